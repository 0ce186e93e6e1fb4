"""SameDiff <-> FlatBuffers ``FlatGraph`` (the ``.fb`` file format).

Reference: nd4j-api autodiff/samediff/serde/FlatBuffersMapper.java, SameDiff.asFlatBuffers / fromFlatBuffers,
schemas libnd4j/include/graph/scheme/{graph,node,variable,array,properties,utils}.fbs (field order below
is the schema's declaration order = vtable slot), ADR 0001. Arrays are stored as FlatArray{shape = nd4j
shapeInfo, buffer = raw little-endian bytes, dtype, byteOrder}.
"""
from __future__ import annotations

import struct
from typing import Dict, List

import numpy as np
import torch

from ..nd4j import serde as nserde
from . import flatbuf as fb
from .flatbuf import S

# DType enum (array.fbs) <-> torch
_DT2ID = {torch.bool: 1, torch.float16: 3, torch.float32: 5, torch.float64: 6, torch.int8: 7, torch.int16: 8, torch.int32: 9,
          torch.int64: 10, torch.uint8: 11, torch.bfloat16: 17}
_ID2DT = {v: k for k, v in _DT2ID.items()}
_ID2DT.update({12: torch.int32, 13: torch.int64, 14: torch.int64})       # unsigned 16/32/64 widen
_ID2NAME = {1: "BOOL", 3: "HALF", 5: "FLOAT", 6: "DOUBLE", 7: "BYTE", 8: "SHORT", 9: "INT", 10: "LONG", 11: "UBYTE", 17: "BFLOAT16"}
_NAME2ID = {v: k for k, v in _ID2NAME.items()}
_NAME2ID.update({"FLOAT32": 5, "FLOAT16": 3, "INT32": 9, "INT64": 10, "FLOAT64": 6, "INT8": 7, "INT16": 8, "UINT8": 11})
_VT = {"VARIABLE": 0, "CONSTANT": 1, "ARRAY": 2, "PLACEHOLDER": 3}
_VT_INV = {v: k for k, v in _VT.items()}


# ----------------------------------------------------------------------------- arrays
def flat_array(t: torch.Tensor) -> S.Tbl:
    t = t.detach().cpu().contiguous()
    name = nserde._TORCH2NAME[t.dtype]
    si = nserde.shape_info(list(t.shape), "c", name, empty=(t.numel() == 0))
    if t.dtype == torch.bfloat16:
        raw = t.view(torch.int16).numpy().tobytes()
    elif t.dtype == torch.bool:
        raw = t.to(torch.uint8).numpy().tobytes()
    else:
        raw = t.numpy().tobytes()
    return S.Tbl({0: S.Vec("long", si), 1: S.Bytes(raw), 2: S.Scalar("byte", _DT2ID[t.dtype]), 3: S.Scalar("byte", 0)})


def read_flat_array(a: fb.Table) -> torch.Tensor:
    si = a.scalar_vector(0, "long")
    raw = a.bytes_vector(1)
    dtid = a.scalar(2, "byte")
    if dtid >= 50:
        return None            # UTF8/16/32 string arrays (assert messages etc.) carry no numeric payload
    dt = _ID2DT[dtid]
    be = a.scalar(3, "byte") == 1
    shape, strides, order, _, empty = nserde.parse_shape_info(si) if si else ([], [], "c", None, False)
    npdt = {torch.bool: "u1", torch.float16: "f2", torch.float32: "f4", torch.float64: "f8", torch.int8: "i1", torch.int16: "i2",
            torch.int32: "i4", torch.int64: "i8", torch.uint8: "u1", torch.bfloat16: "i2"}[dt]
    arr = np.frombuffer(raw, dtype=np.dtype((">" if be else "<") + npdt))
    arr = arr.astype(arr.dtype.newbyteorder("="), copy=True)
    t = torch.from_numpy(arr)
    if dt == torch.bfloat16:
        t = t.view(torch.bfloat16)
    elif dt == torch.bool:
        t = t.bool()
    if empty:
        return t.reshape([0] * max(len(shape), 1))
    n = 1
    for s in shape:
        n *= s
    t = t[:n]
    if order == "f" and len(shape) > 1:
        return t.reshape(list(reversed(shape))).permute(*reversed(range(len(shape)))).contiguous()
    return t.reshape(shape)


def _pair(a, b):
    return S.Tbl({0: S.Scalar("int", a), 1: S.Scalar("int", b)})


# ----------------------------------------------------------------------------- write
def to_flatbuffers(sd, saveUpdaterState=True) -> bytes:
    from .samediff import OpType
    ids: Dict[str, tuple] = {}
    counter = 0
    op_id: Dict[str, int] = {}
    for on, o in sd.ops.items():
        counter += 1
        op_id[on] = counter
        for k, out in enumerate(o.outputs):
            ids[out] = (counter, k)
    for n, v in sd.variables.items():
        if n not in ids:
            counter += 1
            ids[n] = (counter, 0)
    fvars = []
    for n, v in sd.variables.items():
        f = {0: _pair(*ids[n]), 1: S.Str(n), 2: S.Scalar("byte", _DT2ID.get(v.dtype, 5) if v.dtype is not None else 5),
             5: S.Scalar("int", -1), 6: S.Scalar("byte", _VT[v.variableType])}
        arr = sd._arrays.get(n)
        if arr is not None and v.variableType in ("VARIABLE", "CONSTANT"):
            f[4] = flat_array(arr)
        elif v.shape is not None:
            f[3] = S.Vec("long", [int(s) if s is not None else -1 for s in v.shape])
        fvars.append(S.Tbl(f))
    fnodes = []
    for on, o in sd.ops.items():
        props = []
        if o.frameName:
            props.append(S.Tbl({0: S.Str("frameName"), 6: S.Vec("string", [o.frameName])}))
        if o.isConstantEnter:
            props.append(S.Tbl({0: S.Str("isConstant"), 5: S.Vec("bool", [True])}))
        f = {0: S.Scalar("int", op_id[on]), 1: S.Str(on), 2: S.Scalar("byte", o.opType), 3: S.Scalar("long", int(o.opNum)),
             6: S.Vec("table", [_pair(*ids[i]) for i in o.inputs]),
             8: S.Vec("double", [float(x) for x in o.tArgs]) if o.tArgs else None,
             9: S.Vec("long", [int(x) for x in o.iArgs]) if o.iArgs else None,
             10: S.Vec("bool", [bool(x) for x in o.bArgs]) if o.bArgs else None,
             11: S.Vec("int", [int(x) for x in o.dimensions]) if o.dimensions else None,
             12: S.Scalar("int", -1), 15: S.Vec("string", list(o.outputs)), 16: S.Str(o.opName),
             19: S.Vec("string", list(o.controlDeps)) if o.controlDeps else None,
             22: S.Vec("byte", [_NAME2ID.get(str(x).upper(), 5) for x in o.dArgs]) if o.dArgs else None,
             23: S.Vec("string", [str(x) for x in o.sArgs]) if o.sArgs else None}
        if props:
            f[4] = S.Vec("table", props)
        if o.scalar is not None:
            f[18] = flat_array(torch.as_tensor(o.scalar, dtype=torch.float64 if isinstance(o.scalar, float) else None))
        fnodes.append(S.Tbl(f))
    outs = [_pair(*ids[n]) for n in sd.outputs() if n in ids]
    g = {0: S.Scalar("long", 0), 1: S.Vec("table", fvars), 2: S.Vec("table", fnodes), 3: S.Vec("table", outs),
         5: S.Vec("string", sd.inputs()), 6: S.Vec("string", list(sd.lossVariables))}
    if sd.trainingConfig is not None:
        g[7] = S.Str(sd.trainingConfig.toJson())
    if saveUpdaterState and sd.updaterState:
        us = []
        for n, st in sd.updaterState.items():
            us.append(S.Tbl({0: S.Str(n), 1: S.Vec("string", ["STATE"]), 2: S.Vec("table", [flat_array(st)])}))
        g[8] = S.Vec("table", us)
    return fb.Writer().finish(S.Tbl(g))


# ----------------------------------------------------------------------------- read
def from_flatbuffers(buf: bytes, loadUpdaterState=True, device=None):
    from .samediff import SameDiff, SameDiffOp, TrainingConfig, OpType, LOGIC
    g = fb.root(buf)
    sd = SameDiff(device)
    id2name: Dict[tuple, str] = {}
    for v in g.table_vector(1):
        name = v.string(1)
        idp = v.table(0)
        vid = (idp.scalar(0, "int"), idp.scalar(1, "int")) if idp is not None else None
        vt = _VT_INV.get(v.scalar(6, "byte"), "VARIABLE")
        dt = _ID2DT.get(v.scalar(2, "byte"))
        arr_t = v.table(4)
        arr = read_flat_array(arr_t) if arr_t is not None else None
        arr = arr.to(sd.device) if arr is not None else None
        shape = v.scalar_vector(3, "long") if v.has(3) else (list(arr.shape) if arr is not None else None)
        sd._add_var(name, vt, arr.dtype if arr is not None else dt, shape, arr)
        if vid is not None:
            id2name[vid] = name
    for n in g.table_vector(2):
        nid = n.scalar(0, "int")
        name = n.string(1) or "op_%d" % nid
        optype = n.scalar(2, "byte")
        optype = optype if optype >= 0 else optype + 256
        opnum = n.scalar(3, "long")
        opname = n.string(16)
        if opname is None:
            from ..nd4j.ops.custom import name_by_hash
            opname = name_by_hash(opnum) if optype == OpType.CUSTOM else (LOGIC.get(int(opnum)) if optype == OpType.LOGIC else None)
            opname = opname or "unknown_%d" % opnum
        ins = []
        for p in n.table_vector(6):
            key = (p.scalar(0, "int"), p.scalar(1, "int"))
            ins.append(id2name.get(key) or id2name.get((key[0], 0)))
        if not ins and n.has(5):
            ins = [id2name.get((k, 0)) for k in n.scalar_vector(5, "int")]
        outs = n.string_vector(15)
        if not outs:
            outs = [id2name.get((nid, 0), name)]
        o = SameDiffOp(name, opname, [i for i in ins if i is not None], outs, n.scalar_vector(9, "long"), n.scalar_vector(8, "double"),
                       n.scalar_vector(10, "bool"), [_ID2NAME.get(x, "FLOAT") for x in n.scalar_vector(22, "byte")], optype, opnum,
                       n.scalar_vector(11, "int"), None, n.string_vector(23))
        sc = n.table(18)
        if sc is not None:
            t = read_flat_array(sc)
            o.scalar = float(t.reshape(-1)[0]) if t.numel() else None
        for pr in n.table_vector(4):
            pname = pr.string(0)
            if pname in ("frameName", "frame_name"):
                ss = pr.string_vector(6)
                o.frameName = ss[0] if ss else None
            elif pname in ("isConstant", "is_constant"):
                bb = pr.scalar_vector(5, "bool")
                o.isConstantEnter = bool(bb[0]) if bb else False
        o.controlDeps = n.string_vector(19)
        sd._add_op(o)
    sd.lossVariables = g.string_vector(6)
    tc = g.string(7)
    if tc:
        try:
            sd.trainingConfig = TrainingConfig.fromJson(tc)
        except Exception:
            sd.trainingConfig = None         # foreign (Java) training-config JSON: keep the graph usable
    if loadUpdaterState:
        for us in g.table_vector(8):
            vals = us.table_vector(2)
            if vals:
                sd.updaterState[us.string(0)] = read_flat_array(vals[0]).to(sd.device)
    return sd
