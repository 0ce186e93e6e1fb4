"""Native graph runtime front-end (reference: libnd4j/include/graph/** — Graph, GraphExecutioner,
GraphHolder, VariableSpace, FlowPath/FrameState, Logic*.cpp, GraphProfile/NodeProfile; C ABI
`executeFlatGraph` NativeOps.h:1474, `registerGraph/executeStoredGraph/unregisterGraph` :1525-1543;
Java entry OpExecutioner.registerGraph/executeGraph).

The C++ runtime (native/csrc/rt_graph.cpp) parses the FlatGraph, layers it topologically, keeps it in
the GraphHolder registry and owns the schedule including TF-style control flow (frames / iterations,
Switch / Merge / Enter / Exit / NextIteration / LoopCond). Tensors stay on the GPU: the runtime only
sees 64-bit handles and calls back into the op library (HIP kernels on the current stream) per node.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from ..native import lib as L

_EXEC_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int64), C.c_int)
_PRED_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64)
_declared = False


def _rt():
    global _declared
    lib = L.rt()
    if not _declared:
        i32, i64, vp, cp = C.c_int, C.c_int64, C.c_void_p, C.c_char_p
        P64, PI = C.POINTER(C.c_int64), C.POINTER(C.c_int)
        sig = {
            "graph_register": (i32, [i64, vp, i64]), "graph_unregister": (i32, [i64]), "graph_registered": (i32, [i64]),
            "graph_count": (i32, []), "graph_num_nodes": (i32, [i64]), "graph_num_vars": (i32, [i64]),
            "graph_num_layers": (i32, [i64]), "graph_layer_size": (i32, [i64, i32]), "graph_layer_nodes": (i32, [i64, i32, PI, i32]),
            "graph_var_name": (cp, [i64, i32]), "graph_var_index": (i32, [i64, cp]), "graph_var_info": (i32, [i64, i32, P64]),
            "graph_var_shape": (i32, [i64, i32, P64, i32]), "graph_var_shape_info": (i32, [i64, i32, P64, i32]),
            "graph_var_data": (vp, [i64, i32, P64]),
            "graph_node_name": (cp, [i64, i32]), "graph_node_op": (cp, [i64, i32]), "graph_node_frame": (cp, [i64, i32]),
            "graph_node_info": (i32, [i64, i32, P64]), "graph_node_io": (i32, [i64, i32, PI, PI]),
            "graph_node_args": (i32, [i64, i32, C.POINTER(C.c_double), P64, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), PI,
                                      C.POINTER(C.c_double)]),
            "graph_node_sarg": (cp, [i64, i32, i32]),
            "graph_num_outputs": (i32, [i64]), "graph_output_var": (i32, [i64, i32]),
            "graph_num_placeholders": (i32, [i64]), "graph_placeholder": (cp, [i64, i32]), "graph_training_config": (cp, [i64]),
            "graph_execute": (i32, [i64, PI, P64, i32, PI, i32, P64, vp, vp, vp, i32]),
            "graph_node_profile": (i32, [i64, i32, P64]), "graph_profile": (i32, [i64, P64]), "graph_profile_reset": (i32, [i64]),
        }
        for n, (res, args) in sig.items():
            fn = getattr(lib, n)
            fn.restype, fn.argtypes = res, args
        _declared = True
    return lib


class NativeNode:
    __slots__ = ("index", "name", "opName", "opType", "opNum", "layer", "kind", "inputs", "outputs", "tArgs", "iArgs", "bArgs", "dArgs",
                 "dimensions", "sArgs", "scalar", "frameName")

    def __repr__(self):
        return "NativeNode(%s: %s layer=%d)" % (self.name, self.opName, self.layer)


class NativeGraph:
    """A graph registered in the native GraphHolder."""

    def __init__(self, graph_id: int, device=None):
        self.id = int(graph_id)
        self.device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        lib = _rt()
        if not lib.graph_registered(self.id):
            raise KeyError("graph %d is not registered" % self.id)
        self._lib = lib
        self._nodes: Optional[List[NativeNode]] = None
        self._var_arrays: Dict[int, torch.Tensor] = {}
        self._var_names = [lib.graph_var_name(self.id, v).decode() for v in range(lib.graph_num_vars(self.id))]
        self._load_arrays()

    # ------------------------------------------------------------------ structure
    def numNodes(self):
        return self._lib.graph_num_nodes(self.id)

    def numVariables(self):
        return len(self._var_names)

    def variableNames(self):
        return list(self._var_names)

    def layers(self) -> List[List[int]]:
        out = []
        for l in range(self._lib.graph_num_layers(self.id)):
            n = self._lib.graph_layer_size(self.id, l)
            buf = (C.c_int * max(n, 1))()
            self._lib.graph_layer_nodes(self.id, l, buf, n)
            out.append(list(buf[:n]))
        return out

    def placeholders(self):
        return [self._lib.graph_placeholder(self.id, k).decode() for k in range(self._lib.graph_num_placeholders(self.id))]

    def outputs(self):
        return [self._var_names[self._lib.graph_output_var(self.id, k)] for k in range(self._lib.graph_num_outputs(self.id))]

    def nodes(self) -> List[NativeNode]:
        if self._nodes is None:
            lib, gid = self._lib, self.id
            self._nodes = []
            info = (C.c_int64 * 14)()
            for n in range(lib.graph_num_nodes(gid)):
                lib.graph_node_info(gid, n, info)
                nd = NativeNode()
                nd.index = n
                nd.name = lib.graph_node_name(gid, n).decode()
                nd.opName = lib.graph_node_op(gid, n).decode()
                nd.frameName = lib.graph_node_frame(gid, n).decode() or None
                nd.opType, nd.opNum, nd.layer, nd.kind = int(info[0]), int(info[1]), int(info[2]), int(info[3])
                n_in, n_out, nT, nI, nB, nD, nDims, nS = (int(info[k]) for k in range(4, 12))
                ins, outs = (C.c_int * max(n_in, 1))(), (C.c_int * max(n_out, 1))()
                lib.graph_node_io(gid, n, ins, outs)
                nd.inputs = [self._var_names[i] for i in ins[:n_in]]
                nd.outputs = [self._var_names[i] for i in outs[:n_out]]
                t, i = (C.c_double * max(nT, 1))(), (C.c_int64 * max(nI, 1))()
                b, d = (C.c_uint8 * max(nB, 1))(), (C.c_uint8 * max(nD, 1))()
                dims, sc = (C.c_int * max(nDims, 1))(), C.c_double(0)
                lib.graph_node_args(gid, n, t, i, b, d, dims, C.byref(sc))
                nd.tArgs, nd.iArgs = list(t[:nT]), list(i[:nI])
                nd.bArgs, nd.dArgs = [bool(x) for x in b[:nB]], list(d[:nD])
                nd.dimensions = list(dims[:nDims])
                nd.sArgs = [lib.graph_node_sarg(gid, n, k).decode() for k in range(nS)]
                nd.scalar = float(sc.value) if info[12] else None
                self._nodes.append(nd)
        return self._nodes

    def _load_arrays(self):
        """Wrap the FlatArray payloads of VARIABLE / CONSTANT nodes (zero-copy view of the registered buffer → one upload)."""
        from . import serde as S
        lib, gid = self._lib, self.id
        info = (C.c_int64 * 7)()
        for v in range(len(self._var_names)):
            lib.graph_var_info(gid, v, info)
            if not info[4]:
                continue
            dtid = int(info[5])
            if dtid >= 50 or dtid not in S._ID2DT:
                continue
            si = (C.c_int64 * 70)()
            nsi = lib.graph_var_shape_info(gid, v, si, 70)
            ln = C.c_int64(0)
            p = lib.graph_var_data(gid, v, C.byref(ln))
            dt = S._ID2DT[dtid]
            npdt = {torch.bool: "u1", torch.float16: "f2", torch.float32: "f4", torch.float64: "f8", torch.int8: "i1", torch.int16: "i2",
                    torch.int32: "i4", torch.int64: "i8", torch.uint8: "u1", torch.bfloat16: "i2"}[dt]
            from ..nd4j import serde as nserde
            shape, _, order, _, empty = nserde.parse_shape_info(list(si[:nsi])) if nsi > 0 else ([], [], "c", None, False)
            if empty or not p or ln.value == 0:
                t = torch.zeros([0] * max(len(shape), 1), dtype=dt)
            else:
                raw = (C.c_uint8 * ln.value).from_address(p)
                arr = np.frombuffer(raw, dtype=np.dtype((">" if info[6] == 1 else "<") + npdt))
                arr = arr.astype(arr.dtype.newbyteorder("="), copy=True)
                t = torch.from_numpy(arr)
                if dt == torch.bfloat16:
                    t = t.view(torch.bfloat16)
                elif dt == torch.bool:
                    t = t.bool()
                n = int(np.prod(shape)) if shape else 1
                t = t[:n]
                if order == "f" and len(shape) > 1:
                    t = t.reshape(list(reversed(shape))).permute(*reversed(range(len(shape)))).contiguous()
                else:
                    t = t.reshape(shape)
            self._var_arrays[v] = t.to(self.device)

    def getVariableArray(self, name):
        return self._var_arrays.get(self._lib.graph_var_index(self.id, name.encode()))

    def setVariableArray(self, name, arr):
        v = self._lib.graph_var_index(self.id, name.encode())
        if v < 0:
            raise KeyError(name)
        self._var_arrays[v] = torch.as_tensor(getattr(arr, "_t", arr)).to(self.device)

    # ------------------------------------------------------------------ execution
    def execute(self, placeholders: Optional[Dict[str, torch.Tensor]] = None, outputs: Optional[Sequence[str]] = None,
                maxIterations: int = 100000) -> Dict[str, torch.Tensor]:
        from .samediff import SameDiff, SameDiffOp
        from . import serde as S
        lib, gid = self._lib, self.id
        nodes = self.nodes()
        outputs = list(outputs) if outputs else self.outputs()
        if not outputs:
            # FlatGraph without declared outputs: every variable nobody consumes
            consumed = {i for nd in nodes for i in nd.inputs}
            outputs = [o for nd in nodes for o in nd.outputs if o not in consumed]
        handles: Dict[int, torch.Tensor] = {}
        next_h = [1]

        def new_handle(t):
            h = next_h[0]
            next_h[0] += 1
            handles[h] = t
            return h

        feeds_v, feeds_h = [], []
        for v, t in self._var_arrays.items():
            feeds_v.append(v)
            feeds_h.append(new_handle(t))
        for name, t in (placeholders or {}).items():
            v = lib.graph_var_index(gid, str(name).encode())
            if v < 0:
                raise KeyError("unknown placeholder '%s'" % name)
            t = torch.as_tensor(getattr(t, "_t", t)).to(self.device)
            feeds_v.append(v)
            feeds_h.append(new_handle(t))

        sd = SameDiff(self.device)          # op dispatch only (legacy / custom executioners)
        op_cache: Dict[int, SameDiffOp] = {}
        err: List[BaseException] = []

        def exec_cb(_user, ni, ins, n_in, outs, n_out):
            try:
                nd = nodes[ni]
                o = op_cache.get(ni)
                if o is None:
                    o = SameDiffOp(nd.name, nd.opName, nd.inputs, nd.outputs, nd.iArgs, nd.tArgs, nd.bArgs,
                                   [S._ID2NAME.get(x, "FLOAT") for x in nd.dArgs], nd.opType, nd.opNum, nd.dimensions, None, nd.sArgs)
                    o.scalar = nd.scalar
                    op_cache[ni] = o
                with torch.no_grad():
                    res = sd._exec_op(o, [handles[ins[k]] for k in range(n_in)])
                for k in range(min(n_out, len(res))):
                    if res[k] is not None:
                        outs[k] = new_handle(res[k])
                return 0
            except BaseException as e:          # never unwind through the C++ frames
                err.append(e)
                return 1

        def pred_cb(_user, h):
            try:
                t = handles[h]
                return int(bool(t.reshape(-1)[0] != 0)) if t.numel() else 0
            except BaseException as e:
                err.append(e)
                return -1

        want_v = []
        for n in outputs:
            v = lib.graph_var_index(gid, n.encode())
            if v < 0:
                raise KeyError("unknown output '%s'" % n)
            want_v.append(v)
        ecb, pcb = _EXEC_CB(exec_cb), _PRED_CB(pred_cb)
        out_h = (C.c_int64 * max(len(want_v), 1))()
        rc = lib.graph_execute(gid, (C.c_int * max(len(feeds_v), 1))(*feeds_v), (C.c_int64 * max(len(feeds_h), 1))(*feeds_h), len(feeds_v),
                               (C.c_int * max(len(want_v), 1))(*want_v), len(want_v), out_h, C.cast(ecb, C.c_void_p),
                               C.cast(pcb, C.c_void_p), None, int(maxIterations))
        if err:
            raise err[0]
        L.rt_check(rc, "graph_execute")
        res = {}
        for n, h in zip(outputs, out_h):
            if h < 0:
                raise RuntimeError("output '%s' was not produced (dead branch or missing input)" % n)
            res[n] = handles[h]
        return res

    # ------------------------------------------------------------------ profiling
    def profile(self):
        """{'executions', 'total_ms', 'nodes': [(name, op, count, ms)]} — GraphProfile / NodeProfile."""
        lib, gid = self._lib, self.id
        g = (C.c_int64 * 2)()
        lib.graph_profile(gid, g)
        rows = []
        p = (C.c_int64 * 2)()
        for nd in self.nodes():
            lib.graph_node_profile(gid, nd.index, p)
            if p[0]:
                rows.append((nd.name, nd.opName, int(p[0]), p[1] / 1e6))
        return {"executions": int(g[0]), "total_ms": g[1] / 1e6, "nodes": sorted(rows, key=lambda r: -r[3])}

    def resetProfile(self):
        self._lib.graph_profile_reset(self.id)


class NativeGraphExecutioner:
    """`OpExecutioner.registerGraph / executeGraph / forgetGraph` + `executeFlatGraph` over the native GraphHolder."""

    @staticmethod
    def registerGraph(graph_id: int, graph, device=None) -> NativeGraph:
        buf = graph if isinstance(graph, (bytes, bytearray)) else graph.asFlatBuffers(False)
        lib = _rt()
        cb = (C.c_uint8 * len(buf)).from_buffer_copy(bytes(buf))
        L.rt_check(lib.graph_register(int(graph_id), cb, len(buf)), "graph_register")
        dev = device if device is not None else getattr(graph, "device", None)
        return NativeGraph(graph_id, dev)

    @staticmethod
    def executeStoredGraph(graph_id: int, placeholders=None, outputs=None, device=None):
        return NativeGraph(graph_id, device).execute(placeholders, outputs)

    executeGraph = executeStoredGraph

    @staticmethod
    def unregisterGraph(graph_id: int):
        _rt().graph_unregister(int(graph_id))

    forgetGraph = unregisterGraph

    @staticmethod
    def executeFlatGraph(buf: bytes, placeholders=None, outputs=None, device=None):
        """One-shot: register under a scratch id, run, unregister (GraphExecutioner::executeFlatGraph)."""
        gid = -abs(hash(bytes(buf[:64]))) - 1
        g = NativeGraphExecutioner.registerGraph(gid, bytes(buf), device)
        try:
            return g.execute(placeholders, outputs)
        finally:
            NativeGraphExecutioner.unregisterGraph(gid)
