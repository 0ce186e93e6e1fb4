"""Differentiable y = x . W (+ b) for SameDiff / op-level matmul on the in-tree MFMA kernels (forward AND backward), so no
rocBLAS / hipBLASLt kernel appears in an imported BERT step:

    forward   y  = gemm_nt(x, W^T, bias)                 W [K, N] as declared by xw_plus_b / matmul; W^T is a small copy (or the
                                                         transposed replica kept by SameDiff's mixed-precision state)
    backward  dx = gemm_nt(dy, W)                        W [K, N] IS the "NT" B operand of this product: no copy
              dW = split-K weight-gradient kernel: x^T . dy -> [K, N] fp32
              db = column sums of dy (fused with the activation derivative: native/csrc/fused_eltwise.hip)

Reference: ops/declarable/generic/nn/xw_plus_b.cpp (+ xw_plus_b_bp), generic/blas/matmul.cpp, helpers/cuda_off/MmulHelper.cu.
"""
from __future__ import annotations

import torch

from . import gemm as G
from . import dense as DENSE
from . import streams as STREAMS


def usable(x, w):
    """x [..., K], w [K, N] on the GPU, one of bf16 / fp16 / fp32, K and N multiples of the 16-byte chunk."""
    if not (isinstance(x, torch.Tensor) and isinstance(w, torch.Tensor) and x.is_cuda and w.is_cuda and x.dtype == w.dtype
            and x.dtype in (torch.bfloat16, torch.float16, torch.float32) and w.dim() == 2 and x.dim() >= 2):
        return False
    ch = G.chunk(x.dtype)
    K, N = w.shape
    return x.shape[-1] == K and K % ch == 0 and N % ch == 0 and x.numel() // K >= 16


def linear(x, w, bias=None):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, w, bias)):
        return FusedDenseFn.apply(x, w, bias, None, 0)
    x2 = x.reshape(-1, x.shape[-1])
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    y = G.gemm_nt(x2, w.t().contiguous(), bias=None if bias is None else bias.reshape(-1).float())
    return y.view(*x.shape[:-1], w.shape[1])


# ----------------------------------------------------------------------------- fused dense op for mixed-precision training graphs
# ``fused_dense`` = act(x . W + b) (+ residual): what autodiff/optimize.py LinearFusion rewrites xw_plus_b -> activation / residual-add
# chains to. Besides the GEMM epilogues it removes the per-variable glue of a mixed-precision SameDiff step: during a training
# iteration SameDiff registers, for every compute-precision leaf that is a view of the bf16 shadow of its flat fp32 variable
# vector, a Link to (a) the fp32 master view (bias / gain operands of the kernels: no cast launch), (b) the fp32 view of the flat
# gradient vector (weight / bias gradients are accumulated THERE by the split-K and column-sum kernels: no zero-fill, transpose or
# cast launch per variable) and (c) a transposed replica written by the same one-launch cast table that refreshes the shadow
# (the NT GEMM's B operand: no transpose launch per linear op).
ACT_IDS = {None: 0, "identity": 0, "relu": 1, "gelu_tanh": 2, "precise_gelu": 2, "gelu_erf": 3, "tanh": 4, "sigmoid": 5, "gelu": 6,
           "swish": 7}
_EPI_ACT = {0: None, 1: "relu", 2: "gelu_tanh", 3: "gelu_erf", 4: "tanh", 5: "sigmoid"}     # ids the GEMM epilogue knows


class Link:
    __slots__ = ("master", "grad", "wt", "numel", "dtype", "hits")

    def __init__(self, master, grad, wt, numel, dtype):
        self.master, self.grad, self.wt, self.numel, self.dtype, self.hits = master, grad, wt, numel, dtype, 0


_LINKS = None


def set_links(links):
    """Activate (dict data_ptr -> Link) or clear (None) the master links for the current training step."""
    global _LINKS
    _LINKS = links
    if links:
        for l in links.values():
            l.hits = 0


def link_of(t):
    if _LINKS is None or t is None:
        return None
    l = _LINKS.get(t.data_ptr())
    return l if (l is not None and l.numel == t.numel() and l.dtype == t.dtype) else None


def f32_operand(t):
    """fp32 image of a small parameter operand (bias, gain): the master view when linked, else a cast."""
    if t is None:
        return None
    l = link_of(t)
    if l is not None:
        return l.master.reshape(-1)
    t = t.reshape(-1)
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _ew_ok(t):
    return t.is_cuda and t.dtype in G._DT and t.is_contiguous() and t.data_ptr() % 16 == 0 and t.shape[-1] % 8 == 0


def act_fwd(z, act_id):
    from ..native import lib as L
    y = torch.empty_like(z)
    L.check(L.hip().dl4j_act_fwd(L.ptr(z), L.ptr(y), L.c_int64(z.numel()), L.c_int(act_id), L.c_int(G._DT[z.dtype]), L.stream()),
            "act_fwd")
    return y


def act_bwd_colsum(z, dy, act_id, db=None, want_dz=True):
    """dz = dy * act'(z) (z None: dz is dy and nothing is written); db (fp32 [N], accumulated into) += column sums of dz."""
    from ..native import lib as L
    M, N = dy.shape
    dz = torch.empty_like(dy) if (want_dz and z is not None) else None
    L.check(L.hip().dl4j_act_bwd_colsum(L.ptr(z), L.ptr(dy), L.ptr(dz), L.ptr(db), L.c_int64(M), L.c_int(N), L.c_int(act_id),
                                        L.c_int(G._DT[dy.dtype]), L.stream()), "act_bwd_colsum")
    return dz if dz is not None else dy


class ActFn(torch.autograd.Function):
    """Element-wise activation, one HIP pass in each direction (forward y = act(x); backward dx = dy * act'(x))."""

    @staticmethod
    def forward(ctx, x, act_id):
        xc = x if x.is_contiguous() else x.contiguous()
        ctx.save_for_backward(xc)
        ctx.act_id = act_id
        return act_fwd(xc, act_id)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        d2 = dy.to(x.dtype).contiguous().view(-1, x.shape[-1])
        return act_bwd_colsum(x.view(-1, x.shape[-1]), d2, ctx.act_id, None).view(x.shape), None


def activation(x, name):
    """Fused activation when the tensor qualifies (GPU, 16-byte rows), else None (caller keeps its generic path)."""
    a = ACT_IDS.get(name)
    if a is None or not isinstance(x, torch.Tensor) or x.dim() < 1 or not _ew_ok(x):
        return None
    if torch.is_grad_enabled() and x.requires_grad:
        return ActFn.apply(x, a)
    return act_fwd(x, a)


FUSE_DB = __import__("os").environ.get("DL4J_AMD_DENSE_FUSE_DB", "1") == "1"


class FusedDenseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, residual, act_id):
        K, N = w.shape
        x2 = x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        lw = link_of(w)
        wt = lw.wt if (lw is not None and lw.wt is not None) else w.t().contiguous()
        b32 = f32_operand(bias)
        r2 = None
        if residual is not None:
            r2 = residual.reshape(-1, N)
            if not r2.is_contiguous():
                r2 = r2.contiguous()
        z = None
        if act_id and (any(ctx.needs_input_grad[:3]) or act_id not in _EPI_ACT):      # the backward needs the pre-activation
            z = G.gemm_nt(x2, wt, bias=b32)
            y = act_fwd(z, act_id)
            if r2 is not None:
                y = y + r2
        else:
            y = G.gemm_nt(x2, wt, bias=b32, act=_EPI_ACT.get(act_id), residual=r2)
        ctx.save_for_backward(x2, w, z)
        ctx.act_id = act_id
        ctx.bias_meta = None if bias is None else (bias.dtype, bias.shape, link_of(bias))
        ctx.wlink = lw
        ctx.xshape = x.shape
        ctx.rshape = None if residual is None else residual.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w, z = ctx.saved_tensors
        K, N = w.shape
        d2 = dy.reshape(-1, N)
        if not d2.is_contiguous() or d2.dtype != x2.dtype:
            d2 = d2.to(x2.dtype).contiguous()
        dres = dy.reshape(ctx.rshape) if (ctx.rshape is not None and ctx.needs_input_grad[3]) else None
        dx = dw = db = None
        want_b = ctx.bias_meta is not None and ctx.needs_input_grad[2]
        lw = ctx.wlink
        # linear layer whose dW goes straight into the flat gradient vector on the split-K kernel: that launch also sums the columns of dy
        # into the bias gradient (no separate bias-gradient pass)
        db_fused = None
        if (want_b and z is None and FUSE_DB and ctx.bias_meta[2] is not None and ctx.needs_input_grad[1] and lw is not None and lw.hits == 0
                and not STREAMS.dense_active() and d2.is_cuda and DENSE.dw_ok(x2, d2, lw.grad.view(K, N)) and DENSE.dw_takes_db(x2, d2)
                and ctx.bias_meta[2].grad.dtype == torch.float32 and ctx.bias_meta[2].grad.is_contiguous()):
            db_fused = ctx.bias_meta[2].grad.reshape(-1)
            ctx.bias_meta[2].hits += 1
        if (want_b and db_fused is None) or z is not None:
            tgt = None
            if want_b:
                bl = ctx.bias_meta[2]
                if bl is not None:
                    tgt = bl.grad.reshape(-1)
                    bl.hits += 1
                else:
                    tgt = torch.zeros(N, device=d2.device, dtype=torch.float32)
                    db = tgt
            d2 = act_bwd_colsum(z, d2, ctx.act_id, tgt)
            if db is not None:
                db = db.to(ctx.bias_meta[0]).reshape(ctx.bias_meta[1])
        direct = ctx.needs_input_grad[1] and lw is not None and lw.hits == 0 and DENSE.dw_ok(x2, d2, lw.grad.view(K, N))
        side = direct and STREAMS.dense_active()
        if side:
            # dW is off the critical path: issued first, on the side stream, so that it overlaps dx and what follows (ops/streams.py)
            lw.hits += 1
            with STREAMS.side_region(STREAMS.fork(x2, d2), 'dense weight-gradient'):
                DENSE.dw(x2, d2, lw.grad.view(K, N), True)
        if ctx.needs_input_grad[0]:
            wc = w if w.is_contiguous() else w.contiguous()
            dx = G.gemm_nt(d2, wc).view(ctx.xshape)
        if ctx.needs_input_grad[1] and not side:
            if direct:
                lw.hits += 1
                # x^T . dy -> [K, N] fp32, straight into the flat gradient vector (+ the bias gradient when fused)
                DENSE.dw(x2, d2, lw.grad.view(K, N), True, db_out=db_fused, db_of_x=db_fused is not None)
            else:
                if lw is not None:
                    STREAMS.join()                                 # an earlier use of this weight may still be accumulating on the side stream
                t = torch.zeros(K, N, device=w.device, dtype=torch.float32)
                if DENSE.dw_ok(x2, d2, t):
                    DENSE.dw(x2, d2, t, True)
                else:
                    t = x2.float().t() @ d2.float()
                if lw is not None:
                    lw.grad.view(K, N).add_(t)
                    lw.hits += 1
                else:
                    dw = t.to(w.dtype)
        return dx, dw, db, dres, None


def fused_dense(x, w, bias=None, residual=None, act=None):
    a = ACT_IDS[act] if not isinstance(act, int) else act
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, w, bias, residual)):
        return FusedDenseFn.apply(x, w, bias, residual, a)
    K, N = w.shape
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    r2 = None if residual is None else residual.reshape(-1, N).contiguous()
    wt = w.t()                                   # [N, K] view of the [K, N] weight: read K-major by the layout-aware kernel, no copy
    mm = (lambda **kw: G.gemm(x2, wt, splits=1, **kw)) if G.supported_ex(x2, wt) else (lambda **kw: G.gemm_nt(x2, wt.contiguous(), **kw))
    if a in _EPI_ACT:
        y = mm(bias=f32_operand(bias), act=_EPI_ACT[a], residual=r2)
    else:
        y = act_fwd(mm(bias=f32_operand(bias)), a)
        if r2 is not None:
            y = y + r2
    return y.view(*x.shape[:-1], N)


def fused_dense_reference(x, w, bias=None, residual=None, act=None):
    """Plain PyTorch math of fused_dense (CPU path / test oracle)."""
    a = ACT_IDS[act] if not isinstance(act, int) else act
    y = torch.matmul(x, w)
    if bias is not None:
        y = y + bias.reshape(-1)
    if a == 1: y = torch.relu(y)
    elif a == 2: y = torch.nn.functional.gelu(y, approximate="tanh")
    elif a == 3: y = torch.nn.functional.gelu(y)
    elif a == 4: y = torch.tanh(y)
    elif a == 5: y = torch.sigmoid(y)
    elif a == 6: y = y * torch.sigmoid(1.702 * y)
    elif a == 7: y = y * torch.sigmoid(y)
    if residual is not None:
        y = y + residual
    return y


class TransposeTable:
    """One-launch refresh of the transposed compute-precision replicas of 2-D weights living in a flat fp32 vector
    (native/csrc/fused_eltwise.hip transpose_cast_table_kernel). entries: (src_off, dst_off, rows, cols)."""

    def __init__(self, entries, device):
        import struct
        import numpy as np
        self.entries = [tuple(int(v) for v in e) for e in entries]
        self.device = torch.device(device)
        raw, seg, tile = b"", [], []
        for k, (so, do, r, c) in enumerate(self.entries):
            tc = (c + 63) // 64
            nt = ((r + 63) // 64) * tc
            raw += struct.pack("<qqiiii", so, do, r, c, tc, 0)
            seg.append(np.full(nt, k, dtype=np.int32))
            tile.append(np.arange(nt, dtype=np.int32))
        self.num_blocks = int(sum(len(s) for s in seg))
        if self.entries and self.device.type == "cuda":
            self._seg = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
            self._blk_seg = torch.from_numpy(np.concatenate(seg)).to(self.device)
            self._blk_tile = torch.from_numpy(np.concatenate(tile)).to(self.device)

    def run(self, src_f32, dst):
        if not self.entries:
            return
        if src_f32.is_cuda:
            from ..native import lib as L
            L.check(L.hip().dl4j_transpose_cast_table(L.ptr(src_f32), L.ptr(dst), L.ptr(self._seg), L.ptr(self._blk_seg),
                                                      L.ptr(self._blk_tile), L.c_int(self.num_blocks), L.c_int(G._DT[dst.dtype]),
                                                      L.stream()), "transpose_cast_table")
            return
        for so, do, r, c in self.entries:
            dst[do:do + r * c] = src_f32[so:so + r * c].view(r, c).t().reshape(-1).to(dst.dtype)
