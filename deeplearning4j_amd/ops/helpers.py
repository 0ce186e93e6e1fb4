"""Python launchers for the declarable-helper HIP kernels (native/csrc/helpers_nn.hip, helpers_data.hip).

Each function validates shapes / dtypes / contiguity on the host before the launch (a faulting kernel can take the
whole GPU down) and returns new tensors. They are registered as ``@platform`` helpers of the matching custom ops in
nd4j/ops/impl_platform.py — the same isUsable-then-invoke rule as the reference's PlatformHelper
(ops/declarable/impl/DeclarableOp.cpp:692-739) — and are called directly by the layer implementations.
Reference kernel families: libnd4j/include/ops/declarable/helpers/cuda/{activations,addBias,dropout,lrn,clip,gather,
scatter,one_hot,prefix,im2col,col2im,convolutions_upsampling2d,merge,segment_sum,image_resize,sg_cb}.cu.
"""
from __future__ import annotations

import ctypes

import torch

from ..native import lib as L

_F = (torch.float32, torch.bfloat16)


def ok(t, dtypes=_F):
    return L.use_hip(t) and t.dtype in dtypes and t.is_contiguous() and t.numel() > 0


def _f32(t):
    """fp32 image of a parameter operand; a compute-precision leaf linked to its fp32 master (ops/linear.py Links) costs no launch."""
    if t is None:
        return None
    from . import linear as LIN
    return LIN.f32_operand(t)


def _idx_ok(i):
    return i.is_cuda and i.dtype in (torch.int32, torch.int64) and i.is_contiguous()


# ----------------------------------------------------------------------------- softmax family (any rank, last axis)
def softmax_last(x, log=False):
    from . import softmax as SM
    return SM.softmax(x.reshape(-1, x.shape[-1]), log).reshape(x.shape)


def softmax_bwd_last(x, g, log=False):
    """dL/dx for y = (log_)softmax(x, -1) given dL/dy = g (y is recomputed from x inside the kernel)."""
    cols = x.shape[-1]
    dx = torch.empty_like(x)
    L.check(L.hip().dl4j_softmax_bwd_x(L.ptr(x), L.ptr(g), L.ptr(dx), L.c_int64(x.numel() // cols), L.c_int(cols),
                                       L.c_int(1 if log else 0), L.c_int(L.dtype_code(x)), L.stream()), "softmax_bwd_x")
    return dx


# ----------------------------------------------------------------------------- layer norm over the last axis
def layer_norm(x, gamma=None, beta=None, eps=1e-5, want_stats=False):
    cols = x.shape[-1]
    rows = x.numel() // cols
    y = torch.empty_like(x)
    mean = rstd = None
    if want_stats:
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    ga, be = _f32(gamma), _f32(beta)
    assert ga is None or ga.numel() == cols
    assert be is None or be.numel() == cols
    L.check(L.hip().dl4j_layer_norm_fwd(L.ptr(x), L.ptr(ga), L.ptr(be), L.ptr(y), L.ptr(mean), L.ptr(rstd), L.c_int64(rows),
                                        L.c_int(cols), L.c_float(eps), L.c_int(L.dtype_code(x)), L.stream()), "layer_norm_fwd")
    return (y, mean, rstd) if want_stats else y


def layer_norm_bwd(x, g, gamma, mean, rstd, want_param_grads=True, sums_out=None, defer_sums=False):
    """Returns (dx, dgamma, dbeta) — dgamma/dbeta fp32 [cols] (None unless want_param_grads). ``sums_out``: fp32 [2 * cols] the
    finishing kernel writes [dgamma | dbeta] into (adjacent views of a flat gradient vector)."""
    cols = x.shape[-1]
    rows = x.numel() // cols
    dx = torch.empty_like(x)
    ga = _f32(gamma)
    partial = sums = None
    gy = int(L.hip().dl4j_layer_norm_bwd_rows(L.c_int64(rows)))
    if want_param_grads:
        partial = torch.empty(gy * 2 * cols, device=x.device, dtype=torch.float32)
        sums = sums_out if sums_out is not None else torch.empty(2 * cols, device=x.device, dtype=torch.float32)
    L.check(L.hip().dl4j_layer_norm_bwd(L.ptr(x), L.ptr(g), L.ptr(ga), L.ptr(mean), L.ptr(rstd), L.ptr(dx), L.ptr(partial),
                                        L.c_int64(rows), L.c_int(cols), L.c_int(L.dtype_code(x)), L.stream()), "layer_norm_bwd")
    if not want_param_grads:
        return dx, None, None
    if defer_sums and sums_out is not None:
        # nobody reads the sums before the optimizer: they ride on a later weight-gradient launch (or the end-of-backward flush)
        from .conv_igemm import PendingSums
        PendingSums.push(partial, gy, cols, sums)
        return dx, None, None
    L.check(L.hip().dl4j_bn_sum_partials(L.ptr(partial), L.c_int(gy), L.c_int(cols), L.ptr(sums), None, None, L.stream()),
            "layer_norm_param_sums")
    return dx, sums[:cols], sums[cols:]


# ----------------------------------------------------------------------------- bias / prelu / lrn / clip / dropout
def _channel_geom(x, channel_axis):
    ax = channel_axis % x.dim()
    inner = 1
    for s in x.shape[ax + 1:]:
        inner *= int(s)
    return int(x.shape[ax]), inner


def bias_add(x, bias, channel_axis=-1):
    C, inner = _channel_geom(x, channel_axis)
    b = _f32(bias)
    assert b.numel() == C
    y = torch.empty_like(x)
    L.check(L.hip().dl4j_bias_add(L.ptr(x), L.ptr(b), L.ptr(y), L.c_int64(x.numel()), L.c_int(C), L.c_int64(inner),
                                  L.c_int(L.dtype_code(x)), L.stream()), "bias_add")
    return y


def prelu(x, alpha, channel_axis=1):
    C, inner = _channel_geom(x, channel_axis)
    a = _f32(alpha)
    assert a.numel() == C
    y = torch.empty_like(x)
    L.check(L.hip().dl4j_prelu(L.ptr(x), L.ptr(a), L.ptr(y), L.c_int64(x.numel()), L.c_int(C), L.c_int64(inner),
                               L.c_int(L.dtype_code(x)), L.stream()), "prelu")
    return y


def lrn_last(x, depth, bias, alpha, beta, want_scale=False):
    """LRN across the last (channel) axis of a contiguous tensor (channels_last activations viewed as [M, C])."""
    C = x.shape[-1]
    M = x.numel() // C
    y = torch.empty_like(x)
    sc = torch.empty(x.shape, device=x.device, dtype=torch.float32) if want_scale else None
    L.check(L.hip().dl4j_lrn_fwd(L.ptr(x), L.ptr(y), L.ptr(sc), L.c_int64(M), L.c_int(C), L.c_int(int(depth)), L.c_float(bias),
                                 L.c_float(alpha), L.c_float(beta), L.c_int(L.dtype_code(x)), L.stream()), "lrn_fwd")
    return (y, sc) if want_scale else y


def clip_by_norm(x, clip, want_norm=False):
    dev = x.device
    ws = torch.empty(int(L.hip().dl4j_clip_ws_floats()), device=dev, dtype=torch.float32)     # stream-ordered scratch
    y = torch.empty_like(x)
    nrm = torch.empty(1, device=dev, dtype=torch.float32) if want_norm else None
    L.check(L.hip().dl4j_clip_by_norm(L.ptr(x), L.ptr(y), L.ptr(ws), L.c_int64(x.numel()), L.c_float(clip), L.ptr(nrm),
                                      L.c_int(L.dtype_code(x)), L.stream()), "clip_by_norm")
    return (y, nrm) if want_norm else y


def dropout(x, keep_prob, seed, stream_id=0, want_mask=True):
    """Inverted dropout with DL4J's retain-probability convention; returns (y, uint8 mask)."""
    y = torch.empty_like(x)
    mask = torch.empty(x.shape, device=x.device, dtype=torch.uint8) if want_mask else None
    L.check(L.hip().dl4j_dropout(L.ptr(x), L.ptr(y), L.ptr(mask), L.c_int64(x.numel()), L.c_float(keep_prob),
                                 ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), ctypes.c_uint64(int(stream_id) & (2 ** 64 - 1)),
                                 L.c_int(L.dtype_code(x)), L.stream()), "dropout")
    return y, mask


def dropout_bwd(g, mask, keep_prob):
    assert mask.dtype == torch.uint8 and mask.numel() == g.numel() and mask.is_contiguous()
    dx = torch.empty_like(g)
    L.check(L.hip().dl4j_dropout_bwd(L.ptr(g), L.ptr(mask), L.ptr(dx), L.c_int64(g.numel()), L.c_float(keep_prob),
                                     L.c_int(L.dtype_code(g)), L.stream()), "dropout_bwd")
    return dx


# ----------------------------------------------------------------------------- indexed
def gather_rows(table, idx):
    """table [V, D] (any 1/2/4/8-byte dtype), idx int32/int64 of any shape -> [*idx.shape, D]."""
    V, D = table.shape
    out = torch.empty(tuple(idx.shape) + (D,), device=table.device, dtype=table.dtype)
    L.check(L.hip().dl4j_gather_rows(L.ptr(table), L.ptr(idx), L.ptr(out), L.c_int64(idx.numel()), L.c_int(D), L.c_int64(V),
                                     L.c_int(table.element_size()), L.c_int(1 if idx.dtype == torch.int64 else 0), L.stream()),
            "gather_rows")
    return out


def scatter_add_rows_(table_f32, idx, src, alpha=1.0):
    """table[idx[r]] += alpha * src[r] in place (fp32 table; duplicate indices accumulate)."""
    V, D = table_f32.shape
    assert table_f32.dtype == torch.float32 and table_f32.is_contiguous()
    assert src.numel() == idx.numel() * D and src.dtype in _F and src.is_contiguous()
    L.check(L.hip().dl4j_scatter_add_rows(L.ptr(src), L.ptr(idx), L.ptr(table_f32), L.c_int64(idx.numel()), L.c_int(D), L.c_int64(V),
                                          L.c_float(alpha), L.c_int(L.dtype_code(src)),
                                          L.c_int(1 if idx.dtype == torch.int64 else 0), L.stream()), "scatter_add_rows")
    return table_f32


def one_hot(idx, depth, on=1.0, off=0.0, dtype=torch.float32):
    out = torch.empty(tuple(idx.shape) + (int(depth),), device=idx.device, dtype=dtype)
    L.check(L.hip().dl4j_one_hot(L.ptr(idx), L.ptr(out), L.c_int64(idx.numel()), L.c_int(int(depth)), L.c_float(on), L.c_float(off),
                                 L.c_int(L.dtype_code(out)), L.c_int(1 if idx.dtype == torch.int64 else 0), L.stream()), "one_hot")
    return out


def segment_sum(x, seg, num_segments):
    """x [rows, ...] -> [num_segments, ...] (fp32 accumulate, cast back); ids outside [0, num_segments) are dropped."""
    rows = x.shape[0]
    D = x.numel() // max(rows, 1)
    out = torch.zeros((int(num_segments),) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
    L.check(L.hip().dl4j_segment_sum(L.ptr(x), L.ptr(seg), L.ptr(out), L.c_int64(rows), L.c_int(D), L.c_int64(int(num_segments)),
                                     L.c_int(L.dtype_code(x)), L.c_int(1 if seg.dtype == torch.int64 else 0), L.stream()),
            "segment_sum")
    return out if x.dtype == torch.float32 else out.to(x.dtype)


def cumsum_last(x, exclusive=False, reverse=False):
    cols = x.shape[-1]
    y = torch.empty_like(x)
    L.check(L.hip().dl4j_cumsum_rows(L.ptr(x), L.ptr(y), L.c_int64(x.numel() // cols), L.c_int(cols), L.c_int(int(exclusive)),
                                     L.c_int(int(reverse)), L.c_int(L.dtype_code(x)), L.stream()), "cumsum_rows")
    return y


def merge(arrays, mode):
    """mode: 'add' | 'avg' | 'max' over same-shape contiguous tensors."""
    a0 = arrays[0]
    for a in arrays:
        assert a.shape == a0.shape and a.dtype == a0.dtype and a.is_contiguous() and a.is_cuda
    ptrs = torch.tensor([a.data_ptr() for a in arrays], dtype=torch.int64).to(a0.device, non_blocking=False)
    out = torch.empty_like(a0)
    L.check(L.hip().dl4j_merge(L.ptr(ptrs), L.c_int(len(arrays)), L.ptr(out), L.c_int64(a0.numel()),
                               L.c_int({"add": 0, "avg": 1, "max": 2}[mode]), L.c_int(L.dtype_code(a0)), L.stream()), "merge")
    return out


# ----------------------------------------------------------------------------- im2col / col2im / upsampling / resize
def _geom(N, C, H, W, kH, kW, sH, sW, pH, pW, dH, dW):
    oH = (H + 2 * pH - dH * (kH - 1) - 1) // sH + 1
    oW = (W + 2 * pW - dW * (kW - 1) - 1) // sW + 1
    return (ctypes.c_int * 14)(N, C, H, W, kH, kW, oH, oW, sH, sW, pH, pW, dH, dW), oH, oW


def im2col(x, kH, kW, sH, sW, pH, pW, dH=1, dW=1, pad_value=0.0):
    N, C, H, W = x.shape
    g, oH, oW = _geom(N, C, H, W, kH, kW, sH, sW, pH, pW, dH, dW)
    assert oH > 0 and oW > 0
    col = torch.empty((N, C, kH, kW, oH, oW), device=x.device, dtype=x.dtype)
    L.check(L.hip().dl4j_im2col(L.ptr(x), L.ptr(col), g, L.c_float(pad_value), L.c_int(L.dtype_code(x)), L.stream()), "im2col")
    return col


def col2im(col, H, W, sH, sW, pH, pW, dH=1, dW=1):
    N, C, kH, kW, oH, oW = col.shape
    g, oH2, oW2 = _geom(N, C, H, W, kH, kW, sH, sW, pH, pW, dH, dW)
    assert (oH2, oW2) == (oH, oW), "col2im: column shape does not match the image geometry"
    x = torch.empty((N, C, H, W), device=col.device, dtype=col.dtype)
    L.check(L.hip().dl4j_col2im(L.ptr(col), L.ptr(x), g, L.c_int(L.dtype_code(col)), L.stream()), "col2im")
    return x


def upsample2d(x, fH, fW):
    N, C, H, W = x.shape
    y = torch.empty((N, C, H * fH, W * fW), device=x.device, dtype=x.dtype)
    L.check(L.hip().dl4j_upsample2d_fwd(L.ptr(x), L.ptr(y), L.c_int64(N * C), L.c_int(H), L.c_int(W), L.c_int(fH), L.c_int(fW),
                                        L.c_int(x.element_size()), L.stream()), "upsample2d_fwd")
    return y


def upsample2d_bwd(g, fH, fW):
    N, C, oH, oW = g.shape
    assert oH % fH == 0 and oW % fW == 0
    H, W = oH // fH, oW // fW
    dx = torch.empty((N, C, H, W), device=g.device, dtype=g.dtype)
    L.check(L.hip().dl4j_upsample2d_bwd(L.ptr(g), L.ptr(dx), L.c_int64(N * C), L.c_int(H), L.c_int(W), L.c_int(fH), L.c_int(fW),
                                        L.c_int(L.dtype_code(g)), L.stream()), "upsample2d_bwd")
    return dx


def resize_nhwc(x, oH, oW, mode="bilinear", align_corners=False, half_pixel=False):
    N, H, W, C = x.shape
    y = torch.empty((N, int(oH), int(oW), C), device=x.device, dtype=x.dtype)
    L.check(L.hip().dl4j_resize_nhwc(L.ptr(x), L.ptr(y), L.c_int(N), L.c_int(H), L.c_int(W), L.c_int(C), L.c_int(int(oH)),
                                     L.c_int(int(oW)), L.c_int(0 if mode == "nearest" else 1), L.c_int(int(align_corners)),
                                     L.c_int(int(half_pixel)), L.c_int(L.dtype_code(x)), L.stream()), "resize_nhwc")
    return y


# ----------------------------------------------------------------------------- word2vec batched step
def sg_pairs_(syn0, syn1, in_idx, tgt_idx, label, lr, hidden=None, scale=1.0, saturation=0):
    """One batched skip-gram / CBOW update: syn1 rows are updated in place, returns dh [n_pairs, D] (the input-side
    gradient of each pair, to be scatter-added into syn0 — or distributed over the context for CBOW). ``saturation``: the reference's
    MAX_EXP = 6 rule, 1 = hierarchic softmax (a saturated node is skipped), 2 = negative sampling (sigmoid clamped to 0 / 1), 0 = none."""
    n = tgt_idx.numel()
    D = syn0.shape[1]
    for t in (syn0, syn1):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
    assert in_idx.dtype == torch.int64 and tgt_idx.dtype == torch.int64 and label.dtype == torch.float32
    assert in_idx.numel() == n and label.numel() == n and in_idx.is_contiguous() and tgt_idx.is_contiguous() and label.is_contiguous()
    if hidden is not None:
        assert hidden.dtype == torch.float32 and hidden.is_contiguous() and hidden.shape == (n, D)
    dh = torch.empty(n, D, device=syn0.device, dtype=torch.float32)
    L.check(L.hip().dl4j_sg_pairs(L.ptr(syn0), L.ptr(syn1), L.ptr(hidden), L.ptr(in_idx), L.ptr(tgt_idx), L.ptr(label), L.ptr(dh),
                                  L.c_int64(n), L.c_int(D), L.c_float(lr), L.c_float(scale), L.c_int64(syn0.shape[0]),
                                  L.c_int64(syn1.shape[0]), L.c_int(int(saturation)), L.stream()), "sg_pairs")
    return dh


# ----------------------------------------------------------------------------- index-map family (native/csrc/index_map.hip)
SHIFT, REVERSE, WRAP, REFLECT, SYMMETRIC, STRIDE = 0, 1, 2, 3, 4, 5


def _pad_bits(value, dtype):
    t = torch.tensor([value], dtype=dtype)
    nbytes = t.element_size()
    raw = t.view(torch.uint8).numpy().tobytes() if dtype != torch.bfloat16 else t.view(torch.int16).numpy().tobytes()
    return int.from_bytes(raw[:nbytes], "little")


def index_map(x, out, maps, out_base=0, pad_value=0.0):
    """out[out_base + sum c_d*out_stride_d] = x[sum f_d(c_d)*in_stride_d] over the region; ``maps`` = list of
    (out_stride, in_stride, region, in_dim, a, b, mode) per dimension. x must be contiguous-or-strided with the given strides
    and every produced offset in range (validated here on the host)."""
    assert x.is_cuda and out.is_cuda and x.dtype == out.dtype and 1 <= len(maps) <= 8
    assert x.element_size() in (1, 2, 4, 8)
    lo = hi = 0
    olo = ohi = int(out_base)
    for (os_, is_, region, in_dim, a, b, mode) in maps:
        assert region >= 0 and in_dim > 0
        if region == 0:
            return out
        if is_ >= 0:
            hi += (in_dim - 1) * is_
        else:
            lo += (in_dim - 1) * is_
        if os_ >= 0:
            ohi += (region - 1) * os_
        else:
            olo += (region - 1) * os_
    xs, os2 = x.untyped_storage().nbytes() // x.element_size(), out.untyped_storage().nbytes() // out.element_size()
    assert x.storage_offset() + lo >= 0 and x.storage_offset() + hi < xs, "index_map: input range outside the tensor's storage"
    assert out.storage_offset() + olo >= 0 and out.storage_offset() + ohi < os2, "index_map: output range outside the tensor's storage"
    maps = _collapse_maps(maps)
    flat = []
    for m in maps:
        flat += [int(v) for v in m]
    arr = (ctypes.c_int64 * len(flat))(*flat)
    rc = L.hip().dl4j_index_map(L.ptr(x), L.ptr(out), L.c_int(len(maps)), arr, L.c_int64(int(out_base)),
                                ctypes.c_uint64(_pad_bits(pad_value, x.dtype)), L.c_int(x.element_size()), L.stream())
    L.check(rc, "index_map")
    return out


def _collapse_maps(maps):
    """Merge neighbouring dimensions that are copied unchanged (shift by 0 over the whole extent) and are laid out as one dimension on both
    sides: fewer coordinates to decompose per row and longer innermost rows for the kernel (TensorIterator-style coalescing)."""
    out = []
    for m in maps:
        os_, is_, region, in_dim, a, b, mode = [int(v) for v in m]
        if region == 1 and in_dim == 1 and mode in (SHIFT, REVERSE, WRAP) and len(maps) > 1 and (mode != SHIFT or a == 0):
            continue                                     # size-1 dimension: contributes nothing
        if out:
            po, pi, pr, pd, pa, pb, pm = out[-1]
            ident = mode == SHIFT and a == 0 and region == in_dim
            pident = pm == SHIFT and pa == 0 and pr == pd
            if ident and pident and po == os_ * region and pi == is_ * in_dim and pr * region < 2 ** 31:
                out[-1] = (os_, is_, pr * region, pd * in_dim, 0, 0, SHIFT)
                continue
        out.append((os_, is_, region, in_dim, a, b, mode))
    return out or [tuple(int(v) for v in maps[-1])]


def _cstrides(shape):
    st, acc = [], 1
    for s in reversed(shape):
        st.append(acc)
        acc *= int(s)
    return list(reversed(st))


def _merge_to_8(x):
    if x.dim() > 8:
        raise ValueError("index_map supports rank <= 8")
    return x


def pad_nd(x, pads, mode="constant", value=0.0):
    """pads: [(lo, hi)] per dimension (negative = crop); mode constant | reflect | symmetric."""
    _merge_to_8(x)
    oshape = [int(s) + int(lo) + int(hi) for s, (lo, hi) in zip(x.shape, pads)]
    out = torch.empty(oshape, device=x.device, dtype=x.dtype)
    md = {"constant": SHIFT, "reflect": REFLECT, "symmetric": SYMMETRIC}[mode]
    ost = _cstrides(oshape)
    maps = [(ost[k], x.stride(k), oshape[k], x.shape[k], -int(pads[k][0]), 0, md) for k in range(x.dim())]
    return index_map(x, out, maps, 0, value)


def reverse_dims(x, dims):
    out = torch.empty(x.shape, device=x.device, dtype=x.dtype)
    ost = _cstrides(x.shape)
    ds = {d % x.dim() for d in dims}
    maps = [(ost[k], x.stride(k), x.shape[k], x.shape[k], 0, 0, REVERSE if k in ds else SHIFT) for k in range(x.dim())]
    return index_map(x, out, maps)


def tile_nd(x, reps):
    reps = [1] * (x.dim() - len(reps)) + [int(r) for r in reps]
    xv = x.reshape([1] * (len(reps) - x.dim()) + list(x.shape))
    oshape = [int(s) * r for s, r in zip(xv.shape, reps)]
    out = torch.empty(oshape, device=x.device, dtype=x.dtype)
    ost = _cstrides(oshape)
    maps = [(ost[k], xv.stride(k), oshape[k], xv.shape[k], 0, 0, WRAP) for k in range(xv.dim())]
    return index_map(xv, out, maps)


def roll_nd(x, shifts, dims):
    out = torch.empty(x.shape, device=x.device, dtype=x.dtype)
    ost = _cstrides(x.shape)
    sh = {d % x.dim(): int(s) for s, d in zip(shifts, dims)}
    maps = [(ost[k], x.stride(k), x.shape[k], x.shape[k], -sh.get(k, 0), 0, WRAP) for k in range(x.dim())]
    return index_map(x, out, maps)


def permute_copy(x, perm):
    """Materialised permute (contiguous result)."""
    xv = x.permute(*perm)
    out = torch.empty(xv.shape, device=x.device, dtype=x.dtype)
    ost = _cstrides(xv.shape)
    maps = [(ost[k], xv.stride(k), xv.shape[k], xv.shape[k], 0, 0, SHIFT) for k in range(xv.dim())]
    return index_map(xv, out, maps)


def concat_nd(xs, axis):
    axis = axis % xs[0].dim()
    oshape = list(xs[0].shape)
    oshape[axis] = sum(int(t.shape[axis]) for t in xs)
    out = torch.empty(oshape, device=xs[0].device, dtype=xs[0].dtype)
    ost = _cstrides(oshape)
    off = 0
    for t in xs:
        maps = [(ost[k], t.stride(k), t.shape[k], t.shape[k], 0, 0, SHIFT) for k in range(t.dim())]
        index_map(t, out, maps, off * ost[axis])
        off += int(t.shape[axis])
    return out


def strided_slice_nd(x, begins, steps, counts):
    oshape = [int(c) for c in counts]
    out = torch.empty(oshape, device=x.device, dtype=x.dtype)
    ost = _cstrides(oshape)
    maps = [(ost[k], x.stride(k), oshape[k], x.shape[k], int(begins[k]), int(steps[k]), STRIDE) for k in range(x.dim())]
    return index_map(x, out, maps)


DEFER_LN_SUMS = __import__("os").environ.get("DL4J_AMD_LN_DEFER_SUMS", "1") == "1"


class LayerNormFn(torch.autograd.Function):
    """Differentiable layer norm over the last axis: HIP kernels in both directions (one pass forward with saved mean / rstd,
    dx pass + partial column reduction backward). Used by the layer_norm platform helper while SameDiff records a training graph."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        xc = x.contiguous()
        y, mean, rstd = layer_norm(xc, gamma, beta, eps, want_stats=True)
        ctx.save_for_backward(xc, gamma, mean, rstd)
        ctx.has_beta = beta is not None
        ctx.bdt = None if beta is None else beta.dtype
        from . import linear as LIN
        ctx.links = (LIN.link_of(gamma), LIN.link_of(beta))
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, mean, rstd = ctx.saved_tensors
        need_p = ctx.needs_input_grad[1] or (ctx.has_beta and ctx.needs_input_grad[2])
        gl, bl = ctx.links
        cols = x.shape[-1]
        if (need_p and gl is not None and bl is not None and gl.hits == 0 and bl.hits == 0 and ctx.needs_input_grad[1]
                and ctx.needs_input_grad[2] and bl.grad.data_ptr() == gl.grad.data_ptr() + 4 * cols):
            # gain and bias gradients are adjacent in the flat gradient vector: the finishing kernel writes them in place
            gl.hits += 1
            bl.hits += 1
            dx, _, _ = layer_norm_bwd(x, g.to(x.dtype).contiguous(), gamma, mean, rstd, True,
                                      sums_out=torch.as_strided(gl.grad, (2 * cols,), (1,)), defer_sums=DEFER_LN_SUMS)
            return dx, None, None, None
        dx, dg, db = layer_norm_bwd(x, g.to(x.dtype).contiguous(), gamma, mean, rstd, want_param_grads=need_p)
        dgamma = dg.reshape(gamma.shape).to(gamma.dtype) if (dg is not None and ctx.needs_input_grad[1]) else None
        dbeta = db.reshape(gamma.shape).to(ctx.bdt) if (db is not None and ctx.has_beta and ctx.needs_input_grad[2]) else None
        return dx, dgamma, dbeta, None


# ----------------------------------------------------------------------------- selection / indexed reductions (helpers_select.hip)
def segment_reduce(x, seg, num_segments, kind):
    """kind: 'max' | 'min' | 'prod' | 'mean'. Empty segments keep the identity (-/+ finfo.max, 1) or 0 for mean."""
    if kind == "mean":
        s = segment_sum(x, seg, num_segments).float()
        cnt = segment_sum(torch.ones(x.shape[0], 1, device=x.device), seg, num_segments).clamp_min(1.0)
        return (s / cnt.reshape([-1] + [1] * (x.dim() - 1))).to(x.dtype)
    rows = x.shape[0]
    D = x.numel() // max(rows, 1)
    big = torch.finfo(x.dtype).max
    init = {"max": -big, "min": big, "prod": 1.0}[kind]
    out = torch.full((int(num_segments),) + tuple(x.shape[1:]), init, device=x.device, dtype=torch.float32)
    L.check(L.hip().dl4j_segment_reduce(L.ptr(x), L.ptr(seg), L.ptr(out), L.c_int64(rows), L.c_int(D), L.c_int64(int(num_segments)),
                                        L.c_int({"max": 1, "min": 2, "prod": 3}[kind]), L.c_int(L.dtype_code(x)),
                                        L.c_int(1 if seg.dtype == torch.int64 else 0), L.stream()), "segment_reduce")
    return out if x.dtype == torch.float32 else out.to(x.dtype)


def nd_flatten(idx, shape):
    """idx [..., k] int tuples into the leading k dims `shape` -> int64 flat row numbers [...], -1 where out of range."""
    k = idx.shape[-1]
    assert 1 <= k <= 8 and len(shape) == k and _idx_ok(idx)
    flat = torch.empty(idx.shape[:-1], device=idx.device, dtype=torch.int64)
    shp = (ctypes.c_int64 * k)(*[int(v) for v in shape])
    L.check(L.hip().dl4j_nd_flatten(L.ptr(idx), L.ptr(flat), L.c_int64(flat.numel()), L.c_int(k), shp,
                                    L.c_int(1 if idx.dtype == torch.int64 else 0), L.stream()), "nd_flatten")
    return flat


def gather_nd(x, idx):
    k = idx.shape[-1]
    rows = 1
    for v in x.shape[:k]:
        rows *= int(v)
    flat = nd_flatten(idx, x.shape[:k])
    out = gather_rows(x.reshape(rows, -1), flat)
    return out.reshape(tuple(idx.shape[:-1]) + tuple(x.shape[k:]))


def scatter_set_rows_(table, flat_idx, src):
    V, D = table.shape
    assert table.is_contiguous() and src.is_contiguous() and src.dtype == table.dtype and src.numel() == flat_idx.numel() * D
    assert flat_idx.dtype == torch.int64 and flat_idx.is_contiguous()
    L.check(L.hip().dl4j_scatter_set_rows(L.ptr(src), L.ptr(flat_idx), L.ptr(table), L.c_int64(flat_idx.numel()), L.c_int(D), L.c_int64(V),
                                          L.c_int(table.element_size()), L.stream()), "scatter_set_rows")
    return table


def scatter_nd(x, idx, upd, kind):
    """kind 'add' | 'sub' | 'update'; returns a new tensor shaped like x."""
    k = idx.shape[-1]
    rows = 1
    for v in x.shape[:k]:
        rows *= int(v)
    flat = nd_flatten(idx, x.shape[:k]).reshape(-1)
    D = x.numel() // max(rows, 1)
    u = upd.to(x.dtype).contiguous()
    assert u.numel() == flat.numel() * D, "scatter_nd: updates do not match indices x slice size"
    if kind == "update":
        out = x.clone().reshape(rows, D)
        return scatter_set_rows_(out, flat, u).reshape(x.shape)
    acc = x.float().reshape(rows, D).contiguous()
    if acc.data_ptr() == x.data_ptr():
        acc = acc.clone()
    scatter_add_rows_(acc, flat, u, -1.0 if kind == "sub" else 1.0)
    return acc.reshape(x.shape).to(x.dtype)


def top_k(x, k):
    """Values and int64 indices of the k largest entries along the last axis, sorted descending (ties: lower index first)."""
    cols = x.shape[-1]
    assert 0 < k <= cols
    rows = x.numel() // cols
    vals = torch.empty(tuple(x.shape[:-1]) + (k,), device=x.device, dtype=x.dtype)
    inds = torch.empty(tuple(x.shape[:-1]) + (k,), device=x.device, dtype=torch.int64)
    L.check(L.hip().dl4j_top_k(L.ptr(x), L.ptr(vals), L.ptr(inds), L.c_int64(rows), L.c_int(cols), L.c_int(int(k)), L.c_int(L.dtype_code(x)),
                               L.stream()), "top_k")
    return vals, inds


def histogram_fixed_width(x, lo, hi, bins):
    counts = torch.zeros(int(bins), device=x.device, dtype=torch.int64)
    L.check(L.hip().dl4j_histogram(L.ptr(x), L.ptr(counts), L.c_int64(x.numel()), L.c_int(int(bins)), L.c_float(lo), L.c_float(hi),
                                   L.c_int(L.dtype_code(x)), L.stream()), "histogram")
    return counts


def lrn_bwd_last(x, g, depth, bias, alpha, beta):
    C = x.shape[-1]
    assert C <= 4096 and g.shape == x.shape and g.dtype == x.dtype and g.is_contiguous()
    dx = torch.empty_like(x)
    L.check(L.hip().dl4j_lrn_bwd(L.ptr(x), L.ptr(g), L.ptr(dx), L.c_int64(x.numel() // C), L.c_int(C), L.c_int(int(depth)), L.c_float(bias),
                                 L.c_float(alpha), L.c_float(beta), L.c_int(L.dtype_code(x)), L.stream()), "lrn_bwd")
    return dx


def sru_recurrence(xm, z, bias, c0):
    """The elementwise SRU recurrence given the projections: xm [B, K, T], z [B, 3K, T] fp32 contiguous, bias [2K], c0 [B, K] -> (h, c)."""
    B, K, T = xm.shape
    assert tuple(z.shape) == (B, 3 * K, T) and xm.is_contiguous() and z.is_contiguous() and xm.dtype == z.dtype == torch.float32
    h = torch.empty_like(xm)
    cs = torch.empty_like(xm)
    L.check(L.hip().dl4j_sru_fwd(L.ptr(xm), L.ptr(z), L.ptr(_f32(bias)), L.ptr(c0.float().contiguous()), L.ptr(h), L.ptr(cs), L.c_int(B),
                                 L.c_int(K), L.c_int(T), L.stream()), "sru_fwd")
    return h, cs


def sru_fwd(x, w, bias, c0, mask=None):
    """x [B, K, T] fp32, w [3K, K], bias [2K], c0 [B, K] -> (h, c) [B, K, T]; the projection is a library GEMM, the recurrence one kernel."""
    B, K, T = x.shape
    assert w.shape == (3 * K, K) and bias.numel() == 2 * K and c0.shape == (B, K)
    xm = (x * mask.unsqueeze(-1) if mask is not None else x).float().contiguous()
    z = torch.matmul(w.float(), xm).contiguous()                      # [B, 3K, T]
    h, cs = sru_recurrence(xm, z, bias, c0)
    return h, cs, z, xm


def sru_bi_fwd(x, w, bias, c0):
    """Bidirectional SRU: x [T, B, 2K], w [2K, 6K], bias [4K], c0 [B, 2K] -> (h, c) [T, B, 2K]; the backward direction runs the
    same recurrence kernel on the time-reversed sequence."""
    T, B, K2 = x.shape
    K = K2 // 2
    z = torch.matmul(x.float(), w.float()).reshape(T, B, 2, 3 * K)
    hs, cs = [], []
    for dirn in range(2):
        zd = z[:, :, dirn].permute(1, 2, 0)                           # [B, 3K, T]
        xd = x[:, :, dirn * K:(dirn + 1) * K].float().permute(1, 2, 0)
        if dirn == 1:
            zd, xd = zd.flip(-1), xd.flip(-1)
        h, c = sru_recurrence(xd.contiguous(), zd.contiguous(), bias[dirn * 2 * K:(dirn + 1) * 2 * K], c0[:, dirn * K:(dirn + 1) * K])
        if dirn == 1:
            h, c = h.flip(-1), c.flip(-1)
        hs.append(h.permute(2, 0, 1))
        cs.append(c.permute(2, 0, 1))
    return torch.cat(hs, -1).contiguous(), torch.cat(cs, -1).contiguous()


def barnes_edge_forces(rowP, colP, val, y):
    """Attractive forces of Barnes-Hut t-SNE over a CSR neighbour graph; y [N, D] (D <= 8) fp32."""
    N, D = y.shape
    assert 1 <= D <= 8 and rowP.numel() >= N + 1 and rowP.dtype == colP.dtype and _idx_ok(rowP) and _idx_ok(colP)
    yc, vc = y.float().contiguous(), val.float().contiguous()
    out = torch.empty_like(yc)
    L.check(L.hip().dl4j_barnes_edge_forces(L.ptr(rowP), L.ptr(colP), L.ptr(vc), L.ptr(yc), L.ptr(out), L.c_int(N), L.c_int(D),
                                            L.c_int64(min(colP.numel(), vc.numel())), L.c_int(1 if rowP.dtype == torch.int64 else 0),
                                            L.stream()), "barnes_edge_forces")
    return out.to(y.dtype)


def sru_bwd(x, w, bias, c0, gh, gc_last=None, mask=None):
    """Returns (dx, dw, db, dc0) for sru_fwd given dL/dh (and optionally dL/dc_T)."""
    B, K, T = x.shape
    h, cs, z, xm = sru_fwd(x, w, bias, c0, mask)
    dz = torch.empty_like(z)
    dxd = torch.empty_like(xm)
    dc0 = torch.empty(B, K, device=x.device, dtype=torch.float32)
    ghc = gh.float().contiguous()
    gcl = None if gc_last is None else gc_last.float().contiguous()
    L.check(L.hip().dl4j_sru_bwd(L.ptr(xm), L.ptr(z), L.ptr(_f32(bias)), L.ptr(c0.float().contiguous()), L.ptr(cs), L.ptr(ghc), L.ptr(gcl),
                                 L.ptr(dz), L.ptr(dxd), L.ptr(dc0), L.c_int(B), L.c_int(K), L.c_int(T), L.stream()), "sru_bwd")
    wf = w.float()
    dx = torch.matmul(wf.t(), dz) + dxd
    if mask is not None:
        dx = dx * mask.unsqueeze(-1)
    dw = torch.einsum("bot,bkt->ok", dz, xm)
    db = dz[:, K:].sum((0, 2))
    return dx, dw, db, dc0


def color_convert(x, mode, matrix=None, param=0.0):
    """x [..., 3] channel-last; mode 'rgb_to_hsv' | 'hsv_to_rgb' | 'matrix' (3x3 nested list, out = M @ pixel) |
    'adjust_hue' (param = delta) | 'adjust_saturation' (param = factor)."""
    assert x.shape[-1] == 3
    y = torch.empty_like(x)
    m = None
    if mode == "matrix":
        m = (ctypes.c_float * 9)(*[float(v) for row in matrix for v in row])
    elif mode in ("adjust_hue", "adjust_saturation"):
        m = (ctypes.c_float * 9)(float(param), *([0.0] * 8))
    code = {"rgb_to_hsv": 0, "hsv_to_rgb": 1, "matrix": 2, "adjust_hue": 3, "adjust_saturation": 4}[mode]
    L.check(L.hip().dl4j_color_convert(L.ptr(x), L.ptr(y), L.c_int64(x.numel() // 3), L.c_int(code), m, L.c_int(L.dtype_code(x)),
                                       L.stream()), "color_convert")
    return y


def scatter_reduce_rows(x, idx, upd, kind):
    """x[idx[r], ...] = op(x[idx[r], ...], upd[r, ...]) for kind 'max' | 'min' | 'mul' (fp32 CAS atomics); returns a new tensor."""
    rows = x.shape[0]
    D = x.numel() // max(rows, 1)
    acc = x.float().reshape(rows, D).contiguous()
    if acc.data_ptr() == x.data_ptr():
        acc = acc.clone()
    u = upd.to(x.dtype).contiguous()
    assert u.numel() == idx.numel() * D
    L.check(L.hip().dl4j_segment_reduce(L.ptr(u), L.ptr(idx), L.ptr(acc), L.c_int64(idx.numel()), L.c_int(D), L.c_int64(rows),
                                        L.c_int({"max": 1, "min": 2, "mul": 3}[kind]), L.c_int(L.dtype_code(u)),
                                        L.c_int(1 if idx.dtype == torch.int64 else 0), L.stream()), "scatter_reduce_rows")
    return acc.reshape(x.shape).to(x.dtype)


def space_to_batch_nd(x, block, pads):
    """x [N, spatial..., rest...]: zero-pad the spatial dims, then move the block offsets into the batch (one pad + one permute launch)."""
    m = len(block)
    full = [(0, 0)] + [(int(lo), int(hi)) for lo, hi in pads] + [(0, 0)] * (x.dim() - 1 - m)
    xp = pad_nd(x, full) if any(lo or hi for lo, hi in full) else x
    n, sp, rest = xp.shape[0], list(xp.shape[1:1 + m]), list(xp.shape[1 + m:])
    shp = [n]
    for s_, bl in zip(sp, block):
        shp += [s_ // bl, bl]
    v = xp.reshape(shp + rest)
    perm = [2 + 2 * k for k in range(m)] + [0] + [1 + 2 * k for k in range(m)] + list(range(1 + 2 * m, 1 + 2 * m + len(rest)))
    nb = 1
    for bl in block:
        nb *= int(bl)
    return permute_copy(v, perm).reshape([n * nb] + [s_ // bl for s_, bl in zip(sp, block)] + rest)


def batch_to_space_nd(x, block, crops):
    m = len(block)
    nb = 1
    for bl in block:
        nb *= int(bl)
    n, sp, rest = x.shape[0] // nb, list(x.shape[1:1 + m]), list(x.shape[1 + m:])
    v = x.reshape(list(block) + [n] + sp + rest)
    perm = [m]
    for k in range(m):
        perm += [m + 1 + k, k]
    perm += list(range(2 * m + 1, 2 * m + 1 + len(rest)))
    y = permute_copy(v, perm).reshape([n] + [s_ * bl for s_, bl in zip(sp, block)] + rest)
    key = [slice(None)] + [slice(int(lo), y.shape[1 + k] - int(hi)) for k, (lo, hi) in enumerate(crops)]
    return y[tuple(key)]


def nms(boxes, scores, max_out, iou_thr, score_thr=float("-inf"), overlaps=None):
    """Greedy NMS; boxes [N, 4] fp32 (or a precomputed overlaps [N, N]); returns int32 kept indices in selection order."""
    N = scores.numel()
    if N == 0 or max_out <= 0:
        return torch.empty(0, dtype=torch.int32, device=scores.device)
    sc = scores.float().contiguous()
    order = torch.argsort(sc, descending=True, stable=True)
    keep = torch.empty(int(min(max_out, N)), device=sc.device, dtype=torch.int32)
    count = torch.empty(1, device=sc.device, dtype=torch.int32)
    bx = None if boxes is None else boxes.float().contiguous()
    ov = None if overlaps is None else overlaps.float().contiguous()
    assert (bx is not None and bx.shape == (N, 4)) or (ov is not None and ov.shape == (N, N))
    st = max(float(score_thr), -3.0e38)
    L.check(L.hip().dl4j_nms(L.ptr(bx), L.ptr(ov), L.ptr(sc), L.ptr(order), L.ptr(keep), L.ptr(count), L.c_int(N), L.c_int(keep.numel()),
                             L.c_float(iou_thr), L.c_float(st), L.stream()), "nms")
    return keep[:int(count.item())]


# ----------------------------------------------------------------------------- spatial (helpers_spatial.hip)
def _dw_geom(x, w, s, p_lo, dl, out_hw=None):
    N, C, H, W = x.shape
    OC, kH, kW = w.shape
    assert OC % C == 0
    if out_hw is None:
        oH = (H + 2 * p_lo[0] - dl[0] * (kH - 1) - 1) // s[0] + 1
        oW = (W + 2 * p_lo[1] - dl[1] * (kW - 1) - 1) // s[1] + 1
    else:
        oH, oW = out_hw
    assert oH > 0 and oW > 0
    return (ctypes.c_int * 15)(N, C, OC // C, H, W, kH, kW, oH, oW, s[0], s[1], p_lo[0], p_lo[1], dl[0], dl[1]), oH, oW


def depthwise_fwd(x, w, bias, s, p_lo, dl, out_hw=None):
    """x [N, C, H, W], w [C*mC, kH, kW] (same dtype), bias fp32-able [C*mC]; p_lo = top/left padding, out_hw overrides the output size
    (asymmetric SAME padding: the bottom/right padding is implied by the bounds checks)."""
    g, oH, oW = _dw_geom(x, w, s, p_lo, dl, out_hw)
    y = torch.empty((x.shape[0], w.shape[0], oH, oW), device=x.device, dtype=x.dtype)
    L.check(L.hip().dl4j_depthwise_fwd(L.ptr(x), L.ptr(w), L.ptr(_f32(bias)), L.ptr(y), g, L.c_int(L.dtype_code(x)), L.stream()), "depthwise_fwd")
    return y


def depthwise_bwd(x, w, dy, s, p_lo, dl, want_bias=True):
    """Returns (dx, dw fp32 [C*mC, kH, kW], db fp32 [C*mC] or None)."""
    g, oH, oW = _dw_geom(x, w, s, p_lo, dl, (dy.shape[2], dy.shape[3]))
    assert dy.shape[:2] == (x.shape[0], w.shape[0]) and dy.dtype == x.dtype and dy.is_contiguous()
    dx = torch.empty_like(x)
    L.check(L.hip().dl4j_depthwise_bwd_x(L.ptr(dy), L.ptr(w), L.ptr(dx), g, L.c_int(L.dtype_code(x)), L.stream()), "depthwise_bwd_x")
    dw = torch.empty(w.shape, device=x.device, dtype=torch.float32)
    db = torch.empty(w.shape[0], device=x.device, dtype=torch.float32) if want_bias else None
    L.check(L.hip().dl4j_depthwise_bwd_w(L.ptr(dy), L.ptr(x), L.ptr(dw), L.ptr(db), g, L.c_int(L.dtype_code(x)), L.stream()), "depthwise_bwd_w")
    return dx, dw, db


def _p3_geom(shape, k, s, p, dl, mode, include_pad, out=None):
    D, H, W = shape
    if out is None:
        out = [(n + 2 * p[a] - dl[a] * (k[a] - 1) - 1) // s[a] + 1 for a, n in enumerate((D, H, W))]
    assert all(o > 0 for o in out)
    return (ctypes.c_int * 20)(D, H, W, *k, *out, *s, *p, *dl, {"max": 0, "avg": 1, "pnorm": 2}[mode], int(include_pad)), out


def pool3d(x, k, s, p, dl=(1, 1, 1), mode="max", include_pad=True, pnorm=2.0, out_dhw=None):
    """x [N, C, D, H, W] contiguous -> [N, C, oD, oH, oW]; p = front/top/left padding (trailing padding is implied by out_dhw)."""
    N, C = x.shape[:2]
    g, out = _p3_geom(x.shape[2:], list(k), list(s), list(p), list(dl), mode, include_pad, out_dhw)
    y = torch.empty((N, C) + tuple(out), device=x.device, dtype=x.dtype)
    L.check(L.hip().dl4j_pool3d_fwd(L.ptr(x), L.ptr(y), L.c_int64(N * C), g, L.c_float(pnorm), L.c_int(L.dtype_code(x)), L.stream()), "pool3d_fwd")
    return y


def pool3d_bwd(x, y, gy, k, s, p, dl=(1, 1, 1), mode="max", include_pad=True, pnorm=2.0):
    N, C = x.shape[:2]
    assert y.shape == gy.shape and y.dtype == x.dtype == gy.dtype and y.is_contiguous() and gy.is_contiguous()
    g, _ = _p3_geom(x.shape[2:], list(k), list(s), list(p), list(dl), mode, include_pad, list(y.shape[2:]))
    dx = torch.empty_like(x)
    L.check(L.hip().dl4j_pool3d_bwd(L.ptr(x), L.ptr(y), L.ptr(gy), L.ptr(dx), L.c_int64(N * C), g, L.c_float(pnorm), L.c_int(L.dtype_code(x)),
                                    L.stream()), "pool3d_bwd")
    return dx


def upsample3d(x, fD, fH, fW):
    N, C, D, H, W = x.shape
    y = torch.empty((N, C, D * fD, H * fH, W * fW), device=x.device, dtype=x.dtype)
    L.check(L.hip().dl4j_upsample3d_fwd(L.ptr(x), L.ptr(y), L.c_int64(N * C), L.c_int(D), L.c_int(H), L.c_int(W), L.c_int(fD), L.c_int(fH),
                                        L.c_int(fW), L.c_int(x.element_size()), L.stream()), "upsample3d_fwd")
    return y


def upsample3d_bwd(g, fD, fH, fW):
    N, C, oD, oH, oW = g.shape
    assert oD % fD == 0 and oH % fH == 0 and oW % fW == 0
    dx = torch.empty((N, C, oD // fD, oH // fH, oW // fW), device=g.device, dtype=g.dtype)
    L.check(L.hip().dl4j_upsample3d_bwd(L.ptr(g), L.ptr(dx), L.c_int64(N * C), L.c_int(oD // fD), L.c_int(oH // fH), L.c_int(oW // fW),
                                        L.c_int(fD), L.c_int(fH), L.c_int(fW), L.c_int(L.dtype_code(g)), L.stream()), "upsample3d_bwd")
    return dx


def _nhwc_geom(x, kH, kW, s, p_lo, r, out_hw):
    N, H, W, C = x.shape
    return (ctypes.c_int * 14)(N, H, W, C, kH, kW, out_hw[0], out_hw[1], s[0], s[1], p_lo[0], p_lo[1], r[0], r[1])


def dilation2d(x, w, s, p_lo, r, out_hw):
    """x [N, H, W, C], w [kH, kW, C] -> [N, oH, oW, C] = max over taps of (x + w); out-of-image taps are skipped."""
    assert w.shape[2] == x.shape[3] and w.dtype == x.dtype and w.is_contiguous()
    y = torch.empty((x.shape[0], out_hw[0], out_hw[1], x.shape[3]), device=x.device, dtype=x.dtype)
    L.check(L.hip().dl4j_dilation2d(L.ptr(x), L.ptr(w), L.ptr(y), _nhwc_geom(x, w.shape[0], w.shape[1], s, p_lo, r, out_hw),
                                    L.c_int(L.dtype_code(x)), L.stream()), "dilation2d")
    return y


def extract_patches(x, kH, kW, s, p_lo, r, out_hw):
    """x [N, H, W, C] -> [N, oH, oW, kH*kW*C] (zero padded)."""
    y = torch.empty((x.shape[0], out_hw[0], out_hw[1], kH * kW * x.shape[3]), device=x.device, dtype=x.dtype)
    L.check(L.hip().dl4j_extract_patches(L.ptr(x), L.ptr(y), _nhwc_geom(x, kH, kW, s, p_lo, r, out_hw), L.c_int(x.element_size()), L.stream()),
            "extract_patches")
    return y
