"""Python entry points of ``native/csrc/helpers_misc.hip``: order statistics, row sort, vol2col / col2vol, generators, TF-exact
bicubic / area resize, adjust_contrast, draw_bounding_boxes, t-SNE gains + CSR symmetrisation, batched skip-gram / CBOW.

Each function launches the HIP kernel when its operands live on the GPU and otherwise runs a plain PyTorch composition with the same
semantics (the ``*_reference`` functions, which are also what the GPU tests compare against).

Reference behaviour: libnd4j/include/ops/declarable/helpers/cuda/{nth_element,percentile,image_resize,image_draw_bounding_boxes,
BarnesHutTsne,sg_cb,convolutions_vol2col,convolutions_col2vol,range,diag,meshgrid}.cu and helpers/cpu/sg_cb.cpp:34-120.
"""
from __future__ import annotations

import ctypes
import math

import torch

from ..native import lib as L

_FLOAT = (torch.float32, torch.bfloat16)


def _hip_float(x):
    return L.use_hip(x) and x.dtype in _FLOAT and x.is_contiguous()


# ----------------------------------------------------------------------------- order statistics / sort
def nth_element(x, n, reverse=False):
    """n-th smallest (largest when ``reverse``) element along the last axis; output shape = x.shape[:-1]."""
    cols = x.shape[-1]
    n = int(n)
    if not 0 <= n < cols:
        raise ValueError("nth_element: n=%d out of range for %d elements" % (n, cols))
    if _hip_float(x):
        rows = x.numel() // cols
        out = torch.empty(x.shape[:-1], device=x.device, dtype=x.dtype)
        L.check(L.hip().dl4j_nth_element(L.ptr(x), L.ptr(out), L.c_int64(rows), L.c_int(cols), L.c_int(cols - 1 - n if reverse else n),
                                         L.c_int(L.dtype_code(x)), L.stream()), "nth_element")
        return out
    return torch.sort(x, dim=-1, descending=bool(reverse)).values[..., n]


def percentile_positions(q, length, interpolation):
    """Index of the order statistic the percentile op returns. interpolation: 0 lower, 1 higher, 2 nearest."""
    pos = q / 100.0 * (length - 1)
    if interpolation == 0:
        return int(math.floor(pos))
    if interpolation == 1:
        return int(math.ceil(pos))
    return int(round(pos))


def sort_rows_supported(x):
    return _hip_float(x) and x.dim() >= 1 and 0 < x.shape[-1] <= int(L.hip().dl4j_sort_rows_max_cols())


def sort_rows(x, descending=False, values=True, indices=True):
    """Stable sort of every row (last axis) in LDS. Returns (values or None, int64 indices or None)."""
    cols = x.shape[-1]
    if sort_rows_supported(x):
        rows = x.numel() // cols
        v = torch.empty_like(x) if values else None
        ix = torch.empty(x.shape, device=x.device, dtype=torch.int64) if indices else None
        L.check(L.hip().dl4j_sort_rows(L.ptr(x), L.ptr(v), L.ptr(ix), L.c_int64(rows), L.c_int(cols), L.c_int(1 if descending else 0),
                                       L.c_int(L.dtype_code(x)), L.stream()), "sort_rows")
        return v, ix
    v, ix = torch.sort(x, dim=-1, descending=bool(descending), stable=True)
    return (v if values else None), (ix if indices else None)


# ----------------------------------------------------------------------------- vol2col / col2vol
def _vol_geom(N, C, D, H, W, k, s, p, d):
    oD = (D + 2 * p[0] - (k[0] - 1) * d[0] - 1) // s[0] + 1
    oH = (H + 2 * p[1] - (k[1] - 1) * d[1] - 1) // s[1] + 1
    oW = (W + 2 * p[2] - (k[2] - 1) * d[2] - 1) // s[2] + 1
    g = [N, C, D, H, W, k[0], k[1], k[2], oD, oH, oW, s[0], s[1], s[2], p[0], p[1], p[2], d[0], d[1], d[2]]
    return (ctypes.c_int * 20)(*[int(v) for v in g]), (oD, oH, oW)


def vol2col_reference(x, k, s, p, d):
    N, C, D, H, W = x.shape
    xp = torch.nn.functional.pad(x, [p[2], p[2], p[1], p[1], p[0], p[0]])
    _, (oD, oH, oW) = _vol_geom(N, C, D, H, W, k, s, p, d)
    col = x.new_zeros(N, C, k[0], k[1], k[2], oD, oH, oW)
    for a in range(k[0]):
        for b in range(k[1]):
            for c in range(k[2]):
                col[:, :, a, b, c] = xp[:, :, a * d[0]:a * d[0] + (oD - 1) * s[0] + 1:s[0], b * d[1]:b * d[1] + (oH - 1) * s[1] + 1:s[1],
                                        c * d[2]:c * d[2] + (oW - 1) * s[2] + 1:s[2]]
    return col


def vol2col(x, k, s=(1, 1, 1), p=(0, 0, 0), d=(1, 1, 1)):
    """[N, C, D, H, W] -> [N, C, kD, kH, kW, oD, oH, oW] (zero padded)."""
    if L.use_hip(x) and x.is_contiguous() and x.element_size() in (1, 2, 4, 8):
        N, C, D, H, W = x.shape
        geom, (oD, oH, oW) = _vol_geom(N, C, D, H, W, k, s, p, d)
        col = torch.empty(N, C, k[0], k[1], k[2], oD, oH, oW, device=x.device, dtype=x.dtype)
        L.check(L.hip().dl4j_vol2col(L.ptr(x), L.ptr(col), geom, L.c_int(x.element_size()), L.stream()), "vol2col")
        return col
    return vol2col_reference(x, k, s, p, d)


def col2vol_reference(col, vol_shape, s, p, d):
    N, C, kD, kH, kW, oD, oH, oW = col.shape
    D, H, W = vol_shape
    xp = col.new_zeros(N, C, D + 2 * p[0], H + 2 * p[1], W + 2 * p[2], dtype=torch.float32)
    for a in range(kD):
        for b in range(kH):
            for c in range(kW):
                xp[:, :, a * d[0]:a * d[0] + (oD - 1) * s[0] + 1:s[0], b * d[1]:b * d[1] + (oH - 1) * s[1] + 1:s[1],
                   c * d[2]:c * d[2] + (oW - 1) * s[2] + 1:s[2]] += col[:, :, a, b, c].float()
    return xp[:, :, p[0]:p[0] + D, p[1]:p[1] + H, p[2]:p[2] + W].to(col.dtype)


def col2vol(col, vol_shape, s=(1, 1, 1), p=(0, 0, 0), d=(1, 1, 1)):
    """Adjoint of vol2col: [N, C, kD, kH, kW, oD, oH, oW] -> [N, C, D, H, W] (deterministic gather)."""
    if _hip_float(col):
        N, C, kD, kH, kW, oD, oH, oW = col.shape
        D, H, W = [int(v) for v in vol_shape]
        geom, o = _vol_geom(N, C, D, H, W, (kD, kH, kW), s, p, d)
        if o != (oD, oH, oW):
            raise ValueError("col2vol: column shape %s does not match the volume geometry %s" % ((oD, oH, oW), o))
        x = torch.empty(N, C, D, H, W, device=col.device, dtype=col.dtype)
        L.check(L.hip().dl4j_col2vol(L.ptr(col), L.ptr(x), geom, L.c_int(L.dtype_code(col)), L.stream()), "col2vol")
        return x
    return col2vol_reference(col, vol_shape, s, p, d)


# ----------------------------------------------------------------------------- generators
def fill_range(start, step, n, dtype, device):
    """start + i * step for i < n, generated on the device."""
    n = int(n)
    code = L.dtype_code_all(torch.empty(0, dtype=dtype))
    dev = torch.device(device) if device is not None else torch.device("cpu")
    if dev.type == "cuda" and code >= 0 and L.use_hip(torch.empty(0, device=dev)):
        out = torch.empty(max(n, 0), device=dev, dtype=dtype)
        if code in (3, 4) and float(start).is_integer() and float(step).is_integer():
            L.check(L.hip().dl4j_fill_range_i64(L.ptr(out), L.c_int64(n), L.c_int64(int(start)), L.c_int64(int(step)), L.c_int(code), L.stream()),
                    "fill_range")
        else:
            L.check(L.hip().dl4j_fill_range(L.ptr(out), L.c_int64(n), ctypes.c_double(float(start)), ctypes.c_double(float(step)), L.c_int(code),
                                            L.stream()), "fill_range")
        return out
    i = torch.arange(max(n, 0), device=dev, dtype=torch.float64 if dtype.is_floating_point else torch.int64)
    return (start + i * step).to(dtype)


def diag(x):
    """Vector [n] -> matrix [n, n] (libnd4j diag on a flattened input; callers reshape for rank > 1)."""
    v = x.reshape(-1).contiguous()
    n = v.numel()
    if L.use_hip(v) and v.element_size() in (1, 2, 4, 8):
        out = torch.empty(n, n, device=v.device, dtype=v.dtype)
        L.check(L.hip().dl4j_diag(L.ptr(v), L.ptr(out), L.c_int64(n), L.c_int(v.element_size()), L.stream()), "diag")
        return out
    return torch.diag(v)


def diag_part(x):
    """Matrix [n, n] (contiguous) -> its diagonal [n]."""
    n = int(math.isqrt(x.numel()))
    m = x.reshape(n, n).contiguous()
    if L.use_hip(m) and m.element_size() in (1, 2, 4, 8):
        out = torch.empty(n, device=m.device, dtype=m.dtype)
        L.check(L.hip().dl4j_diag_part(L.ptr(m), L.ptr(out), L.c_int64(n), L.c_int(m.element_size()), L.stream()), "diag_part")
        return out
    return torch.diagonal(m).clone()


def meshgrid(vectors, xy=True):
    """Materialised coordinate grids (TF / numpy semantics: 'xy' swaps the first two output axes)."""
    vs = [v.reshape(-1).contiguous() for v in vectors]
    lens = [v.numel() for v in vs]
    if not (vs and all(L.use_hip(v) and v.element_size() in (1, 2, 4, 8) for v in vs)):
        return tuple(t.contiguous() for t in torch.meshgrid(*vs, indexing="xy" if xy else "ij"))
    shape = list(lens)
    axis_of = list(range(len(vs)))
    if xy and len(vs) >= 2:
        shape[0], shape[1] = shape[1], shape[0]
        axis_of[0], axis_of[1] = 1, 0
    total = 1
    for s in shape:
        total *= s
    outs = []
    for k, v in enumerate(vs):
        ax = axis_of[k]
        inner = 1
        for s in shape[ax + 1:]:
            inner *= s
        out = torch.empty(shape, device=v.device, dtype=v.dtype)
        L.check(L.hip().dl4j_meshgrid_axis(L.ptr(v), L.ptr(out), L.c_int64(total), L.c_int64(inner), L.c_int64(lens[k]), L.c_int(v.element_size()),
                                           L.stream()), "meshgrid")
        outs.append(out)
    return tuple(outs)


# ----------------------------------------------------------------------------- TF-exact resize
_CUBIC_TABLE = 1024


def _scale(n_in, n_out, align):
    return (n_in - 1) / (n_out - 1) if (align and n_out > 1) else n_in / n_out


def _f32(v):
    return float(torch.tensor(v, dtype=torch.float32))


def _cubic_matrix(n_in, n_out, align, half_pixel):
    """[n_out, n_in] interpolation matrix of TensorFlow's ResizeBicubic (quantised Keys coefficients), built in fp32 arithmetic."""
    import numpy as np
    f = np.float32
    scale = f(_scale(n_in, n_out, align))
    A = f(-0.5 if half_pixel else -0.75)

    def near(off):
        x = f(off) / f(_CUBIC_TABLE)
        return ((A + f(2)) * x - (A + f(3))) * x * x + f(1)

    def far(off):
        x = f(off) / f(_CUBIC_TABLE) + f(1)
        return ((A * x - f(5) * A) * x + f(8) * A) * x - f(4) * A

    M = np.zeros((n_out, n_in), dtype=np.float64)
    for o in range(n_out):
        loc = (f(o) + f(0.5)) * scale - f(0.5) if half_pixel else f(o) * scale
        il = int(np.floor(loc))
        delta = f(loc - f(il))
        off = int(np.rint(delta * f(_CUBIC_TABLE)))
        taps = [(il - 1, far(off)), (il, near(off)), (il + 1, near(_CUBIC_TABLE - off)), (il + 2, far(_CUBIC_TABLE - off))]
        if half_pixel:
            w = [wt if 0 <= ix <= n_in - 1 else f(0) for ix, wt in taps]
            s = f(sum(w))
            if abs(s) >= 1000.0 * 1.17549435e-38:
                w = [f(v / s) for v in w]
        else:
            w = [wt for _, wt in taps]
        for (ix, _), wt in zip(taps, w):
            M[o, min(max(ix, 0), n_in - 1)] += float(wt)
    return torch.from_numpy(M)


def _area_matrix(n_in, n_out, align):
    import numpy as np
    f = np.float32
    scale = f(_scale(n_in, n_out, align))
    M = np.zeros((n_out, n_in), dtype=np.float64)
    for o in range(n_out):
        a, b = f(o) * scale, f(o + 1) * scale
        for i in range(int(np.floor(a)), int(np.ceil(b))):
            if i < a:
                w = scale if i + 1 > b else f(i + 1) - a
            else:
                w = b - f(i) if i + 1 > b else f(1)
            M[o, min(max(i, 0), n_in - 1)] += float(w)
    return torch.from_numpy(M / float(scale))


def resize_tf_reference(x, oh, ow, mode, align=False, half_pixel=False):
    """NHWC bicubic / area resize with TensorFlow semantics as two separable interpolation matrices (fp32 output)."""
    H, W = x.shape[1], x.shape[2]
    if mode == "bicubic":
        My, Mx = _cubic_matrix(H, oh, align, half_pixel), _cubic_matrix(W, ow, align, half_pixel)
    else:
        My, Mx = _area_matrix(H, oh, align), _area_matrix(W, ow, align)
    xd = x.double()
    y = torch.einsum("oh,nhwc,pw->nopc", My.to(x.device), xd, Mx.to(x.device))
    return y.float()


def resize_tf(x, oh, ow, mode, align=False, half_pixel=False):
    if _hip_float(x) and x.dim() == 4:
        N, H, W, C = x.shape
        y = torch.empty(N, int(oh), int(ow), C, device=x.device, dtype=torch.float32)
        L.check(L.hip().dl4j_resize_tf(L.ptr(x), L.ptr(y), L.c_int(N), L.c_int(H), L.c_int(W), L.c_int(C), L.c_int(int(oh)), L.c_int(int(ow)),
                                       L.c_int(0 if mode == "bicubic" else 1), L.c_int(1 if align else 0), L.c_int(1 if half_pixel else 0),
                                       L.c_int(L.dtype_code(x)), L.stream()), "resize_tf")
        return y
    return resize_tf_reference(x, int(oh), int(ow), mode, align, half_pixel)


# ----------------------------------------------------------------------------- adjust_contrast / draw_bounding_boxes
def adjust_contrast(x, factor):
    """(x - mean_HW) * factor + mean_HW on [..., H, W, C]."""
    if _hip_float(x) and x.dim() >= 3 and x.shape[-1] <= 256:
        C = x.shape[-1]
        HW = x.shape[-3] * x.shape[-2]
        N = x.numel() // (HW * C) if HW * C else 0
        y = torch.empty_like(x)
        sums = torch.zeros(max(N, 1), C, device=x.device, dtype=torch.float32)
        L.check(L.hip().dl4j_adjust_contrast(L.ptr(x), L.ptr(y), L.ptr(sums), L.c_int(N), L.c_int64(HW), L.c_int(C), L.c_float(float(factor)),
                                             L.c_int(L.dtype_code(x)), L.stream()), "adjust_contrast")
        return y
    m = x.float().mean((-3, -2), keepdim=True)
    return ((x.float() - m) * factor + m).to(x.dtype)


def draw_boxes_reference(img, boxes, colors):
    out = img.clone()
    N, H, W, C = img.shape
    for n in range(N):
        for k in range(boxes.shape[1]):
            y1, x1, y2, x2 = boxes[n, k].tolist()
            r1, c1, r2, c2 = int(y1 * (H - 1)), int(x1 * (W - 1)), int(y2 * (H - 1)), int(x2 * (W - 1))
            if r1 > r2 or c1 > c2 or r1 >= H or c1 >= W or r2 < 0 or c2 < 0:
                continue
            col = colors[k % colors.shape[0]].to(img.dtype)
            r1c, r2c, c1c, c2c = max(r1, 0), min(r2, H - 1), max(c1, 0), min(c2, W - 1)
            if r1 >= 0:
                out[n, r1, c1c:c2c + 1] = col
            if r2 < H:
                out[n, r2, c1c:c2c + 1] = col
            if c1 >= 0:
                out[n, r1c:r2c + 1, c1] = col
            if c2 < W:
                out[n, r1c:r2c + 1, c2] = col
    return out


def draw_boxes(img, boxes, colors):
    """NHWC images, boxes [N, B, 4] (y1, x1, y2, x2 normalised), colors [K, C]; one launch instead of the per-box host loop."""
    if _hip_float(img) and img.dim() == 4 and L.use_hip(boxes, colors):
        N, H, W, C = img.shape
        bx = boxes.float().contiguous()
        cl = colors.float().contiguous()
        out = torch.empty_like(img)
        L.check(L.hip().dl4j_draw_boxes(L.ptr(img), L.ptr(bx), L.ptr(cl), L.ptr(out), L.c_int(N), L.c_int(H), L.c_int(W), L.c_int(C),
                                        L.c_int(bx.shape[1]), L.c_int(cl.shape[0]), L.c_int(L.dtype_code(img)), L.stream()), "draw_boxes")
        return out
    return draw_boxes_reference(img, boxes, colors)


# ----------------------------------------------------------------------------- t-SNE
def barnes_gains(gains, gradx, eps):
    if L.use_hip(gains, gradx, eps) and all(t.dtype == torch.float32 and t.is_contiguous() for t in (gains, gradx, eps)) \
            and gains.shape == gradx.shape == eps.shape:
        out = torch.empty_like(gains)
        L.check(L.hip().dl4j_barnes_gains(L.ptr(gains), L.ptr(gradx), L.ptr(eps), L.ptr(out), L.c_int64(gains.numel()), L.stream()), "barnes_gains")
        return out
    same = torch.sign(gradx) == torch.sign(eps)
    return torch.where(same, gains * 0.8, gains + 0.2).clamp_min(0.01)


def barnes_symmetrize_reference(rowP, colP, valP, N):
    rowP, colP = rowP.long(), colP.long()
    rows = torch.repeat_interleave(torch.arange(N, device=valP.device), rowP[1:N + 1] - rowP[:N])
    P = torch.sparse_coo_tensor(torch.stack([rows, colP[:rows.numel()]]), valP[:rows.numel()], (N, N)).coalesce()
    S = ((P + P.t()) * 0.5).coalesce()
    r, c = S.indices()
    out_rows = torch.zeros(N + 1, dtype=torch.int32, device=valP.device)
    out_rows[1:] = torch.cumsum(torch.bincount(r, minlength=N), 0).to(torch.int32)
    return out_rows, c.to(torch.int32), S.values()


def barnes_symmetrize(rowP, colP, valP, N):
    """CSR (rowP, colP, valP) of P -> CSR of (P + P^T) / 2 with ascending columns per row."""
    N = int(N)
    if L.use_hip(rowP, colP, valP) and valP.dtype == torch.float32:
        rp = rowP.to(torch.int32).contiguous()
        cp = colP.to(torch.int32).contiguous()
        vp = valP.contiguous()
        counts = torch.zeros(N, device=vp.device, dtype=torch.int32)
        h = L.hip()
        L.check(h.dl4j_barnes_sym_count(L.ptr(rp), L.ptr(cp), L.ptr(counts), L.c_int(N), L.stream()), "barnes_sym_count")
        sym_row = torch.zeros(N + 1, device=vp.device, dtype=torch.int32)
        sym_row[1:] = torch.cumsum(counts, 0).to(torch.int32)
        nnz = int(sym_row[-1])
        sym_col = torch.empty(nnz, device=vp.device, dtype=torch.int32)
        sym_val = torch.empty(nnz, device=vp.device, dtype=torch.float32)
        cursor = torch.zeros(N, device=vp.device, dtype=torch.int32)
        L.check(h.dl4j_barnes_sym_fill(L.ptr(rp), L.ptr(cp), L.ptr(vp), L.ptr(sym_row), L.ptr(cursor), L.ptr(sym_col), L.ptr(sym_val), L.c_int(N),
                                       L.stream()), "barnes_sym_fill")
        return sym_row, sym_col, sym_val
    return barnes_symmetrize_reference(rowP, colP, valP, N)


# ----------------------------------------------------------------------------- skip-gram / CBOW batch
SG_MAX_EXP = 6.0


def sg_cb_supported(syn0):
    return L.use_hip(syn0) and syn0.dtype == torch.float32 and syn0.is_contiguous() and syn0.dim() == 2 \
        and syn0.shape[1] <= int(L.hip().dl4j_sg_cb_max_dim())


def sg_cb_batch(syn0, syn1, syn1neg, neg_table, target, ng_starter, context, indices, codes, alpha, rnd, ns_rounds, train_words=True,
                infer=None):
    """One batched word2vec step on the GPU, tables updated in place (reference: sg_cb.cu skipgramBatchExec / cbowBatchExec).

    target / ng_starter: [B] ints; context: [B, ctxW] (CBOW, -1 padded) or None (skip-gram); indices / codes: [B, codeLen] Huffman points
    (-1 padded) and their codes, or None; alpha: [B] learning rates; rnd: [B] int64 LCG seeds; infer: optional [B, D] inference vectors
    (then only they are updated)."""
    dev = syn0.device
    B = int(target.numel())
    D, V = int(syn0.shape[1]), int(syn0.shape[0])

    def i32(t, cols=None):
        if t is None:
            return None
        t = torch.as_tensor(t, device=dev).to(torch.int32)
        return (t.reshape(B, -1) if cols is None else t.reshape(B, cols)).contiguous()

    tgt = torch.as_tensor(target, device=dev).to(torch.int32).reshape(-1).contiguous()
    ngs = torch.as_tensor(ng_starter, device=dev).to(torch.int32).reshape(-1).contiguous()
    ctx = i32(context)
    idx = i32(indices)
    cds = None if codes is None else torch.as_tensor(codes, device=dev).to(torch.int8).reshape(B, -1).contiguous()
    if idx is not None and (cds is None or cds.shape != idx.shape):
        raise ValueError("sg_cb_batch: codes must match indices")
    al = torch.as_tensor(alpha, device=dev, dtype=torch.float32).reshape(-1)
    al = (al.expand(B) if al.numel() == 1 else al).contiguous()
    rn = torch.as_tensor(rnd, device=dev).to(torch.int64).reshape(-1)
    rn = (rn.expand(B) if rn.numel() == 1 else rn).contiguous()
    nt = None if neg_table is None or neg_table.numel() == 0 else neg_table.to(device=dev, dtype=torch.int32).contiguous()
    for name, t in (("syn1", syn1), ("syn1neg", syn1neg)):
        if t is not None and not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == syn0.shape):
            raise ValueError("sg_cb_batch: %s must be a contiguous fp32 table shaped like syn0" % name)
    if infer is not None and not (infer.is_cuda and infer.dtype == torch.float32 and infer.is_contiguous() and infer.numel() == B * D):
        raise ValueError("sg_cb_batch: inference vectors must be contiguous fp32 [B, D]")
    if tgt.numel() != B or ngs.numel() != B or al.numel() != B or rn.numel() != B:
        raise ValueError("sg_cb_batch: per-item operands must have one entry per batch item")
    L.check(L.hip().dl4j_sg_cb_batch(L.ptr(syn0), L.ptr(syn1), L.ptr(syn1neg), L.ptr(nt), L.ptr(tgt), L.ptr(ngs), L.ptr(ctx), L.ptr(idx),
                                     L.ptr(cds), L.ptr(al), L.ptr(rn), L.ptr(infer), L.c_int(B), L.c_int(D), L.c_int(V),
                                     L.c_int(0 if idx is None else idx.shape[1]), L.c_int(0 if ctx is None else ctx.shape[1]),
                                     L.c_int(int(ns_rounds)), L.c_int(0 if nt is None else nt.numel()), L.c_int(1 if train_words else 0),
                                     L.c_int(0 if ctx is None else 1), L.stream()), "sg_cb_batch")
