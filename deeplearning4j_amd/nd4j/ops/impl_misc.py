"""Custom-op implementations, part 3: linalg, images, segment / parity ops, TensorArray lists, NLP
(skipgram / cbow), t-SNE, control-flow markers, decoder, misc.

Reference contracts: libnd4j ops/declarable/headers/{parity_ops,images,list,nlp,tsne,decoder,util,
compat,strings,third_party,tests}.h and generic/**. Dense linalg goes to rocSOLVER/hipBLAS through
ATen (the reference: cuSOLVER gesvd/gesvdj + hand-rolled LU, helpers/cuda/{svd,lup}.cu).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .custom import _IMPL, op
from .impl_core import _unbroadcast, bp_of, dtype_arg, first

# ----------------------------------------------------------------------------- linalg
op("matrix_determinant")(lambda ins, i, t, b, d: torch.linalg.det(ins[0]))
op("log_matrix_determinant")(lambda ins, i, t, b, d: tuple(torch.linalg.slogdet(ins[0])))
op("logdet")(lambda ins, i, t, b, d: torch.linalg.slogdet(ins[0])[1])
op("matrix_inverse")(lambda ins, i, t, b, d: torch.linalg.inv(ins[0]))
op("cholesky")(lambda ins, i, t, b, d: torch.linalg.cholesky(ins[0]))
op("sqrtm")(lambda ins, i, t, b, d: _sqrtm(ins[0]))
op("trace")(lambda ins, i, t, b, d: torch.diagonal(ins[0], dim1=-2, dim2=-1).sum(-1))
op("cross")(lambda ins, i, t, b, d: torch.linalg.cross(ins[0], ins[1], dim=-1))
op("diag")(lambda ins, i, t, b, d: torch.diag_embed(ins[0].reshape(-1)).reshape(list(ins[0].shape) * 2) if ins[0].dim() > 1 else torch.diag(ins[0]))
op("diag_part")(lambda ins, i, t, b, d: torch.diagonal(ins[0].reshape(int(math.isqrt(ins[0].numel())), -1)).reshape(
    ins[0].shape[:ins[0].dim() // 2]))
op("matrix_diag")(lambda ins, i, t, b, d: torch.diag_embed(ins[0]))
op("matrix_diag_part")(lambda ins, i, t, b, d: torch.diagonal(ins[0], dim1=-2, dim2=-1))
op("lgamma")(lambda ins, i, t, b, d: torch.lgamma(ins[0]))
op("digamma")(lambda ins, i, t, b, d: torch.digamma(ins[0]))
op("polygamma")(lambda ins, i, t, b, d: _polygamma(ins[0], ins[1]))
op("zeta")(lambda ins, i, t, b, d: torch.special.zeta(ins[0], ins[1]))


def _polygamma(n, x):
    n, x = torch.broadcast_tensors(n, x)
    out = torch.empty_like(x)
    for k in torch.unique(n.long()).tolist():
        m = n.long() == k
        out[m] = torch.polygamma(int(k), x[m])
    return out


def _sqrtm(a):
    if torch.allclose(a, a.transpose(-1, -2), atol=1e-6):
        w, v = torch.linalg.eigh(a)
        return (v * torch.sqrt(w.clamp_min(0)).unsqueeze(-2)) @ v.transpose(-1, -2)
    w, v = torch.linalg.eig(a)
    return (v @ torch.diag_embed(torch.sqrt(w)) @ torch.linalg.inv(v)).real.to(a.dtype)


@op("betainc")
def _betainc(ins, i, t, b, d):
    """Regularized incomplete beta I_x(a,b) by Lentz continued fraction (vectorised, fp64)."""
    a, bb, x = [v.double() for v in torch.broadcast_tensors(ins[0], ins[1], ins[2])]
    swap = x > (a + 1) / (a + bb + 2)
    aa, b2, xx = torch.where(swap, bb, a), torch.where(swap, a, bb), torch.where(swap, 1 - x, x)
    lbeta = torch.lgamma(aa) + torch.lgamma(b2) - torch.lgamma(aa + b2)
    front = torch.exp(aa * torch.log(xx.clamp_min(1e-300)) + b2 * torch.log((1 - xx).clamp_min(1e-300)) - lbeta) / aa
    tiny = 1e-300
    f, c, dd = torch.ones_like(xx), torch.ones_like(xx), torch.zeros_like(xx)
    for m in range(0, 300):
        if m == 0:
            num = torch.ones_like(xx)
        elif m % 2 == 0:
            k = m // 2
            num = k * (b2 - k) * xx / ((aa + 2 * k - 1) * (aa + 2 * k))
        else:
            k = (m - 1) // 2
            num = -(aa + k) * (aa + b2 + k) * xx / ((aa + 2 * k) * (aa + 2 * k + 1))
        dd = 1 + num * dd
        dd = torch.where(dd.abs() < tiny, torch.full_like(dd, tiny), dd)
        dd = 1 / dd
        c = 1 + num / c
        c = torch.where(c.abs() < tiny, torch.full_like(c, tiny), c)
        f = f * c * dd
    r = front * (f - 1)
    r = torch.where(swap, 1 - r, r)
    r = torch.where(x <= 0, torch.zeros_like(r), torch.where(x >= 1, torch.ones_like(r), r))
    return r.to(ins[2].dtype if ins[2].is_floating_point() else torch.float32)


@op("eye")
def _eye(ins, i, t, b, d):
    if ins and not i:
        vals = [int(v) for v in ins[0].reshape(-1).tolist()]
    else:
        vals = list(i[1:]) if len(i) > 1 else list(i)
    rows = vals[0]
    cols = vals[1] if len(vals) > 1 else rows
    batch = vals[2:]
    dt = dtype_arg(d, None, 0, torch.float32)
    e = torch.eye(rows, cols, dtype=dt)
    return e.expand(*batch, rows, cols).clone() if batch else e


op("tri")(lambda ins, i, t, b, d: torch.tril(torch.ones(i[0], i[1] if len(i) > 1 else i[0], dtype=dtype_arg(d, None, 0, torch.float32)),
                                              diagonal=i[2] if len(i) > 2 else 0))
op("triu")(lambda ins, i, t, b, d: torch.triu(ins[0], diagonal=i[0] if i else 0))
op("triu_bp")(lambda ins, i, t, b, d: torch.triu(ins[1], diagonal=i[0] if i else 0))


@op("matrix_band_part")
def _band(ins, i, t, b, d):
    x = ins[0]
    lo = int(i[0]) if i else int(ins[1])
    hi = int(i[1]) if len(i) > 1 else int(ins[2])
    r, c = x.shape[-2], x.shape[-1]
    ii = torch.arange(r, device=x.device).unsqueeze(1)
    jj = torch.arange(c, device=x.device).unsqueeze(0)
    m = ((lo < 0) | (ii - jj <= lo)) & ((hi < 0) | (jj - ii <= hi))
    return x * m.to(x.dtype)


@op("matrix_set_diag")
def _set_diag(ins, i, t, b, d):
    x = ins[0].clone()
    torch.diagonal(x, dim1=-2, dim2=-1).copy_(ins[1])
    return x


@op("lu")
def _lu(ins, i, t, b, d):
    LU, piv = torch.linalg.lu_factor(ins[0])
    # permutation vector (0-based) like the reference's second output
    n = ins[0].shape[-1]
    perm = torch.arange(n, device=ins[0].device).expand(list(ins[0].shape[:-2]) + [n]).clone()
    pv = (piv - 1).long()
    for k in range(pv.shape[-1]):
        a = perm[..., k].clone()
        idx = pv[..., k:k + 1]
        bsel = torch.gather(perm, -1, idx).squeeze(-1)
        perm[..., k] = bsel
        perm.scatter_(-1, idx, a.unsqueeze(-1))
    return LU, perm.to(dtype_arg(d, None, 0, torch.int32))


@op("qr")
def _qr(ins, i, t, b, d):
    full = bool(b[0]) if b else (bool(i[0]) if i else False)
    return tuple(torch.linalg.qr(ins[0], mode="complete" if full else "reduced"))


@op("svd")
def _svd(ins, i, t, b, d):
    full_uv, calc_uv = (bool(i[0]) if i else False), (bool(i[1]) if len(i) > 1 else True)
    if not calc_uv:
        return torch.linalg.svdvals(ins[0])
    U, S, Vh = torch.linalg.svd(ins[0], full_matrices=full_uv)
    return S, U, Vh.transpose(-1, -2)


@op("eig")
def _eig(ins, i, t, b, d):
    w, v = torch.linalg.eig(ins[0])
    return torch.stack([w.real, w.imag], -1), torch.stack([v.real, v.imag], -1)


@op("solve")
def _solve(ins, i, t, b, d):
    a = ins[0].transpose(-1, -2) if (b and b[0]) or (i and i[0]) else ins[0]
    return torch.linalg.solve(a, ins[1])


@op("triangular_solve")
def _tri_solve(ins, i, t, b, d):
    lower = bool(b[0]) if b else (bool(i[0]) if i else True)
    adj = bool(b[1]) if len(b) > 1 else (bool(i[1]) if len(i) > 1 else False)
    a = ins[0].transpose(-1, -2) if adj else ins[0]
    return torch.linalg.solve_triangular(a, ins[1], upper=(not lower) != adj if adj else not lower)


@op("lstsq", "solve_ls")
def _lstsq(ins, i, t, b, d):
    a, y = ins[0], ins[1]
    l2 = t[0] if t else 0.0
    fast = bool(b[0]) if b else True
    if fast or l2 != 0.0:
        at = a.transpose(-1, -2)
        n = a.shape[-1]
        return torch.linalg.solve(at @ a + l2 * torch.eye(n, dtype=a.dtype, device=a.device), at @ y)
    return torch.linalg.lstsq(a, y).solution


# ----------------------------------------------------------------------------- images
def _ch_last(x, i, idx=0):
    dim = (int(i[idx]) if len(i) > idx else -1) % x.dim()
    return x.movedim(dim, -1), dim


@op("rgb_to_grs")
def _rgb_to_grs(ins, i, t, b, d):
    x, dim = _ch_last(ins[0], i)
    g = 0.2989 * x[..., 0] + 0.5870 * x[..., 1] + 0.1140 * x[..., 2]
    return g.unsqueeze(-1).movedim(-1, dim)


def _mat_color(m):
    def f(ins, i, t, b, d):
        x, dim = _ch_last(ins[0], i)
        M = torch.tensor(m, dtype=x.dtype, device=x.device)
        return (x @ M.t()).movedim(-1, dim)
    return f


_YIQ = [[0.299, 0.587, 0.114], [0.59590059, -0.27455667, -0.32134392], [0.21153661, -0.52273617, 0.31119955]]
_YUV = [[0.299, 0.587, 0.114], [-0.14714119, -0.28886916, 0.43601035], [0.61497538, -0.51496512, -0.10001026]]
op("rgb_to_yiq")(_mat_color(_YIQ))
op("yiq_to_rgb")(_mat_color(torch.linalg.inv(torch.tensor(_YIQ, dtype=torch.float64)).tolist()))
op("rgb_to_yuv")(_mat_color(_YUV))
op("yuv_to_rgb")(_mat_color(torch.linalg.inv(torch.tensor(_YUV, dtype=torch.float64)).tolist()))


def _rgb2hsv(x):
    r, g, bl = x[..., 0], x[..., 1], x[..., 2]
    mx, mn = x.amax(-1), x.amin(-1)
    dl = mx - mn
    s = torch.where(mx > 0, dl / mx.clamp_min(1e-30), torch.zeros_like(mx))
    dz = dl.clamp_min(1e-30)
    h = torch.where(mx == r, (g - bl) / dz, torch.where(mx == g, 2 + (bl - r) / dz, 4 + (r - g) / dz))
    h = torch.where(dl == 0, torch.zeros_like(h), (h / 6) % 1.0)
    return torch.stack([h, s, mx], -1)


def _hsv2rgb(x):
    h, s, v = x[..., 0], x[..., 1], x[..., 2]
    k = lambda n: (n + h * 6) % 6
    f = lambda n: v - v * s * torch.clamp(torch.minimum(k(n), 4 - k(n)), 0, 1)
    return torch.stack([f(5), f(3), f(1)], -1)


op("rgb_to_hsv")(lambda ins, i, t, b, d: (lambda x, dim: _rgb2hsv(x).movedim(-1, dim))(*_ch_last(ins[0], i)))
op("hsv_to_rgb")(lambda ins, i, t, b, d: (lambda x, dim: _hsv2rgb(x).movedim(-1, dim))(*_ch_last(ins[0], i)))


@op("adjust_hue")
def _adjust_hue(ins, i, t, b, d):
    x, dim = _ch_last(ins[0], i)
    delta = t[0] if t else float(ins[1])
    hsv = _rgb2hsv(x)
    hsv = torch.stack([(hsv[..., 0] + delta) % 1.0, hsv[..., 1], hsv[..., 2]], -1)
    return _hsv2rgb(hsv).movedim(-1, dim)


@op("adjust_saturation")
def _adjust_sat(ins, i, t, b, d):
    x, dim = _ch_last(ins[0], i)
    f = t[0] if t else float(ins[1])
    hsv = _rgb2hsv(x)
    hsv = torch.stack([hsv[..., 0], (hsv[..., 1] * f).clamp(0, 1), hsv[..., 2]], -1)
    return _hsv2rgb(hsv).movedim(-1, dim)


@op("adjust_contrast", "adjust_contrast_v2")
def _adjust_contrast(ins, i, t, b, d):
    x = ins[0]
    f = t[0] if t else float(ins[1])
    m = x.mean((-3, -2), keepdim=True)
    return (x - m) * f + m


def _resize(mode):
    def f(ins, i, t, b, d):
        x = ins[0]                        # NHWC
        if len(ins) > 1:
            h, w = [int(v) for v in ins[1].reshape(-1).tolist()]
        else:
            h, w = int(i[0]), int(i[1])
        align = bool(b[0]) if b else False
        half = bool(b[1]) if len(b) > 1 else False
        squeeze = x.dim() == 3
        xc = (x.unsqueeze(0) if squeeze else x).permute(0, 3, 1, 2)
        xf = xc.float() if not xc.is_floating_point() else xc
        if mode == "nearest":
            H, W = xf.shape[2], xf.shape[3]
            sh, sw = ((H - 1) / max(h - 1, 1), (W - 1) / max(w - 1, 1)) if align and h > 1 and w > 1 else (H / h, W / w)
            off = 0.5 if half else 0.0
            ih = torch.clamp(((torch.arange(h, device=x.device) + off) * sh).round().long() if align else
                             torch.floor((torch.arange(h, device=x.device) + off) * sh).long(), 0, H - 1)
            iw = torch.clamp(((torch.arange(w, device=x.device) + off) * sw).round().long() if align else
                             torch.floor((torch.arange(w, device=x.device) + off) * sw).long(), 0, W - 1)
            y = xf[:, :, ih][:, :, :, iw]
        elif mode in ("area", "bicubic"):
            # TensorFlow semantics (quantised Keys cubic taps / exact fractional footprint), not torch's: ops/misc_kernels.py
            from ...ops.misc_kernels import resize_tf_reference
            y = resize_tf_reference(xf.permute(0, 2, 3, 1), h, w, mode, align, half)
            return y.squeeze(0) if squeeze else y
        elif mode == "bilinear" and not align and not half:
            # TF1 legacy sampling: src = dst * scale (no half-pixel shift)
            H, W = xf.shape[2], xf.shape[3]
            ys = torch.arange(h, device=x.device, dtype=xf.dtype) * (H / h)
            xs = torch.arange(w, device=x.device, dtype=xf.dtype) * (W / w)
            y0, x0 = ys.floor().long().clamp(0, H - 1), xs.floor().long().clamp(0, W - 1)
            y1, x1 = (y0 + 1).clamp(0, H - 1), (x0 + 1).clamp(0, W - 1)
            wy, wx = (ys - y0).reshape(1, 1, h, 1), (xs - x0).reshape(1, 1, 1, w)
            g = lambda yy, xx: xf[:, :, yy][:, :, :, xx]
            y = (g(y0, x0) * (1 - wy) * (1 - wx) + g(y0, x1) * (1 - wy) * wx + g(y1, x0) * wy * (1 - wx) + g(y1, x1) * wy * wx)
        else:
            y = F.interpolate(xf, (h, w), mode=mode, align_corners=align if mode != "nearest" else None)
        y = y.permute(0, 2, 3, 1)
        if mode in ("nearest",):
            y = y.to(x.dtype)
        return y.squeeze(0) if squeeze else y
    return f


op("resize_bilinear")(_resize("bilinear"))
op("resize_nearest_neighbor")(_resize("nearest"))
op("resize_bicubic")(_resize("bicubic"))
op("resize_area")(_resize("area"))
op("resize_images")(lambda ins, i, t, b, d: _resize({0: "bilinear", 1: "nearest", 2: "bicubic", 3: "area"}.get(int(i[-1]) if i else 0, "bilinear"))(ins, i, t, b, d))
op("image_resize")(lambda ins, i, t, b, d: _resize({0: "bilinear", 1: "nearest", 2: "bicubic", 3: "area", 4: "bicubic", 5: "bilinear"}.get(
    int(i[0]) if i else 0, "bilinear"))(ins, list(i[1:]), t, b, d))


@op("crop_and_resize")
def _crop_and_resize(ins, i, t, b, d):
    img, boxes, box_idx, size = ins[0], ins[1].float(), ins[2].long(), [int(v) for v in ins[3].tolist()]
    method = "bilinear" if (not i or i[0] == 0) else "nearest"
    extrap = t[0] if t else 0.0
    ch, cw = size
    H, W = img.shape[1], img.shape[2]
    outs = []
    for k in range(boxes.shape[0]):
        y1, x1, y2, x2 = boxes[k].tolist()
        ys = torch.linspace(y1 * (H - 1), y2 * (H - 1), ch, device=img.device) if ch > 1 else torch.tensor([0.5 * (y1 + y2) * (H - 1)], device=img.device)
        xs = torch.linspace(x1 * (W - 1), x2 * (W - 1), cw, device=img.device) if cw > 1 else torch.tensor([0.5 * (x1 + x2) * (W - 1)], device=img.device)
        gy = ys / max(H - 1, 1) * 2 - 1
        gx = xs / max(W - 1, 1) * 2 - 1
        grid = torch.stack(torch.meshgrid(gx, gy, indexing="xy"), -1).unsqueeze(0)
        src = img[int(box_idx[k])].permute(2, 0, 1).unsqueeze(0).float()
        o = F.grid_sample(src, grid, mode=method, padding_mode="zeros", align_corners=True)
        inside = ((gy.abs() <= 1).reshape(-1, 1) & (gx.abs() <= 1).reshape(1, -1)).reshape(1, 1, ch, cw)
        o = torch.where(inside, o, torch.full_like(o, extrap))
        outs.append(o.squeeze(0).permute(1, 2, 0))
    return torch.stack(outs) if outs else img.new_zeros(0, ch, cw, img.shape[3])


@op("extract_image_patches")
def _extract_patches(ins, i, t, b, d):
    x = ins[0]                              # NHWC
    kH, kW, sH, sW, rH, rW = i[:6]
    same = bool(i[6]) if len(i) > 6 else False
    xc = x.permute(0, 3, 1, 2)
    if same:
        from .impl_nn import _same_pad
        ph, pw = _same_pad(xc.shape[2], kH, sH, rH), _same_pad(xc.shape[3], kW, sW, rW)
        xc = F.pad(xc, [pw[0], pw[1], ph[0], ph[1]])
    n, c = xc.shape[0], xc.shape[1]
    oH = (xc.shape[2] - (kH - 1) * rH - 1) // sH + 1
    oW = (xc.shape[3] - (kW - 1) * rW - 1) // sW + 1
    cols = F.unfold(xc.float(), (kH, kW), (rH, rW), 0, (sH, sW)).reshape(n, c, kH * kW, oH, oW)
    return cols.permute(0, 3, 4, 2, 1).reshape(n, oH, oW, kH * kW * c).to(x.dtype)


@op("draw_bounding_boxes")
def _draw_bb(ins, i, t, b, d):
    img, boxes, colors = ins[0].clone(), ins[1], ins[2]
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
                img[n, r1, c1c:c2c + 1] = col
            if r2 < H:
                img[n, r2, c1c:c2c + 1] = col
            if c1 >= 0:
                img[n, r1c:r2c + 1, c1] = col
            if c2 < W:
                img[n, r1c:r2c + 1, c2] = col
    return img


def _iou(boxes):
    y1, x1 = torch.minimum(boxes[:, 0], boxes[:, 2]), torch.minimum(boxes[:, 1], boxes[:, 3])
    y2, x2 = torch.maximum(boxes[:, 0], boxes[:, 2]), torch.maximum(boxes[:, 1], boxes[:, 3])
    area = (y2 - y1) * (x2 - x1)
    iy = (torch.minimum(y2[:, None], y2[None]) - torch.maximum(y1[:, None], y1[None])).clamp_min(0)
    ix = (torch.minimum(x2[:, None], x2[None]) - torch.maximum(x1[:, None], x1[None])).clamp_min(0)
    inter = iy * ix
    union = area[:, None] + area[None] - inter
    return torch.where(union > 0, inter / union.clamp_min(1e-30), torch.zeros_like(inter))


def _nms_select(overlaps, scores, max_out, thr, score_thr):
    order = torch.argsort(scores, descending=True, stable=True).tolist()
    keep = []
    for k in order:
        if len(keep) >= max_out:
            break
        if float(scores[k]) <= score_thr:
            continue
        if all(float(overlaps[k, j]) <= thr for j in keep):
            keep.append(k)
    return torch.tensor(keep, dtype=torch.int32, device=scores.device)


@op("non_max_suppression", "non_max_suppression_v3")
def _nms(ins, i, t, b, d):
    boxes, scores = ins[0].float(), ins[1].float()
    max_out = int(i[0]) if i else int(ins[2])
    thr = t[0] if t else (float(ins[3]) if len(ins) > 3 else 0.5)
    sthr = t[1] if len(t) > 1 else (float(ins[4]) if len(ins) > 4 else float("-inf"))
    return _nms_select(_iou(boxes), scores, max_out, thr, sthr)


@op("non_max_suppression_overlaps")
def _nms_overlaps(ins, i, t, b, d):
    max_out = int(i[0]) if i else int(ins[2])
    thr = t[0] if t else 0.5
    sthr = t[1] if len(t) > 1 else float("-inf")
    return _nms_select(ins[0].float(), ins[1].float(), max_out, thr, sthr)


# ----------------------------------------------------------------------------- segment ops
def _segment(kind, sorted_ids):
    def f(ins, i, t, b, d):
        x, ids = ins[0], ins[1].long()
        n = (int(i[0]) if i else int(ins[2])) if not sorted_ids else (int(ids.max()) + 1 if ids.numel() else 0)
        flat = x.reshape(x.shape[0], -1)
        ix = ids.reshape(-1, 1).expand_as(flat)
        if kind in ("sum", "mean", "sqrt_n"):
            out = torch.zeros(n, flat.shape[1], dtype=x.dtype, device=x.device).scatter_add_(0, ix, flat)
            if kind != "sum":
                cnt = torch.bincount(ids, minlength=n).clamp_min(1).to(x.dtype).unsqueeze(1)
                out = out / (cnt if kind == "mean" else torch.sqrt(cnt))
        else:
            red = {"max": "amax", "min": "amin", "prod": "prod"}[kind]
            if kind == "prod":
                init = torch.ones(n, flat.shape[1], dtype=x.dtype, device=x.device)
            else:
                big = torch.finfo(x.dtype).max if x.is_floating_point() else torch.iinfo(x.dtype).max
                init = torch.full((n, flat.shape[1]), -big if kind == "max" else big, dtype=x.dtype, device=x.device)
            out = init.scatter_reduce(0, ix, flat, reduce=red, include_self=True)
        return out.reshape([n] + list(x.shape[1:]))
    return f


for _k in ("sum", "mean", "max", "min", "prod"):
    op("segment_" + _k)(_segment(_k, True))
    op("unsorted_segment_" + _k)(_segment(_k, False))
    op("segment_%s_bp" % _k)(bp_of("segment_" + _k, 1))
    op("unsorted_segment_%s_bp" % _k)(bp_of("unsorted_segment_" + _k, 1))
op("unsorted_segment_sqrt_n")(_segment("sqrt_n", False))
op("unsorted_segment_sqrt_n_bp")(bp_of("unsorted_segment_sqrt_n", 1))


# ----------------------------------------------------------------------------- parity ops
@op("Assert")
def _assert(ins, i, t, b, d):
    if ins[0].device.type != "meta" and not bool((ins[0] != 0).all()):
        raise AssertionError("Assert op failed")
    return ins[0]


op("bincount")(lambda ins, i, t, b, d: (torch.bincount(ins[0].long().reshape(-1), weights=ins[1].reshape(-1) if len(ins) > 1 and ins[1].numel() == ins[0].numel() else None,
                                                       minlength=int(i[0]) if i else 0)[:(int(i[1]) if len(i) > 1 and i[1] > 0 else None)]))


@op("broadcast_dynamic_shape")
def _bds(ins, i, t, b, d):
    return torch.tensor(list(torch.broadcast_shapes(tuple(int(v) for v in ins[0].tolist()), tuple(int(v) for v in ins[1].tolist()))),
                        dtype=ins[0].dtype)


@op("broadcastgradientargs")
def _bga(ins, i, t, b, d):
    a, bb = [int(v) for v in ins[0].tolist()], [int(v) for v in ins[1].tolist()]
    r = max(len(a), len(bb))
    pa, pb = [1] * (r - len(a)) + a, [1] * (r - len(bb)) + bb
    ra = [k for k in range(r) if pa[k] == 1 and pb[k] != 1 or k < r - len(a)]
    rb = [k for k in range(r) if pb[k] == 1 and pa[k] != 1 or k < r - len(bb)]
    return torch.tensor(ra, dtype=torch.int64), torch.tensor(rb, dtype=torch.int64)


@op("check_numerics")
def _check_numerics(ins, i, t, b, d):
    if ins[0].device.type != "meta" and not bool(torch.isfinite(ins[0]).all()):
        raise FloatingPointError("check_numerics: tensor has NaN or Inf")
    return ins[0]


@op("compare_and_bitpack")
def _cab(ins, i, t, b, d):
    x, thr = ins[0], ins[1]
    bits = (x > thr).reshape(list(x.shape[:-1]) + [x.shape[-1] // 8, 8]).to(torch.uint8)
    w = torch.tensor([128, 64, 32, 16, 8, 4, 2, 1], dtype=torch.uint8, device=x.device)
    return (bits * w).sum(-1).to(torch.uint8)


@op("confusion_matrix")
def _confusion(ins, i, t, b, d):
    lab, pred = ins[0].long().reshape(-1), ins[1].long().reshape(-1)
    n = int(i[0]) if i else int(max(lab.max(), pred.max())) + 1
    w = ins[2].reshape(-1) if len(ins) > 2 else None
    dt = dtype_arg(d, None, 0, w.dtype if w is not None else torch.int64)
    return torch.bincount(lab * n + pred, weights=w, minlength=n * n).reshape(n, n).to(dt)


op("expose", "Expose")(lambda ins, i, t, b, d: tuple(x for x in ins))
# Eager forms of the dataflow control ops (generic/flow/flow_control_ops.cpp). Inside a graph they are interpreted by the frame
# executors (autodiff/samediff.py _run_graph, native rt_graph.cpp); executed on their own they are the value-passing identities.
op("Merge")(lambda ins, i, t, b, d: next(x for x in ins if x is not None))
op("LoopCond")(lambda ins, i, t, b, d: ins[0])
op("NextIteration")(lambda ins, i, t, b, d: ins[0])
op("Exit")(lambda ins, i, t, b, d: ins[0])


def _fake_quant(x, mn, mx, num_bits, narrow):
    qmin = 1.0 if narrow else 0.0
    qmax = float(2 ** num_bits - 1)
    scale = (mx - mn) / (qmax - qmin)
    zp = qmin - mn / scale
    nudged_zp = torch.where(zp < qmin, torch.full_like(zp, qmin), torch.where(zp > qmax, torch.full_like(zp, qmax), torch.round(zp)))
    nmin, nmax = (qmin - nudged_zp) * scale, (qmax - nudged_zp) * scale
    c = torch.minimum(torch.maximum(x, nmin), nmax)
    return torch.floor((c - nmin) / scale + 0.5) * scale + nmin


op("fake_quant_with_min_max_vars")(lambda ins, i, t, b, d: _fake_quant(
    ins[0], ins[1] if len(ins) > 1 else torch.tensor(t[0]), ins[2] if len(ins) > 2 else torch.tensor(t[1]),
    int(i[0]) if i else 8, bool(b[0]) if b else (bool(i[1]) if len(i) > 1 else False)))
op("fake_quant_with_min_max_vars_per_channel")(lambda ins, i, t, b, d: _fake_quant(
    ins[0], ins[1], ins[2], int(i[0]) if i else 8, bool(b[0]) if b else (bool(i[1]) if len(i) > 1 else False)))


@op("in_top_k")
def _in_top_k(ins, i, t, b, d):
    pred, tgt = ins[0], ins[1].long()
    k = int(i[0])
    tv = pred.gather(1, tgt.reshape(-1, 1))
    return (pred > tv).sum(1) < k


@op("top_k")
def _top_k(ins, i, t, b, d):
    k = int(i[0]) if i else int(ins[1])
    srt = bool(b[0]) if b else True
    v, ix = torch.topk(ins[0], k, dim=-1, sorted=True)
    return v, ix.to(dtype_arg(d, None, 0, torch.int32) if d else torch.int64)


@op("nth_element")
def _nth(ins, i, t, b, d):
    n = int(ins[1])
    rev = bool(i[0]) if i else False
    s = torch.sort(ins[0], dim=-1, descending=rev).values
    return s[..., n]


@op("listdiff")
def _listdiff(ins, i, t, b, d):
    x, y = ins[0], ins[1]
    m = ~torch.isin(x, y)
    return x[m], torch.nonzero(m).reshape(-1).to(torch.int32)


@op("onehot")
def _onehot(ins, i, t, b, d):
    idx = ins[0].long()
    axis = int(i[0]) if i else -1
    depth = int(i[1]) if len(i) > 1 else int(ins[1])
    on = t[0] if t else (float(ins[2]) if len(ins) > 2 else 1.0)
    off = t[1] if len(t) > 1 else (float(ins[3]) if len(ins) > 3 else 0.0)
    dt = dtype_arg(d, None, 0, torch.float32)
    valid = (idx >= 0) & (idx < depth)
    oh = F.one_hot(idx.clamp(0, depth - 1), depth).bool() & valid.unsqueeze(-1)
    r = torch.where(oh, torch.tensor(on, dtype=dt, device=idx.device), torch.tensor(off, dtype=dt, device=idx.device))
    return r if axis in (-1, idx.dim()) else r.movedim(-1, axis)


@op("sequence_mask")
def _seq_mask(ins, i, t, b, d):
    lens = ins[0].long()
    mx = int(i[0]) if i else (int(ins[1]) if len(ins) > 1 else int(lens.max()))
    return (torch.arange(mx, device=lens.device) < lens.unsqueeze(-1)).to(dtype_arg(d, None, 0, torch.bool))


@op("unique")
def _unique(ins, i, t, b, d):
    x = ins[0].reshape(-1)
    u, inv = torch.unique(x, return_inverse=True)
    # TF semantics: order of first occurrence
    first_pos = torch.full((u.numel(),), x.numel(), dtype=torch.int64, device=x.device).scatter_reduce(
        0, inv, torch.arange(x.numel(), device=x.device), reduce="amin")
    order = torch.argsort(first_pos)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(order.numel(), device=x.device)
    return u[order], rank[inv]


@op("unique_with_counts")
def _unique_counts(ins, i, t, b, d):
    vals, idx = _unique(ins, i, t, b, d)
    return vals, idx, torch.bincount(idx, minlength=vals.numel())


op("noop")(lambda ins, i, t, b, d: ())
op("invoke")(lambda ins, i, t, b, d: tuple(ins))
op("print_variable")(lambda ins, i, t, b, d: (print(ins[0]), ins[0])[1])
op("print_affinity")(lambda ins, i, t, b, d: (print("device:", ins[0].device), ins[0])[1])
op("testop2i2o")(lambda ins, i, t, b, d: (ins[0] + 1.0, ins[1] + 2.0))
op("testcustom")(lambda ins, i, t, b, d: ins[0].clone())
op("test_scalar")(lambda ins, i, t, b, d: ins[0] + 2)
op("test_output_reshape")(lambda ins, i, t, b, d: ins[0].reshape(-1))
op("testreduction")(lambda ins, i, t, b, d: ins[0].sum([k % ins[0].dim() for k in i]) if i else ins[0].sum())


@op("knn_mindistance")
def _knn_mindist(ins, i, t, b, d):
    x, lo, hi = ins[0], ins[1], ins[2]
    dl = torch.maximum(torch.maximum(lo - x, x - hi), torch.zeros_like(x))
    return torch.sqrt((dl * dl).sum())


@op("firas_sparse")
def _firas_sparse(ins, i, t, b, d):
    x = ins[0]
    cols = torch.tensor(list(i), dtype=torch.int64, device=x.device)
    return x.index_select(1, cols)


# ----------------------------------------------------------------------------- TensorArray list ops
class NDArrayList:
    _nd4j_opaque = True
    """TensorArray used by the list ops (reference: array/NDArrayList.h)."""

    def __init__(self, height=0, expandable=True):
        self.items = {}
        self.height = height
        self.expandable = expandable

    def write(self, idx, arr):
        self.items[int(idx)] = arr
        return self

    def read(self, idx):
        return self.items[int(idx)]

    def stack(self):
        return torch.stack([self.items[k] for k in sorted(self.items)])

    def size(self):
        return max(len(self.items), (max(self.items) + 1) if self.items else 0)

    def clone(self):
        n = NDArrayList(self.height, self.expandable)
        n.items = {k: v.clone() for k, v in self.items.items()}
        return n


def _list_of(ins):
    for x in ins:
        if isinstance(x, NDArrayList):
            return x
    raise TypeError("list op needs an NDArrayList input")


op("create_list")(lambda ins, i, t, b, d: (NDArrayList(int(i[0]) if i else 0, bool(i[1]) if len(i) > 1 else True),))
op("write_list")(lambda ins, i, t, b, d: (_list_of(ins).write(int(i[0]) if i else int(ins[2]), ins[1]),))
op("read_list")(lambda ins, i, t, b, d: _list_of(ins).read(int(i[0]) if i else int(ins[1])))
op("stack_list")(lambda ins, i, t, b, d: _list_of(ins).stack())
op("size_list")(lambda ins, i, t, b, d: torch.tensor(_list_of(ins).size(), dtype=torch.int32))
op("clone_list")(lambda ins, i, t, b, d: (_list_of(ins).clone(),))
op("pick_list")(lambda ins, i, t, b, d: torch.stack([_list_of(ins).read(k) for k in (i if i else ins[1].tolist())]))
op("gather_list")(lambda ins, i, t, b, d: torch.stack([_list_of(ins).read(int(k)) for k in ins[1].tolist()]))


@op("unstack_list")
def _unstack_list(ins, i, t, b, d):
    lst = ins[0] if isinstance(ins[0], NDArrayList) else NDArrayList()
    src = ins[1] if isinstance(ins[0], NDArrayList) else ins[0]
    for k in range(src.shape[0]):
        lst.write(k, src[k])
    return (lst,)


@op("scatter_list")
def _scatter_list(ins, i, t, b, d):
    lst, idx, arr = _list_of(ins), ins[1], ins[2]
    for k, ix in enumerate(idx.tolist()):
        lst.write(int(ix), arr[k])
    return (lst,)


@op("split_list")
def _split_list(ins, i, t, b, d):
    lst, arr, sizes = _list_of(ins), ins[1], ins[2].tolist()
    off = 0
    for k, s in enumerate(sizes):
        lst.write(k, arr[off:off + int(s)])
        off += int(s)
    return (lst,)


@op("delete_list")
def _delete_list(ins, i, t, b, d):
    lst = _list_of(ins)
    lst.items.pop(int(i[0]) if i else int(ins[1]), None)
    return (lst,)


# ----------------------------------------------------------------------------- control-flow markers
# Switch/Merge/Enter/Exit/NextIteration/LoopCond are interpreted by the graph sessions
# (autodiff/samediff/session.py); as standalone ops they have their dataflow meaning.
op("Switch")(lambda ins, i, t, b, d: (ins[0], ins[0]))
op("While", "Scope", "Conditional", "Return")(lambda ins, i, t, b, d: tuple(ins))


# ----------------------------------------------------------------------------- NLP: skipgram / cbow
_SG_MAX_EXP = 6.0


def _sg_step(syn0, syn1, syn1neg, exp_table, neg_table, target, ngram_starter, indices, codes, alpha, rnd, n_neg, neu1=None,
             update_syn0=True):
    """One skip-gram/cbow training step on rows of syn0 (reference: helpers/cuda/sg_cb.cu hSoftmax_/nSampling_).
    ``neu1``: input vector (syn0[target] for skip-gram, context mean for cbow). Returns the accumulated error neu1e."""
    l1 = syn0[target] if neu1 is None else neu1
    neu1e = torch.zeros_like(l1)
    if indices is not None and len(indices):
        idx = torch.as_tensor(indices, dtype=torch.int64, device=syn0.device)
        cd = torch.as_tensor(codes, dtype=syn0.dtype, device=syn0.device)
        rows = syn1[idx]
        dot = rows @ l1
        f = torch.sigmoid(dot)
        g = (1 - cd - f) * alpha
        g = torch.where((dot < -_SG_MAX_EXP) | (dot >= _SG_MAX_EXP), torch.zeros_like(g), g)      # reference skips saturated code points
        neu1e += g @ rows
        syn1.index_add_(0, idx, g.unsqueeze(1) * l1.unsqueeze(0))
    if n_neg > 0 and syn1neg is not None:
        tg = [ngram_starter]
        r = int(rnd)
        V = syn1neg.shape[0]
        for _ in range(n_neg):
            r = (r * 25214903917 + 11) & 0xFFFFFFFFFFFFFFFF
            if neg_table is not None and neg_table.numel():
                cand = int(neg_table[(r >> 16) % neg_table.numel()])
            else:
                cand = (r >> 16) % V
            if cand == ngram_starter:
                continue
            tg.append(cand)
        idx = torch.as_tensor(tg, dtype=torch.int64, device=syn0.device)
        lab = torch.zeros(len(tg), dtype=syn0.dtype, device=syn0.device)
        lab[0] = 1
        rows = syn1neg[idx]
        dot = rows @ l1
        f = torch.where(dot > _SG_MAX_EXP, torch.ones_like(dot), torch.where(dot < -_SG_MAX_EXP, torch.zeros_like(dot), torch.sigmoid(dot)))
        g = (lab - f) * alpha
        neu1e += g @ rows
        syn1neg.index_add_(0, idx, g.unsqueeze(1) * l1.unsqueeze(0))
    if update_syn0 and neu1 is None:
        syn0[target] += neu1e
    return neu1e


@op("skipgram")
def _skipgram(ins, i, t, b, d):
    """inputs: target, ngStarter, indices, codes, syn0, syn1, syn1neg, expTable, negTable, alpha, randomValue,
    [inferenceVector]; iArgs [numWorkers, nsRounds]; bArgs [isInference, isPreciseMode]. Updates tables in place."""
    target, ngs, indices, codes, syn0, syn1, syn1neg, expT, negT, alpha, rnd = ins[:11]
    n_neg = int(i[1]) if len(i) > 1 else 0
    if target.dim() == 0 or target.numel() == 1 and indices.dim() <= 1:
        idx = [int(v) for v in indices.reshape(-1).tolist() if v >= 0]
        cds = [int(v) for v in codes.reshape(-1).tolist()][:len(idx)]
        _sg_step(syn0, syn1 if syn1.numel() else None, syn1neg if syn1neg.numel() else None, expT, negT, int(target), int(ngs), idx, cds,
                 float(alpha), int(rnd), n_neg)
    else:
        for k in range(target.numel()):
            idx = [int(v) for v in indices[k].tolist() if v >= 0]
            cds = [int(v) for v in codes[k].tolist()][:len(idx)]
            _sg_step(syn0, syn1 if syn1.numel() else None, syn1neg if syn1neg.numel() else None, expT, negT, int(target.reshape(-1)[k]),
                     int(ngs.reshape(-1)[k]), idx, cds, float(alpha.reshape(-1)[k] if alpha.numel() > 1 else alpha),
                     int(rnd.reshape(-1)[k] if rnd.numel() > 1 else rnd), n_neg)
    return syn0, syn1, syn1neg


op("skipgram_inference")(lambda ins, i, t, b, d: _skipgram(ins, i, t, b, d))


@op("cbow", "cbow_inference")
def _cbow(ins, i, t, b, d):
    """inputs: target, ngStarter, context, indices, codes, syn0, syn1, syn1neg, expTable, negTable, alpha, randomValue,
    numLabels, lockedWords, [inferenceVector]; iArgs [numWorkers, nsRounds]; bArgs [trainWords, isInference]."""
    target, ngs, context, indices, codes, syn0, syn1, syn1neg, expT, negT, alpha, rnd = ins[:12]
    n_neg = int(i[1]) if len(i) > 1 else 0
    train_words = bool(b[0]) if b else True
    ctx = [int(v) for v in context.reshape(-1).tolist() if v >= 0]
    if not ctx:
        return syn0, syn1, syn1neg
    cidx = torch.as_tensor(ctx, dtype=torch.int64, device=syn0.device)
    neu1 = syn0[cidx].mean(0)
    idx = [int(v) for v in indices.reshape(-1).tolist() if v >= 0]
    cds = [int(v) for v in codes.reshape(-1).tolist()][:len(idx)]
    neu1e = _sg_step(syn0, syn1 if syn1.numel() else None, syn1neg if syn1neg.numel() else None, expT, negT, int(target), int(ngs), idx, cds,
                     float(alpha), int(rnd), n_neg, neu1=neu1)
    if train_words:
        syn0.index_add_(0, cidx, neu1e.unsqueeze(0).expand(len(ctx), -1))
    return syn0, syn1, syn1neg


# ----------------------------------------------------------------------------- t-SNE helpers
@op("barnes_symmetrized")
def _barnes_sym(ins, i, t, b, d):
    rowP, colP, valP = ins[0].long(), ins[1].long(), ins[2]
    N = int(i[0]) if i else rowP.numel() - 1
    rows = torch.repeat_interleave(torch.arange(N, device=valP.device), rowP[1:] - rowP[:-1])
    P = torch.sparse_coo_tensor(torch.stack([rows, colP[:rows.numel()]]), valP[:rows.numel()], (N, N)).coalesce()
    S = ((P + P.t()) * 0.5).coalesce()
    r, c = S.indices()
    out_rows = torch.zeros(N + 1, dtype=torch.int32, device=valP.device)
    out_rows[1:] = torch.cumsum(torch.bincount(r, minlength=N), 0).to(torch.int32)
    return out_rows, c.to(torch.int32), S.values()


@op("barnes_edge_forces")
def _barnes_edge(ins, i, t, b, d):
    rowP, colP, valP, data = ins[0].long(), ins[1].long(), ins[2], ins[3]
    N = int(i[0]) if i else data.shape[0]
    rows = torch.repeat_interleave(torch.arange(N, device=data.device), rowP[1:N + 1] - rowP[:N])
    cols = colP[:rows.numel()]
    diff = data[rows] - data[cols]
    w = valP[:rows.numel()] / (1 + (diff * diff).sum(1))
    return torch.zeros_like(data).index_add_(0, rows, diff * w.unsqueeze(1))


@op("barnes_gains")
def _barnes_gains(ins, i, t, b, d):
    gains, gradx, eps = ins[0], ins[1], ins[2]
    same = torch.sign(gradx) == torch.sign(eps)
    return torch.where(same, gains * 0.8, gains + 0.2).clamp_min(0.01)


@op("cell_contains")
def _cell_contains(ins, i, t, b, d):
    corner, width, point = ins[0], ins[1], ins[2]
    return torch.tensor(bool(((corner - width <= point) & (point <= corner + width)).all()))


# ----------------------------------------------------------------------------- decoder / strings / compat
@op("ctc_beam")
def _ctc_beam(ins, i, t, b, d):
    """logits [B, T, C] (log-probs or logits), seqLen [B]; iArgs [blankIndex, beamWidth, nbest]; bArgs [normalizeLogits].
    Prefix beam search; returns (paths [B, nbest, T] padded with -1... here blank-padded with 0 length info), logProbs, lengths."""
    logits, seq_len = ins[0].float().cpu(), ins[1].long().cpu()
    blank = int(i[0]) if i else 0
    width = int(i[1]) if len(i) > 1 else 10
    nbest = int(i[2]) if len(i) > 2 else 1
    lp = torch.log_softmax(logits, -1) if (not b or b[0]) else logits
    B, T, C = lp.shape
    paths = torch.zeros(B, nbest, T, dtype=torch.int32)
    scores = torch.full((B, nbest), float("-inf"))
    lens = torch.zeros(B, nbest, dtype=torch.int32)
    NEG = float("-inf")

    def lse(a, bb):
        if a == NEG:
            return bb
        if bb == NEG:
            return a
        m = max(a, bb)
        return m + math.log(math.exp(a - m) + math.exp(bb - m))

    for bi in range(B):
        beams = {(): (0.0, NEG)}      # prefix -> (p_blank, p_nonblank)
        for tt in range(int(seq_len[bi])):
            row = lp[bi, tt].tolist()
            nxt = {}
            for pre, (pb, pnb) in beams.items():
                tot = lse(pb, pnb)
                e = nxt.get(pre, (NEG, NEG))
                nxt[pre] = (lse(e[0], tot + row[blank]), e[1])
                for c in range(C):
                    if c == blank:
                        continue
                    np_ = pre + (c,)
                    e = nxt.get(np_, (NEG, NEG))
                    add = (pb + row[c]) if (pre and pre[-1] == c) else (tot + row[c])
                    nxt[np_] = (e[0], lse(e[1], add))
                    if pre and pre[-1] == c:
                        e2 = nxt.get(pre, (NEG, NEG))
                        nxt[pre] = (e2[0], lse(e2[1], pnb + row[c]))
            beams = dict(sorted(nxt.items(), key=lambda kv: -lse(*kv[1]))[:width])
        best = sorted(beams.items(), key=lambda kv: -lse(*kv[1]))[:nbest]
        for k, (pre, pr) in enumerate(best):
            paths[bi, k, :len(pre)] = torch.tensor(pre, dtype=torch.int32) if pre else paths[bi, k, :0]
            scores[bi, k] = lse(*pr)
            lens[bi, k] = len(pre)
    return paths, scores, lens


class StringArray:
    """Minimal UTF8 array stand-in for the string ops (reference: NDArray UTF8/16/32 dtypes)."""

    def __init__(self, values, shape=None):
        self.values = list(values)
        self.shape = list(shape) if shape is not None else [len(self.values)]

    def to(self, *a, **k):          # strings live on the host whatever device the numeric operands are on
        return self

    def __eq__(self, other):
        return isinstance(other, StringArray) and self.values == other.values and self.shape == other.shape

    __hash__ = None


@op("split_string")
def _split_string(ins, i, t, b, d):
    arr, delim = ins[0], ins[1]
    dl = delim.values[0] if isinstance(delim, StringArray) else str(delim)
    out = []
    for s in arr.values:
        out += s.split(dl)
    return (StringArray(out),)


@op("compat_string_split")
def _compat_string_split(ins, i, t, b, d):
    arr, delim = ins[0], ins[1]
    dl = delim.values[0] if isinstance(delim, StringArray) else str(delim)
    idx, vals = [], []
    for r, s in enumerate(arr.values):
        for c, tok in enumerate(s.split(dl)):
            idx.append([r, c])
            vals.append(tok)
    return torch.tensor(idx, dtype=torch.int64).reshape(-1, 2), StringArray(vals)


@op("compat_sparse_to_dense")
def _sparse_to_dense(ins, i, t, b, d):
    idx, shape, vals = ins[0].long(), [int(v) for v in ins[1].tolist()], ins[2]
    default = ins[3] if len(ins) > 3 else None
    if isinstance(vals, StringArray):
        flat = [default.values[0] if isinstance(default, StringArray) else ""] * int(math.prod(shape))
        for k, ix in enumerate(idx.reshape(len(vals.values), -1).tolist()):
            off = 0
            for dd, v in zip(shape, ix):
                off = off * dd + v
            flat[off] = vals.values[k]
        return (StringArray(flat, shape),)
    out = torch.full(shape, float(default) if default is not None else 0, dtype=vals.dtype, device=vals.device)
    out[tuple(idx.reshape(vals.numel(), -1).t())] = vals.reshape(-1)
    return out
