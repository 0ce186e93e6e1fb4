"""Runtime graph vertices (reference: deeplearning4j-nn/.../nn/graph/vertex/impl/*.java).
Each vertex: forward(list of inputs) -> tensor, backward(eps) -> list of per-input epsilons."""
from __future__ import annotations

import torch

from ..conf import graph as G
from ..layers.base import AutogradLayerImpl


class VertexImpl:
    def __init__(self, conf, ctx):
        self.conf, self.ctx = conf, ctx
        self.params, self.grads, self.cparams = {}, {}, {}

    def param_specs(self):
        return []

    def compute_replicas(self):
        return []

    def set_views(self, p, g):
        self.params, self.grads = p, g

    def forward(self, xs, train=False):
        raise NotImplementedError

    def backward(self, eps):
        raise NotImplementedError

    def clear(self):
        pass


class MergeVertexImpl(VertexImpl):
    def forward(self, xs, train=False):
        if len(xs) == 1:
            self._sizes = None
            return xs[0]
        ax = self.conf.mergeAxis
        self._sizes = [x.shape[ax] for x in xs]
        y = torch.cat(xs, dim=ax)
        if y.dim() == 4 and y.is_cuda:
            y = y.contiguous(memory_format=torch.channels_last)
        return y

    def backward(self, eps):
        if self._sizes is None:
            return [eps]
        return list(torch.split(eps, self._sizes, dim=self.conf.mergeAxis))


class ElementWiseVertexImpl(VertexImpl):
    def forward(self, xs, train=False):
        op = self.conf.op
        self._n = len(xs)
        if op == "Add":
            y = xs[0]
            for x in xs[1:]:
                y = y + x
            return y
        if op == "Subtract":
            return xs[0] - xs[1]
        if op == "Product":
            self._xs = xs
            y = xs[0]
            for x in xs[1:]:
                y = y * x
            return y
        if op == "Average":
            return sum(xs[1:], xs[0]) / len(xs)
        if op == "Max":
            st = torch.stack(xs, 0)
            y, self._arg = st.max(0)
            return y
        raise ValueError(op)

    def backward(self, eps):
        op, n = self.conf.op, self._n
        if op == "Add":
            return [eps] * n
        if op == "Subtract":
            return [eps, -eps]
        if op == "Product":
            outs = []
            for i in range(n):
                g = eps
                for j, x in enumerate(self._xs):
                    if j != i:
                        g = g * x
                outs.append(g)
            return outs
        if op == "Average":
            return [eps / n] * n
        return [eps * (self._arg == i).to(eps.dtype) for i in range(n)]


class SubsetVertexImpl(VertexImpl):
    def forward(self, xs, train=False):
        x = xs[0]
        self._shape = x.shape
        return x[:, self.conf.from_:self.conf.to + 1]

    def backward(self, eps):
        out = torch.zeros(self._shape, dtype=eps.dtype, device=eps.device)
        out[:, self.conf.from_:self.conf.to + 1] = eps
        return [out]


class StackVertexImpl(VertexImpl):
    def forward(self, xs, train=False):
        self._sizes = [x.shape[0] for x in xs]
        return torch.cat(xs, dim=0)

    def backward(self, eps):
        return list(torch.split(eps, self._sizes, dim=0))


class UnstackVertexImpl(VertexImpl):
    def forward(self, xs, train=False):
        x = xs[0]
        self._shape = x.shape
        step = x.shape[0] // self.conf.stackSize
        self._sl = slice(self.conf.from_ * step, (self.conf.from_ + 1) * step)
        return x[self._sl]

    def backward(self, eps):
        out = torch.zeros(self._shape, dtype=eps.dtype, device=eps.device)
        out[self._sl] = eps
        return [out]


class ScaleVertexImpl(VertexImpl):
    def forward(self, xs, train=False):
        return xs[0] * self.conf.scaleFactor

    def backward(self, eps):
        return [eps * self.conf.scaleFactor]


class ShiftVertexImpl(VertexImpl):
    def forward(self, xs, train=False):
        return xs[0] + self.conf.shiftFactor

    def backward(self, eps):
        return [eps]


class L2VertexImpl(VertexImpl):
    def forward(self, xs, train=False):
        a, b = xs
        self._d = (a - b).reshape(a.shape[0], -1)
        self._shape = a.shape
        self._n = torch.sqrt((self._d * self._d).sum(1, keepdim=True) + self.conf.eps)
        return self._n

    def backward(self, eps):
        g = (eps * self._d / self._n).reshape(self._shape)
        return [g, -g]


class L2NormalizeVertexImpl(VertexImpl):
    def forward(self, xs, train=False):
        x = xs[0]
        dims = tuple(self.conf.dimension) if self.conf.dimension else tuple(range(1, x.dim()))
        self._x = x
        self._nrm = torch.sqrt((x * x).sum(dims, keepdim=True) + self.conf.eps)
        self._dims = dims
        return x / self._nrm

    def backward(self, eps):
        x, n = self._x, self._nrm
        dot = (eps * x).sum(self._dims, keepdim=True)
        return [eps / n - x * dot / (n ** 3)]


class ReshapeVertexImpl(VertexImpl):
    def forward(self, xs, train=False):
        self._shape = xs[0].shape
        s = list(self.conf.newShape)
        if s[0] <= 0 or s[0] != xs[0].shape[0]:
            s[0] = -1 if s.count(-1) == 0 else s[0]
        return xs[0].reshape(s)

    def backward(self, eps):
        return [eps.reshape(self._shape)]


class PreprocessorVertexImpl(VertexImpl):
    def forward(self, xs, train=False):
        self._mb = xs[0].shape[0]
        return self.conf.preProcessor.preProcess(xs[0], self._mb, train)

    def backward(self, eps):
        return [self.conf.preProcessor.backprop(eps, self._mb)]


class PoolHelperVertexImpl(VertexImpl):
    def forward(self, xs, train=False):
        self._shape = xs[0].shape
        return xs[0][:, :, 1:, 1:]

    def backward(self, eps):
        out = torch.zeros(self._shape, dtype=eps.dtype, device=eps.device)
        out[:, :, 1:, 1:] = eps
        return [out]


class LastTimeStepVertexImpl(VertexImpl):
    mask = None

    def forward(self, xs, train=False):
        x = xs[0]
        self._shape = x.shape
        m = self.mask
        if m is None:
            self._idx = None
            return x[:, :, -1]
        self._idx = (m.shape[1] - 1 - torch.flip(m > 0, dims=[1]).int().argmax(dim=1)).long()
        return x[torch.arange(x.shape[0], device=x.device), :, self._idx]

    def backward(self, eps):
        out = torch.zeros(self._shape, dtype=eps.dtype, device=eps.device)
        if self._idx is None:
            out[:, :, -1] = eps
        else:
            out[torch.arange(eps.shape[0], device=eps.device), :, self._idx] = eps
        return [out]


class DuplicateToTimeSeriesVertexImpl(VertexImpl):
    ref_T = 1

    def forward(self, xs, train=False):
        return xs[0].unsqueeze(2).expand(-1, -1, self.ref_T)

    def backward(self, eps):
        return [eps.sum(2)]


class ReverseTimeSeriesVertexImpl(VertexImpl):
    mask = None

    def _rev(self, x):
        m = self.mask
        if m is None:
            return torch.flip(x, dims=[2])
        T = x.shape[2]
        lens = (m > 0).sum(1).long()
        idx = torch.arange(T, device=x.device).unsqueeze(0).expand(x.shape[0], T)
        ridx = torch.where(idx < lens.unsqueeze(1), lens.unsqueeze(1) - 1 - idx, idx)
        return torch.gather(x, 2, ridx.unsqueeze(1).expand_as(x))

    def forward(self, xs, train=False):
        return self._rev(xs[0])

    def backward(self, eps):
        return [self._rev(eps)]


class FrozenVertexImpl(VertexImpl):
    def __init__(self, conf, ctx):
        super().__init__(conf, ctx)
        self.inner = create_vertex(conf.underlying, ctx)

    def forward(self, xs, train=False):
        return self.inner.forward(xs, False)

    def backward(self, eps):
        return self.inner.backward(eps)


class _AttnConf:
    """Adapter so AttentionVertex can reuse the autograd layer machinery."""
    idropout = None
    layerName = None

    def __init__(self, v):
        self.v = v

    def param_specs(self):
        from ..conf.layers import ParamSpec
        v = self.v
        if not v.projectInput:
            return []
        h, hs = v.nHeads, v.headSize
        return [ParamSpec("Wq", (h, hs, v.nInQueries), "c", v.nInQueries, hs, "W"), ParamSpec("Wk", (h, hs, v.nInKeys), "c", v.nInKeys, hs, "W"),
                ParamSpec("Wv", (h, hs, v.nInValues), "c", v.nInValues, hs, "W"), ParamSpec("Wo", (h * hs, v.nOut), "c", h * hs, v.nOut, "W")]

    weightInitFn = None

    def getUpdaterByParam(self, spec):
        return self._upd

    def getRegularizationByParam(self, spec):
        return []


class AttentionVertexImpl(VertexImpl):
    """Multi-head attention over (queries, keys, values) inputs (conf/graph/AttentionVertex.java)."""

    def param_specs(self):
        return _AttnConf(self.conf).param_specs()

    def forward(self, xs, train=False):
        from ..layers.extra import _mha
        q, k, v = xs
        if not train:
            with torch.no_grad():
                return _mha(q, k, v, self.params, self.conf.nHeads, None, self.conf.projectInput)
        self._in = [t.detach().requires_grad_(True) for t in xs]
        self._leaves = {k2: p.detach().requires_grad_(True) for k2, p in self.params.items()}
        with torch.enable_grad():
            self._y = _mha(self._in[0], self._in[1], self._in[2], self._leaves, self.conf.nHeads, None, self.conf.projectInput)
        return self._y.detach()

    def backward(self, eps):
        keys = list(self._leaves)
        gs = torch.autograd.grad(self._y, self._in + [self._leaves[k] for k in keys], eps, allow_unused=True)
        for k, g in zip(keys, gs[3:]):
            if g is not None:
                self.grads[k].copy_(g)
        return list(gs[:3])


class DotProductAttentionVertexImpl(VertexImpl):
    def forward(self, xs, train=False):
        from ...ops.attention import sdpa
        q, v = xs[0], xs[1]
        k = xs[2] if len(xs) > 2 else v
        self._in = [t.detach().requires_grad_(True) for t in xs]
        with torch.enable_grad():
            qq, kk, vv = (t.permute(0, 2, 1).unsqueeze(1) for t in (self._in[0], self._in[2] if len(xs) > 2 else self._in[1], self._in[1]))
            sc = self.conf.scaleFactor or None
            o = sdpa(qq, kk, vv, None, sc, self.conf.useCausalMask, self.conf.dropoutProbability, train)
            self._y = o.squeeze(1).permute(0, 2, 1)
        return self._y.detach()

    def backward(self, eps):
        return list(torch.autograd.grad(self._y, self._in, eps, allow_unused=True))


class SameDiffVertexImpl(VertexImpl):
    """Runs a user-defined SameDiffVertex / SameDiffLambdaVertex; gradients for inputs and parameters come from autograd."""

    def param_specs(self):
        return self.conf.param_specs()

    def forward(self, xs, train=False):
        if not train:
            with torch.no_grad():
                return self.conf.defineVertex(list(xs), dict(self.params))
        self._in = [t.detach().requires_grad_(True) if t.is_floating_point() else t for t in xs]
        self._leaves = {k: p.detach().requires_grad_(True) for k, p in self.params.items()}
        with torch.enable_grad():
            self._y = self.conf.defineVertex(list(self._in), self._leaves)
        return self._y.detach()

    def backward(self, eps):
        keys = list(self._leaves)
        diff_in = [t for t in self._in if t.requires_grad]
        gs = torch.autograd.grad(self._y, diff_in + [self._leaves[k] for k in keys], eps.to(self._y.dtype), allow_unused=True)
        for k, g in zip(keys, gs[len(diff_in):]):
            if g is None:
                self.grads[k].zero_()
            else:
                self.grads[k].copy_(g.reshape(self.grads[k].shape))
        it = iter(gs[:len(diff_in)])
        return [next(it) if t.requires_grad else None for t in self._in]


_MAP = {G.MergeVertex: MergeVertexImpl, G.ElementWiseVertex: ElementWiseVertexImpl, G.SubsetVertex: SubsetVertexImpl,
        G.StackVertex: StackVertexImpl, G.UnstackVertex: UnstackVertexImpl, G.ScaleVertex: ScaleVertexImpl,
        G.ShiftVertex: ShiftVertexImpl, G.L2Vertex: L2VertexImpl, G.L2NormalizeVertex: L2NormalizeVertexImpl,
        G.ReshapeVertex: ReshapeVertexImpl, G.PreprocessorVertex: PreprocessorVertexImpl,
        G.PoolHelperVertex: PoolHelperVertexImpl, G.LastTimeStepVertex: LastTimeStepVertexImpl,
        G.DuplicateToTimeSeriesVertex: DuplicateToTimeSeriesVertexImpl, G.ReverseTimeSeriesVertex: ReverseTimeSeriesVertexImpl,
        G.FrozenVertex: FrozenVertexImpl, G.AttentionVertex: AttentionVertexImpl,
        G.DotProductAttentionVertex: DotProductAttentionVertexImpl}


def create_vertex(conf, ctx):
    for k, v in _MAP.items():
        if type(conf) is k:
            return v(conf, ctx)
    if isinstance(conf, G.SameDiffVertex):
        return SameDiffVertexImpl(conf, ctx)
    raise ValueError("No runtime implementation for vertex %s" % type(conf).__name__)
