"""IActivation implementations (reference: nd4j-api/.../linalg/activations/impl/*, enum
org.nd4j.linalg.activations.Activation). Each has ``forward(z)`` (= getActivation) and
``backprop(z, eps)`` (= dL/dz given dL/da). Closed forms for the hot ones, autograd for the tail.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

_PKG = "org.nd4j.linalg.activations.impl."


class IActivation:
    JAVA_CLASS = None
    NAME = None

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def forward(self, z, training=False):
        raise NotImplementedError

    def backprop(self, z, eps):
        zz = z.detach().requires_grad_(True)
        with torch.enable_grad():
            y = self.forward(zz, True)
        return torch.autograd.grad(y, zz, eps)[0]

    # DL4J naming
    def getActivation(self, z, training=False):
        return self.forward(z, training)

    def params(self):
        return {}

    def to_json(self):
        d = {"@class": self.JAVA_CLASS}
        d.update(self.params())
        return d

    def __eq__(self, o):
        return type(self) is type(o) and self.params() == o.params()

    def __hash__(self):
        return hash((type(self).__name__, tuple(sorted(self.params().items()))))

    def __repr__(self):
        return "Activation." + str(self.NAME)


class ActivationIdentity(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationIdentity", "IDENTITY"

    def forward(self, z, training=False):
        return z

    def backprop(self, z, eps):
        return eps


class ActivationReLU(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationReLU", "RELU"

    def forward(self, z, training=False):
        return torch.relu(z)

    def backprop(self, z, eps):
        return eps * (z > 0).to(eps.dtype)


class ActivationReLU6(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationReLU6", "RELU6"

    def forward(self, z, training=False):
        return torch.clamp(z, 0.0, 6.0)

    def backprop(self, z, eps):
        return eps * ((z > 0) & (z < 6)).to(eps.dtype)


class ActivationLReLU(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationLReLU", "LEAKYRELU"
    alpha = 0.01

    def forward(self, z, training=False):
        return F.leaky_relu(z, self.alpha)

    def backprop(self, z, eps):
        return eps * torch.where(z >= 0, torch.ones_like(z), torch.full_like(z, self.alpha))

    def params(self):
        return {"alpha": self.alpha}


class ActivationELU(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationELU", "ELU"
    alpha = 1.0

    def forward(self, z, training=False):
        return F.elu(z, self.alpha)

    def params(self):
        return {"alpha": self.alpha}


class ActivationSELU(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationSELU", "SELU"

    def forward(self, z, training=False):
        return F.selu(z)


class ActivationSigmoid(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationSigmoid", "SIGMOID"

    def forward(self, z, training=False):
        return torch.sigmoid(z)

    def backprop(self, z, eps):
        s = torch.sigmoid(z)
        return eps * s * (1 - s)


class ActivationHardSigmoid(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationHardSigmoid", "HARDSIGMOID"

    def forward(self, z, training=False):
        return torch.clamp(0.2 * z + 0.5, 0.0, 1.0)


class ActivationTanH(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationTanH", "TANH"

    def forward(self, z, training=False):
        return torch.tanh(z)

    def backprop(self, z, eps):
        t = torch.tanh(z)
        return eps * (1 - t * t)


class ActivationHardTanH(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationHardTanH", "HARDTANH"

    def forward(self, z, training=False):
        return torch.clamp(z, -1.0, 1.0)

    def backprop(self, z, eps):
        return eps * ((z >= -1) & (z <= 1)).to(eps.dtype)


class ActivationRationalTanh(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationRationalTanh", "RATIONALTANH"

    def forward(self, z, training=False):
        # 1.7159 * tanh_approx(2x/3), tanh_approx(y) = sgn(y) * (1 - 1/(1+|y|+y^2+1.41645*y^4))
        y = 2.0 * z / 3.0
        a = y.abs()
        return 1.7159 * torch.sign(y) * (1.0 - 1.0 / (1.0 + a + y * y + 1.41645 * y ** 4))


class ActivationRectifiedTanh(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationRectifiedTanh", "RECTIFIEDTANH"

    def forward(self, z, training=False):
        return torch.clamp_min(torch.tanh(z), 0.0)


class ActivationSoftmax(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationSoftmax", "SOFTMAX"

    def forward(self, z, training=False):
        # the HIP kernel has no autograd node: inside a recorded graph (generic loss gradients, SameDiff sessions) stay on ATen
        if z.dim() == 2 and not (z.requires_grad and torch.is_grad_enabled()):
            from ..ops.softmax import softmax
            return softmax(z)
        return torch.softmax(z, dim=1)

    def backprop(self, z, eps):
        p = self.forward(z.detach())
        if z.dim() == 2:
            from ..ops.softmax import softmax_bp
            return softmax_bp(p, eps.contiguous())
        return p * (eps - (eps * p).sum(1, keepdim=True))


class ActivationSoftPlus(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationSoftPlus", "SOFTPLUS"

    def forward(self, z, training=False):
        return F.softplus(z)

    def backprop(self, z, eps):
        return eps * torch.sigmoid(z)


class ActivationSoftSign(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationSoftSign", "SOFTSIGN"

    def forward(self, z, training=False):
        return z / (1 + z.abs())

    def backprop(self, z, eps):
        d = 1 + z.abs()
        return eps / (d * d)


class ActivationCube(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationCube", "CUBE"

    def forward(self, z, training=False):
        return z * z * z

    def backprop(self, z, eps):
        return eps * 3 * z * z


class ActivationSwish(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationSwish", "SWISH"

    def forward(self, z, training=False):
        return z * torch.sigmoid(z)


class ActivationMish(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationMish", "MISH"

    def forward(self, z, training=False):
        return z * torch.tanh(F.softplus(z))


class ActivationGELU(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationGELU", "GELU"
    precise = False

    def forward(self, z, training=False):
        if self.precise:
            return 0.5 * z * (1 + torch.tanh(math.sqrt(2 / math.pi) * (z + 0.044715 * z ** 3)))
        # libnd4j GELU = x * sigmoid(1.702 x)
        return z * torch.sigmoid(1.702 * z)

    def params(self):
        return {"precise": self.precise}


class ActivationThresholdedReLU(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationThresholdedReLU", "THRESHOLDEDRELU"
    theta = 1.0

    def forward(self, z, training=False):
        return torch.where(z > self.theta, z, torch.zeros_like(z))

    def backprop(self, z, eps):
        return eps * (z > self.theta).to(eps.dtype)

    def params(self):
        return {"theta": self.theta}


class ActivationRReLU(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationRReLU", "RRELU"
    l, u = 1.0 / 8, 1.0 / 3

    def forward(self, z, training=False):
        if training:
            self._alpha = torch.empty_like(z).uniform_(self.l, self.u)
        else:
            self._alpha = torch.full_like(z, (self.l + self.u) / 2)
        return torch.where(z >= 0, z, z * self._alpha)

    def backprop(self, z, eps):
        return eps * torch.where(z >= 0, torch.ones_like(z), self._alpha)

    def params(self):
        return {"l": self.l, "u": self.u}


class ActivationPReLU(IActivation):
    JAVA_CLASS, NAME = _PKG + "ActivationPReLU", "PRELU"
    alpha = None

    def forward(self, z, training=False):
        return torch.where(z >= 0, z, z * self.alpha)


_ALL = [ActivationIdentity, ActivationReLU, ActivationReLU6, ActivationLReLU, ActivationELU, ActivationSELU,
        ActivationSigmoid, ActivationHardSigmoid, ActivationTanH, ActivationHardTanH, ActivationRationalTanh,
        ActivationRectifiedTanh, ActivationSoftmax, ActivationSoftPlus, ActivationSoftSign, ActivationCube,
        ActivationSwish, ActivationMish, ActivationGELU, ActivationThresholdedReLU, ActivationRReLU]
BY_NAME = {c.NAME: c for c in _ALL}
BY_CLASS = {c.JAVA_CLASS: c for c in _ALL + [ActivationPReLU]}


class Activation:
    """Enum facade: ``Activation.RELU`` etc. resolve to fresh IActivation instances."""
    CUBE = "CUBE"; ELU = "ELU"; HARDSIGMOID = "HARDSIGMOID"; HARDTANH = "HARDTANH"; IDENTITY = "IDENTITY"
    LEAKYRELU = "LEAKYRELU"; RATIONALTANH = "RATIONALTANH"; RELU = "RELU"; RELU6 = "RELU6"; RRELU = "RRELU"
    SIGMOID = "SIGMOID"; SOFTMAX = "SOFTMAX"; SOFTPLUS = "SOFTPLUS"; SOFTSIGN = "SOFTSIGN"; TANH = "TANH"
    RECTIFIEDTANH = "RECTIFIEDTANH"; SELU = "SELU"; SWISH = "SWISH"; THRESHOLDEDRELU = "THRESHOLDEDRELU"
    GELU = "GELU"; MISH = "MISH"

    @staticmethod
    def values():
        return list(BY_NAME.keys())


def get_activation(a) -> IActivation:
    if a is None:
        return ActivationIdentity()
    if isinstance(a, IActivation):
        return a
    if isinstance(a, dict):
        cls = BY_CLASS[a["@class"]]
        kw = {k: v for k, v in a.items() if k != "@class"}
        return cls(**kw)
    return BY_NAME[str(a).upper()]()
