"""ZooModel base (reference: deeplearning4j-zoo/.../zoo/ZooModel.java). Pretrained weights are
downloaded resources in the reference; there is no network here, so initPretrained() loads only
from an explicit local path."""
from __future__ import annotations


class ZooModel:
    def init(self):
        raise NotImplementedError

    def conf(self):
        raise NotImplementedError

    def initPretrained(self, path=None):
        if path is None:
            raise RuntimeError("Pretrained weights are not bundled (no network access); pass a local model zip path")
        from ..utils.model_serializer import ModelGuesser
        return ModelGuesser.loadModelGuess(path)

    def metaData(self):
        return {"inputShape": getattr(self, "inputShape", None)}

    @classmethod
    def builder(cls, **kw):
        return _ZooBuilder(cls, kw)


class _ZooBuilder:
    def __init__(self, cls, kw):
        self._cls, self._kw = cls, dict(kw)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)

        def setter(*v):
            self._kw[name] = v[0] if len(v) == 1 else list(v)
            return self
        return setter

    def build(self):
        return self._cls(**self._kw)
