"""OmniHub analogue: a local model hub that stages foreign-framework models and converts them to this framework's
formats (reference: omnihub/ Java client + contrib/omnihub Python staging scripts ``omnihub/frameworks/{onnx,tensorflow,
keras,pytorch,huggingface}.py`` and ``model_hub.py``; ADRs 0011-0014). There is no network in the target environment, so
``download`` only resolves models already present in the cache directory; conversion is fully functional:

    hub = OmniHub(tmp)                       # default ~/.omnihub
    hub.stage("onnx", "add", "/path/add.onnx")
    sd  = hub.load("onnx", "add")            # converted on first use -> <cache>/samediff/add.fb
"""
from __future__ import annotations

import json
import os
import shutil
from pathlib import Path
from typing import Dict, List, Optional

FRAMEWORKS = ("onnx", "tensorflow", "keras", "samediff", "dl4j", "pytorch", "huggingface")


class OmniHub:
    def __init__(self, cacheDir: Optional[str] = None):
        self.root = Path(cacheDir or os.environ.get("OMNIHUB_HOME") or (Path.home() / ".omnihub"))
        for f in FRAMEWORKS:
            (self.root / f).mkdir(parents=True, exist_ok=True)
        self._index_path = self.root / "index.json"
        self._index: Dict[str, dict] = json.loads(self._index_path.read_text()) if self._index_path.exists() else {}

    # ------------------------------------------------------------------ staging
    def _save_index(self):
        self._index_path.write_text(json.dumps(self._index, indent=2, sort_keys=True))

    def stage(self, framework: str, name: str, path, **meta) -> Path:
        """Copy a model file / directory into the hub (what the reference's staging scripts do after downloading)."""
        if framework not in FRAMEWORKS:
            raise ValueError("Unknown framework %s (known: %s)" % (framework, ", ".join(FRAMEWORKS)))
        src = Path(path)
        dst = self.root / framework / (name + "".join(src.suffixes) if src.is_file() else name)
        if src.is_dir():
            if dst.exists():
                shutil.rmtree(dst)
            shutil.copytree(src, dst)
        else:
            shutil.copyfile(src, dst)
        self._index["%s/%s" % (framework, name)] = dict(meta, framework=framework, name=name, path=str(dst))
        self._save_index()
        return dst

    def listModels(self, framework: Optional[str] = None) -> List[str]:
        return sorted(k for k, v in self._index.items() if framework is None or v["framework"] == framework)

    def modelPath(self, framework: str, name: str) -> Path:
        k = "%s/%s" % (framework, name)
        if k not in self._index:
            raise FileNotFoundError("Model %s is not staged in %s. This environment has no network access: stage a local "
                                    "file with OmniHub.stage(framework, name, path)." % (k, self.root))
        return Path(self._index[k]["path"])

    def download(self, framework: str, name: str, url: Optional[str] = None) -> Path:
        """Resolves from the local cache only (no egress)."""
        return self.modelPath(framework, name)

    # ------------------------------------------------------------------ conversion
    def convert(self, framework: str, name: str, device=None, **kw):
        """Foreign model -> SameDiff (onnx / tensorflow) or MultiLayerNetwork / ComputationGraph (keras); result is cached."""
        p = self.modelPath(framework, name)
        if framework == "onnx":
            from .autodiff.import_onnx import OnnxImporter
            sd = OnnxImporter.importGraph(str(p), device=device, **kw)
        elif framework == "tensorflow":
            from .autodiff.import_tf import TFGraphMapper
            sd = TFGraphMapper.importGraph(str(p), device=device, **kw)
        elif framework == "keras":
            from .modelimport import KerasModelImport
            from .utils.model_serializer import ModelSerializer
            cfg = json.loads(Path(p).read_text()) if str(p).endswith(".json") else None
            if cfg is not None:
                seq = cfg.get("class_name") == "Sequential"
                weights = kw.get("weights")
                net = (KerasModelImport.importKerasSequentialModelAndWeights(str(p), weights) if seq
                       else KerasModelImport.importKerasModelAndWeights(str(p), weights)) if weights else None
                if net is None:
                    raise ValueError("keras conversion needs weights=<npz/h5 path> next to the JSON config")
            else:
                net = KerasModelImport.importKerasModelAndWeights(str(p))
            out = self.root / "dl4j" / (name + ".zip")
            ModelSerializer.writeModel(net, str(out), True)
            self._index["dl4j/%s" % name] = {"framework": "dl4j", "name": name, "path": str(out), "source": "%s/%s" % (framework, name)}
            self._save_index()
            return net
        elif framework in ("pytorch", "huggingface"):
            raise NotImplementedError("%s checkpoints: export to ONNX (torch.onnx.export) and stage the .onnx file" % framework)
        else:
            return self.load(framework, name, device=device)
        out = self.root / "samediff" / (name + ".fb")
        sd.save(str(out), False)
        self._index["samediff/%s" % name] = {"framework": "samediff", "name": name, "path": str(out), "source": "%s/%s" % (framework, name)}
        self._save_index()
        return sd

    def load(self, framework: str, name: str, device=None, **kw):
        if framework == "samediff":
            from .autodiff.samediff import SameDiff
            return SameDiff.load(str(self.modelPath(framework, name)), device=device)
        if framework == "dl4j":
            from .utils.model_serializer import ModelGuesser
            return ModelGuesser.loadModelGuess(str(self.modelPath(framework, name)))
        converted = "samediff/%s" % name if framework in ("onnx", "tensorflow") else "dl4j/%s" % name
        if converted in self._index and Path(self._index[converted]["path"]).exists():
            return self.load(converted.split("/")[0], name, device=device)
        return self.convert(framework, name, device=device, **kw)
