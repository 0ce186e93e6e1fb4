"""ConvolutionalIterationListener (reference: deeplearning4j-ui-parent/deeplearning4j-ui/src/main/java/org/deeplearning4j/ui/weights/
ConvolutionalIterationListener.java + deeplearning4j-ui-model/.../weights/ConvolutionListenerPersistable.java): every ``frequency``
iterations the activations of the convolutional layers for the first example of the minibatch are tiled into one grey-scale image (one
row of tiles per layer, one tile per channel, min-max normalised per tile) and posted to the stats storage as a PNG; the UI's
ConvolutionalListenerModule serves the latest one.

The PNG is written with zlib + struct only (8-bit grey, no filter), so no imaging library is needed; the tiling itself runs on the device
and a single uint8 image crosses to the host."""
from __future__ import annotations

import base64
import os
import platform
import struct
import time
import uuid
import zlib
from typing import List, Optional

import torch

from ..optimize.listeners import TrainingListener
from .stats import Persistable

TYPE_ID = "ConvolutionalListener"


def encode_png_gray(img: torch.Tensor) -> bytes:
    """uint8 [H, W] tensor -> PNG bytes (colour type 0, bit depth 8)."""
    a = img.detach().to("cpu", torch.uint8).contiguous()
    H, W = a.shape
    raw = b"".join(b"\x00" + bytes(a[r].tolist()) for r in range(H))

    def chunk(tag, data):
        c = tag + data
        return struct.pack(">I", len(data)) + c + struct.pack(">I", zlib.crc32(c) & 0xFFFFFFFF)

    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


def decode_png_gray(data: bytes) -> torch.Tensor:
    """Inverse of encode_png_gray for files it wrote (filter type 0 only); used by the tests."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, W, H = 8, b"", 0, 0
    while pos < len(data):
        n, = struct.unpack(">I", data[pos:pos + 4])
        tag = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            W, H = struct.unpack(">II", body[:8])
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = zlib.decompress(idat)
    rows = [list(raw[r * (W + 1) + 1:(r + 1) * (W + 1)]) for r in range(H)]
    return torch.tensor(rows, dtype=torch.uint8)


def _tt(v):
    """INDArray / tensor -> tensor."""
    t = getattr(v, "tensor", None)
    return v if t is None else (t() if callable(t) else t)


def tile_activations(acts: List[torch.Tensor], max_channels=64, border=1) -> torch.Tensor:
    """List of [C, H, W] activation maps -> one uint8 image: a row of per-channel tiles for each layer."""
    rows = []
    for a in acts:
        a = a[:max_channels].float()
        C, H, W = a.shape
        lo = a.amin((1, 2), keepdim=True)
        hi = a.amax((1, 2), keepdim=True)
        t = ((a - lo) / (hi - lo).clamp_min(1e-12) * 255.0).round().to(torch.uint8)
        t = torch.nn.functional.pad(t, [border, border, border, border], value=0)
        rows.append(t.permute(1, 0, 2).reshape(H + 2 * border, C * (W + 2 * border)))
    width = max(r.shape[1] for r in rows)
    rows = [torch.nn.functional.pad(r, [0, width - r.shape[1]], value=0) for r in rows]
    return torch.cat(rows, 0)


class ConvolutionalIterationListener(TrainingListener):
    def __init__(self, router, frequency=10, sessionID=None, workerID=None, maxChannels=64, probeInput=None):
        self.router, self.freq, self.maxChannels = router, max(int(frequency), 1), int(maxChannels)
        self.probeInput = probeInput          # example(s) to visualise; default: the network's current input (setInput / last fit batch)
        self.sessionID = sessionID or str(uuid.uuid4())
        self.workerID = workerID or "%s_%d" % (platform.node() or "worker", os.getpid())
        self._acts: Optional[list] = None
        self._meta_sent = False

    def getSessionID(self):
        return self.sessionID

    def onForwardPass(self, model, activations):
        """``activations``: list (MultiLayerNetwork) or dict name -> tensor (ComputationGraph) of this iteration's layer outputs."""
        items = activations.items() if isinstance(activations, dict) else enumerate(activations)
        self._acts = [(str(k), _tt(v)) for k, v in items if torch.is_tensor(_tt(v)) and _tt(v).dim() == 4]

    def iterationDone(self, model, iteration, epoch):
        if iteration % self.freq != 0:
            return
        acts = self._acts
        if acts is None:                       # the network does not push activations to listeners: run the stored input forward
            x = self.probeInput if self.probeInput is not None else getattr(model, "_input", None)
            if x is None or not hasattr(model, "feedForward"):
                return
            x = _tt(x)
            outs = model.feedForward(x[:1] if torch.is_tensor(x) else x, False)
            items = outs.items() if isinstance(outs, dict) else enumerate(outs)
            acts = [(str(k), _tt(v)) for k, v in items]
            acts = [(k, v) for k, v in acts if torch.is_tensor(v) and v.dim() == 4]
        if not acts:
            return
        img = tile_activations([a[0] for _, a in acts], self.maxChannels)
        now = int(time.time() * 1000)
        if not self._meta_sent:
            self.router.putStorageMetaData(Persistable(self.sessionID, TYPE_ID, self.workerID, now, {"updateTypeClass": "ConvolutionListenerPersistable"}))
            self._meta_sent = True
        self.router.putUpdate(Persistable(self.sessionID, TYPE_ID, self.workerID, now,
                                          {"iteration": int(iteration), "layers": [k for k, _ in acts], "width": int(img.shape[1]),
                                           "height": int(img.shape[0]), "png": base64.b64encode(encode_png_gray(img)).decode()}))
        self._acts = None
