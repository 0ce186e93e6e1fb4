"""Image loading + augmentation (reference: datavec-data-image/.../image/loader/{NativeImageLoader,ImageLoader}.java,
recordreader/{ImageRecordReader,objdetect/ObjectDetectionRecordReader}.java, transform/*ImageTransform.java,
datavec-api io/labels/{ParentPathLabelGenerator,PatternPathLabelGenerator}.java).

The reference decodes through JavaCV/OpenCV; here PIL decodes on the host and the tensor math (resize is done by PIL, flips /
crops / colour ops on tensors) stays in torch, NCHW float32 like NativeImageLoader.asMatrix."""
from __future__ import annotations

import os
import random
import re
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from .records import FileSplit, InputSplit, Record, RecordMetaData, RecordReader
from .writable import IntWritable, NDArrayWritable, Text, Writable

ALLOWED_FORMATS = ("tif", "tiff", "jpg", "jpeg", "png", "bmp", "gif", "ppm", "pgm", "webp")


# ----------------------------------------------------------------------------- label generators
class PathLabelGenerator:
    def getLabelForPath(self, path: str) -> Writable:
        raise NotImplementedError

    def inferLabelClasses(self):
        return True


class ParentPathLabelGenerator(PathLabelGenerator):
    def getLabelForPath(self, path):
        return Text(os.path.basename(os.path.dirname(os.fspath(path))))


class PatternPathLabelGenerator(PathLabelGenerator):
    def __init__(self, pattern, patternPosition=0):
        self.pattern, self.pos = pattern, patternPosition

    def getLabelForPath(self, path):
        return Text(re.split(self.pattern, os.path.basename(os.fspath(path)))[self.pos])


# ----------------------------------------------------------------------------- transforms
class ImageTransform:
    def transform(self, img: torch.Tensor, rng: random.Random) -> torch.Tensor:
        """img: float32 [C, H, W]"""
        raise NotImplementedError


class FlipImageTransform(ImageTransform):
    """flipMode 0 = around x axis (vertical flip), 1 = around y axis (horizontal), -1 = both; None = random choice."""

    def __init__(self, flipMode: Optional[int] = None):
        self.mode = flipMode

    def transform(self, img, rng):
        m = self.mode if self.mode is not None else rng.choice([0, 1, -1, None])
        if m is None:
            return img
        dims = {0: [1], 1: [2], -1: [1, 2]}[m]
        return img.flip(dims)


class CropImageTransform(ImageTransform):
    def __init__(self, top=0, left=None, bottom=None, right=None, random_=True):
        self.top = top
        self.left = top if left is None else left
        self.bottom = top if bottom is None else bottom
        self.right = self.left if right is None else right
        self.random = random_

    def transform(self, img, rng):
        t, l, b, r = ((rng.randint(0, v) if (self.random and v > 0) else v) for v in (self.top, self.left, self.bottom, self.right))
        H, W = img.shape[1:]
        return img[:, t:H - b, l:W - r]


class RandomCropTransform(ImageTransform):
    def __init__(self, height, width, seed=None):
        self.h, self.w = int(height), int(width)

    def transform(self, img, rng):
        H, W = img.shape[1:]
        y, x = rng.randint(0, max(H - self.h, 0)), rng.randint(0, max(W - self.w, 0))
        return img[:, y:y + self.h, x:x + self.w]


class ResizeImageTransform(ImageTransform):
    def __init__(self, newWidth, newHeight):
        self.w, self.h = int(newWidth), int(newHeight)

    def transform(self, img, rng):
        return torch.nn.functional.interpolate(img.unsqueeze(0), size=(self.h, self.w), mode="bilinear", align_corners=False)[0]


class ScaleImageTransform(ImageTransform):
    def __init__(self, dx, dy=None):
        self.dx, self.dy = dx, dx if dy is None else dy

    def transform(self, img, rng):
        H, W = img.shape[1:]
        nh = max(1, int(round(H + rng.uniform(-self.dy, self.dy))))
        nw = max(1, int(round(W + rng.uniform(-self.dx, self.dx))))
        return torch.nn.functional.interpolate(img.unsqueeze(0), size=(nh, nw), mode="bilinear", align_corners=False)[0]


class RotateImageTransform(ImageTransform):
    def __init__(self, angle, centerx=0, centery=0, scale=0):
        self.angle = float(angle)

    def transform(self, img, rng):
        a = np.deg2rad(rng.uniform(-self.angle, self.angle))
        theta = torch.tensor([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0]], dtype=img.dtype).unsqueeze(0)
        grid = torch.nn.functional.affine_grid(theta, [1] + list(img.shape), align_corners=False)
        return torch.nn.functional.grid_sample(img.unsqueeze(0), grid, align_corners=False)[0]


class WarpImageTransform(RotateImageTransform):
    def __init__(self, delta):
        self.delta = float(delta)

    def transform(self, img, rng):
        H, W = img.shape[1:]
        d = self.delta
        theta = torch.eye(2, 3, dtype=img.dtype)
        theta += torch.tensor([[rng.uniform(-d, d) / W for _ in range(3)], [rng.uniform(-d, d) / H for _ in range(3)]], dtype=img.dtype)
        grid = torch.nn.functional.affine_grid(theta.unsqueeze(0), [1] + list(img.shape), align_corners=False)
        return torch.nn.functional.grid_sample(img.unsqueeze(0), grid, align_corners=False)[0]


class ColorConversionTransform(ImageTransform):
    """BGR/RGB -> grayscale (the one conversion that changes tensor semantics without OpenCV colour tables)."""

    def __init__(self, conversionCode="GRAY"):
        self.code = conversionCode

    def transform(self, img, rng):
        if img.shape[0] == 3 and str(self.code).upper().endswith("GRAY"):
            w = torch.tensor([0.114, 0.587, 0.299], dtype=img.dtype).view(3, 1, 1)      # channels are BGR like OpenCV
            return (img * w).sum(0, keepdim=True)
        return img


class EqualizeHistTransform(ImageTransform):
    def transform(self, img, rng):
        out = []
        for c in img:
            v = c.clamp(0, 255).round().long().reshape(-1)
            cdf = torch.bincount(v, minlength=256).cumsum(0).float()
            lo = cdf[cdf > 0].min()
            lut = ((cdf - lo) / (cdf[-1] - lo).clamp_min(1) * 255).round()
            out.append(lut[v].reshape(c.shape))
        return torch.stack(out).to(img.dtype)


class BoxImageTransform(ImageTransform):
    """Pads (or crops) to a fixed box, centred."""

    def __init__(self, width, height):
        self.w, self.h = int(width), int(height)

    def transform(self, img, rng):
        C, H, W = img.shape
        out = torch.zeros(C, self.h, self.w, dtype=img.dtype)
        h, w = min(H, self.h), min(W, self.w)
        ys, xs = (H - h) // 2, (W - w) // 2
        yd, xd = (self.h - h) // 2, (self.w - w) // 2
        out[:, yd:yd + h, xd:xd + w] = img[:, ys:ys + h, xs:xs + w]
        return out


class PipelineImageTransform(ImageTransform):
    """Sequence of (transform, probability) pairs, optionally shuffled (PipelineImageTransform.java)."""

    def __init__(self, *steps, shuffle=False, seed=None):
        if len(steps) == 1 and isinstance(steps[0], list):
            steps = steps[0]
        self.steps = [(s, 1.0) if isinstance(s, ImageTransform) else (s[0], float(s[1])) for s in steps]
        self.shuffle = shuffle

    def transform(self, img, rng):
        steps = list(self.steps)
        if self.shuffle:
            rng.shuffle(steps)
        for t, p in steps:
            if p >= 1.0 or rng.random() < p:
                img = t.transform(img, rng)
        return img


class MultiImageTransform(PipelineImageTransform):
    pass


# ----------------------------------------------------------------------------- loader
class NativeImageLoader:
    """Decode -> float32 NCHW (channels in BGR order like OpenCV), optionally resized to (height, width)."""

    def __init__(self, height=-1, width=-1, channels=3, imageTransform: Optional[ImageTransform] = None, centerCropIfNeeded=False):
        self.height, self.width, self.channels, self.transform, self.centerCrop = int(height), int(width), int(channels), imageTransform, centerCropIfNeeded
        self._rng = random.Random(42)

    def _decode(self, src) -> torch.Tensor:
        from PIL import Image
        img = src if isinstance(src, Image.Image) else Image.open(src)
        img = img.convert("L" if self.channels == 1 else ("RGBA" if self.channels == 4 else "RGB"))
        if self.centerCrop and self.height > 0 and self.width > 0:
            W, H = img.size
            tgt = self.width / self.height
            if W / H > tgt:
                nw = int(round(H * tgt))
                img = img.crop(((W - nw) // 2, 0, (W - nw) // 2 + nw, H))
            else:
                nh = int(round(W / tgt))
                img = img.crop((0, (H - nh) // 2, W, (H - nh) // 2 + nh))
        if self.height > 0 and self.width > 0 and img.size != (self.width, self.height) and self.transform is None:
            img = img.resize((self.width, self.height), Image.BILINEAR)
        a = np.asarray(img, dtype=np.float32)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(a).permute(2, 0, 1).contiguous()
        if self.channels == 3:
            t = t.flip(0)                                    # RGB -> BGR (NativeImageLoader keeps OpenCV's channel order)
        return t

    def asMatrix(self, src) -> torch.Tensor:
        t = self._decode(src)
        if self.transform is not None:
            t = self.transform.transform(t, self._rng)
            if self.height > 0 and self.width > 0 and tuple(t.shape[1:]) != (self.height, self.width):
                t = torch.nn.functional.interpolate(t.unsqueeze(0), size=(self.height, self.width), mode="bilinear", align_corners=False)[0]
        return t.unsqueeze(0)

    def asRowVector(self, src):
        return self.asMatrix(src).reshape(1, -1)

    def asImageMatrix(self, src):
        return self.asMatrix(src)

    @staticmethod
    def toImage(t: torch.Tensor):
        from PIL import Image
        t = t[0] if t.dim() == 4 else t
        if t.shape[0] == 3:
            t = t.flip(0)
        a = t.permute(1, 2, 0).clamp(0, 255).round().to(torch.uint8).numpy()
        return Image.fromarray(a[:, :, 0] if a.shape[2] == 1 else a)


ImageLoader = NativeImageLoader


class ImageRecordReader(RecordReader):
    """[NDArrayWritable(1xCxHxW), IntWritable(label)] per image (ImageRecordReader.java / BaseImageRecordReader.java)."""

    def __init__(self, height, width, channels=3, labelGenerator: Optional[PathLabelGenerator] = None, imageTransform: Optional[ImageTransform] = None,
                 appendLabel=None):
        super().__init__()
        self.loader = NativeImageLoader(height, width, channels, imageTransform)
        self.labelGen = labelGenerator
        self.appendLabel = labelGenerator is not None if appendLabel is None else appendLabel
        self.height, self.width, self.channels = height, width, channels

    def initialize(self, split: InputSplit, imageTransform=None):
        if imageTransform is not None:
            self.loader.transform = imageTransform
        self.split = split
        self._locs = [l for l in split.locations() if l.lower().rsplit(".", 1)[-1] in ALLOWED_FORMATS]
        if self.appendLabel and self.labelGen is not None:
            self._labels = sorted({self.labelGen.getLabelForPath(l).toString() for l in self._locs})
        self._i = 0
        return self

    def reset(self):
        self._i = 0

    def hasNext(self):
        return self._i < len(self._locs)

    def numLabels(self):
        return len(self._labels or [])

    def next(self):
        loc = self._locs[self._i]
        self._i += 1
        out = [NDArrayWritable(self.loader.asMatrix(loc))]
        if self.appendLabel and self.labelGen is not None:
            out.append(IntWritable(self._labels.index(self.labelGen.getLabelForPath(loc).toString())))
        return out

    def nextRecord(self):
        loc, i = self._locs[self._i], self._i
        return Record(self.next(), RecordMetaData(loc, i))

    def batchesSupported(self):
        return True

    def nextBatch(self, n) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        xs, ys = [], []
        while len(xs) < n and self.hasNext():
            r = self.next()
            xs.append(r[0].value)
            if len(r) > 1:
                ys.append(r[1].toInt())
        return torch.cat(xs, 0), (torch.tensor(ys, dtype=torch.int64) if ys else None)


class ImageObject:
    def __init__(self, x1, y1, x2, y2, label):
        self.x1, self.y1, self.x2, self.y2, self.label = x1, y1, x2, y2, label


class ObjectDetectionRecordReader(ImageRecordReader):
    """[image, label grid [4 + C, gridH, gridW]] in the Yolo2OutputLayer label format (ObjectDetectionRecordReader.java):
    box corners in grid units at the cell containing the box centre, plus a one-hot class."""

    def __init__(self, height, width, channels, gridH, gridW, labelProvider, imageTransform=None):
        super().__init__(height, width, channels, None, imageTransform, appendLabel=False)
        self.gridH, self.gridW, self.provider = gridH, gridW, labelProvider

    def initialize(self, split, imageTransform=None):
        super().initialize(split, imageTransform)
        labs = set()
        for l in self._locs:
            labs.update(o.label for o in self.provider.getImageObjectsForPath(l))
        self._labels = sorted(labs)
        return self

    def next(self):
        from PIL import Image
        loc = self._locs[self._i]
        with Image.open(loc) as im:
            W, H = im.size
        img = super().next()[0]
        C = len(self._labels)
        grid = torch.zeros(1, 4 + C, self.gridH, self.gridW)
        for o in self.provider.getImageObjectsForPath(loc):
            cx, cy = (o.x1 + o.x2) / 2 / W * self.gridW, (o.y1 + o.y2) / 2 / H * self.gridH
            gx, gy = min(int(cx), self.gridW - 1), min(int(cy), self.gridH - 1)
            grid[0, 0, gy, gx], grid[0, 1, gy, gx] = o.x1 / W * self.gridW, o.y1 / H * self.gridH
            grid[0, 2, gy, gx], grid[0, 3, gy, gx] = o.x2 / W * self.gridW, o.y2 / H * self.gridH
            grid[0, 4 + self._labels.index(o.label), gy, gx] = 1
        return [img, NDArrayWritable(grid)]
