"""Standard data set iterators + fetchers (reference: deeplearning4j-data/deeplearning4j-datasets/.../datasets/
{fetchers/{MnistDataFetcher,EmnistDataFetcher,Cifar10Fetcher,TinyImageNetFetcher,UciSequenceDataFetcher,SvhnDataFetcher},
iterator/impl/{MnistDataSetIterator,EmnistDataSetIterator,Cifar10DataSetIterator,TinyImageNetDataSetIterator,LFWDataSetIterator,
UciSequenceDataSetIterator}.java, mnist/{MnistImageFile,MnistLabelFile}.java) and deeplearning4j-utility-iterators.

The reference downloads the archives on first use.  There is no network here, so every fetcher reads the *same on-disk formats*
from a local directory (``DL4J_RESOURCES`` / ``~/.deeplearning4j`` / an explicit ``root``): MNIST/EMNIST IDX files (optionally
.gz), CIFAR-10 binary batches, TinyImageNet / LFW image folders.  When the files are absent a deterministic synthetic stand-in of
the right shape / label space is generated (``synthetic=True`` is recorded on the iterator) so pipelines and tests still run.
"""
from __future__ import annotations

import gzip
import os
import struct
from typing import List, Optional, Sequence

import numpy as np
import torch

from .dataset import DataSet
from .iterators import DataSetIterator, ListDataSetIterator


def resources_root() -> str:
    return os.environ.get("DL4J_RESOURCES") or os.path.join(os.path.expanduser("~"), ".deeplearning4j")


def _find(root, names: Sequence[str]) -> Optional[str]:
    for base in ([root] if root else []) + [resources_root()]:
        for n in names:
            for cand in (os.path.join(base, n), os.path.join(base, n + ".gz")):
                if os.path.exists(cand):
                    return cand
    return None


# ----------------------------------------------------------------------------- IDX (MNIST family)
def read_idx(path) -> np.ndarray:
    """IDX format: magic (0,0,dtype,ndim) big-endian dims, then row-major data (MnistImageFile / MnistLabelFile)."""
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "rb") as f:
        buf = f.read()
    zero, dt, nd = struct.unpack(">HBB", buf[:4])
    if zero != 0:
        raise ValueError("%s is not an IDX file" % path)
    dims = struct.unpack(">" + "I" * nd, buf[4:4 + 4 * nd])
    dtype = {0x08: np.uint8, 0x09: np.int8, 0x0B: ">i2", 0x0C: ">i4", 0x0D: ">f4", 0x0E: ">f8"}[dt]
    return np.frombuffer(buf, dtype=dtype, offset=4 + 4 * nd).reshape(dims)


def write_idx(path, arr: np.ndarray):
    code = {np.dtype(np.uint8): 0x08, np.dtype(np.int8): 0x09, np.dtype(np.int32): 0x0C, np.dtype(np.float32): 0x0D}[arr.dtype]
    op = gzip.open if str(path).endswith(".gz") else open
    with op(path, "wb") as f:
        f.write(struct.pack(">HBB", 0, code, arr.ndim) + struct.pack(">" + "I" * arr.ndim, *arr.shape))
        f.write(arr.astype(arr.dtype.newbyteorder(">")).tobytes())


def _synthetic_images(n, c, h, w, classes, seed):
    """Class-dependent blob patterns + noise: learnable, deterministic."""
    g = np.random.default_rng(seed)
    y = g.integers(0, classes, n)
    yy, xx = np.mgrid[0:h, 0:w]
    cy = (np.arange(classes) * 7 % h)[y][:, None, None]
    cx = (np.arange(classes) * 13 % w)[y][:, None, None]
    blob = np.exp(-(((yy[None] - cy) ** 2 + (xx[None] - cx) ** 2) / (2.0 * (min(h, w) / 5.0) ** 2)))
    img = (blob[:, None] * 200 + g.normal(0, 12, (n, c, h, w))).clip(0, 255).astype(np.uint8)
    return img, y.astype(np.int64)


class _ArrayFetcherIterator(DataSetIterator):
    """Shared engine: uint8 image array + int labels -> shuffled minibatches of float features / one-hot labels."""

    def __init__(self, images: np.ndarray, labels: np.ndarray, numClasses, batch, numExamples=None, shuffle=True, seed=123, binarize=False,
                 flatten=False, scale=1.0 / 255.0, labelNames: Optional[List[str]] = None, synthetic=False):
        n = len(labels) if numExamples is None else min(int(numExamples), len(labels))
        self._x, self._y = images[:n], labels[:n]
        self._nc, self._bs, self._shuffle, self._seed = numClasses, int(batch), shuffle, seed
        self._binarize, self._flatten, self._scale = binarize, flatten, scale
        self._labelNames = labelNames
        self.synthetic = synthetic
        self._epoch = 0
        self.reset()

    def reset(self):
        n = len(self._y)
        self._order = np.random.default_rng(self._seed + self._epoch).permutation(n) if self._shuffle else np.arange(n)
        self._epoch += 1
        self._pos = 0

    def hasNext(self):
        return self._pos < len(self._y)

    def next(self, num=None):
        bs = num or self._bs
        idx = self._order[self._pos:self._pos + bs]
        self._pos += len(idx)
        x = torch.from_numpy(np.ascontiguousarray(self._x[idx])).float()
        if self._binarize:
            x = (x > 30.0).float()
        else:
            x = x * self._scale
        if self._flatten:
            x = x.reshape(x.shape[0], -1)
        y = torch.zeros(len(idx), self._nc)
        y[torch.arange(len(idx)), torch.from_numpy(self._y[idx])] = 1
        return self._pp(DataSet(x, y))

    def batch(self):
        return self._bs

    def totalOutcomes(self):
        return self._nc

    def inputColumns(self):
        return int(np.prod(self._x.shape[1:]))

    def totalExamples(self):
        return len(self._y)

    numExamples = totalExamples

    def getLabels(self):
        return self._labelNames or [str(i) for i in range(self._nc)]


class MnistDataSetIterator(_ArrayFetcherIterator):
    """MnistDataSetIterator(batch, train, seed) / (batch, numExamples, binarize, train, shuffle, seed): flattened 784 features."""
    NUM_EXAMPLES, NUM_EXAMPLES_TEST = 60000, 10000
    FILES = {True: ("train-images-idx3-ubyte", "train-labels-idx1-ubyte"), False: ("t10k-images-idx3-ubyte", "t10k-labels-idx1-ubyte")}
    SUBDIR = "MNIST"

    def __init__(self, batch, numExamples_or_train=True, binarize=False, train=None, shuffle=True, seed=123, root=None, flatten=True):
        if isinstance(numExamples_or_train, bool):
            train = numExamples_or_train if train is None else train
            numExamples = None
        else:
            numExamples = int(numExamples_or_train)
            train = True if train is None else train
        x, y, synth = self._load(train, root, seed)
        if numExamples is None and synth:
            numExamples = len(y)
        super().__init__(x, y, self._classes(), batch, numExamples, shuffle, seed, binarize, flatten, synthetic=synth)

    @classmethod
    def _classes(cls):
        return 10

    def _load(self, train, root, seed):
        imgs, labs = self.FILES[train]
        base = [root] if root else []
        base += [os.path.join(resources_root(), self.SUBDIR)]
        for b in base:
            pi, pl = _find(b, [imgs]), _find(b, [labs])
            if pi and pl:
                x = read_idx(pi)
                return x.reshape(x.shape[0], 1, x.shape[1], x.shape[2]), read_idx(pl).astype(np.int64), False
        n = 2048 if train else 512
        x, y = _synthetic_images(n, 1, 28, 28, self._classes(), 17 if train else 18)
        return x, y, True


class EmnistDataSetIterator(MnistDataSetIterator):
    """EMNIST splits (EmnistDataSetIterator.Set): COMPLETE(62) MERGE(47) BALANCED(47) LETTERS(26) DIGITS(10) MNIST(10)."""
    SUBDIR = "EMNIST"

    class Set:
        COMPLETE, MERGE, BALANCED, LETTERS, DIGITS, MNIST = "byclass", "bymerge", "balanced", "letters", "digits", "mnist"

    _NC = {"byclass": 62, "bymerge": 47, "balanced": 47, "letters": 26, "digits": 10, "mnist": 10}

    def __init__(self, dataSet, batch, train=True, shuffle=True, seed=123, root=None):
        self.dataSet = dataSet
        self.FILES = {t: ("emnist-%s-%s-images-idx3-ubyte" % (dataSet, "train" if t else "test"),
                          "emnist-%s-%s-labels-idx1-ubyte" % (dataSet, "train" if t else "test")) for t in (True, False)}
        super().__init__(batch, train, False, None, shuffle, seed, root)
        if dataSet == "letters" and not self.synthetic:
            self._y = self._y - 1                     # letters labels are 1..26 in the files

    def _classes(self):
        return self._NC[self.dataSet]

    @staticmethod
    def numLabels(dataSet):
        return EmnistDataSetIterator._NC[dataSet]

    @staticmethod
    def isBalanced(dataSet):
        return dataSet in ("balanced", "letters", "digits", "mnist")


class Cifar10DataSetIterator(_ArrayFetcherIterator):
    """CIFAR-10 binary version: each record = 1 label byte + 3072 bytes (RGB planes, 32x32) (Cifar10Fetcher.java)."""
    LABELS = ["airplane", "automobile", "bird", "cat", "deer", "dog", "frog", "horse", "ship", "truck"]

    def __init__(self, batch, imgDim=None, train=True, shuffle=True, seed=123, root=None, numExamples=None):
        x, y, synth = self._load(train, root)
        if imgDim is not None and tuple(imgDim) != (32, 32):
            t = torch.nn.functional.interpolate(torch.from_numpy(x).float(), size=tuple(imgDim), mode="bilinear", align_corners=False)
            x = t.clamp(0, 255).byte().numpy()
        super().__init__(x, y, 10, batch, numExamples, shuffle, seed, labelNames=self.LABELS, synthetic=synth)

    @staticmethod
    def _load(train, root):
        names = ["data_batch_%d.bin" % i for i in range(1, 6)] if train else ["test_batch.bin"]
        for b in ([root] if root else []) + [os.path.join(resources_root(), "cifar10"), os.path.join(resources_root(), "cifar10", "cifar-10-batches-bin")]:
            paths = [os.path.join(b, n) for n in names]
            if all(os.path.exists(p) for p in paths):
                recs = np.concatenate([np.fromfile(p, dtype=np.uint8).reshape(-1, 3073) for p in paths])
                return recs[:, 1:].reshape(-1, 3, 32, 32), recs[:, 0].astype(np.int64), False
        x, y = _synthetic_images(2048 if train else 512, 3, 32, 32, 10, 31 if train else 32)
        return x, y, True


class _ImageFolderIterator(_ArrayFetcherIterator):
    def __init__(self, folder, h, w, c, batch, numExamples, shuffle, seed, classes_fallback, synth_seed):
        x = y = None
        names = None
        if folder and os.path.isdir(folder):
            from ..datavec.image import NativeImageLoader
            names = sorted(d for d in os.listdir(folder) if os.path.isdir(os.path.join(folder, d)))
            loader = NativeImageLoader(h, w, c)
            xs, ys = [], []
            for k, d in enumerate(names):
                sub = os.path.join(folder, d)
                imgdir = os.path.join(sub, "images") if os.path.isdir(os.path.join(sub, "images")) else sub
                for f in sorted(os.listdir(imgdir)):
                    if f.lower().rsplit(".", 1)[-1] in ("jpg", "jpeg", "png", "bmp"):
                        xs.append(loader.asMatrix(os.path.join(imgdir, f))[0].clamp(0, 255).byte().numpy())
                        ys.append(k)
                        if numExamples and len(xs) >= numExamples:
                            break
            if xs:
                x, y = np.stack(xs), np.asarray(ys, dtype=np.int64)
        synth = x is None
        if synth:
            x, y = _synthetic_images(numExamples or 512, c, h, w, classes_fallback, synth_seed)
            names = None
        super().__init__(x, y, len(names) if names else classes_fallback, batch, numExamples, shuffle, seed, labelNames=names, synthetic=synth)


class TinyImageNetDataSetIterator(_ImageFolderIterator):
    """200 classes, 64x64x3 (TinyImageNetFetcher.java); reads ``tiny-imagenet-200/{train,val}`` class folders."""

    def __init__(self, batch, imgDim=(64, 64), train=True, shuffle=True, seed=123, root=None, numExamples=None):
        base = root or os.path.join(resources_root(), "TINYIMAGENET_200", "tiny-imagenet-200")
        super().__init__(os.path.join(base, "train" if train else "val"), imgDim[0], imgDim[1], 3, batch, numExamples, shuffle, seed, 200, 41)


class LFWDataSetIterator(_ImageFolderIterator):
    """Labeled Faces in the Wild: one folder per person (LFWDataSetIterator.java)."""

    def __init__(self, batch, numExamples=None, imgDim=(250, 250, 3), numLabels=5749, train=True, shuffle=True, seed=123, root=None):
        base = root or os.path.join(resources_root(), "lfw")
        super().__init__(base, imgDim[0], imgDim[1], imgDim[2] if len(imgDim) > 2 else 3, batch, numExamples or 256, shuffle, seed, min(numLabels, 40), 43)


class SvhnDataSetIterator(_ImageFolderIterator):
    def __init__(self, batch, train=True, shuffle=True, seed=123, root=None, numExamples=None):
        base = root or os.path.join(resources_root(), "SVHN", "train" if train else "test")
        super().__init__(base, 32, 32, 3, batch, numExamples, shuffle, seed, 10, 45)


class UciSequenceDataSetIterator(DataSetIterator):
    """UCI synthetic-control chart time series: 600 series x 60 steps, 6 classes (UciSequenceDataFetcher.java).  The file is
    itself synthetic by construction (Alcock & Manolopoulos generators), reproduced here when the download is absent."""

    def __init__(self, batch, train=True, seed=123, root=None):
        p = _find(root or os.path.join(resources_root(), "UCISequence_6"), ["synthetic_control.data"])
        if p:
            data = np.loadtxt(p, dtype=np.float32)
            self.synthetic = False
        else:
            data = self._generate(seed)
            self.synthetic = True
        labels = np.repeat(np.arange(6), 100)
        order = np.random.default_rng(seed).permutation(600)
        cut = 450
        sel = order[:cut] if train else order[cut:]
        x = torch.from_numpy(data[sel]).unsqueeze(1)                  # [n, 1, 60]
        y = torch.zeros(len(sel), 6, 60)
        y[torch.arange(len(sel)), torch.from_numpy(labels[sel]), :] = 1
        self._ds, self._bs, self._pos = DataSet(x, y), int(batch), 0

    @staticmethod
    def _generate(seed):
        g = np.random.default_rng(seed)
        t = np.arange(60, dtype=np.float32)
        out = []
        for cls in range(6):
            for _ in range(100):
                m, r = 30.0, 2.0
                base = m + r * g.uniform(-3, 3, 60)
                if cls == 1:                                         # cyclic
                    base += g.uniform(10, 15) * np.sin(2 * np.pi * t / g.uniform(10, 15))
                elif cls == 2:                                       # increasing trend
                    base += g.uniform(0.2, 0.5) * t
                elif cls == 3:                                       # decreasing trend
                    base -= g.uniform(0.2, 0.5) * t
                elif cls == 4:                                       # upward shift
                    base += (t >= g.integers(20, 40)) * g.uniform(7.5, 20)
                elif cls == 5:                                       # downward shift
                    base -= (t >= g.integers(20, 40)) * g.uniform(7.5, 20)
                out.append(base)
        return np.asarray(out, dtype=np.float32)

    def hasNext(self):
        return self._pos < self._ds.numExamples()

    def next(self, num=None):
        n = num or self._bs
        d = self._ds.getRange(self._pos, min(self._pos + n, self._ds.numExamples()))
        self._pos += n
        return self._pp(d)

    def reset(self):
        self._pos = 0

    def batch(self):
        return self._bs

    def totalOutcomes(self):
        return 6


# ----------------------------------------------------------------------------- utility iterators
class KFoldIterator(DataSetIterator):
    """k-fold cross-validation over one DataSet: next() = training fold, testFold() = the held-out part (KFoldIterator.java)."""

    def __init__(self, k, dataset: DataSet):
        if k <= 1:
            raise ValueError("k must be > 1")
        self.k, self.ds, self._i = int(k), dataset, 0
        n = dataset.numExamples()
        base, extra = divmod(n, self.k)
        sizes = [base + (1 if j < extra else 0) for j in range(self.k)]
        self._bounds = np.concatenate([[0], np.cumsum(sizes)])
        self._test = None

    def hasNext(self):
        return self._i < self.k

    def next(self, num=None):
        lo, hi = int(self._bounds[self._i]), int(self._bounds[self._i + 1])
        n = self.ds.numExamples()
        self._test = self.ds.getRange(lo, hi)
        parts = [self.ds.getRange(0, lo)] if lo > 0 else []
        if hi < n:
            parts.append(self.ds.getRange(hi, n))
        self._i += 1
        return self._pp(DataSet.merge(parts))

    def testFold(self):
        return self._test

    def reset(self):
        self._i = 0

    def batch(self):
        return self.ds.numExamples() - int(self._bounds[1])


class DataSetIteratorSplitter:
    """Train / test views over one iterator by batch counts (DataSetIteratorSplitter.java)."""

    def __init__(self, base: DataSetIterator, totalBatches, ratio):
        self.base, self.total, self.ratio = base, int(totalBatches), float(ratio)
        self.nTrain = int(self.total * self.ratio)

    class _View(DataSetIterator):
        def __init__(self, parent, lo, hi):
            self.p, self.lo, self.hi, self._k = parent, lo, hi, 0

        def reset(self):
            self.p.base.reset()
            for _ in range(self.lo):
                if self.p.base.hasNext():
                    self.p.base.next()
            self._k = self.lo

        def hasNext(self):
            return self._k < self.hi and self.p.base.hasNext()

        def next(self, num=None):
            self._k += 1
            return self.p.base.next()

    def getTrainIterator(self):
        v = self._View(self, 0, self.nTrain)
        v.reset()
        return v

    def getTestIterator(self):
        v = self._View(self, self.nTrain, self.total)
        v.reset()
        return v


class DoublesDataSetIterator(ListDataSetIterator):
    """Iterable of (features, labels) numeric pairs -> minibatches (DoublesDataSetIterator / FloatsDataSetIterator)."""

    def __init__(self, pairs, batchSize):
        xs = torch.tensor([list(p[0]) for p in pairs], dtype=torch.float32)
        ys = torch.tensor([list(p[1]) for p in pairs], dtype=torch.float32)
        super().__init__(DataSet(xs, ys), batchSize)


FloatsDataSetIterator = DoublesDataSetIterator


class FileSplitDataSetIterator(DataSetIterator):
    """One DataSet per file through a user callback (FileSplitDataSetIterator.java)."""

    def __init__(self, files: Sequence[str], callback):
        self.files, self.cb, self._i = list(files), callback, 0

    def hasNext(self):
        return self._i < len(self.files)

    def next(self, num=None):
        f = self.files[self._i]
        self._i += 1
        return self._pp(self.cb(f))

    def reset(self):
        self._i = 0


class DummyPreProcessor:
    def preProcess(self, ds):
        return ds


class CombinedPreProcessor:
    def __init__(self, *ps):
        self.ps = ps

    def preProcess(self, ds):
        for p in self.ps:
            p.preProcess(ds)
        return ds
