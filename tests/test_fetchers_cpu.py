"""Standard data set iterators: on-disk formats (IDX, CIFAR-10 binary, image folders) + synthetic fallback + utility iterators."""
import numpy as np
import torch

from deeplearning4j_amd.datasets.dataset import DataSet
from deeplearning4j_amd.datasets.fetchers import (Cifar10DataSetIterator, DataSetIteratorSplitter, DoublesDataSetIterator, EmnistDataSetIterator,
                                                  KFoldIterator, MnistDataSetIterator, TinyImageNetDataSetIterator, UciSequenceDataSetIterator,
                                                  read_idx, write_idx)
from deeplearning4j_amd.datasets.iterators import ListDataSetIterator


def test_mnist_idx_files_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    imgs = rng.integers(0, 256, (50, 28, 28)).astype(np.uint8)
    labs = rng.integers(0, 10, 50).astype(np.uint8)
    write_idx(tmp_path / "train-images-idx3-ubyte", imgs)
    write_idx(tmp_path / "train-labels-idx1-ubyte.gz", labs)
    assert np.array_equal(read_idx(tmp_path / "train-images-idx3-ubyte"), imgs)
    it = MnistDataSetIterator(16, True, shuffle=False, root=str(tmp_path))
    assert not it.synthetic and it.totalExamples() == 50 and it.inputColumns() == 784 and it.totalOutcomes() == 10
    b = it.next()
    assert b.features.shape == (16, 784) and b.labels.shape == (16, 10)
    assert torch.allclose(b.features[3], torch.from_numpy(imgs[3].reshape(-1)).float() / 255)
    assert b.labels.argmax(1).tolist() == labs[:16].tolist()
    n = 16
    while it.hasNext():
        n += it.next().numExamples()
    assert n == 50
    binar = MnistDataSetIterator(8, 20, True, True, False, 1, root=str(tmp_path)).next()
    assert set(binar.features.unique().tolist()) <= {0.0, 1.0}


def test_synthetic_fallbacks_are_learnable_shapes():
    m = MnistDataSetIterator(32, False, root="/nonexistent")
    assert m.synthetic and m.next().features.shape == (32, 784)
    e = EmnistDataSetIterator(EmnistDataSetIterator.Set.LETTERS, 8, root="/nonexistent")
    assert e.totalOutcomes() == 26 and e.next().labels.shape == (8, 26) and EmnistDataSetIterator.isBalanced("letters")
    t = TinyImageNetDataSetIterator(4, root="/nonexistent", numExamples=16)
    assert t.next().features.shape == (4, 3, 64, 64) and t.totalOutcomes() == 200
    u = UciSequenceDataSetIterator(10, True)
    d = u.next()
    assert d.features.shape == (10, 1, 60) and d.labels.shape == (10, 6, 60) and float(d.labels.sum()) == 600


def test_cifar10_binary(tmp_path):
    rng = np.random.default_rng(1)
    recs = rng.integers(0, 256, (20, 3073)).astype(np.uint8)
    recs[:, 0] = rng.integers(0, 10, 20)
    recs.tofile(tmp_path / "test_batch.bin")
    it = Cifar10DataSetIterator(5, train=False, shuffle=False, root=str(tmp_path))
    assert not it.synthetic and it.getLabels()[3] == "cat"
    b = it.next()
    assert b.features.shape == (5, 3, 32, 32)
    assert torch.allclose(b.features[0], torch.from_numpy(recs[0, 1:].reshape(3, 32, 32)).float() / 255)
    assert b.labels.argmax(1).tolist() == recs[:5, 0].tolist()
    small = Cifar10DataSetIterator(5, imgDim=(16, 16), train=False, root=str(tmp_path)).next()
    assert small.features.shape == (5, 3, 16, 16)


def test_kfold_splitter_doubles():
    ds = DataSet(torch.arange(22.0).reshape(11, 2), torch.arange(11.0).reshape(11, 1))
    kf = KFoldIterator(3, ds)
    seen = []
    while kf.hasNext():
        train = kf.next()
        test = kf.testFold()
        assert train.numExamples() + test.numExamples() == 11
        assert not set(train.labels.reshape(-1).tolist()) & set(test.labels.reshape(-1).tolist())
        seen += test.labels.reshape(-1).tolist()
    assert sorted(seen) == list(map(float, range(11)))
    base = ListDataSetIterator([DataSet(torch.full((2, 1), float(i)), torch.zeros(2, 1)) for i in range(10)])
    sp = DataSetIteratorSplitter(base, 10, 0.7)
    tr = [int(d.features[0, 0]) for d in sp.getTrainIterator()]
    te = [int(d.features[0, 0]) for d in sp.getTestIterator()]
    assert tr == list(range(7)) and te == [7, 8, 9]
    dd = DoublesDataSetIterator([([1.0, 2.0], [0.0]), ([3.0, 4.0], [1.0]), ([5.0, 6.0], [0.0])], 2)
    assert dd.next().features.tolist() == [[1.0, 2.0], [3.0, 4.0]] and dd.next().numExamples() == 1
