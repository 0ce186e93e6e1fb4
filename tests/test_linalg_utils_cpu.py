"""nd4j linalg utility packages (PCA / RandomProjection / InvertMatrix / Eigen / Lapack / DistributedINDArray) against NumPy
float64 references. Reference classes: nd4j-api linalg/dimensionalityreduction, linalg/inverse, linalg/eigen, linalg/api/blas/Lapack,
linalg/api/concurrency."""
import math

import numpy as np
import pytest
import torch

from deeplearning4j_amd.nd4j.factory import Nd4j
from deeplearning4j_amd.nd4j.ndarray import INDArray
from deeplearning4j_amd.nd4j.linalg import (PCA, RandomProjection, InvertMatrix, Eigen, Lapack, ArrayType, BasicDistributedINDArray)


def _arr(a, dtype=torch.float64):
    return INDArray(torch.tensor(np.asarray(a), dtype=dtype), "c")


def _np(a):
    return a.tensor().double().cpu().numpy()


def test_lapack_factorizations_match_numpy():
    rs = np.random.RandomState(0)
    lp = Nd4j.getBlasWrapper().lapack()
    assert isinstance(lp, Lapack)
    # LU: P L U == A
    A0 = rs.randn(5, 5)
    A = _arr(A0)
    piv = lp.getrf(A)
    P, Lm, U = _np(lp.getPFactor(5, piv)), _np(lp.getLFactor(A)), _np(lp.getUFactor(A))
    assert np.allclose(P @ Lm @ U, A0, atol=1e-10)
    assert np.allclose(np.tril(Lm, -1) + np.eye(5), Lm) and np.allclose(np.triu(U), U)
    # getri from the factors
    inv = A.dup()
    lp.getri(5, inv, 5, piv)
    assert np.allclose(_np(inv), np.linalg.inv(A0), atol=1e-9)
    # QR: Q orthonormal, Q R == A
    B0 = rs.randn(7, 4)
    B, R = _arr(B0), _arr(np.zeros((4, 4)))
    lp.geqrf(B, R)
    assert np.allclose(_np(B).T @ _np(B), np.eye(4), atol=1e-10) and np.allclose(_np(B) @ _np(R), B0, atol=1e-10)
    assert np.allclose(np.triu(_np(R)), _np(R))
    # Cholesky, both triangles
    S0 = rs.randn(6, 6)
    S0 = S0 @ S0.T + 6 * np.eye(6)
    for lower in (True, False):
        S = _arr(S0)
        lp.potrf(S, lower)
        F = _np(S)
        assert np.allclose(F @ F.T if lower else F.T @ F, S0, atol=1e-9)
    with pytest.raises(ArithmeticError):
        lp.potrf(_arr(-np.eye(3)), True)
    # syev: ascending eigenvalues, eigenvectors as columns of A
    Sy = _arr(S0)
    w = _arr(np.zeros(6))
    assert lp.syev("V", "L", Sy, w) == 0
    assert np.allclose(_np(w), np.linalg.eigvalsh(S0), atol=1e-9)
    Z = _np(Sy)
    assert np.allclose(S0 @ Z, Z * _np(w)[None, :], atol=1e-8) and np.allclose(Z.T @ Z, np.eye(6), atol=1e-10)
    # gesvd: full U / VT
    G0 = rs.randn(5, 3)
    s, U_, VT = _arr(np.zeros(3)), _arr(np.zeros((5, 5))), _arr(np.zeros((3, 3)))
    lp.gesvd(_arr(G0), s, U_, VT)
    assert np.allclose(_np(s), np.linalg.svd(G0, compute_uv=False), atol=1e-10)
    assert np.allclose(_np(U_)[:, :3] * _np(s)[None, :] @ _np(VT), G0, atol=1e-10)


def test_invert_matrix_family():
    rs = np.random.RandomState(1)
    A0 = rs.randn(4, 4) + 4 * np.eye(4)
    assert np.allclose(_np(InvertMatrix.invert(_arr(A0), False)), np.linalg.inv(A0), atol=1e-10)
    a = _arr(A0)
    assert InvertMatrix.invert(a, True) is a and np.allclose(_np(a), np.linalg.inv(A0), atol=1e-10)
    one = _arr([[4.0]])
    assert float(InvertMatrix.invert(one, False).tensor()) == 0.25                   # [1, 1] edge case
    with pytest.raises(ValueError):
        InvertMatrix.invert(_arr(rs.randn(2, 3)), False)
    with pytest.raises(ArithmeticError):
        InvertMatrix.invert(_arr(np.ones((3, 3))), False)
    T0 = rs.randn(6, 3)                                                               # tall, full column rank
    assert np.allclose(_np(InvertMatrix.pinvert(_arr(T0), False)), np.linalg.pinv(T0), atol=1e-9)
    assert np.allclose(_np(InvertMatrix.pLeftInvert(_arr(T0), False)) @ T0, np.eye(3), atol=1e-9)
    W0 = rs.randn(3, 6)                                                               # wide, full row rank
    assert np.allclose(_np(InvertMatrix.pinvert(_arr(W0), False)), np.linalg.pinv(W0), atol=1e-9)
    assert np.allclose(W0 @ _np(InvertMatrix.pRightInvert(_arr(W0), False)), np.eye(3), atol=1e-9)
    deficient = np.outer(np.arange(1.0, 6.0), np.ones(3))
    with pytest.raises(ValueError):
        InvertMatrix.pinvert(_arr(deficient), False)
    with pytest.raises(ValueError):
        InvertMatrix.pLeftInvert(_arr(deficient), False)


def test_eigen_helpers():
    rs = np.random.RandomState(2)
    S0 = rs.randn(5, 5)
    S0 = S0 + S0.T
    A = _arr(S0)
    w = Eigen.symmetricGeneralizedEigenvalues(A)                                       # vectors overwrite A
    assert np.allclose(_np(w), np.linalg.eigvalsh(S0), atol=1e-10)
    assert np.allclose(S0 @ _np(A), _np(A) * _np(w)[None, :], atol=1e-9)
    keep = _arr(S0)
    w2 = Eigen.symmetricGeneralizedEigenvalues(keep, False)                            # A untouched
    assert np.allclose(_np(keep), S0) and np.allclose(_np(w2), _np(w))
    # (A, B): eigenvalues of inv(B) A (the reference's formulation); with commuting SPD matrices this is the generalized problem
    B0 = np.diag([1.0, 2.0, 3.0, 4.0, 5.0])
    A0 = np.diag([5.0, 4.0, 3.0, 2.0, 1.0])
    w3 = Eigen.symmetricGeneralizedEigenvalues(_arr(A0), _arr(B0))
    assert np.allclose(np.sort(_np(w3)), np.sort(np.diag(A0) / np.diag(B0)), atol=1e-12)
    # general eig: values as (re, im) pairs
    R0 = np.array([[0.0, -1.0], [1.0, 0.0]])                                           # rotation: +-i
    vals, vecs = Eigen.eig(_arr(R0))
    v = _np(vals)
    assert v.shape == (2, 2) and np.allclose(np.sort(v[:, 1]), [-1.0, 1.0]) and np.allclose(v[:, 0], 0.0, atol=1e-12)
    assert tuple(vecs.shape()) == (2, 2, 2)


def _pca_data(n=400, seed=3):
    rs = np.random.RandomState(seed)
    basis, _ = np.linalg.qr(rs.randn(4, 4))
    stds = np.array([5.0, 2.0, 0.5, 0.05])
    return rs.randn(n, 4) * stds[None, :] @ basis.T + np.array([1.0, -2.0, 3.0, 0.5]), basis, stds


def test_pca_instance_api():
    X, basis, stds = _pca_data()
    p = PCA(_arr(X))
    cov = np.cov(X.T, bias=True)
    assert np.allclose(_np(p.getCovarianceMatrix()), cov, atol=1e-9) and np.allclose(_np(p.getMean()), X.mean(0), atol=1e-12)
    # eigenvalues are those of inv(cov), ascending -> component variances descending
    ev = _np(p.getEigenvalues())
    assert np.all(np.diff(ev) >= 0) and np.allclose(np.sort(1.0 / ev), np.sort(np.linalg.eigvalsh(cov)), rtol=1e-6)
    E = _np(p.getEigenvectors())
    assert np.allclose(E.T @ E, np.eye(4), atol=1e-8)
    assert abs(abs(E[:, 0] @ basis[:, 0]) - 1.0) < 1e-2                                # first component = the 5.0-std direction
    comps = p.convertToComponents(_arr(X))
    assert np.allclose(_np(comps).var(0), 1.0 / ev, rtol=1e-6)
    assert np.allclose(_np(p.convertBackToFeatures(comps)), X, atol=1e-8)               # lossless round trip with all components
    rb = p.reducedBasis(0.9)
    assert tuple(rb.shape()) == (4, 2)                                                  # stds 5 + 2 of 7.55 -> 0.927 > 0.9
    assert tuple(PCA.pca2(_arr(X), 0.5).shape()) == (4, 1)
    one = X[7]
    assert 0.0 < p.estimateVariance(_arr(one), 1) <= p.estimateVariance(_arr(one), 3) <= 1.0 + 1e-12
    torch.manual_seed(0)
    gen = _np(p.generateGaussianSamples(20000))
    assert np.allclose(gen.mean(0), X.mean(0), atol=0.15) and np.allclose(np.cov(gen.T, bias=True), cov, atol=0.6, rtol=0.1)


def test_pca_static_svd_factors():
    X, basis, stds = _pca_data(seed=4)
    f2 = PCA.pca_factor(_arr(X), 2, True)
    assert tuple(f2.shape()) == (4, 2)
    _, _, vt = np.linalg.svd(X - X.mean(0), full_matrices=False)
    F = _np(f2)
    for k in range(2):
        assert abs(abs(F[:, k] @ vt[k]) - 1.0) < 1e-9
    red = PCA.pca(_arr(X), 2, True)
    assert tuple(red.shape()) == (400, 2) and np.allclose(_np(red), X @ F, atol=1e-9)
    fv = PCA.pca_factor(_arr(X), 0.8, True)
    assert 1 <= fv.shape()[1] <= 4
    with pytest.raises(ValueError):
        PCA.pca_factor(_arr(X), 9, True)


def test_random_projection_johnson_lindenstrauss():
    assert RandomProjection.johnsonLindenstraussMinDim([100, 1000], 0.5, 0.1) == [
        int(4 * math.log(100) / (0.5 ** 2 / 2 - 0.5 ** 3 / 3)), int(4 * math.log(1000) / (0.5 ** 2 / 2 - 0.5 ** 3 / 3)),
        int(4 * math.log(100) / (0.1 ** 2 / 2 - 0.1 ** 3 / 3)), int(4 * math.log(1000) / (0.1 ** 2 / 2 - 0.1 ** 3 / 3))]
    with pytest.raises(ValueError):
        RandomProjection.johnsonLindenstraussMinDim(10, 1.5)
    rs = np.random.RandomState(5)
    X = rs.randn(60, 2000)
    k = RandomProjection.johnsonLindenstraussMinDim(60, 0.5)[0]
    assert RandomProjection.targetShape(_arr(X), 0.5) == [2000, k]
    rp = RandomProjection(0.5, rng=7)
    Y = _np(rp.project(_arr(X)))
    assert Y.shape == (60, k)
    # pairwise distances preserved within the relative error (with overwhelming probability at this size)
    i, j = np.triu_indices(60, 1)
    ratio = np.linalg.norm(Y[i] - Y[j], axis=1) ** 2 / np.linalg.norm(X[i] - X[j], axis=1) ** 2 * (2000 / k)
    # projection matrix entries ~ N(0, 1/sqrt(n_features)) as upstream: squared norms scale by k / n_features
    assert abs(np.median(ratio) - 1.0) < 0.25
    assert np.allclose(_np(rp.project(_arr(X))), Y)                                     # matrix cached per shape
    fixed = RandomProjection(16, rng=3)
    out = _arr(np.zeros((60, 16)))
    assert fixed.project(_arr(X), out) is out and np.abs(_np(out)).sum() > 0
    with pytest.raises(RuntimeError):
        RandomProjection(0.1).project(_arr(rs.randn(1000, 20)))                         # JL bound exceeds the feature count


def test_distributed_indarray_entries_and_sync():
    d = BasicDistributedINDArray(ArrayType.SYNCABLE, numEntries=3)
    assert d.getINDArrayType() == ArrayType.SYNCABLE and d.numEntries() == 3 and d.numActiveEntries() == 0
    with pytest.raises(KeyError):
        d.entry(1)
    d.allocate(0, "FLOAT", 2, 3)
    d.allocate(1, "FLOAT", 2, 3)
    with pytest.raises(ValueError):
        d.allocate(2, "FLOAT", 4)                                                       # SYNCABLE: same shape everywhere
    d.allocate(2, "FLOAT", 2, 3)
    assert d.numActiveEntries() == 3
    for k in range(3):
        d.entry(k).tensor().fill_(float(k))
    d.sync()
    assert all(torch.allclose(d.entry(k).tensor(), torch.ones(2, 3)) for k in range(3))
    c = BasicDistributedINDArray(ArrayType.CONSTANT, numEntries=2)
    c.propagate(_arr([1.0, 2.0]))
    assert c.numActiveEntries() == 2 and torch.equal(c.entry(1).tensor(), c.entry(0).tensor())
    with pytest.raises(RuntimeError):
        c.propagate(_arr([3.0]))
    v = BasicDistributedINDArray(ArrayType.VARIADIC, numEntries=2)
    v.allocate(0, "FLOAT", 2)
    v.allocate(1, "DOUBLE", 5, 5)
    assert v.entry(1).tensor().dtype == torch.float64
    with pytest.raises(RuntimeError):
        v.sync()
