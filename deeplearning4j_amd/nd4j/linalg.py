"""nd4j linear-algebra utility packages: ``Lapack`` interface, ``Eigen``, ``InvertMatrix``, ``PCA``, ``RandomProjection`` and the
``DistributedINDArray`` family.

Reference (behaviour, not code):
* nd4j-api/.../linalg/api/blas/Lapack.java:42-162 (getrf / geqrf / potrf / syev / gesvd / factor extraction / getri)
* nd4j-api/.../linalg/eigen/Eigen.java:40-113
* nd4j-api/.../linalg/inverse/InvertMatrix.java:41-127
* nd4j-api/.../linalg/dimensionalityreduction/PCA.java:41-381, RandomProjection.java:42-248
* nd4j-api/.../linalg/api/concurrency/{ArrayType,DistributedINDArray,BasicDistributedINDArray}.java

MI355X notes: the factorizations go to rocSOLVER through ATen when the operands live on the GPU (``torch.linalg``); everything
GEMM-shaped (covariance, projections, component conversion) is ONE matrix product instead of the reference's per-row outer-product
loop (PCA.java:341-345 issues `rows` rank-1 updates), so it lands on the MFMA path for device arrays. In-place contracts of the
reference (syev overwrites A with the eigenvectors, potrf / getrf overwrite A with the factors) are kept.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch

from .ndarray import INDArray


def _t(a) -> torch.Tensor:
    return a._t if isinstance(a, INDArray) else torch.as_tensor(a)


def _wrap(t: torch.Tensor) -> INDArray:
    return INDArray(t, "c")


def _compute_dtype(t: torch.Tensor):
    return t.dtype if t.dtype in (torch.float32, torch.float64) else torch.float32


def _assign(dst: INDArray, src: torch.Tensor):
    """Write ``src`` into the storage of ``dst`` (in-place LAPACK contract), keeping dst's dtype / layout."""
    dst._t.copy_(src.to(dst._t.dtype).reshape(dst._t.shape))
    return dst


def _square(a: INDArray, what: str):
    t = _t(a)
    if t.dim() != 2 or t.shape[0] != t.shape[1]:
        raise ValueError("%s must be a square matrix: has shape %s" % (what, list(t.shape)))
    return t


class Lapack:
    """LAPACK-level interface (Lapack.java). Obtained with ``Nd4j.getBlasWrapper().lapack()``."""

    # ---- LU
    def getrf(self, A: INDArray) -> INDArray:
        """LU factorization with partial pivoting, in place: A <- L\\U (unit lower diagonal not stored). Returns the pivot
        indices as an INT array (1-based, LAPACK convention) — reference Lapack.java:42."""
        t = _t(A)
        if t.dim() != 2:
            raise ValueError("getrf: matrix expected")
        LU, piv = torch.linalg.lu_factor(t.to(_compute_dtype(t)))
        _assign(A, LU)
        return _wrap(piv.to(torch.int32))

    def getPFactor(self, M: int, ipiv: INDArray) -> INDArray:
        """Permutation matrix P (M x M) from the pivot vector of getrf such that A = P . L . U (Lapack.java:127)."""
        piv = _t(ipiv).to(torch.int64).reshape(-1).tolist()
        perm = list(range(M))
        for i, p in enumerate(piv):
            p -= 1
            if 0 <= p < M and i < M:
                perm[i], perm[p] = perm[p], perm[i]
        P = torch.zeros(M, M, dtype=torch.float32, device=_t(ipiv).device)
        for i, p in enumerate(perm):
            P[p, i] = 1.0
        return _wrap(P)

    def getLFactor(self, A: INDArray) -> INDArray:
        t = _t(A)
        m, n = t.shape
        k = min(m, n)
        Lm = torch.tril(t[:, :k], -1)
        Lm = Lm + torch.eye(m, k, dtype=t.dtype, device=t.device)
        return _wrap(Lm)

    def getUFactor(self, A: INDArray) -> INDArray:
        t = _t(A)
        k = min(t.shape)
        return _wrap(torch.triu(t[:k, :]))

    def getri(self, N: int, A: INDArray, lda: int, IPIV, WORK=None, lwork: int = 0, INFO: int = 0):
        """Inverse from the getrf factors, in place (Lapack.java:162)."""
        t = _t(A)
        piv = torch.as_tensor(IPIV._t if isinstance(IPIV, INDArray) else IPIV, dtype=torch.int32, device=t.device)
        cd = _compute_dtype(t)
        inv = torch.linalg.lu_solve(t.to(cd), piv, torch.eye(N, dtype=cd, device=t.device))
        _assign(A, inv)

    # ---- QR
    def geqrf(self, A: INDArray, R: Optional[INDArray] = None):
        """QR factorization: A <- Q (m x n, orthonormal columns), R (n x n upper) written into ``R`` when given
        (Lapack.java:59)."""
        t = _t(A)
        m, n = t.shape
        if R is not None and tuple(_t(R).shape) != (n, n):
            raise ValueError("geqrf: R must be n x n")
        Q, Rm = torch.linalg.qr(t.to(_compute_dtype(t)), mode="reduced")
        if m >= n:
            _assign(A, Q)
        else:
            A._t[:, :m].copy_(Q.to(t.dtype))
        if R is not None:
            _assign(R, Rm[:n, :n] if Rm.shape[0] >= n else torch.nn.functional.pad(Rm, (0, 0, 0, n - Rm.shape[0])))

    # ---- Cholesky
    def potrf(self, A: INDArray, lower: bool):
        """Cholesky factor in place; the other triangle is zeroed (Lapack.java:78)."""
        t = _square(A, "potrf: A")
        Lm, info = torch.linalg.cholesky_ex(t.to(_compute_dtype(t)), upper=not lower)
        if int(info) != 0:
            raise ArithmeticError("potrf: the leading minor of order %d is not positive definite" % int(info))
        _assign(A, Lm)

    # ---- symmetric eigen decomposition
    def syev(self, jobz: str, uplo: str, A: INDArray, V: INDArray) -> int:
        """Eigenvalues (ascending) of the symmetric matrix A into V; with jobz == 'V' the orthonormal eigenvectors overwrite A as
        its COLUMNS (Lapack.java:92). ``uplo`` selects the triangle that is read."""
        t = _square(A, "syev: A")
        w, Z = torch.linalg.eigh(t.to(_compute_dtype(t)), UPLO="L" if str(uplo).upper().startswith("L") else "U")
        _assign(V, w)
        if str(jobz).upper().startswith("V"):
            _assign(A, Z)
        return 0

    # ---- SVD
    def gesvd(self, A: INDArray, S: INDArray, U: Optional[INDArray], VT: Optional[INDArray]):
        """A = U . diag(S) . VT with full U (m x m) / VT (n x n) when the outputs are given (Lapack.java:111)."""
        t = _t(A)
        Um, s, Vh = torch.linalg.svd(t.to(_compute_dtype(t)), full_matrices=True)
        _assign(S, s)
        if U is not None:
            _assign(U, Um)
        if VT is not None:
            _assign(VT, Vh)


_LAPACK = Lapack()


def lapack() -> Lapack:
    return _LAPACK


class InvertMatrix:
    """Matrix inverses (InvertMatrix.java)."""

    @staticmethod
    def invert(arr: INDArray, inPlace: bool = False) -> INDArray:
        t = _t(arr)
        if t.dim() == 2 and t.numel() == 1:
            return arr.rdivi(1.0) if inPlace else arr.rdiv(1.0)
        if t.dim() != 2 or t.shape[0] != t.shape[1]:
            raise ValueError("invalid array: must be square matrix")
        inv, info = torch.linalg.inv_ex(t.to(_compute_dtype(t)))
        if int(info) != 0:
            raise ArithmeticError("invert: matrix is singular")
        if inPlace:
            return _assign(arr, inv)
        return _wrap(inv.to(t.dtype if t.is_floating_point() else inv.dtype))

    @staticmethod
    def pinvert(arr: INDArray, inPlace: bool = False) -> INDArray:
        """Moore-Penrose pseudo inverse through QR (the reference solves with a commons-math QRDecomposition and rejects rank
        deficient input, InvertMatrix.java:64-90): for A = Q R (m >= n, full column rank) pinv(A) = R^-1 Q^T."""
        t = _t(arr)
        cd = _compute_dtype(t)
        a = t.to(cd)
        m, n = a.shape
        wide = m < n
        if wide:
            a = a.t()
        Q, R = torch.linalg.qr(a, mode="reduced")
        d = R.diagonal().abs()
        if bool((d <= torch.finfo(cd).eps * max(a.shape) * d.max()).any()):
            raise ValueError("invalid array: must be singular matrix")          # message kept from the reference
        pinv = torch.linalg.solve_triangular(R, Q.t(), upper=True)
        if wide:
            pinv = pinv.t()
        out = _wrap(pinv.to(t.dtype if t.is_floating_point() else cd).contiguous())
        if inPlace:
            if tuple(arr._t.shape) == tuple(pinv.shape):
                arr.assign(out)
            else:
                arr._t = out._t
        return out

    @staticmethod
    def pLeftInvert(arr: INDArray, inPlace: bool = False) -> INDArray:
        """(A^T A)^-1 A^T — needs full column rank."""
        a = _t(arr)
        cd = _compute_dtype(a)
        g = a.to(cd).t() @ a.to(cd)
        inv, info = torch.linalg.inv_ex(g)
        if int(info) != 0:
            raise ValueError("Full column rank condition for left pseudo inverse was not met.")
        out = _wrap((inv @ a.to(cd).t()).to(a.dtype if a.is_floating_point() else cd))
        if inPlace:
            arr._t = out._t
        return out

    @staticmethod
    def pRightInvert(arr: INDArray, inPlace: bool = False) -> INDArray:
        """A^T (A A^T)^-1 — needs full row rank."""
        a = _t(arr)
        cd = _compute_dtype(a)
        g = a.to(cd) @ a.to(cd).t()
        inv, info = torch.linalg.inv_ex(g)
        if int(info) != 0:
            raise ValueError("Full row rank condition for right pseudo inverse was not met.")
        out = _wrap((a.to(cd).t() @ inv).to(a.dtype if a.is_floating_point() else cd))
        if inPlace:
            arr._t = out._t
        return out


class Eigen:
    """Eigenvalue helpers (Eigen.java). The "generalized" names are the reference's: with one argument they are the plain
    symmetric problem; with (A, B) the reference forms inv(B) . A and runs syev on it (Eigen.java:70-104)."""

    @staticmethod
    def symmetricGeneralizedEigenvalues(A: INDArray, B=None, calculateVectors: Optional[bool] = None) -> INDArray:
        if isinstance(B, bool) and calculateVectors is None:          # (A, calculateVectors) overload
            B, calculateVectors = None, B
        _square(A, "Argument A")
        t = _t(A)
        w = _wrap(torch.empty(t.shape[0], dtype=_compute_dtype(t), device=t.device))
        if B is None:
            vec = True if calculateVectors is None else calculateVectors
            lapack().syev("V", "L", A if vec else A.dup(), w)
            return w
        _square(B, "Argument B")
        prod = _t(InvertMatrix.invert(B, False)).to(_compute_dtype(t)) @ t.to(_compute_dtype(t))
        if calculateVectors:
            _assign(A, prod)
            lapack().syev("V", "L", A, w)
        else:
            lapack().syev("V", "L", _wrap(prod), w)
        return w

    @staticmethod
    def eig(A: INDArray) -> List[INDArray]:
        """General (non-symmetric) eigen decomposition: [values [n, 2] (re, im), vectors [n, n, 2]] — the layout of the libnd4j
        `eig` declarable op (Eigen.java:107)."""
        t = _square(A, "Argument A")
        w, V = torch.linalg.eig(t.to(_compute_dtype(t)))
        return [_wrap(torch.view_as_real(w).contiguous()), _wrap(torch.view_as_real(V).contiguous())]


class PCA:
    """Principal component analysis (PCA.java). Instance form: covariance / mean / eigenvectors / eigenvalues of a data set
    (rows = records); static form: SVD based factor matrices."""

    def __init__(self, dataset: INDArray):
        cov, mean = PCA.covarianceMatrix(dataset)
        self.covarianceMatrix_, self.mean = cov, mean
        self.eigenvectors, self.eigenvalues = PCA.principalComponents(cov)

    # ---- instance API
    def _ndims_for(self, variance: float) -> int:
        vars_ = _t(self.eigenvalues).double().pow(-0.5).reshape(-1)
        res = float(vars_.sum())
        total, nd = 0.0, 0
        for v in vars_.tolist():
            nd += 1
            total += v
            if total / res > variance:
                break
        return nd

    def reducedBasis(self, variance: float) -> INDArray:
        nd = self._ndims_for(variance)
        return _wrap(_t(self.eigenvectors)[:, :nd].clone())

    def convertToComponents(self, data: INDArray) -> INDArray:
        dx = _t(data) - _t(self.mean).reshape(1, -1)
        return _wrap(dx @ _t(self.eigenvectors))                      # (E^T dx^T)^T

    def convertBackToFeatures(self, data: INDArray) -> INDArray:
        return _wrap(_t(data) @ _t(self.eigenvectors).t() + _t(self.mean).reshape(1, -1))

    def estimateVariance(self, data: INDArray, ndims: int) -> float:
        dx = (_t(data).reshape(-1) - _t(self.mean).reshape(-1))
        v = _t(self.eigenvectors).t() @ dx
        t2 = v * v
        return float(t2[:ndims].sum() / t2.sum())

    def generateGaussianSamples(self, count: int) -> INDArray:
        ev = _t(self.eigenvalues).reshape(-1)
        samples = torch.randn(int(count), ev.numel(), dtype=ev.dtype, device=ev.device) * ev.pow(-0.5).reshape(1, -1)
        return _wrap(samples @ _t(self.eigenvectors).t() + _t(self.mean).reshape(1, -1))

    def getCovarianceMatrix(self): return self.covarianceMatrix_
    def getMean(self): return self.mean
    def getEigenvectors(self): return self.eigenvectors
    def getEigenvalues(self): return self.eigenvalues

    # ---- static API
    @staticmethod
    def covarianceMatrix(in_: INDArray):
        """(covariance [n, n] with divisor = rows, mean [n]) — one GEMM instead of `rows` outer products."""
        x = _t(in_)
        x = x.to(_compute_dtype(x))
        mean = x.sum(0) / x.shape[0]
        d = x - mean.reshape(1, -1)
        return _wrap((d.t() @ d) / x.shape[0]), _wrap(mean)

    @staticmethod
    def principalComponents(cov: INDArray):
        """(eigenvectors as columns, eigenvalues) of inv(cov) in ascending order = largest variance first; the variance of
        component i is eigenvalues[i] ** -1 (PCA.java:358-364)."""
        _square(cov, "Convariance matrix")
        n = _t(cov).shape[0]
        vecs = _wrap(torch.eye(n, dtype=_compute_dtype(_t(cov)), device=_t(cov).device))
        vals = Eigen.symmetricGeneralizedEigenvalues(vecs, cov, True)
        return vecs, vals

    @staticmethod
    def _svd_vt(A: INDArray, normalize: bool):
        a = _t(A).to(_compute_dtype(_t(A))).clone()
        if normalize:
            a = a - a.mean(0, keepdim=True)
        _, s, Vh = torch.linalg.svd(a, full_matrices=True)
        return a, s, Vh

    @staticmethod
    def pca_factor(A: INDArray, nDimsOrVariance, normalize: bool) -> INDArray:
        """Factor matrix [n, k] whose columns are the leading right singular vectors. An int selects k directly; a float in
        (0, 1] keeps the smallest k whose share of sqrt(s_i)/(m-1) (the reference's conversion, PCA.java:262-264) reaches the
        fraction. (The reference returns the transpose on its CPU backend for the int overload — issue 9931 noted at
        PCA.java:197-200; this is the device-backend orientation, the one `A.mmul(factor)` needs.)"""
        a, s, Vh = PCA._svd_vt(A, normalize)
        m, n = a.shape
        if isinstance(nDimsOrVariance, bool) or not isinstance(nDimsOrVariance, (int, float)):
            raise TypeError("nDims (int) or variance (float) expected")
        if isinstance(nDimsOrVariance, int):
            k = int(nDimsOrVariance)
            if not 0 < k <= n:
                raise ValueError("nDims must be in [1, %d]" % n)
        else:
            e = s.double().sqrt() / max(m - 1, 1)
            target = float(e.sum()) * float(nDimsOrVariance)
            run, k = 0.0, -1
            for i, v in enumerate(e.tolist()):
                run += v
                if run >= target:
                    k = i + 1
                    break
            if k == -1:
                raise RuntimeError("No reduction possible for reqd. variance - use smaller variance")
        return _wrap(Vh[:k, :].t().contiguous())

    @staticmethod
    def pca(A: INDArray, nDimsOrVariance, normalize: bool) -> INDArray:
        f = PCA.pca_factor(A, nDimsOrVariance, normalize)
        return _wrap(_t(A).to(_t(f).dtype) @ _t(f))

    @staticmethod
    def pca2(in_: INDArray, variance: float) -> INDArray:
        cov, _ = PCA.covarianceMatrix(in_)
        vecs, vals = PCA.principalComponents(cov)
        vars_ = _t(vals).double().pow(-0.5).reshape(-1)
        res, total, nd = float(vars_.sum()), 0.0, 0
        for v in vars_.tolist():
            nd += 1
            total += v
            if total / res > variance:
                break
        return _wrap(_t(vecs)[:, :nd].clone())


class RandomProjection:
    """Gaussian random projection with the Johnson-Lindenstrauss bound (RandomProjection.java)."""

    def __init__(self, epsOrComponents, rng=None):
        self.rng = rng
        if isinstance(epsOrComponents, bool):
            raise TypeError("eps (float) or components (int) expected")
        if isinstance(epsOrComponents, int):
            self.components, self.eps, self.autoMode = int(epsOrComponents), 0.0, False
        else:
            self.components, self.eps, self.autoMode = 0, float(epsOrComponents), True
        self._shape = None
        self._matrix = None

    @staticmethod
    def johnsonLindenstraussMinDim(n, *eps) -> List[int]:
        ns = list(n) if isinstance(n, (list, tuple)) else [n]
        if len(eps) == 1 and isinstance(eps[0], (list, tuple)):
            eps = tuple(eps[0])
        if not ns or not eps:
            raise ValueError("Johnson-Lindenstrauss dimension estimation requires > 0 components and at least a relative error")
        for e in eps:
            if e <= 0 or e >= 1:
                raise ValueError("A relative error should be in ]0, 1[")
        res = []
        for e in eps:
            denom = e ** 2 / 2 - e ** 3 / 3
            for c in ns:
                res.append(int(4 * math.log(c) / denom))
        return res

    johnsonLindenStraussMinDim = johnsonLindenstraussMinDim       # both spellings exist upstream

    @staticmethod
    def _target_shape(shape: Sequence[int], eps: float, targetDimension: int, auto: bool):
        comp = targetDimension
        if auto:
            comp = RandomProjection.johnsonLindenstraussMinDim(shape[0], eps)[0]
        if auto and (comp <= 0 or comp > shape[1]):
            raise RuntimeError("Estimation led to a target dimension of %d, which is invalid" % comp)
        return [int(shape[1]), int(comp)]

    @staticmethod
    def targetShape(X: INDArray, epsOrDim):
        if isinstance(epsOrDim, int) and not isinstance(epsOrDim, bool):
            return RandomProjection._target_shape(_t(X).shape, -1, epsOrDim, False)
        return RandomProjection._target_shape(_t(X).shape, float(epsOrDim), -1, True)

    def _projection(self, shape, like: torch.Tensor) -> torch.Tensor:
        if self._matrix is None or self._shape != list(shape) or self._matrix.device != like.device:
            g = None
            if self.rng is not None and hasattr(self.rng, "generator"):
                g = self.rng.generator(like.device) if callable(self.rng.generator) else self.rng.generator
            elif isinstance(self.rng, torch.Generator):
                g = self.rng
            elif isinstance(self.rng, int):
                g = torch.Generator(device=like.device).manual_seed(self.rng)
            dt = _compute_dtype(like)
            if g is not None and g.device != like.device:
                m = torch.randn(*shape, generator=g, dtype=dt, device=g.device).to(like.device)
            else:
                m = torch.randn(*shape, generator=g, dtype=dt, device=like.device)
            self._matrix = m * (1.0 / math.sqrt(shape[0]))           # N(0, 1/sqrt(n_features)) as GaussianDistribution(res, 0, 1/sqrt(shape[0]))
            self._shape = list(shape)
        return self._matrix

    def project(self, data: INDArray, result: Optional[INDArray] = None) -> INDArray:
        d = _t(data)
        shp = self._target_shape(d.shape, self.eps, self.components, self.autoMode)
        out = d.to(_compute_dtype(d)) @ self._projection(shp, d)
        if result is not None:
            return _assign(result, out)
        return _wrap(out)

    def projecti(self, data: INDArray, result: Optional[INDArray] = None) -> INDArray:
        """In-place flavour: the product has a different shape than ``data``, so (as upstream's mmuli with a null / given result)
        the result array — or ``data``'s storage reference — receives it."""
        out = self.project(data, result)
        if result is None:
            data._t = out._t
            return data
        return out


# ------------------------------------------------------------------------------------------------ distributed arrays
class ArrayType:
    CONSTANT, SYNCABLE, VARIADIC = "CONSTANT", "SYNCABLE", "VARIADIC"


class DistributedINDArray:
    """One logical array with an entry per device (DistributedINDArray.java)."""

    def getINDArrayType(self): raise NotImplementedError
    def entry(self, entry=None): raise NotImplementedError
    def propagate(self, array): raise NotImplementedError
    def numEntries(self): raise NotImplementedError
    def numActiveEntries(self): raise NotImplementedError
    def allocate(self, entry, dataTypeOrDescriptor, *shape): raise NotImplementedError


class BasicDistributedINDArray(DistributedINDArray):
    """Concrete per-device entries (the reference class is an abstract stub holding only the ArrayType,
    BasicDistributedINDArray.java:25-36; its intended use is ParallelInference-style replication). Entries live on the visible
    GPUs of THIS process (`cuda:k`), or all on the host when there is none; ``sync()`` makes a SYNCABLE array's entries equal to
    their mean with the device all-reduce primitive used by the DP fallback path (Nd4j.averageAndPropagate)."""

    def __init__(self, arrayType: str = ArrayType.SYNCABLE, numEntries: Optional[int] = None):
        if arrayType not in (ArrayType.CONSTANT, ArrayType.SYNCABLE, ArrayType.VARIADIC):
            raise ValueError("arrayType")
        self.arrayType = arrayType
        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
        self._n = int(numEntries) if numEntries is not None else max(n, 1)
        self._entries: List[Optional[INDArray]] = [None] * self._n
        self._frozen = False

    def _device(self, k):
        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
        return torch.device("cuda", k % n) if n > 0 else torch.device("cpu")

    def getINDArrayType(self):
        return self.arrayType

    def numEntries(self):
        return self._n

    def numActiveEntries(self):
        return sum(1 for e in self._entries if e is not None)

    def entry(self, entry: Optional[int] = None) -> INDArray:
        if entry is None:
            entry = torch.cuda.current_device() % self._n if torch.cuda.is_available() else 0
        e = self._entries[entry]
        if e is None:
            raise KeyError("entry %d was never allocated" % entry)
        return e

    def allocate(self, entry: int, dataTypeOrDescriptor, *shape):
        from .ndarray import DataType
        if self._frozen:
            raise RuntimeError("CONSTANT distributed array cannot be modified after replication")
        if hasattr(dataTypeOrDescriptor, "getShape"):                         # LongShapeDescriptor-like
            shp, dt = list(dataTypeOrDescriptor.getShape()), dataTypeOrDescriptor.dataType()
        else:
            shp, dt = list(shape[0]) if len(shape) == 1 and isinstance(shape[0], (list, tuple)) else list(shape), dataTypeOrDescriptor
        tdt = dt if isinstance(dt, torch.dtype) else DataType.to_torch(dt)
        if self.arrayType != ArrayType.VARIADIC:
            for e in self._entries:
                if e is not None and (list(e._t.shape) != shp or e._t.dtype != tdt):
                    raise ValueError("%s entries must share shape and data type" % self.arrayType)
        self._entries[entry] = _wrap(torch.zeros(*shp, dtype=tdt, device=self._device(entry)))

    def propagate(self, array: INDArray):
        if self._frozen:
            raise RuntimeError("CONSTANT distributed array cannot be modified after replication")
        src = _t(array)
        for k in range(self._n):
            self._entries[k] = _wrap(src.to(self._device(k), copy=True))
        if self.arrayType == ArrayType.CONSTANT:
            self._frozen = True

    def sync(self):
        """SYNCABLE only: every entry becomes the mean of all active entries."""
        if self.arrayType != ArrayType.SYNCABLE:
            raise RuntimeError("only SYNCABLE arrays can be synchronised")
        from .factory import Nd4j
        act = [e for e in self._entries if e is not None]
        if len(act) > 1:
            Nd4j.averageAndPropagate(act)
        return self
