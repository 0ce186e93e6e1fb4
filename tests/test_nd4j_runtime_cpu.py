"""Native host runtime (libnd4jrocm_rt.so) + nd4j Python layer on CPU: shapeInfo/TAD math, workspaces,
codecs, npy, RNG, op registry, CSV parsing, INDArray / Nd4j API, legacy + custom op execution."""
import ctypes
import io
import math
import os

import numpy as np
import pytest
import torch

from deeplearning4j_amd.native import lib as L
from deeplearning4j_amd.nd4j.factory import Nd4j
from deeplearning4j_amd.nd4j.ndarray import INDArray, NDArrayIndex
from deeplearning4j_amd.nd4j.ops import legacy as LG
from deeplearning4j_amd.nd4j.ops.custom import DynamicCustomOp, OpExecutioner, describe, exec_op, name_by_hash, op_hash


def _i64(v):
    return (ctypes.c_int64 * len(v))(*v)


# ------------------------------------------------------------------ shape / TAD
def test_shape_info_strides_ews_order():
    rt = L.rt()
    out = (ctypes.c_int64 * 10)()
    assert rt.nd4j_shape_buffer(3, _i64([2, 3, 4]), b"c", 8192, out) == 0
    assert list(out) == [3, 2, 3, 4, 12, 4, 1, 8192, 1, ord("c")]
    rt.nd4j_shape_buffer(3, _i64([2, 3, 4]), b"f", 8192, out)
    assert list(out)[4:7] == [1, 2, 6]
    # view with stride 2 along the only non-unit dim: ews = 2; a transposed matrix: ews = 0
    rt.nd4j_shape_buffer_ex(2, _i64([1, 5]), _i64([10, 2]), b"c", 8192, (o := (ctypes.c_int64 * 8)()))
    assert list(o)[-2] == 2
    rt.nd4j_shape_buffer_ex(2, _i64([3, 4]), _i64([1, 3]), b"c", 8192, o)
    assert list(o)[-2] == 0
    a = Nd4j.create(2, 3, 4)
    assert a.shapeInfo()[:7] == [3, 2, 3, 4, 12, 4, 1] and a.elementWiseStride() == 1


def test_shape_intern_and_index_math():
    rt = L.rt()
    si = (ctypes.c_int64 * 8)()
    rt.nd4j_shape_buffer(2, _i64([3, 4]), b"c", 8192, si)
    p1, p2 = rt.nd4j_shape_intern(si), rt.nd4j_shape_intern(si)
    assert p1 == p2 and rt.nd4j_shape_cache_size() >= 1
    coords = (ctypes.c_int64 * 2)()
    rt.nd4j_index2coords(7, si, coords)
    assert list(coords) == [1, 3] and rt.nd4j_coords2offset(si, coords) == 7 and rt.nd4j_coords2index(si, coords) == 7
    t = torch.arange(12).reshape(3, 4).t()          # strided view [4,3]
    rt.nd4j_shape_buffer_ex(2, _i64(list(t.shape)), _i64(list(t.stride())), b"c", 262144, si)
    offs = (ctypes.c_int64 * 12)()
    rt.nd4j_index_offsets(si, offs)
    assert [int(t.reshape(-1)[k]) for k in range(12)] == list(offs)
    out = (ctypes.c_int64 * 3)()
    assert rt.nd4j_broadcast_shape(2, _i64([3, 1]), 3, _i64([2, 1, 5]), out) == 3 and list(out) == [2, 3, 5]
    assert rt.nd4j_broadcast_shape(1, _i64([3]), 1, _i64([4]), out) == -1 and rt.nd4j_last_error_code() == 3


def test_tad_pack_matches_tensor_along_dimension():
    rt = L.rt()
    x = torch.arange(2 * 3 * 4 * 5).reshape(2, 3, 4, 5).permute(0, 2, 1, 3)      # non-trivial strides
    si = (ctypes.c_int64 * 12)()
    rt.nd4j_shape_buffer_ex(4, _i64(list(x.shape)), _i64(list(x.stride())), b"c", 262144, si)
    pack = rt.nd4j_tad_pack(si, (ctypes.c_int * 2)(1, 3), 2)
    assert rt.nd4j_tad_num(pack) == 2 * 3
    tsi = rt.nd4j_tad_shape_info(pack)
    assert [tsi[k] for k in range(5)] == [2, 4, 5, x.stride(1), x.stride(3)]
    offs = rt.nd4j_tad_offsets(pack)
    arr = INDArray(x)
    flat = x.contiguous().reshape(-1)      # storage order of the base tensor is the arange itself
    for k in range(6):
        tad = arr.tensorAlongDimension(k, 1, 3).tensor()
        assert int(tad[0, 0]) == int(offs[k])                     # base storage value == offset (arange)
    assert rt.nd4j_tad_pack(si, (ctypes.c_int * 2)(1, 3), 2) == pack and rt.nd4j_tad_cache_size() >= 1


def test_reshape_nocopy_rule_matches_torch_view():
    rt = L.rt()
    base = torch.arange(24).reshape(2, 3, 4)
    cases = [(base, [6, 4]), (base, [2, 12]), (base.transpose(0, 1), [6, 4]), (base.transpose(1, 2), [2, 12]),
             (base[:, :, ::2], [6, 2]), (base[:, :, ::2], [2, 6]), (base.transpose(0, 2), [4, 6])]
    for t, new in cases:
        si = (ctypes.c_int64 * (2 * t.dim() + 4))()
        rt.nd4j_shape_buffer_ex(t.dim(), _i64(list(t.shape)), _i64(list(t.stride())), b"c", 262144, si)
        st = (ctypes.c_int64 * len(new))()
        ok = rt.nd4j_reshape_nocopy(si, len(new), _i64(new), b"c", st)
        try:
            v = t.view(new)
            assert ok == 1 and list(st) == list(v.stride()), (t.shape, t.stride(), new)
        except RuntimeError:
            assert ok == 0


# ------------------------------------------------------------------ workspace
def test_workspace_learning_spill_and_cycles():
    from deeplearning4j_amd.nd4j.workspace import (LearningPolicy, MemoryWorkspace, SpillPolicy, WorkspaceConfiguration,
                                                   getWorkspaceManager, ResetPolicy)
    conf = WorkspaceConfiguration(initialSize=0, overallocationLimit=0.5, policyLearning=LearningPolicy.FIRST_LOOP)
    ws = MemoryWorkspace(conf, "T1", device="cpu")
    with ws:
        a = ws.alloc([100], torch.float32)           # no arena yet -> spilled
        b = ws.alloc([50], torch.float64)
        a.fill_(1.0)
        assert ws.getSpilledSize(0) > 0 and ws.getCurrentSize(0) == 0
    need = (400 + 255) // 256 * 256 + (400 + 255) // 256 * 256
    assert ws.getCurrentSize(0) >= int(need * 1.5) // 256 * 256          # learned + over-allocation
    with ws:
        a = ws.alloc([100], torch.float32)
        b2 = ws.alloc([50], torch.float64)
        assert a.data_ptr() != b2.data_ptr() and a.data_ptr() % 256 == 0
        base = ws._slab[0].data_ptr()
        assert base <= a.data_ptr() < base + ws.getCurrentSize(0)
        a.fill_(3.0)
        b2.fill_(4.0)
        assert float(a.sum()) == 300.0 and float(b2.sum()) == 200.0
        assert ws.getCurrentOffset(0) == 1024
    assert ws.getCyclesCount() == 2 and ws.getGenerationId() == 2
    strict = MemoryWorkspace(WorkspaceConfiguration(initialSize=1024, policySpill=SpillPolicy.FAIL, policyLearning=LearningPolicy.NONE),
                             "T2", device="cpu")
    with strict:
        strict.alloc([128], torch.float32)
        with pytest.raises(MemoryError):
            strict.alloc([1024], torch.float32)
    cyc = MemoryWorkspace(WorkspaceConfiguration(initialSize=2048, policyReset=ResetPolicy.ENDOFBUFFER_REACHED,
                                                 policyLearning=LearningPolicy.NONE), "T3", device="cpu")
    with cyc:
        p = [cyc.alloc([128], torch.float32).data_ptr() for _ in range(6)]     # 512 B each: wraps after 4
        assert p[4] == p[0] and p[5] == p[1] and cyc.getSpilledSize(0) == 0
    mgr = getWorkspaceManager()
    w = mgr.getAndActivateWorkspace(conf, "NAMED", device="cpu")
    assert mgr.checkIfWorkspaceExistsAndActive("NAMED") and mgr.getCurrentWorkspace() is w
    with mgr.scopeOutOfWorkspaces():
        assert mgr.getCurrentWorkspace() is None
    w.notifyScopeLeft()
    mgr.destroyAllWorkspacesForCurrentThread()
    rt = L.rt()
    rt.mem_counter_set_limit(0, 1, 1000)
    assert rt.mem_counter_validate(0, 1, 900) == 1 and rt.mem_counter_add(0, 1, 600) == 600
    assert rt.mem_counter_validate(0, 1, 900) == 0
    rt.mem_counter_add(0, 1, -600)
    rt.mem_counter_set_limit(0, 1, 0)


# ------------------------------------------------------------------ codecs
def test_threshold_and_bitmap_codecs_host():
    from deeplearning4j_amd.nd4j.compression import ThresholdCompression as TC, BasicNDArrayCompressor
    g = torch.Generator().manual_seed(0)
    x = torch.randn(200_000, generator=g) * 0.01
    orig = x.clone()
    tau = 0.02
    enc = TC.thresholdEncode(x, tau)
    h = TC.header(enc)
    cnt = int((orig.abs() >= tau).sum())
    assert h["count"] == cnt and h["length"] == x.numel() and abs(h["threshold"] - tau) < 1e-9
    idx = enc[4:].long()
    assert torch.equal(idx.abs() - 1, torch.nonzero(orig.abs() >= tau).reshape(-1))       # index order
    assert torch.equal(idx > 0, orig[idx.abs() - 1] > 0)
    z = torch.zeros_like(x)
    TC.thresholdDecode(enc, z)
    assert torch.allclose(z + x, orig, atol=1e-7)                                        # decoded + residual == original
    assert float(x.abs().max()) < float(orig.abs().max())
    # boundary: only the first k elements are encoded
    y = orig.clone()
    enc2 = TC.thresholdEncode(y, tau, boundary=10)
    assert TC.header(enc2)["count"] == 10 and torch.equal(enc2[4:], enc[4:14])
    # bitmap
    w = orig.clone()
    benc = TC.bitmapEncode(w, tau)
    zz = torch.zeros_like(w)
    TC.bitmapDecode(benc, zz)
    assert torch.allclose(zz + w, orig, atol=1e-7)
    full = orig.abs() >= tau
    halfneg = (~full) & (orig.abs() >= tau / 2) & (orig < 0)
    assert TC.header(benc)["count"] == int(full.sum() + halfneg.sum())
    est = TC.estimateThreshold(orig, 2000)
    assert 1000 < int((orig.abs() >= est).sum()) < 4000
    c = BasicNDArrayCompressor.getInstance()
    assert {"THRESHOLD", "BITMAP", "FLOAT16", "INT8"} <= c.getAvailableCompressors()
    arr = Nd4j.create(np.linspace(-1, 1, 64).astype(np.float32))
    assert c.decompress(c.compress(arr, "FLOAT16")).equalsWithEps(arr, 1e-3)
    assert c.decompress(c.compress(arr, "INT8")).equalsWithEps(arr, 1e-2)


def test_encoded_gradients_accumulator_single_rank():
    from deeplearning4j_amd.nd4j.compression import EncodedGradientsAccumulator, FixedThresholdAlgorithm
    acc = EncodedGradientsAccumulator(FixedThresholdAlgorithm(0.5))
    g = torch.tensor([0.3, -0.6, 0.2, 1.4])
    tgt = torch.zeros(4)
    acc.applyUpdate(acc.storeUpdate(g), tgt)
    assert torch.allclose(tgt, torch.tensor([0.0, -0.5, 0.0, 0.5]))
    acc.applyUpdate(acc.storeUpdate(g), tgt)           # residual [0.3,-0.1,0.2,0.9] + g = [0.6,-0.7,0.4,2.3]
    assert torch.allclose(tgt, torch.tensor([0.5, -0.5, 0.0, 0.5]))


# ------------------------------------------------------------------ buffers / npy / serde
def test_bswap_npy_and_binary_roundtrip(tmp_path):
    rt = L.rt()
    a = np.arange(10, dtype=np.float32)
    out = np.empty_like(a)
    rt.nd4j_bswap_copy(a.ctypes.data, out.ctypes.data, 10, 4)
    assert np.array_equal(out.view(">f4").astype(np.float32), a)
    arr = Nd4j.create(np.arange(24, dtype=np.float32).reshape(2, 3, 4))
    p = tmp_path / "a.npy"
    Nd4j.writeAsNumpy(arr, p)
    assert np.array_equal(np.load(p), arr.numpy())                     # numpy reads what the native writer wrote
    np.save(tmp_path / "b.npy", np.asfortranarray(np.arange(6, dtype=np.float64).reshape(2, 3)))
    b = Nd4j.createFromNpyFile(tmp_path / "b.npy")
    assert b.ordering() == "f" and np.array_equal(b.numpy(), np.arange(6).reshape(2, 3))
    buf = io.BytesIO()
    Nd4j.write(arr, buf)
    assert Nd4j.read(io.BytesIO(buf.getvalue())).equals(arr)


# ------------------------------------------------------------------ rng
def test_philox_host_matches_python_reference_and_moments():
    from deeplearning4j_amd.nd4j import rng as R
    r = R.Random(12345)
    rt = L.rt()
    for idx in (0, 1, 2, 3, 4, 1000, 2 ** 33 + 5):
        assert rt.rng_relative_int(r._h, idx) == R.philox_bits(12345, 0, idx)
    u = Nd4j.rand(200_000, seed=7, device="cpu").tensor()
    assert 0 < float(u.min()) and float(u.max()) < 1 and abs(float(u.mean()) - 0.5) < 5e-3 and abs(float(u.var()) - 1 / 12) < 2e-3
    n = Nd4j.randn(200_000, seed=7, device="cpu").tensor()
    assert abs(float(n.mean())) < 1e-2 and abs(float(n.std()) - 1) < 1e-2
    assert torch.equal(Nd4j.rand(100, seed=3, device="cpu").tensor(), Nd4j.rand(100, seed=3, device="cpu").tensor())
    rr = R.Random(3)
    a, b = rr.nextFloatArray([50], "cpu"), rr.nextFloatArray([50], "cpu")
    assert not torch.equal(a, b)                                        # stream advances
    z = torch.empty(100_000)
    R.exec_random("BernoulliDistribution", R.Random(1), z, None, None, [0.3])
    assert abs(float(z.mean()) - 0.3) < 1e-2
    R.exec_random("TruncatedNormalDistribution", R.Random(1), z, None, None, [1.0, 2.0])
    assert float(z.min()) >= 1.0 - 4.0 - 1e-5 and float(z.max()) <= 1.0 + 4.0 + 1e-5
    R.exec_random("ExponentialDistribution", R.Random(1), z, None, None, [2.0])
    assert abs(float(z.mean()) - 0.5) < 1e-2


# ------------------------------------------------------------------ op registry / env / io
def test_op_registry_hash_and_descriptors():
    d = describe("conv2d")
    assert (d.numInputs, d.numOutputs, d.numTArgs, d.numIArgs) == (2, 1, 0, 9)
    assert name_by_hash(op_hash("lstmLayer")) == "lstmLayer"
    ops = OpExecutioner.getInstance().getCustomOperations()
    assert len(ops) >= 500 and ops["softmax"].hash == op_hash("softmax")
    assert OpExecutioner.getInstance().missingOps() == []
    with pytest.raises(ValueError):
        exec_op("conv2d", [torch.zeros(1, 1, 3, 3), torch.zeros(1, 1, 1, 1)], iArgs=[1, 1])      # needs >= 9 iArgs


def test_environment_launch_dims_and_profiler(monkeypatch):
    from deeplearning4j_amd.nd4j.environment import Environment, OpProfiler, ProfilerConfig
    env = Environment.getInstance()
    env.setDebug(True)
    assert env.isDebug()
    env.setDebug(False)
    assert env.launchDims("reduceFloat") == (1024, 256, 2048)
    monkeypatch.setenv("GRID_SIZE_REDUCEFLOAT", "77")
    assert env.launchDims("reduceFloat")[0] == 77
    env.setLaunchDims("myKernel", 5, 64, 128)
    assert env.launchDims("myKernel") == (5, 64, 128) and "updater" in env.launchDimNames()
    prof = OpProfiler.getInstance()
    prof.reset()
    prof.setConfig(ProfilerConfig(checkForNAN=True, checkElapsedTime=True, nativeStatistics=True))
    exec_op("add", [torch.ones(3), torch.ones(3)])
    assert prof.getInvocationsCount("add") == 1 and prof.nativeTraces()[-1][0] == "add"
    with pytest.raises(FloatingPointError):
        exec_op("divide", [torch.zeros(2), torch.zeros(2)])
    prof.setConfig(ProfilerConfig())


def test_native_csv_and_svmlight(tmp_path):
    rt = L.rt()
    p = tmp_path / "d.csv"
    p.write_text("a,b,c\n1,2.5,3\n4,5,\"6\"\n7,x,9\n")
    h = rt.dv_csv_parse(str(p).encode(), b",", 1)
    assert (rt.dv_rows(h), rt.dv_cols(h)) == (3, 3)
    out = torch.empty(3, 3)
    rt.dv_copy(h, out.data_ptr())
    rt.dv_free(h)
    assert out[0].tolist() == [1, 2.5, 3] and out[1].tolist() == [4, 5, 6] and math.isnan(float(out[2, 1]))
    q = tmp_path / "d.svm"
    q.write_text("1 1:0.5 3:2\n0 2:1.5 # c\n")
    h = rt.dv_svmlight_parse(str(q).encode(), 3, 0)
    out = torch.empty(2, 4)
    rt.dv_copy(h, out.data_ptr())
    rt.dv_free(h)
    assert out.tolist() == [[1, 0.5, 0, 2], [0, 0, 1.5, 0]]


# ------------------------------------------------------------------ INDArray / Nd4j
def test_indarray_api():
    a = Nd4j.create([[1.0, 2, 3], [4, 5, 6]], device="cpu")
    assert a.shape() == [2, 3] and a.rank() == 2 and a.length() == 6 and a.dataType() == "FLOAT"
    assert a.add(1).sumNumber() == 27 and a.mul(a).getDouble(1, 2) == 36 and a.rsub(10).getDouble(0, 0) == 9
    b = a.dup()
    b.addi(a).muli(2)
    assert b.getDouble(1, 1) == 20 and a.getDouble(1, 1) == 5
    assert a.addRowVector(Nd4j.create([10.0, 20, 30], device="cpu")).getRow(1).toDoubleVector() == [14, 25, 36]
    assert a.mulColumnVector(Nd4j.create([2.0, 3], device="cpu")).getColumn(0).toDoubleVector() == [2, 12]
    assert a.mmul(a.transpose()).toDoubleMatrix() == [[14, 32], [32, 77]]
    assert a.sum(0).toDoubleVector() == [5, 7, 9] and a.mean(1).toDoubleVector() == [2, 5]
    assert a.max().getDouble(0) == 6 and a.argMax(1).toIntVector() == [2, 2] and abs(a.norm2Number() - math.sqrt(91)) < 1e-5
    assert abs(a.std(1).getDouble(0) - 1.0) < 1e-6 and abs(a.var(biasCorrected=False).getDouble(0) - 35 / 12) < 1e-5
    assert a.gt(3).castTo("INT").sumNumber() == 3 and a.eq(a).all()
    # f-order reshape/ravel semantics
    f = a.reshape("f", 3, 2)
    assert f.toDoubleMatrix() == [[1, 5], [4, 3], [2, 6]] and a.ravel("f").toDoubleVector() == [1, 4, 2, 5, 3, 6]
    assert a.getDouble(4) == 5 and f.getDouble(1) == 4           # linear index follows the array's order
    # views alias
    v = a.get(NDArrayIndex.point(1), NDArrayIndex.interval(0, 2))
    v.assign(0)
    assert a.toDoubleMatrix() == [[1, 2, 3], [0, 0, 6]] and v.isView() and not a.isView()
    a.put([NDArrayIndex.all(), NDArrayIndex.point(2)], Nd4j.create([7.0, 8], device="cpu"))
    assert a.getColumn(2).toDoubleVector() == [7, 8]
    t3 = Nd4j.arange(24, device="cpu").reshape(2, 3, 4)
    assert t3.tensorsAlongDimension(2) == 6 and t3.tensorAlongDimension(4, 2).toDoubleVector() == [16, 17, 18, 19]
    assert t3.tensorAlongDimension(1, 0, 2).shape() == [2, 4]
    assert Nd4j.concat(0, a, a).shape() == [4, 3] and Nd4j.hstack(a, a).shape() == [2, 6] and Nd4j.vstack(a, a).shape() == [4, 3]
    assert Nd4j.toFlattened(a, a, order="f").length() == 12
    assert Nd4j.pullRows(Nd4j.arange(12, device="cpu").reshape(4, 3), 1, [3, 0]).toDoubleMatrix() == [[9, 10, 11], [0, 1, 2]]
    assert Nd4j.gemm(a, a, None, False, True).shape() == [2, 2]
    xs = [Nd4j.create([1.0, 2], device="cpu"), Nd4j.create([3.0, 6], device="cpu")]
    Nd4j.averageAndPropagate(xs)
    assert xs[0].toDoubleVector() == [2, 4] and xs[1].toDoubleVector() == [2, 4]
    tgt = Nd4j.zeros(2, device="cpu")
    Nd4j.accumulate(tgt, xs)
    assert tgt.toDoubleVector() == [4, 8]
    assert Nd4j.getAffinityManager().getNumberOfDevices() >= 1 and Nd4j.getBackend()


# ------------------------------------------------------------------ legacy executioner (CPU formulations)
def test_legacy_opnum_tables_and_exec():
    assert LG.PAIRWISE[0] == "Add" and LG.SCALAR[39] == "RELU" and LG.TRANSFORM_STRICT[53] == "GELU" and LG.BROADCAST[13] == "MinPairwise"
    assert 8 not in LG.TRANSFORM_SAME and 35 not in LG.TRANSFORM_STRICT          # gaps are preserved
    assert LG.op_num("reduce_same", "Prod") == 3 and LG.op_num("random", "GaussianDistribution") == 6
    ex = LG.executioner()
    x = torch.tensor([[1.0, -2.0, 3.0], [4.0, 0.0, -6.0]])
    y = torch.tensor([[2.0, 2.0, 2.0], [0.5, 4.0, 3.0]])
    assert torch.equal(ex.execPairwiseTransform(6, x, y), x - y)
    assert torch.equal(ex.execPairwiseTransform(19, x, y), torch.remainder(x, y))
    assert torch.equal(ex.execScalar(39, x, 0.0), torch.relu(x))
    assert torch.equal(ex.execScalar(42, x, 2.5), x.clamp(-2.5, 2.5))
    assert torch.equal(ex.execBroadcast(2, x, torch.tensor([1.0, 2.0, 3.0]), dimensions=[1]), x * torch.tensor([1.0, 2.0, 3.0]))
    assert torch.equal(ex.execBroadcast(0, x, torch.tensor([10.0, 20.0]), dimensions=[0]), x + torch.tensor([[10.0], [20.0]]))
    assert torch.equal(ex.execPairwiseTransformBool(1, x, y), x > y)
    assert torch.equal(ex.execTransformSame(0, x), x.abs())
    assert torch.allclose(ex.execTransformStrict(53, x), x * torch.sigmoid(1.702 * x))
    assert torch.equal(ex.execTransformBool(6, x), x > 0)
    assert torch.allclose(ex.execReduceFloat(3, x, [1]), x.norm(dim=1))
    assert torch.equal(ex.execReduceSame(0, x, [0], keepDims=True), x.sum(0, keepdim=True))
    assert torch.equal(ex.execReduceLong(0, x), (x != 0).sum())
    assert bool(ex.execReduceBool(0, x)) and not bool(ex.execReduceBool(1, x))
    assert torch.allclose(ex.execReduce3(2, x, y, [1]), torch.nn.functional.cosine_similarity(x, y, 1))
    assert ex.execReduce3All(1, x, y, [1]).shape == (2, 2)
    assert ex.execIndexReduce(2, x, [1]).tolist() == [2, 2]
    assert torch.allclose(ex.execSummaryStats(0, x, [1], True), x.var(1, unbiased=True))
    assert torch.equal(ex.execIndexReduce(4, x, [1], extraParams=[0.0, 0.0, 2]), torch.tensor([1, 2]))       # first x < 0
    assert torch.equal(ex.execPairwiseTransformInt(2, torch.tensor([1, -2], dtype=torch.int32), torch.tensor([1, 4], dtype=torch.int32)),
                       torch.tensor([2, -17], dtype=torch.int32))      # cyclic shift left (32-bit)


# ------------------------------------------------------------------ custom ops (spot checks against closed forms)
def test_custom_ops_spot_checks():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 6, generator=g)
    assert torch.allclose(exec_op("softmax", [x])[0], torch.softmax(x, -1))
    xr = x.clone().requires_grad_(True)
    gy = torch.randn(4, 6, generator=g)
    torch.softmax(xr, -1).backward(gy)
    assert torch.allclose(exec_op("softmax_bp", [x, gy, torch.softmax(x, -1)])[0], xr.grad, atol=1e-6)
    # conv2d, NCHW, wFormat 0 ([kH,kW,iC,oC]) vs torch
    img, w, b = torch.randn(2, 3, 8, 8, generator=g), torch.randn(3, 3, 3, 5, generator=g), torch.randn(5, generator=g)
    ref = torch.nn.functional.conv2d(img, w.permute(3, 2, 0, 1), b, 1, 1)
    out = exec_op("conv2d", [img, w, b], iArgs=[3, 3, 1, 1, 1, 1, 1, 1, 0, 0, 0])[0]
    assert torch.allclose(out, ref, atol=1e-5)
    same = exec_op("conv2d", [img, w, b], iArgs=[3, 3, 2, 2, 0, 0, 1, 1, 1, 0, 0])[0]
    assert same.shape == (2, 5, 4, 4)
    gi, gw, gb = exec_op("conv2d_bp", [img, w, b, torch.ones_like(ref)], iArgs=[3, 3, 1, 1, 1, 1, 1, 1, 0, 0, 0])
    assert gi.shape == img.shape and gw.shape == w.shape and torch.allclose(gb, torch.full((5,), 2.0 * 64))
    mp = exec_op("maxpool2d", [img], iArgs=[2, 2, 2, 2, 0, 0, 1, 1, 0, 0, 0])[0]
    assert torch.equal(mp, torch.nn.functional.max_pool2d(img, 2))
    bn = exec_op("batchnorm", [img, torch.zeros(3), torch.ones(3), torch.full((3,), 2.0), torch.ones(3)], iArgs=[1, 1, 1], tArgs=[0.0])[0]
    assert torch.allclose(bn, img * 2 + 1, atol=1e-6)
    a, bm = torch.randn(3, 4, generator=g), torch.randn(5, 4, generator=g)
    assert torch.allclose(exec_op("matmul", [a, bm], iArgs=[0, 1, 0])[0], a @ bm.t())
    assert torch.equal(exec_op("reduce_sum", [x], iArgs=[1], bArgs=[True])[0], x.sum(1, keepdim=True))
    assert torch.allclose(exec_op("reduce_mean_bp", [x, torch.ones(4)], iArgs=[1])[0], torch.full_like(x, 1 / 6))
    assert exec_op("argmax", [x], iArgs=[1])[0].tolist() == x.argmax(1).tolist()
    assert torch.equal(exec_op("gather", [x, torch.tensor([2, 0])], iArgs=[1])[0], x[:, [2, 0]])
    with pytest.raises(IndexError):
        exec_op("gather", [x, torch.tensor([9])], iArgs=[1])
    assert torch.equal(exec_op("onehot", [torch.tensor([0, 2])], iArgs=[-1, 3], tArgs=[1.0, 0.0])[0], torch.eye(3)[[0, 2]])
    assert torch.equal(exec_op("scatter_add", [torch.zeros(3, 2), torch.tensor([0, 0, 2]), torch.ones(3, 2)])[0],
                       torch.tensor([[2.0, 2], [0, 0], [1, 1]]))
    assert torch.equal(exec_op("strided_slice", [x], iArgs=[0, 0, 0, 0, 0, 1, 0, 3, 4, 1, 2])[0], x[1:3, 0:4:2])
    assert torch.equal(exec_op("pad", [torch.ones(2, 2), torch.tensor([[1, 0], [0, 2]])], iArgs=[0], tArgs=[5.0])[0],
                       torch.nn.functional.pad(torch.ones(2, 2), [0, 2, 1, 0], value=5.0))
    assert torch.equal(exec_op("space_to_depth", [exec_op("depth_to_space", [img[:, :, :4, :4].repeat(1, 4, 1, 1)[:, :12]], iArgs=[2, 0])[0]],
                               iArgs=[2, 0])[0], img[:, :, :4, :4].repeat(1, 4, 1, 1)[:, :12])
    s2b = exec_op("space_to_batch", [torch.arange(16.0).reshape(1, 4, 4, 1), torch.zeros(2, 2, dtype=torch.int64)], iArgs=[2])[0]
    assert s2b.shape == (4, 2, 2, 1) and torch.equal(
        exec_op("batch_to_space", [s2b, torch.zeros(2, 2, dtype=torch.int64)], iArgs=[2])[0], torch.arange(16.0).reshape(1, 4, 4, 1))
    u, idx = exec_op("unique", [torch.tensor([3, 1, 3, 2, 1])])
    assert u.tolist() == [3, 1, 2] and idx.tolist() == [0, 1, 0, 2, 1]
    assert torch.equal(exec_op("segment_sum", [torch.tensor([1.0, 2, 3, 4]), torch.tensor([0, 0, 1, 1])])[0], torch.tensor([3.0, 7.0]))
    assert torch.equal(exec_op("unsorted_segment_max", [torch.tensor([1.0, 5, 3]), torch.tensor([1, 0, 1])], iArgs=[2])[0], torch.tensor([5.0, 3.0]))
    assert exec_op("top_k", [torch.tensor([[1.0, 9, 3]])], iArgs=[2])[1].tolist() == [[1, 2]]
    assert exec_op("non_max_suppression", [torch.tensor([[0, 0, 1, 1], [0, 0, 0.9, 0.9], [2, 2, 3, 3.0]]), torch.tensor([0.9, 0.8, 0.7])],
                   iArgs=[3], tArgs=[0.5])[0].tolist() == [0, 2]
    S, U, V = exec_op("svd", [a], iArgs=[0, 1, 16])
    assert torch.allclose((U * S) @ V.t(), a, atol=1e-5)
    m = torch.randn(4, 4, generator=g)
    spd = m @ m.t() + 4 * torch.eye(4)
    assert torch.allclose(exec_op("sqrtm", [spd])[0] @ exec_op("sqrtm", [spd])[0], spd, atol=1e-4)
    assert torch.allclose(exec_op("matrix_inverse", [spd])[0] @ spd, torch.eye(4), atol=1e-4)
    bi = exec_op("betainc", [torch.tensor([2.0, 0.5]), torch.tensor([3.0, 0.5]), torch.tensor([0.4, 0.3])])[0]
    assert torch.allclose(bi, torch.tensor([0.5248, 0.36901]), atol=1e-3)
    hsv = exec_op("rgb_to_hsv", [torch.rand(5, 3, generator=g)])[0]
    assert hsv.shape == (5, 3)
    rgb = torch.rand(2, 4, 4, 3, generator=g)
    assert torch.allclose(exec_op("hsv_to_rgb", [exec_op("rgb_to_hsv", [rgb])[0]])[0], rgb, atol=1e-5)
    assert torch.allclose(exec_op("yiq_to_rgb", [exec_op("rgb_to_yiq", [rgb])[0]])[0], rgb, atol=1e-5)
    rz = exec_op("resize_bilinear", [rgb, torch.tensor([8, 8])], bArgs=[False, False])[0]
    assert rz.shape == (2, 8, 8, 3) and torch.allclose(rz[:, ::2, ::2], rgb, atol=1e-6)
    # updaters: adam against the closed form
    gr = torch.tensor([0.1, -0.2])
    upd, v, mm = exec_op("adam_updater", [gr, torch.zeros(2), torch.zeros(2)], tArgs=[0.01, 0.9, 0.999, 1e-8], iArgs=[0])
    m1, v1 = 0.1 * gr, 0.001 * gr * gr
    alpha = 0.01 * math.sqrt(1 - 0.999) / (1 - 0.9)
    assert torch.allclose(upd, m1 * alpha / (torch.sqrt(v1) + 1e-8), rtol=1e-4)
    # lstmLayer: TNS, fwd, bias, full sequence + last h/c vs a manual loop with gate order [i,f,c',o]
    T, B, nIn, n = 3, 2, 4, 5
    xs, Wx, Wr, bb = torch.randn(T, B, nIn, generator=g), torch.randn(nIn, 4 * n, generator=g), torch.randn(n, 4 * n, generator=g), torch.randn(4 * n, generator=g)
    hs, hl, cl = exec_op("lstmLayer", [xs, Wx, Wr, bb], iArgs=[0, 0, 2, 0, 0], tArgs=[0.0],
                         bArgs=[True, False, False, False, False, True, True, True])
    h, c = torch.zeros(B, n), torch.zeros(B, n)
    for tt in range(T):
        z = xs[tt] @ Wx + h @ Wr + bb
        i_, f_, g_, o_ = torch.sigmoid(z[:, :n]), torch.sigmoid(z[:, n:2 * n]), torch.tanh(z[:, 2 * n:3 * n]), torch.sigmoid(z[:, 3 * n:])
        c = f_ * c + i_ * g_
        h = o_ * torch.tanh(c)
        assert torch.allclose(hs[tt], h, atol=1e-6)
    assert torch.allclose(hl, h, atol=1e-6) and torch.allclose(cl, c, atol=1e-6)
    out, scores, logits = exec_op("dot_product_attention_v2", [torch.randn(2, 3, 4, generator=g), torch.randn(2, 5, 4, generator=g)],
                                  tArgs=[0.5, 0.0], bArgs=[False, False])
    assert out.shape == (2, 3, 4) and scores.shape == (2, 3, 5) and torch.allclose(scores.sum(-1), torch.ones(2, 3))
    loss = exec_op("mean_sqerr_loss", [torch.tensor([[1.0, 2.0]]), torch.ones(1, 1), torch.tensor([[0.0, 0.0]])], iArgs=[1])[0]
    assert float(loss) == 5.0
    paths, sc, ln = exec_op("ctc_beam", [torch.log(torch.tensor([[[0.1, 0.8, 0.1], [0.1, 0.8, 0.1], [0.8, 0.1, 0.1], [0.1, 0.1, 0.8]]])),
                                         torch.tensor([4])], iArgs=[0, 5, 1], bArgs=[False])
    assert paths[0, 0, :int(ln[0, 0])].tolist() == [1, 2]


def test_calculate_output_shape_and_dynamic_custom_op():
    o = (DynamicCustomOp.builder("conv2d").addInputs(torch.zeros(2, 3, 9, 9), torch.zeros(3, 3, 3, 7))
         .addIntegerArguments(3, 3, 2, 2, 0, 0, 1, 1, 0, 0, 0).build())
    assert o.calculateOutputShape() == [[2, 7, 4, 4]]
    assert DynamicCustomOp("reduce_max", [torch.zeros(4, 5, 6)], iArgs=[0, 2]).calculateOutputShape() == [[5]]
    z = torch.zeros(2, 2)
    DynamicCustomOp("add", [torch.ones(2, 2), torch.ones(2, 2)], outputs=[z]).exec()
    assert torch.equal(z, torch.full((2, 2), 2.0))
    from deeplearning4j_amd.nd4j.ops.custom import UserDefinedCustomOp

    class Twice(UserDefinedCustomOp):
        def doExec(self, ins, i, t, b, d):
            return ins[0] * 2
    assert torch.equal(Twice("my_twice", [torch.ones(3)]).exec()[0], torch.full((3,), 2.0))
    assert describe("my_twice") is not None


def test_backend_spi_device_local_events_validator_inspect():
    import gc
    from deeplearning4j_amd.nd4j.backend import (CpuBackend, DeallocatorService, DeviceLocalNDArray, DeviceValidator, EventLogger,
                                                 Nd4jBackend, RocmBackend, UnifiedProfiler, inspectArray, DL4JSystemProperties)
    b = Nd4jBackend.load()
    assert isinstance(b, RocmBackend if torch.cuda.is_available() else CpuBackend)
    assert "opexec" in b.getProperties() and b.device().type in ("cpu", "cuda")
    assert RocmBackend().getPriority() > CpuBackend().getPriority()
    assert DL4JSystemProperties.get(DL4JSystemProperties.UI_SERVER_PORT_PROPERTY, "9000") == os.environ.get("DL4J_UI_PORT", "9000")
    # device-local replicas: delayed materialisation, update in place
    src = torch.arange(6.0).reshape(2, 3)
    dl = DeviceLocalNDArray(src, delayedMode=True)
    r0 = dl.get()
    assert torch.equal(r0.tensor().cpu(), src)
    dl.update(src * 2)
    assert torch.equal(dl.get().tensor().cpu(), src * 2)
    # event logger + deallocator service
    ev = EventLogger.getInstance()
    ev.events.clear()
    ev.setEnabled(True)
    seen = []
    ev.addEventLogListener(seen.append)
    freed = []

    class Owner:
        pass
    o = Owner()
    ds = DeallocatorService.getInstance()
    ds.pickObject(o, lambda: freed.append(1), nbytes=128)
    assert ds.numLive() >= 1
    del o
    gc.collect()
    assert freed == [1]
    agg = ev.aggregate()
    assert agg[("ALLOCATION", "DATA_BUFFER")] == [1, 128] and agg[("DEALLOCATION", "DATA_BUFFER")] == [1, 128] and len(seen) == 2
    ev.setEnabled(False)
    # blocked deallocation (DeallocationExtension)
    o2 = Owner()
    ds.toggleDeallocationBlock(True)
    ds.pickObject(o2, lambda: freed.append(2))
    del o2
    gc.collect()
    assert freed == [1]
    ds.toggleDeallocationBlock(False)
    assert freed == [1, 2]
    # unified profiler wraps op statistics + memory snapshots
    up = UnifiedProfiler.getInstance()
    up.start()
    from deeplearning4j_amd.nd4j.ops.custom import OpExecutioner
    OpExecutioner.getInstance().exec("add", [torch.ones(3), torch.ones(3)])
    rep = up.stop()
    assert len(rep["memory"]) >= 2 and rep["memory"][0]["tag"] == "start"
    # launch validation against gfx950 limits
    lim = DeviceValidator.limits()
    assert lim["waveSize"] == 64 and lim["numCu"] == 256 and lim["ldsPerCu"] == 160 * 1024
    ok = DeviceValidator.validate(2048, 256, 64 * 1024, 128)
    assert ok["valid"] and ok["wavesPerBlock"] == 4 and ok["blocksPerCu"] == 2 and ok["ldsLimited"]
    assert not DeviceValidator.validate(1, 2048)["valid"]
    assert not DeviceValidator.validate(1, 256, 200 * 1024)["valid"]
    assert DeviceValidator.validate(1, 96)["mask"] == DeviceValidator.PARTIAL_WAVE
    info = inspectArray(torch.tensor([0.0, 1.0, -2.0, float("nan"), float("inf"), 3.0]))
    assert (info.min, info.max, info.zeroCount, info.positiveCount, info.negativeCount, info.nanCount, info.infCount) == (-2.0, 3.0, 1, 3, 1, 1, 1)
    assert abs(info.mean - 0.5) < 1e-12


def test_layer_workspace_mgr():
    from deeplearning4j_amd.nd4j.workspace import ArrayType, LayerWorkspaceMgr, WorkspaceConfiguration
    mgr = (LayerWorkspaceMgr.builder()
           .with_(ArrayType.ACTIVATIONS, LayerWorkspaceMgr.WS_ALL_LAYERS_ACT, WorkspaceConfiguration(initialSize=1 << 20))
           .with_(ArrayType.FF_WORKING_MEM, LayerWorkspaceMgr.WS_LAYER_WORKING_MEM, WorkspaceConfiguration(initialSize=1 << 20))
           .noWorkspaceFor(ArrayType.INPUT).build())
    assert mgr.getWorkspaceName(ArrayType.ACTIVATIONS) == "WS_ALL_LAYERS_ACT" and mgr.isScopedOut(ArrayType.INPUT)
    with pytest.raises(ValueError):
        mgr.getWorkspaceName(ArrayType.BP_WORKING_MEM)
    mgr.assertNotOpen(ArrayType.ACTIVATIONS)
    with mgr.notifyScopeEntered(ArrayType.ACTIVATIONS) as ws:
        mgr.assertOpen(ArrayType.ACTIVATIONS)
        a = mgr.create(ArrayType.ACTIVATIONS, torch.float32, 4, 5)
        assert a.shape == (4, 5) and float(a.abs().sum()) == 0 and ws.getThisCycleAllocations() >= 80
        b = mgr.leverageTo(ArrayType.ACTIVATIONS, torch.ones(3))
        assert torch.equal(b, torch.ones(3))
        c = mgr.castTo(ArrayType.ACTIVATIONS, torch.float64, torch.ones(2))
        assert c.dtype == torch.float64
    mgr.assertNotOpen(ArrayType.ACTIVATIONS)
    x = mgr.createUninitialized(ArrayType.INPUT, torch.float32, [2, 2])     # scoped out: plain allocation
    assert x.shape == (2, 2)
    none = LayerWorkspaceMgr.noWorkspaces()
    assert all(none.isScopedOut(t) for t in ArrayType.ALL)


def test_op_profiler_sees_legacy_ops_and_panics_on_nan():
    """OpProfiler is hooked into the legacy executioner too (every transform / pairwise / reduce call), not only custom ops."""
    from deeplearning4j_amd.nd4j.environment import OpProfiler, ProfilerConfig
    from deeplearning4j_amd.nd4j.ops import legacy as LG
    prof = OpProfiler.getInstance()
    prof.reset()
    old = prof.config
    try:
        prof.setConfig(ProfilerConfig(checkForNAN=True, notOptimalArguments=True))
        ex = LG.executioner()
        x = torch.rand(4, 6) + 0.5
        ex.execTransformStrict(LG.op_num("transform_strict", "Sin"), x)
        ex.execPairwiseTransform(LG.op_num("pairwise", "Add"), x, x)
        ex.execReduceSame(LG.op_num("reduce_same", "Sum"), x.t(), [0])
        assert prof.getInvocationsCount("Sin") == 1 and prof.getInvocationsCount("Add") == 1 and prof.getInvocationsCount("Sum") == 1
        assert prof.penalties.get("STRIDED_ACCESS:Sum") == 1
        with pytest.raises(FloatingPointError):
            ex.execTransformStrict(LG.op_num("transform_strict", "Log"), -x)
    finally:
        prof.setConfig(old)
        prof.reset()
