"""Typed native exceptions, stack traces and the crash handler (reference: libnd4j/include/exceptions/*.h, backward.hpp,
nd4j-api linalg/exception/*.java)."""
import subprocess
import sys

import pytest

from deeplearning4j_amd.native import lib as L
from deeplearning4j_amd.nd4j import exceptions as E


@pytest.fixture(autouse=True)
def _backtraces_off_afterwards():
    yield
    E.enable_native_backtraces(False)
    L.rt().nd4j_clear_error()


@pytest.mark.parametrize("kind,cls,needle", [
    (1, E.DeviceException, "Error code: [719]"), (2, E.AllocationException, "Limit bytes: [1073741824]; Requested bytes: [719]"),
    (3, E.DatatypeException, "Expected: [FLOAT32]; Actual: [BFLOAT16]"), (4, E.GraphExecutionException, "graph execution: [719]"),
    (5, E.GraphExistsException, "already exists: [719]"), (6, E.NoResultsException, "no results"), (7, E.UnknownGraphException, "Unknown graph: [719]"),
    (8, E.ND4JIllegalArgumentException, "rank must be positive"), (9, E.ND4JIllegalArgumentException, "dimension 5"),
    (10, E.AllocationException, "bad_alloc"), (0, E.NativeException, "generic failure")])
def test_every_native_exception_type_arrives_as_its_python_class(kind, cls, needle):
    rc = L.rt().nd4j_raise_for_test(kind, 719)
    assert rc != 0
    with pytest.raises(cls) as ei:
        L.rt_check(rc, "probe")
    assert needle in str(ei.value) and isinstance(ei.value, E.ND4JException) and isinstance(ei.value, RuntimeError)
    assert ei.value.nativeCode == rc and ei.value.nativeBacktrace == ""          # capture is off by default
    assert L.rt().nd4j_last_error_code() == 0                                     # slot cleared by the raise


def test_backtrace_names_the_throw_site_when_enabled():
    E.enable_native_backtraces(True)
    assert L.rt().nd4j_backtrace_enabled() == 1
    rc = L.rt().nd4j_raise_for_test(3, 1)
    with pytest.raises(E.DatatypeException) as ei:
        L.rt_check(rc, "probe")
    bt = ei.value.nativeBacktrace
    assert "nd4j_raise_for_test" in bt and "libnd4jrocm_rt.so" in bt and bt.startswith("#0 "), bt
    assert "native stack:" in str(ei.value)
    here = E.native_backtrace()
    assert "nd4j_current_backtrace" in here and len(here.splitlines()) >= 3


def test_graph_abi_reports_typed_failures():
    from deeplearning4j_amd.autodiff import native_graph as NG
    lib = NG._lib() if hasattr(NG, "_lib") else L.rt()
    rc = L.rt().graph_register(991, b"\x00\x01", 2)                                # truncated FlatBuffers
    with pytest.raises(E.NativeException, match="graph_register"):
        L.rt_check(rc, "graph_register")
    assert L.rt().graph_num_nodes(424242) == -1
    assert L.rt().nd4j_last_error_code() == 114
    with pytest.raises(E.UnknownGraphException, match=r"Unknown graph: \[424242\]"):
        L.rt_check(L.rt().nd4j_last_error_code(), "graph_num_nodes")
    del lib


def test_java_named_exception_hierarchy():
    for name in ("ND4JIllegalStateException", "ND4JIllegalArgumentException", "ND4JArraySizeException", "ND4JIllegalAccessException",
                 "ND4JOpProfilerException", "ND4JUnknownDataTypeException", "ND4UnresolvedOutputVariables", "Nd4jNoSuchWorkspaceException"):
        assert issubclass(getattr(E, name), E.ND4JException)
    assert issubclass(E.ND4JIllegalArgumentException, ValueError) and issubclass(E.Nd4jNoSuchWorkspaceException, KeyError)
    assert E.exception_class(111) is E.GraphExecutionException and E.exception_class(999) is E.NativeException


def test_crash_handler_prints_the_native_stack():
    code = ("import ctypes, sys\n"
            "from deeplearning4j_amd.nd4j.exceptions import install_crash_handler\n"
            "install_crash_handler()\n"
            "sys.stderr.write('armed\\n'); sys.stderr.flush()\n"
            "ctypes.string_at(0)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "armed" in r.stderr and "fatal signal, native backtrace follows" in r.stderr, r.stderr[-2000:]
