"""Exception types under the reference's names and the mapping of native failures onto them.

Reference: nd4j-api/.../linalg/exception/{ND4JException, ND4JIllegalStateException, ND4JIllegalArgumentException,
ND4JArraySizeException, ND4JIllegalAccessException, ND4JOpProfilerException, ND4JUnknownDataTypeException,
ND4UnresolvedOutputVariables, Nd4jNoSuchWorkspaceException}.java and, on the native side, libnd4j/include/exceptions/*.h
(cuda_exception, allocation_exception, datatype_exception, graph_exception and its four subclasses) whose messages reach Java through
lastErrorCode / lastErrorMessage. The host runtime (native/csrc/rt_exceptions.*) reports a category code with every failure and, when
``ND4J_BACKTRACE=1`` or ``enable_native_backtraces()``, the demangled C++ stack of the throw site; ``raise_native`` turns that into the
matching class below with the stack attached as ``nativeBacktrace``."""
from __future__ import annotations


class ND4JException(RuntimeError):
    nativeBacktrace: str = ""
    nativeCode: int = 0


class ND4JIllegalStateException(ND4JException):
    pass


class ND4JIllegalArgumentException(ND4JException, ValueError):
    pass


class ND4JArraySizeException(ND4JException, OverflowError):
    pass


class ND4JIllegalAccessException(ND4JException):
    pass


class ND4JOpProfilerException(ND4JException):
    pass


class ND4JUnknownDataTypeException(ND4JException, TypeError):
    pass


class ND4UnresolvedOutputVariables(ND4JException):
    pass


class Nd4jNoSuchWorkspaceException(ND4JException, KeyError):
    pass


# native categories (libnd4j/include/exceptions/*.h)
class NativeException(ND4JException):
    """Base of failures reported by the native libraries."""


class DeviceException(NativeException):           # cuda_exception
    pass


class AllocationException(NativeException, MemoryError):
    pass


class DatatypeException(NativeException, ND4JUnknownDataTypeException):
    pass


class GraphException(NativeException):
    pass


class GraphExecutionException(GraphException):
    pass


class GraphExistsException(GraphException):
    pass


class NoResultsException(GraphException):
    pass


class UnknownGraphException(GraphException, KeyError):
    pass


_BY_CATEGORY = {
    100: NativeException, 101: DeviceException, 102: AllocationException, 103: DatatypeException, 110: GraphException,
    111: GraphExecutionException, 112: GraphExistsException, 113: NoResultsException, 114: UnknownGraphException,
    120: ND4JIllegalArgumentException, 121: ND4JIllegalArgumentException, 122: AllocationException,
}


def exception_class(code: int):
    return _BY_CATEGORY.get(int(code), NativeException)


def enable_native_backtraces(on: bool = True):
    """Capture the C++ stack at every native throw site (costs a stack walk per failure; off by default, also ND4J_BACKTRACE=1)."""
    from ..native import lib as L
    L.rt().nd4j_backtrace_toggle(1 if on else 0)


def native_backtrace() -> str:
    """Demangled native stack of the CALLER (debugging aid, the role of the reference's backward.hpp printer)."""
    import ctypes
    from ..native import lib as L
    n = L.rt().nd4j_current_backtrace(None, 0)
    buf = ctypes.create_string_buffer(int(n) + 1)
    L.rt().nd4j_current_backtrace(buf, int(n) + 1)
    return buf.value.decode(errors="replace")


def install_crash_handler():
    """Fatal signals inside native code print the native stack to stderr before the process dies."""
    from ..native import lib as L
    L.rt().nd4j_install_crash_handler()


def raise_native(code: int, what: str = "native runtime"):
    """Raise the Python class for a non-zero status of a runtime entry point, with message and (if captured) native stack; clears
    the error slot."""
    from ..native import lib as L
    rt = L.rt()
    msg = (rt.nd4j_last_error_message() or b"").decode(errors="replace")
    bt = (rt.nd4j_last_error_backtrace() or b"").decode(errors="replace")
    cat = (rt.nd4j_error_category(int(code)) or b"").decode()
    rt.nd4j_clear_error()
    exc = exception_class(code)("%s failed: %s (code %d%s)" % (what, msg, code, ", " + cat if cat else ""))
    exc.nativeCode, exc.nativeBacktrace = int(code), bt
    if bt:
        exc.args = (exc.args[0] + "\nnative stack:\n" + bt,)
    raise exc
