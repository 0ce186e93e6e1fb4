"""CrashReportingUtil — memory crash dumps when training runs out of device memory.

Reference: deeplearning4j-nn util/CrashReportingUtil.java (writeMemoryCrashDump / generateMemoryStatus; enabled by
default, ``crashDumpsEnabled(false)`` or system property ``org.deeplearning4j.crash.reporting.enabled`` turns it off,
``crashDumpOutputDirectory`` redirects the dump). MI355X version: reports HBM usage from the HIP caching allocator,
host RSS, workspace / native memory counters and the network's MemoryReport.
"""
from __future__ import annotations

import datetime
import os
import platform
import traceback

import torch

_ENABLED = os.environ.get("DL4J_CRASH_REPORTING_ENABLED", "true").lower() != "false"
_DIR = os.environ.get("DL4J_CRASH_REPORTING_DIR")


class CrashReportingUtil:
    @staticmethod
    def crashDumpsEnabled(enabled: bool):
        global _ENABLED
        _ENABLED = bool(enabled)

    @staticmethod
    def crashDumpOutputDirectory(path):
        global _DIR
        _DIR = None if path is None else str(path)

    @staticmethod
    def writeMemoryCrashDump(net, exc: BaseException):
        if not _ENABLED:
            return None
        d = _DIR or os.getcwd()
        os.makedirs(d, exist_ok=True)
        ts = datetime.datetime.now().strftime("%Y%m%d-%H%M%S")
        path = os.path.join(d, "dl4j-memory-crash-dump-%s_%d.txt" % (ts, os.getpid()))
        with open(path, "w") as f:
            f.write("Deeplearning4j (MI355X runtime) OOM Exception Encountered for %s\n" % type(net).__name__)
            f.write("Timestamp: %s\nPID: %d\n\n" % (datetime.datetime.now().isoformat(), os.getpid()))
            f.write("Exception: %s\n%s\n" % (exc, "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))))
            f.write(CrashReportingUtil.generateMemoryStatus(net))
        return path

    @staticmethod
    def generateMemoryStatus(net, minibatch=None, *inputTypes) -> str:
        out = ["========== Memory Information =========="]
        try:
            import psutil
            p = psutil.Process()
            vm = psutil.virtual_memory()
            out.append("Host RSS: %d bytes; host total %d, available %d" % (p.memory_info().rss, vm.total, vm.available))
        except Exception:
            pass
        out.append("Host: %s %s, python %s, torch %s" % (platform.system(), platform.release(), platform.python_version(), torch.__version__))
        if torch.cuda.is_available():
            for i in range(torch.cuda.device_count()):
                free, total = torch.cuda.mem_get_info(i)
                out.append("Device %d (%s): total %d, free %d, allocated %d, reserved %d, peak allocated %d" % (
                    i, torch.cuda.get_device_name(i), total, free, torch.cuda.memory_allocated(i), torch.cuda.memory_reserved(i),
                    torch.cuda.max_memory_allocated(i)))
        try:
            from ..nd4j.workspace import Nd4jWorkspaceManager
            mgr = Nd4jWorkspaceManager.getInstance() if hasattr(Nd4jWorkspaceManager, "getInstance") else None
            if mgr is not None and hasattr(mgr, "getAllWorkspacesForCurrentThread"):
                out.append("========== Workspace Information ==========")
                for ws in mgr.getAllWorkspacesForCurrentThread():
                    out.append("  %s: size %s" % (getattr(ws, "id", ws), getattr(ws, "getCurrentSize", lambda: "?")()))
        except Exception:
            pass
        out.append("========== Network Information ==========")
        try:
            out.append("Network # parameters: %d" % net.numParams())
            out.append("Iteration %d, epoch %d" % (net.getIterationCount(), net.getEpochCount()))
            out.append("Updater state elements: %d" % int(getattr(net, "_state_size", 0)))
            ctx = getattr(net, "ctx", None)
            if ctx is not None:
                out.append("Parameter dtype %s, compute dtype %s, device %s" % (ctx.dtype, ctx.cdtype, ctx.device))
        except Exception as e:
            out.append("(network details unavailable: %s)" % e)
        try:
            conf = net.conf
            its = list(inputTypes) or ([conf.inputType] if getattr(conf, "inputType", None) is not None else list(getattr(conf, "inputTypes", None) or []))
            if its and all(t is not None for t in its):
                rep = conf.getMemoryReport(*its)
                out.append(rep.toString(int(minibatch or getattr(net, "_last_batch", None) or 32)))
        except Exception as e:
            out.append("(memory report unavailable: %s)" % e)
        try:
            out.append("========== Layer summary ==========")
            out.append(net.summary())
        except Exception:
            pass
        return "\n".join(out) + "\n"


def _is_oom(e: BaseException) -> bool:
    return isinstance(e, torch.cuda.OutOfMemoryError) or (isinstance(e, (RuntimeError, MemoryError)) and "out of memory" in str(e).lower())


def install(cls):
    """Wrap ``cls.fit``: an out-of-memory error writes the crash dump, then propagates."""
    if getattr(cls.fit, "_crash_wrapped", False):
        return cls
    orig = cls.fit

    def fit(self, *a, **kw):
        try:
            return orig(self, *a, **kw)
        except BaseException as e:
            if _is_oom(e):
                try:
                    p = CrashReportingUtil.writeMemoryCrashDump(self, e)
                    if p:
                        print("Memory crash dump written to %s" % p)
                except Exception:
                    pass
            raise
    fit._crash_wrapped = True
    fit.__doc__ = orig.__doc__
    cls.fit = fit
    return cls
