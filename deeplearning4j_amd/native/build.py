"""In-tree build of the native libraries (gfx950 only).

Three shared objects are produced next to this file:

* ``libdl4j_hip.so``   – every hand-written HIP kernel (``csrc/*.hip``), compiled with
  ``hipcc --offload-arch=gfx950`` and exposed through a flat C ABI (ctypes on the Python side,
  the same ABI a JavaCPP preset would bind — mirrors the role of libnd4j's NativeOps.h).
* ``libnd4jrocm_dev.so`` – the memory / device / stream / event / launch-context / sort groups of the C ABI on the HIP runtime and
  rocPRIM (``csrc/abi_runtime.hip``), under the reference's function names.
* ``libnd4jrocm_rt.so`` – the host C++ runtime (``csrc/*.cpp``): shapeInfo/TAD math, workspaces,
  DataBuffer serde, threshold codecs (host), record readers + prefetch loader, op registry.

hipcc cross-compiles without a GPU, so this runs on the CPU build box; the ``.so`` files travel
to the GPU box with the tree.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = HERE / "_obj"
ARCH = "gfx950"
HIP_LIB = HERE / "libdl4j_hip.so"
RT_LIB = HERE / "libnd4jrocm_rt.so"
DEV_LIB = HERE / "libnd4jrocm_dev.so"
DEV_SRCS = ("abi_runtime.hip",)      # device-runtime ABI under the reference's unprefixed names: its own library, loaded RTLD_LOCAL


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


# Per-file compiler flags. conv_igemm.hip keeps its in-flight loads in accumulation registers that only inline asm names, so the
# compiler must keep the MFMA accumulators in vector registers there (with accumulation registers in the clobber lists it would
# otherwise select the accumulation-register form of the MFMAs). tools/check_async_loads.py verifies the generated code.
EXTRA_FLAGS = {"conv_igemm.hip": ("-mllvm", "-amdgpu-mfma-vgpr-form")}


def _newer(src: Path, dst: Path, extra=()) -> bool:
    if not dst.exists():
        return True
    t = dst.stat().st_mtime
    return any(p.stat().st_mtime > t for p in (src, *extra))


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(map(str, cmd)), r.stdout))
    return r.stdout


def build(force: bool = False, verbose: bool = False, jobs: int | None = None) -> None:
    OBJ.mkdir(exist_ok=True)
    hipcc = _hipcc()
    headers = list(CSRC.glob("*.h")) + list(CSRC.glob("*.hpp")) + list(CSRC.glob("*.inc"))
    jobs = jobs or min(8, os.cpu_count() or 4)

    hip_srcs = sorted(CSRC.glob("*.hip"))
    cpp_srcs = sorted(CSRC.glob("*.cpp"))

    tasks = []
    for s in hip_srcs:
        o = OBJ / (s.stem + ".hip.o")
        if force or _newer(s, o, headers):
            tasks.append([hipcc, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden",
                          "-Wno-unused-result", "-Wno-unused-value", *EXTRA_FLAGS.get(s.name, ()), "-I", str(CSRC), "-c", str(s),
                          "-o", str(o)])
    for s in cpp_srcs:
        o = OBJ / (s.stem + ".cpp.o")
        if force or _newer(s, o, headers):
            tasks.append(["g++", "-O2", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-pthread", "-I", str(CSRC),
                          "-c", str(s), "-o", str(o)])
    if tasks:
        with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
            for out in ex.map(_run, tasks):
                if verbose and out.strip():
                    print(out)
    hip_objs = [str(OBJ / (s.stem + ".hip.o")) for s in hip_srcs if s.name not in DEV_SRCS]
    dev_objs = [str(OBJ / (s.stem + ".hip.o")) for s in hip_srcs if s.name in DEV_SRCS]
    cpp_objs = [str(OBJ / (s.stem + ".cpp.o")) for s in cpp_srcs]
    if hip_objs and (force or tasks or not HIP_LIB.exists()):
        _run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(HIP_LIB), *hip_objs])
    if dev_objs and (force or tasks or not DEV_LIB.exists()):
        _run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(DEV_LIB), *dev_objs])
    if cpp_objs and (force or tasks or not RT_LIB.exists()):
        _run(["g++", "-shared", "-fPIC", "-pthread", "-o", str(RT_LIB), *cpp_objs])


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print("built:", HIP_LIB if HIP_LIB.exists() else "-", RT_LIB if RT_LIB.exists() else "-")
