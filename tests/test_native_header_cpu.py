"""CPU: the public C header of the native runtime (native/include/nd4jrocm/NativeOps.h, what a JavaCPP preset parses - the reference's
nd4j-cuda-preset does the same with libnd4j's NativeOps.h) is current, self-contained in C and C++, and a program built against it
links to libnd4jrocm_rt.so and executes an op through the OpaqueContext ABI."""
import importlib.util
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
INC = ROOT / "deeplearning4j_amd" / "native" / "include"
LIBDIR = ROOT / "deeplearning4j_amd" / "native"


def _gen():
    spec = importlib.util.spec_from_file_location("gen_native_header", ROOT / "tools" / "gen_native_header.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_header_is_current_and_covers_the_exported_symbols():
    from deeplearning4j_amd.native import build as B
    B.build()
    gen = _gen()
    text, names = gen.render()
    assert gen.OUT.read_text() == text, "stale header: run python tools/gen_native_header.py"
    assert len(names) >= 150
    nm = shutil.which("nm")
    if nm is None:
        pytest.skip("nm not available")
    out = subprocess.run([nm, "-D", "--defined-only", str(LIBDIR / "libnd4jrocm_rt.so")], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    missing = [n for n in names if n not in exported]
    assert not missing, missing[:10]                       # every declared function is an exported symbol of the library


PROGRAM = r"""
#include "nd4jrocm/NativeOps.h"
#include <stdio.h>
int main(void) {
  /* z = x + y on host buffers through the OpaqueContext ABI (the sequence NativeOpExecutioner.exec(CustomOp) performs) */
  float x[6] = {1, 2, 3, 4, 5, 6}, y[6] = {10, 20, 30, 40, 50, 60}, z[6] = {0};
  LongType shape[2] = {2, 3};
  LongType si[16];
  if (nd4j_shape_buffer(2, shape, 'c', 8192 /* FLOAT32 flag of the shapeInfo extras word */, si) != 0) return 2;
  void* in_buf[2] = {x, y}; void* in_si[2] = {si, si};
  void* out_buf[1] = {z}; void* out_si[1] = {si};
  OpaqueContext* ctx = createGraphContext(1);
  setGraphContextInputArrays(ctx, 2, in_buf, in_si, 0, 0);
  setGraphContextOutputArrays(ctx, 1, out_buf, out_si, 0, 0);
  int rc = execCustomOp2(0, ops_hash("add"), ctx);
  deleteGraphContext(ctx);
  if (rc != 0) { printf("rc %d\n", rc); return 3; }
  for (int i = 0; i < 6; ++i) if (z[i] != x[i] + y[i]) return 4;
  printf("ok %g %g\n", z[0], z[5]);
  return 0;
}
"""


@pytest.mark.parametrize("cc,std,ext", [("gcc", "-std=c11", "c"), ("g++", "-std=c++17", "cpp")])
def test_program_against_the_header_links_and_runs_an_op(tmp_path, cc, std, ext):
    if shutil.which(cc) is None:
        pytest.skip(cc + " not available")
    from deeplearning4j_amd.native import build as B
    B.build()
    src = tmp_path / ("prog." + ext)
    src.write_text(PROGRAM)
    exe = tmp_path / "prog"
    r = subprocess.run([cc, std, "-Wall", "-Werror", "-I", str(INC), str(src), "-L", str(LIBDIR), "-lnd4jrocm_rt",
                        "-Wl,-rpath," + str(LIBDIR), "-o", str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.startswith("ok 11 66"), (r.returncode, r.stdout[-500:])
