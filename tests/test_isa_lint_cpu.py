"""CPU: ISA lint of the hand-scheduled convolution loaders (tools/check_async_loads.py).

conv_igemm.hip keeps loads in flight that the compiler does not know about; that is only correct while the compiler never touches
their destination registers. The check compiles the file to assembly with the flags of the build (hipcc cross-compiles without a
GPU) and inspects every kernel."""
import importlib.util
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _lint():
    spec = importlib.util.spec_from_file_location("check_async_loads", ROOT / "tools" / "check_async_loads.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_hidden_register_loaders_are_never_touched_by_the_compiler(tmp_path):
    from deeplearning4j_amd.native import build as B
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("hipcc not available")
    out = tmp_path / "conv_igemm.s"
    cmd = [hipcc, f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", *B.EXTRA_FLAGS.get("conv_igemm.hip", ()), "-I", str(B.CSRC), "-S",
           "--cuda-device-only", str(B.CSRC / "conv_igemm.hip"), "-o", str(out)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    bad, seen = _lint().check(str(out))
    assert len(seen) >= 12, sorted(seen)                    # every forward / data-gradient / weight-gradient instantiation was inspected
    assert not bad, bad[:10]
    assert "scratch_" not in out.read_text()                # and none of them spills to scratch memory


def test_lint_flags_the_unsafe_forms(tmp_path):
    """The checker itself: an inline-asm load into vector registers, and a compiler instruction naming a hidden register."""
    src = tmp_path / "k.s"
    src.write_text("""
_Z1kv:
\t;;#ASMSTART
\tglobal_load_dwordx4 v[4:7], v[2:3], off
\t;;#ASMEND
\t;;#ASMSTART
\tglobal_load_dwordx4 a[0:3], v[2:3], off
\t;;#ASMEND
\tv_accvgpr_read_b32 v9, a2
\tv_accvgpr_read_b32 v9, a8
\ts_endpgm
\t.amdhsa_kernel _Z1kv
\t\t.amdhsa_next_free_vgpr 16
\t\t.amdhsa_accum_offset 12
""")
    bad, seen = _lint().check(str(src))
    assert seen == {"_Z1kv": 3}
    assert len(bad) == 2 and "vector registers" in bad[0] and "a0..a3" in bad[1], bad
