"""Host runtime under ASan + UBSan. Compiling 11 translation units with sanitizers takes ~1-2 minutes, so the run is
opt-in (DL4J_AMD_RUN_SANITIZER=1); its recorded output lives in profiles/r2_sanitizer_host_runtime.md."""
import glob
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
NATIVE = os.path.join(os.path.dirname(HERE), "deeplearning4j_amd", "native")


@pytest.mark.skipif(os.environ.get("DL4J_AMD_RUN_SANITIZER") != "1", reason="opt-in: set DL4J_AMD_RUN_SANITIZER=1")
def test_host_runtime_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "rt_sanitize")
    srcs = sorted(glob.glob(os.path.join(NATIVE, "csrc", "*.cpp")))
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I",
           os.path.join(NATIVE, "csrc"), os.path.join(NATIVE, "tests_native", "rt_sanitize.cpp"), *srcs, "-o", exe]
    subprocess.run(cmd, check=True)
    fixtures = sorted(glob.glob("/root/reference/libnd4j/tests_cpu/resources/*.fb"))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", TMPDIR=str(tmp_path))
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe, *fixtures], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "rt_sanitize OK" in r.stdout, r.stdout[-3000:]


def test_sanitizer_workload_source_is_part_of_the_native_tests():
    assert os.path.exists(os.path.join(NATIVE, "tests_native", "rt_sanitize.cpp"))
    cm = open(os.path.join(NATIVE, "CMakeLists.txt")).read()
    assert "SD_SANITIZE" in cm and "rt_sanitize" in cm
