"""not-gpu: the HOST side of libaesr_hip under AddressSanitizer + UBSan (round-5 verdict, next 7; SURVEY section 5 asks for sanitizer builds of the
host code in tests -- GPU sanitizers are not available on this pool).

csrc/Makefile target ``asan`` compiles the SAME sources with the host code instrumented (device code untouched) into libaesr_hip_asan.so;
tests/host_sanitized_sweep.py, a torch-free child process with the sanitizer runtime preloaded, sweeps the planners (plan_conv / plan_wino /
plan_wgrad / plan_ring), kernel selection, workspace and slab arithmetic and the item decompositions over BASELINE, random and oversized shapes --
queries directly, launch paths through the real entry points with fake device pointers (no GPU is visible: the launch itself fails with a HIP error
after all the host code has run).  Any sanitizer report aborts the child.

Found on its first run (fixed in the same commit): ceil_div(INT_MAX, 2) overflowing inside plan_wino when a QUERY was asked about an absurd batch
(the launch path validated sizes only after planning), and int overflow of H * W in the VIF workspace layout for a 46341 x 46341 slice."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "superresolution_aniso_mri_amd", "csrc")
LIB = os.path.join(ROOT, "superresolution_aniso_mri_amd", "libaesr_hip_asan.so")


def _runtime():
    c = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return c[-1] if c else None


@pytest.mark.parametrize("seed", [1, 2])
def test_planners_and_tables_under_asan_ubsan(seed):
    rt = _runtime()
    if rt is None:
        pytest.skip("no clang AddressSanitizer runtime in this image")
    r = subprocess.run(["make", "-C", CSRC, "-j", str(min(6, os.cpu_count() or 1)), "asan"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and os.path.exists(LIB), r.stdout[-3000:]
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               HIP_VISIBLE_DEVICES="-1")            # launch entry points get fake pointers: no device may be visible
    env.pop("AESR_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "host_sanitized_sweep.py"), LIB, str(seed), "400"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    if r.returncode == 3 and "REFUSED" in r.stdout:
        pytest.skip("a GPU is visible to the HIP runtime: the sweep only runs on CPU-only boxes")
    assert r.returncode == 0 and "SANITIZED SWEEP OK" in r.stdout, r.stdout[-4000:]
    assert "runtime error" not in r.stdout and "AddressSanitizer" not in r.stdout, r.stdout[-4000:]
