"""The host packer (pack.cpp: validation + DFS compaction of the 26-byte node records, threaded over forests) under
AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer — CPU build only (GPU sanitizers are not
available on the pool).  The harness is tests/native/pack_fuzz.cpp."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_and_run(tmp_path, name, flags):
    # pack.cpp shares common.h (device helpers) with the kernels, so it is compiled as HIP, host side only, by ROCm's clang
    cxx = "/opt/rocm/lib/llvm/bin/clang++" if os.path.exists("/opt/rocm/lib/llvm/bin/clang++") else shutil.which("hipcc")
    if cxx is None:
        pytest.skip("no ROCm clang")
    exe = os.path.join(str(tmp_path), name)
    obj = os.path.join(str(tmp_path), name + "_pack.o")
    common = ["-O1", "-g", "-std=c++17", "-pthread", *flags]
    build = subprocess.run([cxx, *common, "-x", "hip", "--cuda-host-only", "--offload-arch=gfx950", "-I/opt/rocm/include",
                            "-c", os.path.join(ROOT, "bark_amd", "csrc", "pack.cpp"), "-o", obj], capture_output=True, text=True)
    if build.returncode == 0:
        build = subprocess.run([cxx, *common, os.path.join(ROOT, "tests", "native", "pack_fuzz.cpp"), obj, "-o", exe,
                                "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    if build.returncode != 0 and ("cannot find" in build.stderr or "unrecognized" in build.stderr):
        pytest.skip("sanitizer runtime not installed: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert run.stdout.startswith("ok "), run.stdout
    n_ok, n_rejected = (int(x) for x in run.stdout.split()[1:3])
    assert n_ok >= 30 and n_rejected >= 5  # both the accepting and the rejecting paths ran
    assert "ERROR" not in run.stderr and "WARNING: ThreadSanitizer" not in run.stderr, run.stderr[-3000:]


def test_packer_under_address_and_ub_sanitizers(tmp_path):
    _build_and_run(tmp_path, "pack_asan", ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"])


def test_packer_under_thread_sanitizer(tmp_path):
    _build_and_run(tmp_path, "pack_tsan", ["-fsanitize=thread"])
