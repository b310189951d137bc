"""ThreadSanitizer and AddressSanitizer/UBSan over the host side of the library (worker pool, threaded quadtree
build with its folds running beside everything else, pipelined flatten, threaded nb_draw): host_ops.cpp and host_tree.cpp
are compiled with g++ (it needs no device) together with tools/sanitize/host_main.cpp and must run without a single report."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rust-exp_amd", "csrc")


@pytest.mark.parametrize("san,threads", [("thread", "8"), ("address,undefined", "8"), ("address,undefined", "1")])
def test_host_code_is_sanitizer_clean(tmp_path, san, threads):
    cxx = shutil.which("g++")
    if not cxx or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("needs g++ and the HIP headers")
    exe = os.path.join(str(tmp_path), "host_san")
    cmd = [cxx, "-std=c++17", "-O1", "-g", "-fsanitize=" + san, "-ffp-contract=off", "-I/opt/rocm/include",
           "-D__HIP_PLATFORM_AMD__", "-I" + CSRC, os.path.join(CSRC, "host_ops.cpp"), os.path.join(CSRC, "host_tree.cpp"),
           os.path.join(ROOT, "tools", "sanitize", "host_main.cpp"), "-o", exe, "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("this g++ has no -fsanitize=" + san)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, NBX_HOST_THREADS=threads, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="halt_on_error=0")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "Sanitizer" not in out and "runtime error" not in out, out[-3000:]
    assert out.count("rc=0") == 5 and "lit=" in out
