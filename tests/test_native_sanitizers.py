"""The kernels' numerics and wavefront orchestration (host-compiled, tests/native/*.cpp) under AddressSanitizer +
UndefinedBehaviorSanitizer: an out-of-bounds index into an LDS-resident array or an uninitialised read is silent corruption on the
GPU; here it aborts the run."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_host_compiled_kernel_code_is_clean_under_asan_ubsan(oracle_lib):
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan is not available")
    out_dir = os.path.join(HERE, "native", "_build")
    os.makedirs(out_dir, exist_ok=True)
    sos = []
    for name in ("relpose_core_host", "guided_host"):
        so = os.path.join(out_dir, name + "_asan.so")
        subprocess.check_call(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-ffp-contract=off", "-fPIC",
                               "-shared", "-std=c++17", "-o", so, os.path.join(HERE, "native", name + ".cpp")])
        sos.append(so)
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "native", "sanitizer_run.py")] + sos, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "relpose harness under ASan/UBSan: clean" in r.stdout and "guided harness under ASan/UBSan: clean" in r.stdout
