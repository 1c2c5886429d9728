"""Sanitizer runs of the host side (CPU): the shim's hand-rolled WorkerPool (one atomic ticket word + a generation
counter, two std::function slots: host/SfMBundleAdjustmentUtils.cpp) under ThreadSanitizer and under AddressSanitizer + UBSan, the CPU oracle under
AddressSanitizer + UBSan -- `make -C sfm-toy-library_amd/host tsan|asan`, `make -C oracle asan`.  Each is one instrumented executable (a Python process
loading an instrumented .so is not a whole-program-instrumented process); halt_on_error / -fno-sanitize-recover: any report fails the target."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "sfm-toy-library_amd", "host")
CSRC = os.path.join(ROOT, "sfm-toy-library_amd", "csrc")


def _san_works(flag):
    """the sanitizer runtimes are part of gcc here; probe instead of assuming"""
    if shutil.which("g++") is None:
        return False
    r = subprocess.run("echo 'int main(){return 0;}' | g++ -x c++ - %s -o /tmp/_sfmba_san_probe && /tmp/_sfmba_san_probe" % flag, shell=True, capture_output=True)
    return r.returncode == 0


def _make(directory, target, timeout=900):
    return subprocess.run(["make", "-C", directory, target], capture_output=True, text=True, timeout=timeout)


@pytest.mark.skipif(not os.path.exists(os.path.join(CSRC, "libsfmba_hip.so")), reason="libsfmba_hip.so not built (the shim links it)")
def test_worker_pool_under_thread_sanitizer():
    if not _san_works("-fsanitize=thread"):
        pytest.skip("no ThreadSanitizer runtime")
    r = _make(HOST, "tsan")
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "pool_sanitize: 0 wrong batch(es)" in r.stdout and "ThreadSanitizer" not in r.stdout + r.stderr


@pytest.mark.skipif(not os.path.exists(os.path.join(CSRC, "libsfmba_hip.so")), reason="libsfmba_hip.so not built (the shim links it)")
def test_worker_pool_under_address_and_ub_sanitizer():
    if not _san_works("-fsanitize=address,undefined"):
        pytest.skip("no AddressSanitizer runtime")
    r = _make(HOST, "asan")
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "pool_sanitize: 0 wrong batch(es)" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_oracle_under_address_and_ub_sanitizer():
    if not _san_works("-fsanitize=address,undefined"):
        pytest.skip("no AddressSanitizer runtime")
    r = _make(os.path.join(ROOT, "oracle"), "asan")
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "oracle sanitize driver: ok" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
