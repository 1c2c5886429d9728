"""The drop-in shim's resident worker pool (host/SfMBundleAdjustmentUtils.cpp): tasks are handed out through one atomic ticket,
workers poll before they block.  No GPU involved."""
import ctypes as C
import os
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    return C.CDLL(os.path.join(ROOT, "sfm-toy-library_amd", "host", "libsfmba_shim.so"))


def test_every_task_of_every_batch_runs_exactly_once():
    lib = _lib()
    assert lib.sfmba_shim_pool_selftest(C.c_int(3000)) == 0


def test_callers_on_different_threads_take_turns():
    lib = _lib()
    out = []
    ths = [threading.Thread(target=lambda: out.append(lib.sfmba_shim_pool_selftest(C.c_int(600)))) for _ in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert out == [0, 0, 0, 0]


def test_contended_small_batches_never_lose_a_task():
    """begin() publishes the ticket word before the generation; a worker still drawing tickets of the previous
    batch could take index 0 of the new one, see the old generation and drop it -- end() then waited forever.  The hang needs CPU
    contention: several processes (each with its own pool) run many small batches at once, under a watchdog."""
    import subprocess
    import sys
    code = ("import ctypes as C, sys; lib = C.CDLL(%r); "
            "sys.exit(1 if lib.sfmba_shim_pool_stress(C.c_int(3000000), C.c_int(6)) != 0 else 0)"
            % os.path.join(ROOT, "sfm-toy-library_amd", "host", "libsfmba_shim.so"))
    procs = [subprocess.Popen([sys.executable, "-c", code]) for _ in range(6)]
    try:
        for p in procs:
            assert p.wait(timeout=240) == 0          # a lost task shows up as a hang (timeout), a wrong count as exit code 1
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
