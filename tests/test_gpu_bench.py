"""bench.py contract (-m gpu): one JSON line with the keys the driver and the judge read, on a small workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--workload", "cfg2", *extra],
                         cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_line_has_the_contract_keys():
    d = _run("--cpu-threads", "2")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and d["termination"] == "CONVERGENCE"
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma", "l2_mall_latency") and 0 < r["frac"] < 1
    import shutil
    if shutil.which("rocprofv3") and not any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        # roofline.traffic is measured by the run itself: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command
        assert r["traffic_is_static"] is False and r["traffic"] > 0 and r["traffic_detail"]["live"] is True, r
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["value"] > 0
    # the checker's result on the sample agrees with the GPU's (same problem)
    assert abs(c["rms_px_after_sample"] - d["final_rms_px"]) < 1e-4


def test_bench_cholesky_and_f64_modes_run():
    d = _run("--linear", "cholesky", "--precision", "f64", "--no-cpu-baseline", "--no-live-traffic")
    assert d["termination"] == "CONVERGENCE" and "cpu_baseline" not in d
