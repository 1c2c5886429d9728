"""bench.py's launcher (CPU; VERDICT r5 "next" 1): `python bench.py --gpus N` with no WORLD_SIZE in the environment must start N ranks itself -- there
is no path on which `--gpus 8` runs one rank and prints n_gpus = 1 -- must refuse when fewer than N devices are visible, and must refuse a --gpus that
contradicts the launcher's WORLD_SIZE.  `--launch-only` is the dry mode: every rank prints RANK / LOCAL_RANK / WORLD_SIZE and exits without a GPU.
What the N ranks then reproduce is the reference's whole-problem solve (BA.cpp:160-179) on N devices."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_plain_invocation_with_two_gpus_starts_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-only"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 2, r.stdout
    assert sorted(x["rank"] for x in rows) == [0, 1]
    assert sorted(x["local_rank"] for x in rows) == [0, 1]             # one device each
    assert all(x["world_size"] == 2 and x["gpus_arg"] == 2 and x["master_addr"] == "127.0.0.1" for x in rows)
    assert rows[0]["pid"] != rows[1]["pid"]


def test_single_gpu_invocation_stays_one_process():
    r = subprocess.run([sys.executable, BENCH, "--launch-only"], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert rows == [dict(rows[0], rank=0, local_rank=0, world_size=1, gpus_arg=1)] and rows[0]["master_addr"] is None


def test_refuses_when_fewer_devices_than_ranks_are_visible():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 64:
        pytest.skip("this box really has 64 GPUs")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "64"], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "refusing" in r.stderr and "--gpus 64" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_refuses_a_gpus_argument_that_contradicts_the_launcher():
    env = dict(_clean_env(), WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-only"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
    # without --gpus the launcher's count is taken
    r = subprocess.run([sys.executable, BENCH, "--launch-only"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["world_size"] == 4


def test_parity_check_beside_every_number():
    """parity_vs_oracle: what `parity_ok` means on the headline, cfg4 and `sharded.*` entries (stored oracle results; no oracle call)."""
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_final_costs.json")))
    for g in range(8):
        assert "cfg4.%d" % g in want and "cfg3.%d" % g in want
        assert want["cfg4.%d" % g]["n_cam"] == 25 and want["cfg4.%d" % g]["n_obs"] == 125000
    w = want["cfg5"]
    good = {"termination_name": "CONVERGENCE", "iterations": w["iterations"], "final_cost": w["final_cost"] * (1 + 5e-7)}
    assert b.parity_vs_oracle(good, w["n_obs"], "cfg5")["parity_ok"] is True
    assert b.parity_vs_oracle(dict(good, final_cost=w["final_cost"] * 1.00001), w["n_obs"], "cfg5")["parity_ok"] is False
    assert b.parity_vs_oracle(dict(good, iterations=w["iterations"] + 1), w["n_obs"], "cfg5")["parity_ok"] is False
    assert b.parity_vs_oracle(dict(good, termination_name="NO_CONVERGENCE"), w["n_obs"], "cfg5")["parity_ok"] is False
    assert b.parity_vs_oracle(good, w["n_obs"], "no_such_problem")["parity_ok"] is None
    # the headline's n_gpus is the rank count RCCL reports (rccl_rank_count -> sfmba_comm_size), never the --gpus argument
    src = open(BENCH).read()
    assert '"n_gpus": args.gpus' not in src and src.count('"n_gpus": n_gpus') == 2 and "comm.size()" in src
