#!/usr/bin/env python3
"""bench.py -- bundle-adjustment throughput on MI355X (metric of BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Both forms run N ranks, one device each: started plain with N > 1 the script starts torch.distributed.run on itself (ensure_ranks), it refuses when
fewer than N devices are visible or when --gpus contradicts the launcher's WORLD_SIZE, and the line's `n_gpus` is the rank count RCCL itself reports for the
library's own communicator (rccl_rank_count).  Every number of the line -- headline, extra_workloads, cfg4, sharded.* -- carries `parity_ok` against the
oracle's stored result for its problem (tests/golden/oracle_final_costs.json).

A "step" is one full pass of the hot path over one batch of synthetic input: one complete
adjustBundle()-equivalent solve (ceres::Solve semantics, reference options BA.cpp:171-177 except the
10 s wall limit, which is disabled as BASELINE.md prescribes) of the device-resident problem, restarted
from the same initial parameters every step (device-to-device reset, inside the timed region).

Workload (config.workload): BASELINE.json configs[2] == BASELINE.md cfg 3, the configuration the metric
is quoted on: synthetic 200 cams / 100k pts / 1M obs, fp32 Jacobian blocks + fp64 accumulation,
seeded generator of sfm-toy-library_amd/synthetic.py.  With N > 1 every rank solves an independent
problem of that size (different seed): the reference has no exchange step between independent
reconstructions, so there is no data-path collective and scaling is "weak".
The same invocation then ALSO runs the path that does have an exchange step -- ONE problem (cfg 3, then the
1000-camera cfg 5) with its points sharded over the N ranks and the reduced camera system all-reduced over
RCCL / xGMI once per LM iteration (sfmba_problem_solve_sharded, strong scaling) -- and reports it in the
"sharded" object of the same JSON line (--sharded-extras 0 turns that off; at N = 1 it is off unless asked for).
BASELINE config 4 as written (the 200-camera problem as 8 independent sub-problems of 25 cams / 12.5k pts / 125k obs) is timed too: at N > 1 as
`cfg4_replicas` (rank g solves sub-problem g), at N = 1 as `extra_workloads.cfg4_8_concurrent` (the eight on eight streams of the one GPU).

value = LM iterations (successful + unsuccessful) of all ranks / max-over-ranks wall time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node (default: WORLD_SIZE when started by torch.distributed.run, else 1).  Started as a plain "
                         "`python bench.py --gpus N` with N > 1 the script starts its own N ranks (torch.distributed.run, one device each) and "
                         "refuses when fewer than N devices are visible")
    ap.add_argument("--launch-only", action="store_true",
                    help="dry run of the launcher: every rank prints its RANK / LOCAL_RANK / WORLD_SIZE as one JSON line and exits without touching a GPU "
                         "(tests/test_bench_launcher_cpu.py)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", help="name in synthetic.CONFIGS (default: the BASELINE metric config)")
    ap.add_argument("--linear", default="pcg", choices=["cholesky", "pcg", "auto"],
                    help="reduced-system solver: pcg = two-level block-Jacobi PCG on the dense reduced system, tolerance --pcg-tol (north_star; "
                         "the headline), auto = the library / shim default: the DENSE_SCHUR result through the CG at 1e-12 with the Cholesky as "
                         "fallback, cholesky = always factorise (the reference's literal configuration)")
    ap.add_argument("--precision", default="f32j", choices=["f32j", "f64"])
    ap.add_argument("--pcg-tol", type=float, default=1e-8, help="CG tolerance (library default 1e-8, anchored to the first LM iteration)")
    ap.add_argument("--mode", default="independent", choices=["independent", "sharded"],
                    help="N > 1: 'independent' = one whole problem per GPU (weak scaling, no data-path collective; the default the "
                         "driver runs); 'sharded' = ONE problem with its points sharded over the ranks and the reduced camera "
                         "system all-reduced over RCCL every LM iteration (strong scaling; BASELINE config 5 is quoted this way)")
    ap.add_argument("--sharded-extras", type=int, default=-1,
                    help="after the headline run also time ONE problem sharded over all ranks (cfg3 and cfg5) with the RCCL all-reduce "
                         "of the reduced camera system: 1 = yes, 0 = no, -1 (default) = only when N > 1")
    ap.add_argument("--distributed-cg", action="store_true", help="--mode sharded: the CG without the redundant solve (reduce-scatter + one small all-reduce per CG iteration)")
    ap.add_argument("--implicit-cg", action="store_true", help="--mode sharded: the CG with the product formed implicitly from every rank's own points (no exchange of the reduced matrix at all)")
    ap.add_argument("--row-sharded", action="store_true", help="--mode sharded: block ROWS of the reduced matrix per rank (every rank holds the whole problem; per-point table all-gathered; "
                                                              "multi-workgroup distributed CG on the owned rows)")
    ap.add_argument("--extra-workloads", type=int, default=-1,
                    help="after the timed region also run the other BASELINE configurations (cfg 2 in fp64 with the reference's solver, cfg 3 in all-fp64, cfg3_banded, cfg 5) and hold "
                         "each to the oracle's stored final cost: 1 = yes, 0 = no, -1 (default) = at N = 1 in the driver's plain invocation (default workload and solver, "
                         "CPU baseline not switched off: the tool runs under rocprofv3 stay lean)")
    ap.add_argument("--extras-timeout", type=int, default=420, help="seconds after which the sharded extras are abandoned (the headline line is printed regardless)")
    ap.add_argument("--opt", action="append", default=[], metavar="FIELD=VALUE",
                    help="set a field of sfmba_options for the headline solve (A/B runs: --opt pcg_symmetric=-1); may be repeated")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not spawn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic live; the committed "
                         "profiles/*_pmc_traffic.txt summary is used instead")
    ap.add_argument("--cpu-iters", type=int, default=0, help="LM iterations of the CPU sample (0 = auto)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="OpenMP threads of the CPU sample (0 = swept upwards from 16 while it still gets faster; "
                                                               "nproc itself is 650x SLOWER than 16 threads on the 256-thread box, see cpu_baseline())")
    return ap.parse_args()


def emit(obj):
    """One JSON line in ONE write: torch.distributed.run starts its ranks with `python -u`, where print() hands the text and the newline to the
    pipe separately -- two ranks' lines then come out glued together now and then."""
    sys.stdout.write(json.dumps(obj) + "\n")
    sys.stdout.flush()


def ensure_ranks(args):
    """The contract is `python bench.py --gpus N`; the N > 1 form normally arrives wrapped in torch.distributed.run, but a plain invocation
    must not silently run ONE rank and print n_gpus = 1.  Returns (rank, local_rank, world) of THIS process; when N > 1 ranks
    are asked for and none were started, this process starts `python -m torch.distributed.run --nproc-per-node N bench.py <same args>` as a child, waits for it and exits with its code."""
    started = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if started:
        world = int(os.environ["WORLD_SIZE"])
        if args.gpus is not None and args.gpus != world:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks: refusing to report a line for either" % (args.gpus, world))
        args.gpus = world
        return int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"])), world
    if args.gpus is None:
        args.gpus = 1
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus == 1:
        return 0, 0, 1
    if not args.launch_only:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit("bench.py: --gpus %d asked for, %d HIP device(s) visible: refusing (a run on fewer devices would not be an N-GPU number)"
                             % (args.gpus, have))
    import socket
    import subprocess
    # The ranks run as a child `python -m torch.distributed.run ...` whose stdio is this process's.  The rendezvous port is picked here and
    # bound by the child a moment later: if somebody else takes it in between the child dies within seconds -- a launch that fails that
    # early (before any rank could have printed a line) is repeated on a fresh port, twice at most.
    rc = 1
    for attempt in range(3):
        with socket.socket() as sk:                   # a free port on the loop-back interface for the rendezvous
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush(); sys.stderr.flush()
        t0 = time.time()
        child = subprocess.Popen(cmd)
        import signal
        old = {sg: signal.signal(sg, lambda signum, frame: child.send_signal(signum)) for sg in (signal.SIGTERM, signal.SIGINT)}      # (what ends this process ends the ranks)
        try:
            rc = child.wait()
        finally:
            for sg, h in old.items():
                signal.signal(sg, h)
        if rc == 0 or time.time() - t0 > 20.0:
            break
        sys.stderr.write("bench.py: the launcher exited with %d after %.1f s (attempt %d of 3)\n" % (rc, time.time() - t0, attempt + 1))
    raise SystemExit(rc)


def parity_vs_oracle(summ, n_obs, key, tol_rel_cost=1e-6):
    """A solve's summary against the ORACLE's stored result (tests/golden/oracle_final_costs.json): termination, LM iteration count, final cost
    (north_star's 1e-6 relative unless a tighter bar is passed) and RMS within 1e-4 px.  Returns a dict with `parity_ok` (None: nothing stored)."""
    global _ORACLE_WANT
    if _ORACLE_WANT is None:
        with open(os.path.join(ROOT, "tests", "golden", "oracle_final_costs.json")) as f:
            _ORACLE_WANT = json.load(f)
    w = _ORACLE_WANT.get(key)
    if w is None:
        return {"parity_ok": None, "oracle_key": key, "note": "no stored oracle result for this problem"}
    rms = float(np.sqrt(2.0 * summ["final_cost"] / n_obs))
    rel = abs(summ["final_cost"] - w["final_cost"]) / w["final_cost"]
    ok = (summ["termination_name"] == w["termination"] and summ["iterations"] == w["iterations"] and rel <= tol_rel_cost and abs(rms - w["final_rms_px"]) < 1e-4)
    return {"parity_ok": bool(ok), "oracle_key": key, "oracle_final_cost": w["final_cost"], "oracle_iterations": w["iterations"],
            "rel_cost_diff_vs_oracle": rel, "rms_diff_vs_oracle_px": rms - w["final_rms_px"], "parity_tolerance_rel_cost": tol_rel_cost}


_ORACLE_WANT = None


def algorithmic_bytes_per_iteration(n_obs, n_pt, n_cam, s_o, n_lin):
    """B_iter = 2 N_obs (8 + 2 s_o) + 72 N_pt + 96 N_cam + 8 d^2 (1 + n_lin)."""
    d = 6 * n_cam + 1
    return 2 * n_obs * (8 + 2 * s_o) + 72 * n_pt + 96 * n_cam + 8 * d * d * (1 + n_lin)


def cpu_baseline(prob_name, seed_sub, args, gpu_rms):
    """Host restatement of the reference CPU path (oracle = Ceres-equivalent LM + DENSE_SCHUR, NOT Ceres), timed on a bounded
    sample of the same workload: the same problem solved to the same termination, at nproc threads (BASELINE.md section 3:
    OMP_NUM_THREADS = nproc; `value`), at 16 threads (the r01 / r02 rows) and -- the reference's actual configuration,
    num_threads = 1 -- one LM iteration on one thread."""
    nproc = os.cpu_count() or 1
    try:
        nproc = min(nproc, len(os.sched_getaffinity(0)))      # (what this process may actually run on)
    except (AttributeError, OSError):
        pass
    import sfm_toy_library_amd as sfm
    from oracle import oracle_py as oracle           # checker/baseline only -- never part of the product path
    prob = sfm.make_problem(prob_name, sub=seed_sub)
    iters = args.cpu_iters or (4 if prob.n_obs >= 500000 else 50)
    opt = sfm.SfmbaOptions.defaults(max_seconds=0.0, max_iters=iters)

    def timed(nthreads, repeats, o=opt):
        oracle.set_num_threads(nthreads)
        best = None
        for _ in range(repeats):
            t0 = time.time()
            summ = oracle.solve(prob, o)[3]
            dt = time.time() - t0
            if best is None or summ["seconds"] < best[0]["seconds"]:
                best = (summ, dt)
        return best

    # BASELINE.md section 3 asks for OMP_NUM_THREADS = nproc.  Measured on the 256-thread GPU box (profiles/r03_a_cfg3_pcg_bench.json): the
    # restatement takes 347 s per cfg-3 solve at 256 threads against 0.53 s at 16 (its Schur elimination keeps a reduced matrix per
    # thread) -- a sample of that size does not belong in a bench that has to finish in minutes, and it would flatter the GPU by 650x.
    # So: the thread count is swept upwards from 16 on ONE LM iteration each and the sweep stops at the first count that is not at
    # least 10 % faster than the best so far; `value` is the full solve at the best count found (`cores`), --cpu-threads forces a count.
    one = sfm.SfmbaOptions.defaults(max_seconds=0.0, max_iters=1)
    sweep = {}
    if args.cpu_threads:
        threads = args.cpu_threads
    else:
        threads, best_t = min(16, nproc), None
        for cand in [c for c in (16, 32, 64, 128, 256, 512) if c <= nproc] or [nproc]:
            s1c = timed(cand, 1, one)[0]
            sweep[str(cand)] = round(s1c["seconds"], 4)
            if best_t is not None and s1c["seconds"] > 0.9 * best_t:
                break
            threads, best_t = cand, s1c["seconds"]
    summ, dt = timed(threads, 2)
    n_it = max(summ["iterations"], 1)
    rows = {"thread_sweep_first_iteration_seconds": sweep}
    if threads != 16 and nproc >= 16:
        s16, dt16 = timed(16, 2)
        rows["threads_16"] = {"value": max(s16["iterations"], 1) / s16["seconds"], "unit": "LM iterations/s", "cores": 16,
                              "sample": "%s, %d LM iterations, best of 2 (%.1f s)" % (prob_name, s16["iterations"], s16["seconds"])}
    # the reference's own configuration is num_threads = 1 (Ceres default, BA.cpp:171-177): one LM iteration of it
    oracle.set_num_threads(1)
    s1 = oracle.solve(prob, one)[3]
    oracle.set_num_threads(min(threads, 16))
    rows["single_thread"] = {"value": max(s1["iterations"], 1) / s1["seconds"], "unit": "LM iterations/s", "cores": 1,
                             "sample": "%s, first LM iteration, one thread (%.1f s)" % (prob_name, s1["seconds"])}
    out = {
        "value": n_it / summ["seconds"],
        "unit": "LM iterations/s",
        "cores": threads,
        "kind": "port",
        "sample": "%s, %d LM iterations of the same problem to %s, best of 2 (%.2f s of solve, %.2f s wall, cost %.6e -> %.6e); "
                  "host restatement of Ceres LM + DENSE_SCHUR with Jet autodiff (oracle/sfmba_oracle.c), not Ceres itself"
                  % (prob_name, n_it, summ["termination_name"], summ["seconds"], dt, summ["initial_cost"], summ["final_cost"]),
        "residuals_per_s": 2.0 * prob.n_obs * (summ["residual_evals"] + summ["jacobian_evals"]) / summ["seconds"],
        "seconds_per_iteration": summ["seconds"] / n_it,
        "rms_px_after_sample": float(np.sqrt(2 * summ["final_cost"] / prob.n_obs)),
        "host_cpu": _cpu_model(),
        "host_nproc": os.cpu_count(), "host_usable_cpus": nproc,
    }
    out.update(rows)
    return out


SOLVER_NAMES = {0: "DENSE_SCHUR-equivalent Cholesky", 1: "two-level block-Jacobi PCG", 2: "AUTO (CG to 1e-12, Cholesky fallback)"}
# what is summed in which precision in SFMBA_PRECISION_F32J (include/sfmba.h; csrc/ba_kernels.hip pair_block_reduce / k_point_build / cam_diag_finish)
DTYPE_F32J = ("f32 Jacobian blocks and observation coordinates; a lane's own partial sum (<= 8 pair products of a block, ceil(track / 4) point-block products of a "
              "point) in f32, every sum across lanes, chunks, points and ranks in f64; residuals, cost, reduced system and solve in f64")


def _flush_c_stdio():
    """librccl prints a version banner into C stdio's buffer when the first communicator is created; left there it comes out at process exit,
    BEHIND the JSON line.  Flushed here, the JSON line is the last line of stdout."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    args = parse()
    rank, local_rank, world = ensure_ranks(args)
    if args.launch_only:
        emit({"launch_only": True, "rank": rank, "local_rank": local_rank, "world_size": world, "gpus_arg": args.gpus,
              "master_addr": os.environ.get("MASTER_ADDR"), "pid": os.getpid()})
        return
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP back end has no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d wants device %d, %d visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    n_gpus, rccl_check = 1, None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        if args.mode == "sharded":
            n_gpus, rccl_check = rccl_rank_count(torch, dist, rank, local_rank, world)
        # (headline mode: the library's own communicator is counted AFTER the timed region, under the watchdog below -- a collective of
        # the N > 1 RCCL path that never returns must not cost the headline line; until then n_gpus is torch.distributed's world size)

    import sfm_toy_library_amd as sfm
    from sfm_toy_library_amd import capi

    precision = 1 if args.precision == "f32j" else 0
    linear = {"cholesky": 0, "pcg": 1, "auto": 2}[args.linear]
    if args.mode == "sharded":
        return main_sharded(args, rank, local_rank, world, torch, dist, sfm, capi, precision, linear, n_gpus)
    sub = rank if world > 1 else None
    globals()["PMC_WORKLOAD"] = args.workload
    prob = sfm.make_problem(args.workload, sub=sub)
    P = capi.Problem(prob, precision=precision, device=local_rank)
    opt = capi.default_options(max_seconds=0.0, precision=precision, linear_solver=linear, pcg_tolerance=args.pcg_tol)
    for kv in args.opt:
        k, _, v = kv.partition("=")
        setattr(opt, k, type(getattr(opt, k))(float(v)))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        P.reset()
        P.solve(opt)
    # ---- timed region: exactly K steps, barrier + synchronize on both sides, no instrumentation ----
    barrier()
    t0 = time.perf_counter()
    iters = res_evals = jac_evals = lin_iters = 0
    summ = None
    for _ in range(args.steps):
        P.reset()
        summ, _ = P.solve(opt)
        iters += summ["iterations"]
        res_evals += summ["residual_evals"]
        jac_evals += summ["jacobian_evals"]
        lin_iters += summ["linear_iters"]
    barrier()
    dt = time.perf_counter() - t0
    # ---- the same K steps again with every launch bracketed by HIP events on the solver's stream:
    # per-kernel average launch durations for the roofline object (not part of `value`) ----
    P.set_profiling(True)
    t1 = time.perf_counter()
    lin_iters_prof = 0
    for _ in range(args.steps):
        P.reset()
        sp, _ = P.solve(opt)
        lin_iters_prof += sp["linear_iters"]
    barrier()
    dt_prof = time.perf_counter() - t1
    profile = P.get_profile()
    P.set_profiling(False)

    # every rank's own result against the oracle's stored one for ITS problem (rank g of N > 1 solves make_problem(workload, sub=g))
    par = parity_vs_oracle(summ, prob.n_obs, args.workload if sub is None else "%s.%d" % (args.workload, sub))
    tot = torch.tensor([float(iters), float(res_evals + jac_evals), dt, 0.0 if par["parity_ok"] is False else 1.0], dtype=torch.float64, device="cuda")
    if dist is not None:
        tmax, tmin = tot[2:3].clone(), tot[3:4].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(tot[:2], op=dist.ReduceOp.SUM)
        tot[2], tot[3] = tmax[0], tmin[0]
    g_iters, g_evals, g_dt, g_par = [float(v) for v in tot.tolist()]
    if world > 1:
        n_gpus = dist.get_world_size()

    line = None
    if rank == 0:
        n_obs, n_pt, n_cam = prob.n_obs, prob.n_pt, prob.n_cam
        s_o = 4 if precision == 1 else 8
        rms = float(np.sqrt(2 * summ["final_cost"] / n_obs))
        line = {
            "metric": "BA LM iterations/sec (200 cams, 100k pts, 1M obs)" if args.workload == "cfg3" else "BA LM iterations/sec",
            "value": g_iters / g_dt,
            "unit": "LM iterations/s",
            "n_gpus": n_gpus,               # N > 1: ncclCommCount of the library's own communicator (rccl_rank_count), never the argument
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * g_dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if precision == 0 else DTYPE_F32J,
            "data": "synthetic",
            "config": {"workload": "%s: %d cams / %d pts / %d obs, shared focal, %s, one independent problem per GPU"
                                   % (args.workload, n_cam, n_pt, n_obs, SOLVER_NAMES[linear]),
                       "step": "one full LM solve to ceres CONVERGENCE from the resident initial point",
                       "lm_iterations_per_step": g_iters / (args.steps * world),
                       "linear_solver": ("two-level block-Jacobi PCG, tolerance %.0e anchored to the first LM iteration" % args.pcg_tol) if linear == 1 else
                                        ("exact Cholesky (DENSE_SCHUR equivalent)" if linear == 0 else
                                         "AUTO (library / shim default): the DENSE_SCHUR result through the two-level CG at relative 1e-12, Cholesky fallback")},
            "residuals_per_sec": 2.0 * n_obs * g_evals / g_dt,
            "ms_per_lm_iteration": 1e3 * g_dt * world / g_iters,
            "final_rms_px": rms,
            "final_cost": summ["final_cost"],
            "termination": summ["termination_name"],
            # every rank's result against the oracle's stored one for its own problem (None: nothing stored for this workload)
            "parity_ok": None if par["parity_ok"] is None else bool(g_par == 1.0), "parity_rank0": par,
        }
        # fraction of the HBM roofline of the whole LM iteration (algorithmic bytes of SURVEY 8d)
        n_lin = 1 if linear == 0 else max(1.0, lin_iters / max(iters, 1))
        b_iter = algorithmic_bytes_per_iteration(n_obs, n_pt, n_cam, s_o, n_lin)
        line["whole_iteration_hbm"] = {"algorithmic_bytes_per_iteration": b_iter,
                                       "achieved_GBps": b_iter * (g_iters / world) / g_dt / 1e9,
                                       "frac_of_8TBps": b_iter * (g_iters / world) / g_dt / 8.0e12}
        if world == 1 and not args.no_live_traffic:
            globals()["LIVE_PMC"] = live_pmc_passes(args)
        line["roofline"] = roofline_entry(profile, prob, precision)
        line["roofline_all_kernels"] = roofline_entry(profile, prob, precision, all_kernels=True)
        if linear != 0 and "pcg_iter" in profile and lin_iters_prof > 0:
            # the launch-per-iteration CG is enqueued in batches sized from the previous solve (+1): launches behind the converging one
            # test the done flag and return.  The event brackets see whole batches, so the surplus cannot be timed on its own here:
            # both counts, the plain average, and the time per REAL iteration with all of the surplus charged to it (an upper bound).
            pl = profile["pcg_iter"]
            cg = {"launches": pl["launches"], "cg_iterations_reported_by_the_solver": lin_iters_prof,
                  "surplus_early_exit_launches": max(0, pl["launches"] - lin_iters_prof),
                  "avg_us_per_launch": pl["avg_us"], "us_per_real_iteration_upper_bound": pl["total_us"] / lin_iters_prof,
                  "note": "rocprofv3 (profiles/*_kernel_stats.txt) times every launch on its own: the early-exit launches are the ~2 us tail of its histogram"}
            for ent in [line["roofline"]] + list(line["roofline_all_kernels"] or []):
                if ent and ent.get("kernel") == "pcg_iter":
                    ent["launch_accounting"] = cg
                    if ent.get("algorithmic_bytes_per_launch"):
                        b = ent["algorithmic_bytes_per_launch"]
                        ent["per_real_iteration"] = {"us_upper_bound": cg["us_per_real_iteration_upper_bound"],
                                                     "frac_of_8TBps": b / (cg["us_per_real_iteration_upper_bound"] * 1e-6) / 8.0e12,
                                                     "frac_of_L2_34.5TBps": b / (cg["us_per_real_iteration_upper_bound"] * 1e-6) / 34.5e12}
        line["ms_per_step_with_event_bracketing"] = 1e3 * dt_prof / args.steps
        line["kernel_profile_us"] = {k: round(v["avg_us"], 2) for k, v in profile.items()}
        line["kernel_profile_share"] = {k: round(v["total_us"] / max(1e-9, sum(x["total_us"] for x in profile.values())), 4)
                                        for k, v in profile.items()}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, sub, args, rms)
    if rank == 0 and linear == 1 and world == 1:
        # the library / shim default (AUTO) on the same resident problem, untimed extra: what a drop-in caller gets without opting in
        o2 = capi.default_options(max_seconds=0.0, precision=precision)
        for _ in range(2):
            P.reset(); P.solve(o2)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        it2 = li2 = fb2 = 0
        for _ in range(args.steps):
            P.reset()
            s2, _ = P.solve(o2)
            it2 += s2["iterations"]; li2 += s2["linear_iters"]; fb2 += s2["cholesky_fallbacks"]
        torch.cuda.synchronize()
        da = time.perf_counter() - ta
        line["default_solver_auto"] = {"value": it2 / da, "unit": "LM iterations/s", "ms_per_step": 1e3 * da / args.steps,
                                       "cg_iterations_per_step": li2 / args.steps, "cholesky_fallbacks": fb2,
                                       "final_cost": s2["final_cost"], "note": "same problem, sfmba_options_default (SFMBA_LINEAR_AUTO)"}
    if rank == 0 and world == 1 and not os.environ.get("SFMBA_BENCH_PMC_CHILD") and \
            (args.extra_workloads == 1 or (args.extra_workloads == -1 and args.workload == "cfg3" and linear == 1 and precision == 1 and not args.no_cpu_baseline)):
        try:
            line["extra_workloads"] = extra_workloads(args, torch, sfm, capi)
        except Exception as e:                                  # never lose the headline line to the extras
            line["extra_workloads"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and linear == 1 and world == 1 and args.pcg_tol < 1e-3:
        # Inexact Newton: the same solver with the CG stopped at 1e-3 relative (Ceres' own default for its iterative Schur solvers is
        # eta = 1e-1, the reference sets eta = 1e-2, BA.cpp:173) -- NOT the headline: the headline keeps the 1e-8 of rounds 1 / 2, which
        # reproduces the exact solve's PARAMETERS to 1e-8.  Reported because it is the time-to-solution optimum at the north_star's bar:
        # same number of LM iterations, final RMS within 1e-4 px (measured: 4e-8) of the exact solve.
        o3 = capi.default_options(max_seconds=0.0, linear_solver=1, precision=precision, pcg_tolerance=1e-3)
        for _ in range(3):
            P.reset(); P.solve(o3)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        it3 = li3 = 0
        for _ in range(args.steps):
            P.reset()
            s3, _ = P.solve(o3)
            it3 += s3["iterations"]; li3 += s3["linear_iters"]
        torch.cuda.synchronize()
        da = time.perf_counter() - ta
        rms3 = float(np.sqrt(2.0 * s3["final_cost"] / n_obs))
        line["inexact_newton_pcg_tol_1e-3"] = {"value": it3 / da, "unit": "LM iterations/s", "ms_per_step": 1e3 * da / args.steps,
                                               "lm_iterations_per_step": it3 / args.steps, "cg_iterations_per_step": li3 / args.steps,
                                               "final_rms_px": rms3, "final_rms_minus_headline_px": rms3 - line["final_rms_px"],
                                               "termination": s3["termination_name"],
                                               "note": "same resident problem and solver, CG tolerance 1e-3 instead of 1e-8; not the headline"}
    P.close()
    # Everything from here on is beside the headline and must never cost it: exceptions are caught; against a collective that never returns
    # (the N > 1 RCCL path has only ever run on one rank here) a watchdog prints the headline line as it stands and ends the process ...
    import signal
    import threading
    want_sharded = args.sharded_extras == 1 or (args.sharded_extras == -1 and world > 1)
    sh = {} if want_sharded else None
    stage = ["the RCCL rank count"]
    watchdog, old_term = None, None

    def bail(why):
        if rank == 0:
            line["abandoned"] = "%s during %s" % (why, stage[0])
            if sh:
                line["sharded"] = dict(sh)
            emit(line)
        os._exit(0)
    if world > 1 or want_sharded:
        watchdog = threading.Timer(args.extras_timeout, lambda: bail("not finished within %d s: abandoned" % args.extras_timeout))
        watchdog.daemon = True
        watchdog.start()
        # ... and against a PEER that dies in them: the launcher then sends the surviving ranks SIGTERM -- rank 0 prints the line first
        old_term = signal.signal(signal.SIGTERM, lambda signum, frame: bail("terminated (signal %d: a peer rank ended)" % signum))
    if world > 1:
        # n_gpus = what RCCL itself reports for the library's own communicator (a mismatch is refused: SystemExit, no line)
        try:
            n_gpus, rccl_check = rccl_rank_count(torch, dist, rank, local_rank, world)
        except Exception as e:                  # (n_gpus stays torch.distributed's world size, and the line says so)
            rccl_check = {"error": "%s: %s" % (type(e).__name__, e), "n_gpus_is": "torch.distributed world size"}
        if rank == 0:
            line["n_gpus"] = n_gpus
            line["rccl_check"] = rccl_check
    stage[0] = "cfg4_replicas"
    if world > 1 and args.workload == "cfg3":
        # BASELINE config 4 as written: one 25-camera sub-problem per GPU (all ranks take part; rank 0 reports)
        try:
            c4 = cfg4_replicas(args, rank, local_rank, world, torch, dist, sfm, capi)
        except Exception as e:
            c4 = {"error": "%s: %s" % (type(e).__name__, e)}
        if rank == 0:
            line["cfg4_replicas"] = c4
    # ---- the path with a real exchange step: one problem, points sharded over the ranks (all ranks take part) ----
    if want_sharded:
        # BASELINE config 5 first, and of its four forms the row-sharded one first: should the extras run out of their time,
        # what is lost is the least interesting; one synthetic problem per workload, shared by its forms
        for wl in (["cfg5", "cfg3"] if args.workload == "cfg3" else [args.workload]):
            wl_prob = sfm.make_problem(wl)
            for variant in ("row_sharded_cg", "replicated_cg", "distributed_cg", "implicit_schur_cg"):
                key = wl if variant == "replicated_cg" else wl + "_" + variant
                stage[0] = "the sharded extras (%s)" % key
                try:
                    sh[key] = sharded_run(wl, args, rank, local_rank, world, torch, dist, sfm, capi, precision, linear, steps=max(3, min(args.steps, 10)),
                                          distributed={"replicated_cg": 0, "distributed_cg": 1, "implicit_schur_cg": 2, "row_sharded_cg": 3}[variant], prob=wl_prob)
                except Exception as e:                       # never lose the headline line to the extras
                    sh[key] = {"error": "%s: %s" % (type(e).__name__, e)}
    if watchdog is not None:
        watchdog.cancel()
        signal.signal(signal.SIGTERM, old_term)
    if rank == 0:
        if sh is not None:
            line["sharded"] = sh
        _flush_c_stdio()
        emit(line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def rccl_rank_count(torch, dist, rank, local_rank, world):
    """`n_gpus` of the JSON line is what RCCL itself reports (ncclCommCount of the library's own communicator, sfmba_comm_size), not an argument;
    beside it: every rank sits on its own device, and one all-reduce through that communicator sums to the rank count."""
    import ctypes as C
    from sfm_toy_library_amd import sharded
    comm = sharded.RcclComm(dist, rank, world, device=local_rank)
    try:
        n, r = comm.size()
        ones = torch.ones(8, dtype=torch.float64, device="cuda")
        rc = comm.L.sfmba_comm_allreduce(comm._h, C.c_void_p(ones.data_ptr()), C.c_int64(8), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        summed = float(ones[0].item())
    finally:
        comm.close()
    try:
        ident = str(torch.cuda.get_device_properties(local_rank).uuid)
    except Exception:
        ident = "device %d" % local_rank
    ids = [None] * world
    dist.all_gather_object(ids, (os.uname().nodename, local_rank, ident))
    check = {"nccl_comm_count": n, "nccl_user_rank_matches": r == rank, "allreduce_of_ones": summed, "allreduce_rc": int(rc),
             "distinct_devices": len(set(ids)) == world, "devices": [list(i) for i in ids]}
    if n != world or r != rank or rc != 0 or summed != float(world) or len(set(ids)) != world:
        raise SystemExit("bench.py: RCCL reports %d ranks (rank %d) for WORLD_SIZE=%d rank %d, all-reduce of ones = %r, devices %r: refusing" % (n, r, world, rank, summed, ids))
    return n, check


def cfg4_replicas(args, rank, local_rank, world, torch, dist, sfm, capi):
    """BASELINE config 4 AS WRITTEN at N > 1: the 200-camera problem as independent sub-problems of 25 cams / 12.5k pts / 125k obs, rank g solves
    sub-problem g (mod 8) -- no data-path collective; every rank's result held to the oracle's stored one."""
    sub = rank % 8
    prob = sfm.make_problem("cfg4", sub=sub)
    opt = capi.default_options(max_seconds=0.0, precision=1, linear_solver=1, pcg_tolerance=args.pcg_tol)
    steps = max(args.steps, 10)
    with capi.Problem(prob, precision=1, device=local_rank) as P:
        for _ in range(3):
            P.reset(); P.solve(opt)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        iters = 0
        for _ in range(steps):
            P.reset()
            s, _ = P.solve(opt)
            iters += s["iterations"]
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    par = parity_vs_oracle(s, prob.n_obs, "cfg4.%d" % sub)
    t = torch.tensor([float(iters), dt, 1.0 if par["parity_ok"] else 0.0], dtype=torch.float64, device="cuda")
    tmax, tmin = t[1:2].clone(), t[2:3].clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX); dist.all_reduce(tmin, op=dist.ReduceOp.MIN); dist.all_reduce(t[:1], op=dist.ReduceOp.SUM)
    return {"workload": "cfg4: %d independent sub-problems of %d cams / %d pts / %d obs, one per GPU (rank g: sub-problem g mod 8)" % (world, prob.n_cam, prob.n_pt, prob.n_obs),
            "scaling": "weak", "ranks": world, "steps": steps, "value": float(t[0].item()) / float(tmax[0].item()), "unit": "LM iterations/s",
            "ms_per_step": 1e3 * float(tmax[0].item()) / steps, "dtype": DTYPE_F32J, "linear_solver": SOLVER_NAMES[1],
            "parity_ok": bool(tmin[0].item() == 1.0), "rank0": par}


def cfg4_concurrent(args, torch, sfm, capi, nprob=8):
    """BASELINE config 4 on ONE GPU: its eight sub-problems resident side by side, each on its own stream with its own host thread (the C ABI call
    releases the GIL), solved concurrently -- and one after the other for comparison; each held to the oracle's stored result."""
    import threading
    probs = [sfm.make_problem("cfg4", sub=g) for g in range(nprob)]
    opt = capi.default_options(max_seconds=0.0, precision=1, linear_solver=1, pcg_tolerance=args.pcg_tol)
    steps = 20
    P = [capi.Problem(p, precision=1, device=torch.cuda.current_device()) for p in probs]
    try:
        for h in P:
            for _ in range(2):
                h.reset(); h.solve(opt)
        last = [None] * nprob

        def work(k, out, gate=None):
            if gate is not None:
                gate.wait()
            its = 0
            for _ in range(steps):
                P[k].reset()
                s, _ = P[k].solve(opt)
                its += s["iterations"]
            out[k] = its
            last[k] = s
        torch.cuda.synchronize()
        t0 = time.perf_counter(); seq = [0] * nprob
        for k in range(nprob):
            work(k, seq)
        torch.cuda.synchronize()
        dt_seq = time.perf_counter() - t0
        con = [0] * nprob
        gate = threading.Barrier(nprob + 1)
        th = [threading.Thread(target=work, args=(k, con, gate)) for k in range(nprob)]
        for t in th:
            t.start()
        gate.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        dt_con = time.perf_counter() - t0
    finally:
        for h in P:
            h.close()
    pars = [parity_vs_oracle(last[k], probs[k].n_obs, "cfg4.%d" % k) for k in range(nprob)]
    out = {"workload": "cfg4: the %d independent sub-problems of %d cams / %d pts / %d obs, all resident on ONE GPU, one stream + one host thread each"
                       % (nprob, probs[0].n_cam, probs[0].n_pt, probs[0].n_obs),
           "dtype": DTYPE_F32J, "linear_solver": SOLVER_NAMES[1], "steps": steps, "value": sum(con) / dt_con, "unit": "LM iterations/s",
           "ms_per_round_of_%d_solves" % nprob: 1e3 * dt_con / steps,
           "one_after_the_other": {"value": sum(seq) / dt_seq, "unit": "LM iterations/s", "ms_per_solve": 1e3 * dt_seq / (steps * nprob)},
           "lm_iterations_per_step": sum(con) / (steps * nprob), "parity_ok": all(p["parity_ok"] for p in pars),
           "max_rel_cost_diff_vs_oracle": max(p["rel_cost_diff_vs_oracle"] for p in pars),
           "max_abs_rms_diff_vs_oracle_px": max(abs(p["rms_diff_vs_oracle_px"]) for p in pars)}
    assert out["parity_ok"], "cfg4_8_concurrent: a sub-problem's result differs from the oracle's stored one: %r" % pars
    return out


def extra_workloads(args, torch, sfm, capi):
    """The other BASELINE configurations under the same clock as the headline, untimed extras of the default run: each
    problem resident, 2 warm-up solves, K timed solves bracketed by synchronisations; LM iterations, final cost and RMS held to the ORACLE's
    stored result (tests/golden/oracle_final_costs.json, regenerated by tests/golden/make_oracle_final_costs.py) -- `parity_ok`."""
    with open(os.path.join(ROOT, "tests", "golden", "oracle_final_costs.json")) as f:
        want = json.load(f)
    runs = [
        # key, workload, precision, linear solver, options, timed solves
        ("cfg2_f64_dense_schur", "cfg2", 0, 0, {}, 20),                       # BASELINE config 2: fp64, the reference's solver (BA.cpp:172)
        ("cfg2_f64_default", "cfg2", 0, 2, {}, 20),
        ("cfg3_f64_pcg", "cfg3", 0, 1, {"pcg_tolerance": args.pcg_tol}, 10),  # the headline's problem and solver in the reference's own arithmetic
        ("cfg3_f64_default", "cfg3", 0, 2, {}, 10),
        ("cfg3_banded_pcg", "cfg3_banded", 1, 1, {"pcg_tolerance": args.pcg_tol}, 10),      # realistic co-visibility (cameras on a path)
        ("cfg3_banded_default", "cfg3_banded", 1, 2, {}, 10),
        ("cfg5_pcg", "cfg5", 1, 1, {"pcg_tolerance": args.pcg_tol}, 4),       # BASELINE config 5's problem on ONE GPU
    ]
    out = {}
    cache = {}
    for key, wl, prec, lin, okw, steps in runs:
        try:
            if wl not in cache:
                cache.clear()
                cache[wl] = sfm.make_problem(wl)
            prob = cache[wl]
            with capi.Problem(prob, precision=prec, device=torch.cuda.current_device()) as P:
                opt = capi.default_options(max_seconds=0.0, precision=prec, linear_solver=lin, **okw)
                for _ in range(2):
                    P.reset(); P.solve(opt)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                iters = lin_it = 0
                for _ in range(steps):
                    P.reset()
                    s, _ = P.solve(opt)
                    iters += s["iterations"]; lin_it += s["linear_iters"]
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            w = want[wl]
            rms = float(np.sqrt(2.0 * s["final_cost"] / prob.n_obs))
            rel = abs(s["final_cost"] - w["final_cost"]) / w["final_cost"]
            # the bar: fp64 with a factorised (or 1e-12) solve 1e-9 relative, anything inexact or F32J north_star's 1e-6 cost / 1e-4 px
            tol = 1e-9 if (prec == 0 and lin != 1) else 1e-6
            ok = (s["termination_name"] == w["termination"] and s["iterations"] == w["iterations"] and rel <= tol and abs(rms - w["final_rms_px"]) < 1e-4)
            out[key] = {"workload": "%s: %d cams / %d pts / %d obs" % (wl, prob.n_cam, prob.n_pt, prob.n_obs), "dtype": "f64" if prec == 0 else DTYPE_F32J,
                        "linear_solver": SOLVER_NAMES[lin], "steps": steps, "value": iters / dt, "unit": "LM iterations/s", "ms_per_step": 1e3 * dt / steps,
                        "lm_iterations_per_step": iters / steps, "cg_iterations_per_step": lin_it / steps, "termination": s["termination_name"],
                        "final_cost": s["final_cost"], "final_rms_px": rms, "oracle_final_cost": w["final_cost"], "oracle_iterations": w["iterations"],
                        "rel_cost_diff_vs_oracle": rel, "rms_diff_vs_oracle_px": rms - w["final_rms_px"], "parity_tolerance_rel_cost": tol, "parity_ok": bool(ok)}
            assert ok, "%s: result differs from the oracle's stored one: %r" % (key, out[key])
        except AssertionError as e:
            out.setdefault(key, {})["error"] = str(e)
        except Exception as e:
            out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
    cache.clear()
    try:                                       # BASELINE config 4 as written, on the one GPU there is
        out["cfg4_8_concurrent"] = cfg4_concurrent(args, torch, sfm, capi)
    except Exception as e:
        out["cfg4_8_concurrent"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def sharded_run(workload, args, rank, local_rank, world, torch, dist, sfm, capi, precision, linear, steps, distributed=False, prob=None):
    """ONE problem, points sharded over the ranks; the LM loop runs inside the C library (sfmba_problem_solve_sharded) with
    ncclAllReduce on the solver's stream (sharded.RcclComm); every rank times the same K solves.  Returns the result dict."""
    from sfm_toy_library_amd import sharded
    import ctypes as C
    if prob is None:
        prob = sfm.make_problem(workload)
    be = (sharded.HipRowShardBackend if int(distributed) == 3 else sharded.HipShardBackend)(prob, rank, world, device=local_rank, precision=precision)
    opt = capi.default_options(max_seconds=0.0, precision=precision, linear_solver=linear, pcg_tolerance=args.pcg_tol,
                               shard_distributed_cg=int(distributed))
    comm = sharded.RcclComm(dist, rank, world, device=local_rank)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None and world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    try:
        for _ in range(2):
            be.reset(); sharded.solve_sharded_native(be, opt, comm=comm)
        barrier()
        t0 = time.perf_counter()
        iters = 0
        summ = None
        for _ in range(steps):
            be.reset()
            summ = sharded.solve_sharded_native(be, opt, comm=comm)
            iters += summ["iterations"]
        barrier()
        dt = time.perf_counter() - t0
        # one rank: the same loop WITHOUT a communicator (the C loop then issues no collective call; pack / unpack kernels and the rest of
        # the sharded machinery stay) -- separates the loop's own overhead from what RCCL's one-rank all-reduce launches (two fills and a
        # copy per call in the r04_f kernel statistics)
        no_comm = None
        if world == 1:
            for _ in range(2):
                be.reset(); sharded.solve_sharded_native(be, opt, comm=None)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            it2 = 0
            for _ in range(steps):
                be.reset()
                it2 += sharded.solve_sharded_native(be, opt, comm=None)["iterations"]
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t2
            no_comm = {"value": it2 / dt2, "unit": "LM iterations/s", "ms_per_step": 1e3 * dt2 / steps,
                       "note": "one rank, no communicator call: pack / unpack kernels and loop bookkeeping only"}
        # the exchange step on its own, K back-to-back all-reduces of the large block.  CG path (csrc/ba_kernels.hip, k_shard_diag /
        # k_shard_offdiag): (A) diagonal blocks + vectors + scalars, (B) the off-diagonal blocks of the preconditioned matrix, (C) 80 scalars
        L = be.L
        nc = prob.n_cam
        ld = (int(L.sfmba_shard_setup_len(be._h)) - 80) // 2
        n_a, n_red = 27 * nc + 3 * ld + 80, 18 * nc * (nc - 1)
        ex_bytes, b_fp32 = summ.get("exchange_bytes", [8 * n_a, 8 * n_red, 640]), summ.get("exchange_b_fp32", False)
        stream = C.c_void_p(L.sfmba_problem_stream(be._h))
        buf = C.c_void_p(L.sfmba_shard_reduce_buf(be._h))
        reps = 10
        barrier()
        t1 = time.perf_counter()
        for _ in range(reps):          # the same call, count and element type as exchange (B) inside the solve
            (L.sfmba_comm_allreduce_f32 if b_fp32 else L.sfmba_comm_allreduce)(comm._h, buf, C.c_int64(n_red), stream)
        barrier()
        t_ar = (time.perf_counter() - t1) / reps
        # row-sharded: its two collectives on their own -- the all-gather of the per-point table (the one large message of a linearisation) and the
        # small all-reduce of a CG iteration (latency-bound: what DESIGN.md section 6 prices at ~20 us)
        t_ag = t_small = 0.0
        if summ.get("row_sharded"):
            stride = -(-prob.n_pt // world)
            nbytes = stride * (88 if precision == 1 else 128)
            scratch = torch.zeros(world * nbytes, dtype=torch.uint8, device="cuda")
            small = torch.zeros(ld + 16 * ((nc + 3) // 4 + 1), dtype=torch.float64, device="cuda")
            barrier()
            t1 = time.perf_counter()
            for _ in range(reps):
                L.sfmba_comm_allgather(comm._h, C.c_void_p(scratch.data_ptr()), C.c_int64(nbytes), stream)
            barrier()
            t_ag = (time.perf_counter() - t1) / reps
            t1 = time.perf_counter()
            for _ in range(50):
                L.sfmba_comm_allreduce(comm._h, C.c_void_p(small.data_ptr()), C.c_int64(small.numel()), stream)
            barrier()
            t_small = (time.perf_counter() - t1) / 50
        # the result of EVERY rank against the oracle's stored solve of the whole problem (the replicas of a sharded solve must all agree with it)
        g_par = parity_vs_oracle(summ, prob.n_obs, workload)
        tmax = torch.tensor([dt, t_ar, t_ag, t_small, 1.0 if g_par["parity_ok"] is False else 0.0], dtype=torch.float64, device="cuda")
        if dist is not None and world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        g_dt, g_ar, g_ag, g_small, bad = [float(v) for v in tmax.tolist()]
        if g_par["parity_ok"] is not None:
            g_par["parity_ok"] = bool(bad == 0.0)
        dist_note = None
        if summ.get("row_sharded"):
            dist_note = ("block ROWS of the preconditioned reduced matrix per rank: every rank holds the whole problem, eliminates its own range of points, the per-point table "
                         "is all-gathered (%d bytes received per rank and linearisation), the pair pass forms the rank's own block rows from all their pairs (no partial block "
                         "crosses a rank), per CG iteration ONE all-reduce of %d + 16 (cameras / 4 + 1) doubles (partial product + its partial dot products); "
                         "vector updates multi-workgroup, replicated" % (ex_bytes[1], ld))
        elif summ.get("implicit_schur_cg"):
            dist_note = ("no exchange of the reduced matrix: per CG iteration every rank applies its own points' W V^-1 W^T to the vector (two passes over "
                         "its observations) and ONE all-reduce of %d doubles sums the partial products; vector updates replicated" % ld)
        elif summ.get("distributed_cg"):
            dist_note = ("reduce-scatter of the upper-triangle blocks of the preconditioned matrix into ranges of block rows (%d bytes in the buffer, a rank "
                         "receives 1 / %d of it), then per CG iteration ONE all-reduce of %d doubles (the partial product from the owned blocks); "
                         "vector updates replicated" % (ex_bytes[1], world, ld))
        return {"workload": "%s: %d cams / %d pts / %d obs, ONE problem, %s sharded over %d rank(s)" % (workload, prob.n_cam, prob.n_pt, prob.n_obs, "points and block rows" if summ.get("row_sharded") else "points", world),
                "reduced_system_solve": ("row-sharded: distributed CG on the rank's own block rows, formed from all their pairs" if summ.get("row_sharded") else
                                         "implicit Schur CG (no reduced matrix formed or exchanged)" if summ.get("implicit_schur_cg") else
                                         "distributed CG (no redundant solve)" if summ.get("distributed_cg") else "every rank runs the CG on the summed matrix (redundant)"),
                "distributed_cg_exchange": dist_note,
                "scaling": "strong", "ranks": world, "steps": steps, "value": iters / g_dt, "unit": "LM iterations/s",
                "ms_per_step": 1e3 * g_dt / steps, "lm_iterations_per_step": iters / steps,
                "one_rank_without_collective": no_comm,
                "allreduce_bytes_per_lm_iteration": int(sum(ex_bytes)),
                # one rank: the "all-reduce" is a no-op of the communicator, its time says nothing about xGMI
                "allreduce_ms": 1e3 * g_ar if world > 1 else None,
                "allreduce_GBps_algorithmic": ex_bytes[1] / g_ar / 1e9 if world > 1 else None, "exchange_b_dtype": "f32" if b_fp32 else "f64",
                # (row-sharded only; one rank: no-ops of the communicator)
                "point_table_allgather_ms": (1e3 * g_ag if world > 1 else None) if summ.get("row_sharded") else None,
                "cg_iteration_allreduce_us": (1e6 * g_small if world > 1 else None) if summ.get("row_sharded") else None,
                "collective": ("per linearisation on the solver stream: two ncclAllGather (the halves of the per-point table: %d bytes per rank), one ncclAllReduce(SUM) of [6x6 diagonal "
                               "blocks | camera-focal column | rhs | diagonals | scalars] (%d doubles), one of 8 ld + 64 (cameras / 4 + 1) doubles at the CG's set-up and one of "
                               "ld + 16 (cameras / 4 + 1) = %d doubles per CG iteration, one of 80 trial-step scalars; one ncclAllGather of the final points per solve; `allreduce_ms` "
                               "times an all-reduce of the size of exchange (B) of the point-sharded forms for comparison -- this form has no such exchange"
                               % (-(-prob.n_pt // world) * (88 if precision == 1 else 128), n_a, ld + 16 * ((nc + 3) // 4 + 1))) if summ.get("row_sharded") else
                              ("three ncclAllReduce(SUM) per LM iteration on the solver stream: [6x6 diagonal blocks | camera-focal column | rhs | "
                               "diagonals | scalars] (%d doubles), the off-diagonal blocks of the block-Jacobi-preconditioned reduced matrix (%d values, %s: the "
                               "one timed here), 80 trial-step scalars; every rank runs the CG on the summed matrix redundantly" % (n_a, n_red, "fp32 like the CG's stored matrix" if b_fp32 else "fp64")),
                "final_rms_px": float(np.sqrt(2 * summ["final_cost"] / prob.n_obs)), "final_cost": summ["final_cost"],
                "termination": summ["termination_name"], "linear_iters_per_step": summ["linear_iters"],
                "lm_iterations_last_step": summ["iterations"], **g_par}
    finally:
        comm.close()
        be.close()


def main_sharded(args, rank, local_rank, world, torch, dist, sfm, capi, precision, linear, n_gpus):
    """--mode sharded: the sharded run IS the headline line (strong scaling)."""
    r = sharded_run(args.workload, args, rank, local_rank, world, torch, dist, sfm, capi, precision, linear, steps=args.steps,
                    distributed=3 if args.row_sharded else 2 if args.implicit_cg else 1 if args.distributed_cg else 0)
    if rank == 0:
        _flush_c_stdio()
        emit({
            "metric": "BA LM iterations/sec", "value": r["value"], "unit": "LM iterations/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64" if precision == 0 else DTYPE_F32J,
            "data": "synthetic", "config": {"workload": r["workload"], "step": "one full LM solve to ceres CONVERGENCE",
                                            "lm_iterations_per_step": r["lm_iterations_per_step"], "collective": r["collective"]},
            "sharded": r, "final_rms_px": r["final_rms_px"], "final_cost": r["final_cost"], "termination": r["termination"],
            "parity_ok": r.get("parity_ok")})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


PMC_KERNEL_NAMES = {"pcg_iter": "k_pcg_iter_fast", "schur_pairs": "k_schur_pairs", "cam_diag": "k_cam_diag",
                    "point_build": "k_point_build", "point_update": "k_point_update", "chol_panel": "k_chol_step",
                    "chol_update": "k_chol_update"}


PMC_WORKLOAD = "cfg3"  # set from --workload in main(): which committed summary the static fallback may read
LIVE_PMC = None      # {kernel name as rocprofv3 prints it: (FETCH_SIZE KB, WRITE_SIZE KB) per launch}, filled by live_pmc_passes()


def live_pmc_passes(args):
    """roofline.traffic measured by THIS run: two child runs of this script (2 steps, 1 warm-up, same workload and solver) under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` -- separate passes, counters only, as
    MI355X_MICROARCH.md's HBM section prescribes.  Returns None (and the committed summary is used) when rocprofv3 is missing,
    a pass fails or takes longer than 150 s."""
    import shutil
    import signal
    import subprocess
    import tempfile
    if os.environ.get("SFMBA_BENCH_PMC_CHILD") or shutil.which("rocprofv3") is None:
        return None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):      # this run is itself being profiled: no nested profiler
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_summary import per_kernel
    out = {}
    env = dict(os.environ, SFMBA_BENCH_PMC_CHILD="1", TMPDIR="/tmp")
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory(prefix="sfmba_pmc_", dir="/tmp") as tmp:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-live-traffic",
                   "--workload", args.workload, "--linear", args.linear, "--precision", args.precision, "--pcg-tol", repr(args.pcg_tol)]
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = proc.wait(timeout=150)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)      # the process group started here, nothing else
                proc.wait()
                return None
            if rc != 0:
                return None
            vals = per_kernel(d, counter)
            if not vals:
                return None
            out[counter] = vals
    names = set(out["FETCH_SIZE"]) | set(out["WRITE_SIZE"])
    return {"kernels": {k: (out["FETCH_SIZE"].get(k, 0.0), out["WRITE_SIZE"].get(k, 0.0)) for k in names},
            "seconds": time.perf_counter() - t0}


def pmc_traffic(kernel):
    """HBM-side bytes per launch (FETCH_SIZE + WRITE_SIZE): from the two rocprofv3 --pmc passes this run spawned
    (live_pmc_passes), else from the committed summary of the same command under profiles/.  FETCH_SIZE is doubled for the
    streaming kernels as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950."""
    import glob
    want = PMC_KERNEL_NAMES.get(kernel)
    if not want:
        return None
    wide = kernel in ("pcg_iter",)     # 16-byte-per-lane coalesced streams (the point passes stream 4- and 8-byte items since round 4: no correction)

    def entry(fetch_kb, write_kb, source, live):
        return {"bytes": (fetch_kb * (2.0 if wide else 1.0) + write_kb) * 1024.0, "fetch_kb": fetch_kb, "write_kb": write_kb,
                "fetch_x2_correction": wide, "source": source, "live": live}
    if LIVE_PMC is not None:
        hits = [v for k, v in LIVE_PMC["kernels"].items() if want in k]
        if hits:
            # several instantiations of one kernel (k_schur_pairs<..., MODE, ...>): the one that moves the most is the per-iteration pass
            f, w = max(hits, key=lambda v: v[0] + v[1])
            return entry(f, w, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes spawned by this run (%.0f s)" % LIVE_PMC["seconds"], True)
    # the committed summary must be of the SAME workload (r04_f_cfg3_pcg_pmc_traffic.txt / r04_f_cfg5_...): anything else is not this kernel's traffic
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_pcg_pmc_traffic.txt" % PMC_WORKLOAD)))
    if not files:
        return None
    for line in open(files[-1]):
        if want in line and not line.startswith("#"):
            parts = line.split()
            try:
                return entry(float(parts[-2]), float(parts[-1]), os.path.basename(files[-1]), False)
            except ValueError:
                return None
    return None


LIMITERS = {
    "pcg_iter": "L2 / Infinity-Cache latency + the dependent-launch boundary (one launch per CG iteration; the 11.5 MB matrix never leaves the "
                "256 MiB Infinity Cache, so FETCH_SIZE counts cache hits): not an HBM-bandwidth-bound kernel",
    "schur_pairs": "VALU issue: every wave instruction costs the SIMD ~4 cycles whatever it is (rocprofv3: SQ_ACTIVE_INST_VALU = SQ_INSTS_VALU quad-cycles; "
                   "88 % of the SIMD time in the unfactored form), so the pass is priced by its instruction COUNT: the factored form (camera factor "
                   "diag(R K', I) applied once per block) runs 295 instead of 363 instructions per 64 pairs and the launch went from 68 to 59 us; what "
                   "is left is 2 re-evaluations + one 6x6 product per pair and ~170 instructions of reduction / transform per block",
    "cam_diag": "VALU issue (11.5 M wave instructions per launch = 77 % of the SIMD time at one quad-cycle each): ~740 instructions per wave of 64 observations, "
                "a third of them the fp64 reduction tree of the 47 sums; index -> point-table gather in front of them",
    "point_build": "VALU issue at ~60 % + wave lifetime: two dependent memory levels and a 7-lanes-of-64 per-point phase per wave; since round 4 it "
                   "writes nothing per observation and gathers three component quads of the camera table instead of six (48 -> 35 us)",
    "point_update": "texture addresser (TA_TA_BUSY ~90 % of the launch, VALU 26 %): every lane gathers another camera's rows, each 16-byte load is an instruction over 64 "
                    "distinct lines; F32J gathers one 80-byte fp32 record per camera in the first sweep (5 loads instead of 10: 27.7 -> 21 us at cfg 3, 181 -> 135 at cfg 5), "
                    "the trial sweep its 6 loads of the fp64 trial pose",
    "chol_panel": "the serial chain of 64 pivots in the diagonal tile (one workgroup: ~250 cycles per pivot) + one launch boundary per block column",
    "chol_update": "fp64 MFMA, short launches",
}


def roofline_one(name, profile, model, overhead_us):
    avg_us = profile[name]["avg_us"]
    m = model.get(name)
    tr = pmc_traffic(name)
    base = {"kernel": name, "avg_launch_us": avg_us, "empty_event_bracket_us": overhead_us,
            "launches": profile[name]["launches"], "traffic": None if tr is None else tr["bytes"], "traffic_detail": tr,
            "traffic_is_static": not (tr or {}).get("live", False),   # False: measured by this run's own rocprofv3 --pmc passes; True: committed summary
            "limiter": LIMITERS.get(name),
            "timing": ("HIP events on the solver stream around batches of back-to-back launches (per-launch average includes the "
                       "~1.5-2.6 us launch boundary and the early-exit surplus launches of a batch)" if name == "pcg_iter" else
                       "HIP events on the solver stream around every launch; an event pair around nothing reads empty_event_bracket_us, "
                       "so launches shorter than ~10 us are overstated -- compare with the rocprofv3 average in profiles/")}
    if m is None:
        base.update({"bound": "hbm", "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None})
        return base
    if m.get("overhead_only"):
        # a pass a fused design would not have: no roofline fraction (its algorithmic bytes are a few KB), reported as pure overhead
        base.update({"overhead_only": True, "bound": "overhead", "achieved": None, "peak": None, "unit": None, "frac": None,
                     "design_bytes_per_launch": m["moved"], "note": m["note"]})
        return base
    if m["bound"] == "hbm":
        ach = m["bytes"] / (avg_us * 1e-6) / 1e9
        # `bound` names what limits the kernel; kernels whose bytes never leave the caches are priced against the same 8 TB/s (the
        # only byte peak the guide states) but labelled for what they are
        base.update({"bound": m.get("label", "hbm"), "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                     "algorithmic_bytes_per_launch": m["bytes"], "note": m["note"]})
        if m.get("cache_resident"):
            # the matrix of this kernel never leaves the caches (11.5 MB at cfg 3): 8 TB/s is not its roof.  The same bytes against the
            # guide's L2 row (MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate), labelled as such
            base.update({"cache_side": {"peak_GBps": 34500.0, "frac": ach / 34500.0,
                                        "label": "same algorithmic bytes / launch time against the aggregate L2 bandwidth of MI355X_MICROARCH.md (34.5 TB/s): "
                                                 "the matrix is Infinity-Cache / L2 resident, HBM is not this kernel's roof"}})
        if "moved" in m:     # bytes this design moves by construction (index lists, per-point tables): NOT algorithmic
            base.update({"design_bytes_per_launch": m["moved"], "overhead_ratio": m["moved"] / max(m["bytes"], 1.0),
                         "frac_on_design_bytes": m["moved"] / (avg_us * 1e-6) / 8.0e12})
    else:
        ach = m["flops"] / (avg_us * 1e-6) / 1e12
        base.update({"bound": "mfma", "achieved": ach, "peak": m["peak_tflops"], "unit": "TFLOP/s", "frac": ach / m["peak_tflops"],
                     "algorithmic_flops_per_launch": m["flops"], "note": m["note"]})
    return base


def roofline_entry(profile, prob, precision, all_kernels=False):
    """Roofline of the dominant kernel (largest share of the timed region).  Launch durations are measured live with
    HIP events recorded on the solver's own stream (the stream the kernels run on)."""
    if not profile:
        return None
    t = 4 if precision == 1 else 8
    model = kernel_models(prob.n_obs, prob.n_pt, prob.n_cam, 6 * prob.n_cam + 1, t)
    overhead = profile.get("empty_bracket", {}).get("avg_us", 0.0)
    names = [k for k in profile if k != "empty_bracket"]
    if all_kernels:
        out = [roofline_one(k, profile, model, overhead) for k in sorted(names, key=lambda k: -profile[k]["total_us"]) if k in model]
        # the linearisation stage as a whole against SURVEY 8(d)'s FUSED model (no stored Jacobian): what the four kernels together
        # would have to move, what they do move by design, and the measured counter traffic
        stage = [k for k in ("point_build", "cam_diag", "finalize", "schur_pairs") if k in profile]
        if len(stage) >= 3:
            launches = max(profile["point_build"]["launches"], 1)
            us = sum(profile[k]["total_us"] for k in stage) / launches
            alg = prob.n_obs * (8 + 2 * t) + 24 * prob.n_pt + 48 * prob.n_cam + model["schur_pairs"]["bytes"]
            moved = sum(model[k].get("moved", model[k].get("bytes", 0)) for k in stage if k in model)
            tr = {k: pmc_traffic(k) for k in stage}
            covered = [k for k in stage if tr[k] is not None]
            counter = sum(tr[k]["bytes"] for k in covered) if covered else None
            out.append({"kernel": "linearisation stage (point_build + cam_diag + finalize + schur_pairs)", "us_per_lm_iteration": us,
                        "bound": "hbm", "algorithmic_bytes": alg, "achieved": alg / (us * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": alg / (us * 1e-6) / 8.0e12, "design_bytes": moved, "overhead_ratio": moved / alg,
                        "counter_traffic_bytes": counter, "counter_traffic_kernels": covered,
                        "counter_overhead_ratio": None if counter is None else counter / alg,
                        "note": "algorithmic = SURVEY 8(d) fused model: observations once, points once, cameras once, S written once; "
                                "design bytes add the index lists, the per-point table and the pair-point list (nothing is materialised per observation)"})
        return out
    name = max(names, key=lambda k: profile[k]["total_us"])
    return roofline_one(name, profile, model, overhead)


def kernel_models(n_obs, n_pt, n_cam, d, t):
    """Per launch: ALGORITHMIC bytes after SURVEY 8(d) (fused model, no stored Jacobian) -- what `roofline.frac` is priced on --
    and, where this design materialises intermediate data, the bytes it moves by construction (`moved`).  Derivations in DESIGN.md."""
    yrec = 16 * t
    npair = n_obs * (n_obs / max(n_pt, 1) - 1) / 2      # pairs of observations of one point, a < b
    nb = 64
    nblk = (d + 1 + nb - 1) // nb
    b_res = n_obs * (8 + 2 * t) + 24 * n_pt + 48 * n_cam          # one residual evaluation (B_res)
    # the reduced matrix as this path stores it: both triangles in fp64 up to d = 1280 (the register-resident CG); above, ONE triangle
    # (symmetric streaming CG, round 6), in fp32 when the Jacobians are (F32J)
    sym = d > 1280
    b_mat = (t * d * (d + 1) // 2) if sym else 8 * d * d
    mat_txt = ("one triangle of the reduced matrix in %s (%d d (d + 1) / 2)" % ("fp32" if t == 4 else "fp64", t)) if sym else "the reduced matrix (8 d^2)"
    pa = 64 if t == 4 else 80          # PtRecA / PtRecB of the per-point table (sfmba_device.h)
    pb = 24 if t == 4 else 48
    return {
        "point_build": {"bound": "hbm", "bytes": b_res,
                        "moved": n_obs * (4 + 4 + 2 * t) + n_pt * (24 + 24 + 24 + 48 + pa + pb),
                        "note": "algorithmic: one pass over observations, points and cameras (B_res); moved: the two observation indices and the coordinates, "
                                "per point the scales, t, M and the table entry every other pass re-evaluates from (64 + 24 bytes); nothing is written per "
                                "observation (rounds 1 - 3 wrote a 64-byte record each)"},
        "schur_pairs": {"bound": "hbm", "bytes": b_mat,
                        "moved": 4 * npair + n_pt * pa + b_mat,
                        "note": "algorithmic: " + mat_txt + " written once; moved: + the pair-point list (4 bytes per pair) and the point table once "
                                "(it stays in L2: every pair re-reads its entry from there); no per-observation record is gathered"},
        "cam_diag": {"overhead_only": True, "bytes": 96 * n_cam + 8 * d,
                     "moved": n_obs * (4 + 2 * t) + n_pt * (pa + pb),
                     "note": "a fused design has no such pass (its result is 96 bytes per camera): everything this kernel moves and every microsecond it "
                             "takes is overhead of forming the camera-diagonal blocks in a pass of their own; moved: the camera-major point index and "
                             "observation coordinates (coalesced) and the point table once (L2-resident gathers)"},
        "point_update": {"bound": "hbm", "bytes": b_res + 24 * n_pt,
                         "moved": n_obs * (4 + 4 + 2 * t) + n_pt * (pa + 24 + 48 + 24 + 24),
                         "note": "one residual evaluation (B_res) + the trial points written; moved: observation indices and coordinates, per point the table "
                                 "entry, t, M, the point and the trial point (no per-observation record since round 4)"},
        "pcg_iter": ({"bound": "hbm", "bytes": b_mat + 9 * 8 * d,
                      "note": "one CG iteration = two launches (k_sy_vec + k_sy_prod): " + mat_txt + " read once, every entry used twice, + the x/r/p/q "
                              "vectors; the event bracket spans whole batches INCLUDING their early-exit launches: compare profiles/r06_ab_sy_prod_bisect.txt "
                              "(17.9 + 7.5 us per real iteration at d = 6001)"} if sym else
                     {"bound": "hbm", "label": "l2_mall_latency", "bytes": 8 * d * d + 9 * 8 * d, "cache_resident": 8 * d * d <= 64 << 20,
                      "note": "one CG iteration = one launch: reads its rows of the preconditioned reduced matrix S~ once "
                              "(8 d^2 bytes) + the x/r/p/q vectors; launch/latency bound (d = %d): the matrix is re-read from "
                              "L2/MALL every launch because L2 does not survive the kernel boundary" % d}),
        "chol_update": {"bound": "mfma", "flops": d * d * d / 3.0 / max(nblk - 1, 1), "peak_tflops": 78.6,
                        "note": "fp64 trailing update on v_mfma_f64_16x16x4_f64; d^3/3 flops of the factorisation spread over its launches; "
                                "peak = fp64 matrix 78.6 TF (171 tiles of 64^3 at most: latency bound, not MFMA bound)"},
        "chol_panel": {"bound": "mfma", "flops": d * d * d / 3.0 / max(nblk, 1), "peak_tflops": 78.6,
                       "note": "k_chol_step, one launch per block column of 64: panel solve as GEMMs with the inverse of the diagonal factor, trailing "
                               "update and the in-LDS factorisation of the next diagonal tile; d^3/3 flops of the factorisation spread over its launches; "
                               "peak = fp64 matrix 78.6 TF (latency bound on the pivot chain, not MFMA bound)"},
    }


if __name__ == "__main__":
    main()
