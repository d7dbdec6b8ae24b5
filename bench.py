#!/usr/bin/env python
"""bench.py — LM iterations/s of the full-BA hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config C4]

A "step" is one LM iteration (linearise -> Schur -> dense reduced solve ->
back-substitution/update -> trial cost -> trust-region control; reference
core/full_bundle_adjustment_solver.cpp:709-1007) of BASELINE.json's flagship
workload C4 (stereo, 1 000 poses / 500 000 landmarks / 5 000 000
observations, synthetic, seeded).  N > 1: one process per GPU, landmarks
sharded across ranks, one RCCL all-reduce of the packed reduced camera system
and one of four LM scalars per iteration, issued from C++ inside the library
(ba_rccl_allreduce_hook) — strong scaling by default (the problem is fixed),
`--weak` multiplies landmarks and observations by N at fixed poses.  Started
either by torchrun (RANK / WORLD_SIZE in the environment) or as plain
`python bench.py --gpus N`: the script then starts the N ranks itself, as
child processes, before anything touches the GPU.  Inputs are resident in HBM
before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_MFMA_PEAK_TFLOPS = 78.6  # public MI355X datasheet (not in local guides)
FP64_MFMA_MEASURED_TFLOPS = 47.9  # v_mfma_f64_16x16x4 issue peak measured on the part (tools/mfma_f64_peak.hip)
FP64_FMA_MEASURED_TFLOPS = 65.0   # v_fma_f64 issue peak measured on the part (same tool)

WORKLOADS = {
    "C1": "C1 test_ba.cpp stereo scene: 60 poses / 660 landmarks / 34019 obs",
    "C2": "C2 mono 6-DoF: 200 poses / 50k landmarks / 500k obs",
    "C3": "C3 stereo 6-DoF: 500 poses / 200k landmarks / 2M obs",
    "C4": "C4 stereo 6-DoF: 1000 poses / 500k landmarks / 5M obs",
    # off the headline's happy path (not BASELINE configs; measured because the
    # headline scene avoids these code paths)
    "C4R": "off-path C4R stereo: C4 with 15 % of the observations dropped at random "
           "(1000 poses / 500k landmarks / ~4.3M obs): irregular observation patterns",
    "W20": "off-path W20 mono: 500 poses / 100k landmarks / 2M obs, 20-pose windows "
           "(k_schur_partial: global triple list)",
    "DENSE1K": "off-path DENSE1K mono: 1000 poses / 60k landmarks / 480k obs, 8 random "
               "views per landmark (fully dense reduced system)",
}


def algorithmic_bytes(O, P, M, N):
    """bytes per LM iteration by stage (fp64)."""
    n6 = 6 * N
    build = 32 * O + 144 * P + 96 * M + 312 * N
    schur = 144 * P + 168 * M + 264 * N + 8 * n6 * n6
    solve = 16 * n6 * n6 + 144 * P + 192 * M + 144 * N
    control = 32 * O + 120 * M + 384 * N
    return dict(build=build, schur=schur, solve=solve, control=control,
                total=build + schur + solve + control)


PMC_SUMMARY = {"C4": "profiles/r03_c4_pmc_fetch_write_v2.txt"}


def pmc_traffic(kernel, config):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (one --pmc FETCH_SIZE pass, one --pmc WRITE_SIZE pass over this bench, see
    profiles/): counters are KiB per dispatch; on gfx950 FETCH_SIZE reports half
    of a wide coalesced read (MI355X_MICROARCH.md, HBM section), so traffic =
    (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  bench.py cannot collect PMC counters
    inside its own timed run; None when no summary is committed for the config."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        PMC_SUMMARY.get(config, ""))
    if not os.path.isfile(path):
        return None, None
    vals = {}
    for line in open(path):
        f = line.split()
        # "k_lin_grp<true, false> FETCH_SIZE 5 45628.2": the counter name follows the
        # (possibly templated) kernel name, the mean per dispatch is the last field
        if len(f) >= 4 and f[0].split("<")[0] == kernel:
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                if c in f:
                    vals[c] = float(f[-1])
    if len(vals) != 2:
        return None, None
    return ((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
            os.path.relpath(path, os.path.dirname(os.path.abspath(__file__))))


DENSE_GRID_GB = {"C1": 0.021, "C2": 5.6, "C3": 57.0, "C4": 287.0}


def dense_faithful_baseline(O, scenes, config, pr, huber):
    """One LM iteration of the oracle with the reference's dense N x M block
    grids allocated and re-zeroed (ba_oracle_set_dense_faithful).  C3 / C4: the
    grids need 57 / 287 GB -> reported as not runnable, and the C2 figure is
    measured instead so that the default (C4) line still carries one."""
    out = {}
    runnable = config if config in ("C1", "C2") else "C2"
    if runnable != config:
        out["note"] = ("%s: the reference layout needs %.0f GB of dense N x M "
                       "block grids - not runnable; measured at C2 instead"
                       % (config, DENSE_GRID_GB[config]))
    if pr is None or runnable != config:
        pr = scenes.scaled_problem(scenes.config_scene(runnable))
    o = O.Oracle(pr)
    o.set_dense_faithful(True)
    t = time.perf_counter()
    o.solve(O.make_options(max_iter=1, thr_step=-1.0, thr_cost=-1.0, huber=huber))
    dt = time.perf_counter() - t
    out.update({"config": runnable, "value": 1.0 / dt, "unit": "it/s", "cores": 1,
                "kind": "port", "dense_grid_GB": DENSE_GRID_GB[runnable],
                "sample": "1 LM iteration at %s with the four dense N x M grids "
                          "allocated and re-zeroed, %.2f s; stage ms build/schur/"
                          "solve/control = %s" % (runnable, dt, ", ".join(
                              "%.0f" % v for v in o.stage_ms()))})
    return out


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without torchrun: start the N ranks as CHILD
    processes of torch.distributed.run (nothing in this process has touched the
    GPU or imported torch yet; no exec), relay their output and exit code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


def bench_stream(args, pr, sc, t_gen):
    """LM iterations/s of the streamed solve (include/ba_hip.h ba_stream_*): the
    landmark-side data lives in pinned host memory and crosses PCIe twice per
    iteration — the honest rate for a problem that does not fit the device."""
    import torch
    from bundle_adjustment_solver_amd._lib import make_options
    from bundle_adjustment_solver_amd.solver import BaStream

    def load(st):
        st.set_cameras(pr["cam_intr"], pr["cam_T"])
        st.set_poses(pr["pose_T"], pr["pose_fixed"])
        st.set_points(pr["pt_X"], pr["pt_fixed"])
        st.set_observations(pr["obs_cam"], pr["obs_pose"], pr["obs_pt"], pr["obs_uv"])
        st.finalize()
        return st

    arena = int(args.arena_mb * 1e6)
    if arena <= 0:   # size the arenas to the largest chunk
        probe = load(BaStream(0, args.stream, 8 << 30))
        arena = int(probe.info()["largest_chunk_bytes"] * 1.02) + (1 << 20)
        probe.close()
    t_fin = time.time()
    st = load(BaStream(0, args.stream, arena))
    t_fin = time.time() - t_fin
    opt = make_options(max_iter=args.warmup + args.steps + 1, thr_step=-1.0, thr_cost=-1.0,
                       huber=args.huber)
    st.lm_begin(opt)
    st.lm_iterate(args.warmup)
    st.lm_sync()
    torch.cuda.synchronize()
    i0 = st.info()
    t0 = time.perf_counter()
    st.lm_iterate(args.steps)
    st.lm_sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    i1 = st.info()
    rows, n_done, _, _ = st.lm_sync(cap=args.warmup + args.steps + 1)
    assert n_done == args.warmup + args.steps, n_done
    up = (i1["bytes_h2d"] - i0["bytes_h2d"]) / args.steps
    down = (i1["bytes_d2h"] - i0["bytes_d2h"]) / args.steps
    n_obs = int(pr["obs_cam"].shape[0])
    result = {
        "metric": "lm_iterations_per_sec", "value": args.steps / elapsed, "unit": "it/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": WORKLOADS[args.config] +
            ("" if args.scale == 1.0 else " (scaled x%g, debug)" % args.scale),
            "n_opt_poses": int((pr["pose_fixed"] == 0).sum()),
            "n_opt_landmarks": int((pr["pt_fixed"] == 0).sum()), "n_observations": n_obs,
            "parallelism": "single GPU, observation streaming: %d landmark chunks through two "
                           "device arenas (PCIe inside the timed region)" % args.stream},
        "final_cost": rows[-1].cost if rows else None,
        "host_prep_s": {"scene_gen": t_gen, "finalize_upload": t_fin},
        "streaming": {
            "n_chunks": args.stream, "arena_MB_each": arena / 1e6,
            "arenas_MB": i1["arena_bytes"] / 1e6,
            "largest_chunk_MB": i1["largest_chunk_bytes"] / 1e6,
            "all_chunks_MB": i1["all_chunks_bytes"] / 1e6,
            "host_to_device_GB_per_iteration": up / 1e9,
            "device_to_host_GB_per_iteration": down / 1e9,
            "pcie_GBs_achieved": (up + down) / (elapsed / args.steps) / 1e9},
        "roofline": None, "cpu_baseline": None,
    }
    print(json.dumps(result))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C4", choices=sorted(WORKLOADS))
    ap.add_argument("--scale", type=float, default=1.0,
                    help="shrink the config (debug only; result is then "
                         "labelled as such)")
    ap.add_argument("--sigma", type=float, default=0.0,
                    help="pixel noise of the synthetic measurements [px] "
                         "(SURVEY.md 8d: the headline is 0, second run 0.5)")
    ap.add_argument("--huber", type=float, default=1.0,
                    help="threshold_huber_loss (solver units of 0.01 px)")
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling: landmarks and observations x N at fixed "
                         "poses (the default is strong: the problem is fixed)")
    ap.add_argument("--stream", type=int, default=0, metavar="K",
                    help="observation streaming (ba_stream_*): K landmark chunks through two "
                         "device arenas; the timed region then INCLUDES the PCIe traffic of "
                         "every iteration by construction (single GPU only)")
    ap.add_argument("--arena-mb", type=float, default=0.0,
                    help="size of each of the two arenas with --stream (default: the largest "
                         "chunk + 2 %%)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="with one rank: still run the N > 1 code path (process group, RCCL "
                         "exchange from C++, separate control kernel, final landmark gather) — "
                         "what a one-GPU box can rehearse of the multi-GPU bench with real RCCL")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # native artefacts are checked once, here, so the ranks find them built
        import __graft_entry__ as ge
        ge.build(only_if_missing=True)
        raise SystemExit(self_launch(args.gpus))
    if args.gpus > 1 and world == 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=1" % args.gpus)

    # native artefacts first, before anything initialises the GPU or the process
    # group: rank 0 alone checks the source stamps (and rebuilds what is stale);
    # the other ranks block in init_process_group / the barrier below until it
    # has joined, so nobody loads a library that is being rewritten
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    if rank == 0:
        ge.build(only_if_missing=True)

    # BA_BENCH_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs
    # than ranks (ranks share cards, the exchange is staged through the host);
    # the line is then labelled as a rehearsal, never a result.
    backend = os.environ.get("BA_BENCH_BACKEND", "nccl")
    rehearsal = backend != "nccl"
    if rehearsal:
        local_rank %= max(1, torch.cuda.device_count())
    if torch.cuda.device_count() > 0:   # (without a GPU ba_create below fails loudly: no CPU path)
        torch.cuda.set_device(local_rank)
    sharded = world > 1 or args.force_exchange
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if rehearsal:
            dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    if sharded:
        dist.barrier()   # rank 0 has finished build()
    from bundle_adjustment_solver_amd import scenes
    from bundle_adjustment_solver_amd._lib import make_options
    from bundle_adjustment_solver_amd.solver import BaProblem

    t_gen = time.time()
    factor = world if args.weak else 1
    sc = scenes.config_scene(args.config, args.scale, args.sigma, landmark_factor=factor)
    pr = scenes.scaled_problem(sc)
    t_gen = time.time() - t_gen

    if args.stream > 0:
        if world > 1:
            raise SystemExit("--stream is a single-GPU mode")
        return bench_stream(args, pr, sc, t_gen)
    p = BaProblem(local_rank)
    p.set_cameras(pr["cam_intr"], pr["cam_T"])
    p.set_poses(pr["pose_T"], pr["pose_fixed"])
    p.set_points(pr["pt_X"], pr["pt_fixed"])
    p.set_observations(pr["obs_cam"], pr["obs_pose"], pr["obs_pt"],
                       pr["obs_uv"])
    keep = []
    if sharded:
        p.set_shard(rank, world)   # (the handle keeps its own stream: the exchange
        #                            hook is told which one, see sharding.TorchExchange)
    t_fin = time.time()
    p.finalize()
    t_fin = time.time() - t_fin
    n_ranks_seen = 1
    exchange = "none"
    if sharded:
        # the per-iteration collectives: RCCL from C++ inside the library (no Python
        # between the kernels); BA_BENCH_EXCHANGE=torch keeps them in torch.distributed
        # (a Python callback twice per iteration), which is also what a gloo rehearsal
        # uses, staged through the host
        from bundle_adjustment_solver_amd.sharding import RcclExchange, TorchExchange
        if rehearsal or os.environ.get("BA_BENCH_EXCHANGE") == "torch":
            keep.append(TorchExchange(p, dist, torch.device("cuda", local_rank),
                                      stage_host=rehearsal))
            n_ranks_seen = dist.get_world_size()
            exchange = "torch.distributed/%s%s" % (backend, " (host-staged)" if rehearsal else "")
        else:
            ex = RcclExchange(p, dist, rank, world, local_rank)
            keep.append(ex)
            n_ranks_seen = ex.size()
            exchange = "RCCL all-reduce issued by libba_hip.so (ba_rccl_allreduce_hook)"

    n_obs = int(pr["obs_cam"].shape[0])
    N = p.N
    M_glob = int((pr["pt_fixed"] == 0).sum())
    # the LM loop must not stop inside the timed region: thresholds < 0
    opt = make_options(max_iter=args.warmup + args.steps + 1, thr_step=-1.0,
                       thr_cost=-1.0, huber=args.huber)

    def sync_all():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    p.lm_begin(opt)
    p.lm_iterate(args.warmup)
    p.lm_sync()
    sync_all()
    t0 = time.perf_counter()
    p.lm_iterate(args.steps)
    p.lm_sync()
    sync_all()
    elapsed = time.perf_counter() - t0
    if sharded:
        te = torch.tensor([elapsed], dtype=torch.float64,
                          device="cpu" if rehearsal else "cuda")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    rows, n_done, conv, done = p.lm_sync(cap=args.warmup + args.steps + 1)
    assert n_done == args.warmup + args.steps, (n_done, args)
    if os.environ.get("BA_BENCH_DUMP_ROWS"):  # developer aid: the trajectory of every rank
        sys.stderr.write("rank %d: %s\n" % (rank, " ".join(
            "%d:%.6g" % (r.iteration_status, r.trial_cost) for r in rows)))

    ms_per_step = elapsed / args.steps * 1e3
    result = {
        "metric": "lm_iterations_per_sec",
        # strong: LM iterations/s of the fixed problem.  weak: the problem is N times
        # the one-GPU problem, so the whole-job rate is N x (iterations/s) in units of
        # one-GPU-problem iterations (= obs_iterations_per_sec / observations of the
        # one-GPU problem)
        "value": args.steps / elapsed * factor,
        "unit": "it/s" if factor == 1 else "it/s x N (iterations of the N-times problem x N)",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak" if args.weak else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": WORKLOADS[args.config] +
            ("" if args.scale == 1.0 else " (scaled x%g, debug)" % args.scale) +
            ("" if args.sigma == 0.0 else ", pixel noise sigma %g px" % args.sigma) +
            ("" if args.huber == 1.0 else ", huber %g" % args.huber) +
            ("" if factor == 1 else ", landmarks and observations x%d (weak scaling)" % factor),
            "n_opt_poses": N, "n_opt_landmarks": M_glob,
            "n_observations": n_obs,
            "parallelism": ("landmark-shard x%d + all-reduce(S|rhs)" % world +
                            (" (REHEARSAL: %s, shared GPU)" % backend
                             if rehearsal else ""))
            if world > 1 else "single GPU",
            "exchange": exchange,
        },
        "n_ranks_seen": n_ranks_seen,
        "obs_iterations_per_sec": n_obs * args.steps / elapsed,
        "final_cost": rows[-1].cost if rows else None,
        "host_prep_s": {"scene_gen": t_gen, "finalize_upload": t_fin},
    }

    # ---- per-stage / per-kernel device times + roofline: a separate
    # instrumented pass (hipEvents around every launch, on the stream the
    # kernels run on) over the same problem, so that the headline region stays
    # un-instrumented ------------------------------------------------------
    if not args.no_roofline:
        p.enable_stage_timing(True)
        p.get_stage_ms(reset=True)
        p.get_kernel_ms(reset=True)
        n_prof = min(5, max(1, args.steps))
        opt2 = make_options(max_iter=n_prof, thr_step=-1.0, thr_cost=-1.0,
                            huber=args.huber)
        p.lm_begin(opt2)
        p.lm_iterate(n_prof)
        p.lm_sync()
        st = p.get_stage_ms(reset=True) / n_prof
        km = p.get_kernel_ms(reset=True)
        p.enable_stage_timing(False)
        names = ["build", "schur", "solve", "backsub_update", "cost",
                 "control", "exchange"]
        result["stage_ms"] = {k: float(v) for k, v in zip(names, st)}
        result["kernel_us_per_iter"] = {
            k: round(v[0] / n_prof * 1e3, 2) for k, v in km.items() if v[1]}
        result["kernel_launches_per_iter"] = {
            k: v[1] / n_prof for k, v in km.items() if v[1]}
        di = p.get_dense_info()
        result["dense_solve"] = {
            "levels": di["levels"], "tile_fill": di["fill"],
            "executed_gflop": di["flops"] / 1e9,
            "dense_gflop": ((6 * N) ** 3 / 3.0 + 4.0 * (6 * N) ** 2) / 1e9,
            "order_padded": di["npad"]}
        if world == 1:
            P = int(p.P)
            n_pose = int(pr["pose_T"].shape[0])
            O_opt = int((pr["pose_fixed"][pr["obs_pose"]] == 0).sum())
            T = int(p.lib.ba_num_schur_triples(p.h))
            B = int(p.lib.ba_num_schur_blocks(p.h))
            si = p.get_schur_info()
            Mg, Pg, Tg = si["grouped_landmarks"], si["grouped_pairs"], si["grouped_triples"]
            li = p.get_lin_info()
            Og, Op = li["group_observations"], li["pose_major_observations"]
            lin_grp = li["group_pieces"] > 0
            Ml, Pl = (M_glob - Mg, P - Pg) if lin_grp else (M_glob, P)  # the chunk kernel's share
            # algorithmic HBM bytes per launch of each streaming kernel
            # (every array the kernel must read or write, once)
            kbytes = {
                "k_cost": 24 * n_obs + 24 * M_glob + 96 * n_pose,
                # grouped landmarks, landmark and pose side in one pass: the observation
                # stream (uv only: pose and camera are the pattern's), points in; compact
                # W, undamped C (48) + b (24) out
                "k_lin_grp": 16 * Og + 24 * Mg + 96 * Pg + 72 * Mg,
                # observation records + points in; compact W, undamped C (48) + b (24) out
                "k_lin_landmarks": 32 * (n_obs - (Og if lin_grp else 0)) + 24 * Ml + 96 * Pl + 72 * Ml,
                "k_lin_poses": 24 * Op + 24 * (M_glob if Op else 0),
                "k_damp_invert": 96 * M_glob,
                "k_schur_grp": 96 * Pg + 72 * Mg,
                "k_schur_lds": 96 * (P - Pg) + 72 * (M_glob - Mg) + 4 * (T - Tg),
                "k_schur_final": 288 * B,
                "k_backsub_update": 96 * P + 192 * M_glob,
            }
            # arithmetic of the two Schur kernels: executed MFMA flops of the group
            # kernel (2048 per v_mfma_f64_16x16x4), FMA flops of the super-run kernel
            # (63 FMAs per half-slot triple, two half slots)
            kflops = {"k_schur_grp": 2048.0 * si["group_mfma"],
                      "k_schur_lds": 2.0 * 126.0 * (T - Tg - si["list_triples"])}
            peaks = {"k_schur_grp": ("mfma", FP64_MFMA_MEASURED_TFLOPS),
                     "k_schur_lds": ("fma", FP64_FMA_MEASURED_TFLOPS)}
            tot = {k: v[0] / n_prof for k, v in km.items() if v[1]}
            dense = sum(tot.get(k, 0.0) for k in
                        ("k_chol_diag", "k_chol_trsm", "k_chol_diag_trsm", "k_chol_update",
                         "k_chol_back", "k_chol_level", "k_chol_tail"))
            live = [k for k in kbytes if tot.get(k) and kbytes[k] > 0]
            dom = max(live, key=lambda k: tot[k])
            dom_ms = tot[dom] / km[dom][1] * n_prof   # average launch duration
            ach = kbytes[dom] / (dom_ms * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic(dom, args.config)
            result["roofline"] = {
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": dom, "avg_launch_us": dom_ms * 1e3,
                "algorithmic_bytes_per_launch": kbytes[dom]}
            if dom in kflops and kflops[dom] > 0:
                # the Schur kernels are arithmetic- (matrix-core / LDS-) bound, not
                # HBM-bound: the honest fraction is against the issue peak measured
                # on this part (tools/mfma_f64_peak.hip)
                kind, pk = peaks[dom]
                tf = kflops[dom] / (dom_ms * 1e-3) / 1e12
                result["roofline"].update({
                    "bound": "mfma", "achieved": tf, "peak": pk, "unit": "TFLOP/s",
                    "frac": tf / pk, "pipe": kind,
                    "hbm": {"achieved_GBs": ach, "frac": ach / HBM_PEAK_GBS}})
            result["roofline_all_hbm_kernels"] = {
                k: {"us": round(tot[k] * 1e3, 1),
                    "GB/s": round(kbytes[k] / (tot[k] * 1e-3) / 1e9, 1),
                    "frac": round(kbytes[k] / (tot[k] * 1e-3) / 1e9 /
                                  HBM_PEAK_GBS, 4),
                    "pmc_traffic_bytes": pmc_traffic(k, args.config)[0]}
                for k in live}
            result["roofline_schur_arithmetic"] = {
                k: {"pipe": peaks[k][0], "TFLOP/s": round(kflops[k] / (tot[k] * 1e-3) / 1e12, 2),
                    "measured_peak_TFLOP/s": peaks[k][1],
                    "frac": round(kflops[k] / (tot[k] * 1e-3) / 1e12 / peaks[k][1], 4)}
                for k in kflops if tot.get(k) and kflops[k] > 0}
            result["schur_paths"] = si
            result["linearisation_paths"] = li
            result["roofline_dense_solve"] = {
                "bound": "mfma",
                "achieved": di["flops"] / (dense * 1e-3) / 1e12 if dense else 0,
                "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": (di["flops"] / (dense * 1e-3) / 1e12 /
                         FP64_MFMA_PEAK_TFLOPS) if dense else 0,
                "frac_of_measured_peak": (di["flops"] / (dense * 1e-3) / 1e12 /
                                          FP64_MFMA_MEASURED_TFLOPS) if dense else 0,
                "us": dense * 1e3,
                "note": "executed flops of the structure-aware factorisation; "
                        "latency-bound (%d dependent levels); measured fp64 "
                        "MFMA issue peak on this part: %.1f TFLOP/s "
                        "(tools/mfma_f64_peak.hip)" % (di["levels"], FP64_MFMA_MEASURED_TFLOPS)}

    # ---- the whole Solve() of a caller: finalize (host plan + uploads) + 50 LM
    # iterations with the reference's default thresholds + write-back, fresh handle,
    # host arrays as inputs (PCIe inside) — what Summary::total_time covers --------
    if rank == 0 and world == 1 and not args.no_roofline:
        t_a = time.perf_counter()
        q = BaProblem(local_rank)
        q.set_cameras(pr["cam_intr"], pr["cam_T"])
        q.set_poses(pr["pose_T"], pr["pose_fixed"])
        q.set_points(pr["pt_X"], pr["pt_fixed"])
        q.set_observations(pr["obs_cam"], pr["obs_pose"], pr["obs_pt"], pr["obs_uv"])
        q.finalize()
        t_b = time.perf_counter()
        erows, econv = q.solve(make_options(max_iter=50, thr_step=1e-6, thr_cost=1e-6,
                                            huber=args.huber))
        t_c = time.perf_counter()
        q.get_poses()
        q.get_points()
        t_d = time.perf_counter()
        # a second solve of the same structure with new values: no re-planning
        q.update_values(pr["pose_T"], pr["pt_X"])
        t_e = time.perf_counter()
        q.solve(make_options(max_iter=50, thr_step=1e-6, thr_cost=1e-6, huber=args.huber))
        q.get_poses()
        q.get_points()
        t_f = time.perf_counter()
        result["end_to_end_solve_s"] = {
            "finalize": t_b - t_a, "lm_iterations": len(erows), "converged": bool(econv),
            "solve": t_c - t_b, "write_back": t_d - t_c, "total": t_d - t_a,
            "resolve_update_values": t_e - t_d, "resolve_total": t_f - t_d}
        q.close()

    # ---- CPU baseline: the oracle on the box's host cores, 1 thread -------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle_py as O   # the checker: this leg only
        o = O.Oracle(pr)
        # SURVEY.md 8(d) / BASELINE.md 3: median of >= 5 iterations after the first —
        # where that fits the bounded sample (C1 - C3: 6 iterations); at C4 one
        # iteration is ~20 s of the reference-style unblocked pivoted LDLT
        n_cpu = 6 if args.config in ("C1", "C2", "C3") else 1
        t = time.perf_counter()
        orows, _ = o.solve(O.make_options(max_iter=n_cpu, thr_step=-1.0,
                                          thr_cost=-1.0, huber=args.huber))
        dt = time.perf_counter() - t
        per_it = (float(np.median([r.iter_time_ms for r in orows[1:]])) * 1e-3
                  if n_cpu > 1 else dt)
        result["cpu_baseline"] = {
            "value": 1.0 / per_it, "unit": "it/s", "cores": 1, "kind": "port",
            "sample": "%s of the same %s problem (block-sparse oracle, reference's "
                      "unblocked pivoted LDLT for the dense solve), %.1f s in all; "
                      "stage ms build/schur/solve/control (sum) = %s" % (
                          "median of iterations 2-%d" % n_cpu if n_cpu > 1 else
                          "1 LM iteration", args.config, dt, ", ".join(
                              "%.0f" % v for v in o.stage_ms())),
            "host_cpus": os.cpu_count(),
        }
        # same first iteration on GPU and CPU: parity spot check
        if rows:
            result["cpu_baseline"]["first_iter_trial_cost_rel_diff"] = abs(
                rows[0].trial_cost - orows[0].trial_cost) / abs(
                    orows[0].trial_cost)
        del o
        # SURVEY.md 8(d), second CPU figure: the reference's OWN storage (dense
        # N x M grids of 6x3 / 3x6 blocks, re-zeroed by ResetStorageMatrices
        # every iteration, reference :343-379).  Runnable at C1 (21 MB) and C2
        # (5.6 GB) only.
        if args.config in DENSE_GRID_GB:   # (the BASELINE configs; not the off-path ones)
            result["cpu_baseline"]["dense_faithful"] = dense_faithful_baseline(
                O, scenes, args.config, pr if args.scale == 1.0 else None, args.huber)

    if sharded:
        # the end of a sharded Solve: every rank gathers every landmark (reference
        # :1018-1022 writes back all of them); timed apart from the iterations
        t_g = time.perf_counter()
        p.gather_points()
        Xall, owned = p.get_points()
        result["gather_points_ms"] = (time.perf_counter() - t_g) * 1e3
        assert bool(owned.all())
    if rank == 0:
        print(json.dumps(result))
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
