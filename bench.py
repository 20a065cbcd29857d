#!/usr/bin/env python3
"""bench.py — analysis frames/s of the eaQHM hot path on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one complete run of the adaptation loop (functions.py:163-402) over the workload:
adaptation 0..maxAdpt or until the reference's stop rule fires.  Inputs (signal, pitch grid, frame
tables) are resident in HBM before the timed region; result packing into Python structs is outside it
(SURVEY.md §8d).  Workload at N GPUs = SA19.WAV tiled N times ("sa19x<N>", weak scaling: the frames per
GPU stay fixed), `female`, maxAdpt=5 — at N=1 this is BASELINE.json configs[1].  Pitch tracks come from
committed fixtures produced by the reference's SWIPE' (tests/golden), because the reference cannot travel
to the GPU box.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

PEAK_FP64_TFLOPS = 78.6   # MI355X FP64 (vector = matrix) spec: half the 157.3 TF FP32 vector peak
                          # of MI355X_MICROARCH.md's chip table; measured here: mfma_f64 48, v_fma_f64 63 TF/s


def ls_flops(N, Kc):
    """Algorithmic FP64 flops of one frame's least squares:
    Hermitian 3-block Gramian + RHS + complex Cholesky + two triangular solves."""
    N = np.asarray(N, dtype=np.float64)
    Kc = np.asarray(Kc, dtype=np.float64)
    return 12 * N * Kc * (Kc + 1) + 8 * N * Kc + (32.0 / 3.0) * Kc ** 3 + 32 * Kc ** 2


def load_workload(reps):
    from scipy.io import wavfile
    from eaqhm_amd import prologue
    fs, x = wavfile.read(os.path.join(GOLDEN, "SA19.WAV"))
    x = np.tile(x, reps)
    s = x / 32768.0
    if reps == 1:
        track = np.load(os.path.join(GOLDEN, "sa19_female_default.npz"))["swipe_track"]
    else:
        g = np.load(os.path.join(GOLDEN, "prep_fixtures.npz"))
        key = "sa19x%d_f0s_5ms" % reps
        if key not in g.files:
            raise SystemExit("no pitch fixture for SA19 x%d (available: x2, x4, x8, x10)" % reps)
        track = g[key]
    grid = prologue.resample_track(track, np.arange(0, len(s) - 1, round(fs * 5 / 1000)) / fs)
    frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "female")
    prologue.apply_full_waveform(frames, len(s), 32 * 15)
    return fs, s, grid, frames, fstep


def cpu_baseline():
    """The oracle (NumPy port of the reference's path) on a bounded sample of the same workload:
    SA19.WAV, adaptations 0 and 1 (8,338 LS frames), on this host's cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import eaqhm_oracle as O
    g = np.load(os.path.join(GOLDEN, "sa19_female_default.npz"))
    from scipy.io import wavfile
    fs, x = wavfile.read(os.path.join(GOLDEN, "SA19.WAV"))
    s = x / 32768.0
    from threadpoolctl import threadpool_limits
    threads = min(8, os.cpu_count() or 1)      # small matrices: more BLAS threads only add contention
    with threadpool_limits(limits=threads):
        t0 = time.time()
        r = O.analyse(s, fs, g["f0s_5ms"], g["vuv_ti"], g["vuv_isSpeech"], g["vuv_isVoiced"], int(g["frame_step"]),
                      f0min=160, maxAdpt=1)
        dt = time.time() - t0
    return {"value": r["n_ls_frames"] / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "SA19.WAV, adaptations 0-1 (%d LS frames, %.1f s), oracle/eaqhm_oracle.py, host has %d cores"
                      % (r["n_ls_frames"], dt, os.cpu_count()),
            "srer_db": [float(v) for v in r["SRER"]]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--max-adpt", type=int, default=5)
    ap.add_argument("--reps", type=int, default=0, help="tile SA19 this many times (default: --gpus)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    # stdout carries exactly ONE JSON line: everything else (RCCL prints a version banner to stdout) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs a torch.distributed.run launch with %d ranks" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    use_dist = world > 1 or os.environ.get("EAQHM_FORCE_DIST") == "1"   # the latter: 1-rank rehearsal of the RCCL path
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    from eaqhm_amd.engine import DeviceAnalysis, FramePlan, Sharding
    reps = args.reps or args.gpus
    fs, s, grid, frames, fstep = load_workload(reps)
    plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
    shard = Sharding(rank, world, dist.group.WORLD if use_dist else None)
    eng = DeviceAnalysis(s, s, plan, 160, args.max_adpt, device_index=local, shard=shard)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.reset()
        eng.run()
    eng.profile = True               # HIP events around every LS launch, on the launch stream
    barrier()
    t0 = time.perf_counter()
    frames_done = 0
    n_adpt = 0
    for _ in range(args.steps):
        eng.reset(keep_timeline=True)
        n_adpt = eng.run()
        frames_done += eng.n_ls_frames
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt, float(frames_done)], dtype=torch.float64, device="cuda")
    if use_dist:
        tmax = tt[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt[1:].clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, frames_total = float(tmax.item()), float(tsum.item())
    else:
        frames_total = float(frames_done)
    srer = [float(v) for v in eng.SRER]

    # ---- roofline of the dominant kernel (eaqhm_ls_kernel), from the events of the timed region
    torch.cuda.synchronize()
    ls_ms = [e0.elapsed_time(e1) for (a, st, e0, e1) in eng.timeline if st == "ls"]
    post_ms = [e0.elapsed_time(e1) for (a, st, e0, e1) in eng.timeline if st == "post"]
    gather_ms = [e0.elapsed_time(e1) for (a, st, e0, e1) in eng.timeline if st == "gather"]
    N = 2 * plan.frame_wl[eng.f_lo:eng.f_hi].astype(np.int64) + 1
    flops_per_launch = [float(ls_flops(N, 2 * plan.frame_K[eng.f_lo:eng.f_hi].astype(np.int64) + 1).sum())]
    for nc in eng.ncol_hist[:max(n_adpt - 1, 0)]:
        flops_per_launch.append(float(ls_flops(N, 2 * nc.cpu().numpy().astype(np.int64) + 1).sum()))
    launches_per_step = len(flops_per_launch)
    flops_step = sum(flops_per_launch)
    ls_total_s = sum(ls_ms) / 1e3
    achieved = flops_step * args.steps / ls_total_s / 1e12 if ls_total_s > 0 else 0.0
    # HBM traffic of that kernel: not measurable from inside this process; taken from the committed rocprofv3 PMC
    # passes of this same command (profiles/r01_final/pmc_hbm_traffic.json), N=1 workload only
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_final", "pmc_hbm_traffic.json")
    if world == 1 and reps == 1 and os.path.exists(pmc):
        traffic = json.load(open(pmc)).get("eaqhm_ls_tile_kernel", {}).get("hbm_bytes_per_launch_fetch_doubled")
    roofline = {"bound": "mfma", "kernel": "eaqhm_ls_tile_kernel", "achieved": achieved, "peak": PEAK_FP64_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_TFLOPS, "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_final)",
                "algorithmic_bytes_per_launch": float(np.sum(8 * N) + 32 * np.sum(2 * plan.frame_K[eng.f_lo:eng.f_hi] + 1)),
                "flops_per_launch_mean": flops_step / max(launches_per_step, 1),
                "launch_ms_mean": float(np.mean(ls_ms)) if ls_ms else None,
                "launches_timed": len(ls_ms),
                "post_stage_ms_mean": float(np.mean(post_ms)) if post_ms else None,
                "all_gather_ms_mean": float(np.mean(gather_ms)) if gather_ms else None}

    out = {"metric": "analysis_frames_per_sec", "value": frames_total / dt, "unit": "frames/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
           "data": "SA19.WAV (the reference's sample recording) tiled x%d; pitch track = fixture from the "
                   "reference's SWIPE'" % reps,
           "config": {"workload": "sa19x%d_female_maxAdpt%d" % (reps, args.max_adpt), "samples": int(plan.L),
                      "fs": int(fs), "ls_frames_per_adaptation": int(plan.n_frames), "adaptations_executed": n_adpt,
                      "Kmax": int(plan.Kmax), "parallelism": "frames sharded x%d, all-gather of boundary records per adaptation" % world},
           "final_srer_db": max(srer), "srer_db": srer, "roofline": roofline}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    else:
        out["cpu_baseline"] = None
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
