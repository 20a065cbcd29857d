"""GPU test of the frame-sharded path with the REAL kernels: two processes share the one GPU of the test box
(NCCL needs one device per rank, so the two collectives are staged through the host over gloo here — the
buffers, ranges, halos and kernels are exactly those of the multi-GPU run)."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_golden

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, out_path, workload="sa19", budget=None):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import eaqhm_amd  # noqa: F401
    from eaqhm_amd import prologue
    from eaqhm_amd.engine import DeviceAnalysis, FramePlan, Sharding
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    class HostStaged(Sharding):
        def _all_gather_into(self, out, part):
            host = out.cpu()
            dist.all_gather_into_tensor(host, part.cpu().clone(), group=self.group)
            out.copy_(host)

        def all_reduce_sum(self, t):
            host = t.cpu()
            dist.all_reduce(host, group=self.group)
            t.copy_(host)

    if workload == "sa19":
        g = load_golden("sa19_female_default.npz")
        fs, s = prologue.read_signal(os.path.join(GOLDEN, "SA19.WAV"))
        max_adpt = 5
    else:           # the signal with a span of digital zeros (empty-row seeding), cut by the rank boundary
        g = load_golden("seed16k_1p2s_adpt6.npz")
        fs, s = 16000, g["wav_int16"] / 32768.0
        max_adpt = 6
    grid = prologue.resample_track(g["swipe_track"], np.arange(0, len(s) - 1, round(fs * 5 / 1000)) / fs)
    frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "female")
    prologue.apply_full_waveform(frames, len(s), 480)
    plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
    eng = DeviceAnalysis(s, s, plan, 160, max_adpt, shard=HostStaged(rank, world, dist.group.WORLD),
                         track_budget_bytes=budget)
    seen = {}

    def hook(a, e):
        rec = e.records[0].clone()
        e.shard.all_gather_rows(rec, e.bounds)        # test-only: complete rows of this adaptation
        seen["rec%d" % a] = rec[:plan.No_ti].cpu().numpy()

    eng.run(on_adaptation=hook if workload != "sa19" else None)
    fin = eng.final_arrays()
    if rank == 0:
        np.savez(out_path, SRER=np.array(eng.SRER), frames_rank0=eng.n_ls_frames, bounds=np.array(eng.bounds),
                 blocks=len(eng.blocks), **fin, **seen)
    dist.destroy_process_group()


def test_two_ranks_one_gpu_matches_reference(tmp_path, sa19_golden):
    import torch.multiprocessing as mp
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(2, 29600 + os.getpid() % 300, out), nprocs=2, join=True)
    got = np.load(out)
    g = sa19_golden
    assert len(got["SRER"]) == 6 and np.abs(got["SRER"] - g["SRER"]).max() < 1e-6
    assert np.abs(got["s_recon"] - g["s_recon"]).max() <= 1e-9
    assert 0 < int(got["frames_rank0"]) < 6 * 4169
    cells = g["det_cells"]
    i, k = cells[:, 0], cells[:, 1]
    ok = got["am"][i, k] != 0
    assert ok.mean() > 0.999
    assert np.abs(got["am"][i, k][ok] - g["det_am"][ok]).max() <= 1e-8 * g["det_am"].max()
    assert np.abs(got["fm"][i, k][ok] - g["det_fm"][ok]).max() <= 1e-3
    d = (got["pk"][i, k][ok] - g["det_pk"][ok] + np.pi) % (2 * np.pi) - np.pi
    assert np.abs(d).max() <= 1e-5


def test_two_ranks_seeding_across_the_rank_boundary(tmp_path):
    """The silent span (instants 534-719) straddles the boundary between the two ranks' instant ranges: seeded rows of
    rank 0 are visible inside rank 1's first windows only through the halo frames' flags (engine.py: the frame tables start at ext0, before the rank's first frame)."""
    import torch.multiprocessing as mp
    from test_gpu_parity import check_seeding_result
    g = load_golden("seed16k_1p2s_adpt6.npz")
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(2, 29300 + os.getpid() % 300, out, "seed"), nprocs=2, join=True)
    got = np.load(out)
    b = got["bounds"]
    z0, z1 = g["zero_span"] // 15
    assert z0 + 20 < b[1] < z1 - 20, "rank boundary %d not inside the silent span %d..%d" % (b[1], z0, z1)
    seen = {a: got["rec%d" % a] for a in range(4)}
    check_seeding_result(g, got["SRER"], seen, {k: got[k] for k in ("am", "fm", "pk", "a0", "s_recon")})


def test_two_ranks_with_time_block_streaming(tmp_path):
    """Ranks AND time blocks: the seeding signal on two ranks, each working its range off in blocks of <= 1500 samples
    of tracks (the silent span crosses both the rank boundary and block boundaries).  Same checks as the resident run."""
    import torch.multiprocessing as mp
    from test_gpu_parity import check_seeding_result
    g = load_golden("seed16k_1p2s_adpt6.npz")
    out = str(tmp_path / "r0.npz")
    budget = 18 * 59 * 1500            # DeviceAnalysis.TRACK_BYTES_PER_CELL * Kmax * samples
    mp.spawn(_worker, args=(2, 29000 + os.getpid() % 300, out, "seed", budget), nprocs=2, join=True)
    got = np.load(out)
    assert int(got["blocks"]) >= 4
    seen = {a: got["rec%d" % a] for a in range(4)}
    check_seeding_result(g, got["SRER"], seen, {k: got[k] for k in ("am", "fm", "pk", "a0", "s_recon")})
