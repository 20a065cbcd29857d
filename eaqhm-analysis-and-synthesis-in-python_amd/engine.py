"""Device-side adaptation loop (functions.py:148-402) on one MI355X — or on N of them with the analysis
instants of each adaptation sharded over the ranks (one process per GPU).

Per adaptation a rank runs
    [frame_prep] -> ls_batch            its own contiguous range of analysis instants
    all-gather (RCCL over xGMI) of the frame-centre records its neighbours' interpolation reads: the boundary
    rows of every rank's range (Sharding.share_rows); the complete records are gathered once, with the results
    spline_solve                        every rank, the instants its own time range looks at
    eval_synth                          its own time range plus a halo of max(wl) samples, so the dense
                                        tracks its frames will window in the next adaptation are local
    all-reduce of the two error sums    (16 bytes; skipped when world_size == 1)
and the host reads back the error sums once to apply the stop rule of functions.py:394-396 — every rank
computes the same SRER from the same reduced sums, so no broadcast of the decision is needed.

Dense state lives in HBM for the whole run: am_current / fm_current as [Kmax][L] (time contiguous),
frame-centre records as [No_ti][3*Kmax+1].  All numerics are in libeaqhm_hip.so; this module only
allocates buffers (torch-ROCm tensors), fills the small per-frame tables and sequences the launches.
"""
import numpy as np

from .hip import Context


class FramePlan:
    """Host bookkeeping of functions.py:148-157, :180-191: analysis instants, which of them are
    analysed, and the adaptation-0 frame set-up (pitch, harmonic count, half window)."""

    def __init__(self, length, fs, f0_grid, frames, frame_step, step, pitch_periods, analysis_window, partials):
        f0_grid = np.asarray(f0_grid, dtype=np.float64)
        self.L, self.fs, self.step = int(length), fs, int(step)
        self.Fmax = int(fs / 2 - 200)                                            # functions.py:115
        self.Kmax = int(partials) if partials > 0 else int(round(self.Fmax / np.min(f0_grid[:, 1])) + 10)
        aws = analysis_window * step                                             # functions.py:123
        self.ti = np.arange(1, self.L, step)                                     # 1-based instants
        self.No_ti = len(self.ti)
        pos = self.ti / frame_step
        idx = pos.astype(int)
        voiced5 = np.array([bool(f.isVoiced) for f in frames])
        self.in_bounds = (self.ti > aws) & (self.ti < self.L - aws)              # functions.py:180
        self.analysed = self.in_bounds.copy()
        ib = self.in_bounds
        self.analysed[ib] = voiced5[idx[ib] - 1] & voiced5[idx[ib]]              # functions.py:181
        sel = np.flatnonzero(self.analysed)
        frac = pos[sel] - idx[sel]
        f0 = (1 - frac) * f0_grid[idx[sel] - 1, 1] + frac * f0_grid[idx[sel], 1]  # functions.py:185
        self.frame_inst = sel.astype(np.int32)
        self.frame_c = (self.ti[sel] - 1).astype(np.int32)
        self.frame_f0 = f0
        self.frame_K = np.minimum(self.Kmax, (self.Fmax / f0).astype(int)).astype(np.int32)   # :187
        self.frame_wl = np.maximum(120, np.round((pitch_periods / 2) * (fs / f0))).astype(np.int32)  # :191
        self.n_frames = len(sel)
        # the loop variable f0 survives adaptation 0, so adaptations >= 1 see the LAST frame's pitch
        self.f0_stale = float(f0[-1]) if self.n_frames else 0.0
        self.wl_max = int(self.frame_wl.max()) if self.n_frames else 0
        if self.No_ti < 4:
            raise ValueError("signal too short: fewer than 4 analysis instants (interp1d kind=3 needs 4)")
        if self.n_frames:
            if (self.frame_c - self.frame_wl).min() < 0 or (self.frame_c + self.frame_wl).max() >= self.L:
                raise ValueError("an analysis window reaches outside the signal (analysisWindow too small)")


def ls_cost(N, Kc):
    """Algorithmic FP64 flops of one frame's least squares: Hermitian 3-block Gramian + right-hand
    side + complex Cholesky + two triangular solves."""
    N = np.asarray(N, dtype=np.float64)
    Kc = np.asarray(Kc, dtype=np.float64)
    return 12 * N * Kc * (Kc + 1) + 8 * N * Kc + (32.0 / 3.0) * Kc ** 3 + 32 * Kc ** 2


class Sharding:
    """Contiguous ranges of analysis instants per rank, balanced by the LS cost of their frames (SURVEY.md §8e:
    "balanced by sum F(N,Kc), not by count": the cost of a frame varies ~2.5x with the pitch); `group` is a
    torch.distributed process group or None for a single process.  Every rank derives the same bounds from the
    same frame plan, so no communication is needed to agree on them."""

    def __init__(self, rank=0, world=1, group=None):
        self.rank, self.world, self.group = int(rank), int(world), group
        self.collective = group is not None      # a 1-rank group still goes through the collectives (rehearsal)
        self.bounds = None

    def balance(self, n_instants, cost=None):
        """Fix the instant ranges: bounds[r] .. bounds[r+1] is rank r's.  `cost` = per-instant LS cost (zero for the
        instants that are not analysed); None or all-zero -> equal counts."""
        T, W = int(n_instants), self.world
        if cost is None or W == 1 or float(np.sum(cost)) <= 0.0:
            c = -(-T // W)
            b = [min(r * c, T) for r in range(W + 1)]
        else:
            cum = np.cumsum(np.asarray(cost, dtype=np.float64))
            cuts = np.searchsorted(cum, cum[-1] * np.arange(1, W) / W, side="left") + 1
            b = [0] + [int(min(v, T)) for v in cuts] + [T]
            for r in range(1, W + 1):
                b[r] = max(b[r], b[r - 1])
        b[W] = T
        self.bounds = b
        return b

    def instants(self, n_instants, rank=None):
        if self.bounds is None or self.bounds[-1] != n_instants:
            self.balance(n_instants)
        r = self.rank if rank is None else rank
        return self.bounds[r], self.bounds[r + 1]

    def _all_gather_into(self, out, part):
        """out = concatenation over ranks of `part` (RCCL all-gather of equal parts)."""
        import torch.distributed as dist
        dist.all_gather_into_tensor(out, part, group=self.group)

    def all_gather_rows(self, buf, n_instants):
        """Every rank ends up with every rank's rows of `buf` (rank r owns rows bounds[r] .. bounds[r+1]).  The ranges
        differ in length, so this is a sum over ranks of the buffer with the foreign rows zeroed — done once per run,
        when the results are collected, or per adaptation only for inputs too short for the boundary exchange."""
        if not self.collective:
            return
        lo, hi = self.instants(n_instants)
        buf[:lo].zero_()
        buf[hi:].zero_()
        self.all_reduce_sum(buf)

    def share_rows(self, buf, n_instants, margin):
        """What the interpolation of one rank's time range reads from the other ranks' rows of `buf`: the `margin`
        rows on either side of its own range and the first rows of the file (pad knots of short runs).  Every rank
        contributes the first and the last `margin` rows of its range to one small all-gather and copies its two
        neighbours' parts into place; the full rows are gathered once, when the results are collected
        (all_gather_rows).  Falls back to the full exchange when some range is shorter than the margin.
        Returns True if only boundary rows were exchanged."""
        if not self.collective:
            return False
        self.instants(n_instants)
        b = self.bounds
        if margin < 4 or min(b[r + 1] - b[r] for r in range(self.world)) < margin:
            self.all_gather_rows(buf, n_instants)
            return False
        import torch
        r, m = self.rank, margin
        lo, hi = b[r], b[r + 1]
        part = torch.cat((buf[lo:lo + m], buf[hi - m:hi]))
        got = buf.new_empty((self.world * 2 * m, buf.shape[1]))
        self._all_gather_into(got, part)
        if r > 0:
            buf[lo - m:lo].copy_(got[(r - 1) * 2 * m + m:(r - 1) * 2 * m + 2 * m])     # left neighbour's last rows
            k = min(m, lo - m)                                                          # rank 0's first rows
            if k > 0:
                buf[0:k].copy_(got[0:k])
        if r < self.world - 1:
            buf[hi:hi + m].copy_(got[(r + 1) * 2 * m:(r + 1) * 2 * m + m])             # right neighbour's first rows
        return True

    def all_reduce_sum(self, t):
        if not self.collective:
            return
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)


class DeviceAnalysis:
    """Buffers + launch sequence of one analysis run on one GPU (rank)."""

    def __init__(self, s, target, plan, f0min, max_adpt, device_index=0, shard=None, keep_raw=False, ctx=None):
        import torch
        self.torch = torch
        self.ctx = ctx if ctx is not None else Context(device_index)
        self.plan = plan
        self.f0min = float(f0min)
        self.max_adpt = int(max_adpt)
        self.shard = shard if shard is not None else Sharding()
        dev = self.ctx.device
        p = plan
        f64, i32 = torch.float64, torch.int32
        self.s = torch.as_tensor(np.ascontiguousarray(s), dtype=f64, device=dev)
        self.target = self.s if target is s else torch.as_tensor(np.ascontiguousarray(target), dtype=f64, device=dev)
        self.std_det = float(np.std(target))                                     # functions.py:161
        K, L, T = p.Kmax, p.L, p.No_ti
        self.RS = 3 * K + 1
        # this rank's instants, frames and time range
        sh = self.shard
        cost = np.zeros(T)
        cost[p.frame_inst] = ls_cost(2 * p.frame_wl.astype(np.int64) + 1, 2 * p.frame_K.astype(np.int64) + 1)
        sh.balance(T, cost)
        self.i_lo, self.i_hi = sh.instants(T)
        mine = np.flatnonzero((p.frame_inst >= self.i_lo) & (p.frame_inst < self.i_hi))
        lo, hi = (int(mine[0]), int(mine[-1]) + 1) if len(mine) else (0, 0)
        self.f_lo, self.f_hi, self.nf = lo, hi, hi - lo
        # exclusive time ranges partition [0, L): boundary r = first sample of rank r's first instant
        def bound(r):
            if r <= 0:
                return 0
            if r >= sh.world or sh.bounds[r] >= T:
                return L
            return sh.bounds[r] * p.step
        self.s_lo, self.s_hi = bound(sh.rank), bound(sh.rank + 1)
        self.t_lo = max(0, self.s_lo - p.wl_max)
        self.t_hi = min(L, self.s_hi + p.wl_max)
        # rows of the other ranks' records the interpolation of [t_lo, t_hi) reads: the halo, the spline range beyond it
        # (+-2, +4), run detection (+-41)
        self.margin = -(-p.wl_max // p.step) + 48
        self.partial_rows = False
        # instants whose run codes / spline moments the evaluation of [t_lo, t_hi) looks at
        self.sp_lo = max(0, (max(self.t_lo, 1) - 1) // p.step - 2)
        self.sp_hi = max(self.sp_lo + 1, min(T, (max(self.t_hi, 1) - 1) // p.step + 4))

        def dev_i32(a):
            return torch.as_tensor(np.ascontiguousarray(a[lo:hi]), dtype=i32, device=dev)

        self.frame_inst, self.frame_c, self.frame_wl = dev_i32(p.frame_inst), dev_i32(p.frame_c), dev_i32(p.frame_wl)
        self.frame_K = dev_i32(p.frame_K)
        self.frame_f0 = torch.as_tensor(np.ascontiguousarray(p.frame_f0[lo:hi]), dtype=f64, device=dev)
        # frame_prep also visits the frames just before this rank's first one whose empty-row seeding
        # (functions.py:204-210) would be visible inside this rank's windows
        ext = lo
        if self.nf:
            ext = int(np.searchsorted(p.frame_c, p.frame_c[lo] - p.wl_max, side="left"))
        self.n_ext, self.e0 = hi - ext, lo - ext
        self.frame_c_ext = torch.as_tensor(np.ascontiguousarray(p.frame_c[ext:hi]), dtype=i32, device=dev)
        self.ncol_ext = torch.zeros(max(self.n_ext, 1), dtype=i32, device=dev)
        self.cols_ext = torch.zeros(max(self.n_ext, 1) * K, dtype=i32, device=dev)
        self.ncol = self.ncol_ext[self.e0:]
        self.cols = self.cols_ext[self.e0 * K:]
        self.seeded = torch.zeros(L, dtype=torch.uint8, device=dev)
        self.any_seed = torch.zeros(1, dtype=i32, device=dev)
        # dense tracks (functions.py:159-160) — harmonic-major
        self.am_cur = torch.zeros(K, L, dtype=f64, device=dev)
        self.fm_cur = torch.zeros(K, L, dtype=f64, device=dev)
        # frame-centre records, double-buffered: [0] = adaptation in flight, [1] = last accepted
        self.records = [torch.zeros(T, self.RS, dtype=f64, device=dev) for _ in range(2)]
        self.ph_knot = [torch.zeros(T, K, dtype=f64, device=dev) for _ in range(2)]
        self.s_hat = [torch.zeros(L, dtype=f64, device=dev) for _ in range(2)]
        self.code = torch.zeros(T, K, dtype=torch.uint8, device=dev)
        self.mom = torch.zeros(T, K + 1, dtype=f64, device=dev)
        self.partials = torch.zeros(self.ctx.eval_partials_len(0, L, p.step), dtype=f64, device=dev)
        self.sums = torch.zeros(8, dtype=f64, device=dev)    # {sum d, sum d^2, n, SRER dB, LS faults, -, -, -}
        self.raw = None
        if keep_raw:
            self.raw = (torch.zeros(max(self.nf, 1), 2 * (2 * K + 1), dtype=f64, device=dev),
                        torch.zeros(max(self.nf, 1), 2 * (2 * K + 1), dtype=f64, device=dev))
        self.ncol_hist = []
        self.SRER = []
        self.n_ls_frames = 0
        self.seeded_on_break = None
        self.timeline = []          # (adaptation, stage, start_event, end_event) when profiling is on
        self.profile = False

    # ------------------------------------------------------------------ stages
    def _mark(self):
        if not self.profile:
            return None
        e = self.torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def ls_stage(self, a):
        """Per-frame LS of adaptation `a` for this rank's frames -> rows of records[0]."""
        p, c = self.plan, self.ctx
        # No clearing of records[0]: every analysed instant's row is rewritten completely by its frame (the set of
        # analysed instants never changes), the rows of the others are never written and stay zero from the
        # allocation, and the rows of other ranks arrive with the all-gather.
        if self.nf == 0:
            return
        if a > 0:
            c.frame_prep(self.fm_cur, p.L, p.Kmax, self.frame_c_ext, self.n_ext, self.ncol_ext, self.cols_ext,
                         self.seeded, self.any_seed)
            if self.profile:
                self.ncol_hist.append(self.ncol.clone())
        raw_a, raw_s = self.raw if self.raw is not None else (None, None)
        e0 = self._mark()
        c.ls_batch(0 if a == 0 else 1, self.s, p.L, p.fs, self.am_cur, self.fm_cur, p.Kmax, self.frame_inst,
                   self.frame_c, self.frame_wl, self.frame_f0, self.frame_K, self.ncol, self.cols, self.seeded,
                   self.any_seed, self.nf, p.wl_max, a, p.f0_stale, self.f0min, self.records[0], raw_a, raw_s)
        e1 = self._mark()
        if self.profile:
            self.timeline.append((a, "ls", e0, e1))
        self.n_ls_frames += self.nf

    def post_launch(self, a):
        """Enqueue interpolation + synthesis + error sums of adaptation `a` (no host read)."""
        p, c, sh = self.plan, self.ctx, self.shard
        g0 = self._mark()
        self.partial_rows = sh.share_rows(self.records[0], p.No_ti, self.margin) or self.partial_rows
        e0 = self._mark()
        if self.profile and sh.collective:
            self.timeline.append((a, "gather", g0, e0))
        c.spline_solve(self.records[0], p.No_ti, p.Kmax, p.step, self.code, self.mom, self.sp_lo, self.sp_hi)
        if self.s_hi > self.s_lo:
            c.eval_synth(self.records[0], self.code, self.mom, p.No_ti, p.Kmax, p.step, p.fs, p.L,
                         self.t_lo, self.t_hi, self.s_lo, self.s_hi, self.target, self.std_det, self.am_cur,
                         self.fm_cur, self.ph_knot[0], self.s_hat[0], self.partials, self.sums)
        else:
            self.sums.zero_()
        e1 = self._mark()
        if self.profile:
            self.timeline.append((a, "post", e0, e1))

    def post_result(self):
        """SRER of the adaptation enqueued last (host float): the one device->host read of an adaptation.  The same
        read carries the count of singular LS systems of that adaptation; like the reference, whose inv() raises
        there (functions.py:465, :530), the run aborts with numpy.linalg.LinAlgError."""
        p, sh = self.plan, self.shard
        if not sh.collective:
            srer, faults = (float(v) for v in self.sums[3:5].cpu())
        else:
            red = self.sums[[0, 1, 4]]
            sh.all_reduce_sum(red)
            tot, tot2, faults = (float(v) for v in red.cpu())
            n = float(p.L)
            mean = tot / n
            srer = float(20.0 * np.log10(self.std_det / np.sqrt(tot2 / n - mean * mean)))
        if faults > 0:
            raise np.linalg.LinAlgError("Singular matrix (%d frame(s) of this adaptation: a collapsed Cholesky pivot "
                                        "in the normal equations)" % int(faults))
        if not np.isfinite(srer):
            raise FloatingPointError("SRER of the adaptation is not finite (%r)" % srer)
        return srer

    def post_stage(self, a):
        """Interpolation + synthesis + SRER of adaptation `a`; returns the SRER (host float)."""
        self.post_launch(a)
        return self.post_result()

    def reset(self, keep_timeline=False):
        """Forget the previous run (the buffers are rewritten by the next one)."""
        self.SRER = []
        self.n_ls_frames = 0
        self.seeded_on_break = None
        if not keep_timeline:
            self.timeline = []
            self.ncol_hist = []

    def adaptations(self, on_adaptation=None):
        """functions.py:163-402 as a generator: yields once per adaptation, right after its kernels have been
        enqueued and before the host reads its SRER, so that a caller driving several analyses (run_interleaved)
        can enqueue the others' work before this one blocks."""
        if hasattr(self.ctx, "bind_stream"):
            self.ctx.bind_stream()
        for a in range(self.max_adpt + 1):
            self.ls_stage(a)
            self.post_launch(a)
            yield a
            if hasattr(self.ctx, "bind_stream"):
                self.ctx.bind_stream()
            srer = self.post_result()
            self.SRER.append(np.float64(srer))
            if on_adaptation is not None:
                on_adaptation(a, self)
            if a != 0 and self.SRER[a] <= self.SRER[a - 1]:                      # functions.py:394-396
                # Q8: the empty-row seeding of THIS adaptation wrote into the array the previous
                # adaptation's result aliases (functions.py:210, :383, :400)
                flag = self.any_seed.clone()
                self.shard.all_reduce_sum(flag)
                if int(flag.item()):
                    seeded = self.seeded.to(self.torch.int32)
                    self.shard.all_reduce_sum(seeded)
                    self.seeded_on_break = self.torch.nonzero(seeded).flatten().cpu().numpy()
                break
            # accept: functions.py:397-402
            for buf in (self.records, self.ph_knot, self.s_hat):
                buf[0], buf[1] = buf[1], buf[0]

    def run(self, on_adaptation=None):
        """Runs the adaptation loop to its end.  Returns the number of executed adaptations."""
        for _ in self.adaptations(on_adaptation):
            pass
        return len(self.SRER)

    # ------------------------------------------------------------------ results
    def final_arrays(self):
        """Host copies of what functions.py:404-411 packs: a0, am, fm, phase at every instant of the
        last accepted adaptation, and s_recon.  (With several ranks each holds its own time range of
        s_recon and its own instants of the phases; they are merged here, once, outside the loop.)"""
        p, K, sh = self.plan, self.plan.Kmax, self.shard
        s_hat, ph = self.s_hat[1], self.ph_knot[1]
        if sh.collective and self.partial_rows:      # only boundary rows travelled during the loop
            sh.all_gather_rows(self.records[1], p.No_ti)
        if sh.collective:
            s_hat = s_hat.clone()
            s_hat[:self.s_lo] = 0
            s_hat[self.s_hi:] = 0
            ph = ph.clone()
            ph[:self.i_lo] = 0
            ph[self.i_hi:] = 0
            sh.all_reduce_sum(s_hat)
            sh.all_reduce_sum(ph)
        rec = self.records[1][:p.No_ti].cpu().numpy()
        out = dict(a0=rec[:, 3 * K].copy(), am=rec[:, :K].copy(), fm=rec[:, K:2 * K].copy(),
                   pk=ph.cpu().numpy(), s_recon=s_hat.cpu().numpy())
        if self.seeded_on_break is not None and len(self.seeded_on_break):
            inst = self.seeded_on_break // p.step
            out["am"][inst, 0] = 10e-4
        return out

    def stage_times_ms(self):
        """Per-stage GPU time from the recorded events (profile=True)."""
        self.torch.cuda.synchronize(self.ctx.device)
        acc = {}
        for a, stage, e0, e1 in self.timeline:
            acc.setdefault(stage, []).append(e0.elapsed_time(e1))
        return acc


def run_interleaved(engines, on_adaptation=None):
    """Batch operation: the adaptation loops of several independent analyses on ONE GPU, each on
    its own HIP stream, advanced round-robin.  While the host waits for the SRER of one analysis the kernels of
    the others are already queued, so launch tails and the per-adaptation host round trip of one file are filled
    with the frames of the next.  Results are bit-identical to running the engines one after another (same
    kernels, same inputs; the analyses share nothing but the device)."""
    if not engines:
        return
    torch = engines[0].torch
    main = torch.cuda.current_stream(engines[0].ctx.device)
    jobs = []
    for i, e in enumerate(engines):
        st = torch.cuda.Stream(device=e.ctx.device)
        st.wait_stream(main)                      # inputs were uploaded on the current stream
        cb = (lambda a, eng, i=i: on_adaptation(i, a, eng)) if on_adaptation is not None else None
        jobs.append((e, st, e.adaptations(cb)))
    while jobs:
        for job in list(jobs):
            e, st, gen = job
            with torch.cuda.stream(st):
                try:
                    next(gen)
                except StopIteration:
                    jobs.remove(job)
                    main.wait_stream(st)          # final_arrays() reads on the current stream
