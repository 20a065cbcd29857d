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

State in HBM: the frame-centre records [No_ti][3*Kmax+1] for the whole run; am_current / fm_current as
[Kmax][samples] (time contiguous) for the rank's own time range plus halo — or, for long files, one time block at a
time, regenerated from the records (DeviceAnalysis).  All numerics are in libeaqhm_hip.so; this module only
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
    """One rank's view of a run sharded over `world` processes (`group` = a torch.distributed process group, or None
    for a single process) and the collectives the adaptation loop needs.  Stateless about the ranges: `balance`
    computes the instant ranges of a frame plan — every rank derives the same bounds from the same plan, so no
    communication is needed to agree on them — and the engine that asked keeps them and hands them back to the
    collectives (one Sharding may serve several engines, engine.run_interleaved)."""

    def __init__(self, rank=0, world=1, group=None):
        self.rank, self.world, self.group = int(rank), int(world), group
        self.collective = group is not None      # a 1-rank group still goes through the collectives (rehearsal)

    def balance(self, n_instants, cost=None):
        """Contiguous instant ranges, bounds[r] .. bounds[r+1] for rank r, balanced by the LS cost of their frames
        ("balanced by sum F(N,Kc), not by count": the cost of a frame varies ~2.5x with the pitch).
        `cost` = per-instant LS cost (zero for the instants that are not analysed); None or all-zero -> equal counts."""
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
        return b

    def instants(self, n_instants, rank=None):
        """Equal-count range of `rank` (what balance gives without costs)."""
        b = self.balance(n_instants)
        r = self.rank if rank is None else rank
        return b[r], b[r + 1]

    def _check(self, bounds, n_rows):
        if bounds is None or len(bounds) != self.world + 1 or bounds[0] != 0 or bounds[-1] != n_rows:
            raise ValueError("instant ranges %r do not belong to a buffer of %d rows on %d ranks" % (bounds, n_rows, self.world))

    def _all_gather_into(self, out, part):
        """out = concatenation over ranks of `part` (RCCL all-gather of equal parts)."""
        import torch.distributed as dist
        dist.all_gather_into_tensor(out, part, group=self.group)

    def all_gather_rows(self, buf, bounds):
        """Every rank ends up with every rank's rows of `buf` (rank r owns rows bounds[r] .. bounds[r+1]).  The ranges
        differ in length, so this is a sum over ranks of the buffer with the foreign rows zeroed — done once per run,
        when the results are collected, or per adaptation only for inputs too short for the boundary exchange."""
        if not self.collective:
            return
        self._check(bounds, buf.shape[0])
        lo, hi = bounds[self.rank], bounds[self.rank + 1]
        buf[:lo].zero_()
        buf[hi:].zero_()
        self.all_reduce_sum(buf)

    def share_rows(self, buf, bounds, margin):
        """What the interpolation of one rank's time range reads from the other ranks' rows of `buf`: the `margin`
        rows on either side of its own range and the first rows of the file (pad knots of short runs).  Every rank
        contributes the first and the last `margin` rows of its range to one small all-gather and copies its two
        neighbours' parts into place; the full rows are gathered once, when the results are collected
        (all_gather_rows).  Falls back to the full exchange when some range is shorter than the margin.
        Returns True if only boundary rows were exchanged."""
        if not self.collective:
            return False
        self._check(bounds, buf.shape[0])
        b = bounds
        if margin < 4 or min(b[r + 1] - b[r] for r in range(self.world)) < margin:
            self.all_gather_rows(buf, bounds)
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


class StalledPipeline(RuntimeError):
    """A diagonal-tile pipeline of the LS kernels timed out waiting for its partner wave (eaqhm_ls_chol.h, spin_until):
    a defect of the library, never a property of the input — reported apart from singular systems."""


def srer_from_limbs(limbs, n, std_det):
    """SRER in dB (functions.py:388: 20 log10(std(target) / std(target - s_recon)), population std) from the
    fixed-point error sums of eaqhm_eval_synth (include/eaqhm_hip.h): three base-2^32 limbs of sum(d * 2^60), three of
    sum(d^2 * 2^64), the count of non-finite (or absurdly large) samples.  Integer sums: the same value whatever
    the number of ranks, blocks or time blocks that contributed.  NumPy semantics for the degenerate cases like the
    reference's: a non-finite reconstruction gives nan, a perfect one +inf, a silent target -inf or nan."""
    v = [int(x) for x in limbs]
    if v[6]:
        return np.float64(np.nan)
    tot = (v[0] + (v[1] << 32) + (v[2] << 64)) / (1 << 60)          # int / int: correctly rounded
    tot2 = (v[3] + (v[4] << 32) + (v[5] << 64)) / (1 << 64)
    with np.errstate(all="ignore"):
        mean = np.float64(tot) / np.float64(n)
        var = np.float64(tot2) / np.float64(n) - mean * mean
        return np.float64(20.0) * np.log10(np.float64(std_det) / np.sqrt(var))


def auto_track_budget(resident_bytes, free_bytes):
    """The `track_budget_bytes="auto"` rule: keep the dense tracks resident (None) while they take less than 40 % of the
    free device memory — the signal, the reconstructions, the records and the library's scratch need room too —,
    otherwise stream them in time blocks under 15 % of it, at least 256 MiB, at most 4 GiB (beyond that more blocks are
    no slower: DESIGN.md section 6.2)."""
    if free_bytes is None or resident_bytes <= 0.4 * free_bytes:
        return None
    return int(min(max(0.15 * free_bytes, 256 * 2 ** 20), 4 * 2 ** 30))


class DeviceAnalysis:
    """Buffers + launch sequence of one analysis run on one GPU (rank).

    Dense tracks.  Only two of the reference's seven (L, Kmax) arrays are ever read again (am_current / fm_current,
    functions.py:159-160), and only inside the analysis windows of the frames.  A rank keeps them for ITS time range
    plus a halo of max(wl) samples, [Kmax][t_hi - t_lo].  With `track_budget_bytes` set (long files)
    they are not kept at all: the frame-centre records are the state that survives an adaptation (1/15 of the
    samples), and the frames are worked off in TIME BLOCKS — per block the tracks of the block's samples (+- max(wl))
    are regenerated from the previous adaptation's records into a block-sized buffer (eaqhm_eval_synth without
    synthesis outputs), then the block's LS frames run.  The interpolation is a pure function of the records, so the
    blocks see bit for bit the tracks the resident run sees (tests/test_gpu_fullsize.py)."""

    TRACK_BYTES_PER_CELL = 8 + 8 + 2      # am, fm, and the library's zero counts (u16) per (slot, sample)

    def __init__(self, s, target, plan, f0min, max_adpt, device_index=0, shard=None, keep_raw=False, ctx=None,
                 track_budget_bytes=None):
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
        self.bounds = sh.balance(T, cost)
        self.i_lo, self.i_hi = self.bounds[sh.rank], self.bounds[sh.rank + 1]
        mine = np.flatnonzero((p.frame_inst >= self.i_lo) & (p.frame_inst < self.i_hi))
        lo, hi = (int(mine[0]), int(mine[-1]) + 1) if len(mine) else (0, 0)
        self.f_lo, self.f_hi, self.nf = lo, hi, hi - lo
        # exclusive time ranges partition [0, L): boundary r = first sample of rank r's first instant
        def bound(r):
            if r <= 0:
                return 0
            if r >= sh.world or self.bounds[r] >= T:
                return L
            return self.bounds[r] * p.step
        self.s_lo, self.s_hi = bound(sh.rank), bound(sh.rank + 1)
        # samples whose tracks this rank's frames look at: every window [c - wl, c + wl] and the sample before it
        # (the zero counts of the LS kernels are differences of running counts)
        self.halo = p.wl_max + 1
        self.t_lo = max(0, self.s_lo - self.halo)
        self.t_hi = min(L, self.s_hi + self.halo)
        # rows of the other ranks' records the interpolation of [t_lo, t_hi) reads: the halo, the spline range beyond it
        # (+-2, +4), run detection (+-41)
        self.margin = -(-self.halo // p.step) + 48
        self.partial_rows = False
        # instants whose run codes / spline moments the evaluation of [t_lo, t_hi) looks at
        self.sp_lo = max(0, (max(self.t_lo, 1) - 1) // p.step - 2)
        self.sp_hi = max(self.sp_lo + 1, min(T, (max(self.t_hi, 1) - 1) // p.step + 4))

        # frame tables: this rank's frames, preceded by the frames just before its first one whose empty-row seeding
        # (functions.py:204-210) would be visible inside its windows (frame_prep visits those too)
        ext = lo
        if self.nf:
            ext = int(np.searchsorted(p.frame_c, p.frame_c[lo] - p.wl_max, side="left"))
        self.ext0 = ext

        def dev_i32(a):
            return torch.as_tensor(np.ascontiguousarray(a[ext:hi]), dtype=i32, device=dev)

        self._inst_x, self._c_x, self._wl_x, self._K_x = (dev_i32(p.frame_inst), dev_i32(p.frame_c), dev_i32(p.frame_wl),
                                                           dev_i32(p.frame_K))
        self._f0_x = torch.as_tensor(np.ascontiguousarray(p.frame_f0[ext:hi]), dtype=f64, device=dev)
        n_x = max(hi - ext, 1)
        self.ncol_ext = torch.zeros(n_x, dtype=i32, device=dev)
        self.cols_ext = torch.zeros(n_x * K, dtype=i32, device=dev)
        e0 = lo - ext
        self.frame_inst, self.frame_c, self.frame_wl = self._inst_x[e0:], self._c_x[e0:], self._wl_x[e0:]
        self.frame_K, self.frame_f0 = self._K_x[e0:], self._f0_x[e0:]
        self.ncol = self.ncol_ext[e0:]
        self.cols = self.cols_ext[e0 * K:]
        self.seeded = torch.zeros(L, dtype=torch.uint8, device=dev)
        self.any_seed = torch.zeros(1, dtype=i32, device=dev)

        # time blocks and the dense tracks (functions.py:159-160) — harmonic-major, [Kmax][samples of the window]
        if isinstance(track_budget_bytes, str):
            if track_budget_bytes != "auto":
                raise ValueError("track_budget_bytes: a byte count, None or 'auto'")
            free = torch.cuda.mem_get_info(dev)[0] if dev.type == "cuda" else None
            track_budget_bytes = auto_track_budget(self.TRACK_BYTES_PER_CELL * K * max(self.t_hi - self.t_lo, 0), free)
        self.blocks = self._plan_blocks(track_budget_bytes)
        self.streaming = track_budget_bytes is not None
        if self.streaming:
            self.seeded_all = torch.zeros(L, dtype=torch.uint8, device=dev)
            self.any_seed_all = torch.zeros(1, dtype=i32, device=dev)
        width = max([b[3] - b[2] for b in self.blocks] + [1])
        self.track_cells = K * width
        self._trk_am = torch.zeros(K * width, dtype=f64, device=dev)
        self._trk_fm = torch.zeros(K * width, dtype=f64, device=dev)
        self.track_t0, self.track_len = (self.blocks[0][2], self.blocks[0][3] - self.blocks[0][2]) if self.blocks else (0, 1)
        # frame-centre records, double-buffered: [0] = adaptation in flight, [1] = last accepted
        self.records = [torch.zeros(T, self.RS, dtype=f64, device=dev) for _ in range(2)]
        self.ph_knot = [torch.zeros(T, K, dtype=f64, device=dev) for _ in range(2)]
        self.s_hat = [torch.zeros(L, dtype=f64, device=dev) for _ in range(2)]
        self.code = torch.zeros(T, K, dtype=torch.uint8, device=dev)
        self.mom = torch.zeros(T, K + 1, dtype=f64, device=dev)
        self.partials = torch.zeros(self.ctx.eval_partials_len(0, L, p.step), dtype=f64, device=dev)
        # {sum d, sum d^2, n, SRER dB, LS breakdowns, stalled pipelines, -, -} + 8 int64 limbs (include/eaqhm_hip.h)
        self.sums = torch.zeros(16, dtype=f64, device=dev)
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

    # ------------------------------------------------------------------ tracks and time blocks
    def _plan_blocks(self, budget):
        """[(first frame, end frame, first sample, end sample)] in this rank's frame numbering: runs of consecutive frames
        whose windows (+ the sample before) fit a track buffer of `budget` bytes; one block = the rank's whole range when
        there is no budget."""
        p = self.plan
        if self.nf == 0:
            return []
        if budget is None:
            return [(0, self.nf, self.t_lo, self.t_hi)]
        c = p.frame_c[self.f_lo:self.f_hi].astype(np.int64)
        width = int(budget) // (self.TRACK_BYTES_PER_CELL * p.Kmax)
        need = 2 * self.halo + 1
        if width < need:
            raise ValueError("track_budget_bytes=%d holds %d samples of %d slots; one analysis window needs %d"
                             % (budget, width, p.Kmax, need))
        out, fa = [], 0
        while fa < self.nf:
            w_lo = max(0, int(c[fa]) - self.halo)
            fb = int(np.searchsorted(c, w_lo + width - self.halo, side="left"))     # c + halo <= w_lo + width
            fb = max(fb, fa + 1)
            out.append((fa, min(fb, self.nf), w_lo, min(p.L, int(c[min(fb, self.nf) - 1]) + self.halo)))
            fa = min(fb, self.nf)
        return out

    def _tracks(self, blk):
        """(am, fm) views [Kmax][width] of the track buffer for a block's window, and the window."""
        K = self.plan.Kmax
        w = blk[3] - blk[2]
        return self._trk_am[:K * w].view(K, w), self._trk_fm[:K * w].view(K, w), blk[2], w

    @property
    def am_cur(self):
        """am_current of the resident window (functions.py:383) as [Kmax][samples]; absolute sample t is column
        t - track_t0.  (Streaming runs hold only the block worked on last.)"""
        return self._trk_am[:self.plan.Kmax * self.track_len].view(self.plan.Kmax, self.track_len)

    @property
    def fm_cur(self):
        return self._trk_fm[:self.plan.Kmax * self.track_len].view(self.plan.Kmax, self.track_len)

    def track_bytes(self):
        """Bytes of dense-track state this engine holds (both arrays); the library adds 2 bytes per cell of zero counts."""
        return 2 * 8 * self.track_cells

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
        raw_a, raw_s = self.raw if self.raw is not None else (None, None)
        K = p.Kmax
        if self.streaming and a > 0:
            self.seeded_all.zero_()
            self.any_seed_all.zero_()
        e0 = self._mark()
        for blk in (self.blocks if a > 0 else [(0, self.nf, 0, p.L)]):
            fa, fb = blk[0], blk[1]
            am = fm = None
            t0, w = 0, p.L
            if a > 0:
                am, fm, t0, w = self._tracks(blk)
                self.track_t0, self.track_len = t0, w
                if self.streaming:
                    # the block's tracks from the records of the last accepted adaptation (the interpolation and phase
                    # integration of functions.py:346-375 again, for these samples only; code / mom are still its)
                    c.eval_synth(self.records[1], self.code, self.mom, p.No_ti, K, p.step, p.fs, p.L, t0, t0 + w, 0, 0,
                                 None, 0.0, am, fm, t0, w, None, None, None, None)
                # frame_prep also visits the frames before the block whose seeded rows its windows can see
                xa = int(np.searchsorted(p.frame_c, p.frame_c[self.f_lo + fa] - p.wl_max, side="left")) - self.ext0
                xb = self.f_lo + fb - self.ext0
                c.frame_prep(fm, p.L, t0, w, K, self._c_x[xa:xb], xb - xa, self.ncol_ext[xa:xb],
                             self.cols_ext[xa * K:xb * K], self.seeded, self.any_seed)
                if self.streaming:
                    self.seeded_all |= self.seeded
                    self.any_seed_all |= self.any_seed
            c.ls_batch(0 if a == 0 else 1, self.s, p.L, p.fs, am, fm, t0, w, K, self.frame_inst[fa:fb],
                       self.frame_c[fa:fb], self.frame_wl[fa:fb], self.frame_f0[fa:fb], self.frame_K[fa:fb],
                       self.ncol[fa:fb], self.cols[fa * K:fb * K], self.seeded, self.any_seed, fb - fa, p.wl_max, a,
                       p.f0_stale, self.f0min, self.records[0],
                       None if raw_a is None else raw_a[fa:fb], None if raw_s is None else raw_s[fa:fb])
        e1 = self._mark()
        if self.profile:
            if a > 0:
                self.ncol_hist.append(self.ncol.clone())
            self.timeline.append((a, "ls", e0, e1))
        self.n_ls_frames += self.nf

    def post_launch(self, a):
        """Enqueue interpolation + synthesis + error sums of adaptation `a` (no host read)."""
        p, c, sh = self.plan, self.ctx, self.shard
        g0 = self._mark()
        self.partial_rows = sh.share_rows(self.records[0], self.bounds, self.margin) or self.partial_rows
        e0 = self._mark()
        if self.profile and sh.collective:
            self.timeline.append((a, "gather", g0, e0))
        c.spline_solve(self.records[0], p.No_ti, p.Kmax, p.step, self.code, self.mom, self.sp_lo, self.sp_hi)
        if self.s_hi > self.s_lo:
            if self.streaming or not self.blocks:      # synthesis and error sums only: the tracks are made per block
                c.eval_synth(self.records[0], self.code, self.mom, p.No_ti, p.Kmax, p.step, p.fs, p.L,
                             self.s_lo, self.s_hi, self.s_lo, self.s_hi, self.target, self.std_det, None, None, 0, 0,
                             self.ph_knot[0], self.s_hat[0], self.partials, self.sums)
            else:
                am, fm, t0, w = self._tracks(self.blocks[0])
                c.eval_synth(self.records[0], self.code, self.mom, p.No_ti, p.Kmax, p.step, p.fs, p.L,
                             self.t_lo, self.t_hi, self.s_lo, self.s_hi, self.target, self.std_det, am, fm, t0, w,
                             self.ph_knot[0], self.s_hat[0], self.partials, self.sums)
        else:
            self.sums.zero_()
        e1 = self._mark()
        if self.profile:
            self.timeline.append((a, "post", e0, e1))

    def post_result(self):
        """SRER of the adaptation enqueued last (numpy.float64): the one device->host read of an adaptation.  One
        formula for every world size and block count: the ranks' fixed-point error sums are added as integers
        (srer_from_limbs).  The same read carries the count of LS systems whose factorisation broke down in that
        adaptation; like the reference, whose inv() raises on a singular matrix (functions.py:465, :530), the run
        aborts with numpy.linalg.LinAlgError."""
        p, sh = self.plan, self.shard
        words = self.sums.view(self.torch.int64)[8:16].clone()
        words[7] = 0
        counts = self.sums[4:7].to(self.torch.int64)
        red = self.torch.cat((words, counts))
        sh.all_reduce_sum(red)
        red = red.cpu().numpy()
        faults, stalled, dropped = int(red[8]), int(red[9]), int(red[10])
        if dropped > 0:
            raise ValueError("%d frame(s) had their analysis window outside the resident track window (engine defect or a "
                             "direct caller of eaqhm_ls_batch broke its contract); they were not analysed" % dropped)
        if stalled > 0:
            raise StalledPipeline("%d diagonal-tile pipeline(s) of the LS kernels timed out in this adaptation "
                                  "(library defect, not a singular system)" % stalled)
        if faults > 0:
            raise np.linalg.LinAlgError("Singular matrix (%d frame(s) of this adaptation: Cholesky breakdown "
                                        "of the normal equations)" % faults)
        return srer_from_limbs(red[:8], p.L, self.std_det)

    def post_stage(self, a):
        """Interpolation + synthesis + SRER of adaptation `a`; returns the SRER."""
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
            # functions.py:394-396.  A nan SRER compares False (the loop goes on, as in the reference), inf <= inf breaks.
            if a != 0 and self.SRER[a] <= self.SRER[a - 1]:
                # Q8: the empty-row seeding of THIS adaptation wrote into the array the previous
                # adaptation's result aliases (functions.py:210, :383, :400)
                flag = (self.any_seed_all if self.streaming else self.any_seed).clone()
                self.shard.all_reduce_sum(flag)
                if int(flag.item()):
                    seeded = (self.seeded_all if self.streaming else self.seeded).to(self.torch.int32)
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
            sh.all_gather_rows(self.records[1], self.bounds)
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
