"""ORACLE — CPU (NumPy/SciPy) restatement of the reference's eaQHM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under `oracle/` is part of the product: only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may
import it, and only as the checker / the reported CPU baseline.  The product path
(`eaqhm-analysis-and-synthesis-in-python_amd/`) never imports this module and fails
loudly when its HIP library is missing.

What it restates (all citations are into /root/reference, read as text):
  * functions.py:86-161   driver prologue (signal scaling, pitch limits, Kmax, VUV
                          override, analysis instants, dense state)
  * functions.py:163-402  adaptation loop: frame set-up (182-197), track gather +
                          gap fill (199-292), LS calls (196, 295), frequency
                          correction (297), slot scatter (299-301), acceptance and
                          frame-centre writes (303-325), a0 spline (340), per-harmonic
                          segmentation / am / fm interpolation (346-371), phase
                          integration (373, 537-575), next-iteration frequency (375),
                          synthesis + SRER + stop rule (383-402)
  * functions.py:404-411  result packing (returned here as plain arrays)
  * functions.py:420-470  iqhmLS_complexamps, :472-535 eaqhmLS_complexamps
  * functions.py:577-680  voicedUnvoicedFrames, getLinear; misc.py:167-206
It is written from the algorithm, vectorised where that does not change the
arithmetic, and is NOT a copy of the reference's code.

PINNING.  The oracle is pinned against outputs of the reference itself, generated in
the build container by `tests/golden/make_golden.py` (which imports the reference
unmodified) and committed under `tests/golden/*.npz`: the six SRER values of the
reference's README screenshot (img/SA19out.JPG), per-frame LS inputs/outputs,
frame-centre records after adaptations 0 and 1, dense track slices, the final
`Deterministic` contents and `s_recon`.  `tests/test_oracle_golden.py` checks all of
them without a GPU.
"""
from __future__ import annotations

import numpy as np
from numpy.linalg import inv
from scipy.interpolate import make_interp_spline
from scipy.signal import ellip, filtfilt

NORMALIZE = 32768          # misc.py:13
MIN_INTERP_SIZE = 4        # misc.py:14
EPS_AM = 10e-5             # functions.py:517


# --------------------------------------------------------------------------- prologue pieces
def pitch_limits(gender):
    """functions.py:95-109."""
    if isinstance(gender, tuple):
        return gender[0], gender[1]
    return {"male": (70, 180), "female": (160, 300), "child": (300, 600)}.get(gender, (70, 500))


def get_linear(v, t):
    """functions.py:644-680 getLinear for an array of query times: linear interpolation of every
    column of `v` (first column = time), output column 0 = query time."""
    v = np.asarray(v, dtype=np.float64)
    t = np.asarray(t, dtype=np.float64)
    times = v[:, 0]
    out = np.ones((len(t), v.shape[1]))
    out[:, 0] = t
    prev = np.searchsorted(times, t, side="right") - 1          # last index with times <= t
    for j in range(len(t)):
        p = prev[j]
        if p < 0:                                               # before the first sample (:658-662)
            out[j, 1:] = v[0, 1:]
        elif times[p] == t[j] or p == len(v) - 1:
            # exact hit (:667-670).  (p == len(v)-1 with t beyond the track raises IndexError in the
            # reference at :674; it cannot happen for the 5 ms grid the driver asks for.)
            if times[p] != t[j]:
                raise IndexError("query time beyond the pitch track")
            out[j, 1:] = v[p, 1:]
        else:
            g = (t[j] - times[p]) / (times[p + 1] - times[p])
            if g < 0 or g > 1:
                raise ValueError("linearity factor unbound, g not in [0, 1]")
            out[j, 1:] = v[p, 1:] * (1 - g) + v[p + 1, 1:] * g
    return out


def ellip_filter(s, fs, fc, ftype="highpass"):
    """misc.py:167-182."""
    b, a = ellip(6, 0.5, 60, 2 * fc / fs, ftype)
    return filtfilt(b, a, s)


def medfilt_ref(x, p=5):
    """misc.py:184-206 as written.  With xp = x padded by (p-1)/2 edge copies on both sides, the
    matrix built there is fliplr(toeplitz(flipud(xp[0:L]), xp[L:L+p-1])): it has p-1 columns (an
    even-length 'median' = mean of the two middle values), row i holds xp[L-1-i+j] for j <= i and
    xp[L+j-i] for j > i — i.e. the output runs BACKWARDS in time relative to the input and the first
    rows skip xp[L].  Returns floats in {0, 0.5, 1} for boolean input."""
    x = np.asarray(x, dtype=np.float64)
    L = len(x)
    ad = (p - 1) // 2
    xp = np.concatenate((np.full(ad, x[0]), x, np.full(ad, x[-1])))
    i = np.arange(L)[:, None]
    j = np.arange(p - 1)[None, :]
    idx = np.where(j <= i, L - 1 - i + j, L + j - i)
    rows = np.sort(xp[idx], axis=1)
    h = (p - 1) // 2
    return 0.5 * (rows[:, h - 1] + rows[:, h])


def voiced_unvoiced_frames(s, fs, gender):
    """functions.py:577-642.  Returns (ti5, isSpeech, isVoiced, frame_step); flags are floats in
    {0, 0.5, 1} exactly as the reference's medfilt produces them."""
    s = ellip_filter(np.asarray(s, dtype=np.float64), fs, 30)
    L = len(s)
    s_smooth = ellip_filter(s, fs, 1000 if gender == "male" else 1500, "lowpass")
    wlen = int(round(0.03 * fs))
    if wlen % 2 == 0:
        wlen += 1
    hop = int(round(0.005 * fs))
    mid = (wlen - 1) // 2
    ti = np.arange(1, L, hop)
    sp = np.zeros(len(ti), dtype=bool)
    vo = np.zeros(len(ti), dtype=bool)
    for i, t in enumerate(ti):
        if mid < t < L - mid:
            seg = slice(t - mid - 1, t + mid)
            e = 20 * np.log10(np.std(s[seg]))
            es = 20 * np.log10(np.std(s_smooth[seg]))
            sp[i] = e > -60
            if sp[i]:
                vo[i] = (e - es < 10) and (es > -50)
    return ti, medfilt_ref(sp, 5), medfilt_ref(vo, 5), int(ti[1] - ti[0])


def full_waveform_override(ti5, is_speech, is_voiced, L, analysis_window_samples):
    """functions.py:139-146: with fullWaveform every 5 ms frame away from the edges becomes voiced."""
    sp = np.array(is_speech, dtype=np.float64)
    vo = np.array(is_voiced, dtype=np.float64)
    h = analysis_window_samples / 2
    inb = (ti5 > h) & (ti5 < L - h)
    c1 = inb & (sp != 0) & (vo == 0)
    vo[c1] = 1.0
    c2 = inb & (sp == 0) & (vo == 0)
    sp[c2] = 1.0
    vo[c2] = 1.0
    return sp, vo


def voiced_only_target(s, ti5, is_speech, is_voiced, frame_step):
    """functions.py:127-138 (fullWaveform=False): SRER target = s inside runs of speech&voiced 5 ms
    frames, each run extended by one frame step on both sides; a run still open at the end of the
    list is dropped, exactly as the reference's loop does."""
    ss = np.zeros_like(s)
    run = []
    for t, sp, vo in zip(ti5, is_speech, is_voiced):
        if sp and vo:
            run.append(int(t))
        elif run:
            ss[run[0] - frame_step:run[-1] + frame_step + 1] = s[run[0] - frame_step:run[-1] + frame_step + 1]
            run = []
    return ss


# --------------------------------------------------------------------------- LS seams
def iqhm_ls(s, f0range, window, fs):
    """functions.py:420-470: stationary-harmonic QHM least squares (weight = window applied to both
    basis and signal, i.e. w^2)."""
    s = np.asarray(s, dtype=np.float64).ravel()
    N = len(s)
    mid = (N - 1) / 2
    n = np.arange(-mid, mid + 1)[:, None]
    t = (n * 2 * np.pi * np.asarray(f0range, dtype=np.float64)[None, :]) / fs
    E0 = np.cos(t) + 1j * np.sin(t)
    return _weighted_ls(E0, n, s, window)


def eaqhm_ls(s, am, fm, window, fs):
    """functions.py:472-535: adaptive AM-FM basis; phase = running sum of fm re-centred at the
    window middle, amplitude ratio (am+eps)/(am_mid+eps)."""
    s = np.asarray(s, dtype=np.float64).ravel()
    N = fm.shape[0]
    mid = (N - 1) // 2
    n = np.arange(-mid, mid + 1, dtype=np.float64)[:, None]
    f_an = np.cumsum(fm, axis=0)
    f_an = f_an - f_an[mid]
    t = (2 * np.pi * f_an) / fs
    E2 = (EPS_AM + am) / (am[mid][None, :] + EPS_AM) * (np.cos(t) + 1j * np.sin(t))
    return _weighted_ls(E2, n, s, window)


def _weighted_ls(E0, n, s, window):
    w = np.asarray(window, dtype=np.float64)[:, None]
    Ew = w * np.concatenate((E0, n * E0), axis=1)
    EwH = Ew.conj().T
    R = EwH @ Ew
    rhs = EwH @ (w[:, 0] * s)
    x = inv(R) @ rhs                       # functions.py:465 / :530 use inv(), not a solve
    K = E0.shape[1]
    return x[:K], x[K:]


# --------------------------------------------------------------------------- interpolation pieces
def phase_integr_interpolation(omega, ph, knots):
    """functions.py:537-575 for knots with arbitrary spacing (looped) — the small-case form."""
    out = np.zeros(len(omega))
    for i in range(len(knots) - 1):
        i0, i1 = int(knots[i]), int(knots[i + 1])
        p = np.cumsum(omega[i0:i1 + 1])
        p = p + (ph[i0] - p[0])
        M = round((p[-1] - ph[i1]) / (2 * np.pi))
        er = np.pi * (p[-1] - ph[i1] - 2 * np.pi * M) / (2 * (i1 - i0))
        tt = np.arange(0, i1 - i0 + 1)
        p = p - np.cumsum(np.sin(np.pi * tt / (i1 - i0)) * er)
        out[i0:i1 + 1] = p
    return out[int(knots[0]):int(knots[-1]) + 1]


def _phase_integr_uniform(omega_col, ph_col, knots, step):
    """Same arithmetic as `phase_integr_interpolation`, all knot intervals at once (they all have
    length `step`); returns the dense phase on knots[0]..knots[-1]."""
    i0 = knots[:-1]
    idx = i0[:, None] + np.arange(step + 1)[None, :]
    p = np.cumsum(omega_col[idx], axis=1)
    p = p + (ph_col[i0] - p[:, 0])[:, None]
    e = p[:, -1] - ph_col[knots[1:]]
    M = np.round(e / (2 * np.pi))          # round-half-even like Python's round()
    er = np.pi * (e - 2 * np.pi * M) / (2 * step)
    ft = np.sin(np.pi * np.arange(step + 1) / step)
    p = p - np.cumsum(ft[None, :] * er[:, None], axis=1)
    out = np.empty(knots[-1] - knots[0] + 1)
    out[(idx[:, :-1] - knots[0]).ravel()] = p[:, :-1].ravel()
    out[-1] = p[-1, -1]                     # the last knot keeps the integrated value (Q6)
    return out


def _cubic(x, y, xq, extrapolate=False):
    """interp1d(kind=3) == make_interp_spline(k=3) (not-a-knot)."""
    return make_interp_spline(x, y, k=3)(xq, extrapolate=extrapolate)


def interpolate_tracks(a0_c, am_recon, fm_recon, ph_recon, ti, step, fs, L):
    """functions.py:337-383 on dense (L, Kmax) arrays that hold frame-centre values only.
    Modifies am/fm/ph_recon in place (as the reference does) and returns (a0_dense, fm_current)."""
    Kmax = am_recon.shape[1]
    c = ti - 1
    a0 = _cubic(c, a0_c, np.arange(L), extrapolate=True)               # :340
    fm_current = np.zeros((L, Kmax))
    for k in range(Kmax):
        nz = np.flatnonzero(am_recon[:, k])                              # :350
        if len(nz) == 0:
            continue
        d = np.diff(np.concatenate(([0], nz, [L - 1])))                  # :352
        ind = (d <= step).astype(int)
        dd = np.diff(ind)
        starts = np.flatnonzero(dd == 1)
        ends = np.flatnonzero(dd == -1)
        for st, en in zip(starts, ends):                                 # :360
            knots = nz[st:en + 1]
            rng = np.arange(knots[0], knots[-1] + 1)
            am_recon[rng, k] = np.interp(rng, knots, am_recon[knots, k])  # :364 (linear)
            if len(knots) >= MIN_INTERP_SIZE:
                fm_recon[rng, k] = _cubic(knots, fm_recon[knots, k], rng)                  # :367
            else:
                kt = np.concatenate((np.arange(0, (MIN_INTERP_SIZE - len(knots)) * step, step), knots))
                fm_recon[rng, k] = _cubic(kt, fm_recon[kt, k], rng)                        # :369-371
            if np.all(np.diff(knots) == step):
                ph = _phase_integr_uniform(2 * np.pi / fs * fm_recon[:, k], ph_recon[:, k], knots, step)
            else:  # cannot happen (consecutive instants are `step` apart) but stay literal
                ph = phase_integr_interpolation(2 * np.pi / fs * fm_recon[:, k], ph_recon[:, k], knots)
            ph_recon[rng, k] = ph                                        # :373
            fm_current[rng, k] = np.concatenate(([fm_recon[rng[0], k]],
                                                 fs / (2 * np.pi) * np.diff(np.unwrap(ph))))  # :375
    return a0, fm_current


# --------------------------------------------------------------------------- the adaptation loop
def gather_fill(fm_current, am_current, c, wl, nz):
    """functions.py:244-278: window the tracks of the active slots and bridge zero gaps (positions
    decided on fm, applied to fm and am): interior gaps linearly, edge gaps held."""
    rows = slice(c - wl, c + wl + 1)
    fm = fm_current[rows][:, nz].copy()
    am = am_current[rows][:, nz].copy()
    zero_cols = np.flatnonzero((fm == 0).any(axis=0))
    x = np.arange(fm.shape[0])
    for j in zero_cols:
        kn = np.flatnonzero(fm[:, j])
        fm[:, j] = np.interp(x, kn, fm[kn, j])
        am[:, j] = np.interp(x, kn, am[kn, j])
    return fm, am


class Analysis:
    """State of one run of the adaptation loop (functions.py:115-411) given the pre-processing
    outputs, split into the two stages the GPU build also uses:

      ls_stage(a, instants)   per-frame LS + frequency correction + acceptance  -> frame-centre records
      post_stage(a, records)  interpolation, synthesis, SRER, stop rule

    s        (L,) float64 signal (already /32768 and optionally high-passed)
    f0s      (n5, >=2) 5 ms pitch grid, column 1 = f0 (functions.py:113)
    vuv_*    outputs of voicedUnvoicedFrames BEFORE the fullWaveform override
    """

    def __init__(self, s, fs, f0s, vuv_ti, vuv_speech, vuv_voiced, frame_step, *, f0min, step=15,
                 maxAdpt=10, pitchPeriods=3, analysisWindow=32, fullWaveform=True, partials=0):
        s = np.asarray(s, dtype=np.float64).ravel()
        self.s, self.fs, self.f0min, self.step, self.maxAdpt = s, fs, f0min, step, maxAdpt
        L = self.L = len(s)
        f0s = np.asarray(f0s, dtype=np.float64)
        Fmax = self.Fmax = int(fs / 2 - 200)                                          # :115
        Kmax = self.Kmax = partials if partials > 0 else int(round(Fmax / np.min(f0s[:, 1])) + 10)
        aws = analysisWindow * step                                                   # :123
        if fullWaveform:
            sp, vo = full_waveform_override(vuv_ti, vuv_speech, vuv_voiced, L, aws)
            self.target = s
        else:
            sp, vo = np.array(vuv_speech, dtype=float), np.array(vuv_voiced, dtype=float)
            self.target = voiced_only_target(s, vuv_ti, sp, vo, frame_step)
        ti = self.ti = np.arange(1, L, step)                                          # :148
        No_ti = self.No_ti = len(ti)
        framei = ti / frame_step
        fi = framei.astype(int)
        inb = self.inb = (ti > aws) & (ti < L - aws)                                  # :180
        voiced = inb.copy()
        voiced[inb] = (vo[fi[inb] - 1] != 0) & (vo[fi[inb]] != 0)                      # :181
        self.voiced = voiced
        # adaptation-0 frame set-up (functions.py:183-191), host-side scalars per instant
        dec = framei - fi
        f0 = np.zeros(No_ti)
        v = np.flatnonzero(voiced)
        f0[v] = (1 - dec[v]) * f0s[fi[v] - 1, 1] + dec[v] * f0s[fi[v], 1]             # :185
        self.f0 = f0
        self.K0 = np.zeros(No_ti, dtype=np.int64)
        self.K0[v] = np.minimum(Kmax, (Fmax / f0[v]).astype(int))                      # :187
        self.wls = np.zeros(No_ti, dtype=np.int64)
        self.wls[v] = np.maximum(120, np.round((pitchPeriods / 2) * (fs / f0[v]))).astype(np.int64)  # :191
        self.f0_stale = f0[v[-1]] if len(v) else 0.0        # Q1: f0 of the last frame of adaptation 0
        self.fm_current = np.zeros((L, Kmax))
        self.am_current = np.zeros((L, Kmax))
        self.std_det = np.std(self.target)
        self.SRER = []
        self.fin = None
        self.s_recon = None
        self.n_ls = 0
        self.seeded = []
        self.done = False

    # ---- stage 1
    def ls_stage(self, a, instants=None):
        """Frame-centre records of adaptation `a` for the given instant indices (default: all).
        Returns dict a0 (No_ti,), am/fm/ph (No_ti, Kmax); rows of instants not analysed are zero."""
        s, fs, ti, Kmax = self.s, self.fs, self.ti, self.Kmax
        rec = dict(a0=np.zeros(self.No_ti), am=np.zeros((self.No_ti, Kmax)),
                   fm=np.zeros((self.No_ti, Kmax)), ph=np.zeros((self.No_ti, Kmax)))
        todo = np.flatnonzero(self.voiced)
        want = np.ones(self.No_ti, dtype=bool) if instants is None else np.isin(np.arange(self.No_ti), instants)
        self.seeded = []
        fm_current, am_current = self.fm_current, self.am_current
        for i in todo:
            c = int(ti[i]) - 1
            if a > 0 and not fm_current[c].any():                                    # :204-210 (Q7)
                # the write happens when the sequential loop reaches this frame, whoever analyses it
                fm_current[c, 0] = 140
                am_current[c, 0] = 10e-4
                self.seeded.append(c)
            if not want[i]:
                continue
            self.n_ls += 1
            wl = int(self.wls[i])
            if a == 0:
                f0 = self.f0[i]
                K = int(self.K0[i])
                f0range = np.arange(-K, K + 1) * f0                                  # :189
                amp, _ = iqhm_ls(s[c - wl:c + wl + 1], f0range, np.blackman(2 * wl + 1), fs)
                A = amp[K + 1:]
                eta_pos = np.zeros(K)
                rec["a0"][i] = amp[K].real
            else:
                f0 = self.f0_stale
                nz = np.flatnonzero(fm_current[c])                                   # :202
                K = int(nz[-1]) + 1
                fm, am = gather_fill(fm_current, am_current, c, wl, nz)
                n = len(nz)
                z = np.zeros((fm.shape[0], 1))
                FM = np.concatenate((-fm[::-1], z, fm), axis=1)                      # :284
                AM = np.concatenate((am[::-1], z, am), axis=1)                       # :285
                amp_t, slo_t = eaqhm_ls(s[c - wl:c + wl + 1], AM, FM, np.hamming(2 * wl + 1), fs)
                eta_t = fs / (2 * np.pi) * (amp_t.real * slo_t.imag - amp_t.imag * slo_t.real) / np.abs(amp_t) ** 2
                A = np.zeros(K, dtype=complex)                                       # positive slots only
                eta_pos = np.zeros(K)
                A[nz] = amp_t[n + 1:]
                eta_pos[nz] = eta_t[n + 1:]
                rec["a0"][i] = amp_t[n].real
            absA = np.abs(A)
            with np.errstate(divide="ignore"):
                logA = 20 * np.log10(absA)
            floor = logA.max() - 150                                                 # :309
            h = f0 / (a + 1)                                                         # :310 (stale f0, Q1)
            ka = np.flatnonzero((logA > floor) & (np.abs(eta_pos) < h))              # :315
            rec["am"][i, ka] = absA[ka]
            rec["ph"][i, ka] = np.angle(A[ka])
            if a == 0:
                rec["fm"][i, ka] = (ka + 1) * f0
            elif f0 > self.f0min:
                rec["fm"][i, ka] = fm_current[c, ka] + eta_pos[ka]
            else:
                rec["fm"][i, ka] = fm_current[c, ka]
        return rec

    # ---- stage 2
    def post_stage(self, a, rec):
        """Interpolation + synthesis + SRER + stop rule for adaptation `a` from complete records.
        Returns a dict of the dense state (views, not copies)."""
        L, Kmax, ti = self.L, self.Kmax, self.ti
        c = ti - 1
        am_recon = np.zeros((L, Kmax))
        fm_recon = np.zeros((L, Kmax))
        ph_recon = np.zeros((L, Kmax))
        am_recon[c], fm_recon[c], ph_recon[c] = rec["am"], rec["fm"], rec["ph"]
        a0_dense, fm_next = interpolate_tracks(rec["a0"], am_recon, fm_recon, ph_recon, ti, self.step, self.fs, L)
        self.fm_current = fm_next                                                    # :337, :375
        self.am_current = am_recon                                                   # :383 (alias, Q8)
        s_hat = a0_dense + 2 * (am_recon * np.cos(ph_recon)).sum(axis=1)             # :385
        self.SRER.append(20 * np.log10(self.std_det / np.std(self.target - s_hat)))  # :388
        st = dict(a0=a0_dense, am=am_recon, fm=fm_recon, ph=ph_recon, fm_current=fm_next, s_hat=s_hat,
                  SRER=self.SRER[-1])
        if a != 0 and self.SRER[a] <= self.SRER[a - 1]:                              # :394-396
            self.done = True
            return st
        self.s_recon = s_hat.copy()
        self.fin = (a0_dense, am_recon, fm_recon, ph_recon)
        if a == self.maxAdpt:
            self.done = True
        return st

    def result(self):
        a0_f, am_f, fm_f, pm_f = self.fin
        c = self.ti - 1
        return dict(s_recon=self.s_recon, SRER=list(self.SRER), ti=c, isSpeech=self.inb, isVoiced=self.voiced,
                    Kmax=self.Kmax, a0=a0_f[c], am=am_f[c], fm=fm_f[c], pk=pm_f[c], n_ls_frames=self.n_ls,
                    wls=self.wls)


def analyse(s, fs, f0s, vuv_ti, vuv_speech, vuv_voiced, frame_step, *, on_adaptation=None, **kw):
    """Run the whole adaptation loop; `on_adaptation(a, records, state)` is a test hook."""
    an = Analysis(s, fs, f0s, vuv_ti, vuv_speech, vuv_voiced, frame_step, **kw)
    for a in range(an.maxAdpt + 1):                                                  # :163
        rec = an.ls_stage(a)
        st = an.post_stage(a, rec)
        if on_adaptation is not None:
            on_adaptation(a, rec, st)
        if an.done:
            break
    return an.result()
