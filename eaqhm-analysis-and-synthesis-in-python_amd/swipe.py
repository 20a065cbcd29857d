"""SWIPE' pitch estimation, host-side (NumPy) restatement of the reference's SWIPE.py:14-195.

Runs once per file before the adaptation loop (functions.py:111); it is outside the accelerated hot path
 but its output feeds it, so it reproduces the reference's behaviour,
including the quirks that change numbers:
  * the power spectra come from matplotlib.pyplot.specgram in the reference (SWIPE.py:66: one-sided PSD,
    density scaling 1/(Fs*sum(w^2)), x2 except DC and Nyquist); they are computed here with numpy.fft
    directly — no matplotlib, nothing is drawn;
  * the LAST pitch candidate of each window size is never scored (SWIPE.py:147: range(0, len(pc)-1));
  * 1 counts as a prime in the kernel (SWIPE.py:151-163) and later lobes overwrite instead of accumulate;
  * the frame times handed to the time interpolation are shifted to start at 0 (SWIPE.py:92-97);
  * a strength maximum on either edge of the candidate grid maps to the FIRST candidate (SWIPE.py:118-121).
"""
import numpy as np


def hz2erbs(hz):
    return 21.4 * np.log10(1 + hz / 229)


def erbs2hz(erbs):
    return (np.power(10, erbs / 21.4) - 1) * 229


def _psd_frames(x, nfft, fs, window, noverlap):
    """matplotlib.mlab.specgram(mode='psd', sides='onesided', scale_by_freq=True, detrend none)."""
    step = nfft - noverlap
    nseg = (len(x) - noverlap) // step
    idx = np.arange(nfft)[:, None] + step * np.arange(nseg)[None, :]
    seg = x[idx] * window[:, None]
    spec = np.fft.fft(seg, n=nfft, axis=0)[: nfft // 2 + 1]
    psd = (np.conj(spec) * spec).real
    psd[1:-1] *= 2.0
    psd /= fs
    psd /= (np.abs(window) ** 2).sum()
    freqs = np.arange(nfft // 2 + 1) * (fs / nfft)
    times = np.arange(nfft / 2, len(x) - nfft / 2 + 1, step) / fs
    return psd, freqs, times


def _interp_last_axis(xp, fp, x):
    """scipy.interpolate.interp1d(xp, fp, kind='linear')(x) along the last axis of fp, in-range x only,
    with SciPy's arithmetic: slope * (x - x_lo) + y_lo."""
    hi = np.searchsorted(xp, x)
    hi = np.clip(hi, 1, len(xp) - 1)
    lo = hi - 1
    slope = (fp[..., hi] - fp[..., lo]) / (xp[hi] - xp[lo])
    return slope * (x - xp[lo]) + fp[..., lo]


def _primes_with_one(n):
    """SWIPE.py:161-163: [i for i in range(1, n+2) if is_prime(i)] — 1 passes the trial-division test."""
    out = []
    for i in range(1, n + 2):
        if all(i % d for d in range(2, int(np.sqrt(i)) + 1)):
            out.append(i)
    return out


def _strength_one(f, L, pc):
    """SWIPE.py:166-183."""
    n = int(np.fix(f[-1] / pc - 0.75))
    k = np.zeros(len(f))
    q = f / pc
    for i in _primes_with_one(n):
        a = np.abs(q - i)
        p = a < 0.25
        k[p] = np.cos(2 * np.pi * q[p])
        v = (0.25 < a) & (a < 0.75)
        k[v] = np.cos(2 * np.pi * q[v]) / 2
    k = k * np.sqrt(1.0 / f)
    k = k / np.linalg.norm(k[k > 0.0])
    return k @ L


def _strength_all(f, L, pc):
    """SWIPE.py:135-148 (loudness normalisation; last candidate left at zero)."""
    nrm = np.sqrt((L * L).sum(axis=0))
    nrm = np.where(nrm == 0, np.inf, nrm)
    L = L / nrm[None, :]
    S = np.zeros((len(pc), L.shape[1]))
    for j in range(len(pc) - 1):
        S[j] = _strength_one(f, L, pc[j])
    return S


def swipep(x, fs, plim, speechFile=None):
    """Pitch track of `x` every 1 ms: returns (T, 3) [time s, pitch Hz, strength] like SWIPE.py:14-132."""
    x = np.asarray(x, dtype=np.float64).ravel()
    dt, dlog2p, dERBs = 0.001, 1.0 / 96.0, 0.1
    t = np.arange(0, len(x) / float(fs), dt)
    log2pc = np.arange(np.log2(plim[0]), np.log2(plim[-1]), dlog2p)
    pc = np.power(2, log2pc)
    S = np.zeros((len(pc), len(t)))
    logWs = np.round(np.log2(8 * (float(fs) / np.asarray(plim, dtype=np.float64))))
    ws = np.power(2, np.arange(logWs[0], logWs[1] - 1, -1))
    pO = 8 * fs / ws
    d = 1 + log2pc - np.log2(8 * (fs / ws[0]))
    fERBs = erbs2hz(np.arange(hz2erbs(pc[0] / 4), hz2erbs(fs / 2), dERBs))
    for i in range(len(ws)):
        w_i = int(ws[i])
        dn = int(round(4 * fs / pO[i]))
        xk = np.concatenate((np.zeros(w_i // 2), x, np.zeros(int(dn + w_i / 2))))
        o = max(0, int(round(w_i - dn)))
        X, f, ti = _psd_frames(xk, w_i, fs, np.hanning(w_i), o)
        L = np.sqrt(np.maximum(0, _interp_last_axis(f, X.T, fERBs).T))           # (nERB, nframes)
        if i == len(ws) - 1:
            j = np.flatnonzero(d - (i + 1) > -1)
            k = np.flatnonzero(d[j] - (i + 1) < 0)
        elif i == 0:
            j = np.flatnonzero(d - (i + 1) < 1)
            k = np.flatnonzero(d[j] - (i + 1) > 0)
        else:
            j = np.flatnonzero(np.abs(d - (i + 1)) < 1)
            k = np.arange(len(j))
        Si = _strength_all(fERBs, L, pc[j])
        if Si.shape[1] > 1:
            tshift = np.concatenate(([0.0], ti[:-1]))
            if t[0] < tshift[0] or t[-1] > tshift[-1]:
                raise ValueError("A value in x_new is outside the interpolation range.")   # interp1d bounds_error
            Si = _interp_last_axis(tshift, Si, t)
        else:
            Si = np.full((len(Si), len(t)), np.nan)
        lam = d[j[k]] - (i + 1)
        mu = np.ones(len(j))
        mu[k] = 1 - np.abs(lam)
        S[j, :] = S[j, :] + mu[:, None] * Si
    # Parabolic refinement around the strongest candidate (SWIPE.py:107-131), one 1 ms instant after the other in the
    # reference.  Here the instants that share their strongest candidate i are taken together: numpy.polyfit / polyval
    # on the columns of one 3 x n block give the results of the per-instant calls bit for bit (checked on seven inputs
    # against the per-instant loop, and by tests/test_host_cpu.py against the reference's tracks), the pitch 2 ** (...) stays a scalar operation per distinct (i, k) — NumPy's array power differs from its
    # scalar power in the last bit.  (20 s of speech in the build container: 1.60 s -> 0.18 s for the whole of swipep.)
    p = np.full(len(t), np.nan)
    best = S.argmax(axis=0)
    s = S[best, np.arange(len(t))].copy()
    edge = (best == 0) | (best == len(pc) - 1)
    p[edge] = pc[0]
    for i in np.unique(best[~edge]):
        jt = np.flatnonzero(best == i)
        I = np.arange(i - 1, i + 2)
        tc = 1.0 / pc[I]
        ntc = ((tc / tc[1]) - 1) * 2 * np.pi
        c = np.polyfit(ntc, S[I][:, jt], 2)                       # (3, n): one quadratic per instant
        ftc = 1.0 / np.power(2, np.arange(np.log2(pc[I[0]]), np.log2(pc[I[2]]), 0.0013021))
        nftc = ((ftc / tc[1]) - 1) * 2 * np.pi
        val = np.zeros((len(nftc), len(jt)))
        for ck in c:                                              # numpy.polyval's Horner scheme, column-wise
            val = val * nftc[:, None] + ck[None, :]
        s[jt] = val.max(axis=0)
        k = val.argmax(axis=0)
        for kk in np.unique(k):
            p[jt[k == kk]] = 2 ** (np.log2(pc[I[0]]) + (int(kk) - 1) / 768)
    return np.column_stack((t, p, s))
