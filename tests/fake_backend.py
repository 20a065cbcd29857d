"""TEST-ONLY stand-in for eaqhm_amd.hip.Context: implements the same five calls on CPU tensors with
the oracle, so that the HOST logic of the engine (instant sharding, halo ranges, in-place all-gather,
stop rule, double buffering) can be exercised with the gloo backend and no GPU.  The product never
constructs this class; it lives under tests/ because only tests may touch the oracle."""
import numpy as np
import torch

import eaqhm_oracle as O


class OracleBackend:
    device = torch.device("cpu")

    def eval_partials_len(self, t_lo, t_hi, step):
        return 8

    # ---- functions.py:202-213   (fm_cur: [Kmax][track_len] window that starts at sample track_t0)
    def frame_prep(self, fm_cur, L, track_t0, track_len, Kmax, frame_c, n_frames, ncol, cols, seeded, any_seed):
        fm = fm_cur.numpy()
        seeded.zero_()
        any_seed.zero_()
        for f in range(n_frames):
            c = int(frame_c[f])
            nz = np.flatnonzero(fm[:, c - track_t0])
            if len(nz) == 0:
                seeded[c] = 1
                any_seed[0] = 1
                nz = np.array([0])
            ncol[f] = len(nz)
            cols[f * Kmax:f * Kmax + len(nz)] = torch.as_tensor(nz, dtype=torch.int32)

    # ---- functions.py:187-197 / :244-324
    def ls_batch(self, mode, s, L, fs, am_cur, fm_cur, track_t0, track_len, Kmax, frame_inst, frame_c, frame_wl,
                 frame_f0, frame_K, ncol, cols, seeded, any_seed, n_frames, wl_max, a_iter, f0_stale, f0min, records,
                 raw_amp=None, raw_slope=None):
        sig = s.numpy()
        rec = records.numpy()
        am = fm = None
        if mode == 1:                                            # (L, Kmax) like the reference; NaN where not resident
            am = np.full((L, Kmax), np.nan)
            fm = np.full((L, Kmax), np.nan)
            am[track_t0:track_t0 + track_len] = am_cur.numpy().T
            fm[track_t0:track_t0 + track_len] = fm_cur.numpy().T
        seeds = np.flatnonzero(seeded.numpy()) if mode == 1 and int(any_seed[0]) else np.array([], dtype=int)
        sp = 0
        for f in range(n_frames):
            i, c, wl = int(frame_inst[f]), int(frame_c[f]), int(frame_wl[f])
            row = rec[i]
            row[:] = 0
            if mode == 0:
                f0, K = float(frame_f0[f]), int(frame_K[f])
                amp, _ = O.iqhm_ls(sig[c - wl:c + wl + 1], np.arange(-K, K + 1) * f0, np.blackman(2 * wl + 1), fs)
                A, eta, slots = amp[K + 1:], np.zeros(K), np.arange(K)
                row[3 * Kmax] = amp[K].real
            else:
                f0 = f0_stale
                while sp < len(seeds) and seeds[sp] <= c:        # visible to frames at or after the seeded row
                    fm[seeds[sp], 0] = 140
                    am[seeds[sp], 0] = 10e-4
                    sp += 1
                n = int(ncol[f])
                nz = cols[f * Kmax:f * Kmax + n].numpy().astype(int)
                fw, aw = O.gather_fill(fm, am, c, wl, nz)
                z = np.zeros((fw.shape[0], 1))
                amp_t, slo_t = O.eaqhm_ls(sig[c - wl:c + wl + 1], np.concatenate((aw[::-1], z, aw), axis=1),
                                          np.concatenate((-fw[::-1], z, fw), axis=1), np.hamming(2 * wl + 1), fs)
                eta_t = fs / (2 * np.pi) * (amp_t.real * slo_t.imag - amp_t.imag * slo_t.real) / np.abs(amp_t) ** 2
                A, eta, slots = amp_t[n + 1:], eta_t[n + 1:], nz
                row[3 * Kmax] = amp_t[n].real
            mag = np.abs(A)
            with np.errstate(divide="ignore"):
                lg = 20 * np.log10(mag)
            ok = (lg > lg.max() - 150) & (np.abs(eta) < f0 / (a_iter + 1))
            k = slots[ok]
            row[k] = mag[ok]
            row[2 * Kmax + k] = np.angle(A[ok])
            if mode == 0:
                row[Kmax + k] = (k + 1) * f0
            else:
                row[Kmax + k] = fm[c, k] + (eta[ok] if f0 > f0min else 0.0)

    def spline_solve(self, records, No_ti, Kmax, step, code, mom, i_lo=0, i_hi=None):
        pass                                                      # folded into eval_synth below

    # ---- functions.py:337-388
    def eval_synth(self, records, code, mom, No_ti, Kmax, step, fs, L, t_lo, t_hi, s_lo, s_hi, target, std_det,
                   am_out, fm_out, track_t0, track_len, ph_knot, s_hat, partials, sums_out):
        rec = records.numpy()[:No_ti]
        ti = np.arange(1, L, step)
        c = ti - 1
        key = hash(rec.tobytes())        # (a streaming run asks for the same records once per time block)
        if getattr(self, "_interp_key", None) != key:
            am = np.zeros((L, Kmax))
            fm = np.zeros((L, Kmax))
            ph = np.zeros((L, Kmax))
            am[c], fm[c], ph[c] = rec[:, :Kmax], rec[:, Kmax:2 * Kmax], rec[:, 2 * Kmax:3 * Kmax]
            a0, fm_next = O.interpolate_tracks(rec[:, 3 * Kmax].copy(), am, fm, ph, ti, step, fs, L)
            self._interp_key, self._interp = key, (am, ph, a0, fm_next)
        am, ph, a0, fm_next = self._interp
        sh = a0 + 2 * (am * np.cos(ph)).sum(axis=1)
        if am_out is not None:
            am_out.numpy()[:, t_lo - track_t0:t_hi - track_t0] = am[t_lo:t_hi].T
            fm_out.numpy()[:, t_lo - track_t0:t_hi - track_t0] = fm_next[t_lo:t_hi].T
        if s_hat is None:
            return
        s_hat.numpy()[t_lo:t_hi] = sh[t_lo:t_hi]
        inside = (c >= t_lo) & (c < t_hi)
        ph_knot.numpy()[inside] = ph[c[inside]]
        d = target.numpy()[s_lo:s_hi] - sh[s_lo:s_hi]
        n = float(s_hi - s_lo)
        so = sums_out.numpy()
        so[:] = 0
        so[0], so[1], so[2] = d.sum(), (d * d).sum(), n
        # the fixed-point error sums of include/eaqhm_hip.h: sum of trunc(d * 2^60) and of trunc(d^2 * 2^64) as exact
        # integers, cut into base-2^32 limbs (any cut that adds up to the same integer serves the host's recombination)
        tot = sum(int(v) for v in np.trunc(d * 2.0 ** 60).astype(object))
        tot2 = sum(int(v) for v in np.trunc((d * d) * 2.0 ** 64).astype(object))
        limbs = so.view(np.int64)[8:16]
        for j, v in ((0, tot), (3, tot2)):
            limbs[j], limbs[j + 1], limbs[j + 2] = v & 0xffffffff, (v >> 32) & 0xffffffff, v >> 64
        mean = so[0] / n
        so[3] = 20 * np.log10(std_det / np.sqrt(so[1] / n - mean * mean))
