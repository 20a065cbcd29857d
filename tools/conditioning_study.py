"""Why the mis-scaled-pitch parity case (tests/test_gpu_parity.py, f0scale = 0.62) sits outside the 1e-8 amplitude bar:
the same NumPy restatement solved twice — inv(R) @ rhs as the reference does (functions.py:465/:530) and a Cholesky
solve (what the kernels do) — on every frame of that case, next to the 2-norm condition number of R.  Both are
"the same math, different rounding"; their distance is the rounding sensitivity of the frame, ~ cond(R) * eps.
CPU only (runs in the build container):  python tools/conditioning_study.py > profiles/r02_conditioning.txt"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import eaqhm_oracle as O
import scipy.linalg as sla
from eaqhm_amd import prologue
from eaqhm_amd.synth import synth_speech_int16

stats = []
_orig = O._weighted_ls


def both(E0, n, s, window):
    w = np.asarray(window, dtype=np.float64)[:, None]
    Ew = w * np.concatenate((E0, n * E0), axis=1)
    R = Ew.conj().T @ Ew
    rhs = Ew.conj().T @ (w[:, 0] * s)
    x_inv = np.linalg.inv(R) @ rhs
    x_ch = sla.cho_solve(sla.cho_factor(R, lower=True), rhs)
    d = np.sqrt(np.real(np.diag(R)))
    K = E0.shape[1]
    stats.append((np.linalg.cond(R), np.linalg.cond(R / np.outer(d, d)), np.abs(x_inv[:K] - x_ch[:K]).max(),
                  np.abs(x_inv[:K]).max(), K))
    return x_inv[:K], x_inv[K:]


for f0scale in (1.0, 0.62):
    stats.clear()
    O._weighted_ls = both
    fs = 16000
    s = synth_speech_int16(0.9, fs) / 32768.0
    t = np.arange(0, len(s) / fs, 0.001)
    f0 = f0scale * (220.0 + 40.0 * np.sin(2 * np.pi * 0.31 * t) + 10.0 * np.sin(2 * np.pi * 1.7 * t))
    grid = prologue.resample_track(np.column_stack([t, f0, np.ones_like(t)]), np.arange(0, len(s) - 1, 80) / fs)
    frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "female")
    r = O.analyse(s, fs, grid, np.array([f.ti for f in frames]), np.array([float(f.isSpeech) for f in frames]),
                  np.array([float(f.isVoiced) for f in frames]), fstep, f0min=160 if f0scale == 1.0 else 70, maxAdpt=2)
    O._weighted_ls = _orig
    st = np.array(stats)
    nf = len(st) // 3
    amax = st[:, 3].max()
    print("f0scale %.2f: %d frames x 3 adaptations, Kc %d-%d, SRER %s" % (f0scale, nf, st[:, 4].min(), st[:, 4].max(),
                                                                        ["%.3f" % v for v in r["SRER"]]))
    for a in range(3):
        q = st[a * nf:(a + 1) * nf]
        print("  adaptation %d: cond(R) median %.2e max %.2e | Jacobi-scaled max %.1f | |x_inv - x_chol| max %.2e "
              "= %.2e of the largest amplitude | max over frames of diff/(cond*eps*|x|max) %.2f"
              % (a, np.median(q[:, 0]), q[:, 0].max(), q[:, 1].max(), q[:, 2].max(), q[:, 2].max() / amax,
                 (q[:, 2] / (q[:, 0] * 2.2e-16 * q[:, 3])).max()))
