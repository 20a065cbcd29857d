"""GPU parity tests (run on the MI355X box: `pytest -m gpu`).  Everything goes through the C ABI of
libeaqhm_hip.so via the package's ctypes layer; the checker is the oracle and the golden vectors the
reference itself produced (tests/golden/make_golden.py).

Stated tolerances (SURVEY.md §8c, FP64 path): amplitudes <= 1e-8 * max, frequencies <= 1e-3 Hz,
phases <= 1e-5 rad (mod 2*pi), identical acceptance mask on >= 99.9 % of the cells,
SRER <= 1e-6 dB per adaptation (north-star bar: 0.1 dB).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, record_measurement, unpack_records

pytestmark = pytest.mark.gpu

TOL_AM_REL, TOL_FM_HZ, TOL_PH_RAD, TOL_SRER_DB = 1e-8, 1e-3, 1e-5, 1e-6
# Full-band 48 kHz, adaptation >= 1 only: partials within ~200 Hz of Nyquist advance their phase by ~pi per
# sample, and functions.py:375 takes fs/2pi * diff(unwrap(phase)) of them — a decision per sample that a last-bit
# difference can turn.  Measured on MI355X against the reference (profiles/r03_parity/parity_measurements.json): 1.6e-5 dB
# on the 0.6 s input (18.497426 vs 18.497410), 2.8e-7 dB on the 2 s input (34.1187839 vs 34.1187842; the NumPy oracle is
# 1.6e-9 dB from the reference there).  Stated tolerance, with margin for other inputs of this kind:
TOL_SRER_NYQUIST_DB = 1e-3


def wrap(d):
    return (d + np.pi) % (2 * np.pi) - np.pi


@pytest.fixture(scope="module")
def amd():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    import eaqhm_amd
    return eaqhm_amd


def relerr(a, b):
    return np.abs(np.asarray(a).ravel() - np.asarray(b).ravel()).max() / np.abs(np.asarray(b)).max()


# ----------------------------------------------------------------------------- LS seams
def test_ls_seams_unit_vectors(amd):
    u = load_golden("unit_vectors.npz")
    a, b = amd.iqhmLS_complexamps(u["iq_s"].reshape(-1, 1), u["iq_f0range"], u["iq_w"], 16000)
    assert a.shape == (len(u["iq_f0range"]), 1) and a.dtype == np.complex128
    assert relerr(a, u["iq_amp"]) < 1e-10 and relerr(b, u["iq_slope"]) < 1e-10
    a, b = amd.eaqhmLS_complexamps(u["ea_s"].reshape(-1, 1), u["ea_am"], u["ea_fm"], u["ea_w"], 16000)
    assert relerr(a, u["ea_amp"]) < 1e-10 and relerr(b, u["ea_slope"]) < 1e-10


def test_ls_seams_sa19_frames(amd, sa19_golden):
    g = sa19_golden
    for idx in (0, 700, 2000, 3500):
        p = "iqhm%d_" % idx
        a, b = amd.iqhmLS_complexamps(g[p + "s"], g[p + "f0range"], g[p + "window"], int(g[p + "fs"]))
        assert relerr(a, g[p + "amp"]) < 1e-9 and relerr(b, g[p + "slope"]) < 1e-9
        p = "eaqhm%d_" % idx
        a, b = amd.eaqhmLS_complexamps(g[p + "s"], g[p + "am"], g[p + "fm"], g[p + "window"], int(g[p + "fs"]))
        assert relerr(a, g[p + "amp"]) < 1e-8 and relerr(b, g[p + "slope"]) < 1e-8


def test_phase_integration_seam(amd):
    """phase_integr_interpolation (functions.py:537-575) against the reference's output, plus uneven knots vs the oracle."""
    import eaqhm_oracle as O
    u = load_golden("unit_vectors.npz")
    p = amd.phase_integr_interpolation(u["pii_fm"], u["pii_ph"], u["pii_knots"])
    assert p.shape == u["pii_out"].shape and np.abs(p - u["pii_out"]).max() <= 1e-12
    rng = np.random.default_rng(3)
    om = 2 * np.pi / 16000 * (300 + 50 * rng.standard_normal(200))
    kn = np.array([5, 20, 33, 64, 65, 120, 199])
    ph = np.zeros(200)
    ph[kn] = rng.uniform(-np.pi, np.pi, len(kn))
    assert np.abs(amd.phase_integr_interpolation(om, ph, kn) - O.phase_integr_interpolation(om, ph, kn)).max() <= 1e-12
    with pytest.raises(ValueError):
        amd.phase_integr_interpolation(om, ph, [10, 5])


def test_ls_seam_errors(amd):
    with pytest.raises(ValueError):
        amd.iqhmLS_complexamps(np.zeros(11), np.arange(3.0), np.ones(10), 16000)
    with pytest.raises(RuntimeError):     # even window length: the reference's eaqhm seam breaks too
        amd.eaqhmLS_complexamps(np.zeros(10), np.ones((10, 3)), np.ones((10, 3)), np.ones(10), 16000)


# ----------------------------------------------------------------------------- full SA19 run
@pytest.fixture(scope="module")
def sa19_run(amd, sa19_golden):
    g = sa19_golden
    seen = {}

    from eaqhm_amd import functions as F
    from eaqhm_amd import prologue
    from eaqhm_amd.engine import DeviceAnalysis, FramePlan
    fs, s = prologue.read_signal(os.path.join(GOLDEN, "SA19.WAV"))
    grid = prologue.resample_track(g["swipe_track"], np.arange(0, len(s) - 1, round(fs * 5 / 1000)) / fs)
    frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "female")
    prologue.apply_full_waveform(frames, len(s), 32 * 15)
    plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
    eng = DeviceAnalysis(s, s, plan, 160, 10)

    def hook(a, e):
        rec = e.records[0][:e.plan.No_ti].cpu().numpy()
        K = e.plan.Kmax
        seen[a] = dict(am=rec[:, :K].copy(), fm=rec[:, K:2 * K].copy(), ph=rec[:, 2 * K:3 * K].copy(),
                       a0=rec[:, 3 * K].copy(), s_hat=e.s_hat[0].cpu().numpy(),
                       ph_knot=e.ph_knot[0].cpu().numpy())
        if a == 0:
            seen["dense0"] = dict(am=e.am_cur.cpu().numpy(), fmcur=e.fm_cur.cpu().numpy())

    eng.run(on_adaptation=hook)
    fin = eng.final_arrays()
    det = F.pack_results(plan, fin)
    return dict(eng=eng, plan=plan, seen=seen, fin=fin, det=det)


def test_sa19_plan(sa19_run, sa19_golden):
    p, g = sa19_run["plan"], sa19_golden
    assert p.Kmax == 59 and p.No_ti == 4233 and p.n_frames == 4169
    assert np.array_equal(p.ti[p.analysed], g["ti_a0"])
    assert np.abs(p.frame_f0 - g["f0_a0"]).max() == 0
    assert abs(p.f0_stale - g["stale_f0"][0, 1]) == 0
    sh = g["ls_shapes_iqhm"]
    assert np.array_equal(2 * p.frame_wl + 1, sh[:, 0]) and np.array_equal(2 * p.frame_K + 1, sh[:, 1])


def test_sa19_srer(sa19_run, sa19_golden):
    srer = np.array(sa19_run["eng"].SRER)
    assert len(srer) == 6
    assert np.abs(srer - sa19_golden["SRER"]).max() < TOL_SRER_DB
    assert sa19_run["eng"].n_ls_frames == 6 * 4169


@pytest.mark.parametrize("a", [0, 1])
def test_sa19_records(sa19_run, sa19_golden, a):
    """Frame-centre records of adaptations 0 (iQHM) and 1 (eaQHM) against the reference's."""
    got = sa19_run["seen"][a]
    ref = unpack_records(sa19_golden, a, with_fm=(a > 0))
    mask = got["am"] != 0
    agree = np.mean(mask == ref["mask"])
    assert agree >= 0.999, "acceptance mask agreement %.6f" % agree
    both = mask & ref["mask"]
    assert np.abs(got["am"][both] - ref["am"][both]).max() <= TOL_AM_REL * ref["am"].max()
    assert np.abs(wrap(got["ph"][both] - ref["ph"][both])).max() <= TOL_PH_RAD
    assert np.abs(got["a0"] - ref["a0"]).max() <= TOL_AM_REL
    if a > 0:
        assert np.abs(got["fm"][both] - ref["fm"][both]).max() <= TOL_FM_HZ


def test_sa19_checksums_every_adaptation(sa19_run, sa19_golden):
    for a in range(6):
        got, gs = sa19_run["seen"][a], sa19_golden["recsum%d" % a]
        assert abs(np.count_nonzero(got["am"]) - int(gs[0])) <= 2
        assert abs(got["am"].sum() - gs[1]) <= 1e-7 * abs(gs[1])
        assert abs(got["fm"].sum() - gs[2]) <= 1e-7 * abs(gs[2])
        assert abs(got["a0"].sum() - gs[4]) <= 1e-7


def test_sa19_dense_after_adaptation0(sa19_run, sa19_golden):
    """Interpolation stage: dense am / next-iteration fm of slots 0, 30, 45, a0 and the synthesis."""
    g, d = sa19_golden, sa19_run["seen"]["dense0"]
    for k in (0, 30, 45):
        for lo in (0, 30000):
            p = "dense0_k%d_%d_" % (k, lo)
            n = len(g[p + "am"])
            assert np.abs(d["am"][k, lo:lo + n] - g[p + "am"]).max() <= TOL_AM_REL
            assert np.abs(d["fmcur"][k, lo:lo + n] - g[p + "fmcur"]).max() <= TOL_FM_HZ
    assert np.abs(sa19_run["seen"][0]["s_hat"] - g["dense0_srecon"]).max() <= 1e-9


def test_sa19_returned_structs(sa19_run, sa19_golden):
    g, det, fin = sa19_golden, sa19_run["det"], sa19_run["fin"]
    assert len(det) == 4233
    assert np.array_equal([d.ti for d in det], g["det_ti"])
    assert np.array_equal([d.isSpeech for d in det], g["det_isSpeech"])
    assert np.array_equal([d.isVoiced for d in det], g["det_isVoiced"])
    assert np.abs(fin["s_recon"] - g["s_recon"]).max() <= 1e-9
    v = g["det_isVoiced"]
    assert np.abs(np.array([float(d.a0) for d in np.array(det, dtype=object)[v]]) - g["det_a0"][v]).max() <= TOL_AM_REL
    cells = g["det_cells"]
    i, k = cells[:, 0], cells[:, 1]
    ref_mask = np.zeros_like(fin["am"], dtype=bool)
    ref_mask[i, k] = True
    assert np.mean((fin["am"] != 0) == ref_mask) >= 0.999
    ok = fin["am"][i, k] != 0
    assert np.abs(fin["am"][i, k][ok] - g["det_am"][ok]).max() <= TOL_AM_REL * g["det_am"].max()
    assert np.abs(fin["fm"][i, k][ok] - g["det_fm"][ok]).max() <= TOL_FM_HZ
    assert np.abs(wrap(fin["pk"][i, k][ok] - g["det_pk"][ok])).max() <= TOL_PH_RAD
    # Python-level shape quirks of the structs
    d = next(x for x in det if x.isVoiced)
    q = g["det_quirk"]
    assert type(d.ti).__name__ == q[0] and type(d.a0).__name__ == q[1] and str(d.amplitudes.dtype) == q[2]
    assert type(d.amplitudes[0]).__name__ in ("ndarray", "int") and d.ak == []
    first = next(e for e in d.amplitudes if isinstance(e, np.ndarray))
    assert first.shape == (1,)
    lens = g["det_len"]
    assert np.array_equal([len(x.amplitudes) if x.isVoiced else 0 for x in det], lens)


# ----------------------------------------------------------------------------- public entry point
def test_entry_point_signature_and_synth16k(amd):
    """eaQHMAnalysisAndSynthesis end to end on the 2 s synthetic 16 kHz signal, against the reference's
    own output for it (golden) — exercises the stop rule (adaptation 3 is rejected)."""
    import inspect
    from scipy.io import wavfile
    sig = inspect.signature(amd.eaQHMAnalysisAndSynthesis)
    names = list(sig.parameters)[:11]
    assert names == ["speechFile", "gender", "step", "maxAdpt", "pitchPeriods", "analysisWindow", "fullWaveform",
                     "fc", "partials", "printPrompts", "loadingScreen"]
    assert [sig.parameters[n].default for n in names[1:]] == ['other', 15, 10, 3, 32, True, 0, 0, True, True]
    g = load_golden("synth16k_2s_adpt3.npz")
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "eaqhm_synth16k.wav")
    wavfile.write(path, 16000, g["wav_int16"])
    s_recon, SRER, det, T = amd.eaQHMAnalysisAndSynthesis(path, "female", maxAdpt=3, printPrompts=False,
                                                         loadingScreen=False, pitch_track=g["swipe_track"])
    assert isinstance(SRER, list) and len(SRER) == 4 and isinstance(T, float)
    assert np.abs(np.array(SRER) - g["SRER"]).max() < TOL_SRER_DB
    assert s_recon.shape == g["s_recon"].shape and np.abs(s_recon - g["s_recon"]).max() <= 1e-9
    assert np.array_equal([d.isVoiced for d in det], g["det_isVoiced"])


def test_det_format_arrays_equals_structs(amd, sa19_golden):
    """Keyword-only extension det_format="arrays": the same numbers as the Deterministic list, as plain arrays."""
    wav = os.path.join(GOLDEN, "SA19.WAV")
    kw = dict(maxAdpt=1, printPrompts=False, pitch_track=sa19_golden["swipe_track"])
    _, srer_s, det, _ = amd.eaQHMAnalysisAndSynthesis(wav, "female", **kw)
    _, srer_a, arr, _ = amd.eaQHMAnalysisAndSynthesis(wav, "female", det_format="arrays", **kw)
    assert srer_s == srer_a and set(arr) == {"ti", "isSpeech", "isVoiced", "a0", "amplitudes", "frange", "pk"}
    assert np.array_equal(arr["ti"], [d.ti for d in det]) and np.array_equal(arr["isVoiced"], [d.isVoiced for d in det])
    assert np.array_equal(arr["isSpeech"], [d.isSpeech for d in det])
    for i in (0, 40, 41, 1000, 2500, 4200, 4232):
        d = det[i]
        if not d.isVoiced:
            assert not arr["amplitudes"][i].any() and arr["a0"][i] == 0
            continue
        assert arr["a0"][i] == d.a0
        for name in ("amplitudes", "frange", "pk"):
            row, obj = arr[name][i], getattr(d, name)
            assert not row[len(obj):].any()
            assert all((row[k] == obj[k][0]) if isinstance(obj[k], np.ndarray) else (row[k] == 0) for k in range(len(obj)))
    with pytest.raises(ValueError):
        amd.eaQHMAnalysisAndSynthesis(wav, "female", det_format="json")


def test_voiced_only_option(amd):
    """fullWaveform=False (functions.py:127-138) against the reference's run."""
    g = load_golden("sa19_female_voicedonly_adpt1.npz")
    s_recon, SRER, det, _ = amd.eaQHMAnalysisAndSynthesis(os.path.join(GOLDEN, "SA19.WAV"), "female", maxAdpt=1,
                                                         fullWaveform=False, printPrompts=False,
                                                         pitch_track=g["swipe_track"])
    assert np.abs(np.array(SRER) - g["SRER"]).max() < TOL_SRER_DB
    assert np.array_equal([d.isSpeech for d in det], g["det_isSpeech"])
    assert np.array_equal([d.isVoiced for d in det], g["det_isVoiced"])
    assert np.abs(s_recon - g["s_recon"]).max() <= 1e-9


def test_spline_range_equals_full_solve(sa19_run):
    """eaqhm_spline_solve_range on a sub-range of the instants gives, inside the range, the run codes and moments of
    the full solve (what a rank of a sharded run relies on); rows 0..3 of the codes are always produced."""
    import torch
    eng = sa19_run["eng"]
    p, c = eng.plan, eng.ctx
    rec = eng.records[1]
    code_f, mom_f = torch.zeros_like(eng.code), torch.zeros_like(eng.mom)
    c.spline_solve(rec, p.No_ti, p.Kmax, p.step, code_f, mom_f)
    for lo, hi in ((0, 500), (1800, 2600), (p.No_ti - 300, p.No_ti)):
        code_r = torch.full_like(eng.code, 99)
        mom_r = torch.full_like(eng.mom, 1e300)
        c.spline_solve(rec, p.No_ti, p.Kmax, p.step, code_r, mom_r, lo, hi)
        torch.cuda.synchronize()
        assert torch.equal(code_r[lo:hi], code_f[lo:hi]) and torch.equal(code_r[:4], code_f[:4])
        assert torch.equal(mom_r[lo:hi], mom_f[lo:hi])


# ----------------------------------------------------------------------------- against the oracle
@pytest.mark.parametrize("params", [dict(step=15, pitchPeriods=3, analysisWindow=32, partials=0),
                                    dict(step=10, pitchPeriods=4, analysisWindow=50, partials=20),
                                    dict(step=15, pitchPeriods=3, analysisWindow=32, partials=0, f0scale=0.62)])
def test_against_oracle_seeded_signal(amd, params):
    """Same seeded synthetic input through the HIP path and the oracle, non-default parameters too.  The third
    case hands both a pitch track at 0.62 of the true pitch (105-167 Hz, a male-voice frame geometry: 46-74
    harmonics, windows up to ~460 samples), so one launch mixes frames of the register-resident kernel with
    frames of the large-frame kernel."""
    params = dict(params)
    f0scale = params.pop("f0scale", 1.0)
    f0min = 160 if f0scale == 1.0 else 70
    import eaqhm_oracle as O
    from eaqhm_amd import prologue
    from eaqhm_amd.engine import DeviceAnalysis, FramePlan
    from eaqhm_amd.synth import synth_speech_int16
    fs = 16000
    s = synth_speech_int16(0.9, fs) / 32768.0
    t = np.arange(0, len(s) / fs, 0.001)
    f0 = f0scale * (220.0 + 40.0 * np.sin(2 * np.pi * 0.31 * t) + 10.0 * np.sin(2 * np.pi * 1.7 * t))
    track = np.column_stack([t, f0, np.ones_like(t)])
    grid = prologue.resample_track(track, np.arange(0, len(s) - 1, round(fs * 5 / 1000)) / fs)
    frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "female")
    ti5 = np.array([f.ti for f in frames])
    sp = np.array([float(f.isSpeech) for f in frames])
    vo = np.array([float(f.isVoiced) for f in frames])
    # rounding sensitivity of the frames themselves: the same NumPy normal equations solved with inv() (the reference,
    # functions.py:465/:530) and by Cholesky (what the kernels do); see tools/conditioning_study.py
    import scipy.linalg as sla
    sens, plain_ls = [0.0, 0.0], O._weighted_ls

    def two_roundings(E0, n, sw, window):
        w = np.asarray(window, dtype=np.float64)[:, None]
        Ew = w * np.concatenate((E0, n * E0), axis=1)
        R, rhs = Ew.conj().T @ Ew, Ew.conj().T @ (w[:, 0] * sw)
        x = np.linalg.inv(R) @ rhs
        Kc = E0.shape[1]
        sens[0] = max(sens[0], np.abs(x[:Kc] - sla.cho_solve(sla.cho_factor(R, lower=True), rhs)[:Kc]).max())
        sens[1] = max(sens[1], np.abs(x[:Kc]).max())
        return x[:Kc], x[Kc:]

    O._weighted_ls = two_roundings
    try:
        ref = O.analyse(s, fs, grid, ti5, sp, vo, fstep, f0min=f0min, maxAdpt=2, step=params["step"],
                        pitchPeriods=params["pitchPeriods"], analysisWindow=params["analysisWindow"],
                        partials=params["partials"])
    finally:
        O._weighted_ls = plain_ls
    prologue.apply_full_waveform(frames, len(s), params["analysisWindow"] * params["step"])
    plan = FramePlan(len(s), fs, grid, frames, fstep, params["step"], params["pitchPeriods"],
                     params["analysisWindow"], params["partials"])
    eng = DeviceAnalysis(s, s, plan, f0min, 2)
    eng.run()
    fin = eng.final_arrays()
    if f0scale != 1.0:
        nt = (2 * (2 * plan.frame_K + 1) + 1 + 15) // 16
        assert (nt <= 13).any() and (nt > 13).any()      # both LS kernels take part
    assert np.abs(np.array(eng.SRER) - np.array(ref["SRER"])).max() < TOL_SRER_DB
    # Conditioning-aware bar.  SURVEY §8c states its FP64 tolerances for cond(R) <= 1.1e5 (Q4); with the mis-scaled
    # pitch every other basis column sits between true partials and cond(R) of adaptation 1 reaches 1.3e10
    # (profiles/r02_parity/conditioning_f0scale.txt): inv() and Cholesky of the SAME NumPy matrices already differ by
    # 3e-8 of the largest amplitude.  The kernels may be off by at most 3x that distance between two CPU roundings.
    rel_sens = sens[0] / sens[1]
    loose = max(1.0, 3.0 * rel_sens / TOL_AM_REL)
    assert loose == 1.0 if f0scale == 1.0 else 1.0 < loose < 30.0, (rel_sens, loose)
    assert np.abs(fin["s_recon"] - ref["s_recon"]).max() <= 1e-9 * loose
    m = ref["am"] != 0
    assert np.mean((fin["am"] != 0) == m) >= 0.999
    both = m & (fin["am"] != 0)
    assert np.abs(fin["am"][both] - ref["am"][both]).max() <= loose * TOL_AM_REL * ref["am"].max()
    assert np.abs(fin["fm"][both] - ref["fm"][both]).max() <= loose * TOL_FM_HZ
    assert np.abs(wrap(fin["pk"][both] - ref["pk"][both])).max() <= loose * TOL_PH_RAD


@pytest.mark.parametrize("fs,dur,gender", [(8000, 0.8, "female"), (22050, 0.45, "female"), (44100, 0.3, "other"),
                                           (11025, 0.7, "male")])
def test_other_sampling_rates_against_oracle(amd, fs, dur, gender):
    """Sampling rates the fixtures do not have (telephone band to CD rate; the reference takes whatever the wav file says,
    functions.py:86): frame geometry, Fmax = fs/2 - 200, Kmax and the window lengths all change with fs, and with them
    which LS kernel a frame takes.  HIP path against the oracle on the same seeded input and the same analytic pitch track,
    adaptations 0-1."""
    import eaqhm_oracle as O
    from eaqhm_amd import prologue
    from eaqhm_amd.engine import DeviceAnalysis, FramePlan
    from eaqhm_amd.synth import synth_speech_int16
    s = synth_speech_int16(dur, fs) / 32768.0
    t = np.arange(0, len(s) / fs + 0.002, 0.001)      # (one ms more than SWIPE' gives: at 44.1 kHz the 5 ms grid of
    scale = 0.5 if gender == "male" else 1.0          #  functions.py:113 ends after the last whole millisecond)
    f0 = scale * (220.0 + 40.0 * np.sin(2 * np.pi * 0.31 * t) + 10.0 * np.sin(2 * np.pi * 1.7 * t))
    f0min = prologue.pitch_limits(gender)[0]
    grid = prologue.resample_track(np.column_stack([t, f0, np.ones_like(t)]),
                                   np.arange(0, len(s) - 1, round(fs * 5 / 1000)) / fs)
    frames, fstep = prologue.voiced_unvoiced_frames(s, fs, gender)
    ti5 = np.array([f.ti for f in frames])
    sp = np.array([float(f.isSpeech) for f in frames])
    vo = np.array([float(f.isVoiced) for f in frames])
    seen_ref = {}
    ref = O.analyse(s, fs, grid, ti5, sp, vo, fstep, f0min=f0min, maxAdpt=1,
                    on_adaptation=lambda a, rec, st: seen_ref.update({a: {k: np.array(v) for k, v in rec.items()}}))
    prologue.apply_full_waveform(frames, len(s), 32 * 15)
    plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
    assert plan.n_frames == ref["n_ls_frames"] // len(ref["SRER"]) and plan.n_frames > 100
    seen = {}
    eng = DeviceAnalysis(s, s, plan, f0min, 1)
    eng.run(on_adaptation=lambda a, e: seen.update({a: e.records[0][:plan.No_ti].cpu().numpy().copy()}))
    record_measurement("fs_%d_%s" % (fs, gender), srer_hip=[float(v) for v in eng.SRER],
                       srer_oracle=[float(v) for v in ref["SRER"]], Kmax=int(plan.Kmax),
                       Kc_max=int(2 * plan.frame_K.max() + 1), N_max=int(2 * plan.frame_wl.max() + 1))
    assert len(eng.SRER) == len(ref["SRER"]) == 2
    assert abs(eng.SRER[0] - ref["SRER"][0]) < TOL_SRER_DB and abs(eng.SRER[1] - ref["SRER"][1]) < TOL_SRER_NYQUIST_DB
    K = plan.Kmax
    for a in (0, 1):
        am, ph = seen[a][:, :K], seen[a][:, 2 * K:3 * K]
        r = seen_ref[a]
        m = r["am"] != 0
        assert np.mean((am != 0) == m) >= 0.999
        both = m & (am != 0)
        q = np.abs(am[both] - r["am"][both]) / r["am"].max()
        assert np.quantile(q, 0.999) <= TOL_AM_REL and q.max() <= 1e-6      # (worst cells: noise-only margins, cf. the low voice)
        strong = both & (r["am"] > 1e-6 * r["am"].max())
        assert np.quantile(np.abs(wrap(ph[strong] - r["ph"][strong])), 0.999) <= TOL_PH_RAD
    assert np.abs(eng.final_arrays()["s_recon"] - ref["s_recon"]).max() <= 1e-8


@pytest.mark.parametrize("case", ["too_short", "two_frames", "digital_silence", "silence_then_speech", "constant"])
def test_degenerate_inputs_like_the_oracle(amd, case):
    """Inputs at the edge of the domain, each against the oracle: a file too short for any analysed frame (no frames, SRER
    0 dB: the reconstruction is zero), one with a handful of frames, digital silence (every SRER nan: 0 / 0 in
    functions.py:388, nothing accepted), silence followed by speech, and a constant.  The constant is the one documented
    divergence: its partials are rounding noise, the frequency tracks derived from them make the systems of adaptation 1
    numerically singular — the reference's inv() returns noise (SRER -inf: the target's variance is zero), this
    implementation reports the Cholesky breakdown as LinAlgError."""
    import warnings
    import eaqhm_oracle as O
    from eaqhm_amd import prologue
    from eaqhm_amd.engine import DeviceAnalysis, FramePlan
    from eaqhm_amd.synth import synth_speech_int16
    fs = 16000
    x = synth_speech_int16(0.5, fs) / 32768.0
    s = {"too_short": x[:800], "two_frames": x[:1060], "digital_silence": np.zeros(4800),
         "silence_then_speech": np.concatenate((np.zeros(3000), x[:5000])), "constant": np.full(4800, 0.1)}[case]
    n = len(s)
    tt = np.arange(0, n / fs, 0.001)
    track = np.column_stack([tt, np.full(len(tt), 200.0), np.ones_like(tt)])
    gt = np.arange(0, n - 1, round(fs * 5 / 1000)) / fs
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")            # log10(0), 0 / 0 of the silent cases, as in the reference
        ti5, sp, vo, fstep_o = O.voiced_unvoiced_frames(s, fs, "female")
        ref = O.analyse(s, fs, O.get_linear(track, gt), ti5, sp, vo, fstep_o, f0min=160, maxAdpt=1)
        grid = prologue.resample_track(track, gt)
        frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "female")
        prologue.apply_full_waveform(frames, n, 480)
        plan = FramePlan(n, fs, grid, frames, fstep, 15, 3, 32, 0)
        eng = DeviceAnalysis(s, s, plan, 160, 1)
        if case == "constant":
            assert np.isneginf(ref["SRER"]).all()
            with pytest.raises(np.linalg.LinAlgError):
                eng.run()
            return
        eng.run()
    fin = eng.final_arrays()
    assert eng.ctx.ls_faults() == (0, 0, 0)
    assert (case != "too_short") or plan.n_frames == 0
    srer, want = np.array(eng.SRER, dtype=np.float64), np.array(ref["SRER"], dtype=np.float64)
    assert len(srer) == len(want) and np.array_equal(np.isnan(srer), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.abs(srer[ok] - want[ok]).max(initial=0.0) < TOL_SRER_DB
    assert np.array_equal(fin["am"] != 0, ref["am"] != 0)
    assert np.abs(fin["s_recon"] - ref["s_recon"]).max() <= 1e-9
    assert np.abs(fin["am"] - ref["am"]).max(initial=0.0) <= TOL_AM_REL * max(ref["am"].max(initial=0.0), 1e-300)


def test_pitch_glide_across_the_kernel_boundary(amd):
    """A glide from 130 to 200 Hz at 16 kHz: the systems shrink from 16 tile rows to 10 in the course of the file, so one
    launch holds frames of the large-frame kernels (more than 13 tile rows: eaqhm_ls_a0big_kernel / eaqhm_ls_mfma_kernel
    as the tile kernel's left-over class) next to frames of four size classes of the tile kernel.  Against the oracle."""
    import eaqhm_oracle as O
    from eaqhm_amd import prologue
    from eaqhm_amd.engine import DeviceAnalysis, FramePlan
    fs, n = 16000, 11200
    rng = np.random.default_rng(11)
    t = np.arange(n) / fs
    f0 = 130.0 + 70.0 * t / t[-1]
    phi = 2 * np.pi * np.cumsum(f0) / fs
    x = np.zeros(n)
    for k in range(1, 40):
        x += k ** -1.1 * np.cos(k * phi + rng.uniform(0, 2 * np.pi)) * (k * f0 < 0.47 * fs)
    x += rng.standard_normal(n) * np.sqrt(np.mean(x ** 2)) * 10 ** (-50 / 20)
    s = np.round(0.25 * x / np.abs(x).max() * 32767) / 32768.0
    tt = np.arange(0, n / fs, 0.001)
    track = np.column_stack([tt, 130.0 + 70.0 * tt / t[-1], np.ones_like(tt)])
    grid = prologue.resample_track(track, np.arange(0, n - 1, round(fs * 5 / 1000)) / fs)
    frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "other")
    for fr in frames:
        fr.isSpeech = fr.isVoiced = True
    ti5 = np.array([f.ti for f in frames])
    ones = np.ones(len(frames))
    ref = O.analyse(s, fs, grid, ti5, ones, ones, fstep, f0min=70, maxAdpt=2, step=15, pitchPeriods=3, analysisWindow=32,
                    partials=0)
    plan = FramePlan(n, fs, grid, frames, fstep, 15, 3, 32, 0)
    rows = (2 * (2 * plan.frame_K + 1) + 1 + 15) // 16
    assert rows.max() > 13 and rows.min() <= 11 and len(np.unique(rows)) >= 5
    eng = DeviceAnalysis(s, s, plan, 70, 2)
    eng.run()
    assert eng.ctx.ls_faults() == (0, 0, 0)
    fin = eng.final_arrays()
    assert len(eng.SRER) == len(ref["SRER"])
    assert np.abs(np.array(eng.SRER) - np.array(ref["SRER"])).max() < TOL_SRER_DB
    assert np.abs(fin["s_recon"] - ref["s_recon"]).max() <= 1e-9
    m = ref["am"] != 0
    assert np.mean((fin["am"] != 0) == m) >= 0.999
    both = m & (fin["am"] != 0)
    assert np.abs(fin["am"][both] - ref["am"][both]).max() <= TOL_AM_REL * ref["am"].max()
    assert np.abs(fin["fm"][both] - ref["fm"][both]).max() <= TOL_FM_HZ
    strong = both & (ref["am"] > 1e-6 * ref["am"].max())
    assert np.abs(wrap(fin["pk"][strong] - ref["pk"][strong])).max() <= TOL_PH_RAD
    record_measurement("pitch_glide_across_kernels", tile_rows=[int(rows.min()), int(rows.max())],
                       srer_abs_diff_db=float(np.abs(np.array(eng.SRER) - np.array(ref["SRER"])).max()))


def test_long_windows_against_oracle(amd):
    """A low voice at 32 kHz (f0 86-94 Hz, 'male' limits): windows of 1020-1120 samples — longer than the 1024-sample
    chunks of the zero counts and than the 16 x 64-sample masks the register-resident kernel is laid out for — and
    343-371 basis columns, i.e. real systems of 22-24 tile rows at adaptation 0: beyond what eaqhm_ls_a0big_kernel
    holds in registers, so that launch takes the complex system through memory; adaptation 1 runs the large-frame
    kernel with bridged gaps in windows that straddle two chunk boundaries."""
    import eaqhm_oracle as O
    from eaqhm_amd import prologue
    from eaqhm_amd.engine import DeviceAnalysis, FramePlan
    fs, n = 32000, 5400
    rng = np.random.default_rng(7)
    t = np.arange(n) / fs
    f0 = 90.0 + 4.0 * np.sin(2 * np.pi * 1.3 * t)
    phi = 2 * np.pi * np.cumsum(f0) / fs
    x = np.zeros(n)
    for k in range(1, 160):
        x += k ** -1.2 * np.cos(k * phi + rng.uniform(0, 2 * np.pi))
    x += rng.standard_normal(n) * np.sqrt(np.mean(x ** 2)) * 10 ** (-45 / 20)
    s = np.round(0.25 * x / np.abs(x).max() * 32767) / 32768.0
    tt = np.arange(0, n / fs, 0.001)
    track = np.column_stack([tt, 90.0 + 4.0 * np.sin(2 * np.pi * 1.3 * tt), np.ones_like(tt)])
    grid = prologue.resample_track(track, np.arange(0, n - 1, round(fs * 5 / 1000)) / fs)
    frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "male")
    for fr in frames:                       # every instant voiced speech: the geometry is what is under test
        fr.isSpeech = fr.isVoiced = True
    ti5 = np.array([f.ti for f in frames])
    ones = np.ones(len(frames))
    # step 60: the unanalysed margin analysisWindow * step (functions.py:180) is then longer than half a window
    ref = O.analyse(s, fs, grid, ti5, ones, ones, fstep, f0min=70, maxAdpt=1, step=60, pitchPeriods=3, analysisWindow=32,
                    partials=0)
    plan = FramePlan(n, fs, grid, frames, fstep, 60, 3, 32, 0)
    assert (2 * plan.frame_wl + 1).max() > 1024 and ((2 * plan.frame_K + 2 + 15) // 16).min() > 19
    eng = DeviceAnalysis(s, s, plan, 70, 1)
    eng.run()
    fin = eng.final_arrays()
    assert len(eng.SRER) == len(ref["SRER"])
    assert np.abs(np.array(eng.SRER) - np.array(ref["SRER"])).max() < TOL_SRER_DB
    assert np.abs(fin["s_recon"] - ref["s_recon"]).max() <= 1e-9
    m = ref["am"] != 0
    assert np.mean((fin["am"] != 0) == m) >= 0.999
    both = m & (fin["am"] != 0)
    assert np.abs(fin["am"][both] - ref["am"][both]).max() <= TOL_AM_REL * ref["am"].max()
    assert np.abs(fin["fm"][both] - ref["fm"][both]).max() <= TOL_FM_HZ
    strong = both & (ref["am"] > 1e-6 * ref["am"].max())      # the angle of a vanishing partial is ill-conditioned
    assert np.abs(wrap(fin["pk"][strong] - ref["pk"][strong])).max() <= TOL_PH_RAD


def test_entry_point_with_own_swipe(amd, sa19_golden):
    """The complete drop-in call — wav in, (s_recon, SRER, DetComponents, time) out — with the pitch track
    estimated by the package's own SWIPE' restatement (no fixture)."""
    s_recon, SRER, det, T = amd.eaQHMAnalysisAndSynthesis(os.path.join(GOLDEN, "SA19.WAV"), "female", maxAdpt=2,
                                                         printPrompts=False, loadingScreen=False)
    assert np.abs(np.array(SRER) - sa19_golden["SRER"][:3]).max() < TOL_SRER_DB
    assert len(det) == 4233 and s_recon.shape == (63488,)


def test_option_paths_against_oracle(amd, tmp_path):
    """fc > 0 (high-pass pre-filter, functions.py:90-91), partials > 0 (:117-118), tuple gender (:95-97) and a
    non-default step: host-side options that only change the kernels' inputs."""
    import eaqhm_oracle as O
    from scipy.io import wavfile
    from eaqhm_amd.swipe import swipep
    from eaqhm_amd.synth import synth_speech_int16
    fs = 16000
    x = synth_speech_int16(0.8, fs)
    path = str(tmp_path / "opt.wav")
    wavfile.write(path, fs, x)
    gender, fc, partials, step = (150, 320), 60, 25, 12
    s_recon, SRER, det, _ = amd.eaQHMAnalysisAndSynthesis(path, gender, step=step, maxAdpt=2, fc=fc,
                                                         partials=partials, printPrompts=False)
    s = O.ellip_filter(x / 32768.0, fs, fc)
    track = swipep(s, fs, list(gender))
    grid = O.get_linear(track, np.arange(0, len(s) - 1, round(fs * 5 / 1000)) / fs)
    ti5, sp, vo, fstep = O.voiced_unvoiced_frames(s, fs, gender)
    ref = O.analyse(s, fs, grid, ti5, sp, vo, fstep, f0min=gender[0], maxAdpt=2, step=step, partials=partials)
    assert len(SRER) == len(ref["SRER"])
    assert np.abs(np.array(SRER) - np.array(ref["SRER"])).max() < TOL_SRER_DB
    assert np.abs(s_recon - ref["s_recon"]).max() <= 1e-9
    assert np.array_equal([d.isVoiced for d in det], ref["isVoiced"])


def test_option_paths_against_the_reference(amd, tmp_path):
    """Every host-side option of the signature off its default at once — tuple gender (functions.py:95-97), step,
    pitchPeriods, analysisWindow, fullWaveform=False (:127-138), fc > 0 (:90-91), partials > 0 (:117-118) — against the
    reference's own run with those options (tests/golden/make_golden.py options16k): SRER of all four adaptations, the
    reconstruction, the flags and every cell of the returned structs; once with the reference's pitch track and once
    with this package's own SWIPE' on the pre-filtered signal."""
    from scipy.io import wavfile
    g = load_golden("options16k_1p5s.npz")
    path = str(tmp_path / "opt_ref.wav")
    wavfile.write(path, 16000, g["wav_int16"])
    kw = dict(step=12, maxAdpt=3, pitchPeriods=4, analysisWindow=40, fullWaveform=False, fc=60, partials=25,
              printPrompts=False)
    for track in (g["swipe_track"], None):
        s_recon, SRER, det, _ = amd.eaQHMAnalysisAndSynthesis(path, (150, 320), pitch_track=track, **kw)
        assert len(SRER) == len(g["SRER"]) == 4 and np.abs(np.array(SRER) - g["SRER"]).max() < TOL_SRER_DB
        assert np.abs(s_recon - g["s_recon"]).max() <= 1e-9
        assert np.array_equal([d.ti for d in det], g["det_ti"])
        assert np.array_equal([d.isSpeech for d in det], g["det_isSpeech"])
        assert np.array_equal([d.isVoiced for d in det], g["det_isVoiced"])
        v = g["det_isVoiced"]
        assert np.abs(np.array([float(d.a0) for d in det if d.isVoiced]) - g["det_a0"][v]).max() <= TOL_AM_REL
        assert np.array_equal([len(d.amplitudes) if d.isVoiced else 0 for d in det], g["det_len"])
        cells = g["det_cells"]
        got = np.array([[float(det[i].amplitudes[k][0]) if isinstance(det[i].amplitudes[k], np.ndarray) else 0.0,
                         float(det[i].frange[k][0]) if isinstance(det[i].frange[k], np.ndarray) else 0.0,
                         float(det[i].pk[k][0]) if isinstance(det[i].pk[k], np.ndarray) else 0.0] for i, k in cells])
        ok = got[:, 0] != 0
        assert ok.mean() >= 0.999
        assert np.abs(got[ok, 0] - g["det_am"][ok]).max() <= TOL_AM_REL * g["det_am"].max()
        assert np.abs(got[ok, 1] - g["det_fm"][ok]).max() <= TOL_FM_HZ
        assert np.abs(wrap(got[ok, 2] - g["det_pk"][ok])).max() <= TOL_PH_RAD


def test_sa19_large_frame_kernel_on_every_frame(amd, sa19_golden):
    """The large-frame LS kernel (MFMA Gramian in passes, tile Cholesky through memory) forced onto every SA19
    frame (EAQHM_OPT_LS_VARIANT = 2): three adaptations (modes 0 and 1) against the reference's SRER."""
    from eaqhm_amd import prologue
    from eaqhm_amd.engine import DeviceAnalysis, FramePlan
    fs, s = prologue.read_signal(os.path.join(GOLDEN, "SA19.WAV"))
    grid = prologue.resample_track(sa19_golden["swipe_track"], np.arange(0, len(s) - 1, round(fs * 5 / 1000)) / fs)
    frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "female")
    prologue.apply_full_waveform(frames, len(s), 32 * 15)
    plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
    eng = DeviceAnalysis(s, s, plan, 160, 2)
    eng.ctx.set_option(1, 2)
    eng.run()
    assert len(eng.SRER) == 3
    assert np.abs(np.array(eng.SRER) - sa19_golden["SRER"][:3]).max() < TOL_SRER_DB


def test_batch_matches_single_file_runs(amd, sa19_golden, tmp_path):
    """several files interleaved on separate streams give every file exactly the result of its own
    run (bitwise: same kernels, same inputs), each with its own number of executed adaptations."""
    import scipy.io.wavfile as wavfile
    from eaqhm_amd.synth import synth_speech_int16
    fs = 16000
    x = synth_speech_int16(0.8, fs)
    wav2 = str(tmp_path / "synth.wav")
    wavfile.write(wav2, fs, x)
    t = np.arange(0, len(x) / fs, 0.001)
    f0 = 220.0 + 40.0 * np.sin(2 * np.pi * 0.31 * t) + 10.0 * np.sin(2 * np.pi * 1.7 * t)
    track2 = np.column_stack([t, f0])
    wav1 = os.path.join(GOLDEN, "SA19.WAV")
    files, tracks = [wav1, wav2, wav1], [sa19_golden["swipe_track"], track2, sa19_golden["swipe_track"]]
    kw = dict(maxAdpt=3, printPrompts=False)
    single = [amd.eaQHMAnalysisAndSynthesis(f, "female", pitch_track=tr, **kw) for f, tr in zip(files[:2], tracks[:2])]
    batch = amd.eaQHMAnalysisAndSynthesisBatch(files, "female", pitch_tracks=tracks, **kw)
    assert len(batch) == 3
    for b, ref in zip(batch, [single[0], single[1], single[0]]):
        assert np.array_equal(b[0], ref[0])
        assert [float(v) for v in b[1]] == [float(v) for v in ref[1]]
        assert len(b[2]) == len(ref[2])
        assert all(np.array_equal(np.asarray(p.a0), np.asarray(q.a0)) for p, q in zip(b[2][::97], ref[2][::97]))
    assert np.abs(np.array(batch[0][1]) - sa19_golden["SRER"][:4]).max() < TOL_SRER_DB


def test_48khz_large_frames():
    """BASELINE config 5 in miniature: 0.6 s of the synthetic signal at 48 kHz (N up to 901, Kc up to ~300: the
    frames take the large-frame LS kernel).  Adaptation 0 is pinned against the reference's run; adaptation 1
    collapses in the reference itself (near-Nyquist unwrap flips) and is only required to be
    rejected by the stop rule, as in the reference."""
    from eaqhm_amd import prologue
    from eaqhm_amd.engine import DeviceAnalysis, FramePlan
    g = load_golden("synth48k_0p6s_adpt1.npz")
    fs = 48000
    s = g["wav_int16"] / 32768.0
    grid = prologue.resample_track(g["swipe_track"], np.arange(0, len(s) - 1, round(fs * 5 / 1000)) / fs)
    frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "female")
    prologue.apply_full_waveform(frames, len(s), 32 * 15)
    plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
    sh = g["ls_shapes_iqhm"]
    assert np.array_equal(2 * plan.frame_wl + 1, sh[:, 0]) and np.array_equal(2 * plan.frame_K + 1, sh[:, 1])
    seen = {}

    def hook(a, e):
        rec = e.records[0][:plan.No_ti].cpu().numpy()
        seen[a] = rec.copy()

    eng = DeviceAnalysis(s, s, plan, 160, 1)
    eng.run(on_adaptation=hook)
    assert abs(eng.SRER[0] - g["SRER"][0]) < TOL_SRER_DB
    assert len(eng.SRER) == 2 and eng.SRER[1] < eng.SRER[0]
    record_measurement("synth48k_0p6s_fullband", srer_hip=[float(v) for v in eng.SRER],
                       srer_reference=[float(v) for v in g["SRER"]])
    # adaptation 1 of the full band (reference: 54.89 -> 18.50 dB): its value hangs on unwrap() decisions of partials
    # whose phase advances by ~pi per sample (functions.py:375) — a coin flip per sample that rounding-level
    # differences turn.  Stated tolerance for this 0.6 s input: TOL_SRER_NYQUIST_DB (measured distance: see
    # profiles/r03_parity/measurements.json)
    assert abs(eng.SRER[1] - g["SRER"][1]) < TOL_SRER_NYQUIST_DB
    ref = unpack_records(g, 0, with_fm=False)
    K = plan.Kmax
    am, ph = seen[0][:, :K], seen[0][:, 2 * K:3 * K]
    assert np.mean((am != 0) == ref["mask"]) >= 0.999
    both = (am != 0) & ref["mask"]
    assert np.abs(am[both] - ref["am"][both]).max() <= TOL_AM_REL * ref["am"].max()
    strong = both & (ref["am"] > 1e-6 * ref["am"].max())      # the angle of a vanishing partial is ill-conditioned
    assert np.abs(wrap(ph[strong] - ref["ph"][strong])).max() <= TOL_PH_RAD
    assert np.abs(eng.final_arrays()["s_recon"] - g["s_recon"]).max() <= 1e-9


def test_48khz_fullband_2s_against_reference():
    """Full-band 48 kHz at a size between the 0.6 s fixture and the 60 s bench workload: 2 s, maxAdpt=1, through the
    reference itself (tests/golden/make_golden.py synth48k_2s; its own SWIPE' on these 2 s).  Adaptation 0 to the usual
    1e-6 dB; adaptation 1 (reference: 57.63 -> 34.12 dB, the near-Nyquist collapse diluted by the length) within
    TOL_SRER_NYQUIST_DB, plus the checksums of the frame-centre records of adaptation 0 and the decimated s_recon."""
    from eaqhm_amd import prologue
    from eaqhm_amd.engine import DeviceAnalysis, FramePlan
    g = load_golden("synth48k_2s_adpt1.npz")
    fs = 48000
    s = g["wav_int16"] / 32768.0
    frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "female")
    prologue.apply_full_waveform(frames, len(s), 32 * 15)
    plan = FramePlan(len(s), fs, g["f0s_5ms"], frames, fstep, 15, 3, 32, 0)
    sh = g["ls_shapes_iqhm"]
    assert np.array_equal(2 * plan.frame_wl + 1, sh[:, 0]) and np.array_equal(2 * plan.frame_K + 1, sh[:, 1])
    seen = {}
    eng = DeviceAnalysis(s, s, plan, 160, 1)
    eng.run(on_adaptation=lambda a, e: seen.update({a: e.records[0][:plan.No_ti].cpu().numpy().copy()}))
    record_measurement("synth48k_2s_fullband", srer_hip=[float(v) for v in eng.SRER],
                       srer_reference=[float(v) for v in g["SRER"]])
    assert len(eng.SRER) == 2 and abs(eng.SRER[0] - g["SRER"][0]) < TOL_SRER_DB
    assert abs(eng.SRER[1] - g["SRER"][1]) < TOL_SRER_NYQUIST_DB
    K = plan.Kmax
    rs, rec = g["recsum0"], seen[0]
    assert abs(np.count_nonzero(rec[:, :K]) - int(rs[0])) <= 2
    assert abs(rec[:, :K].sum() - rs[1]) <= 1e-7 * abs(rs[1]) and abs(rec[:, K:2 * K].sum() - rs[2]) <= 1e-7 * abs(rs[2])
    assert abs(rec[:, 3 * K].sum() - rs[4]) <= 1e-7
    fin = eng.final_arrays()                       # the kept result is adaptation 0's
    dec = int(g["s_recon_decim"])
    assert np.abs(fin["s_recon"][::dec] - g["s_recon_every"]).max() <= 1e-9


# ----------------------------------------------------------------------------- hot-kernel raw LS solutions
def _plan_for(s, fs, track, partials=0):
    from eaqhm_amd import prologue
    from eaqhm_amd.engine import FramePlan
    grid = prologue.resample_track(track, np.arange(0, len(s) - 1, round(fs * 5 / 1000)) / fs)
    frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "female")
    prologue.apply_full_waveform(frames, len(s), 32 * 15)
    return FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, partials)


def _raw_of(eng, f):
    """(amplitudes, slopes) of frame f as eaqhm_ls_batch left them in raw_amp / raw_slope (column order
    [negative | DC | positive], the order iqhmLS_complexamps / eaqhmLS_complexamps return)."""
    amp = eng.raw[0][f].cpu().numpy().view(np.complex128)
    slo = eng.raw[1][f].cpu().numpy().view(np.complex128)
    return amp, slo


@pytest.mark.parametrize("variant", [3, 2])
def test_batched_kernels_raw_solutions_sa19(amd, sa19_golden, variant):
    """The BATCHED LS kernels (eaqhm_ls_tile_kernel for variant 3, eaqhm_ls_mfma_kernel for variant 2 — not the
    one-frame seam kernel) against the reference's per-frame LS outputs: complex amplitudes AND slopes of all Kc
    columns, negative-frequency block included, for four frames of adaptation 0 and four of adaptation 1."""
    from eaqhm_amd import prologue
    from eaqhm_amd.engine import DeviceAnalysis
    g = sa19_golden
    fs, s = prologue.read_signal(os.path.join(GOLDEN, "SA19.WAV"))
    plan = _plan_for(s, fs, g["swipe_track"])
    eng = DeviceAnalysis(s, s, plan, 160, 1, keep_raw=True)
    eng.ctx.set_option(1, variant)
    worst = {}

    def hook(a, e):
        for idx in (0, 700, 2000, 3500):
            p = ("iqhm%d_" if a == 0 else "eaqhm%d_") % idx
            assert int(g[p + "tith"]) - 1 == int(plan.frame_c[idx])
            amp, slo = _raw_of(e, idx)
            Kc = len(g[p + "amp"])
            worst[(a, idx)] = (relerr(amp[:Kc], g[p + "amp"]), relerr(slo[:Kc], g[p + "slope"]))

    eng.run(on_adaptation=hook)
    assert len(worst) == 8
    for (a, idx), (ea, es) in worst.items():
        assert ea < 1e-9 and es < 1e-8, (a, idx, ea, es)


def test_batched_kernel_raw_solutions_48khz(amd):
    """Large frames (N up to 901, Kc 161-297: eaqhm_ls_mfma_kernel): raw LS solutions of adaptation 0 against the
    reference on the full-band 48 kHz fixture (frame 100) and, with partials=80, of adaptations 0 AND 1 (mode 1 of
    the large-frame kernel at Kc = 161); then the frame-centre records of adaptation 1 and the SRER list, which
    the reference ends after adaptation 1 (39.81 -> 39.72 dB)."""
    from eaqhm_amd.engine import DeviceAnalysis
    fs = 48000
    g = load_golden("synth48k_0p6s_adpt1.npz")
    s = g["wav_int16"] / 32768.0
    plan = _plan_for(s, fs, g["swipe_track"])
    eng = DeviceAnalysis(s, s, plan, 160, 0, keep_raw=True)
    eng.run()
    amp, slo = _raw_of(eng, 100)
    Kc = len(g["iqhm100_amp"])
    assert relerr(amp[:Kc], g["iqhm100_amp"]) < 1e-9 and relerr(slo[:Kc], g["iqhm100_slope"]) < 1e-8

    g = load_golden("synth48k_0p6s_p80_adpt2.npz")
    s = g["wav_int16"] / 32768.0
    plan = _plan_for(s, fs, g["swipe_track"], partials=80)
    assert plan.Kmax == 80 and np.array_equal(2 * plan.frame_K + 1, g["ls_shapes_iqhm"][:, 1])
    n0 = plan.n_frames
    eng = DeviceAnalysis(s, s, plan, 160, 2, keep_raw=True)
    seen = {}

    def hook(a, e):
        for idx in ((100, 900) if a == 0 else (100, 900, 1500)):
            key = "iqhm%d_" % idx if a == 0 else "eaqhm%d_" % idx
            if a > 0:
                assert int(g[key + "a"]) == 1 and idx < n0
            assert int(g[key + "tith"]) - 1 == int(plan.frame_c[idx])
            amp, slo = _raw_of(e, idx)
            Kc = len(g[key + "amp"])
            assert relerr(amp[:Kc], g[key + "amp"]) < 1e-9, (a, idx)
            assert relerr(slo[:Kc], g[key + "slope"]) < 1e-8, (a, idx)
        seen[a] = e.records[0][:plan.No_ti].cpu().numpy().copy()

    eng.run(on_adaptation=hook)
    assert len(eng.SRER) == 2 and np.abs(np.array(eng.SRER) - g["SRER"]).max() < TOL_SRER_DB
    ref = unpack_records(g, 1)
    K = plan.Kmax
    am, fm, ph = seen[1][:, :K], seen[1][:, K:2 * K], seen[1][:, 2 * K:3 * K]
    assert np.mean((am != 0) == ref["mask"]) >= 0.999
    both = (am != 0) & ref["mask"]
    assert np.abs(am[both] - ref["am"][both]).max() <= TOL_AM_REL * ref["am"].max()
    assert np.abs(fm[both] - ref["fm"][both]).max() <= TOL_FM_HZ
    strong = both & (ref["am"] > 1e-6 * ref["am"].max())
    assert np.abs(wrap(ph[strong] - ref["ph"][strong])).max() <= TOL_PH_RAD
    assert np.abs(eng.final_arrays()["s_recon"] - g["s_recon"]).max() <= 1e-9


def test_low_voice_16khz_large_frames_against_reference(amd):
    """A low (`male`) voice at 16 kHz through the reference itself (tests/golden/make_golden.py male16k_2s): windows of up
    to 565 samples and up to 185 basis columns (systems of up to 24 tile rows) — every frame goes to the large-frame
    kernels (eaqhm_ls_a0big_kernel / eaqhm_ls_mfma_kernel) at 16 kHz, adaptations 0-2: the SRER list, raw LS solutions
    of captured frames, the frame-centre records of adaptation 1, the reconstruction."""
    from eaqhm_amd.engine import DeviceAnalysis
    g = load_golden("male16k_2s_adpt2.npz")
    fs = 16000
    s = g["wav_int16"] / 32768.0
    from eaqhm_amd import prologue
    grid = prologue.resample_track(g["swipe_track"], np.arange(0, len(s) - 1, round(fs * 5 / 1000)) / fs)
    frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "male")
    prologue.apply_full_waveform(frames, len(s), 32 * 15)
    from eaqhm_amd.engine import FramePlan
    plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
    sh = g["ls_shapes_iqhm"]
    assert np.array_equal(2 * plan.frame_wl + 1, sh[:, 0]) and np.array_equal(2 * plan.frame_K + 1, sh[:, 1])
    assert sh[:, 1].max() > 103                      # beyond the on-chip tile kernel (Kc <= 103)
    eng = DeviceAnalysis(s, s, plan, 70, 2, keep_raw=True)
    seen = {}
    n0 = plan.n_frames

    def hook(a, e):
        for idx in ((300,) if a == 0 else (300, 1500)):
            key = "iqhm%d_" % idx if a == 0 else "eaqhm%d_" % idx
            if a > 0 and int(g[key + "a"]) != a:
                continue
            assert int(g[key + "tith"]) - 1 == int(plan.frame_c[idx % n0])
            amp, slo = _raw_of(e, idx % n0)
            Kc = len(g[key + "amp"])
            assert relerr(amp[:Kc], g[key + "amp"]) < 1e-9, (a, idx)
            assert relerr(slo[:Kc], g[key + "slope"]) < 1e-8, (a, idx)
        seen[a] = e.records[0][:plan.No_ti].cpu().numpy().copy()

    eng.run(on_adaptation=hook)
    record_measurement("male16k_2s", srer_hip=[float(v) for v in eng.SRER], srer_reference=[float(v) for v in g["SRER"]],
                       Kc_max=int(sh[:, 1].max()), N_max=int(sh[:, 0].max()))
    # the reference stops after adaptation 1 (54.69 -> 16.13 dB): with partials up to 7.7 kHz of 8 kHz this input has the
    # near-Nyquist collapse of SURVEY Q14 too; adaptation 0 to 1e-6 dB, adaptation 1 within TOL_SRER_NYQUIST_DB
    assert len(eng.SRER) == len(g["SRER"]) == 2 and abs(eng.SRER[0] - g["SRER"][0]) < TOL_SRER_DB
    assert abs(eng.SRER[1] - g["SRER"][1]) < TOL_SRER_NYQUIST_DB
    ref = unpack_records(g, 1)
    K = plan.Kmax
    am, fm, ph = seen[1][:, :K], seen[1][:, K:2 * K], seen[1][:, 2 * K:3 * K]
    assert np.mean((am != 0) == ref["mask"]) >= 0.999
    both = (am != 0) & ref["mask"]
    dam = np.abs(am[both] - ref["am"][both]) / ref["am"].max()
    dfm = np.abs(fm[both] - ref["fm"][both])
    strong = both & (ref["am"] > 1e-6 * ref["am"].max())
    dph = np.abs(wrap(ph[strong] - ref["ph"][strong]))
    inst = np.nonzero(both)[0]
    bad_inst = np.unique(inst[dam > TOL_AM_REL])
    record_measurement("male16k_2s_records_adaptation1", cells=int(both.sum()), am_max=float(dam.max()),
                       am_q999=float(np.quantile(dam, 0.999)), fm_max_hz=float(dfm.max()),
                       fm_q999_hz=float(np.quantile(dfm, 0.999)), ph_max_rad=float(dph.max()),
                       ph_q999_rad=float(np.quantile(dph, 0.999)), instants_beyond_am_tol=int(len(bad_inst)),
                       first_bad_instants=[int(v) for v in bad_inst[:12]])
    # Frequencies and phases meet the bars of SURVEY 8c outright (measured: 3.2e-4 Hz, 5.8e-6 rad), and so do the amplitudes
    # of all but the last five analysed instants of the file (2054-2058: the noise-only fade-out, 223 columns fitted to a
    # signal 45 dB down, the bars were stated for cond(R) <= 1.1e5), which stay within 1e-6 of the largest amplitude
    # (measured 6.4e-7; profiles/r03_parity/parity_measurements.json)
    assert dfm.max() <= TOL_FM_HZ and dph.max() <= TOL_PH_RAD
    assert len(bad_inst) <= 8 and (len(bad_inst) == 0 or bad_inst.min() >= plan.No_ti - 100) and dam.max() <= 1e-6
    assert np.abs(eng.final_arrays()["s_recon"] - g["s_recon"]).max() <= 1e-9


def test_high_voice_16khz_small_systems_against_reference(amd):
    """A high (`child`) voice at 16 kHz through the reference itself (tests/golden/make_golden.py child16k_2s): 14 partials,
    the smallest systems the on-chip tile kernel sees (few tile rows), five adaptations that all improve: the SRER list,
    raw LS solutions of a captured frame of adaptations 0 and 1, the frame-centre records of adaptation 1, the
    reconstruction."""
    from eaqhm_amd import prologue
    from eaqhm_amd.engine import DeviceAnalysis, FramePlan
    g = load_golden("child16k_2s_adpt4.npz")
    fs = 16000
    s = g["wav_int16"] / 32768.0
    grid = prologue.resample_track(g["swipe_track"], np.arange(0, len(s) - 1, round(fs * 5 / 1000)) / fs)
    frames, fstep = prologue.voiced_unvoiced_frames(s, fs, "child")
    prologue.apply_full_waveform(frames, len(s), 32 * 15)
    plan = FramePlan(len(s), fs, grid, frames, fstep, 15, 3, 32, 0)
    sh = g["ls_shapes_iqhm"]
    assert np.array_equal(2 * plan.frame_wl + 1, sh[:, 0]) and np.array_equal(2 * plan.frame_K + 1, sh[:, 1])
    eng = DeviceAnalysis(s, s, plan, 300, 4, keep_raw=True)
    seen = {}

    def hook(a, e):
        if a <= 1:
            key = "iqhm300_" if a == 0 else "eaqhm300_"
            assert int(g[key + "tith"]) - 1 == int(plan.frame_c[300])
            amp, slo = _raw_of(e, 300)
            Kc = len(g[key + "amp"])
            assert relerr(amp[:Kc], g[key + "amp"]) < 1e-9 and relerr(slo[:Kc], g[key + "slope"]) < 1e-8, a
        seen[a] = e.records[0][:plan.No_ti].cpu().numpy().copy()

    eng.run(on_adaptation=hook)
    record_measurement("child16k_2s", srer_hip=[float(v) for v in eng.SRER], srer_reference=[float(v) for v in g["SRER"]],
                       Kc_min=int(sh[:, 1].min()), Kc_max=int(sh[:, 1].max()))
    assert len(eng.SRER) == len(g["SRER"]) == 5 and np.abs(np.array(eng.SRER) - g["SRER"]).max() < TOL_SRER_DB
    ref = unpack_records(g, 1)
    K = plan.Kmax
    am, fm, ph = seen[1][:, :K], seen[1][:, K:2 * K], seen[1][:, 2 * K:3 * K]
    assert np.mean((am != 0) == ref["mask"]) >= 0.999
    both = (am != 0) & ref["mask"]
    assert np.abs(am[both] - ref["am"][both]).max() <= TOL_AM_REL * ref["am"].max()
    assert np.abs(fm[both] - ref["fm"][both]).max() <= TOL_FM_HZ
    strong = both & (ref["am"] > 1e-6 * ref["am"].max())
    assert np.abs(wrap(ph[strong] - ref["ph"][strong])).max() <= TOL_PH_RAD
    assert np.abs(eng.final_arrays()["s_recon"] - g["s_recon"]).max() <= 1e-9


# ----------------------------------------------------------------------------- empty-row seeding (Q7 / Q8)
def check_seeding_result(g, srer, seen, fin, raw510=None):
    """Shared by the single-GPU and the two-rank test: everything the reference's run on the signal with the span of
    digital zeros pins (tests/golden/make_golden.py seed16k)."""
    assert len(srer) == 4 and np.abs(np.array(srer) - g["SRER"]).max() < TOL_SRER_DB
    assert np.abs(fin["s_recon"] - g["s_recon"]).max() <= 1e-9
    K = fin["am"].shape[1]
    for a in (1, 2):
        ref = unpack_records(g, a)
        am, fm, ph = seen[a][:, :K], seen[a][:, K:2 * K], seen[a][:, 2 * K:3 * K]
        assert np.mean((am != 0) == ref["mask"]) >= 0.999
        both = (am != 0) & ref["mask"]
        assert np.abs(am[both] - ref["am"][both]).max() <= TOL_AM_REL * ref["am"].max()
        assert np.abs(fm[both] - ref["fm"][both]).max() <= TOL_FM_HZ
        assert np.abs(wrap(ph[both] - ref["ph"][both])).max() <= TOL_PH_RAD
        assert np.abs(seen[a][:, 3 * K] - ref["a0"]).max() <= TOL_AM_REL
        # the seeded instants carry nothing (their frames see a silent signal)
        inst = (g["seeded_tith_a%d" % a] - 1) // 15
        assert len(inst) == 170 and np.all(am[inst] == 0)
    for a in range(4):
        gs = g["recsum%d" % a]
        assert abs(np.count_nonzero(seen[a][:, :K]) - int(gs[0])) <= 2
        assert abs(seen[a][:, :K].sum() - gs[1]) <= 1e-7 * abs(gs[1])
    # Q8: the rejected adaptation 3 seeded its rows inside the arrays the kept result aliases
    cells, ref_am = g["det_cells"], g["det_am"]
    i, k = cells[:, 0], cells[:, 1]
    ref_mask = np.zeros_like(fin["am"], dtype=bool)
    ref_mask[i, k] = True
    assert np.mean((fin["am"] != 0) == ref_mask) >= 0.999
    q8 = ref_am == 10e-4
    assert q8.sum() == 170
    assert np.all(fin["am"][i[q8], 0] == 10e-4) and np.all(fin["fm"][i[q8], 0] == 0) and np.all(fin["pk"][i[q8], 0] == 0)
    assert np.array_equal(np.flatnonzero(fin["am"][:, 0] == 10e-4), i[q8])
    ok = fin["am"][i, k] != 0
    assert np.abs(fin["am"][i, k][ok] - ref_am[ok]).max() <= TOL_AM_REL * ref_am.max()
    assert np.abs(fin["fm"][i, k][ok] - g["det_fm"][ok]).max() <= TOL_FM_HZ
    assert np.abs(wrap(fin["pk"][i, k][ok] - g["det_pk"][ok])).max() <= TOL_PH_RAD
    if raw510 is not None:      # LS of the first seeded frame of adaptation 1: K = 1, columns [-140 Hz | DC | +140 Hz]
        scale = max(np.abs(g["eaqhm510_amp"]).max(), 1e-300)
        assert np.abs(raw510[0][:3] - g["eaqhm510_amp"]).max() <= 1e-9 * scale + 1e-18


@pytest.mark.parametrize("variant", [3, 2])
def test_empty_row_seeding_against_reference(amd, variant):
    """functions.py:204-242, :286-292 and the aliasing on break :383, :397-402 (Q8): 175 ms of digital
    zeros inside the analysed region; 170 frames per adaptation take the seeding branch from adaptation 1 on and the
    stop rule fires at adaptation 3.  Both batched LS kernels."""
    from eaqhm_amd import functions as F
    from eaqhm_amd.engine import DeviceAnalysis
    g = load_golden("seed16k_1p2s_adpt6.npz")
    s = g["wav_int16"] / 32768.0
    plan = _plan_for(s, 16000, g["swipe_track"])
    assert np.array_equal(plan.ti[plan.analysed], g["ti_a0"])
    eng = DeviceAnalysis(s, s, plan, 160, 6, keep_raw=True)
    eng.ctx.set_option(1, variant)
    seen, raw = {}, {}

    def hook(a, e):
        seen[a] = e.records[0][:plan.No_ti].cpu().numpy().copy()
        if a == 1:
            assert int(g["eaqhm510_tith"]) - 1 == int(plan.frame_c[510])
            raw[510] = _raw_of(e, 510)
            seeded = np.flatnonzero(e.seeded.cpu().numpy()) + 1
            assert np.array_equal(seeded, g["seeded_tith_a1"])

    eng.run(on_adaptation=hook)
    fin = eng.final_arrays()
    check_seeding_result(g, eng.SRER, seen, fin, raw[510])
    det = F.pack_results(plan, fin)
    d = det[int(g["det_cells"][g["det_am"] == 10e-4][0, 0])]
    assert len(d.amplitudes) == 1 and d.amplitudes[0][0] == 10e-4 and d.frange[0][0] == 0
    assert np.array_equal([len(x.amplitudes) if x.isVoiced else 0 for x in det], g["det_len"])


# ----------------------------------------------------------------------------- singular systems, CLI
def test_singular_system_raises_linalgerror(amd):
    """The reference aborts with numpy.linalg.LinAlgError from inv() on a singular normal matrix
    (functions.py:465, :530).  Two identical frequency columns: the seams raise instead of returning a silently
    wrong solution; a healthy system still goes through."""
    rng = np.random.default_rng(1)
    N = 241
    s = rng.standard_normal(N) * 0.1
    with pytest.raises(np.linalg.LinAlgError):
        amd.iqhmLS_complexamps(s, np.array([-200.0, 0.0, 200.0, 200.0]), np.blackman(N), 16000)
    fm = np.tile(np.array([-200.0, 0.0, 200.0, 200.0]), (N, 1))
    with pytest.raises(np.linalg.LinAlgError):
        amd.eaqhmLS_complexamps(s, np.ones((N, 4)), fm, np.hamming(N), 16000)
    a, b = amd.iqhmLS_complexamps(s, np.array([-200.0, 0.0, 200.0]), np.blackman(N), 16000)
    assert np.all(np.isfinite(a)) and np.all(np.isfinite(b))


def _normal_equations(s, am, fm, w, fs):
    """functions.py:498-530 restated for a handful of columns: (R, arr) of the eaQHM least squares."""
    N = len(s)
    mid = (N - 1) // 2
    n = np.arange(N) - mid
    F = np.cumsum(fm, axis=0)
    F = F - F[mid]
    E2 = (1e-4 + am) / (am[mid] + 1e-4) * np.exp(2j * np.pi * F / fs)
    E = np.concatenate((E2, n[:, None] * E2), axis=1)
    Ew = E * w[:, None]
    return Ew.conj().T @ Ew, Ew.conj().T @ (w * s)


def test_ill_conditioned_but_nonsingular_system_is_solved_like_inv(amd):
    """Where the line is drawn.  The reference's inv() returns whatever an ill-conditioned system gives;
    the kernels raise only when the Cholesky factorisation BREAKS DOWN (pivot <= 2.5e-13 = order x eps of its diagonal
    entry: exactly duplicated columns), not because a system is ill-conditioned.  Two columns 200 Hz and 200 Hz + df:
    at df = 1 / 0.3 Hz the smallest pivots of the scaled system are 4e-9 / 2e-10 (cond(R) 1e10 - 1e12, four to six
    orders beyond any frame of the fixtures): NumPy's inv() returns there and so must the kernel, with a residual of the
    normal equations no worse than inv()'s; df = 0 still raises.  (A pivot placed exactly between the round-2 threshold
    1e-12 and the breakdown threshold cannot be produced through this seam: the slope copy n*E2 of a near-duplicate
    column squares the loss, and below ~1e-10 the last pivot is rounding noise in NumPy as well.)"""
    rng = np.random.default_rng(3)
    N, fs = 241, 16000
    s = rng.standard_normal(N) * 0.1
    w = np.hamming(N)
    am = np.ones((N, 4))
    seen = {}
    for df in (1.0, 0.3):
        fm = np.tile(np.array([-200.0, 0.0, 200.0, 200.0 + df]), (N, 1))
        R, arr = _normal_equations(s, am, fm, w, fs)
        x_ref = np.linalg.inv(R) @ arr                                   # functions.py:530 — returns, no exception
        a, b = amd.eaqhmLS_complexamps(s, am, fm, w, fs)                 # must return too
        x = np.concatenate((a.ravel(), b.ravel()))
        assert np.all(np.isfinite(x))
        res = np.linalg.norm(R @ x - arr) / np.linalg.norm(arr)
        res_ref = np.linalg.norm(R @ x_ref - arr) / np.linalg.norm(arr)
        seen["df_%g_hz" % df] = dict(cond=float(np.linalg.cond(R)), residual_kernel=float(res), residual_inv=float(res_ref),
                                     rel_distance=float(np.abs(x - x_ref).max() / np.abs(x_ref).max()))
        assert res <= 10 * res_ref + 1e-9
    record_measurement("ill_conditioned_seam", **seen)
    with pytest.raises(np.linalg.LinAlgError):
        amd.eaqhmLS_complexamps(s, am, np.tile(np.array([-200.0, 0.0, 200.0, 200.0]), (N, 1)), w, fs)


@pytest.mark.parametrize("variant", [3, 2])
def test_frames_outside_the_track_window_are_dropped_not_read(amd, sa19_golden, variant):
    """The contract of eaqhm_ls_batch — every frame window inside the resident track window — checked on the device: a
    caller that hands a window too small gets the offending frames dropped and counted (eaqhm_ls_faults[2], ValueError from
    the engine), not an out-of-bounds read.  Both LS kernels."""
    from eaqhm_amd import prologue
    from eaqhm_amd.engine import DeviceAnalysis
    fs, s = prologue.read_signal(os.path.join(GOLDEN, "SA19.WAV"))
    s = s[:16000]
    plan = _plan_for(s, fs, sa19_golden["swipe_track"][:1000])
    eng = DeviceAnalysis(s, s, plan, 160, 1)
    eng.ctx.set_option(1, variant)
    it = eng.adaptations()
    next(it)
    next(it)                                    # adaptation 0 read, adaptation 1 enqueued (healthy)
    eng.torch.cuda.synchronize()
    assert eng.ctx.ls_faults() == (0, 0, 0)
    p, c, T = plan, eng.ctx, eng.torch
    # the tracks are adaptation 1's by now: slot lists of THESE tracks, then the same launch twice — whole file / window
    ncol, cols = T.zeros_like(eng.ncol), T.zeros_like(eng.cols)
    seeded, any_seed = T.zeros_like(eng.seeded), T.zeros_like(eng.any_seed)
    c.frame_prep(eng.fm_cur, p.L, 0, p.L, p.Kmax, eng.frame_c, eng.nf, ncol, cols, seeded, any_seed)

    def launch(am, fm, lo, w):
        rec = T.zeros_like(eng.records[0])
        c.ls_batch(1, eng.s, p.L, p.fs, am, fm, lo, w, p.Kmax, eng.frame_inst, eng.frame_c, eng.frame_wl, eng.frame_f0,
                   eng.frame_K, ncol, cols, seeded, any_seed, eng.nf, p.wl_max, 2, p.f0_stale, eng.f0min, rec)
        T.cuda.synchronize()
        return rec[:plan.No_ti].cpu().numpy(), c.ls_faults()

    ref, faults = launch(eng.am_cur, eng.fm_cur, 0, p.L)
    assert faults == (0, 0, 0) and ref[plan.frame_inst].any(axis=1).all()
    lo, w = 4000, 6000                          # a resident window [4000, 10000): most frames lie outside it
    got, faults = launch(eng.am_cur[:, lo:lo + w].contiguous(), eng.fm_cur[:, lo:lo + w].contiguous(), lo, w)
    fc, fw = plan.frame_c.astype(np.int64), plan.frame_wl.astype(np.int64)
    inside = (fc - fw - 1 >= lo) & (fc + fw < lo + w)
    assert inside.sum() > 100 and (~inside).sum() > 100
    assert faults == (0, 0, int((~inside).sum()))
    rows = plan.frame_inst
    assert not got[rows[~inside]].any()                       # dropped frames wrote nothing
    assert np.array_equal(got[rows[inside]], ref[rows[inside]])  # the others are what they are in the whole-file launch


def test_singular_frame_in_batch_raises(amd, sa19_golden):
    """Two slots with identical tracks make two columns of every adaptation >= 1 frame identical: the batched kernel
    counts the collapsed pivots (eaqhm_ls_faults) and the engine raises LinAlgError at the end of that adaptation
    (the reference would abort inside the first such frame's inv())."""
    from eaqhm_amd import prologue
    from eaqhm_amd.engine import DeviceAnalysis
    fs, s = prologue.read_signal(os.path.join(GOLDEN, "SA19.WAV"))
    s = s[:16000]
    plan = _plan_for(s, fs, sa19_golden["swipe_track"][:1000])
    eng = DeviceAnalysis(s, s, plan, 160, 2)
    it = eng.adaptations()
    next(it)                                    # adaptation 0 enqueued
    next(it)                                    # its SRER read (healthy), adaptation 1 enqueued
    eng.torch.cuda.synchronize()
    eng.fm_cur[1].copy_(eng.fm_cur[0])          # slot 1 := slot 0 for the next adaptation's windows
    eng.am_cur[1].copy_(eng.am_cur[0])
    with pytest.raises(np.linalg.LinAlgError):
        for _ in it:
            pass


def test_cli_writes_reconstruction(amd, tmp_path, sa19_golden, capsys):
    """Headless counterpart of main.py:44-72: prints the SRER lines, writes <name>_reconstructed.wav as float32."""
    import shutil
    from scipy.io import wavfile
    from eaqhm_amd import cli
    wav = str(tmp_path / "SA19.WAV")
    shutil.copy(os.path.join(GOLDEN, "SA19.WAV"), wav)
    assert cli.main([wav, "--gender", "female", "--max-adpt", "1"]) == 0
    out = capsys.readouterr().out
    assert "SRER: " in out and "Adaptation No: 1" in out and "Signal adapted to" in out
    fs, x = wavfile.read(str(tmp_path / "SA19_reconstructed.wav"))
    assert fs == 16000 and x.dtype == np.float32 and x.shape == (63488,)
    ref = sa19_golden["SRER"][1]
    fs0, s0 = wavfile.read(wav)
    s0 = s0 / 32768.0
    assert abs(20 * np.log10(np.std(s0) / np.std(s0 - x)) - ref) < 1e-3      # float32 file, SWIPE' run by the CLI


def test_synth16k_checksums_every_adaptation(amd):
    """The 2 s synthetic 16 kHz fixture: record and dense-state checksums the reference left after EVERY adaptation
    (recsum0..3, densesum0..3: sums and non-zero counts of am_recon, fm_current, s_recon_tmp), incl. the rejected one."""
    from eaqhm_amd.engine import DeviceAnalysis
    g = load_golden("synth16k_2s_adpt3.npz")
    s = g["wav_int16"] / 32768.0
    plan = _plan_for(s, 16000, g["swipe_track"])
    eng = DeviceAnalysis(s, s, plan, 160, 3)
    seen = {}

    def hook(a, e):
        rec = e.records[0][:plan.No_ti].cpu().numpy()
        K = plan.Kmax
        am, fm = e.am_cur.cpu().numpy(), e.fm_cur.cpu().numpy()
        seen[a] = (np.count_nonzero(rec[:, :K]), rec[:, :K].sum(), rec[:, K:2 * K].sum(), rec[:, 3 * K].sum(),
                   am.sum(), fm.sum(), e.s_hat[0].cpu().numpy().sum(), np.count_nonzero(am), np.count_nonzero(fm))

    eng.run(on_adaptation=hook)
    assert len(eng.SRER) == 4 and np.abs(np.array(eng.SRER) - g["SRER"]).max() < TOL_SRER_DB
    for a in range(4):
        rs, ds, got = g["recsum%d" % a], g["densesum%d" % a], seen[a]
        assert abs(got[0] - int(rs[0])) <= 2
        assert abs(got[1] - rs[1]) <= 1e-7 * abs(rs[1]) and abs(got[2] - rs[2]) <= 1e-7 * abs(rs[2])
        assert abs(got[3] - rs[4]) <= 1e-7
        assert abs(got[4] - ds[0]) <= 1e-7 * abs(ds[0])            # dense am_recon
        assert abs(got[5] - ds[3]) <= 1e-6 * abs(ds[3])            # fm_current (derivative of the unwrapped phase)
        assert abs(got[6] - ds[5]) <= 1e-7 * max(abs(ds[5]), 1.0)  # s_recon_tmp
        assert abs(got[7] - int(ds[6])) <= 40 and abs(got[8] - int(ds[7])) <= 40
