"""CPU tests: the oracle (oracle/eaqhm_oracle.py) against golden vectors produced by the
reference itself (tests/golden/make_golden.py, run in the build container).

Tolerances are the FP64 bars of SURVEY.md §8(c): am <= 1e-8*max, fm <= 1e-3 Hz, phase <= 1e-5 rad
(mod 2*pi), identical acceptance mask, SRER <= 1e-6 dB — the oracle actually lands many orders
of magnitude inside them, and the asserts below use the tighter measured margins.
"""
import numpy as np
import pytest

import eaqhm_oracle as O
from conftest import load_golden, unpack_records

README_SRER = [17.86520945273994, 24.431728752204954, 24.67698055430504, 25.291120491477024,
               25.497403658214047, 25.446628776435006]     # /root/reference/img/SA19out.JPG


def wrap(d):
    return (d + np.pi) % (2 * np.pi) - np.pi


def test_golden_matches_readme_screenshot(sa19_golden):
    """The fixture itself reproduces the only numbers the reference publishes."""
    assert np.allclose(sa19_golden["SRER"], README_SRER, rtol=0, atol=1e-9)
    assert len(sa19_golden["s_recon"]) == 63488


def test_unit_seams():
    u = load_golden("unit_vectors.npz")
    a, b = O.iqhm_ls(u["iq_s"], u["iq_f0range"], u["iq_w"], 16000)
    assert np.abs(a - u["iq_amp"]).max() <= 1e-13 * np.abs(a).max()
    assert np.abs(b - u["iq_slope"]).max() <= 1e-13 * np.abs(b).max()
    a, b = O.eaqhm_ls(u["ea_s"], u["ea_am"], u["ea_fm"], u["ea_w"], 16000)
    assert np.abs(a - u["ea_amp"]).max() <= 1e-13 * np.abs(a).max()
    assert np.abs(b - u["ea_slope"]).max() <= 1e-13 * np.abs(b).max()
    p = O.phase_integr_interpolation(u["pii_fm"], u["pii_ph"], u["pii_knots"])
    assert np.abs(p - u["pii_out"]).max() <= 1e-14
    p = O._phase_integr_uniform(u["pii_fm"], u["pii_ph"], u["pii_knots"], 15)
    assert np.abs(p - u["pii_out"]).max() <= 1e-14
    assert np.abs(O.get_linear(u["gl_v"], u["gl_t"]) - u["gl_out"]).max() == 0
    assert np.abs(O.medfilt_ref(u["mf_x"], 5) - u["mf_out"]).max() == 0


def test_ls_frames_from_sa19(sa19_golden):
    g = sa19_golden
    for idx in (0, 700, 2000, 3500):
        p = "iqhm%d_" % idx
        a, b = O.iqhm_ls(g[p + "s"], g[p + "f0range"], g[p + "window"], int(g[p + "fs"]))
        assert np.abs(a - g[p + "amp"]).max() <= 1e-12 * np.abs(a).max()
        assert np.abs(b - g[p + "slope"]).max() <= 1e-12 * np.abs(b).max()
        p = "eaqhm%d_" % idx
        a, b = O.eaqhm_ls(g[p + "s"], g[p + "am"], g[p + "fm"], g[p + "window"], int(g[p + "fs"]))
        assert np.abs(a - g[p + "amp"]).max() <= 1e-12 * np.abs(a).max()
        assert np.abs(b - g[p + "slope"]).max() <= 1e-12 * np.abs(b).max()


def test_preprocessing_sa19(sa19_golden, sa19_signal):
    g = sa19_golden
    fs, s = sa19_signal
    ti5, sp, vo, fstep = O.voiced_unvoiced_frames(s, fs, "female")
    assert np.array_equal(ti5, g["vuv_ti"]) and fstep == int(g["frame_step"])
    assert np.array_equal(sp, g["vuv_isSpeech"]) and np.array_equal(vo, g["vuv_isVoiced"])
    f0s = O.get_linear(g["swipe_track"], np.arange(0, len(s) - 1, round(fs * 5 / 1000)) / fs)
    assert np.abs(f0s - g["f0s_5ms"]).max() == 0


def test_voiced_only_run(sa19_signal):
    """fullWaveform=False path (functions.py:127-138): SRER target masking + voicing flags."""
    g = load_golden("sa19_female_voicedonly_adpt1.npz")
    fs, s = sa19_signal
    r = O.analyse(s, fs, g["f0s_5ms"], g["vuv_ti"], g["vuv_isSpeech"], g["vuv_isVoiced"], int(g["frame_step"]),
                  f0min=160, maxAdpt=1, fullWaveform=False)
    assert np.abs(np.array(r["SRER"]) - g["SRER"]).max() < 1e-9
    assert np.array_equal(r["isSpeech"], g["det_isSpeech"]) and np.array_equal(r["isVoiced"], g["det_isVoiced"])
    assert np.abs(r["s_recon"] - g["s_recon"]).max() < 1e-11


def test_every_option_off_its_default():
    """Tuple gender, step, pitchPeriods, analysisWindow, fullWaveform=False, fc > 0 and partials > 0 at once
    (tests/golden/make_golden.py options16k): the oracle, fed the pre-processing outputs of the reference's run, against
    the SRER of its four adaptations, its reconstruction and its flags."""
    g = load_golden("options16k_1p5s.npz")
    s = O.ellip_filter(g["wav_int16"] / 32768.0, 16000, 60)                              # functions.py:90-91
    r = O.analyse(s, 16000, g["f0s_5ms"], g["vuv_ti"], g["vuv_isSpeech"], g["vuv_isVoiced"], int(g["frame_step"]),
                  f0min=150, maxAdpt=3, step=12, pitchPeriods=4, analysisWindow=40, fullWaveform=False, partials=25)
    assert len(r["SRER"]) == 4 and np.abs(np.array(r["SRER"]) - g["SRER"]).max() < 1e-9
    assert np.array_equal(r["isSpeech"], g["det_isSpeech"]) and np.array_equal(r["isVoiced"], g["det_isVoiced"])
    assert np.abs(r["s_recon"] - g["s_recon"]).max() < 1e-11


def test_synth16k_run():
    g = load_golden("synth16k_2s_adpt3.npz")
    s = g["wav_int16"] / 32768.0
    r = O.analyse(s, 16000, g["f0s_5ms"], g["vuv_ti"], g["vuv_isSpeech"], g["vuv_isVoiced"], int(g["frame_step"]),
                  f0min=160, maxAdpt=3)
    assert np.abs(np.array(r["SRER"]) - g["SRER"]).max() < 1e-9
    assert np.abs(r["s_recon"] - g["s_recon"]).max() < 1e-11


def test_seeding_run():
    """Empty-row seeding (functions.py:204-242, :286-292) and its aliasing into the kept result when the
    stop rule fires (:383, :397-402; Q8): 1.2 s of synthetic speech with 175 ms of digital zeros.  170 frames per
    adaptation take the branch from adaptation 1 on; the loop breaks at adaptation 3 with the seeds of that
    adaptation already written into the arrays the result of adaptation 2 aliases."""
    g = load_golden("seed16k_1p2s_adpt6.npz")
    s = g["wav_int16"] / 32768.0
    assert np.all(s[g["zero_span"][0]:g["zero_span"][1]] == 0)
    seen, seeded = {}, {}
    an = O.Analysis(s, 16000, g["f0s_5ms"], g["vuv_ti"], g["vuv_isSpeech"], g["vuv_isVoiced"], int(g["frame_step"]),
                    f0min=160, maxAdpt=6)
    for a in range(7):
        rec = an.ls_stage(a)
        seeded[a] = np.array(an.seeded) + 1
        seen[a] = rec
        an.post_stage(a, rec)
        if an.done:
            break
    r = an.result()
    assert len(r["SRER"]) == 4 and np.abs(np.array(r["SRER"]) - g["SRER"]).max() < 1e-9
    assert np.abs(r["s_recon"] - g["s_recon"]).max() < 1e-11
    for a in (1, 2, 3):
        assert np.array_equal(seeded[a], g["seeded_tith_a%d" % a]) and len(seeded[a]) == 170
    for a in (1, 2):
        gr = unpack_records(g, a)
        assert np.array_equal(seen[a]["am"] != 0, gr["mask"])
        assert np.abs(seen[a]["am"] - gr["am"]).max() <= 1e-11 * gr["am"].max()
        assert np.abs(seen[a]["fm"] - gr["fm"]).max() <= 1e-4
        assert np.abs(wrap(seen[a]["ph"] - gr["ph"])).max() <= 1e-7
        assert np.abs(seen[a]["a0"] - gr["a0"]).max() <= 1e-12
    for a in range(4):
        gs = g["recsum%d" % a]
        assert np.count_nonzero(seen[a]["am"]) == int(gs[0])
        assert np.allclose([seen[a]["am"].sum(), seen[a]["fm"].sum()], gs[1:3], rtol=1e-9, atol=1e-9)
    # the LS of seeded frames: K = 1, columns [-140 Hz | DC | +140 Hz] built from the seeded track
    for idx in (510, 511, 1726, 2942):
        p = "eaqhm%d_" % idx
        a_, b_ = O.eaqhm_ls(g[p + "s"], g[p + "am"], g[p + "fm"], g[p + "window"], 16000)
        assert np.array_equal(g[p + "slots"], [0, 1, 2])
        assert np.abs(a_ - g[p + "amp"]).max() <= 1e-12 * max(np.abs(a_).max(), 1e-30)
    # returned structs: the 10e-4 entries of slot 0 at the instants seeded during the rejected adaptation
    cells, am = g["det_cells"], g["det_am"]
    i, k = cells[:, 0], cells[:, 1]
    assert np.count_nonzero(r["am"]) == len(cells)
    assert np.abs(r["am"][i, k] - am).max() <= 1e-11
    q8 = am == 10e-4
    assert q8.sum() == 170 and np.all(k[q8] == 0)
    assert np.array_equal(r["ti"][i[q8]] + 1, g["seeded_tith_a3"])
    assert np.all(r["fm"][i[q8], 0] == 0) and np.all(g["det_fm"][q8] == 0)
    assert np.abs(r["fm"][i, k] - g["det_fm"]).max() <= 1e-5
    assert np.abs(wrap(r["pk"][i, k] - g["det_pk"])).max() <= 1e-7


@pytest.mark.slow
def test_synth48k_partials80():
    """48 kHz with partials=80 (no near-Nyquist partials): adaptation 1 is well behaved, so the eaQHM LS at the
    large-frame sizes (Kc = 161, N up to 901) is pinned; the reference stops after adaptation 1 (39.81 -> 39.72 dB)."""
    g = load_golden("synth48k_0p6s_p80_adpt2.npz")
    s = g["wav_int16"] / 32768.0
    seen = {}
    r = O.analyse(s, 48000, g["f0s_5ms"], g["vuv_ti"], g["vuv_isSpeech"], g["vuv_isVoiced"], int(g["frame_step"]),
                  f0min=160, maxAdpt=2, partials=80, on_adaptation=lambda a, rec, st: seen.update({a: rec}))
    assert len(r["SRER"]) == 2 and np.abs(np.array(r["SRER"]) - g["SRER"]).max() < 1e-8
    gr = unpack_records(g, 1)
    assert np.mean((seen[1]["am"] != 0) == gr["mask"]) >= 0.9999
    both = (seen[1]["am"] != 0) & gr["mask"]
    assert np.abs(seen[1]["am"][both] - gr["am"][both]).max() <= 1e-10 * gr["am"].max()
    assert np.abs(seen[1]["fm"][both] - gr["fm"][both]).max() <= 1e-4
    assert np.abs(r["s_recon"] - g["s_recon"]).max() < 1e-10


@pytest.mark.slow
def test_sa19_full_run(sa19_golden, sa19_signal):
    """BASELINE config 1/2: SA19.WAV, 'female', defaults; all six adaptations, every stage pinned."""
    g = sa19_golden
    fs, s = sa19_signal
    seen = {}

    def hook(a, rec, st):
        seen[a] = dict(cnt=np.count_nonzero(rec["am"]), sums=[rec["am"].sum(), rec["fm"].sum(),
                                                             np.abs(rec["ph"]).sum(), rec["a0"].sum()])
        if a in (0, 1):
            gr = unpack_records(g, a, with_fm=(a > 0))
            assert np.array_equal(rec["am"] != 0, gr["mask"])
            assert np.abs(rec["am"] - gr["am"]).max() <= 1e-11 * gr["am"].max()
            assert np.abs(wrap(rec["ph"] - gr["ph"])).max() <= 1e-7   # weak partials: angle is ill-conditioned
            assert np.abs(rec["a0"] - gr["a0"]).max() <= 1e-12
            if a > 0:
                assert np.abs(rec["fm"] - gr["fm"]).max() <= 1e-4      # weak partials; bar is 1e-3 Hz
        if a == 0:
            for k in (0, 30, 45):
                for lo in (0, 30000):
                    p = "dense0_k%d_%d_" % (k, lo)
                    n = len(g[p + "am"])
                    assert np.abs(st["am"][lo:lo + n, k] - g[p + "am"]).max() <= 1e-13
                    assert np.abs(st["fm"][lo:lo + n, k] - g[p + "fm"]).max() <= 1e-8
                    assert np.abs(st["ph"][lo:lo + n, k] - g[p + "ph"]).max() <= 1e-9
                    assert np.abs(st["fm_current"][lo:lo + n, k] - g[p + "fmcur"]).max() <= 1e-6
            assert np.abs(st["a0"][:9000] - g["dense0_a0_head"]).max() <= 1e-13
            assert np.abs(st["a0"][-2000:] - g["dense0_a0_tail"]).max() <= 1e-13
            assert np.abs(st["s_hat"] - g["dense0_srecon"]).max() <= 1e-12

    r = O.analyse(s, fs, g["f0s_5ms"], g["vuv_ti"], g["vuv_isSpeech"], g["vuv_isVoiced"], int(g["frame_step"]),
                  f0min=160, on_adaptation=hook)
    assert len(r["SRER"]) == 6
    assert np.abs(np.array(r["SRER"]) - g["SRER"]).max() < 1e-9
    assert np.abs(np.array(r["SRER"]) - README_SRER).max() < 1e-9
    assert np.abs(r["s_recon"] - g["s_recon"]).max() < 1e-11
    for a in range(6):
        gs = g["recsum%d" % a]
        assert seen[a]["cnt"] == int(gs[0])
        assert np.allclose(seen[a]["sums"], gs[1:], rtol=1e-9, atol=1e-9)
    # returned structs (functions.py:404-411)
    assert np.array_equal(r["ti"], g["det_ti"])
    assert np.array_equal(r["isSpeech"], g["det_isSpeech"]) and np.array_equal(r["isVoiced"], g["det_isVoiced"])
    assert r["n_ls_frames"] == 6 * 4169
    v = g["det_isVoiced"]
    assert np.abs(r["a0"][v] - g["det_a0"][v]).max() <= 1e-12
    cells = g["det_cells"]
    assert np.count_nonzero(r["am"][v]) == len(cells)
    i, k = cells[:, 0], cells[:, 1]
    assert np.abs(r["am"][i, k] - g["det_am"]).max() <= 1e-11
    assert np.abs(r["fm"][i, k] - g["det_fm"]).max() <= 1e-5
    assert np.abs(wrap(r["pk"][i, k] - g["det_pk"])).max() <= 1e-7


@pytest.mark.slow
def test_synth48k_adaptation0():
    """48 kHz (BASELINE config 5 in miniature: N up to 901, Kc up to ~300).  Adaptation 0 only: adaptation 1
    collapses in the reference itself (near-Nyquist unwrap flips) and pins nothing."""
    g = load_golden("synth48k_0p6s_adpt1.npz")
    s = g["wav_int16"] / 32768.0
    seen = {}
    r = O.analyse(s, 48000, g["f0s_5ms"], g["vuv_ti"], g["vuv_isSpeech"], g["vuv_isVoiced"], int(g["frame_step"]),
                  f0min=160, maxAdpt=0, on_adaptation=lambda a, rec, st: seen.update(rec=rec))
    assert abs(r["SRER"][0] - g["SRER"][0]) < 1e-8
    gr = unpack_records(g, 0, with_fm=False)
    assert np.array_equal(seen["rec"]["am"] != 0, gr["mask"])
    assert np.abs(seen["rec"]["am"] - gr["am"]).max() <= 1e-10 * gr["am"].max()
