import sys, numpy as np, time
sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo')
import eaqhm_oracle as O
g=np.load('/root/repo/tests/golden/synth48k_2s_adpt1.npz')
fs=48000
s=g['wav_int16']/32768.0
t0=time.time()
r=O.analyse(s, fs, g['f0s_5ms'], g['vuv_ti'], g['vuv_isSpeech'], g['vuv_isVoiced'], int(g['frame_step']), f0min=160, maxAdpt=1)
print("oracle SRER", [repr(float(v)) for v in r['SRER']], "ref", [repr(float(v)) for v in g['SRER']], time.time()-t0, flush=True)
