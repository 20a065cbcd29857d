EAQHM_LIB=tools/alt/tmfinal.so timeout -k 10 200 python tools/class_probe.py synth16k_60s 2>&1 | tail -7 | cut -c1-46
timeout -k 10 300 python tools/ls_ab_probe.py tools/alt/tmfinal.so synth16k_60s 2>&1 | tail -2
timeout -k 10 300 python tools/ls_ab_probe.py tools/alt/tmfinal.so sa19 2>&1 | tail -2
EAQHM_LIB=tools/alt/tmfinal.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
