#!/usr/bin/env python3
"""Random-shape parity sweep on the GPU beyond the 48 shapes of tests/test_gpu_parity.py: every shape's multiply
(+relinearise / modulus switch), relinearise and rotations against the C oracle (tests/full_size.py).
Test infrastructure (lives in tests/ because it uses the oracle).
Usage: python tests/random_sweep_gpu.py [seconds [first_idx [last_idx [ks_mode [big|f64|f64wide|- [emu]]]]]]   (big: N = 32768 / 65536 only;
f64: N = 4096 ... 16384 with every modulus below 2^50 -- the FP64-FMA kernels, round 6; f64wide: N = 8192 F64 launches of more
than one workgroup per CU -- the 512-thread key-switch instance;
emu: the same sweep on the host emulation of the kernel sources -- CPU CI evidence, no GPU needed)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
from helpers import load_engine
import full_size
engine = "emu" if len(sys.argv) > 6 and sys.argv[6] == "emu" else "hip"
fhe = load_engine(engine)
t0 = time.time(); done = 0; fails = []
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 400
first = int(sys.argv[2]) if len(sys.argv) > 2 else 48
last = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
ks_mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # fhe_ksk_set_mode of every key the sweep makes (0 = auto)
big = (sys.argv[5] if sys.argv[5] in ("big", "f64", "f64wide") else False) if len(sys.argv) > 5 else False   # "f64": shapes on the FP64-FMA kernels
with fhe.KeySwitchingKey.forced_mode(ks_mode):     # (thread-local, restored on exit)
    for idx in range(first, last):
        try:
            (full_size.check_random_shape_host if engine == "emu" else full_size.check_random_shape)(fhe, idx, big)
            done += 1
        except Exception as e:
            fails.append((idx, full_size.random_shape(idx, big), repr(e)[:200]))
            break
        if time.time() - t0 > budget: break
print(json.dumps({"engine": engine, "shapes_checked": done, "first_idx": first, "ks_mode": ks_mode, "big": big, "failures": fails, "seconds": round(time.time() - t0)}))
