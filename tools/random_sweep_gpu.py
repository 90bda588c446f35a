#!/usr/bin/env python3
"""Random-shape parity sweep on the GPU: tests/full_size.check_random_shape over seeded shapes (degree, moduli, batch,
parts) against the oracle for a time budget in seconds.  usage: random_sweep_gpu.py [seconds] [first index]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
from helpers import load_engine
import full_size
fhe = load_engine("hip")
t0 = time.time(); done = 0; fails = []
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 400
first = int(sys.argv[2]) if len(sys.argv) > 2 else 48
for idx in range(first, first + 4000):
    try:
        full_size.check_random_shape(fhe, idx)
        done += 1
    except Exception as e:
        fails.append((idx, full_size.random_shape(idx), repr(e)[:200]))
        break
    if time.time() - t0 > budget: break
print(json.dumps({"shapes_checked": done, "first_idx": first, "failures": fails, "seconds": round(time.time() - t0)}))
