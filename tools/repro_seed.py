"""Runs tests/test_gpu_parity.py::test_randomized_differential for one seed outside pytest and prints the full failure tuple."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
np.set_printoptions(linewidth=200, threshold=10000)
import oracle
import test_gpu_parity as T
seed = int(sys.argv[1])
try:
    T.test_randomized_differential(oracle, seed)
    print("seed", seed, "passes")
except AssertionError as e:
    print("seed", seed, "FAILS")
    for x in (e.args[0] if e.args and isinstance(e.args[0], tuple) else e.args):
        print("  ", repr(x)[:1500])
