#!/usr/bin/env python3
"""300 more random call sequences of tests/test_random_sequences_gpu.py (seeds 100-399), outside the test suite."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib
import mrs_multirotor_simulator_amd as mrs
from oracle import oracle_swarm as oracle
T = importlib.import_module("test_random_sequences_gpu")
fn = T.test_random_call_sequences_match_oracle
fn = getattr(fn, "__wrapped__", fn)
bad = 0
for seed in range(100, 400):
    try:
        fn.__wrapped__(mrs, oracle, seed, seed % 4 == 3, "mixed" if seed % 2 else "x500") if hasattr(fn, "__wrapped__") else fn(mrs, oracle, seed, seed % 4 == 3, "mixed" if seed % 2 else "x500")
    except Exception as e:
        bad += 1
        print("SEED", seed, "FAILED:", str(e)[:200], flush=True)
print("done, failures:", bad)
