#!/usr/bin/env python3
"""Randomised parity sweep of the HIP path against the CPU oracle (development aid, GPU box).
usage: fuzz_parity.py [n_cases] [first_seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import spada_sim_amd as S
from oracle import oracle
from conftest import assert_parity, to_oracle


from fuzz_cases import random_case


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    engines = [S.Engine(), S.Engine(accumulator=S.ACC_SORT_MERGE)]
    bad = 0
    for seed in range(first, first + ncases):
        a, b, desc = random_case(seed)
        ao, bo = to_oracle(a), to_oracle(b)
        ref = oracle.spgemm_sortmerge(ao, bo)
        for e in engines:
            try:
                assert_parity(e.spgemm(a, b), ref, ao, bo, 1e-9)            # two-phase contract
                assert_parity(e.spgemm_fused(a, b), ref, ao, bo, 1e-9)      # one-pass entry point
            except AssertionError as ex:
                bad += 1
                print("FAIL", desc, "nnzC", ref.nnz, str(ex)[:200])
    print(f"{ncases} cases x 2 accumulators x 2 entry points, {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
