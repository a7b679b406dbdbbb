"""The per-step engine parity test (tests/test_gpu_parity.py::test_engine_vs_oracle_stepwise_resync: auto-reset, TimeLimit,
selectors, episode statistics, 150 steps x 2 048 lanes, oracle re-synchronised every step) with its context table drawn over the
reference's declared bounds instead of the usual moderate ranges (GPU box; a checker's tool).

    python tools/fuzz_engine_wide.py"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import tests.test_gpu_parity as T  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    T.random_table = lambda fam, rng, n: T.wide_context_rows(fam, rng, n, 3.0 if fam == O.ACROBOT else 10.0)[0]
    for fam in range(5):
        for sel, name in ((O.SEL_STATIC, "static"), (O.SEL_ROUND_ROBIN, "rr"), (O.SEL_RANDOM, "random")):
            try:
                T.test_engine_vs_oracle_stepwise_resync.__wrapped__(fam, sel, dev) if hasattr(T.test_engine_vs_oracle_stepwise_resync, "__wrapped__") \
                    else T.test_engine_vs_oracle_stepwise_resync(fam, sel, dev)
                print(f"{O.FAMILY_NAMES[fam]:18s} {name:7s} ok")
            except AssertionError:
                tb = traceback.format_exc().strip().splitlines()
                print(f"{O.FAMILY_NAMES[fam]:18s} {name:7s} ASSERT  {tb[-1][:200]}  @ {[l.strip() for l in tb if 'test_gpu_parity.py' in l][-1][:120]}")


if __name__ == "__main__":
    main()
