"""The two oracles against each other under the mutations of test_fuzz_extended.py (CPU only): oracle/c (5x51-limb C) and oracle/py
(big-int model) were written from the same reading of the reference but share no code, so a slip in one of them shows here without a GPU.
TMX_CROSS_FUZZ=N seeds (default 150, about five seconds)."""
import os

import numpy as np
import pytest

N_SEEDS = int(os.environ.get("TMX_CROSS_FUZZ", "150"))


def test_c_oracle_equals_python_model_on_mutated_batches(oracle, monkeypatch):
    import tmx_model as pm
    monkeypatch.setenv("TMX_FUZZ_NSET", "1,2,4")
    import importlib
    import sys
    import types
    if "test_gpu_parity" not in sys.modules:   # (test_fuzz_extended imports its GPU helper from there; not needed here)
        stub = types.ModuleType("test_gpu_parity")
        stub._check_vs_oracle = None
        monkeypatch.setitem(sys.modules, "test_gpu_parity", stub)
    fz = importlib.import_module("test_fuzz_extended")
    monkeypatch.setattr(fz, "NSET", tuple(int(v) for v in os.environ.get("TMX_CROSS_NSET", "1,2,4").split(",")))
    checked = 0
    for seed in range(N_SEEDS):
        kind, n, proofs, targets, trusteds, chain, skip_max = fz._mutated_batch(seed)
        P = len(proofs) // 2336
        want, reps = oracle.witness_batch(kind, P, proofs, targets, trusteds, n, chain, skip_max, n_threads=1)
        for p in range(min(P, 4)):
            tg = [targets[(p * n + i) * 256:(p * n + i + 1) * 256] for i in range(n)]
            tr = [trusteds[(p * n + i) * 48:(p * n + i + 1) * 48] for i in range(n)] if trusteds else None
            el, rep = pm.witness(kind, proofs[p * 2336:(p + 1) * 2336], tg, tr, chain, skip_max)
            el = np.array(el, dtype=np.uint64)
            assert len(el) == want.shape[1] and np.array_equal(el, want[p]), (seed, p, np.argwhere(el != want[p][:len(el)])[:5].ravel().tolist())
            for k in ("all_ok", "fail_mask", "first_bad_sig", "gt_target", "dist_ok"):
                if k in rep:
                    assert rep[k] == reps[p][k], (seed, p, k)
            if kind == 0:
                assert rep["gt_trusted"] == reps[p]["gt_trusted"], (seed, p)
            checked += 1
    assert checked >= N_SEEDS
