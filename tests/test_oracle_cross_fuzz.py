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


def test_c_oracle_under_sanitizers_on_mutated_batches():
    """oracle/c built with -fsanitize=address,undefined (make asan) walks mutated batches -- field lengths up to 255, lengths beyond the
    buffers they index, undecodable points -- through the witness, the Level-2 generator and its checker, in a subprocess."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cdir = os.path.join(root, "oracle", "c")
    subprocess.check_call(["make", "-s", "-C", cdir, "asan"])
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    code = r'''
import ctypes as C, os, sys, types
import numpy as np
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
stub = types.ModuleType("test_gpu_parity"); stub._check_vs_oracle = None; sys.modules["test_gpu_parity"] = stub
os.environ["TMX_FUZZ_NSET"] = "1,2,4,7,16,33"
import test_fuzz_extended as f
L = C.CDLL(os.path.join(root, "oracle", "c", "libtmx_oracle_asan.so"))
L.tmxo_elem_count.restype = C.c_size_t; L.tmxo_elem_count.argtypes = [C.c_int, C.c_size_t]
L.tmxo_trace_elem_count.restype = C.c_size_t; L.tmxo_trace_elem_count.argtypes = [C.c_int, C.c_size_t]
L.tmxo_trace_check.restype = C.c_longlong
for seed in range(int(sys.argv[2])):
    kind, n, proofs, targets, trusteds, chain, skip_max = f._mutated_batch(seed)
    out = np.zeros(L.tmxo_elem_count(kind, n), dtype=np.uint64)
    rep = (C.c_uint8 * 64)()
    for p in range(len(proofs) // 2336):
        pr, tg = proofs[p * 2336:(p + 1) * 2336], targets[p * n * 256:(p + 1) * n * 256]
        tr = trusteds[p * n * 48:(p + 1) * n * 48] if trusteds else None
        assert L.tmxo_witness(kind, pr, tg, tr, C.c_uint32(n), chain, C.c_uint32(len(chain)), C.c_uint64(skip_max), out.ctypes.data_as(C.c_void_p), rep) == 0
        if n <= 4 and p < 2:
            t = np.zeros(L.tmxo_trace_elem_count(kind, n), dtype=np.uint64)
            assert L.tmxo_trace(kind, pr, tg, tr, C.c_uint32(n), t.ctypes.data_as(C.c_void_p)) == 0
            assert L.tmxo_trace_check(kind, pr, tg, tr, C.c_uint32(n), t.ctypes.data_as(C.c_void_p)) == 0
print("asan fuzz ok")
'''
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([sys.executable, "-c", code, root, os.environ.get("TMX_ASAN_FUZZ", "60")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "asan fuzz ok" in out.stdout, out.stderr[-3000:]
