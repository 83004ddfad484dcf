"""Level-2 trace rows (DESIGN.md "Level-2 trace rows"): the oracle's generator (affine arithmetic, one inversion per operation) against
its own constraint checker here, and the HIP rows (projective ladder, batched inversions, wave-cooperative stores) against both on the
GPU box: bit-exact vs the generator, and accepted row by row by the checker."""
import struct

import numpy as np
import pytest


def _case(cases, name):
    c = cases[name]
    return c["kind"], c["n"], bytes.fromhex(c["proof"]), bytes.fromhex(c["target"]), (bytes.fromhex(c["trusted"]) if c["trusted"] else None)


@pytest.mark.parametrize("name", ["skip_10000_10500_n4", "step_10500_n4", "skip_3000_3100_n4"])
def test_generated_rows_satisfy_every_constraint(oracle, cases, built_lib, name):
    kind, n, _, targets, trusteds = _case(cases, name)
    tr = oracle.trace(kind, targets, trusteds, n)
    assert tr.size == oracle.trace_elem_count(kind, n) == built_lib.tmx_trace_elem_count(kind, n)
    assert int(tr.max()) < 2**32
    assert oracle.trace_check(kind, targets, trusteds, n, tr) == 0
    # the last row of every ladder is the Level-1 point: s*B / h*A of the lane (checked inside trace_check against tmxo_eddsa_trace_lane)
    rng = np.random.default_rng(3)
    for _ in range(300):   # any single-bit change of any element breaks a constraint
        i = int(rng.integers(0, tr.size))
        m = tr.copy()
        m[i] ^= np.uint64(1 << int(rng.integers(0, 33)))
        assert oracle.trace_check(kind, targets, trusteds, n, m) != 0, i


def test_ladder_checker_rejects_a_consistent_trace_of_another_scalar(oracle):
    """Rows that satisfy every curve relation but belong to scalar k' != k fail the bit-composition constraint."""
    import ctypes as C
    L = oracle.lib()
    bx, by = (C.c_uint8 * 32)(), (C.c_uint8 * 32)()
    L.tmxo_base_point(bx, by)
    rows = np.zeros(256 * 65, dtype=np.uint64)
    k1, k2 = bytes(range(1, 33)), bytes(range(2, 34))
    L.tmxo_trace_ladder(k1, bytes(bx), bytes(by), rows.ctypes.data_as(C.POINTER(C.c_uint64)))
    last = rows[255 * 65 + 49:255 * 65 + 65]
    rx = b"".join(struct.pack("<I", int(w)) for w in last[:8]); ry = b"".join(struct.pack("<I", int(w)) for w in last[8:])
    L.tmxo_trace_ladder_check.restype = C.c_int
    assert L.tmxo_trace_ladder_check(rows.ctypes.data_as(C.POINTER(C.c_uint64)), k1, bytes(bx), bytes(by), rx, ry) == 0
    assert L.tmxo_trace_ladder_check(rows.ctypes.data_as(C.POINTER(C.c_uint64)), k2, bytes(bx), bytes(by), rx, ry) % 1000 == 2
    assert L.tmxo_trace_ladder_check(rows.ctypes.data_as(C.POINTER(C.c_uint64)), k1, bytes(bx), bytes(by), ry, rx) == 257008


def _gpu_trace(tmx, kind, n, proofs, targets, trusteds, sections=15):
    import torch
    P = len(proofs) // 2336
    dev = torch.device("cuda", 0)
    d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) if b else None for b in (proofs, targets, trusteds)]
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        te = ctx.trace_elem_count(kind)
        out = torch.zeros((P, ctx.elem_stride(kind)), dtype=torch.int64, device=dev)
        rep = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
        tr = torch.full((P, te), -1, dtype=torch.int64, device=dev)
        ctx.witness_batch_device(kind, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr() if d[2] is not None else None, out.data_ptr(), rep.data_ptr(), 0)
        ctx.trace_rows_device(kind, P, d[1].data_ptr(), d[2].data_ptr() if d[2] is not None else None, tr.data_ptr(), sections, 0)
        torch.cuda.synchronize(dev)
    return tr.cpu().numpy().view(np.uint64)


@pytest.mark.gpu
def test_hip_rows_equal_the_generator_and_pass_the_checker(built_lib, oracle, cases):
    import tendermintx_amd as tmx
    from tendermintx_amd.synth import Workload
    for name in ("skip_10000_10500_n4", "step_10500_n4"):
        kind, n, proof, targets, trusteds = _case(cases, name)
        got = _gpu_trace(tmx, kind, n, proof, targets, trusteds)
        assert np.array_equal(got[0], oracle.trace(kind, targets, trusteds, n)), name
        assert oracle.trace_check(kind, targets, trusteds, n, got[0]) == 0
    for kind, n, P, nb in ((0, 7, 5, 6), (1, 16, 3, 16), (0, 33, 2, 30)):
        wl = Workload(kind, n, P, nb, chain_id=b"celestia", seed=77 + n, signed_permille=800, rounds=(0, 2))
        targets = bytearray(wl.targets)
        targets[1 * 256:1 * 256 + 32] = (2).to_bytes(32, "little")            # proof 0, lane 1: undecodable public key -> zero ladders
        targets[2 * 256 + 40] ^= 0x10                                          # lane 2: corrupted R (decodes or not: both paths are legal)
        targets[(n + 3) * 256 + 64 + 31] |= 0xF0                               # proof 1, lane 3: s >= 2^252 (non-canonical, still a ladder)
        got = _gpu_trace(tmx, kind, n, wl.proofs, bytes(targets), wl.trusteds)
        for p in range(P):
            t = bytes(targets[p * n * 256:(p + 1) * n * 256])
            r = wl.trusteds[p * n * 48:(p + 1) * n * 48] if kind == 0 else None
            assert np.array_equal(got[p], oracle.trace(kind, t, r, n)), (kind, n, p)
            assert oracle.trace_check(kind, t, r, n, got[p]) == 0


@pytest.mark.gpu
def test_hip_rows_at_n128(built_lib, oracle):
    """BASELINE configs[2] shape: two proofs at N = 128 (38 MB of rows each): bit-exact vs the generator, accepted by the checker; a section
    mask leaves the other sections untouched."""
    import tendermintx_amd as tmx
    from tendermintx_amd.synth import bench_workload
    n, P = 128, 2
    wl = bench_workload("survey8d", n, P, seed=5)
    got = _gpu_trace(tmx, 0, n, wl.proofs, wl.targets, wl.trusteds)
    for p in range(P):
        t, r = wl.targets[p * n * 256:(p + 1) * n * 256], wl.trusteds[p * n * 48:(p + 1) * n * 48]
        assert oracle.trace_check(0, t, r, n, got[p]) == 0
        assert np.array_equal(got[p], oracle.trace(0, t, r, n))
    only = _gpu_trace(tmx, 0, n, wl.proofs, wl.targets, wl.trusteds, sections=2 | 8)
    lad = n * 2 * 256 * 65
    assert (only[:, :lad] == np.uint64(2**64 - 1)).all() and np.array_equal(only[:, lad:lad + n * 2880], got[:, lad:lad + n * 2880])
