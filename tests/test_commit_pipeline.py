"""The commit pipeline chained on the device (tmx_trace_commit_device: Level-2 section rows -> columns -> coset LDE -> Poseidon Merkle cap;
SURVEY 8(f) rank 2, what the reference's `prove` does with its traces in one process: reference circuits/skip.rs:119-133) against the CPU
chain under oracle/c: tmxo_trace -> (the same section, columns zero-padded to the power of two) -> tmxo_lde per column ->
tmxo_poseidon_merkle.  Parity pinned against that chain, not against plonky2 (its sources are absent: DESIGN.md)."""
import numpy as np
import pytest

from conftest import ROOT  # noqa: F401


def _section_geom(kind, n, section):
    sets = 2 if kind == 0 else 1
    tn, sz = 0, n
    while sz > 1:
        sz = (sz + 1) // 2
        tn += sz
    o512 = n * 2 * 256 * 65
    o256 = o512 + n * 2 * 80 * 18
    otree = o256 + sets * n * 64 * 9 + (n * n if kind == 0 else 0)
    return {1: (0, 2 * n * 256, 65), 2: (o512, 2 * n * 80, 18), 4: (o256, sets * n * 64, 9), 16: (otree, sets * tn * 128, 9),
            32: (otree + sets * tn * 1152, (4 if kind == 0 else 5) * 5 * 128, 9)}[section]


def test_shape_function(built_lib):
    import ctypes as C
    for kind, n, section in ((0, 128, 2), (0, 128, 1), (1, 4, 32), (0, 32, 16)):
        lg, w = C.c_uint32(), C.c_uint32()
        assert built_lib.tmx_trace_commit_shape(kind, n, section, C.byref(lg), C.byref(w)) == 0
        off, rows, width = _section_geom(kind, n, section)
        assert w.value == width and (1 << lg.value) >= rows and ((1 << lg.value) < 2 * rows or lg.value == 6)
    assert built_lib.tmx_trace_commit_shape(0, 128, 8, None, None) == -1       # the N x N match bits are not a row table
    assert built_lib.tmx_trace_commit_shape(0, 128, 3, None, None) == -1       # one section at a time


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,P,sections", [(0, 4, 3, (1, 2, 4, 16, 32)), (1, 4, 2, (2, 32)), (0, 32, 2, (2, 4, 16))])
def test_cap_equals_the_oracle_chain(built_lib, oracle, kind, n, P, sections):
    import torch
    import tendermintx_amd as tmx
    from tendermintx_amd.synth import Workload
    wl = Workload(kind, n, P, n - 1 if n > 4 else n, chain_id=b"celestia", seed=500 + n + kind, signed_permille=900)
    dev = torch.device("cuda", 0)
    d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) if b else None for b in (wl.proofs, wl.targets, wl.trusteds if kind == 0 else b"")]
    log_blowup, cap_h = 3, 2
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        te = ctx.trace_elem_count(kind)
        out = torch.zeros((P, ctx.elem_stride(kind)), dtype=torch.int64, device=dev)
        rep = torch.zeros(P * 64, dtype=torch.uint8, device=dev)
        tr = torch.zeros((P, te), dtype=torch.int64, device=dev)
        ctx.witness_batch_device(kind, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr() if d[2] is not None else None, out.data_ptr(), rep.data_ptr(), 0)
        ctx.trace_rows_device(kind, P, d[1].data_ptr(), d[2].data_ptr() if d[2] is not None else None, tr.data_ptr(), 63, 0)
        caps = {}
        for sec in sections:
            cap = torch.zeros(4 << cap_h, dtype=torch.int64, device=dev)
            ctx.trace_commit_device(kind, P, sec, log_blowup, cap_h, tr.data_ptr(), cap.data_ptr(), 0)
            torch.cuda.synchronize(dev)
            caps[sec] = cap.cpu().numpy().view(np.uint64).reshape(-1, 4)
            ms = ctx.trace_commit_last_ms()
            assert all(v >= 0 for v in ms.values())
            assert ctx.trace_commit_shape(kind, sec)[1] == _section_geom(kind, n, sec)[2]
        rows_gpu = tr.cpu().numpy().view(np.uint64)
    for sec in sections:
        off, rows, width = _section_geom(kind, n, sec)
        log_n = max(6, (rows - 1).bit_length())
        cols = np.zeros((P * width, 1 << log_n), dtype=np.uint64)
        for p in range(P):
            t = wl.targets[p * n * 256:(p + 1) * n * 256]
            r = wl.trusteds[p * n * 48:(p + 1) * n * 48] if kind == 0 else None
            full = oracle.trace(kind, wl.proofs[p * 2336:(p + 1) * 2336], t, r, n)
            assert np.array_equal(full, rows_gpu[p])
            m = full[off:off + rows * width].reshape(rows, width)
            cols[p * width:(p + 1) * width, :rows] = m.T
        ext = oracle.lde(cols, log_blowup)
        levels = oracle.poseidon_merkle(ext.reshape(-1), log_n + log_blowup, P * width, cap_h)
        assert np.array_equal(caps[sec], levels[-(1 << cap_h):]), (kind, n, sec)
