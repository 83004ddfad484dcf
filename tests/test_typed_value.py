"""The typed value of the hint (include/tmx.h "TYPED VALUE"): SkipInputs<F> / StepInputs<F> of the reference (circuits/input/mod.rs:45-74)
field by field -- what SkipOffchainInputs::hint / StepOffchainInputs::hint hold before `write_value` expands it (circuits/skip.rs:85-100,
circuits/step.rs:75-87) -- plus the derived Level-1 values in packed form.

Three independent statements of the layout meet here: the C structs of include/tmx.h (what the device writes), the oracle's own structs
(oracle/c/tmxo.h, filled beside its element stream) and the ctypes mirror below, through which THIS FILE expands a value into field
elements by the reference's rules (byte = 8 big-endian bits, U32 / Variable / Bool = 1 element, U64 = 2 LE limbs,
U256 = 8 LE limbs, struct fields in declaration order of circuits/variables.rs:35-120).  The expansion must reproduce the hint section H
of the row -- and, with the derived part, the whole row -- bit for bit.

CPU tier: the oracle's value against the oracle's row and the layout function of libtmx.  GPU tier: the HIP path's value, byte for byte
against the oracle's, at N = 4 / 32 / 128 / 512, skip and step, through every entry point (host pageable, host page-locked, device)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN

FX = os.path.join(GOLDEN, "fixtures", "mocha-4")


# ------------------------------------------------------------------------------------------------ value -> field elements (the test's own)
def _bits(b):
    return np.unpackbits(np.frombuffer(bytes(b), dtype=np.uint8)).astype(np.uint64)          # most significant bit first


def _u32(v):
    return np.array([int(v) & 0xFFFFFFFF], dtype=np.uint64)


def _u64(v):
    return np.array([int(v) & 0xFFFFFFFF, (int(v) >> 32) & 0xFFFFFFFF], dtype=np.uint64)


def _u256le(b):
    return np.frombuffer(bytes(b), dtype="<u4").astype(np.uint64)


def _proof4(p):
    return np.concatenate([_bits(p[k]) for k in range(4)])


def _structs(tmxlib, kind, n, row, lay):
    """ctypes views of one proof's value"""
    L = tmxlib
    buf = bytes(row)
    fixed = (L.SkipInputsFixed if kind == 0 else L.StepInputsFixed).from_buffer_copy(buf[:lay.fixed_bytes])
    vals = (L.ValidatorValue * n).from_buffer_copy(buf[lay.off_validators:lay.off_validators + 240 * n])
    hfs = (L.HashFieldValue * n).from_buffer_copy(buf[lay.off_hashfields:lay.off_hashfields + 48 * n]) if kind == 0 else None
    return fixed, vals, hfs


def expand_hint(tmxlib, kind, n, row, lay):
    """VerifySkipVariable<N> / VerifyStepVariable<N> elements (variables.rs:91-120) from the typed value, field by field as the hint
    bodies assign them (skip.rs:85-98, step.rs:75-87)"""
    f, vals, hfs = _structs(tmxlib, kind, n, row, lay)
    out = [_bits(f.target_header if kind == 0 else f.next_header)]
    for v in vals:                                                       # ValidatorVariable, variables.rs:69-79
        out += [_bits(v.pubkey), _bits(v.sig_r), _u256le(v.sig_s), _bits(v.message), _u32(v.message_byte_length), _u64(v.voting_power),
                _u32(v.validator_byte_length), _u32(v.signed_)]
    cid = f.target_block_chain_id_proof if kind == 0 else f.next_block_chain_id_proof
    hp = f.target_block_height_proof if kind == 0 else f.next_block_height_proof
    vp = f.target_block_validators_hash_proof if kind == 0 else f.next_block_validators_hash_proof
    out += [_u32(f.nb_target_validators if kind == 0 else f.nb_validators), _u64(f.round)]
    out += [_proof4(cid.proof), _u32(cid.enc_chain_id_byte_length), _bits(cid.chain_id)]          # ChainIdProofVariable, variables.rs:35-41
    out += [_proof4(hp.proof), _u32(hp.enc_height_byte_length), _u64(hp.height)]                  # HeightProofVariable, variables.rs:49-55
    out += [_proof4(vp.proof), _bits(vp.leaf)]                                                    # MerkleInclusionProofVariable: proof, leaf
    if kind == 0:
        tp = f.trusted_block_validators_hash_proof
        out += [_u32(f.nb_trusted_validators), _proof4(tp.proof), _bits(tp.leaf)]
        for h in hfs:                                                    # ValidatorHashFieldVariable, variables.rs:82-88
            out += [_bits(h.pubkey), _u64(h.voting_power), _u32(h.validator_byte_length)]
    else:
        lb, nv = f.next_block_last_block_id_proof, f.prev_block_next_validators_hash_proof
        out += [_proof4(lb.proof), _bits(lb.leaf), _proof4(nv.proof), _bits(nv.leaf)]
    return np.concatenate(out)


def expand_derived(tmxlib, kind, n, row, lay):
    """section D of the row (DESIGN.md "Witness layout") from the packed derived values"""
    L = tmxlib
    buf = bytes(row)
    tl = (L.TargetLaneDerived * n).from_buffer_copy(buf[lay.off_target_lanes:lay.off_target_lanes + 560 * n])
    pd = L.ProofDerived.from_buffer_copy(buf[lay.off_proof_derived:lay.off_proof_derived + 976])
    tn = lay.tree_nodes
    out = []
    for d in tl:
        out += [_bits(d.marshalled), _bits(d.leaf_hash), _bits(d.sha512_digest)]                   # D.1a
    for d in tl:                                                                                   # D.1b
        out += [_u256le(d.h)] + [_u256le(d.points[k]) for k in range(10)] + [_u32(d.eddsa_ok)] + [_u32(d.flags[k]) for k in range(6)]
        out += [_u64(d.total_prefix), _u64(d.signed_prefix)]
    if kind == 0:
        tr = (L.TrustedLaneDerived * n).from_buffer_copy(buf[lay.off_trusted_lanes:lay.off_trusted_lanes + 112 * n])
        for d in tr:
            out += [_bits(d.marshalled), _bits(d.leaf_hash)]                                       # D.2a
        for d in tr:
            out += [_u32(d.flags[0]), _u32(d.flags[1]), _u64(d.total_prefix), _u64(d.matched_prefix)]   # D.2b
    out.append(_bits(buf[lay.off_nodes_target:lay.off_nodes_target + 32 * tn]))                    # D.3
    if kind == 0:
        out.append(_bits(buf[lay.off_nodes_trusted:lay.off_nodes_trusted + 32 * tn]))              # D.4
    proof_d = lambda q: np.concatenate([_bits(pd.proofs[q][k]) for k in range(5)])
    out += [proof_d(0), _bits(pd.height_leaf), proof_d(1), proof_d(2), proof_d(3)]
    if kind == 1:
        out.append(proof_d(4))
    out += [_u64(pd.tally_target[k]) for k in range(4)] + [_u32(pd.verdicts[0])]
    if kind == 0:
        out += [_u64(pd.tally_trusted[k]) for k in range(4)] + [_u32(pd.verdicts[1]), _u32(pd.verdicts[2]), _u32(pd.verdicts[3])]
    out += [_u32(pd.checks[k]) for k in range(13 if kind == 0 else 15)] + [_u32(pd.all_ok)]
    return np.concatenate(out)


def _inputs(c):
    return bytes.fromhex(c["proof"]), bytes.fromhex(c["target"]), bytes.fromhex(c["trusted"]) if c["trusted"] else None


def _oracle_value(oracle, kind, proofs, targets, trusteds, n, chain_id, skip_max, derived):
    P = len(proofs) // 2336
    rows = []
    for p in range(P):
        v, _ = oracle.witness_value(kind, proofs[2336 * p:2336 * (p + 1)], targets[256 * n * p:256 * n * (p + 1)],
                                    trusteds[48 * n * p:48 * n * (p + 1)] if trusteds else None, chain_id, skip_max, derived)
        rows.append(v)
    return np.stack(rows)


# ------------------------------------------------------------------------------------------------ CPU tier
def test_layout_function_matches_the_structs(built_lib, oracle):
    """tmx_value_layout_of (host arithmetic, no device) against the ctypes mirror and the oracle's byte count"""
    from tendermintx_amd import _lib
    for kind in (0, 1):
        for n in (1, 4, 32, 100, 128, 512):
            tn = int(oracle.lib().tmxo_tree_nodes(n))
            for sections in (_lib.SEC_HINT, _lib.SEC_ALL):
                lay = _lib.ValueLayout()
                assert built_lib.tmx_value_layout_of(kind, n, sections, C.byref(lay)) == 0
                fixed = C.sizeof(_lib.SkipInputsFixed if kind == 0 else _lib.StepInputsFixed)
                want = fixed + 240 * n + (48 * n if kind == 0 else 0)
                assert lay.fixed_bytes == fixed and lay.off_validators == fixed and lay.tree_nodes == tn
                assert lay.off_hashfields == (fixed + 240 * n if kind == 0 else 0)
                if sections == _lib.SEC_ALL:
                    assert lay.off_target_lanes == want
                    want += 560 * n + (112 * n if kind == 0 else 0) + 32 * tn * (2 if kind == 0 else 1) + 976
                    assert lay.off_proof_derived == want - 976
                else:
                    assert lay.off_target_lanes == lay.off_proof_derived == 0
                assert lay.bytes == want == oracle.value_bytes(kind, n, sections == _lib.SEC_ALL) and lay.bytes % 16 == 0
    lay = _lib.ValueLayout()
    for bad in ((2, 4, 1), (0, 0, 1), (0, 513, 1), (0, 4, 2), (0, 4, 0)):       # derived values alone are not a value
        assert built_lib.tmx_value_layout_of(bad[0], bad[1], bad[2], C.byref(lay)) == -1


def test_oracle_value_expands_to_the_row(built_lib, oracle, cases):
    """the oracle's typed value, expanded by this file, IS the oracle's element row: H from SkipInputs / StepInputs, D from the packed
    derived values -- every golden case (fixtures of the reference: N = 2 .. 128, nil and absent votes, skip and step)"""
    from tendermintx_amd import _lib
    for name, c in sorted(cases.items()):
        kind, n = c["kind"], c["n"]
        proof, target, trusted = _inputs(c)
        row, rep = oracle.witness(kind, proof, target, trusted, c["chain_id"].encode(), c["skip_max"])
        val, vrep = oracle.witness_value(kind, proof, target, trusted, c["chain_id"].encode(), c["skip_max"], True)
        assert vrep == rep, name
        lay = _lib.ValueLayout()
        assert built_lib.tmx_value_layout_of(kind, n, _lib.SEC_ALL, C.byref(lay)) == 0 and lay.bytes == val.size
        hint = int(built_lib.tmx_hint_elem_count(kind, n))
        h = expand_hint(_lib, kind, n, val, lay)
        assert h.size == hint and np.array_equal(h, row[:hint]), name
        d = expand_derived(_lib, kind, n, val, lay)
        assert np.array_equal(d, row[hint:]), name
        # the hint-only value is a prefix of the full one
        vh, _ = oracle.witness_value(kind, proof, target, trusted, c["chain_id"].encode(), c["skip_max"], False)
        assert np.array_equal(vh, val[:vh.size])
        f = (_lib.SkipInputsFixed if kind == 0 else _lib.StepInputsFixed).from_buffer_copy(bytes(val[:lay.fixed_bytes]))
        assert bytes(f.report.header).hex() == c["header"] and bool(f.report.all_ok) == c["all_ok"]
        assert bytes(f.target_header if kind == 0 else f.next_header).hex() == c["header"]


# ------------------------------------------------------------------------------------------------ GPU tier
@pytest.fixture(scope="module")
def tmx(built_lib):
    import tendermintx_amd
    return tendermintx_amd


@pytest.mark.gpu
def test_golden_values_bit_exact(tmx, oracle, cases):
    """every golden case through tmx_inputs_value_batch: HIP value == oracle value byte for byte (hint-only and with the derived part),
    and its expansion == the row the element path produces on the same context"""
    from tendermintx_amd import _lib
    for name, c in sorted(cases.items()):
        kind, n = c["kind"], c["n"]
        proof, target, trusted = _inputs(c)
        cid = c["chain_id"].encode()
        with tmx.Context(n, cid, c["skip_max"], max_batch=1) as ctx:
            for sections, derived in ((_lib.SEC_HINT, False), (_lib.SEC_ALL, True)):
                for rerun in range(2):                                        # cold, then with the key cache warm
                    got, lay = ctx.inputs_value_batch(kind, proof, target, trusted, sections)
                    want = _oracle_value(oracle, kind, proof, target, trusted, n, cid, c["skip_max"], derived)
                    if not np.array_equal(got, want):
                        bad = np.argwhere(got != want)
                        raise AssertionError(f"{name} sections={sections} run {rerun}: value differs at (proof, byte) {bad[:12].tolist()} ({len(bad)} bytes)")
            elems, _ = ctx.witness_batch(kind, proof, target, trusted)
            hint = ctx.hint_elem_count(kind)
            assert np.array_equal(expand_hint(_lib, kind, n, got[0], lay), elems[0][:hint]), name
            assert np.array_equal(expand_derived(_lib, kind, n, got[0], lay), elems[0][hint:]), name


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,nb,proofs", [(0, 4, 4, 3), (0, 32, 21, 5), (1, 32, 32, 4), (0, 128, 100, 7), (1, 128, 128, 3), (0, 512, 400, 2), (1, 512, 512, 1),
                                               (0, 128, 100, 40)])
def test_synthetic_values_bit_exact(tmx, oracle, kind, n, nb, proofs):
    """N = 4 / 32 / 128 / 512, skip and step, rounds != 0, nb < N: pageable host buffers, page-locked host buffers (the device writes the
    value itself) and the device entry point all give the oracle's bytes; the last case (40 x 128 = 5120 lanes) is the classic launch graph"""
    import torch
    from tendermintx_amd import _lib
    from tendermintx_amd.synth import Workload
    wl = Workload(kind, n, proofs, nb, chain_id=b"celestia", seed=77 + n + nb + proofs, signed_permille=900, rounds=(0, 3, 0, 2**40 + 7, 1))
    want = _oracle_value(oracle, kind, wl.proofs, wl.targets, wl.trusteds, n, b"celestia", 100800, True)
    with tmx.Context(n, b"celestia", 100800, max_batch=proofs) as ctx:
        lay = ctx.value_layout(kind, _lib.SEC_ALL)
        got, _ = ctx.inputs_value_batch(kind, wl.proofs, wl.targets, wl.trusteds, _lib.SEC_ALL)
        assert np.array_equal(got, want)
        # page-locked in and out
        pin = ctx.host_alloc(proofs * lay.bytes)
        pp, pt = ctx.host_alloc(len(wl.proofs)), ctx.host_alloc(len(wl.targets))
        pp[:] = np.frombuffer(wl.proofs, dtype=np.uint8); pt[:] = np.frombuffer(wl.targets, dtype=np.uint8)
        pr = None
        if kind == 0:
            pr = ctx.host_alloc(len(wl.trusteds))
            pr[:] = np.frombuffer(wl.trusteds, dtype=np.uint8)
        pin[:] = 0xA5
        got2, _ = ctx.inputs_value_batch(kind, pp, pt, pr, _lib.SEC_ALL, out=pin)
        assert np.array_equal(got2, want)
        hl = ctx.value_layout(kind, _lib.SEC_HINT)
        got3, _ = ctx.inputs_value_batch(kind, pp, pt, pr, _lib.SEC_HINT, out=pin)
        assert np.array_equal(got3, want[:, :hl.bytes])
        for a in (pin, pp, pt) + ((pr,) if pr is not None else ()):
            ctx.host_free(a)
        # device entry point on torch's stream
        dev = torch.device("cuda:0")
        d = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
        dp, dt_, dr = d(wl.proofs), d(wl.targets), d(wl.trusteds) if kind == 0 else None
        dout = torch.full((proofs * lay.bytes,), 0x5A, dtype=torch.uint8, device=dev)
        ctx.inputs_value_batch_device(kind, proofs, dp.data_ptr(), dt_.data_ptr(), dr.data_ptr() if dr is not None else None, dout.data_ptr(), _lib.SEC_ALL,
                                      stream=int(torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert np.array_equal(dout.cpu().numpy().reshape(proofs, lay.bytes), want)
        # the hint expands to the reference's element stream at this size too
        hint = ctx.hint_elem_count(kind)
        row, _ = oracle.witness(kind, wl.proofs[:2336], wl.targets[:256 * n], wl.trusteds[:48 * n] if kind == 0 else None, b"celestia", 100800)
        assert np.array_equal(expand_hint(_lib, kind, n, got[0], lay), row[:hint])


@pytest.mark.gpu
def test_value_reports_failures_like_the_row_path(tmx, oracle, cases):
    """a corrupted signature and a wrong trusted hash: the report inside the value is the oracle's, and the derived check words say why"""
    from tendermintx_amd import _lib
    c = cases["skip_10000_10500_n32"]
    proof, target, trusted = _inputs(c)
    t = bytearray(target)
    t[32 + 5] ^= 0x40                                                       # R of lane 0
    p = bytearray(proof)
    p[16] ^= 1                                                              # the public trusted header hash
    cid = c["chain_id"].encode()
    with tmx.Context(32, cid, c["skip_max"]) as ctx:
        got, lay = ctx.inputs_value_batch(0, bytes(p), bytes(t), trusted, _lib.SEC_ALL)
    want = _oracle_value(oracle, 0, bytes(p), bytes(t), trusted, 32, cid, c["skip_max"], True)
    assert np.array_equal(got, want)
    f = _lib.SkipInputsFixed.from_buffer_copy(bytes(got[0][:lay.fixed_bytes]))
    _, orep = oracle.witness(0, bytes(p), bytes(t), trusted, cid, c["skip_max"])
    assert not orep["all_ok"] and (orep["fail_mask"] & 1) and (orep["fail_mask"] >> 9) & 1 and orep["first_bad_sig"] >= 0   # the scenario bites
    assert (bool(f.report.all_ok), f.report.fail_mask, f.report.first_bad_sig) == (orep["all_ok"], orep["fail_mask"], orep["first_bad_sig"])
    pd = _lib.ProofDerived.from_buffer_copy(bytes(got[0][lay.off_proof_derived:lay.off_proof_derived + 976]))
    assert [int(pd.checks[k]) for k in range(13)] == [0 if (orep["fail_mask"] >> k) & 1 else 1 for k in range(13)] and pd.all_ok == 0


@pytest.mark.gpu
def test_value_call_rejects_what_the_reference_asserts(tmx, cases):
    """input/mod.rs:439-444: a validator set larger than N is refused before anything is enqueued; capacity and argument errors"""
    from tendermintx_amd import _lib
    c = cases["skip_10000_10500_n4"]
    proof, target, trusted = _inputs(c)
    with tmx.Context(4, c["chain_id"].encode(), c["skip_max"]) as ctx:
        p = bytearray(proof)
        p[56:60] = (5).to_bytes(4, "little")
        with pytest.raises(_lib.TmxError) as e:
            ctx.inputs_value_batch(0, bytes(p), target, trusted)
        assert e.value.status == -2
        lay = ctx.value_layout(0)
        small = np.zeros(lay.bytes - 16, dtype=np.uint8)
        L = _lib.lib()
        assert L.tmx_inputs_value_batch(ctx._h, 0, 1, proof, target, trusted, _lib.SEC_HINT, small.ctypes.data, small.nbytes) == -4      # TMX_ERR_CAPACITY
        ok = np.zeros(lay.bytes, dtype=np.uint8)
        assert L.tmx_inputs_value_batch(ctx._h, 0, 1, proof, target, None, _lib.SEC_HINT, ok.ctypes.data, ok.nbytes) == -1               # skip without trusted
        assert L.tmx_inputs_value_batch(ctx._h, 0, 1, proof, target, trusted, _lib.SEC_DERIVED, ok.ctypes.data, ok.nbytes) == -1         # derived alone
        assert L.tmx_inputs_value_batch(ctx._h, 0, 2, proof, target, trusted, _lib.SEC_HINT, ok.ctypes.data, 2 * ok.nbytes) == -4        # max_batch = 1
        assert L.tmx_inputs_value_batch(ctx._h, 0, 1, proof, target, trusted, _lib.SEC_HINT, ok.ctypes.data, ok.nbytes) == 0


@pytest.mark.gpu
def test_hint_value_reads_like_the_reference(tmx, kat):
    """test_skip_small / test_step_small (skip.rs:252-265, step.rs:230-241) through SkipCircuit.hint_value / StepCircuit.hint_value: the
    typed SkipInputs / StepInputs, the fields the hint body assigns by name"""
    circ = tmx.SkipCircuit(4, tmx.MOCHA_4_CHAIN_ID_BYTES, tmx.SKIP_MAX, fetcher=tmx.InputDataFetcher(FX))
    try:
        f, vals, hfs = circ.hint_value(10000, bytes.fromhex("A0123D5E4B8B8888A61F931EE2252D83568B97C223E0ECA9795B29B8BD8CBA2D"), 10500)
        assert bytes(f.target_header).hex().upper() == "E2BA1B86926925A69C2FCC32E5178E7E6653D386C956BB975142FA73211A9444" and f.report.all_ok
        assert bytes(f.trusted_header).hex().upper() == "A0123D5E4B8B8888A61F931EE2252D83568B97C223E0ECA9795B29B8BD8CBA2D"
        assert f.target_block_height_proof.height == 10500 and f.target_block_chain_id_proof.enc_chain_id_byte_length == 9
        assert bytes(f.target_block_chain_id_proof.chain_id)[:9] == b"\x0a\x07mocha-4" and f.round == 0
        assert f.nb_target_validators <= 4 and all(v.message_byte_length >= 32 and v.validator_byte_length in range(38, 47) for v in vals)
        assert len(hfs) == 4
        with pytest.raises(AssertionError, match="Trusted header hash doesn't pass sanity check"):
            circ.hint_value(10000, bytes(32), 10500)
    finally:
        circ.close()
    circ = tmx.StepCircuit(2, tmx.MOCHA_4_CHAIN_ID_BYTES, fetcher=tmx.InputDataFetcher(FX))
    try:
        f, vals = circ.hint_value(10000, bytes.fromhex("A0123D5E4B8B8888A61F931EE2252D83568B97C223E0ECA9795B29B8BD8CBA2D"))
        assert bytes(f.next_header).hex().upper() == "F2A340CC2AEF6FE163254B326A52334B45793EB11417029F9548418F88B38E26" and f.report.all_ok
        assert f.next_block_height_proof.height == 10001 and bytes(f.next_block_last_block_id_proof.leaf)[:4] == b"\x0a\x20" + bytes.fromhex("A012")
    finally:
        circ.close()
