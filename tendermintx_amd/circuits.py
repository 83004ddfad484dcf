"""Host-side mirror of the reference's interface for the skip / step path (same names, argument meaning and error
behaviour), over the C ABI.  The Rust host of the reference keeps these types; this module is what the parity
tests drive in its place because no Rust toolchain exists in this image (DESIGN.md "Boundary").

  InputDataFetcher           reference circuits/input/mod.rs:36-116 (fixture mode :188-282)
  SkipCircuit / StepCircuit  reference circuits/skip.rs:104-143, circuits/step.rs:91-127
  SKIP_MAX, chain ids        reference circuits/config.rs:10-31
"""
import ctypes as C
import os

from . import _lib
from ._lib import KIND_SKIP, KIND_STEP, HashFieldRec, ProofRec, TmxError, ValidatorRec, check
from .context import Context

SKIP_MAX = 100800                        # config.rs:12
CELESTIA_CHAIN_ID_BYTES = b"celestia"    # config.rs:15
MOCHA_4_CHAIN_ID_BYTES = b"mocha-4"      # config.rs:25


class InputDataFetcher:
    """Fixture-mode fetcher: reads `<fixture_path>/<height>/commit.json` and `validators_<page>.json`
    (mod.rs:189-193, 249-254) and converts them with the library's codec into packed records.  A height that holds only the
    reference's `signed_block.json` (SignedBlockResponse, tendermint_utils.rs:52-55, 97-112: header, commit and validator set of one
    block in one document) is served from it: the codec reads the same header / commit / validators out of either shape."""

    def __init__(self, fixture_path="./circuits/fixtures/mocha-4"):
        self.fixture_path = fixture_path
        self._L = _lib.lib()

    def _read(self, height, name):
        with open(os.path.join(self.fixture_path, str(height), name), "rb") as f:
            return f.read()

    def get_signed_header_json(self, block_number):
        try:
            return self._read(block_number, "commit.json")
        except FileNotFoundError:   # a height that holds only the SignedBlockResponse
            return self._read(block_number, "signed_block.json")

    def get_validator_set_json(self, block_number):
        """All pages back to back (mod.rs:219-241 pages until count >= total)."""
        import json
        pages, page, so_far = [], 1, 0
        while True:
            try:
                raw = self._read(block_number, f"validators_{page}.json")
            except FileNotFoundError:
                if page != 1:
                    raise
                return self._read(block_number, "signed_block.json")
            r = json.loads(raw)["result"]
            pages.append(raw)
            so_far += int(r["count"])
            if so_far >= int(r["total"]):
                return b"\n".join(pages)
            page += 1

    def get_skip_inputs(self, n_max, trusted_block_number, trusted_block_hash, target_block_number):
        """get_skip_inputs::<N, F> (mod.rs:425-523) up to the packed records: (proof, target[n], trusted[n]) as bytes."""
        proof = ProofRec()
        target = (ValidatorRec * n_max)()
        trusted = (HashFieldRec * n_max)()
        st = self._L.tmx_skip_inputs_from_json(
            self.get_signed_header_json(trusted_block_number), self.get_validator_set_json(trusted_block_number),
            self.get_signed_header_json(target_block_number), self.get_validator_set_json(target_block_number),
            n_max, trusted_block_number, bytes(trusted_block_hash), target_block_number,
            C.byref(proof), target, trusted)
        if st == -2:  # the reference asserts here (mod.rs:439-444)
            raise AssertionError("The validator set size of the trusted or target block is larger than the VALIDATOR_SET_SIZE_MAX.")
        check(st)
        return bytes(proof), bytes(target), bytes(trusted)

    def get_step_inputs(self, n_max, prev_block_number, prev_header_hash):
        """get_step_inputs::<N, F> (mod.rs:316-423) up to the packed records."""
        proof = ProofRec()
        target = (ValidatorRec * n_max)()
        st = self._L.tmx_step_inputs_from_json(
            self.get_signed_header_json(prev_block_number), self.get_signed_header_json(prev_block_number + 1),
            self.get_validator_set_json(prev_block_number + 1), n_max, prev_block_number, bytes(prev_header_hash),
            C.byref(proof), target)
        if st == -2:  # mod.rs:338-342
            raise AssertionError("The validator set size of the next block is larger than the VALIDATOR_SET_SIZE_MAX.")
        check(st)
        return bytes(proof), bytes(target)


    # ---- operator side
    def get_skipcheck_inputs(self, n_max, start_block, target_block):
        """Address/power records of the start set, the target set and the target commit for is_valid_skip."""
        from ._lib import AddrRec
        start, target, sigs = (AddrRec * n_max)(), (AddrRec * n_max)(), (AddrRec * n_max)()
        ns, nt, ng = C.c_uint32(), C.c_uint32(), C.c_uint32()
        st = self._L.tmx_skipcheck_inputs_from_json(self.get_validator_set_json(start_block), self.get_validator_set_json(target_block),
                                                    self.get_signed_header_json(target_block), n_max, start, C.byref(ns), target, C.byref(nt),
                                                    sigs, C.byref(ng))
        check(st)
        return bytes(start), ns.value, bytes(target), nt.value, bytes(sigs), ng.value

    def find_block_to_request(self, ctx, start_block, max_end_block):
        """find_block_to_request (mod.rs:160-186): the reference tests max_end, then keeps halving towards start_block, one RPC round and
        one is_valid_skip per candidate.  Here every candidate on that descent is fetched first and all predicates are evaluated in one
        launch (tmx_valid_skip_batch); the first valid one wins, exactly as in the reference."""
        cands, cur = [], max_end_block
        while cur - start_block != 1:
            cands.append(cur)
            cur = (cur + start_block) // 2
        if not cands:
            return cur
        n = ctx.n_max
        start = None
        targets, sigs, nts, ngs = [], [], [], []
        for cb in cands:
            s, ns, t, nt, g, ng = self.get_skipcheck_inputs(n, start_block, cb)
            start, n_start = s, ns
            targets.append(t); sigs.append(g); nts.append(nt); ngs.append(ng)
        valid, _, _ = ctx.valid_skip_batch(start, n_start, b"".join(targets), nts, b"".join(sigs), ngs)
        for cb, ok in zip(cands, valid):
            if ok:
                return cb
        return cur  # == start_block + 1: fall back to a step


class _Circuit:
    kind = None

    def __init__(self, max_validator_set_size, chain_id_bytes=CELESTIA_CHAIN_ID_BYTES, skip_max=SKIP_MAX,
                 fetcher=None, device=0, max_batch=1):
        self.n = max_validator_set_size
        self.chain_id_bytes = bytes(chain_id_bytes)
        self.skip_max = skip_max
        self.fetcher = fetcher or InputDataFetcher()
        self.ctx = Context(self.n, self.chain_id_bytes, skip_max, device=device, max_batch=max_batch)

    def close(self):
        self.ctx.close()


class SkipCircuit(_Circuit):
    """SkipCircuit<MAX_VALIDATOR_SET_SIZE, CHAIN_ID_SIZE_BYTES, C> (skip.rs:104-143), value level."""
    kind = KIND_SKIP

    def hint(self, trusted_block, trusted_header_hash, target_block):
        """SkipOffchainInputs::hint (skip.rs:64-102): returns (elements [elem_count], report).  The first
        hint_elem_count elements are the VerifySkipVariable<N> value written to the output stream."""
        proof, target, trusted = self.fetcher.get_skip_inputs(self.n, trusted_block, trusted_header_hash, target_block)
        elems, reps = self.ctx.witness_batch(KIND_SKIP, proof, target, trusted)
        rep = reps[0]
        if rep["fail_mask"] & 1:  # mod.rs:450-455 sanity assert on the trusted header hash
            raise AssertionError("Trusted header hash doesn't pass sanity check! An incorrect header was likely pushed to the "
                                 "contract, typically the genesis header.")
        if rep["first_bad_sig"] >= 0:  # conversion.rs:48-49
            raise AssertionError("Signature should be valid for validator")
        return elems[0], rep

    def hint_value(self, trusted_block, trusted_header_hash, target_block):
        """SkipOffchainInputs::hint up to the value it builds: the reference's `SkipInputs<F>` (input/mod.rs:60-74) as the typed structs of
        include/tmx.h -- (SkipInputsFixed, ValidatorValue[N], HashFieldValue[N]) -- from which the hint body assigns `VerifySkipStruct`
        by field name (skip.rs:85-98).  No element row is produced on the device (tmx_skip_inputs_value)."""
        proof, target, trusted = self.fetcher.get_skip_inputs(self.n, trusted_block, trusted_header_hash, target_block)
        val, lay = self.ctx.inputs_value_batch(KIND_SKIP, proof, target, trusted, _lib.SEC_HINT)
        buf = bytes(val[0])
        fixed = _lib.SkipInputsFixed.from_buffer_copy(buf[:lay.fixed_bytes])
        if fixed.report.fail_mask & 1:
            raise AssertionError("Trusted header hash doesn't pass sanity check! An incorrect header was likely pushed to the "
                                 "contract, typically the genesis header.")
        if fixed.report.first_bad_sig >= 0:
            raise AssertionError("Signature should be valid for validator")
        return (fixed, (_lib.ValidatorValue * self.n).from_buffer_copy(buf[lay.off_validators:lay.off_validators + 240 * self.n]),
                (_lib.HashFieldValue * self.n).from_buffer_copy(buf[lay.off_hashfields:lay.off_hashfields + 48 * self.n]))

    def prove_public(self, input_bytes):
        """`circuit.prove(PublicInput::Bytes(..))` at the public-value level (skip.rs:197-212): 48 B in, 32 B out."""
        tb, th, gb = C.c_uint64(), C.create_string_buffer(32), C.c_uint64()
        _lib.lib().tmx_unpack_skip_input(bytes(input_bytes), C.byref(tb), th, C.byref(gb))
        _, rep = self.hint(tb.value, th.raw, gb.value)
        if not rep["all_ok"]:
            raise AssertionError(f"circuit constraints not satisfied (fail_mask={rep['fail_mask']:#x}, gt_target={rep['gt_target']}, "
                                 f"gt_trusted={rep['gt_trusted']}, dist_ok={rep['dist_ok']})")
        return rep["header"]


class StepCircuit(_Circuit):
    """StepCircuit<MAX_VALIDATOR_SET_SIZE, CHAIN_ID_SIZE_BYTES, C> (step.rs:91-127), value level."""
    kind = KIND_STEP

    def hint(self, prev_block_number, prev_header_hash):
        """StepOffchainInputs::hint (step.rs:56-89)."""
        proof, target = self.fetcher.get_step_inputs(self.n, prev_block_number, prev_header_hash)
        elems, reps = self.ctx.witness_batch(KIND_STEP, proof, target, None)
        rep = reps[0]
        if rep["fail_mask"] & (1 << 12):  # mod.rs:324-329 "Prev header hash doesn't pass sanity check"
            raise AssertionError("Prev header hash doesn't pass sanity check")
        if rep["first_bad_sig"] >= 0:
            raise AssertionError("Signature should be valid for validator")
        return elems[0], rep

    def hint_value(self, prev_block_number, prev_header_hash):
        """StepOffchainInputs::hint up to the `StepInputs<F>` value (input/mod.rs:45-58): (StepInputsFixed, ValidatorValue[N])."""
        proof, target = self.fetcher.get_step_inputs(self.n, prev_block_number, prev_header_hash)
        val, lay = self.ctx.inputs_value_batch(KIND_STEP, proof, target, None, _lib.SEC_HINT)
        buf = bytes(val[0])
        fixed = _lib.StepInputsFixed.from_buffer_copy(buf[:lay.fixed_bytes])
        if fixed.report.fail_mask & (1 << 12):
            raise AssertionError("Prev header hash doesn't pass sanity check")
        if fixed.report.first_bad_sig >= 0:
            raise AssertionError("Signature should be valid for validator")
        return fixed, (_lib.ValidatorValue * self.n).from_buffer_copy(buf[lay.off_validators:lay.off_validators + 240 * self.n])

    def prove_public(self, input_bytes):
        pb, ph = C.c_uint64(), C.create_string_buffer(32)
        _lib.lib().tmx_unpack_step_input(bytes(input_bytes), C.byref(pb), ph)
        _, rep = self.hint(pb.value, ph.raw)
        if not rep["all_ok"]:
            raise AssertionError(f"circuit constraints not satisfied (fail_mask={rep['fail_mask']:#x}, gt_target={rep['gt_target']})")
        return rep["header"]
