"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/tmx.h declares, its host codec turns the
reference's fixture JSON into exactly the records the oracle-side Python codec produces, and it refuses to compute
without a GPU (no CPU fallback).  No kernel is launched here."""
import ctypes as C
import os
import re

import pytest

import tmx_model as m
from conftest import GOLDEN, ROOT

FX = os.path.join(GOLDEN, "fixtures", "mocha-4")


def test_exports_every_declared_symbol(built_lib):
    hdr = open(os.path.join(ROOT, "include", "tmx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(tmx_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 20
    for n in sorted(names):
        assert hasattr(built_lib, n), f"libtmx.so does not export {n}"


def test_no_hip_runtime_dt_needed():
    """libtmx.so must not pin a HIP runtime: the process (PyTorch or a C/Rust host) provides it -- INTEGRATION.md"""
    import subprocess
    out = subprocess.check_output(["objdump", "-p", os.path.join(ROOT, "tendermintx_amd", "libtmx.so")]).decode()
    assert "libamdhip64" not in out


def test_elem_counts_match_reference_formulas(built_lib, oracle):
    for n in (1, 2, 4, 32, 100, 128, 512):
        assert built_lib.tmx_hint_elem_count(0, n) == 1776 * n + 5320      # VerifySkipVariable<N>, SURVEY App. B
        assert built_lib.tmx_hint_elem_count(1, n) == 1517 * n + 6919      # VerifyStepVariable<N>
        for kind in (0, 1):
            assert built_lib.tmx_elem_count(kind, n) == oracle.elem_count(kind, n) == m.elem_count(kind, n)
            assert built_lib.tmx_elem_stride(kind, n) % 2 == 0
    assert built_lib.tmx_elem_count(0, 0) == 0 and built_lib.tmx_elem_count(0, 513) == 0 and built_lib.tmx_elem_count(2, 4) == 0


def test_public_input_packing(built_lib, kat):
    """abi.encodePacked(uint64, bytes32, uint64): reference skip.rs:197-199, step.rs:178-180, TendermintX.sol:104-108"""
    for inp, _ in kat["public_io"]["skip"]:
        raw = bytes.fromhex(inp)
        a, h, b = C.c_uint64(), C.create_string_buffer(32), C.c_uint64()
        built_lib.tmx_unpack_skip_input(raw, C.byref(a), h, C.byref(b))
        assert a.value == int.from_bytes(raw[:8], "big") and h.raw == raw[8:40] and b.value == int.from_bytes(raw[40:], "big")
        out = C.create_string_buffer(48)
        built_lib.tmx_pack_skip_input(a.value, h.raw, b.value, out)
        assert out.raw == raw
    for inp, _ in kat["public_io"]["step"]:
        raw = bytes.fromhex(inp)
        a, h = C.c_uint64(), C.create_string_buffer(32)
        built_lib.tmx_unpack_step_input(raw, C.byref(a), h)
        out = C.create_string_buffer(40)
        built_lib.tmx_pack_step_input(a.value, h.raw, out)
        assert out.raw == raw


def test_codec_skip_equals_python_codec(built_lib, cases):
    """tmx_skip_inputs_from_json (C++) == oracle/py fixture codec == committed golden records, incl. the 100-validator set
    with absent (flag 1) and nil (flag 3) votes."""
    from tendermintx_amd.circuits import InputDataFetcher
    f, pf = InputDataFetcher(FX), m.FixtureFetcher(FX)
    for name, a, b, n in [("skip_3000_3100_n4", 3000, 3100, 4), ("skip_10000_10500_n4", 10000, 10500, 4),
                          ("skip_10000_10500_n32", 10000, 10500, 32), ("skip_157001_157001_n128", 157001, 157001, 128),
                          ("skip_10500_157001_n128", 10500, 157001, 128),
                          # heights served from signed_block.json (SignedBlockResponse): the same codec entry points
                          ("skip_11000_11105_n16", 11000, 11105, 16), ("skip_15000_50000_n128", 15000, 50000, 128),
                          ("skip_50000_157000_n128", 50000, 157000, 128)]:
        pr, tg, tr = m.skip_inputs_from_fixtures(pf, a, b, n)
        p2, t2, r2 = f.get_skip_inputs(n, a, m.unpack_proof(pr)["hash"], b)
        assert p2 == pr and t2 == b"".join(tg) and r2 == b"".join(tr), name
        c = cases[name]
        assert p2.hex() == c["proof"] and t2.hex() == c["target"] and r2.hex() == c["trusted"], name


def test_codec_step_equals_python_codec(built_lib, cases):
    from tendermintx_amd.circuits import InputDataFetcher
    f, pf = InputDataFetcher(FX), m.FixtureFetcher(FX)
    for name, prev, n in [("step_3000_n4", 3000, 4), ("step_10000_n2", 10000, 2), ("step_10500_n4", 10500, 4), ("step_10500_n100", 10500, 100),
                          ("step_10002_n2", 10002, 2), ("step_10003_n4", 10003, 4), ("step_11000_n8", 11000, 8)]:
        pr, tg = m.step_inputs_from_fixtures(pf, prev, n)
        p2, t2 = f.get_step_inputs(n, prev, m.unpack_proof(pr)["hash"])
        assert p2 == pr and t2 == b"".join(tg), name
        assert p2.hex() == cases[name]["proof"] and t2.hex() == cases[name]["target"], name


def test_codec_error_behaviour(built_lib):
    """reference input/mod.rs:439-444 / 338-342 assert when the validator set exceeds VALIDATOR_SET_SIZE_MAX"""
    from tendermintx_amd.circuits import InputDataFetcher
    f = InputDataFetcher(FX)
    with pytest.raises(AssertionError, match="larger than the VALIDATOR_SET_SIZE_MAX"):
        f.get_skip_inputs(2, 10000, bytes(32), 10500)       # target set has 3 validators
    with pytest.raises(AssertionError, match="larger than the VALIDATOR_SET_SIZE_MAX"):
        f.get_step_inputs(2, 10500, bytes(32))
    from tendermintx_amd._lib import HashFieldRec, ProofRec, ValidatorRec
    p, t, r = ProofRec(), (ValidatorRec * 4)(), (HashFieldRec * 4)()
    assert built_lib.tmx_skip_inputs_from_json(b"{not json", b"{}", b"{}", b"{}", 4, 1, bytes(32), 3, C.byref(p), t, r) == -5
    assert built_lib.tmx_skip_inputs_from_json(None, b"{}", b"{}", b"{}", 4, 1, bytes(32), 3, C.byref(p), t, r) == -1


def test_round_nonzero_sign_bytes(built_lib):
    """The reference has no fixture with round != 0 (TODO at verify.rs:612): the codec must place LE64(round) at [13..21] and
    the block hash at [25..57] (validator.rs:133, 168)."""
    import json
    commit = json.load(open(os.path.join(FX, "10500", "commit.json")))
    commit["result"]["signed_header"]["commit"]["round"] = 5
    vals = open(os.path.join(FX, "10500", "validators_1.json"), "rb").read()
    from tendermintx_amd._lib import ProofRec, ValidatorRec
    p, t = ProofRec(), (ValidatorRec * 4)()
    prev = open(os.path.join(FX, "10500", "commit.json"), "rb").read()
    assert built_lib.tmx_step_inputs_from_json(prev, json.dumps(commit).encode(), vals, 4, 10499, bytes(32), C.byref(p), t) == 0
    msg = bytes(t[0].message)
    assert p.round == 5 and msg[12] == 0x19 and int.from_bytes(msg[13:21], "little") == 5
    assert msg[25:57].hex().upper() == commit["result"]["signed_header"]["commit"]["block_id"]["hash"]
    assert t[0].message_byte_length == 109 + 9


def test_no_cpu_fallback(built_lib):
    """Without a usable HIP device every compute entry point fails loudly (TMX_ERR_HIP); with one, this test is vacuous."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from tendermintx_amd import Context, TmxError
    with pytest.raises(TmxError) as e:
        Context(4)
    assert e.value.status == -3


def test_product_never_touches_the_oracle():
    """The product package and its native sources must not import, link or execute anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tendermintx_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".hpp", ".h")) or fn == "Makefile":
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle_c" not in text and "tmxo_" not in text and "libtmx_oracle" not in text, fn
    import subprocess
    syms = subprocess.check_output(["nm", "-D", os.path.join(ROOT, "tendermintx_amd", "libtmx.so")]).decode()
    assert "tmxo_" not in syms


def _paged_fixture(tmp_path, height, first_page, extra):
    """A copy of fixture `height` whose validator set is served in two pages: the first `first_page` real validators, then the others plus
    `extra` synthetic ones (absent votes, CommitSig::BlockIdFlagAbsent) -- the shape a /validators?page=2 answer has at > 100 validators."""
    import hashlib
    import json
    import shutil
    src = os.path.join(FX, str(height))
    dst = os.path.join(str(tmp_path), str(height))
    os.makedirs(dst)
    shutil.copy(os.path.join(src, "commit.json"), os.path.join(dst, "commit.json"))
    with open(os.path.join(src, "validators_1.json")) as f:
        page = json.load(f)
    vals = page["result"]["validators"]
    import base64
    synth = []
    for i in range(extra):
        seed = hashlib.sha256(b"tmx-page2" + bytes([i])).digest()
        synth.append({"address": hashlib.sha256(seed).hexdigest()[:40].upper(), "pub_key": {"type": "tendermint/PubKeyEd25519", "value": base64.b64encode(seed).decode()},
                      "voting_power": str(1000 + 7 * i), "proposer_priority": "0"})
    allv = vals + synth
    total = str(len(allv))
    for k, part in enumerate((allv[:first_page], allv[first_page:]), start=1):
        doc = {"jsonrpc": "2.0", "id": -1, "result": {"block_height": str(height), "validators": part, "count": str(len(part)), "total": total}}
        with open(os.path.join(dst, f"validators_{k}.json"), "w") as f:
            json.dump(doc, f)
    if extra:
        with open(os.path.join(dst, "commit.json")) as f:
            c = json.load(f)
        c["result"]["signed_header"]["commit"]["signatures"] += [
            {"block_id_flag": 1, "validator_address": "", "timestamp": "0001-01-01T00:00:00Z", "signature": None} for _ in range(extra)]
        with open(os.path.join(dst, "commit.json"), "w") as f:
            json.dump(c, f)
    return str(tmp_path)


def test_codec_two_validator_pages(built_lib, tmp_path):
    """reference circuits/input/mod.rs:219-241: the /validators RPC pages 100 per request, so a Celestia-size set (N = 128) arrives as
    page 1 (100) + page 2 (28).  tmx_skip_inputs_from_json takes the pages back to back: the records must equal the Python codec's on the
    same two files, and -- with no synthetic validators -- the one-page fixture's."""
    from tendermintx_amd.circuits import InputDataFetcher
    # (a) the real 100-validator set of 157001 split 60 + 40 == the committed one-page records
    fx_a = _paged_fixture(tmp_path / "a", 157001, 60, 0)
    one = InputDataFetcher(FX).get_skip_inputs(128, 157001, bytes(32), 157001)
    two = InputDataFetcher(fx_a).get_skip_inputs(128, 157001, bytes(32), 157001)
    assert one == two
    # (b) 128 validators = 100 + 28 through both codecs
    fx_b = _paged_fixture(tmp_path / "b", 157001, 100, 28)
    pr, tg, tr = m.skip_inputs_from_fixtures(m.FixtureFetcher(fx_b), 157001, 157001, 128)
    p2, t2, r2 = InputDataFetcher(fx_b).get_skip_inputs(128, 157001, m.unpack_proof(pr)["hash"], 157001)
    assert p2 == pr and t2 == b"".join(tg) and r2 == b"".join(tr)
    assert m.unpack_proof(pr)["nb_a"] == 128 and m.unpack_proof(pr)["nb_b"] == 128
    lanes = [t2[256 * i:256 * (i + 1)] for i in range(128)]
    assert all(l[223] & 2 for l in lanes) and not any(l[223] & 1 for l in lanes[100:])   # present, not signed
    # 129 validators do not fit N = 128 (mod.rs:439-444)
    fx_c = _paged_fixture(tmp_path / "c", 157001, 100, 29)
    with pytest.raises(AssertionError, match="larger than the VALIDATOR_SET_SIZE_MAX"):
        InputDataFetcher(fx_c).get_skip_inputs(128, 157001, bytes(32), 157001)
