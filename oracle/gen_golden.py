#!/usr/bin/env python3
"""Generates tests/golden/ from the reference's fixture DATA (run in the build container, where /root/reference exists).

TEST INFRASTRUCTURE.  Reads only data files of the reference (circuits/fixtures/mocha-4/**.json) and the
known-answer tables of its in-file unit tests (transcribed below with file:line); computes expected values with the
pure-Python model in oracle/py.  Outputs:
  tests/golden/fixtures/mocha-4/<h>/{commit.json,validators_1.json}   verbatim data files the reference's tests read
  tests/golden/fixtures/mocha-4/<h>/signed_block.json                  verbatim, the nine heights that hold nothing else (SignedBlockResponse,
                                                                       tendermint_utils.rs:52-55, 97-112): all 17 fixture heights are pinned
  tests/golden/signed_blocks.json  per signed_block height: header hash, validators_hash, every signature under both verification equations
  tests/golden/cases.json        per case: packed input records (hex), Level-0 header, report, element count, sha256
  tests/golden/elems_<case>.npz  full element streams of the small cases
  tests/golden/kat.json          the five CI known-answer tables + RFC 8032 vectors + dummy-lane constants
Usage: python oracle/gen_golden.py [--reference /root/reference]
"""
import argparse
import hashlib
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "py"))
import ed25519_model as ed  # noqa: E402
import tm_encoding as tm  # noqa: E402
import tmx_model as m  # noqa: E402

SKIP_CASES = [  # (name, trusted, target, N, chain_id)   reference tests: skip.rs:188-217, 252-282
    ("skip_3000_3100_n4", 3000, 3100, 4, "mocha-4"),
    ("skip_10000_10500_n4", 10000, 10500, 4, "mocha-4"),
    ("skip_10000_10500_n32", 10000, 10500, 32, "mocha-4"),
    ("skip_157001_157001_n128", 157001, 157001, 128, "mocha-4"),   # largest real validator set; fails only the distance check
    ("skip_10500_157001_n128", 10500, 157001, 128, "mocha-4"),     # real non-overlapping sets: 1/3 check fails
    ("skip_10000_10500_n4_wrongchain", 10000, 10500, 4, "celestia"),
    # the signed_block.json heights (no reference test reads them; real mocha-4 data with MANY absent votes -- whatever verdict the model gives)
    ("skip_11000_11105_n16", 11000, 11105, 16, "mocha-4"),         # 8 -> 9 validators, 8 of 9 signed
    ("skip_15000_50000_n128", 15000, 50000, 128, "mocha-4"),       # 34 -> 100 validators, 47 of 100 signed: the 2/3 check fails
    ("skip_50000_157000_n128", 50000, 157000, 128, "mocha-4"),     # 100 -> 100 validators, 53 of 100 signed
]
SIGNED_BLOCK_HEIGHTS = (10002, 10003, 10004, 11000, 11001, 11105, 15000, 50000, 157000)
STEP_CASES = [  # reference tests: step.rs:170-268
    ("step_3000_n4", 3000, 4, "mocha-4"),
    ("step_10000_n2", 10000, 2, "mocha-4"),
    ("step_10500_n4", 10500, 4, "mocha-4"),       # test_step_with_dummy: validator 2 voted nil
    ("step_10500_n100", 10500, 100, "mocha-4"),   # test_step_large shape (N = 100, not a power of two)
    ("step_10002_n2", 10002, 2, "mocha-4"),       # signed_block.json heights: 10002 -> 10003 -> 10004, 11000 -> 11001
    ("step_10003_n4", 10003, 4, "mocha-4"),
    ("step_11000_n8", 11000, 8, "mocha-4"),
]
FULL_ELEMS_MAX = 30000


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    fx = os.path.join(args.reference, "circuits", "fixtures", "mocha-4")
    out = os.path.join(os.path.dirname(HERE), "tests", "golden")
    os.makedirs(out, exist_ok=True)
    for h in (3000, 3001, 3100, 10000, 10001, 10500, 10501, 157001):
        dst = os.path.join(out, "fixtures", "mocha-4", str(h))
        os.makedirs(dst, exist_ok=True)
        for name in ("commit.json", "validators_1.json"):
            shutil.copyfile(os.path.join(fx, str(h), name), os.path.join(dst, name))
    for h in SIGNED_BLOCK_HEIGHTS:
        dst = os.path.join(out, "fixtures", "mocha-4", str(h))
        os.makedirs(dst, exist_ok=True)
        shutil.copyfile(os.path.join(fx, str(h), "signed_block.json"), os.path.join(dst, "signed_block.json"))
    f = m.FixtureFetcher(fx)
    cases = {}

    # ---- the nine signed_block.json heights as data: what each pins on its own (SURVEY headline fact 4, verified by probe in round 1, now committed)
    sb = {}
    for h in SIGNED_BLOCK_HEIGHTS:
        sh, vs = f.signed_header(h), f.validators(h)
        leaves = tm.header_leaves(sh["header"])
        hh = tm.root_from_leaf_hashes([tm.leaf_hash(x) for x in leaves])
        vh = tm.root_from_leaf_hashes([tm.leaf_hash(tm.validator_bytes(tm.b64(v["pub_key"]["value"]), int(v["voting_power"]))) for v in vs])
        assert hh.hex().upper() == sh["commit"]["block_id"]["hash"] and vh.hex().upper() == sh["header"]["validators_hash"], h
        bid = sh["commit"]["block_id"]
        block_id = (bytes.fromhex(bid["hash"]), int(bid["parts"]["total"]), bytes.fromhex(bid["parts"]["hash"]))
        lanes = []
        for v, cs in zip(vs, sh["commit"]["signatures"]):
            if cs["block_id_flag"] != 2:
                lanes.append(None)
                continue
            pk, sig = tm.b64(v["pub_key"]["value"]), tm.b64(cs["signature"])
            msg = tm.sign_bytes(sh["header"]["chain_id"], int(sh["commit"]["height"]), int(sh["commit"]["round"]), block_id, cs["timestamp"])
            t = ed.verify_trace(pk, sig, msg)
            mul8 = lambda pt: ed.scalarmult(8, pt)
            cofactored = t["A"] is not None and t["s"] < ed.L and mul8(t["sB"]) == ed.add(mul8(t["R"]), mul8(t["hA"]))   # RFC 8032 5.1.7
            lanes.append(dict(pubkey=pk.hex(), signature=sig.hex(), message=msg.hex(), h=t["h"].to_bytes(32, "little").hex(),
                              ok_cofactorless=bool(t["ok"]), ok_rfc8032=bool(cofactored)))
        sb[str(h)] = dict(header_hash=hh.hex(), validators_hash=vh.hex(), next_validators_hash=sh["header"]["next_validators_hash"].lower(),
                          validators=len(vs), flags=[cs["block_id_flag"] for cs in sh["commit"]["signatures"]], lanes=lanes)
        print("signed_block", h, len(vs), "validators,", sum(x is not None for x in lanes), "signed, all verify:",
              all(x["ok_cofactorless"] and x["ok_rfc8032"] for x in lanes if x))
    with open(os.path.join(out, "signed_blocks.json"), "w") as fh:
        json.dump(sb, fh, indent=0, sort_keys=True)

    def finish(name, kind, n, chain_id, proof, target, trusted):
        w, rep = m.witness(kind, proof, target, trusted, chain_id.encode(), 100800)
        arr = np.array(w, dtype=np.uint64)
        cases[name] = dict(kind=kind, n=n, chain_id=chain_id, skip_max=100800, proof=proof.hex(), target=b"".join(target).hex(),
                           trusted=(b"".join(trusted).hex() if trusted else None), header=rep["header"].hex(),
                           all_ok=rep["all_ok"], fail_mask=rep["fail_mask"], first_bad_sig=rep["first_bad_sig"],
                           gt_target=rep["gt_target"], gt_trusted=rep["gt_trusted"], elem_count=len(w),
                           elems_sha256=hashlib.sha256(arr.tobytes()).hexdigest())
        if len(w) <= FULL_ELEMS_MAX:
            np.savez_compressed(os.path.join(out, f"elems_{name}.npz"), elems=arr)
        print(name, len(w), rep["header"].hex()[:16], rep["all_ok"], bin(rep["fail_mask"]))

    for name, a, b, n, cid in SKIP_CASES:
        proof, target, trusted = m.skip_inputs_from_fixtures(f, a, b, n)
        finish(name, m.KIND_SKIP, n, cid, proof, target, trusted)
    for name, prev, n, cid in STEP_CASES:
        proof, target = m.step_inputs_from_fixtures(f, prev, n)
        finish(name, m.KIND_STEP, n, cid, proof, target, None)
    with open(os.path.join(out, "cases.json"), "w") as fh:
        json.dump(cases, fh, indent=1, sort_keys=True)

    kat = dict(
        # reference circuits/builder/shared.rs:236-250 (test_marshal_int64_varint)
        varint=[[1, "01"], [3804, "dc1d"], [1234567890, "d285d8cc04"], [38957235239, "a7f8a0909101"],
                [9999999999999, "ffbfcaf384a302"], [724325643436111, "cf80b7a5d3d8a401"], [9223372036854775807, "ffffffffffffffff7f"]],
        # reference circuits/builder/validator.rs:282-287 (test_marshal_tendermint_validator)
        marshal=dict(pubkey="de25aec935b10f657b43fa97e5a8d4e523bdb0f9972605f0b064eff7b17048ba", power=100010,
                     expected="0a220a20de25aec935b10f657b43fa97e5a8d4e523bdb0f9972605f0b064eff7b17048ba10aa8d06"),
        # reference circuits/builder/validator.rs:333-338 (test_generate_validators_hash); expected root = native
        # RFC-6962 root of the byte slices (validator.rs:359-362)
        validators_hash=[[
            "0a220a20de25aec935b10f657b43fa97e5a8d4e523bdb0f9972605f0b064eff7b17048ba10aa8d06",
            "0a220a208de6ad1a569a223e7bb0dade194abb9487221210e1fa8154bf654a10fe6158a610aa8d06",
            "0a220a20e9b7638ca1c42da37d728970632fda77ec61dcc520395ab5d3a645b9c2b8e8b1100a",
            "0a220a20bd60452e7f056b22248105e7fd298961371da0d9332ef65fa81691bf51b2e5051001"], [
            "364db94241a02b701d0dc85ac016fab2366fba326178e6f11d8294931969072b7441fd6b0ff5129d6867",
            "6fa0cef8f328eb8e2aef2084599662b1ee0595d842058966166029e96bd263e5367185f19af67b099645ec08aa",
            "0a220a20bd60452e7f056b22248105e7fd298961371da0d9332ef65fa81691bf51b2e5051001",
            "0a220a20bd60452e7f056b22248105e7fd298961371da0d9332ef65fa81691bf51b2e5051001"]],
        # reference circuits/builder/verify.rs:597-601 (test_verify_hash_in_message)
        hash_in_message=dict(
            header="8909e1b73b7d987e95a7541d96ed484c17a4b0411e98ee4b7c890ad21302ff8c", round=0,
            message="6b080211de3202000000000022480a208909e1b73b7d987e95a7541d96ed484c17a4b0411e98ee4b7c890ad21302ff8c12240801122061263df4855e55fcab7aab0a53ee32cf4f29a1101b56de4a9d249d44e4cf96282a0b089dce84a60610ebb7a81932076d6f6368612d33"),
        # reference circuits/builder/voting.rs:127-146 (test_accumulate_voting_power): powers, in_group, expected (2/3)
        threshold=[[[10, 10, 10, 10], [1, 1, 1, 0], True], [[10, 10, 10, 10], [1, 1, 1, 1], True],
                   [[4294967296000, 4294967296, 10, 10], [1, 0, 0, 0], True],
                   [[4294967296000, 4294967296000, 4294967296000, 0], [1, 1, 0, 0], False],
                   [[4294967296000, 4294967296000, 4294967296000, 0], [0, 0, 0, 0], False]],
        # public inputs of the reference's end-to-end tests and the outputs the fixtures imply
        # (skip.rs:197-199, 259-262; step.rs:178-180, 237-240, 250-253)
        public_io=dict(
            skip=[["0000000000000bb8a8512f18c34b70e1533cfd5aa04f251fcb0d7be56ec570051fbad9bdb9435e6a0000000000000c1c",
                   "9b59dfd5ad4ef2c258c81fd1c25a46f99d2ce609100dc128ca3d065261c4c657"],
                  ["0000000000002710a0123d5e4b8b8888a61f931ee2252d83568b97c223e0eca9795b29b8bd8cba2d0000000000002904",
                   "e2ba1b86926925a69c2fcc32e5178e7e6653d386c956bb975142fa73211a9444"]],
            step=[["0000000000000bb8a8512f18c34b70e1533cfd5aa04f251fcb0d7be56ec570051fbad9bdb9435e6a",
                   "5121dc1ed961f6dc518992a3b61d6ccabb9ea2750d50d21a67d66f3d9c81a3cd"],
                  ["0000000000002710a0123d5e4b8b8888a61f931ee2252d83568b97c223e0eca9795b29b8bd8cba2d",
                   "f2a340cc2aef6fe163254b326a52334b45793eb11417029f9548418f88b38e26"],
                  ["0000000000002904e2ba1b86926925a69c2fcc32e5178e7e6653d386c956bb975142fa73211a9444",
                   "cd3e0f3e47fdac9abe1c98cf6be241bc23a8779e67df068832f7f43e2db7b05b"]]),
        # RFC 8032 §7.1 TEST 1-3 (seed, public key, message, signature)
        rfc8032=[
            ["9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60",
             "d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a", "",
             "e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e065224901555fb8821590a33bacc61e39701cf9b46bd25bf5f0595bbe24655141438e7a100b"],
            ["4ccd089b28ff96da9db6c346ec114e0f5b8a319f35aba624da8cf6ed4fb8a6fb",
             "3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c", "72",
             "92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00"],
            ["c5aa8df43f9f837bedb7442f31dcb7b166d38535076f094b85ce3a2e0b4458f7",
             "fc51cd8e6218a1a38da47ed00230f0580816ed13ba3303ac5deb911548908025", "af82",
             "6291d657deec24024827e69c3abe01a30ce548a284743a445e3680d7db5ac3ac18ff9b538d16f290ae67f760984dc6594a7c15e9716ed28dc027beceea1ec40a"]],
        dummy=dict(seed="01" * 32, message="00" * 32, public_key=ed.DUMMY_PUBLIC_KEY.hex(), signature=ed.DUMMY_SIGNATURE.hex()),
    )
    # expected roots for the validators-hash table, computed the way the reference's test does (native RFC 6962)
    kat["validators_hash_roots"] = [tm.root_from_leaf_hashes([tm.leaf_hash(bytes.fromhex(x)) for x in batch]).hex()
                                    for batch in kat["validators_hash"]]
    with open(os.path.join(out, "kat.json"), "w") as fh:
        json.dump(kat, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
