"""Runs tests/ledger.py (the executable form of tests/ASSERTS.md): every assertion of the reference's skip / step circuits against
oracle/c and oracle/py here, and against the HIP path through the C ABI on the GPU box.  The expected verdicts come from the reference
text (ledger.ENTRIES), never from the implementation under test."""
import numpy as np
import pytest

import ledger

IDS = [e[0] for e in ledger.ENTRIES]


@pytest.fixture(scope="module")
def scenarios():
    return {e[0]: ledger.build(e[2], **e[3]) for e in ledger.ENTRIES}


def test_ledger_ids_are_unique_and_documented():
    import os
    assert len(set(IDS)) == len(IDS)
    text = open(os.path.join(os.path.dirname(__file__), "ASSERTS.md")).read()
    missing = [i for i in IDS if f"`{i}`" not in text]
    assert not missing, f"ledger entries without a line in tests/ASSERTS.md: {missing}"


@pytest.mark.parametrize("entry", ledger.ENTRIES, ids=IDS)
def test_oracle_c_gives_the_verdict_the_reference_text_implies(oracle, scenarios, entry):
    eid, _, kind, _, expect = entry
    sc = scenarios[eid]
    _, rep = oracle.witness(kind, sc["proof"], sc["targets"], sc["trusteds"], sc["chain_id"], sc["skip_max"])
    ledger.check(eid, rep, expect)
    assert rep["header"] == sc["header"], eid     # Level-0 output: the header hash as hashlib computes it


@pytest.mark.parametrize("entry", ledger.ENTRIES, ids=IDS)
def test_python_model_agrees_with_oracle_c(oracle, scenarios, entry):
    import tmx_model as m
    eid, _, kind, _, expect = entry
    sc = scenarios[eid]
    n = sc["n"]
    t = [sc["targets"][256 * i:256 * (i + 1)] for i in range(n)]
    r = [sc["trusteds"][48 * i:48 * (i + 1)] for i in range(n)] if kind == ledger.SKIP else None
    w, rep = m.witness(kind, sc["proof"], t, r, sc["chain_id"], sc["skip_max"])
    want, orep = oracle.witness(kind, sc["proof"], sc["targets"], sc["trusteds"], sc["chain_id"], sc["skip_max"])
    assert np.array_equal(np.array(w, dtype=np.uint64), want), eid
    for k in ("all_ok", "fail_mask", "first_bad_sig", "gt_target"):
        assert rep[k] == orep[k], (eid, k)
    ledger.check(eid, dict(rep, gt_trusted=bool(rep["gt_trusted"]), dist_ok=orep["dist_ok"]), expect)


@pytest.mark.gpu
@pytest.mark.parametrize("entry", ledger.ENTRIES, ids=IDS)
def test_hip_path_gives_the_verdict_the_reference_text_implies(built_lib, oracle, scenarios, entry):
    import tendermintx_amd as tmx
    eid, _, kind, _, expect = entry
    sc = scenarios[eid]
    with tmx.Context(sc["n"], sc["chain_id"], sc["skip_max"], max_batch=1) as ctx:
        elems, reps = ctx.witness_batch(kind, sc["proof"], sc["targets"], sc["trusteds"])
    ledger.check(eid, reps[0], expect)
    assert reps[0]["header"] == sc["header"], eid
    want, orep = oracle.witness(kind, sc["proof"], sc["targets"], sc["trusteds"], sc["chain_id"], sc["skip_max"])
    assert np.array_equal(elems[0], want) and reps[0] == orep, eid


@pytest.mark.gpu
def test_hip_path_whole_ledger_in_one_batch(built_lib, oracle, scenarios):
    """The same entries as ONE batch per (kind, chain id): a failing proof must not leak into its neighbours' reports."""
    import tendermintx_amd as tmx
    for kind in (ledger.SKIP, ledger.STEP):
        group = [e for e in ledger.ENTRIES if e[2] == kind and scenarios[e[0]]["chain_id"] == b"celestia" and scenarios[e[0]]["n"] == 4]
        proofs = b"".join(scenarios[e[0]]["proof"] for e in group)
        targets = b"".join(scenarios[e[0]]["targets"] for e in group)
        trusteds = b"".join(scenarios[e[0]]["trusteds"] for e in group) if kind == ledger.SKIP else None
        with tmx.Context(4, b"celestia", ledger.SKIP_MAX, max_batch=len(group)) as ctx:
            _, reps = ctx.witness_batch(kind, proofs, targets, trusteds)
        for e, rep in zip(group, reps):
            ledger.check(e[0], rep, e[4])
