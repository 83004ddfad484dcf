"""The C ABI driven from a compiled C host (no Python, no torch in the process): what a Rust/C maintainer's FFI sees.
The host links libtmx.so and the ROCm HIP runtime itself (libtmx.so deliberately has no DT_NEEDED on it -- INTEGRATION.md).
Every entry point a hint body would bind is driven from C and its FULL output compared with the oracle's, element by element:
tmx_skip_witness (SkipOffchainInputs::hint, reference circuits/skip.rs:64-102), tmx_step_witness (StepOffchainInputs::hint,
circuits/step.rs:56-89), tmx_witness_batch_opts(TMX_SEC_HINT, TMX_OUT_U32), and two host threads with a context each (the calling
pattern of the reference's async hints on a tokio runtime, skip.rs:37-44)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
FX = os.path.join(GOLDEN, "fixtures", "mocha-4")


@pytest.fixture(scope="module")
def host(built_lib, tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("c_host") / "witness_host")
    libdir = os.path.join(ROOT, "tendermintx_amd")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "c_host", "witness_host.c"), "-I" + os.path.join(ROOT, "include"),
                           "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", "-L" + libdir, "-ltmx", "-L/opt/rocm/lib", "-lamdhip64", "-lstdc++", "-lpthread", "-Wl,-rpath," + libdir + ":/opt/rocm/lib"])
    return exe


def _fields(line):
    f = line.split()
    return dict(zip(f[::2], f[1::2]))


def _oracle_row(oracle, c):
    want, rep = oracle.witness(c["kind"], bytes.fromhex(c["proof"]), bytes.fromhex(c["target"]), bytes.fromhex(c["trusted"]) if c["trusted"] else None,
                               c["chain_id"].encode(), c["skip_max"])
    return want, rep


@pytest.mark.parametrize("name,a,b,n", [("skip_10000_10500_n4", 10000, 10500, 4), ("skip_3000_3100_n4", 3000, 3100, 4),
                                        ("skip_10000_10500_n32", 10000, 10500, 32), ("skip_10500_157001_n128", 10500, 157001, 128)])
def test_skip_from_c_host(host, oracle, cases, tmp_path, name, a, b, n):
    c = cases[name]
    trusted_hash = c["proof"][32:96]  # bytes 16..48 of the proof record = the public trusted header hash
    out = str(tmp_path / "row.bin")
    lines = subprocess.check_output([host, "skip", FX, str(n), "mocha-4", out, str(a), trusted_hash, str(b)]).decode().split("\n")
    assert lines[0] == "header " + c["header"]
    f = _fields(lines[1])
    assert (f["all_ok"], f["fail_mask"], f["first_bad_sig"]) == (str(int(c["all_ok"])), str(c["fail_mask"]), str(c["first_bad_sig"]))
    want, _ = _oracle_row(oracle, c)
    got = np.fromfile(out, dtype=np.uint64)
    assert got.size == c["elem_count"] == int(f["elems"]) and np.array_equal(got, want)          # the full row, not a checksum
    k = _fields(lines[2].replace("key_cache ", ""))
    assert k["enabled"] == "1" and int(k["resident"]) == int(k["last_new"]) > 0                   # the cold call made its keys resident


@pytest.mark.parametrize("name,prev,n", [("step_10000_n2", 10000, 2), ("step_3000_n4", 3000, 4), ("step_10500_n4", 10500, 4), ("step_10500_n100", 10500, 100)])
def test_step_from_c_host(host, oracle, cases, tmp_path, name, prev, n):
    c = cases[name]
    out = str(tmp_path / "row.bin")
    lines = subprocess.check_output([host, "step", FX, str(n), "mocha-4", out, str(prev), c["proof"][32:96]]).decode().split("\n")
    assert lines[0] == "header " + c["header"]
    f = _fields(lines[1])
    assert (f["all_ok"], f["fail_mask"], f["first_bad_sig"]) == (str(int(c["all_ok"])), str(c["fail_mask"]), str(c["first_bad_sig"]))
    want, _ = _oracle_row(oracle, c)
    got = np.fromfile(out, dtype=np.uint64)
    assert got.size == c["elem_count"] and np.array_equal(got, want)


def test_hint_section_as_u32_from_c_host(host, oracle, cases, tmp_path, built_lib):
    """tmx_witness_batch_opts(TMX_SEC_HINT, TMX_OUT_U32): exactly the elements SkipOffchainInputs::hint writes to its output stream
    (VerifySkipVariable<N>, reference circuits/variables.rs:91-105), narrowed to u32"""
    c = cases["skip_10000_10500_n32"]
    out = str(tmp_path / "hint.bin")
    subprocess.check_call([host, "hint32", FX, "32", "mocha-4", out, "10000", c["proof"][32:96], "10500"], stdout=subprocess.DEVNULL)
    want, _ = _oracle_row(oracle, c)
    hint = int(built_lib.tmx_hint_elem_count(0, 32))
    got = np.fromfile(out, dtype=np.uint32)
    assert hint == 1776 * 32 + 5320 and got.size == hint and np.array_equal(got.astype(np.uint64), want[:hint])


@pytest.mark.parametrize("kind,a,b,n", [(0, 10000, 10500, 4), (0, 10000, 10500, 32), (0, 10500, 157001, 128), (0, 10500, 157001, 512),
                                         (1, 10000, 0, 2), (1, 10500, 0, 4), (1, 10500, 0, 100), (1, 10500, 0, 512)])
def test_typed_value_from_c_host(host, oracle, built_lib, tmp_path, kind, a, b, n):
    """tmx_skip_inputs_value / tmx_step_inputs_value from compiled C into page-locked memory (tmx_host_alloc): the packed SkipInputs<F> /
    StepInputs<F> value (reference circuits/input/mod.rs:45-74) equals the oracle's byte for byte, and -- expanded by THIS test by the
    reference's element rules -- reproduces section H (and, through the derived part, the whole row) at N = 4 / 32 / 128 / 512 and for step"""
    import ctypes as C
    from tendermintx_amd import _lib
    from tendermintx_amd.circuits import InputDataFetcher
    from test_typed_value import expand_derived, expand_hint
    f = InputDataFetcher(FX)
    import tmx_model as m
    pf = m.FixtureFetcher(FX)
    if kind == 0:
        pr, tg, tr = m.skip_inputs_from_fixtures(pf, a, b, n)
        targets, trusteds = b"".join(tg), b"".join(tr)
        h = m.unpack_proof(pr)["hash"].hex()
        args = [host, "value", FX, str(n), "mocha-4", str(tmp_path / "v.bin"), str(a), h, str(b)]
    else:
        pr, tg = m.step_inputs_from_fixtures(pf, a, n)
        targets, trusteds = b"".join(tg), None
        h = m.unpack_proof(pr)["hash"].hex()
        args = [host, "stepvalue", FX, str(n), "mocha-4", str(tmp_path / "v.bin"), str(a), h]
    r = subprocess.run(args, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.split("\n")
    want, orep = oracle.witness_value(kind, pr, targets, trusteds, b"mocha-4", 100800, True)
    got = np.fromfile(str(tmp_path / "v.bin"), dtype=np.uint8)
    assert got.size == want.size and np.array_equal(got, want)
    assert lines[0] == "header " + orep["header"].hex() and _fields(lines[1])["all_ok"] == str(int(orep["all_ok"]))   # (10500 -> 157001 is too far: dist_ok fails, as in the goldens)
    named = _fields(lines[2])
    row, _ = oracle.witness(kind, pr, targets, trusteds, b"mocha-4", 100800)
    lay = _lib.ValueLayout()
    assert built_lib.tmx_value_layout_of(kind, n, _lib.SEC_ALL, C.byref(lay)) == 0
    hint = int(built_lib.tmx_hint_elem_count(kind, n))
    assert np.array_equal(expand_hint(_lib, kind, n, got, lay), row[:hint])
    assert np.array_equal(expand_derived(_lib, kind, n, got, lay), row[hint:])
    # the fields the C program read by name are the record's
    if kind == 0:
        assert named["chain_id"] == "mocha-4" and int(named["height"]) == b and int(named["nb_target"]) == m.unpack_proof(pr)["nb_a"]
        print("typed value from C host, N =", n, "host-to-host ms per call:", named["ms_per_call"])
    else:
        assert int(named["height"]) == a + 1 and int(named["nb_validators"]) == m.unpack_proof(pr)["nb_a"]


def test_two_host_threads_two_contexts(host, oracle, cases, tmp_path):
    """Two threads, a context each, 40 skip witnesses each at the same time: every row equals the single-threaded one and the oracle's
    (the contexts share the device's three internal side streams; each owns its scratch, events and key cache)."""
    c = cases["skip_10000_10500_n32"]
    out = str(tmp_path / "row.bin")
    r = subprocess.run([host, "threads", FX, "32", "mocha-4", out, "10000", c["proof"][32:96], "10500", "40"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    print(r.stdout.strip())          # mismatch counts, ms per call alone and with both threads running (-s shows it; DESIGN.md quotes it)
    want, _ = _oracle_row(oracle, c)
    assert np.array_equal(np.fromfile(out, dtype=np.uint64), want)


def test_sharded_entry_points_from_c_host(host, oracle, cases, tmp_path):
    """tmx_comm_unique_id -> tmx_comm_create -> tmx_witness_validator_sharded_device / tmx_witness_batch_sharded_device from compiled C with
    device buffers of its own: the exchange runs through a real (one-rank) RCCL communicator that libtmx dlopens -- the program does not
    link librccl -- and the row equals the oracle's (the Rust shim binds exactly these calls: rust-shim/gpu.rs.example)"""
    c = cases["skip_10000_10500_n32"]
    out = str(tmp_path / "row.bin")
    r = subprocess.run([host, "sharded", FX, "32", "mocha-4", out, "10000", c["proof"][32:96], "10500"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.split("\n") if ln.startswith(("header ", "all_ok ", "rank "))]   # (RCCL prints a version banner of its own)
    assert lines[0] == "header " + c["header"]
    f = _fields(lines[2])
    assert (f["rank"], f["world"], f["shard_lo"], f["shard_hi"], f["rows_equal"]) == ("0", "1", "0", "32", "1")
    want, _ = _oracle_row(oracle, c)
    assert np.array_equal(np.fromfile(out, dtype=np.uint64), want)
