"""The C ABI driven from a compiled C host (no Python, no torch in the process): what a Rust/C maintainer's FFI sees.
The host links libtmx.so and the ROCm HIP runtime itself (libtmx.so deliberately has no DT_NEEDED on it -- INTEGRATION.md)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def test_skip_from_c_host(built_lib, oracle, cases, tmp_path):
    exe = str(tmp_path / "skip_host")
    libdir = os.path.join(ROOT, "tendermintx_amd")
    subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(ROOT, "tests", "c_host", "skip_host.c"), "-I" + os.path.join(ROOT, "include"),
                           "-L" + libdir, "-ltmx", "-L/opt/rocm/lib", "-lamdhip64", "-lstdc++", "-Wl,-rpath," + libdir + ":/opt/rocm/lib"])
    fx = os.path.join(GOLDEN, "fixtures", "mocha-4")
    for name, a, b, n in [("skip_10000_10500_n4", 10000, 10500, 4), ("skip_3000_3100_n4", 3000, 3100, 4), ("skip_10000_10500_n32", 10000, 10500, 32)]:
        c = cases[name]
        trusted_hash = c["proof"][32:96]  # bytes 16..48 of the proof record = the public trusted header hash
        out = subprocess.check_output([exe, fx, str(a), trusted_hash, str(b), str(n), "mocha-4"]).decode().split("\n")
        assert out[0] == "header " + c["header"]
        fields = dict(zip(out[1].split()[::2], out[1].split()[1::2]))
        assert fields["all_ok"] == "1" and fields["fail_mask"] == "0" and fields["first_bad_sig"] == "-1"
        assert int(fields["elems"]) == c["elem_count"]
        want, _ = oracle.witness(c["kind"], bytes.fromhex(c["proof"]), bytes.fromhex(c["target"]), bytes.fromhex(c["trusted"]), b"mocha-4", 100800)
        s = 0
        for v in want.tolist():
            s = (s * 1099511628211 + v) & (2**64 - 1)
        assert int(fields["checksum"]) == s
