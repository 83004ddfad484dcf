"""Robustness of the C++ fixture / RPC JSON codec (tendermintx_amd/csrc/codec.cpp): mutated and truncated JSON must come back
as a status code (TMX_OK or TMX_ERR_PARSE / SET_TOO_LARGE / MSG_TOO_LONG / BAD_ARG), never as a crash or an out-of-bounds access.
The codec is compiled on its own under -fsanitize=address,undefined and driven in a subprocess (the sanitizer runtime loads first).
TMX_CODEC_FUZZ=N mutations per entry point (default 1500)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = os.path.join(ROOT, "tests", "golden", "fixtures", "mocha-4")

DRIVER = r'''
import ctypes as C, os, random, sys
lib, fx, n_iter = C.CDLL(sys.argv[1]), sys.argv[2], int(sys.argv[3])
def rd(h, f): return open(os.path.join(fx, str(h), f), "rb").read()
tc, tv = rd(10000, "commit.json"), rd(10000, "validators_1.json")
gc, gv = rd(10500, "commit.json"), rd(10500, "validators_1.json")
pc = rd(10499, "commit.json") if os.path.exists(os.path.join(fx, "10499", "commit.json")) else gc
P = (C.c_uint8 * 2336)(); T = (C.c_uint8 * (256 * 8))(); R = (C.c_uint8 * (48 * 8))()
rng = random.Random(20260929)
TOKENS = [b"{", b"}", b"[", b"]", b'"', b":", b",", b"null", b"true", b"-1", b"1e999", b"99999999999999999999999999", b'"\\u0000"', b'"\\"',
          b'"AAAA"', b'""', b"\x00", b"\xff", b'"' + b"A" * 300 + b'"', b"[" * 200, b'{"a":' * 100]
def mutate(b):
    b = bytearray(b)
    for _ in range(rng.randint(1, 4)):
        k = rng.randint(0, 7)
        i = rng.randrange(len(b)) if b else 0
        if k == 0 and b: b[i] ^= 1 << rng.randrange(8)
        elif k == 1 and b: b[i] = rng.randrange(256)
        elif k == 2: b = b[:i]
        elif k == 3 and b: del b[i:i + rng.randint(1, 64)]
        elif k == 4: b[i:i] = rng.choice(TOKENS)
        elif k == 5 and b: b[i:i + rng.randint(1, 32)] = rng.choice(TOKENS)
        elif k == 6 and b: j = rng.randrange(len(b)); b[i:i] = b[j:j + rng.randint(1, 200)]
        else: b = b[rng.randrange(len(b) + 1):] if b else b
    return bytes(b)
ok = {0, -1, -2, -4, -5, -6}
seen = {}
for it in range(n_iter):
    docs = [tc, tv, gc, gv]
    for j in rng.sample(range(4), rng.randint(1, 2)): docs[j] = mutate(docs[j])
    n_max = rng.choice((1, 3, 4, 8))
    st = lib.tmx_skip_inputs_from_json(docs[0], docs[1], docs[2], docs[3], C.c_uint32(n_max), C.c_uint64(10000), bytes(32), C.c_uint64(10500), P, T, R)
    assert st in ok, st
    seen[st] = seen.get(st, 0) + 1
    docs = [pc, gc, gv]
    docs[rng.randrange(3)] = mutate(docs[rng.randrange(3)])
    st = lib.tmx_step_inputs_from_json(docs[0], docs[1], docs[2], C.c_uint32(n_max), C.c_uint64(10499), bytes(32), P, T)
    assert st in ok, st
    seen[st] = seen.get(st, 0) + 1
A, B_, S = (C.c_uint8 * (64 * 8))(), (C.c_uint8 * (64 * 8))(), (C.c_uint8 * (64 * 8))()
na, nb, ns = C.c_uint32(), C.c_uint32(), C.c_uint32()
for it in range(n_iter // 2):
    docs = [tv, gv, gc]
    docs[rng.randrange(3)] = mutate(docs[rng.randrange(3)])
    st = lib.tmx_skipcheck_inputs_from_json(docs[0], docs[1], docs[2], C.c_uint32(rng.choice((1, 4, 8))), A, C.byref(na), B_, C.byref(nb), S, C.byref(ns))
    assert st in ok, st
    seen[("skipcheck", st)] = seen.get(("skipcheck", st), 0) + 1
print("codec fuzz ok", sorted(seen.items(), key=str))
'''


def test_codec_survives_mutated_json_under_sanitizers(tmp_path):
    so = str(tmp_path / "libcodec_asan.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tendermintx_amd", "csrc", "codec.cpp"), "-o", so])
    asan_rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan_rt, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([sys.executable, "-c", DRIVER, so, FX, os.environ.get("TMX_CODEC_FUZZ", "1500")], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0 and "codec fuzz ok" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])
