"""The persistent per-key table cache of a context (include/tmx.h "persistent per-key table cache", DESIGN.md "Key cache"): whatever the
cache does -- hit, miss, eviction, a key one bit away from a resident one, disabled, flushed, resized -- the witness is bit-exact vs the
CPU oracle, and the cache's own counters say which path the call took.  A light client re-verifies the same validator set from call to
call (reference bin/tendermintx.rs:171): the hint bodies `SkipOffchainInputs::hint` / `StepOffchainInputs::hint` (reference
circuits/skip.rs:64-102, circuits/step.rs:56-89) are called once per proof on a long-lived context."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tmx(built_lib):
    import tendermintx_amd
    return tendermintx_amd


def _check(ctx, oracle, wl, n, kind=0, targets=None):
    targets = wl.targets if targets is None else targets
    P = len(wl.proofs) // 2336
    elems, reps = ctx.witness_batch(kind, wl.proofs, targets, wl.trusteds)
    want, oreps = oracle.witness_batch(kind, P, wl.proofs, targets, wl.trusteds, n, b"celestia", 100800, n_threads=8)
    assert np.array_equal(elems, want), np.argwhere(elems != want)[:8].tolist()
    assert reps == oreps
    return elems, reps


def _sets(n, P, seeds, nb=None):
    from tendermintx_amd.synth import Workload
    return [Workload(0, n, P, nb or n, chain_id=b"celestia", seed=s, signed_permille=1000) for s in seeds]


def test_hit_and_miss_counters(tmx, oracle):
    """cold call: every key new, tables built; same batch again: every lane hits, nothing new; another validator set: all new again"""
    n, P = 32, 80                # 2560 lanes: above the 2048 up to which a launch takes the small path, which never waits for fresh tables
    a, b = _sets(n, P, (11, 12))
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        st0 = ctx.key_cache_stats()
        assert st0["enabled"] == 1 and st0["resident_keys"] == 0 and st0["capacity_keys"] >= 1024 and st0["bytes_per_key"] > 200_000
        ctx.key_cache_config(True, max_keys=4096)   # (room for more than one launch's worth of new keys: no eviction in this test)
        first, _ = _check(ctx, oracle, a, n)
        st = ctx.key_cache_stats()
        assert (st["last_new_keys"], st["last_hit_keys"], st["last_hit_lanes"], st["last_built_keys"]) == (n, 0, 0, n)
        assert st["resident_keys"] == n and ctx.last_dedup() == (n, True)      # 20 lanes per key: the fresh tables were walked
        again, _ = _check(ctx, oracle, a, n)
        st = ctx.key_cache_stats()
        assert (st["last_new_keys"], st["last_hit_keys"], st["last_hit_lanes"], st["last_built_keys"]) == (0, n, n * P, 0)
        assert st["resident_keys"] == n and ctx.last_dedup() == (n, True) and np.array_equal(first, again)
        _check(ctx, oracle, b, n)
        st = ctx.key_cache_stats()
        assert st["last_new_keys"] == n and st["last_hit_lanes"] == 0 and st["resident_keys"] == 2 * n
        assert st["hit_lanes"] == n * P and st["miss_lanes"] == 2 * n * P and st["built_keys"] == 2 * n and st["launches"] == 3


def test_single_proof_cold_then_warm(tmx, oracle):
    """BASELINE configs[2]: one proof at N = 128.  The cold call takes the table-free form and builds the tables off its critical path;
    the next call -- the same set, another block -- walks them."""
    from tendermintx_amd.synth import Workload
    n = 128
    wl = Workload(0, n, 3, 100, chain_id=b"celestia", seed=77, signed_permille=1000)   # 100 keys + the dummy key of lanes 100..127
    one = lambda p: type("W", (), dict(proofs=wl.proofs[p * 2336:(p + 1) * 2336], targets=wl.targets[p * n * 256:(p + 1) * n * 256],
                                       trusteds=wl.trusteds[p * n * 48:(p + 1) * n * 48]))
    with tmx.Context(n, b"celestia", max_batch=1) as ctx:
        _check(ctx, oracle, one(0), n)
        st = ctx.key_cache_stats()
        assert st["last_hit_lanes"] == 0 and st["last_built_keys"] == st["last_new_keys"] == 101 and ctx.last_dedup() == (101, False)
        _check(ctx, oracle, one(1), n)
        st = ctx.key_cache_stats()
        assert st["last_hit_lanes"] == n and st["last_new_keys"] == 0 and ctx.last_dedup()[1]
        _check(ctx, oracle, one(2), n)
        assert ctx.key_cache_stats()["hit_lanes"] == 2 * n


def test_key_one_bit_away_from_a_resident_key(tmx, oracle):
    """all 32 key bytes are compared on a probe: a key that differs from a resident one in a single bit -- the first bit, the last byte,
    the sign bit of x -- is a new key (its own decode, its own table), never the resident key's table"""
    n, P = 16, 4
    (a,) = _sets(n, P, (21,))
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        _check(ctx, oracle, a, n)
        for lane, byte, bit in ((0, 0, 0), (5, 31, 0), (9, 31, 7), (n + 3, 17, 4)):
            t = bytearray(a.targets)
            t[lane * 256 + byte] ^= 1 << bit
            _, reps = _check(ctx, oracle, a, n, targets=bytes(t))
            st = ctx.key_cache_stats()
            assert st["last_new_keys"] == 1 and st["last_hit_lanes"] == n * P - 1, (lane, byte, bit, st)
            assert not reps[lane // n]["all_ok"]                      # the signature was made with the original key
        _check(ctx, oracle, a, n)                                     # and the resident keys are still themselves
        assert ctx.key_cache_stats()["last_hit_lanes"] == n * P


def test_lru_eviction(tmx, oracle):
    """capacity 64 keys, validator sets of 16 keys: when the cache runs short of room for a launch's new keys the least recently USED set
    goes -- A, B, A (refreshed), C evicts B, not A -- and an evicted set is simply new again.  Bit-exact throughout."""
    n, P = 16, 2
    a, b, c, d = _sets(n, P, (31, 32, 33, 34))
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        ctx.key_cache_config(True, max_keys=64)
        assert ctx.key_cache_stats()["capacity_keys"] == 64
        for wl in (a, b, a):
            _check(ctx, oracle, wl, n)
        st = ctx.key_cache_stats()
        assert st["resident_keys"] == 32 and st["evicted_keys"] == 0 and st["last_hit_lanes"] == n * P
        _check(ctx, oracle, c, n)                                     # 48 resident, 16 free < 32 (the most a launch may add): evict
        st = ctx.key_cache_stats()
        assert st["evicted_keys"] == 16 and st["evictions"] == 1 and st["resident_keys"] == 32
        _check(ctx, oracle, a, n)
        assert ctx.key_cache_stats()["last_hit_lanes"] == n * P        # A survived (used more recently than B)
        _check(ctx, oracle, b, n)
        st = ctx.key_cache_stats()
        assert st["last_hit_lanes"] == 0 and st["last_new_keys"] == n  # B was evicted: new again
        for wl in (d, c, b, a, d, c):                                  # keep cycling through more sets than fit
            _check(ctx, oracle, wl, n)
        st = ctx.key_cache_stats()
        assert st["resident_keys"] <= 64 and st["evictions"] >= 3


def test_more_new_keys_than_room(tmx, oracle):
    """a launch with more new keys than free slots: the keys that get a slot walk tables, the others take the table-free form; nothing
    resident is lost that the launch itself uses"""
    n = 16
    (big,) = _sets(n, 6, (41,))
    from tendermintx_amd.synth import Workload
    many = Workload(0, n, 6, n, chain_id=b"celestia", seed=42, signed_permille=1000, n_sets=6)     # 96 distinct keys in one launch
    with tmx.Context(n, b"celestia", max_batch=6) as ctx:
        ctx.key_cache_config(True, max_keys=40)
        _check(ctx, oracle, many, n)
        st = ctx.key_cache_stats()
        assert st["last_new_keys"] == 96 and st["last_built_keys"] == 40 and st["resident_keys"] == 40
        _check(ctx, oracle, many, n)                                   # 40 keys hit, 56 are new again and find no room
        st = ctx.key_cache_stats()
        assert st["last_hit_keys"] == 40 and st["last_new_keys"] == 56 and st["last_built_keys"] == 0
        _check(ctx, oracle, big, n)
        _check(ctx, oracle, many, n)


def test_cache_disabled_flush_and_resize(tmx, oracle, monkeypatch):
    n, P = 16, 4
    a, b = _sets(n, P, (51, 52))
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        ctx.key_cache_config(False)
        for _ in range(2):                                             # disabled: every call is cold, nothing becomes resident
            _check(ctx, oracle, a, n)
            st = ctx.key_cache_stats()
            assert st["enabled"] == 0 and st["last_hit_lanes"] == 0 and st["last_new_keys"] == n and st["resident_keys"] == 0
        ctx.key_cache_config(True)
        _check(ctx, oracle, a, n)
        _check(ctx, oracle, a, n)
        assert ctx.key_cache_stats()["last_hit_lanes"] == n * P
        ctx.key_cache_flush()
        assert ctx.key_cache_stats()["resident_keys"] == 0
        _check(ctx, oracle, a, n)
        assert ctx.key_cache_stats()["last_new_keys"] == n
        ctx.key_cache_config(True, max_keys=20)                        # a new capacity flushes too
        st = ctx.key_cache_stats()
        assert st["capacity_keys"] == 20 and st["resident_keys"] == 0
        for wl in (a, b, a, b):
            _check(ctx, oracle, wl, n)
    monkeypatch.setenv("TMX_KEY_CACHE", "0")
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        _check(ctx, oracle, a, n)
        _check(ctx, oracle, a, n)
        st = ctx.key_cache_stats()
        assert st["enabled"] == 0 and st["hit_lanes"] == 0


def test_undecodable_and_edge_keys_are_cached_as_such(tmx, oracle):
    """a public key that does not decode (y with no x on the curve), y >= p, x = 0 with the sign bit set: the negative result is cached
    like any key, and a second call gives the same (zero-point) lane records as the first"""
    n, P = 16, 3
    (a,) = _sets(n, P, (61,))
    t = bytearray(a.targets)
    bad_keys = [bytes([2] + [0] * 31), bytes([0xff] * 32), bytes([1] + [0] * 30 + [0x80]), bytes([0xee] * 31 + [0x7f])]
    for k, pk in enumerate(bad_keys):
        for p in range(P):
            t[(p * n + 2 + k) * 256:(p * n + 2 + k) * 256 + 32] = pk
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        e1, r1 = _check(ctx, oracle, a, n, targets=bytes(t))
        e2, r2 = _check(ctx, oracle, a, n, targets=bytes(t))
        assert ctx.key_cache_stats()["last_hit_lanes"] == n * P and np.array_equal(e1, e2) and r1 == r2


def test_step_and_skip_share_the_cache(tmx, oracle):
    """a step proof of a validator set a skip proof has made resident hits (kind is not part of the key)"""
    from tendermintx_amd.synth import Workload
    n = 32
    sk = Workload(0, n, 2, n, chain_id=b"celestia", seed=71, signed_permille=1000)
    st_ = Workload(1, n, 2, n, chain_id=b"celestia", seed=71, signed_permille=1000)
    with tmx.Context(n, b"celestia", max_batch=2) as ctx:
        _check(ctx, oracle, sk, n, kind=0)
        _check(ctx, oracle, st_, n, kind=1)
        assert ctx.key_cache_stats()["last_hit_lanes"] == 2 * n


@pytest.mark.parametrize("fresh_proofs", [1, 4])
def test_a_few_proofs_bring_new_keys_into_a_warm_batch(tmx, oracle, fresh_proofs):
    """The daily churn of a validator set: a warm batch (the warm walk split by residency: resident lanes at once, the lanes of new keys behind
    their tables on the side streams) in which one / four proofs bring 40 keys each that the cache has not seen -- N = 64, 40 proofs, one of
    the new-key lanes with a corrupted signature.  Bit-exact vs the oracle, the failing lane found, and the counters say the keys are
    resident afterwards."""
    from tendermintx_amd.synth import Workload
    n, P = 64, 40
    base = Workload(0, n, P, 50, chain_id=b"celestia", seed=9001, signed_permille=900)
    fresh = Workload(0, n, fresh_proofs, 40, chain_id=b"celestia", seed=9100 + fresh_proofs, signed_permille=1000, n_sets=fresh_proofs)
    k = fresh_proofs
    proofs = fresh.proofs + base.proofs[k * 2336:]
    targets = bytearray(fresh.targets + base.targets[k * n * 256:])
    trusteds = fresh.trusteds + base.trusteds[k * n * 48:]
    lane = next(l for l in range(n) if targets[l * 256 + 223] & 1)
    targets[lane * 256 + 45] ^= 0x04      # a failing signature on a lane of a NEW key
    mixed = type("W", (), {"proofs": proofs, "targets": bytes(targets), "trusteds": trusteds})
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        _check(ctx, oracle, base, n)
        _check(ctx, oracle, base, n)          # warm: the schedule hint now says so
        st0 = ctx.key_cache_stats()
        _, reps = _check(ctx, oracle, mixed, n)
        st1 = ctx.key_cache_stats()
        assert reps[0]["first_bad_sig"] == lane and all(r["first_bad_sig"] == -1 for r in reps[1:])
        assert st1["last_new_keys"] == 40 * fresh_proofs and st1["resident_keys"] == st0["resident_keys"] + 40 * fresh_proofs
        _check(ctx, oracle, mixed, n)          # the same batch again: every key resident, its table built by the call above
        assert ctx.key_cache_stats()["last_new_keys"] == 0


@pytest.mark.gpu
def test_validator_set_cache(tmx, oracle, monkeypatch):
    """The validator-set cache of a context (include/tmx.h tmx_set_cache_stats; layout.h SetCache): a batch of 24 proofs over three validator
    sets computes each (target, trusted) set once per workgroup that finds it missing and inserts it once; the second call serves all 48 sets
    from the cache; a set that differs in ONE byte (a voting power) is a different set; rows bit-exact vs the oracle every time, and equal to a
    context without the cache (TMX_SET_CACHE=0)."""
    from tendermintx_amd.synth import Workload
    n, P = 128, 24
    wl = Workload(0, n, P, 100, chain_id=b"celestia", seed=9001, signed_permille=900, n_sets=3)
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        e1, _ = ctx.witness_batch(0, wl.proofs, wl.targets, wl.trusteds)
        s1 = ctx.set_cache_stats()
        assert s1["served"] == 0 and s1["computed"] == 2 * P and 1 <= s1["resident"] <= 2 * P and s1["inserted"] == s1["resident"]
        e2, _ = ctx.witness_batch(0, wl.proofs, wl.targets, wl.trusteds)
        s2 = ctx.set_cache_stats()
        assert s2["served"] == 2 * P and s2["computed"] == s1["computed"] and s2["resident"] == s1["resident"]
        want, _ = oracle.witness_batch(0, P, wl.proofs, wl.targets, wl.trusteds, n, b"celestia", 100800, n_threads=8)
        assert np.array_equal(e1, want) and np.array_equal(e2, want)
        # one byte of one lane's voting power in proof 5's target set: that proof's target set is new, everything else is served
        t2 = bytearray(wl.targets)
        t2[(5 * n + 17) * 256 + 224] ^= 1
        e3, _ = ctx.witness_batch(0, wl.proofs, bytes(t2), wl.trusteds)
        s3 = ctx.set_cache_stats()
        assert s3["computed"] == s2["computed"] + 1 and s3["served"] == s2["served"] + 2 * (P - 1) and s3["resident"] == s2["resident"] + 1
        want3, _ = oracle.witness_batch(0, P, wl.proofs, bytes(t2), wl.trusteds, n, b"celestia", 100800, n_threads=8)
        assert np.array_equal(e3, want3)
        ctx.key_cache_flush()
        assert ctx.set_cache_stats() == {"resident": 0, "served": 0, "computed": 0, "inserted": 0, "evicted": 0, "capacity": 256}
    monkeypatch.setenv("TMX_SET_CACHE", "0")
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        e4, _ = ctx.witness_batch(0, wl.proofs, wl.targets, wl.trusteds)
        assert ctx.set_cache_stats()["computed"] == 0 and np.array_equal(e4, want)


@pytest.mark.gpu
def test_validator_set_cache_evicts_least_recently_used(tmx, oracle, monkeypatch):
    """A prover process lives for months (reference bin/tendermintx.rs:171) and sees more than 256 validator sets: the set cache evicts the
    least recently used ones at launch granularity (layout.h SetCache; the last workgroup of a k_proof launch keeps an eighth of the slots
    free).  300 distinct target sets go through the 256-slot cache in twelve batches of 25 proofs (a set = the base set with one voting
    power changed: every byte of a set is its key); then the LAST 200 are served from the cache -- nothing recomputed --, the oldest are
    gone, and every row of every call equals the oracle's.  A 16-slot cache (TMX_SET_CACHE_SETS) thrashes and still gives the same bits."""
    from tendermintx_amd.synth import Workload
    n, P, B = 128, 25, 12
    wl = Workload(0, n, P, 100, chain_id=b"celestia", seed=9100, signed_permille=900, n_sets=1)

    def batch(j):   # proof q of batch j: its own target set (one power word differs in lane q)
        t = bytearray(wl.targets)
        for q in range(P):
            off = (q * n + q) * 256 + 224
            t[off:off + 4] = (int.from_bytes(t[off:off + 4], "little") ^ (1 + j * P + q)).to_bytes(4, "little")
        return bytes(t)

    targets = [batch(j) for j in range(B)]
    want = [oracle.witness_batch(0, P, wl.proofs, t, wl.trusteds, n, b"celestia", 100800, n_threads=8)[0] for t in targets]
    for sets, expect_all_served in ((None, True), ("16", False)):
        if sets:
            monkeypatch.setenv("TMX_SET_CACHE_SETS", sets)
        with tmx.Context(n, b"celestia", max_batch=P) as ctx:
            cap = ctx.set_cache_stats()["capacity"]
            assert cap == (int(sets) if sets else 256)
            for j in range(B):
                e, _ = ctx.witness_batch(0, wl.proofs, targets[j], wl.trusteds)
                assert np.array_equal(e, want[j]), (sets, j)
            s = ctx.set_cache_stats()
            assert s["computed"] >= B * P and s["evicted"] > 0 and s["resident"] == s["inserted"] - s["evicted"] <= cap - cap // 8
            if expect_all_served:
                assert s["inserted"] == B * P + 1          # 300 target sets + the one trusted set: nothing was refused for want of room
            for j in range(B - 8, B):                      # the last 200 sets again
                before = ctx.set_cache_stats()
                e, _ = ctx.witness_batch(0, wl.proofs, targets[j], wl.trusteds)
                assert np.array_equal(e, want[j]), (sets, "again", j)
                after = ctx.set_cache_stats()
                if expect_all_served:
                    assert after["computed"] == before["computed"] and after["served"] == before["served"] + 2 * P, j
                    assert after["evicted"] == before["evicted"] and after["resident"] == before["resident"]
            if expect_all_served:                          # the oldest sets are gone: batch 0 is recomputed (and inserted again)
                before = ctx.set_cache_stats()
                e, _ = ctx.witness_batch(0, wl.proofs, targets[0], wl.trusteds)
                after = ctx.set_cache_stats()
                assert np.array_equal(e, want[0]) and after["computed"] == before["computed"] + P


@pytest.mark.gpu
def test_validator_set_cache_random_stream_through_a_tiny_cache(tmx, oracle, monkeypatch):
    """LRU under churn: an 8-slot set cache (TMX_SET_CACHE_SETS) and 150 calls of 20 proofs drawn at random from 48 proofs over 48 different
    target sets (and one trusted set): every call hits, misses, inserts, is refused and evicts in some mix -- every row of every call equals
    the oracle's row of that proof (rows are per proof, so the oracle runs once per variant), and the counters stay consistent."""
    from tendermintx_amd.synth import Workload
    monkeypatch.setenv("TMX_SET_CACHE_SETS", "8")
    n, V, P = 128, 48, 20
    wl = Workload(0, n, V, 100, chain_id=b"celestia", seed=9200, signed_permille=900, n_sets=1)
    t = bytearray(wl.targets)
    for q in range(V):          # proof q: its own target set (one power word differs in lane q % 100)
        off = (q * n + q % 100) * 256 + 224
        t[off:off + 4] = (int.from_bytes(t[off:off + 4], "little") ^ (0x100 + q)).to_bytes(4, "little")
    targets = bytes(t)
    want, oreps = oracle.witness_batch(0, V, wl.proofs, targets, wl.trusteds, n, b"celestia", 100800, n_threads=8)
    rng = np.random.default_rng(77)
    with tmx.Context(n, b"celestia", max_batch=P) as ctx:
        for call in range(150):
            hot = rng.integers(0, V, 3)                       # a few hot sets + a random rest: hits and misses in every call
            pick = [int(hot[i % 3]) if rng.random() < 0.5 else int(rng.integers(0, V)) for i in range(P)]
            pr = b"".join(wl.proofs[2336 * q:2336 * (q + 1)] for q in pick)
            tg = b"".join(targets[256 * n * q:256 * n * (q + 1)] for q in pick)
            tr = b"".join(wl.trusteds[48 * n * q:48 * n * (q + 1)] for q in pick)
            e, reps = ctx.witness_batch(0, pr, tg, tr)
            assert np.array_equal(e, want[pick]), (call, pick)
            assert [r["all_ok"] for r in reps] == [oreps[q]["all_ok"] for q in pick]
        s = ctx.set_cache_stats()
        assert s["capacity"] == 8 and s["resident"] <= 8 and s["resident"] == s["inserted"] - s["evicted"] and s["evicted"] > 50
        assert s["served"] > 0 and s["computed"] > 0
