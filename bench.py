#!/usr/bin/env python3
"""Headline benchmark: skip-circuit witness generation at VALIDATOR_SET_SIZE_MAX=128 on N MI355X (one rank per GPU).

A "step" = one pass of the whole hot path (EdDSA kernels, k_proof, k_serialize) over one batch of synthetic skip proofs whose packed
input records are already resident in HBM; outputs (Goldilocks elements + reports) stay in HBM.  `python bench.py --gpus N` spawns its
own N ranks (torch.distributed.run, RCCL) when it was not started by a launcher; under a launcher it reads RANK / WORLD_SIZE.

  --scaling weak    (default) every rank processes --proofs proofs (256 = BASELINE configs[3]'s batch); no data-path collective
  --scaling strong  --proofs proofs in total, split over the ranks (configs[3] as written: 256 proofs sharded across 8);
                    --gather adds the all-gather of the witness rows (RCCL over xGMI) to the step and times it separately
  --mode c5         BASELINE configs[4]: ONE proof at N = 512, validator lanes sharded across the ranks, one all-gather of the
                    448-byte EdDSA lane records, replicated finish (tendermintx_amd/sharding.py)
  --workload survey8d (default) SURVEY 8(d)'s measured workload: 100 validators in 128 lanes, four distinct validator sets per batch,
                    Bernoulli(0.9) signing re-drawn until > 2/3, rounds {0,0,0,3}; `best_case` (one set, 128 of 128, everybody signs --
                    round 1's headline) is timed beside it and reported in the same line

Prints ONE JSON line on rank 0 (contract: task description "bench.py").  Beside the contract's fields:
  single_proof   BASELINE configs[2]: device-resident latency of one proof and its host-to-host time (SURVEY 8(d)'s metric definition)
  host_to_host   the batch through the host-buffer entry point (H2D + step + D2H), full rows and hint-only rows
  key_cache      the timed steps run with a WARM per-key table cache (the same validator sets step after step: what a light client does);
                 `cold` = the same step after tmx_key_cache_flush (every key new: decode, doubling chain, tables built inside the step)
  roofline       the step against HBM (algorithmic bytes / HIP-event time of the launch sequence), k_serialize alone, VALU issue
  cpu_baseline   oracle/c on this box's host cores: one thread, and a persistent pool on all cores (>= 16 proofs per thread) with its
                 scaling efficiency; OpenSSL's EVP_DigestVerify per signature as an independent datapoint
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
SIMDS, CLOCK_GHZ = 1024, 2.4   # 256 CUs x 4 SIMDs
# Issue cost of a wave64 VALU instruction per SIMD with the SIMD saturated, measured per opcode by tools/microbench/valu_isa.hip
# (profiles/r03_valu_isa.txt): 4.05-4.08 cycles for everything these kernels are made of (v_mad_u64_u32 / v_mad_i64_i32, every VOP3 and DPP
# form, v_mul_*, v_bfe_u32, carry pairs) -- only plain two-operand v_add_u32 / v_xor_b32 (and v_fma_f32) co-issue at ~2.3.  profiles/r03_isa_mix.json
# holds each kernel's static share of that fast class; the issue roof is reported between "all of them co-issued" and "none".
CYCLES_SLOW, CYCLES_FAST = 4.07, 2.3


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--proofs", type=int, default=256, help="proofs per GPU per step (weak) / in total (strong)")
    ap.add_argument("--n-max", type=int, default=None, help="VALIDATOR_SET_SIZE_MAX (default 128; 512 in --mode c5)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--gather", action="store_true", help="strong scaling: all-gather the witness rows on every rank inside the step")
    ap.add_argument("--mode", choices=("batch", "c5"), default="batch")
    ap.add_argument("--workload", choices=("survey8d", "one_set"), default="survey8d")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the timed region (profiling runs)")
    ap.add_argument("--other-scaling", action="store_true",
                    help="more than one GPU: after the timed region also run the scaling mode the command line did not ask for (strong: one batch sharded "
                         "through tmx_witness_batch_sharded_device without / with the row exchange, the sharded Level-2 rows).  Off by default: these are "
                         "collective calls that have only run through the stand-in RCCL (tests/test_world2_one_gpu.py), and the scaling record of the "
                         "driver's run must not depend on them")
    return ap.parse_args(argv)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script the way the driver does."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def gbs(nbytes, ms):
    return nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args)

    import torch  # first: libtmx must share PyTorch's HIP runtime (tendermintx_amd/_lib.py)
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (there is no CPU fallback)")
    # TMX_BENCH_SHARE_GPU=1 (development aid for 1-GPU boxes): every rank on cuda:0 with gloo as the control backend, so that the
    # multi-rank control flow (sharding of the proofs, barrier, max over ranks, one JSON line) can be exercised without a second GPU.
    # The numbers of such a run mean nothing (the ranks share one device) and the line says so.
    share_gpu = os.environ.get("TMX_BENCH_SHARE_GPU") == "1"
    # ... together with TMX_RCCL_LIB=<tests/fake_rccl>: libtmx's own communicator is made at world > 1 too (real RCCL refuses two ranks on one
    # device), so that the strong-scaling / row-exchange / --mode c5 code of this file runs before a real node runs it (tests/test_world2_one_gpu.py)
    share_gpu_comm = share_gpu and bool(os.environ.get("TMX_RCCL_LIB"))
    control_cpu = share_gpu  # the control plane's tensors live on the CPU (gloo)
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.mode == "c5"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
        fail_control = os.environ.get("TMX_BENCH_FAIL_RCCL_CONTROL") == "1"  # (test hook: the fallback below on a 1-GPU box)
        if share_gpu and not fail_control:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            # the control plane (barriers, max over ranks, the unique id) over RCCL; its first collective runs HERE so that a broken RCCL
            # shows as an exception in front of the timed region -- the control plane then falls back to gloo (no data-path collective is
            # part of the weak-scaled step; the line's rccl.control_backend says which one carried the barriers)
            try:
                if fail_control:
                    raise RuntimeError("TMX_BENCH_FAIL_RCCL_CONTROL=1")
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                probe = torch.ones(1, device=dev)
                dist.all_reduce(probe)
                torch.cuda.synchronize(dev)
                if int(probe.item()) != world:
                    raise RuntimeError(f"all_reduce over {world} ranks gave {probe.item()}")
            except Exception as e:  # noqa: BLE001
                print(f"warning: rank {rank}: RCCL control plane failed ({type(e).__name__}: {str(e)[:200]}); falling back to gloo", file=sys.stderr)
                if dist.is_initialized():
                    dist.destroy_process_group()
                dist.init_process_group("gloo", rank=rank, world_size=world)
                control_cpu = True
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using {world}", file=sys.stderr)

    import numpy as np
    from tendermintx_amd import KIND_SKIP, Context, sharding
    from tendermintx_amd.synth import bench_workload

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if control_cpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def dev_bytes(b):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)

    stream = torch.cuda.current_stream(dev)

    # ------------------------------------------------------------------------------------------------ configs[4]: one proof, lanes sharded
    if args.mode == "c5":
        n = args.n_max or 512
        wl = bench_workload("survey8d", n, 1, seed=0x544D58)      # identical on every rank (nb = N at C5)
        d_p, d_t, d_r = dev_bytes(wl.proofs), dev_bytes(wl.targets), dev_bytes(wl.trusteds)
        ctx = Context(n, b"celestia", 100800, device=local_rank, max_batch=1)

        if world == 1:   # a real one-rank RCCL communicator, so that the exchange runs through RCCL on a 1-GPU box too
            ctx.comm_create(sharding.unique_id(), 0, 1)
            comm_world = 1
        else:            # the 128-byte id travels through torch.distributed once; the data path is libtmx's own communicator
            comm_world = sharding.connect(ctx)[1]

        def step():
            return sharding.validator_sharded_skip(ctx, KIND_SKIP, d_p, d_t, d_r)   # tmx_witness_validator_sharded_device

        for _ in range(args.warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            elems, rep = step()
        barrier()
        ms_per_step = 1e3 * max_over_ranks(time.perf_counter() - t0) / args.steps
        gather_ms = None
        ok = int(rep.cpu().numpy()[32:36].view(np.uint32)[0])
        if rank == 0:
            out_bytes = ctx.elem_stride(KIND_SKIP) * 8
            emit({
                "metric": "skip-circuit witness-gen ms at VALIDATOR_SET_SIZE_MAX=512, one proof, validator-sharded", "value": round(ms_per_step, 5),
                "unit": "ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
                "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
                "config": {"workload": f"BASELINE configs[4]: SkipCircuit VALIDATOR_SET_SIZE_MAX={n}, ONE proof, {wl.nb} validators, lanes sharded over "
                                       f"{world} GPU(s), one all-gather of {n} x 448 B EdDSA lane records, finish replicated on every rank",
                           "n_max": n, "parallelism": f"validator-sharded x{world}"},
                "rccl": {"torch_world": world, "libtmx_comm_world": comm_world, "entry_point": "tmx_witness_validator_sharded_device"},
                "all_proofs_ok": bool(ok),
                **({"debug_shared_gpu": "TMX_BENCH_SHARE_GPU=1: every rank ran on cuda:0 (control-flow test, timings meaningless)"} if share_gpu else {}),
                "roofline": {"kernel": "step", "bound": "hbm", "achieved": round(gbs(out_bytes, ms_per_step), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(gbs(out_bytes, ms_per_step) / HBM_PEAK_GBS, 5), "traffic": None,
                             "note": "a single proof is latency-bound (dependent EdDSA chain), not bandwidth-bound: DESIGN.md section 6"}})
        barrier()
        ctx.close()
        if use_dist:
            dist.destroy_process_group()
        return 0

    # ------------------------------------------------------------------------------------------------ batch of independent proofs
    n = args.n_max or 128
    if args.scaling == "weak":
        P_total, lo, hi = args.proofs * world, rank * args.proofs, (rank + 1) * args.proofs
    else:
        P_total = args.proofs
        lo, hi = sharding.shard_range(P_total, rank, world)
    P = hi - lo
    seed = 0x544D58 + (rank if args.scaling == "weak" else 0)
    wl_all = bench_workload(args.workload, n, P_total if args.scaling == "strong" else P, seed=seed)
    # (strong: every rank holds the whole batch and a full-size row buffer; tmx_witness_batch_sharded_device fills this rank's rows in place)
    proofs, targets, trusteds = wl_all.proofs, wl_all.targets, wl_all.trusteds
    ctx = Context(n, b"celestia", 100800, device=local_rank, max_batch=max(P, 1))
    stride, count = ctx.elem_stride(KIND_SKIP), ctx.elem_count(KIND_SKIP)
    d_proofs, d_targets, d_trusteds = dev_bytes(proofs), dev_bytes(targets), dev_bytes(trusteds)
    rows_buf = P_total if args.scaling == "strong" else max(P, 1)
    d_out = torch.empty((rows_buf, stride), dtype=torch.int64, device=dev)
    d_rep = torch.zeros(rows_buf * 64, dtype=torch.uint8, device=dev)
    gather = args.gather and world > 1
    comm_world, comm_error = 1, None
    need_comm = args.scaling == "strong" or args.other_scaling   # (weak scaling alone has no data-path collective: no libtmx communicator at all)
    if use_dist and need_comm and (not share_gpu or share_gpu_comm):
        try:
            comm_world = sharding.connect(ctx)[1]
        except Exception as e:   # (weak scaling has no data-path collective: a communicator that cannot be made must not cost the scaling record)
            if args.scaling == "strong":
                raise
            comm_error = repr(e)[:200]
        # every rank must agree on whether the communicator exists (the strong-scaling extras below are collective calls)
        flag = torch.tensor([0 if comm_error else 1], device="cpu" if control_cpu else dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0 and comm_error is None:
            comm_error = "another rank could not create its communicator"
        if comm_error:
            comm_world = 1

    def run(c, k, bufs=None, n_proofs=None):
        dp, dt, dr = bufs or (d_proofs, d_targets, d_trusteds)
        for _ in range(k):
            c.witness_batch_device(KIND_SKIP, P if n_proofs is None else n_proofs, dp.data_ptr(), dt.data_ptr(), dr.data_ptr(), d_out.data_ptr(),
                                   d_rep.data_ptr(), stream.cuda_stream)

    def step():
        if args.scaling == "strong" and comm_world > 1:
            sharding.proof_sharded_batch(ctx, KIND_SKIP, P_total, d_proofs, d_targets, d_trusteds, d_out, d_rep, gather=gather, stream=stream.cuda_stream)
        else:
            run(ctx, 1)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    ms_per_step = 1e3 * elapsed / args.steps
    kms = ctx.kernel_ms_mean(min(args.steps, 128))  # HIP events recorded on the launch stream inside the timed region
    n_unique, used_tables = ctx.last_dedup()
    kc = ctx.key_cache_stats()

    gather_ms = None
    if gather:   # the same step without the exchange, for the difference
        for _ in range(3):
            sharding.proof_sharded_batch(ctx, KIND_SKIP, P_total, d_proofs, d_targets, d_trusteds, d_out, d_rep, gather=False, stream=stream.cuda_stream)
        barrier()
        t0 = time.perf_counter()
        for _ in range(10):
            sharding.proof_sharded_batch(ctx, KIND_SKIP, P_total, d_proofs, d_targets, d_trusteds, d_out, d_rep, gather=False, stream=stream.cuda_stream)
        barrier()
        gather_ms = ms_per_step - 1e3 * max_over_ranks(time.perf_counter() - t0) / 10

    # every proof of this rank must have verified (synthetic inputs are well-formed)
    row0 = lo if (args.scaling == "strong" and comm_world > 1) else 0   # (strong: this rank's rows sit at their place in the full-size buffer)
    rep = d_rep.cpu().numpy().reshape(-1, 64)[row0:row0 + P]
    all_ok = int(rep[:, 32:36].copy().view(np.uint32).sum())
    ok_flag = torch.tensor([1 if all_ok == P else 0], device="cpu" if control_cpu else dev)
    if world > 1:
        dist.all_reduce(ok_flag, op=dist.ReduceOp.MIN)

    if rank == 0:
        lanes = P * n
        in_bytes = P * (2336 + n * (256 + 48))
        out_bytes = P * stride * 8
        alg_bytes = in_bytes + out_bytes      # 33 504 B per lane + 42.6 KB per proof (1.109 GB at 256 x 128), counted exactly
        step_ms_events = kms["k_eddsa"] + kms["k_serialize"]   # ev0 -> ev3 of the launch sequence on the caller's stream
        result = {
            "metric": "skip-circuit witness-gen ms at VALIDATOR_SET_SIZE_MAX=128", "value": round(ms_per_step / P_total, 6), "unit": "ms",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": False,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": f"SkipCircuit VALIDATOR_SET_SIZE_MAX={n}, batch of {P_total} proofs over {world} GPU(s) "
                                   f"({'BASELINE configs[3] batch per GPU, weak-scaled' if args.scaling == 'weak' else 'BASELINE configs[3]: one batch sharded'}), "
                                   f"{wl_all.describe}; inputs resident in HBM, witness rows stay in HBM",
                       "n_max": n, "proofs_total": P_total, "proofs_per_gpu": P, "workload_name": args.workload,
                       "parallelism": f"proof-sharded x{world}, " + ("all-gather of the witness rows inside the step" if gather else "no data-path collective")},
            "value_note": "SURVEY 8(d)'s primary metric (host-resident in -> host-resident out) is `value_host_to_host_ms` (per proof of the batch) and "
                          "`value_single_proof_host_to_host_ms` (BASELINE configs[2]), both through the typed value of the hint (`typed_value`); "
                          "`value` = ms_per_step / proofs_total is the device-resident, batch-amortised kernel-side figure with full element rows "
                          "written in HBM (the driver-facing number, unchanged in definition since round 1)",
            "throughput": {"proofs_per_s": round(P_total / (ms_per_step * 1e-3), 1), "lanes_per_s": round(P_total * n / (ms_per_step * 1e-3), 1)},
            "kernels_ms": dict({k: round(v, 4) for k, v in kms.items()}, step_events=round(step_ms_events, 4)),
            "all_proofs_ok": bool(int(ok_flag.item())),
            **({"debug_shared_gpu": "TMX_BENCH_SHARE_GPU=1: every rank ran on cuda:0 (control-flow test, timings meaningless)"} if share_gpu else {}),
            "dedup": {"lanes": lanes, "distinct_keys": n_unique, "per_key_tables": used_tables},
            "key_cache": {"state_of_the_timed_steps": "warm" if kc["last_new_keys"] == 0 and kc["last_hit_lanes"] == lanes else "cold/mixed",
                          "last_step": {k: kc[k] for k in ("last_new_keys", "last_hit_keys", "last_hit_lanes", "last_built_keys")},
                          "resident_keys": kc["resident_keys"], "capacity_keys": kc["capacity_keys"], "bytes_per_key": kc["bytes_per_key"],
                          "note": "the warm-up steps made the batch's validator sets resident; a light client re-verifies the same slowly changing set "
                                  "from call to call (reference bin/tendermintx.rs:171).  `cold` below = the same step with an empty cache"},
        }
        try:  # what the step does NOT recompute (round 5): lanes that did not sign, validator sets seen before
            flags = np.frombuffer(wl_all.targets, dtype=np.uint8).reshape(-1, 256)[:, 223] & 1
            result["memoized"] = {
                "lanes_that_did_not_sign": int((flags == 0).sum()), "lanes": int(flags.size),
                "lanes_note": "absent / nil votes and the lanes behind the validator count are evaluated on plonky2x's dummy triple (verify.rs:248-259): one "
                              "record computed once per context, copied into those lanes; the EdDSA kernels run over the dense list of the others",
                "validator_set_cache": ctx.set_cache_stats(),
                "validator_set_cache_note": "marshalled validators, leaf hashes and tree nodes of a (target | trusted) set depend on the set alone: "
                                            "served from the context's content-addressed cache (every key byte compared) once seen; totals since creation"}
        except Exception as e:
            result["memoized"] = {"error": repr(e)}
        if gather_ms is not None:
            result["gather_rows"] = {"ms": round(gather_ms, 4), "bytes_per_rank_out": P_total * stride * 8,
                                     "note": "the step with the grouped RCCL exchange of the row slices (tmx_witness_batch_sharded_device, gather = 1) minus the "
                                             "same step without it (10 repetitions): every rank ends with all rows"}
        roofline = {"kernel": "step", "kernel_note": "the whole launch sequence of one batch (EdDSA kernels, k_proof, k_serialize on four streams), "
                    "timed by HIP events on the caller's stream", "bound": "hbm", "achieved": round(gbs(alg_bytes, step_ms_events), 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(gbs(alg_bytes, step_ms_events) / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes": alg_bytes,
                    "host_clock": {"achieved": round(gbs(alg_bytes, ms_per_step), 1), "frac": round(gbs(alg_bytes, ms_per_step) / HBM_PEAK_GBS, 4)},
                    "note": "the step is bounded by VALU issue and by the dependent EdDSA chain (valu_issue), k_serialize by HBM writes; "
                            "nothing on this path is a dense contraction (no MFMA)"}
        result["roofline"] = roofline

        result["rccl"] = {"torch_world": world, "libtmx_comm_world": comm_world, **({"comm_error": comm_error} if comm_error else {}),
                          "control_backend": (dist.get_backend() if use_dist and dist.is_initialized() else None),
                          "note": "the data-path exchange (strong scaling with --gather, --mode c5) is RCCL inside libtmx (tmx_comm_create); torch.distributed "
                                  "carries the unique id, the barriers and the max over ranks"}
        if not args.no_extras and world == 1:
            extras(args, result, roofline, ctx, run, wl_all, n, P, stride, count, dev, stream, d_out, d_rep, dev_bytes, kms, ms_per_step, alg_bytes)
    # ---- more than one GPU: the OTHER scaling of the same record (a SCALE run of the default command then carries both the weak-scaled value
    # and BASELINE configs[3] as written: 256 proofs sharded over the ranks, without and with the row exchange)
    if world > 1 and args.other_scaling and not args.no_extras and (not share_gpu or share_gpu_comm) and comm_error is None:
        other = both_scalings(args, ctx if args.scaling == "strong" else None, n, world, rank, local_rank, dev, stream, dev_bytes, barrier, max_over_ranks)
        if rank == 0:
            result["other_scaling"] = other
    if rank == 0:
        emit(result)
    barrier()
    if use_dist:
        dist.destroy_process_group()
    ctx.close()
    return 0


LINE_CAP = 6144   # bytes; the driver keeps the last 8 KB of stdout and parses the LAST line (round 5's 20.5-KB line came back `parsed: null`)


def pick(d, *keys):
    """Sub-dict of the keys that exist (and are not None / prose)."""
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(full):
    """The ONE driver-facing JSON line: the contract's fields + roofline + cpu_baseline + the handful of secondary figures VERDICT r5 #1 lists,
    numbers only (no prose), built from the full record.  Everything else lives in bench_extras.json and on an EARLIER stdout line."""
    c = pick(full, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling")
    c["vs_baseline"] = full.get("vs_baseline")   # null: BASELINE.md holds no published number for this metric
    c.update(pick(full, "dtype", "data"))
    cfg = full.get("config", {})
    c["config"] = pick(cfg, "n_max", "proofs_total", "proofs_per_gpu", "workload_name", "parallelism")
    c["config"]["workload"] = str(cfg.get("workload", ""))[:200]
    c["all_proofs_ok"] = full.get("all_proofs_ok")
    if "debug_shared_gpu" in full:
        c["debug_shared_gpu"] = True
    r = full.get("roofline", {})
    rr = pick(r, "kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes")
    rr.setdefault("traffic", None)
    if "host_clock" in r:
        rr["host_clock"] = pick(r["host_clock"], "achieved", "frac")
    if "traffic_detail" in r:
        rr["traffic_split"] = pick(r["traffic_detail"], "fetch_bytes", "write_bytes")
    if "per_kernel" in r:   # top 6 by time alone
        rr["per_kernel"] = {k.split("<")[0] + (k[k.index("<"):] if "<" in k and k.startswith("k_ed") else ""):
                            {"insts": v.get("valu_insts"), "alone_us": v.get("alone_us"), "issue_frac": v.get("issue_frac"), "hbm_frac": v.get("hbm_frac")}
                            for k, v in list(r["per_kernel"].items())[:6]}
    if "valu_issue" in r:
        rr["valu_issue"] = {"step": r["valu_issue"].get("step")}
    if "k_serialize" in r:
        rr["k_serialize"] = pick(r["k_serialize"], "ms_alone", "achieved", "frac", "traffic", "algorithmic_bytes", "split_alone_us")
    c["roofline"] = rr
    if "kernels_ms" in full:
        c["kernels_ms"] = full["kernels_ms"]
    if "cpu_baseline" in full:
        b = full["cpu_baseline"]
        cb = pick(b, "value", "unit", "cores", "kind", "compute_only_ms")
        cb["sample"] = str(b.get("sample", ""))[:160]
        if "all_cores" in b:
            cb["all_cores"] = pick(b["all_cores"], "value", "cores", "scaling_efficiency")
        if "openssl_evp_digestverify" in b:
            cb["openssl_us_per_verify"] = b["openssl_evp_digestverify"].get("us_per_verify")
        c["cpu_baseline"] = cb
    if "parity_vs_oracle" in full:
        c["parity_vs_oracle"] = full["parity_vs_oracle"]
    kc = full.get("key_cache", {})
    c["key_cache"] = {"state": kc.get("state_of_the_timed_steps"), "resident_keys": kc.get("resident_keys"),
                      "cold_ms": kc.get("cold", {}).get("ms_per_step"), "warm_ms": kc.get("warm", {}).get("ms_per_step"),
                      "churn_ms_by_new_keys": kc.get("churn", {}).get("ms_per_step_by_new_keys")}
    if "memoized" in full and "validator_set_cache" in full["memoized"]:
        c["memoized"] = {"lanes_that_did_not_sign": full["memoized"].get("lanes_that_did_not_sign"), "set_cache": full["memoized"]["validator_set_cache"]}
    if "batch_sizes_ms" in full:
        c["batch_sizes_ms"] = full["batch_sizes_ms"]
    if "batch_sizes_frac" in full:
        c["batch_sizes_frac"] = full["batch_sizes_frac"]
    if "single_proof" in full:
        c["single_proof"] = pick(full["single_proof"], "device_ms", "device_ms_cold", "host_to_host_ms", "host_to_host_typed_value_ms")
    if "typed_value" in full and "batch" in full["typed_value"]:
        t = full["typed_value"]
        c["typed_value"] = {"bytes_per_proof": t.get("bytes_per_proof", {}).get("value_hint"), "single_proof_host_to_host_ms": t.get("single_proof", {}).get("host_to_host_ms"),
                            "batch_host_to_host_ms": t["batch"].get("host_to_host_ms_per_step"), "batch_device_ms": t["batch"].get("device_resident_ms_per_step"),
                            "bit_exact_vs_oracle": t.get("bit_exact_vs_oracle")}
    c.update(pick(full, "value_host_to_host_ms", "value_single_proof_host_to_host_ms"))
    for k in ("best_case", "survey8d"):
        if k in full:
            c[k] = pick(full[k], "ms_per_step", "ms_per_step_cold", "all_proofs_ok")
    l2 = full.get("level2_trace_rows", {})
    if "roofline" in l2:
        c["level2"] = {"ms_per_batch": l2.get("ms_per_batch"), **pick(l2["roofline"], "frac", "achieved", "algorithmic_bytes", "traffic"),
                       "traffic_over_rows": pick(l2["roofline"].get("traffic_over_rows", {}), "raw", "fetch_x2"),
                       "violations": l2.get("constraint_check", {}).get("violations")}
    elif "error" in l2:
        c["level2"] = {"error": str(l2["error"])[:120]}
    cp = full.get("commit_pipeline", {}).get("sections")
    if cp:
        c["commit_pipeline"] = {k: {"ms_total": v.get("ms_total"), "lde_frac_passes": v.get("lde_stage", {}).get("frac_passes"),
                                    "lde_frac_1r1w": v.get("lde_stage", {}).get("frac_1r1w"), "gperm_per_s": v.get("merkle_stage", {}).get("gperm_per_s")}
                                for k, v in cp.items() if k in ("sha512", "ladders")}
    if "rccl" in full:
        c["rccl"] = pick(full["rccl"], "torch_world", "libtmx_comm_world", "backend", "control_backend", "comm_error")
    if "gather_rows" in full:
        c["gather_rows"] = pick(full["gather_rows"], "ms", "bytes_per_rank_out")
    if "other_scaling" in full:
        c["other_scaling"] = {k: (pick(v, "ms_per_step", "value", "proofs_total", "proofs_per_gpu", "libtmx_comm_world", "error") if isinstance(v, dict) else None)
                              for k, v in full["other_scaling"].items() if k != "note"}
    c["extras"] = "bench_extras.json"
    # never over the cap: shed the secondary objects, least important first
    for k in ("commit_pipeline", "best_case", "survey8d", "memoized", "kernels_ms", "other_scaling", "level2", "typed_value", "batch_sizes_frac", "batch_sizes_ms", "single_proof"):
        if len(json.dumps(c)) <= LINE_CAP:
            break
        c.pop(k, None)
    return c


def emit(full):
    """Full record -> bench_extras.json (+ gpurun_out/) and an EARLIER stdout line; the compact record is the LAST line."""
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, "bench_extras.json"), "w") as f:
                    json.dump(full, f, indent=1)
        except OSError:
            pass
    print("# bench_extras " + json.dumps(full), flush=True)   # (prefixed: no parser can take it for the record line)
    line = json.dumps(compact_line(full))
    assert len(line) <= LINE_CAP, len(line)
    print(line, flush=True)


def both_scalings(args, strong_ctx, n, world, rank, local_rank, dev, stream, dev_bytes, barrier, max_over_ranks):
    """Run by EVERY rank after the timed region: the scaling mode the command line did not ask for, 20 steps after 5 warm-up.
    strong = --proofs proofs in total over the ranks through tmx_witness_batch_sharded_device (gather off / on); weak = --proofs per rank."""
    import torch
    from tendermintx_amd import KIND_SKIP, Context, sharding
    from tendermintx_amd.synth import bench_workload
    out = {}

    def timed(fn, k=20, w=5):
        for _ in range(w):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        barrier()
        return 1e3 * max_over_ranks(time.perf_counter() - t0) / k

    if args.scaling == "weak":
        Pt = args.proofs
        wl = bench_workload(args.workload, n, Pt, seed=0x544D58)
        lo, hi = sharding.shard_range(Pt, rank, world)
        c = Context(n, b"celestia", 100800, device=local_rank, max_batch=max(hi - lo, 1))
        cw = sharding.connect(c)[1]
        d = [dev_bytes(b) for b in (wl.proofs, wl.targets, wl.trusteds)]
        o = torch.empty((Pt, c.elem_stride(KIND_SKIP)), dtype=torch.int64, device=dev)
        r = torch.zeros(Pt * 64, dtype=torch.uint8, device=dev)
        for g in (False, True):
            ms = timed(lambda: sharding.proof_sharded_batch(c, KIND_SKIP, Pt, d[0], d[1], d[2], o, r, gather=g, stream=stream.cuda_stream))
            out["strong" + ("_with_row_exchange" if g else "")] = {"ms_per_step": round(ms, 4), "value": round(ms / Pt, 6), "proofs_total": Pt,
                                                                  "proofs_per_gpu": hi - lo, "libtmx_comm_world": cw}
        # Level-2 trace rows of the same sharded batch (judge row 8(e)-T): every rank writes the rows of its proofs, then one exchange of the
        # 41-MB blocks (tmx_trace_rows_sharded_device) -- the one payload of this path big enough for xGMI to matter
        try:
            te = c.trace_elem_count(KIND_SKIP)
            tr = torch.empty((Pt, te), dtype=torch.int64, device=dev)
            sharding.proof_sharded_batch(c, KIND_SKIP, Pt, d[0], d[1], d[2], o, r, gather=False, stream=stream.cuda_stream)
            for g in (False, True):
                ms = timed(lambda: c.trace_rows_sharded_device(KIND_SKIP, Pt, d[1].data_ptr(), d[2].data_ptr(), tr.data_ptr(), 63, gather=g, stream=stream.cuda_stream), k=5, w=2)
                out["level2_trace_rows" + ("_with_exchange" if g else "")] = {"ms_per_step": round(ms, 4), "bytes_total": Pt * te * 8, "proofs_per_gpu": hi - lo,
                                                                              "entry_point": "tmx_trace_rows_sharded_device"}
            del tr
        except Exception as e:  # (never lose the scaling record to the secondary line)
            out["level2_trace_rows"] = {"error": str(e)[:200]}
        out["note"] = ("BASELINE configs[3] as written: ONE batch of --proofs proofs sharded over the ranks (tmx_witness_batch_sharded_device), without a "
                       "data-path collective and with the grouped RCCL exchange that leaves every row on every rank; value = ms per proof")
        c.close()
    else:
        Pw = args.proofs
        wl = bench_workload(args.workload, n, Pw, seed=0x544D58 + rank)
        c = Context(n, b"celestia", 100800, device=local_rank, max_batch=Pw)
        d = [dev_bytes(b) for b in (wl.proofs, wl.targets, wl.trusteds)]
        o = torch.empty((Pw, c.elem_stride(KIND_SKIP)), dtype=torch.int64, device=dev)
        r = torch.zeros(Pw * 64, dtype=torch.uint8, device=dev)
        ms = timed(lambda: c.witness_batch_device(KIND_SKIP, Pw, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), o.data_ptr(), r.data_ptr(), stream.cuda_stream))
        out["weak"] = {"ms_per_step": round(ms, 4), "value": round(ms / (Pw * world), 6), "proofs_total": Pw * world, "proofs_per_gpu": Pw}
        c.close()
    return out


def extras(args, result, roofline, ctx, run, wl, n, P, stride, count, dev, stream, d_out, d_rep, dev_bytes, kms, ms_per_step, alg_bytes):
    """Everything beside the timed region (rank 0, one GPU): secondary workload, single proof, host-to-host, kernel-alone figures,
    PMC-derived figures from profiles/, CPU baseline."""
    import numpy as np
    import torch
    from tendermintx_amd import KIND_SKIP, Context
    from tendermintx_amd.synth import bench_workload

    def timed(c, k, bufs=None, n_proofs=None):
        torch.cuda.synchronize(dev)
        a = time.perf_counter()
        run(c, k, bufs, n_proofs)
        stream.synchronize()   # the call is complete when the caller's stream is (a cold single proof leaves the table build of its keys
        b = time.perf_counter()  # running on the library's side stream: work for the NEXT call, not part of this one's latency)
        torch.cuda.synchronize(dev)
        return 1e3 * (b - a) / k

    # ---- cold: the same step with an empty key cache (every key new: decoded, its table built inside the step), 12 flush + step pairs
    def cold_ms(bufs=None, n_proofs=None, reps=12):
        xs = []
        for _ in range(reps):
            ctx.key_cache_flush()
            xs.append(timed(ctx, 1, bufs, n_proofs))
        return median(xs)
    result["key_cache"]["cold"] = {"ms_per_step": round(cold_ms(), 4), "note": "tmx_key_cache_flush + one step, host clock around enqueue + synchronize, median of 12"}
    timed(ctx, 3)
    result["key_cache"]["warm"] = {"ms_per_step": round(timed(ctx, 30), 4), "note": "30 steps, host clock (the timed region above is the driver-facing figure)"}

    # ---- the other workload (best case <-> SURVEY 8(d)) through the same context
    other_name = "one_set" if args.workload == "survey8d" else "survey8d"
    wo = bench_workload(other_name, n, P, seed=0x544D58)
    bo = tuple(dev_bytes(b) for b in (wo.proofs, wo.targets, wo.trusteds))
    ms_other_cold = cold_ms(bo)
    timed(ctx, 10, bo)
    ms_other = timed(ctx, 30, bo)
    uo, to = ctx.last_dedup()
    ok_o = int(d_rep.cpu().numpy().reshape(-1, 64)[:P, 32:36].copy().view(np.uint32).sum()) == P
    result["best_case" if other_name == "one_set" else "survey8d"] = {
        "workload": wo.describe, "ms_per_step": round(ms_other, 4), "ms_per_step_cold": round(ms_other_cold, 4), "value": round(ms_other / P, 6),
        "distinct_keys": uo, "per_key_tables": to,
        "all_proofs_ok": ok_o, "note": "same context, 30 steps after 10 warm-up, host clock around the enqueue + synchronize"}

    # ---- other batch sizes through the same entry point (warm key cache, rows resident, host clock around 20 pipelined steps): the latency regime
    # (32 ... 128 proofs) and the throughput regime (512 / 1024 proofs: chain + tail)
    try:
        sizes = {}
        for Pb in (32, 64, 128, 512, 1024):
            wb = bench_workload(args.workload, n, Pb, seed=0x544D58)
            cb = ctx if Pb <= P else Context(n, b"celestia", 100800, device=dev.index, max_batch=Pb)
            bb = tuple(dev_bytes(b) for b in (wb.proofs, wb.targets, wb.trusteds))
            ob = d_out if Pb <= P else torch.empty((Pb, stride), dtype=torch.int64, device=dev)
            rb = d_rep if Pb <= P else torch.zeros(Pb * 64, dtype=torch.uint8, device=dev)

            def go(k):
                for _ in range(k):
                    cb.witness_batch_device(KIND_SKIP, Pb, bb[0].data_ptr(), bb[1].data_ptr(), bb[2].data_ptr(), ob.data_ptr(), rb.data_ptr(), stream.cuda_stream)
                torch.cuda.synchronize(dev)
            go(6)
            a = time.perf_counter()
            go(20)
            ms = 1e3 * (time.perf_counter() - a) / 20
            ok_b = int(rb.cpu().numpy().reshape(-1, 64)[:Pb, 32:36].copy().view(np.uint32).sum()) == Pb
            sizes[str(Pb)] = round(ms, 4) if ok_b else None
            if cb is not ctx:
                cb.close()
            del bb
        result["batch_sizes_ms"] = sizes
        # the same sizes as fractions of the HBM roof (SURVEY 8d bytes of that many proofs / host clock / 8 TB/s): the throughput regime's figure
        result["batch_sizes_frac"] = {k: (round(gbs(alg_bytes * int(k) // P, v) / HBM_PEAK_GBS, 4) if v else None) for k, v in sizes.items()}
        run(ctx, 3)   # (the timed batch resident again)
    except Exception as e:
        result["batch_sizes_ms"] = {"error": repr(e)[:200]}

    # ---- BASELINE configs[2]: ONE proof.  Device-resident latency (host clock around one call) and host-to-host (host buffers in,
    # witness row in page-locked host memory out: SURVEY 8(d)'s metric definition)
    lat_cold = cold_ms(None, 1, reps=20)
    run(ctx, 3, None, 1)
    lat = []
    for _ in range(30):
        lat.append(timed(ctx, 1, None, 1))
    pinned = torch.empty(P * stride, dtype=torch.int64, pin_memory=True)
    host_out = pinned.numpy().view(np.uint64)
    p1 = (wl.proofs[:2336], wl.targets[:n * 256], wl.trusteds[:n * 48])
    hh1, hh1_hint = [], []
    for _ in range(3):
        ctx.witness_batch(KIND_SKIP, *p1, out=host_out)
    for _ in range(20):
        a = time.perf_counter()
        ctx.witness_batch(KIND_SKIP, *p1, out=host_out)
        hh1.append(1e3 * (time.perf_counter() - a))
    hint = getattr(ctx, "witness_batch_hint", None)
    if hint is not None:
        for _ in range(3):
            hint(KIND_SKIP, *p1, out=host_out)
        for _ in range(20):
            a = time.perf_counter()
            hint(KIND_SKIP, *p1, out=host_out)
            hh1_hint.append(1e3 * (time.perf_counter() - a))
    result["single_proof"] = {"device_ms": round(median(lat), 4), "device_ms_cold": round(lat_cold, 4), "host_to_host_ms": round(median(hh1), 4),
                              "host_to_host_hint_only_ms": round(median(hh1_hint), 4) if hh1_hint else None,
                              "row_bytes": stride * 8, "workload": "proof 0 of the batch above (BASELINE configs[2]); device_ms / host_to_host with the "
                              "proof's validator set resident in the key cache (warm), device_ms_cold after tmx_key_cache_flush"}
    result["latency_single_proof_ms"] = result["single_proof"]["device_ms"]
    # SURVEY 8(d)'s own metric definition beside `value` (which is the device-resident, batch-amortised kernel-side figure)
    result["value_single_proof_ms"] = result["single_proof"]["device_ms"]
    result["value_single_proof_host_to_host_ms"] = result["single_proof"]["host_to_host_ms"]

    # ---- the batch, host to host (PCIe-bound; never `value`)
    hh = []
    for _ in range(3):
        a = time.perf_counter()
        ctx.witness_batch(KIND_SKIP, wl.proofs, wl.targets, wl.trusteds, out=host_out)
        hh.append(1e3 * (time.perf_counter() - a))
    in_bytes = P * (2336 + n * (256 + 48))
    hb = in_bytes + P * stride * 8 + P * 64
    h2h = {"ms_per_step": round(min(hh), 3), "ms_per_proof": round(min(hh) / P, 5), "bytes_over_pcie": hb,
           "effective_gbs": round(hb / (min(hh) * 1e-3) / 1e9, 1), "note": "pageable inputs, page-locked output rows; PCIe-bound"}
    if hint is not None:
        hq = []
        for _ in range(3):
            a = time.perf_counter()
            hint(KIND_SKIP, wl.proofs, wl.targets, wl.trusteds, out=host_out)
            hq.append(1e3 * (time.perf_counter() - a))
        hbh = in_bytes + P * ctx.hint_elem_count(KIND_SKIP) * 8 + P * 64
        h2h["hint_only"] = {"ms_per_step": round(min(hq), 3), "ms_per_proof": round(min(hq) / P, 5), "bytes_over_pcie": hbh,
                            "note": "only the hint section H of every row (what SkipOffchainInputs::hint writes, skip.rs:85-100) leaves the device"}
    result["host_to_host"] = h2h
    del pinned, host_out

    # ---- SURVEY 8(d)'s PRIMARY metric (host-resident records in -> host-resident witness out) through the TYPED VALUE of the hint
    # (tmx_inputs_value_batch: the reference's SkipInputs<F>, circuits/input/mod.rs:60-74, field by field -- what SkipOffchainInputs::hint
    # holds before write_value expands it, skip.rs:85-100): 38 KB per proof instead of a 4.3 MB element row, page-locked buffers both ways,
    # no element row produced on the device at all.  `value_host_to_host_ms` / `value_single_proof_host_to_host_ms` are these figures.
    from tendermintx_amd import _lib
    layh, laya = ctx.value_layout(KIND_SKIP, _lib.SEC_HINT), ctx.value_layout(KIND_SKIP, _lib.SEC_ALL)
    pin_out = ctx.host_alloc(P * laya.bytes)
    pin_in = [ctx.host_alloc(len(b)) for b in (wl.proofs, wl.targets, wl.trusteds)]
    for a, b in zip(pin_in, (wl.proofs, wl.targets, wl.trusteds)):
        a[:] = np.frombuffer(b, dtype=np.uint8)

    def t_value(k, sections, reps):
        a, b, c = pin_in[0][:2336 * k], pin_in[1][:256 * n * k], pin_in[2][:48 * n * k]
        for _ in range(3):
            ctx.inputs_value_batch(KIND_SKIP, a, b, c, sections, out=pin_out)
        xs = []
        for _ in range(reps):
            t0 = time.perf_counter()
            ctx.inputs_value_batch(KIND_SKIP, a, b, c, sections, out=pin_out)
            xs.append(1e3 * (time.perf_counter() - t0))
        return median(xs)
    tv1, tv1a = t_value(1, _lib.SEC_HINT, 40), t_value(1, _lib.SEC_ALL, 40)
    tvb, tvba = t_value(P, _lib.SEC_HINT, 10), t_value(P, _lib.SEC_ALL, 10)
    # the device-resident form of the same call (records and values stay in HBM): the Level-1 kernels + k_pack_value, no serializer
    d_val = torch.empty(P * laya.bytes, dtype=torch.uint8, device=dev)
    dp, dt, dr = (dev_bytes(b) for b in (wl.proofs, wl.targets, wl.trusteds))
    for _ in range(5):
        ctx.inputs_value_batch_device(KIND_SKIP, P, dp.data_ptr(), dt.data_ptr(), dr.data_ptr(), d_val.data_ptr(), _lib.SEC_HINT, stream=stream.cuda_stream)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.inputs_value_batch_device(KIND_SKIP, P, dp.data_ptr(), dt.data_ptr(), dr.data_ptr(), d_val.data_ptr(), _lib.SEC_HINT, stream=stream.cuda_stream)
    stream.synchronize()
    tv_dev = 1e3 * (time.perf_counter() - t0) / 20
    # the value the timed path produces IS the oracle's (one proof, byte for byte; the GPU suite checks every size): checked here so that the
    # host-to-host figures cannot come from a value nobody looked at
    sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
    import oracle_c as oc  # checker only, outside every timed region
    got1, _ = ctx.inputs_value_batch(KIND_SKIP, wl.proofs[:2336], wl.targets[:256 * n], wl.trusteds[:48 * n], _lib.SEC_ALL)
    want1, _ = oc.witness_value(KIND_SKIP, wl.proofs[:2336], wl.targets[:256 * n], wl.trusteds[:48 * n], b"celestia", 100800, True)
    in_bytes = P * (2336 + n * (256 + 48))
    result["typed_value"] = {
        "what": "tmx_inputs_value_batch: SkipInputs<F> of the reference (circuits/input/mod.rs:60-74) packed field by field + tmx_report; with "
                "`derived` also the packed Level-1 derived values (section D); page-locked host buffers in and out; host clock around the blocking call",
        "bytes_per_proof": {"value_hint": int(layh.bytes), "value_with_derived": int(laya.bytes), "element_row_u64": stride * 8},
        "single_proof": {"host_to_host_ms": round(tv1, 4), "host_to_host_with_derived_ms": round(tv1a, 4)},
        "batch": {"proofs": P, "host_to_host_ms_per_step": round(tvb, 4), "host_to_host_ms_per_proof": round(tvb / P, 6),
                  "with_derived_ms_per_step": round(tvba, 4), "bytes_over_pcie": in_bytes + P * int(layh.bytes),
                  "device_resident_ms_per_step": round(tv_dev, 4)},
        "bit_exact_vs_oracle": bool(np.array_equal(got1[0], want1)),
        "rows_path_for_comparison": {"single_proof_host_to_host_ms": result["single_proof"]["host_to_host_ms"], "batch_host_to_host_ms_per_step": h2h["ms_per_step"]}}
    result["value_host_to_host_ms"] = round(tvb / P, 6)
    result["value_single_proof_host_to_host_ms"] = round(tv1, 4)
    result["single_proof"]["host_to_host_typed_value_ms"] = round(tv1, 4)
    result["host_to_host"]["typed_value_ms_per_step"] = round(tvb, 4)
    for a in [pin_out] + pin_in:
        ctx.host_free(a)
    del d_val

    # ---- k_serialize on its own (the step spreads it over overlapped launches): a second context with the split disabled
    os.environ["TMX_SER_SPLIT"] = "0"
    ctx1 = Context(n, b"celestia", 100800, device=dev.index, max_batch=P)
    del os.environ["TMX_SER_SPLIT"]
    run(ctx1, 30)  # (the first launches after a context creation run at ramping clocks)
    torch.cuda.synchronize(dev)
    k_s = ctx1.kernel_ms_mean(20)["k_serialize"]
    ctx1.close()
    ser_bytes = P * stride * 8 + P * (n * (448 + 2 * 112 + 256 + 48) + 1920 + 2336)
    roofline["k_serialize"] = {"bound": "hbm", "ms_alone": round(k_s, 4), "achieved": round(gbs(ser_bytes, k_s), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(gbs(ser_bytes, k_s) / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes": ser_bytes,
                               "note": "one unsplit launch over the whole batch, HIP events around it (TMX_SER_SPLIT=0)"}
    # ---- transparency: the same step with the per-key tables switched off (every lane does its own 252 doublings for h*A)
    os.environ["TMX_DEDUP"] = "0"
    ctx0 = Context(n, b"celestia", 100800, device=dev.index, max_batch=P)
    del os.environ["TMX_DEDUP"]
    timed(ctx0, 3)
    result["dedup"]["without_key_tables"] = {"ms_per_step": round(timed(ctx0, 10), 4), "k_eddsa_ms": round(ctx0.kernel_ms_mean(10)["k_eddsa"], 4)}
    ctx0.close()

    # ---- HBM traffic of the step measured IN THIS RUN when rocprofv3 is on PATH: two --pmc passes (FETCH_SIZE / WRITE_SIZE never fit one) over
    # tools/profile_step.py, the same workload; the committed profile is replayed below only if this fails
    measured_traffic = measure_traffic(n, P, args.workload)
    if measured_traffic:
        roofline["traffic"] = measured_traffic["bytes_per_step"]
        roofline["traffic_source"] = measured_traffic["source"]
        roofline["traffic_detail"] = measured_traffic
        # per kernel: wave-level VALU instructions and time ALONE (the counter pass serializes the kernels) -> the fraction of the chip's issue
        # slots the kernel fills on its own; HBM bytes beside it.  The chain kernels (hash, walk, finish) and k_proof are latency-bound
        # launches of 512 - 4096 waves: their roof is this one, not HBM.
        pkr = {}
        for k, v in measured_traffic["per_kernel"].items():
            if "SQ_INSTS_VALU" not in v or not v.get("alone_us"):
                continue
            pkr[k] = {"valu_insts": int(v["SQ_INSTS_VALU"]), "alone_us": v["alone_us"],
                      "issue_frac": round(v["SQ_INSTS_VALU"] * CYCLES_SLOW / (SIMDS * CLOCK_GHZ * 1e3 * v["alone_us"]), 3),
                      "hbm_bytes": int(v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)),
                      "hbm_frac": round((v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) / (v["alone_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 3)}
        roofline["per_kernel"] = dict(sorted(pkr.items(), key=lambda kv: -kv[1]["alone_us"]))
        roofline["per_kernel_note"] = ("issue_frac = wave-level VALU instructions x 4.07 cycles / (1024 SIMDs x 2.4 GHz x the kernel's time alone); alone_us = "
                                       "its launches of one batch summed, kernels serialized by the counter pass (inside a step they overlap: kernels_ms)")
        ser = {k: v for k, v in measured_traffic["per_kernel"].items() if k.startswith("k_serialize")}
        if ser:   # the serializer launches of the SPLIT configuration the step runs (five k_serialize + two k_serialize_few + the seams)
            roofline["k_serialize"]["traffic"] = int(sum(v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0) for v in ser.values()))
            roofline["k_serialize"]["traffic_source"] = measured_traffic["source"] + " (the split launches of the step; `ms_alone` is the unsplit launch of TMX_SER_SPLIT=0)"
            roofline["k_serialize"]["split_alone_us"] = round(sum(v.get("alone_us", 0) for v in ser.values()), 1)

    # ---- PMC figures of the committed rocprofv3 passes (profiles/pmc_latest.json), only if they were taken on this configuration.
    # NOT measured by this run: replayed from the builder's profile collection, and marked as such in the line
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
        if pmc["config"] == {"n_max": n, "proofs_per_gpu": P, "workload": args.workload}:
            src = "profiles/pmc_latest.json (rocprofv3 --pmc passes over tools/profile_step.py, collected by the builder; replayed, not measured in this run)"
            pk = {g: v for g, v in pmc["kernels"].items() if g != "setup"}   # (setup = the once-per-context table of B)
            if not measured_traffic:
                roofline["traffic"] = int(sum(pk[g].get("fetch_kb", 0) + pk[g].get("write_kb", 0) for g in pk) * 1024)
                roofline["traffic_source"] = src
            if "traffic" not in roofline["k_serialize"]:
                roofline["k_serialize"]["traffic"] = int((pk["k_serialize"]["fetch_kb"] + pk["k_serialize"]["write_kb"]) * 1024)
                roofline["k_serialize"]["traffic_source"] = src
            try:
                mix = json.load(open(os.path.join(ROOT, "profiles", "r03_isa_mix.json")))["kernels"]
            except (OSError, KeyError, ValueError):
                mix = {}

            def cycles(group_or_kernel, insts, fast_share):
                return insts * ((1 - fast_share) * CYCLES_SLOW + fast_share * CYCLES_FAST), insts * CYCLES_SLOW

            lo = hi = 0.0
            per_kernel_insts = {k: {"valu_insts": v} for k, v in measured_traffic["valu_insts_per_kernel"].items()} if (measured_traffic and measured_traffic.get("valu_insts_per_kernel")) else pmc.get("per_kernel", {})
            if measured_traffic and measured_traffic.get("valu_insts_per_kernel"):
                src = measured_traffic["source"]
                grp = lambda k: "k_serialize" if "k_serialize" in k else "k_proof" if "k_proof" in k else "k_verdict" if "k_verdict" in k else "k_eddsa"
                pk = {}
                for k, v in per_kernel_insts.items():
                    pk.setdefault(grp(k), {"valu_insts": 0})["valu_insts"] += v["valu_insts"]
                for g in ("k_eddsa", "k_proof", "k_serialize"):
                    pk.setdefault(g, {"valu_insts": 0})
            for kname, v in per_kernel_insts.items():
                if "valu_insts" not in v or "init_base" in kname:
                    continue
                share = mix.get(kname.split("<")[0], {}).get("fast_class_share", 0.0)
                a_, b_ = cycles(kname, v["valu_insts"], share)
                lo, hi = lo + a_, hi + b_
            step_s = (kms["k_eddsa"] + kms["k_serialize"]) * 1e-3
            step_insts = sum(pk[g].get("valu_insts", 0) for g in pk)
            roofline["valu_issue"] = {
                "unit": "wave-level VALU instructions per batch (SQ_INSTS_VALU)", "insts_source": src,
                "cycles_per_instruction": {"measured_by": "tools/microbench/valu_isa.hip (profiles/r03_valu_isa.txt), per SIMD, saturated",
                                           "default": CYCLES_SLOW, "plain_v_add_u32_v_xor_b32_v_mov_when_co_issued": CYCLES_FAST},
                "step": {"insts": int(step_insts),
                         "frac_of_issue_slots": [round(lo / (SIMDS * step_s * CLOCK_GHZ * 1e9), 4), round(hi / (SIMDS * step_s * CLOCK_GHZ * 1e9), 4)]},
                "k_eddsa": {"insts": int(pk["k_eddsa"]["valu_insts"])}, "k_proof": {"insts": int(pk["k_proof"]["valu_insts"])},
                "k_serialize": {"insts": int(pk["k_serialize"]["valu_insts"])},
                "note": "fraction of the 1024 SIMDs' issue cycles (2.4 GHz) the step's VALU instructions need over the event-timed interval: "
                        "[every fast-class instruction co-issued at 2.3 cycles, none co-issued]; the EdDSA kernels move ~1 % of HBM peak: their roof is this one"}
    except (OSError, KeyError, ValueError):
        pass

    # ---- key-cache churn: warm (0 new keys per step) and cold (all 401) are two points of a curve.  Step j replaces the first proofs of the batch
    # by proofs over FRESH validator keys (a new seed per step), so that exactly `new` keys miss per step; host clock around one call.
    try:
        from tendermintx_amd.synth import Workload
        churn = {}
        for new_keys, reps in ((0, 12), (4, 12), (40, 12), (401, 8)):
            xs = []
            for j in range(reps):
                if new_keys == 0:
                    bufs = None
                elif new_keys == 401:
                    wj = bench_workload(args.workload, n, P, seed=0x600000 + 977 * j)
                    bufs = tuple(dev_bytes(b) for b in (wj.proofs, wj.targets, wj.trusteds))
                else:   # one proof with `new_keys` fresh validators (the lanes behind them carry the dummy key, resident since the warm-up)
                    wj = Workload(0, n, 1, new_keys, chain_id=b"celestia", seed=0x700000 + 31 * j + new_keys, signed_permille=1000)
                    bufs = tuple(dev_bytes(a + b[len(a):]) for a, b in ((wj.proofs, wl.proofs), (wj.targets, wl.targets), (wj.trusteds, wl.trusteds)))
                run(ctx, 2)                     # (the base batch resident again, the schedule hint back to warm)
                xs.append(timed(ctx, 1, bufs))
                seen = ctx.key_cache_stats()["last_new_keys"]   # (a fresh batch brings 400: the dummy key of the unsigned lanes is resident)
            churn[str(seen)] = round(median(xs), 4)
        result["key_cache"]["churn"] = {"ms_per_step_by_new_keys": churn, "proofs": P,
                                        "note": "one 256-proof step in which exactly that many of the batch's 401 distinct keys are new to the context's cache "
                                                "(decoded, table built inside the step), the schedule hint saying warm; median of 8-12, host clock"}
        run(ctx, 3)
    except Exception as e:
        result["key_cache"]["churn"] = {"error": repr(e)}

    # ---- the commit pipeline on the device (reference circuits/skip.rs:119-133): section rows -> columns -> coset LDE x8 -> Poseidon
    # Merkle cap, only the cap leaves the GPU.  The SHA sections of the whole batch (the ladder section's 70 GB of extended columns + scratch do
    # not fit beside the bench's other buffers: timed at 32 proofs)
    try:
        from tendermintx_amd import _lib as _l
        te = ctx.trace_elem_count(KIND_SKIP)
        d_tr = torch.empty(P * te, dtype=torch.int64, device=dev)
        bi = tuple(dev_bytes(b) for b in (wl.proofs, wl.targets, wl.trusteds))
        run(ctx, 1, bi)
        ctx.trace_rows_device(KIND_SKIP, P, bi[1].data_ptr(), bi[2].data_ptr(), d_tr.data_ptr(), _l.TRACE_ALL, stream.cuda_stream)
        cap = torch.zeros(4 << 4, dtype=torch.int64, device=dev)
        cp = {}
        for name, sec, pp in (("sha512", _l.TRACE_SHA512, P), ("sha256_leaves", _l.TRACE_SHA256, P), ("tree", _l.TRACE_TREE, P), ("header", _l.TRACE_HEADER, P),
                              ("ladders", _l.TRACE_LADDERS, min(P, 32))):
            log_rows, width = ctx.trace_commit_shape(KIND_SKIP, sec)
            tn_, sz_ = 0, n
            while sz_ > 1:
                sz_ = (sz_ + 1) // 2
                tn_ += sz_
            rows = {"sha512": 2 * n * 80, "sha256_leaves": 2 * n * 64, "tree": 2 * tn_ * 128, "header": 4 * 5 * 128, "ladders": 2 * n * 256}[name]
            for _ in range(2):
                ctx.trace_commit_device(KIND_SKIP, pp, sec, 3, 4, d_tr.data_ptr(), cap.data_ptr(), stream.cuda_stream)
            ms = ctx.trace_commit_last_ms()
            cols = pp * width
            col_b, lde_b = (cols << log_rows) * 8, (cols << (log_rows + 3)) * 8
            perms = (1 << (log_rows + 3)) * ((cols + 7) // 8) + (1 << (log_rows + 3))
            cp[name] = {"proofs": pp, "columns": cols, "log_rows": log_rows, "log_rows_extended": log_rows + 3,
                        "ms": {k: round(v, 4) for k, v in ms.items()}, "ms_total": round(sum(ms.values()), 4),
                        "rows_per_proof": rows,
                        "columns_stage": {"bytes": cols * rows * 8 + col_b, "frac_of_hbm": round(gbs(cols * rows * 8 + col_b, ms["columns"]) / HBM_PEAK_GBS, 4),
                                          "note": "the section's own rows read once, the zero-padded columns written once"},
                        "lde_stage": {"bytes_one_read_one_write_per_pass": 2 * (2 * col_b) + 2 * (2 * lde_b) + 2 * lde_b,
                                      "frac_passes": round(gbs(4 * col_b + 6 * lde_b, ms["lde"]) / HBM_PEAK_GBS, 4),
                                      "frac_1r1w": round(gbs(col_b + lde_b, ms["lde"]) / HBM_PEAK_GBS, 4),
                                      "note": "frac_passes: what the two-pass inverse + two-pass forward transforms move (a read and a write per pass); "
                                              "frac_1r1w: the algorithmic minimum, the columns read once and the extended columns written once"},
                        "merkle_stage": {"permutations": perms, "gperm_per_s": round(perms / (ms["merkle"] * 1e-3) / 1e9, 3),
                                         "bytes_read": lde_b, "frac_of_hbm": round(gbs(lde_b, ms["merkle"]) / HBM_PEAK_GBS, 4)},
                        "cap0": [int(x) & (2**64 - 1) for x in cap[:4].cpu().numpy()]}
        result["commit_pipeline"] = {"what": "tmx_trace_commit_device per section: rows of every proof -> n_proofs x width columns (tiled transpose) -> coset LDE, "
                                             "blow-up 8 -> Poseidon Merkle tree over the extended rows (leaf = the row across all columns), cap of 16 digests; "
                                             "HIP events between the stages; parity of the chain vs oracle/c: tests/test_commit_pipeline.py (Poseidon constants: the "
                                             "context's, by default the Grain stream -- parity unpinned against plonky2)", "sections": cp}
        del d_tr
    except Exception as e:
        result["commit_pipeline"] = {"error": repr(e)}

    # ---- Level-2 trace rows of the same batch: the writer of 266 KB of ladder rows per lane
    try:
        from tendermintx_amd import _lib
        te = ctx.trace_elem_count(KIND_SKIP)
        d_tr = torch.empty(P * te, dtype=torch.int64, device=dev)
        bi = tuple(dev_bytes(b) for b in (wl.proofs, wl.targets, wl.trusteds))
        run(ctx, 1, bi)                                       # the Level-1 lane records the trace kernels read
        tn, sz = 0, n
        while sz > 1:
            sz = (sz + 1) // 2
            tn += sz
        sec_elems = {"ladders": (_lib.TRACE_LADDERS, n * 2 * 256 * 65), "sha512": (_lib.TRACE_SHA512, n * 2880), "sha256": (_lib.TRACE_SHA256, n * 1152),
                     "match": (_lib.TRACE_MATCH, n * n), "tree": (_lib.TRACE_TREE, 2 * tn * 1152), "header": (_lib.TRACE_HEADER, 20 * 1152),
                     "all": (_lib.TRACE_ALL, te)}
        l2 = {}
        for name, (mask, elems) in sec_elems.items():
            ctx.trace_rows_device(KIND_SKIP, P, bi[1].data_ptr(), bi[2].data_ptr(), d_tr.data_ptr(), mask, stream.cuda_stream)
            torch.cuda.synchronize(dev)
            a = time.perf_counter()
            for _ in range(3):
                ctx.trace_rows_device(KIND_SKIP, P, bi[1].data_ptr(), bi[2].data_ptr(), d_tr.data_ptr(), mask, stream.cuda_stream)
            torch.cuda.synchronize(dev)
            ms = 1e3 * (time.perf_counter() - a) / 3
            l2[name] = {"ms": round(ms, 4), "bytes": P * elems * 8, "gbs": round(gbs(P * elems * 8, ms), 1)}
        result["level2_trace_rows"] = {
            "what": "row-level trace of both scalar multiplications (256 rows x 65 elements each), SHA-512 / leaf SHA-256 round states, N x N match "
                    "bits of every lane of the batch, SHA-256 round states of the inner nodes of both validator trees and of the header proofs: "
                    "this build's own row specification, DESIGN.md 'Level-2 trace rows'",
            "sections": l2, "ms_per_batch": l2["all"]["ms"], "ms_per_proof": round(l2["all"]["ms"] / P, 5),
            "roofline": {"kernel": "k_trace_ladder_pass1 + _pass2", "bound": "hbm", "achieved": l2["ladders"]["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(l2["ladders"]["gbs"] / HBM_PEAK_GBS, 4), "algorithmic_bytes": l2["ladders"]["bytes"], "traffic": None,
                         "note": "pass 1 (the double-and-add chain + the running product of the Z's, one thread per ladder) is one latency-bound "
                                 "wave per SIMD; pass 2 (one wave-shared inversion per 64 x 16 rows, Montgomery's trick walked backwards over the stored "
                                 "prefix products, canonical limbs, stores) runs beside the next segment's chain at four waves per SIMD; ~3.1 k + ~3.3 k "
                                 "instructions per row; the scratch rows are 280 B (round 4: 320): DESIGN.md 'The writer, measured'"}}
        l2_traffic = measure_level2_traffic(n, P, args.workload)  # (extras() runs on rank 0 of a one-GPU run only)
        if l2_traffic:
            result["level2_trace_rows"]["roofline"]["traffic"] = l2_traffic["bytes_per_call"]
            result["level2_trace_rows"]["roofline"]["traffic_detail"] = l2_traffic
            # raw counters, and with the guide's x2 on the fetches of pass 2 (the scratch rows are read back with 16-byte loads)
            p2f = l2_traffic["per_kernel"].get("k_trace_ladder_pass2", {}).get("FETCH_SIZE", 0)
            result["level2_trace_rows"]["roofline"]["traffic_over_rows"] = {
                "raw": round(l2_traffic["bytes_per_call"] / l2["ladders"]["bytes"], 3),
                "fetch_x2": round((l2_traffic["bytes_per_call"] + p2f) / l2["ladders"]["bytes"], 3),
                "note": "HBM bytes of both passes over the bytes of ladder rows; the floor with a scratch buffer is 1 + 2 x 280 / 520 = 2.08 (pass 1 writes "
                        "seven field elements per row, pass 2 reads them back): recomputing them in pass 2 instead costs 15 more field products per row on a "
                        "kernel pair that is instruction-bound (docs/kernels.md 'Level-2 ladders, round 5')"}
        tr0 = d_tr[:te].cpu().numpy().view(np.uint64)
        del d_tr
    except Exception as e:  # the trace rows are a widening row: never let them take the headline line down
        result["level2_trace_rows"] = {"error": repr(e)}
        tr0 = None

    if args.no_cpu_baseline:
        return
    sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
    import oracle_c as oc  # checker + reported CPU baseline only
    S = min(P, 64)
    sl_p, sl_t, sl_r = wl.proofs[:S * 2336], wl.targets[:S * n * 256], wl.trusteds[:S * n * 48]
    a = time.perf_counter()
    o_elems, o_reps = oc.witness_batch(KIND_SKIP, S, sl_p, sl_t, sl_r, n, b"celestia", 100800, n_threads=1)
    t1 = time.perf_counter() - a
    t1c = oc.witness_pool_seconds(KIND_SKIP, S, sl_p, sl_t, sl_r, n, b"celestia", 100800, 1, 1)   # compute only, like the pool
    cores, cores_note = usable_cores()
    rep_all = max(1, (32 * cores + P - 1) // P)            # >= 32 proofs per thread
    tn = min(oc.witness_pool_seconds(KIND_SKIP, P, wl.proofs, wl.targets, wl.trusteds, n, b"celestia", 100800, rep_all, cores) for _ in range(4))
    per_proof_1, per_proof_n = t1c / S, tn / (P * rep_all)
    # parity of the timed GPU output against the oracle on the same sample
    run(ctx, 1)
    torch.cuda.synchronize(dev)
    g = d_out[:S, :count].cpu().numpy().view(np.uint64)
    result["parity_vs_oracle"] = {"proofs_checked": S, "bit_exact": bool(np.array_equal(g, o_elems))}
    result["cpu_baseline"] = {
        "value": round(1e3 * t1 / S, 4), "unit": "ms", "cores": 1, "kind": "port",
        "sample": f"{S} proofs x N={n} ({S * n} validator lanes) of the timed workload, 1 pass, oracle/c single thread, full witness rows written",
        "compute_only_ms": round(1e3 * per_proof_1, 4),
        "all_cores": {"value": round(1e3 * per_proof_n, 5), "unit": "ms", "cores": cores, "cores_note": cores_note,
                      "sample": f"{P * rep_all} proofs ({rep_all} x the batch) dealt round-robin to a persistent pool of {cores} threads "
                                f"({P * rep_all // cores} proofs per thread), compute only, best of 4",
                      "speedup_vs_1_thread": round(per_proof_1 / per_proof_n, 1), "scaling_efficiency": round(per_proof_1 / per_proof_n / cores, 3)}}
    if tr0 is not None:  # constraint checker (oracle/c/tmxo_trace.c) on the rows of proof 0
        a = time.perf_counter()
        code = oc.trace_check(KIND_SKIP, wl.proofs[:2336], wl.targets[:n * 256], wl.trusteds[:n * 48], n, tr0)
        result["level2_trace_rows"]["constraint_check"] = {"proof": 0, "violations": code, "checker_ms": round(1e3 * (time.perf_counter() - a), 1)}
    ossl = openssl_verify_us(wl, n)
    if ossl is not None:
        result["cpu_baseline"]["openssl_evp_digestverify"] = ossl


def measure_traffic(n, P, workload):
    """HBM bytes of one warm step from rocprofv3's FETCH_SIZE / WRITE_SIZE (KB; memory-side request counters of the L2), each in a pass of
    its own over tools/profile_step.py (2 warm + 6 counted batches).  Uncorrected sums, as MI355X_MICROARCH.md prescribes for access patterns
    that are not wide coalesced reads (the reads here are 1- and 4-byte gathers; writes dominate).  None when rocprofv3 is not usable."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    if not shutil.which("rocprofv3") or os.environ.get("TMX_BENCH_NO_PMC") == "1":
        return None
    tot, per_kernel = {}, {}
    warm, steps = 2, 6
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            env = dict(os.environ, TMPDIR="/tmp", P=str(P), N=str(n), WORKLOAD=workload, WARM=str(warm), STEPS=str(steps))
            for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
                out = os.path.join(tmp, counter)
                r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", out, "-o", "step", "--", sys.executable,
                                    os.path.join(ROOT, "tools", "profile_step.py")], cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
                dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
                if r.returncode != 0 or not dbs:
                    return None
                db = sqlite3.connect(dbs[0])
                starts = sorted(x[0] for x in db.execute("select distinct dispatch_id from counters_collection where kernel_name like '%k_proof%'"))
                if len(starts) != warm + steps:
                    return None
                first = starts[warm]
                for name, value in db.execute("select kernel_name, sum(value) from counters_collection where counter_name = ? and dispatch_id >= ? "
                                              "and kernel_name like '%tmx::%' and kernel_name not like '%k_init_base%' group by kernel_name", (counter, first)):
                    tot[counter] = tot.get(counter, 0.0) + value / steps
                    short = name.split("(")[0].replace("void ", "").replace("tmx::", "")
                    per_kernel.setdefault(short, {})[counter] = round(value / steps * (1 if counter == "SQ_INSTS_VALU" else 1024))
                if counter == "SQ_INSTS_VALU":  # the same pass, kernels serialized by the counter collection: each kernel's time ALONE, per batch
                    t_first = sorted(x[0] for x in db.execute("select start from kernels where name like '%k_proof%'"))[warm]
                    for name, ns in db.execute("select name, sum(end - start) from kernels where start >= ? and name like '%tmx::%' "
                                               "and name not like '%k_init_base%' group by name", (t_first,)):
                        short = name.split("(")[0].replace("void ", "").replace("tmx::", "")
                        per_kernel.setdefault(short, {})["alone_us"] = round(ns / steps / 1e3, 2)
        if "FETCH_SIZE" not in tot or "WRITE_SIZE" not in tot:
            return None
        valu = {k: v["SQ_INSTS_VALU"] for k, v in per_kernel.items() if "SQ_INSTS_VALU" in v}
        return {"bytes_per_step": int((tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024), "fetch_bytes": int(tot["FETCH_SIZE"] * 1024),
                "write_bytes": int(tot["WRITE_SIZE"] * 1024), "per_kernel": per_kernel, "valu_insts_per_kernel": valu,
                "source": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_INSTS_VALU (three passes) over "
                          f"tools/profile_step.py, mean of {steps} warm batches"}
    except Exception:
        return None


def measure_level2_traffic(n, P, workload):
    """HBM bytes of one call of the Level-2 ladder section (k_trace_ladder_pass1 + _pass2 of every segment) from rocprofv3's FETCH_SIZE /
    WRITE_SIZE, each in a pass of its own over tools/trace_bench.py (SECTIONS=ladders: 2 warm + 5 timed calls, all counted -- the kernels do
    the same work every call).  None when rocprofv3 is not usable."""
    import glob
    import shutil
    import sqlite3
    import tempfile
    if not shutil.which("rocprofv3") or os.environ.get("TMX_BENCH_NO_PMC") == "1":
        return None
    calls, tot, per_kernel = 7, {}, {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
            env = dict(os.environ, TMPDIR="/tmp", P=str(P), N=str(n), WORKLOAD=workload, SECTIONS="ladders")
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                out = os.path.join(tmp, counter)
                r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", out, "-o", "l2", "--", sys.executable,
                                    os.path.join(ROOT, "tools", "trace_bench.py")], cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
                dbs = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
                if r.returncode != 0 or not dbs:
                    return None
                db = sqlite3.connect(dbs[0])
                for name, value in db.execute("select kernel_name, sum(value) from counters_collection where counter_name = ? "
                                              "and kernel_name like '%k_trace_ladder%' group by kernel_name", (counter,)):
                    tot[counter] = tot.get(counter, 0.0) + value / calls
                    short = "k_trace_ladder_pass1" if "pass1" in name else "k_trace_ladder_pass2"
                    per_kernel.setdefault(short, {})[counter] = per_kernel.get(short, {}).get(counter, 0) + round(value / calls * 1024)
        if "FETCH_SIZE" not in tot or "WRITE_SIZE" not in tot:
            return None
        return {"bytes_per_call": int((tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024), "fetch_bytes": int(tot["FETCH_SIZE"] * 1024),
                "write_bytes": int(tot["WRITE_SIZE"] * 1024), "per_kernel": per_kernel,
                "source": "measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) over tools/trace_bench.py "
                          f"(SECTIONS=ladders), mean of {calls} calls; uncorrected sums (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced "
                          "reads by 2x -- pass 2 reads the 4.7 GB of scratch pass 1 writes with 16-byte loads)"}
    except Exception:
        return None


def usable_cores():
    """Host threads this process can really run at once: the smaller of the affinity mask and the cgroup CPU quota (the GPU boxes report
    256 logical CPUs and grant 16 of them: beyond the quota more threads only get throttled -- measured 15.9x at 16 threads, 11x at 256)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"os.cpu_count() = {os.cpu_count()}, affinity = {n}"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = max(1, int(quota) // int(period))
            note += f", cgroup cpu.max = {quota}/{period} -> {q} CPUs"
            n = min(n, q)
    except (OSError, ValueError):
        try:
            q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                note += f", cfs quota {q}/{per} -> {max(1, q // per)} CPUs"
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n, note


def openssl_verify_us(wl, n, budget_s=2.0):
    """SURVEY 8(d)'s optional independent datapoint: OpenSSL's Ed25519 verification (EVP_DigestVerify) of the signed lanes of the
    workload, one thread.  Verification only -- none of the witness values -- so it is a lower bound for any host path."""
    try:
        import ctypes as C
        from tendermintx_amd import synth
        L = synth._libcrypto()
        L.EVP_PKEY_new_raw_public_key.restype = C.c_void_p
        L.EVP_PKEY_new_raw_public_key.argtypes = [C.c_int, C.c_void_p, C.c_char_p, C.c_size_t]
        L.EVP_DigestVerifyInit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.EVP_DigestVerify.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        lanes = []
        for i in range(min(len(wl.targets) // 256, 4 * n)):
            r = wl.targets[256 * i:256 * (i + 1)]
            if r[223] & 1:
                lanes.append((r[:32], r[32:96], r[96:96 + int.from_bytes(r[220:222], "little")]))
        keys = [L.EVP_PKEY_new_raw_public_key(synth.EVP_PKEY_ED25519, None, pk, 32) for pk, _, _ in lanes]
        done, ok, a = 0, 0, time.perf_counter()
        while time.perf_counter() - a < budget_s:
            for k, (_, sig, msg) in zip(keys, lanes):
                c = L.EVP_MD_CTX_new()
                L.EVP_DigestVerifyInit(c, None, None, None, k)
                ok += L.EVP_DigestVerify(c, sig, 64, msg, len(msg)) == 1
                L.EVP_MD_CTX_free(c)
            done += len(lanes)
        dt = time.perf_counter() - a
        for k in keys:
            L.EVP_PKEY_free(k)
        if ok != done:
            return None
        return {"us_per_verify": round(1e6 * dt / done, 2), "ms_per_proof_equivalent": round(1e3 * dt / done * n, 3), "cores": 1, "verifies": done,
                "note": "signature verification only, through ctypes (a few hundred ns of call overhead per verify); not a witness"}
    except Exception:  # no libcrypto, or an API mismatch: the datapoint is optional
        return None


if __name__ == "__main__":
    sys.exit(main())
