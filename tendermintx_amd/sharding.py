"""Multi-GPU partitioning of the witness path: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Two modes:

* proof-sharded batch (BASELINE config 4): proofs are independent, rank r takes a contiguous slice, there is NO data-path
  collective; `gather_rows` optionally reassembles the rows on every rank with one all-gather.
* validator-sharded single proof (BASELINE config 5): the N lanes of one proof are split across ranks for the EdDSA stage
  (the only expensive stage); the 448-byte lane records are exchanged with ONE all-gather (N*448 B = 224 KB at N = 512:
  latency-bound, a direct all-gather uses every xGMI link once), then each rank finishes the proof on the reassembled
  records (Merkle trees, tallies, serialization are cheap and replicated).

The compute steps are injected callables so that the partition/exchange logic is testable on CPU with gloo (the tests
inject the oracle there; the product default is the HIP path of `Context`, which needs a GPU).
"""
import torch
import torch.distributed as dist

ED_STRIDE = 448


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) of `n_items` for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_rows(local_rows, n_total, group=None):
    """All-gather equally-shaped per-rank row blocks (padded to the largest shard) and strip the padding.
    local_rows: [n_local, width] tensor on the rank's device.  Returns [n_total, width]."""
    world = dist.get_world_size(group)
    per = (n_total + world - 1) // world
    width = local_rows.shape[1]
    pad = torch.zeros((per, width), dtype=local_rows.dtype, device=local_rows.device)
    pad[:local_rows.shape[0]] = local_rows
    out = torch.empty((world * per, width), dtype=local_rows.dtype, device=local_rows.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    pieces = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        pieces.append(out[r * per:r * per + (hi - lo)])
    return torch.cat(pieces, dim=0)


def validator_sharded_eddsa(target_lanes, eddsa_fn, group=None):
    """target_lanes: uint8 tensor [N, 256] (identical on every rank).  Each rank runs `eddsa_fn(lanes[lo:hi]) -> uint8
    [hi-lo, 448]` on its slice; returns the reassembled [N, 448] records on every rank (one all-gather)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = target_lanes.shape[0]
    lo, hi = shard_range(n, rank, world)
    local = eddsa_fn(target_lanes[lo:hi])
    assert local.shape == (hi - lo, ED_STRIDE) and local.dtype == torch.uint8
    return gather_rows(local, n, group)


def make_gpu_eddsa_fn(ctx, stream=None):
    """Product default: k_eddsa on this rank's GPU through the C ABI (tmx_eddsa_lanes_device)."""
    def fn(lanes):
        lanes = lanes.contiguous()
        out = torch.empty((lanes.shape[0], ED_STRIDE), dtype=torch.uint8, device=lanes.device)
        s = stream if stream is not None else int(torch.cuda.current_stream(lanes.device).cuda_stream)
        ctx.eddsa_lanes_device(lanes.shape[0], lanes.data_ptr(), out.data_ptr(), s)
        return out
    return fn


def validator_sharded_skip(ctx, kind, proof, target, trusted, group=None):
    """One proof, lanes split across the ranks' GPUs.  proof/target/trusted: uint8 device tensors (identical on every
    rank).  Returns (elements int64 [elem_count], report uint8 [64]) on every rank."""
    dev = target.device
    n = ctx.n_max
    ed = validator_sharded_eddsa(target.view(n, 256), make_gpu_eddsa_fn(ctx), group).contiguous()
    out = torch.zeros(ctx.elem_stride(kind), dtype=torch.int64, device=dev)
    rep = torch.zeros(64, dtype=torch.uint8, device=dev)
    s = int(torch.cuda.current_stream(dev).cuda_stream)
    ctx.finish_batch_device(kind, 1, proof.data_ptr(), target.data_ptr(), trusted.data_ptr() if trusted is not None else None,
                            ed.data_ptr(), out.data_ptr(), rep.data_ptr(), s)
    return out[:ctx.elem_count(kind)], rep
