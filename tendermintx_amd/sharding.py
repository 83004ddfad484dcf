"""Multi-GPU partitioning of the witness path: one process per GPU, the exchange behind the C ABI (include/tmx.h "multi-GPU").

Two modes, both implemented in libtmx (`tmx_witness_batch_sharded_device`, `tmx_witness_validator_sharded_device`) so that
the host the reference actually has -- the Rust process behind `SkipOffchainInputs::hint`, reference circuits/skip.rs:64-102 -- can use them
without Python; this module is the thin Python caller the tests and bench.py drive:

* proof-sharded batch (BASELINE configs[3]): proofs are independent, rank r computes a contiguous slice of the rows in place in a
  full-size buffer; `gather=True` makes every row resident on every rank with ONE grouped RCCL exchange (each rank broadcasts its slice
  in place: no padding, no staging copy).  Without it there is no data-path collective at all.
* validator-sharded proofs (BASELINE configs[4]): the lanes are split across the ranks for the EdDSA stage, ONE grouped exchange of the
  448-byte lane records (224 KB at N = 512: latency-bound), then every rank finishes the proof on the reassembled records.

The only thing `torch.distributed` carries here is the bootstrap: the 128-byte RCCL unique id from rank 0 to the others (any channel
would do: a file, the host's RPC) and barriers around timed regions.
"""
import ctypes as C

from . import _lib

ED_STRIDE = 448


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) of `n_items` for `rank`; sizes differ by at most one.  The C ABI's tmx_shard_range: one partition rule for every host."""
    lo, hi = C.c_uint64(), C.c_uint64()
    _lib.lib().tmx_shard_range(n_items, rank, world, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def unique_id():
    """128-byte RCCL unique id (rank 0 creates it and hands it to the other ranks)."""
    buf = C.create_string_buffer(128)
    _lib.check(_lib.lib().tmx_comm_unique_id(buf))
    return buf.raw


def connect(ctx, group=None):
    """Give `ctx` an RCCL communicator over the ranks of the torch.distributed `group`: the id travels through the group's own backend (gloo
    or nccl), the data path afterwards is libtmx's.  world_size 1: no id, nothing loaded."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        ctx.comm_create(None, 0, 1)
        return 0, 1
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    ctx.comm_create(box[0], rank, world)
    return rank, world


def proof_sharded_batch(ctx, kind, n_total, d_proofs, d_targets, d_trusteds, d_out, d_reports, gather=False, stream=None):
    """All arguments are device tensors holding ALL n_total proofs / rows (identical inputs on every rank); this rank fills its rows,
    `gather` fills the others'."""
    if stream is None:  # the buffers were produced on torch's current stream: enqueue behind it (the context's own stream would race their fill)
        import torch
        stream = int(torch.cuda.current_stream(d_targets.device).cuda_stream)
    ctx.witness_batch_sharded_device(kind, n_total, d_proofs.data_ptr(), d_targets.data_ptr(), d_trusteds.data_ptr() if d_trusteds is not None else None,
                                     d_out.data_ptr() if d_out is not None else None, d_reports.data_ptr(), gather, stream)


def validator_sharded_skip(ctx, kind, proof, target, trusted, n_proofs=1, stream=None):
    """Lanes split across the ranks' GPUs.  proof/target/trusted: uint8 device tensors (identical on every rank).  Returns
    (elements int64 [n_proofs, elem_stride], reports uint8 [n_proofs * 64]) on every rank; one proof: ([elem_count], [64]) as before."""
    import torch
    dev = target.device
    if stream is None:  # out / rep are zero-filled on torch's current stream and handed back to it: the kernels must be ordered with both
        stream = int(torch.cuda.current_stream(dev).cuda_stream)
    out = torch.zeros((n_proofs, ctx.elem_stride(kind)), dtype=torch.int64, device=dev)
    rep = torch.zeros(n_proofs * 64, dtype=torch.uint8, device=dev)
    ctx.witness_validator_sharded_device(kind, n_proofs, proof.data_ptr(), target.data_ptr(), trusted.data_ptr() if trusted is not None else None,
                                         out.data_ptr(), rep.data_ptr(), stream)
    if n_proofs == 1:
        return out[0, :ctx.elem_count(kind)], rep
    return out, rep
