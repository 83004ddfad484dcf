import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import torch
import torch.distributed as dist
import oracle_c as oc
import tendermintx_amd as tmx
from tendermintx_amd.synth import Workload
from tendermintx_amd import sharding
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29411")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = 512
wl = Workload(0, n, 1, 400, chain_id=b"celestia", seed=2024, signed_permille=900)
d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (wl.proofs, wl.targets, wl.trusteds)]
with tmx.Context(n, b"celestia", max_batch=1) as ctx:
    ed_ref = torch.from_numpy(ctx.eddsa_lanes(wl.targets)).to(dev)
    ed = sharding.validator_sharded_eddsa(d[1].view(n, 256), sharding.make_gpu_eddsa_fn(ctx)).contiguous()
    torch.cuda.synchronize()
    print("gathered ed == ref:", torch.equal(ed, ed_ref), ed.shape, ed.dtype, ed.is_contiguous())
    for trial in range(3):
        elems, rep = sharding.validator_sharded_skip(ctx, 0, d[0], d[1], d[2])
        torch.cuda.synchronize(dev)
        got = elems.cpu().numpy().view(np.uint64)
        want, orep = oc.witness(0, wl.proofs, wl.targets, wl.trusteds, b"celestia", 100800)
        diff = np.nonzero(got != want)[0]
        print("trial", trial, "diffs", len(diff), diff[:10], diff[-5:])
dist.destroy_process_group()
