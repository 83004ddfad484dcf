import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import torch
import oracle_c as oc
import tendermintx_amd as tmx
from tendermintx_amd.synth import Workload
from tendermintx_amd import sharding
dev = torch.device("cuda", 0)
n = 512
wl = Workload(0, n, 1, 400, chain_id=b"celestia", seed=2024, signed_permille=900)
d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (wl.proofs, wl.targets, wl.trusteds)]
with tmx.Context(n, b"celestia", max_batch=1) as ctx:
    fn = sharding.make_gpu_eddsa_fn(ctx)
    ed = fn(d[1].view(n, 256))
    torch.cuda.synchronize()
    ed2 = torch.from_numpy(ctx.eddsa_lanes(wl.targets)).to(dev)
    print("ed device == ed host path:", torch.equal(ed, ed2))
    out = torch.zeros(ctx.elem_stride(0), dtype=torch.int64, device=dev)
    rep = torch.zeros(64, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    ctx.finish_batch_device(0, 1, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), ed.data_ptr(), out.data_ptr(), rep.data_ptr(), s)
    torch.cuda.synchronize()
    got = out[:ctx.elem_count(0)].cpu().numpy().view(np.uint64)
    want, orep = oc.witness(0, wl.proofs, wl.targets, wl.trusteds, b"celestia", 100800)
    diff = np.nonzero(got != want)[0]
    print("diffs", len(diff), diff[:20], "elem_count", len(want))
    print("rep", bytes(rep.cpu().numpy()).hex())
    e2, r2 = ctx.witness_batch(0, wl.proofs, wl.targets, wl.trusteds)
    print("host path equal:", np.array_equal(e2[0], want), r2[0]["all_ok"], orep["all_ok"])
