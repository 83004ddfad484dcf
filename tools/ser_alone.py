# serializer alone (TMX_SER_SPLIT=0): event time of k_serialize for the lib in $TMX_LIB
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from tendermintx_amd import Context, _lib
from tendermintx_amd.context import KIND_SKIP
from tendermintx_amd.synth import bench_workload
P, n = 256, 128
w = bench_workload("survey8d", n, P, seed=7)
dev = torch.device("cuda:0")
d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (w.proofs, w.targets, w.trusteds)]
stride = int(_lib.lib().tmx_elem_stride(KIND_SKIP, n))
out = torch.empty(P * stride, dtype=torch.int64, device=dev); rep = torch.empty(P * 64, dtype=torch.uint8, device=dev)
s = torch.cuda.Stream(dev)
os.environ["TMX_SER_SPLIT"] = "0"
ctx = Context(n, b"celestia", 100800, device=0, max_batch=P)
for _ in range(30):
    ctx.witness_batch_device(KIND_SKIP, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), rep.data_ptr(), s.cuda_stream)
torch.cuda.synchronize()
print("k_serialize alone ms:", round(ctx.kernel_ms_mean(20)["k_serialize"], 4), os.path.basename(os.environ.get("TMX_LIB", "default")))
