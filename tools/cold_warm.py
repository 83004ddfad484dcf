import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from tendermintx_amd import Context
from tendermintx_amd.synth import bench_workload
P, n = int(os.environ.get("P", "256")), 128
dev = torch.device("cuda:0")
for wlname in ("survey8d", "one_set"):
    w = bench_workload(wlname, n, P, seed=7)
    d = [torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev) for b in (w.proofs, w.targets, w.trusteds)]
    ctx = Context(n, b"celestia", 100800, max_batch=P)
    out = torch.empty(P * ctx.elem_stride(0), dtype=torch.int64, device=dev); rep = torch.empty(P * 64, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream(dev)
    def run(k):
        for _ in range(k):
            ctx.witness_batch_device(0, P, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), out.data_ptr(), rep.data_ptr(), s.cuda_stream)
    run(10); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(40); torch.cuda.synchronize(); warm = 1e3 * (time.perf_counter() - t0) / 40
    cold = []
    for _ in range(10):
        ctx.key_cache_flush(); torch.cuda.synchronize(); a = time.perf_counter(); run(1); s.synchronize(); cold.append(1e3 * (time.perf_counter() - a)); torch.cuda.synchronize()
    cold.sort()
    print(os.environ.get("TMX_LIB", "default").split("/")[-1], wlname, "P", P, "warm %.4f cold %.4f" % (warm, cold[5]), flush=True)
    ctx.close()
