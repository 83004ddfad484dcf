#!/usr/bin/env python3
"""A/B of the serializer's span (elements per wave) on the bench workload: k_serialize alone (TMX_SER_SPLIT=0) and the whole step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from tendermintx_amd import Context  # noqa: E402
from tendermintx_amd.context import KIND_SKIP  # noqa: E402
from tendermintx_amd.synth import Workload  # noqa: E402
from tendermintx_amd import _lib  # noqa: E402

P, n = 256, 128
w = Workload(KIND_SKIP, n, P, n, chain_id=b"celestia", seed=7)
dev = torch.device("cuda:0")
d_proofs = torch.frombuffer(bytearray(w.proofs), dtype=torch.uint8).to(dev)
d_targets = torch.frombuffer(bytearray(w.targets), dtype=torch.uint8).to(dev)
d_trusteds = torch.frombuffer(bytearray(w.trusteds), dtype=torch.uint8).to(dev)
stride = int(_lib.lib().tmx_elem_stride(KIND_SKIP, n))
d_out = torch.empty(P * stride, dtype=torch.int64, device=dev)
d_rep = torch.empty(P * 64, dtype=torch.uint8, device=dev)
stream = torch.cuda.Stream(dev)
for rep in range(int(os.environ.get("REPS", "2"))):
    for span in [int(x) for x in os.environ.get("SPANS", "128,256,512").split(",")]:
        os.environ["TMX_SER_SPAN"] = str(span)
        res = []
        for split in ("0", "1"):
            os.environ["TMX_SER_SPLIT"] = split
            ctx = Context(n, b"celestia", 100800, device=0, max_batch=P)
            for _ in range(4):
                ctx.witness_batch_device(KIND_SKIP, P, d_proofs.data_ptr(), d_targets.data_ptr(), d_trusteds.data_ptr(), d_out.data_ptr(),
                                         d_rep.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(20):
                ctx.witness_batch_device(KIND_SKIP, P, d_proofs.data_ptr(), d_targets.data_ptr(), d_trusteds.data_ptr(), d_out.data_ptr(),
                                         d_rep.data_ptr(), stream.cuda_stream)
            torch.cuda.synchronize(dev)
            ms = 1e3 * (time.perf_counter() - t0) / 20
            res.append((ms, ctx.kernel_ms_mean(20)["k_serialize"]))
            ctx.close()
        print(f"span {span}: serialize alone {res[0][1]:.4f} ms (step {res[0][0]:.4f}); split step {res[1][0]:.4f} ms, final part {res[1][1]:.4f}", flush=True)
