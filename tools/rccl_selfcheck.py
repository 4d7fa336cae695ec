#!/usr/bin/env python3
"""RCCL bring-up on one GPU: backend "nccl" (= RCCL on ROCm), world size 1, the very collectives
acgpathtracing_amd.distributed issues for N > 1 — reduce(SUM) of a float4 [1080, 1920, 4] accumulation buffer to rank 0,
all_reduce(MAX) of the elapsed time, all_reduce(SUM) of the counters, barrier — on DEVICE tensors.  With one rank the
collectives move no data between GPUs, but communicator creation, stream ordering against torch's stream and the RCCL
entry points themselves run on the hardware.  Prints NCCL_OK."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"

t = torch.full((1080, 1920, 4), 0.25, device=dev)
dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)          # what distributed.reduce_accumulation calls for world > 1
dist.barrier()
torch.cuda.synchronize()
assert float(t.sum()) == 0.25 * 1080 * 1920 * 4
m = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(m, op=dist.ReduceOp.MAX)             # max_over_ranks
s = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64, device=dev)
dist.all_reduce(s, op=dist.ReduceOp.SUM)             # sum_over_ranks
torch.cuda.synchronize()
assert float(m.item()) == 1.5 and s.tolist() == [1.0, 2.0, 3.0]

# and through the product's own helpers (no-ops at world 1, but the import path and signatures are the bench's)
from acgpathtracing_amd import distributed as D  # noqa: E402
D.barrier(); D.reduce_accumulation(t, dst=0)
assert D.max_over_ranks(2.5, dev) == 2.5 and D.sum_over_ranks([1, 2], dev) == [1.0, 2.0]
dist.destroy_process_group()
print("NCCL_OK")
