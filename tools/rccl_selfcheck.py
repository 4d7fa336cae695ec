import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
from acgpathtracing_amd import distributed as D
t = torch.ones((1080,1920,4), device="cuda")
D.barrier(); D.reduce_accumulation(t, dst=0); torch.cuda.synchronize()
print("reduce ok", float(t.sum()))
print("max", D.max_over_ranks(1.5, torch.device("cuda",0)), "sum", D.sum_over_ranks([1,2,3], torch.device("cuda",0)))
dist.destroy_process_group()
print("NCCL_OK")
