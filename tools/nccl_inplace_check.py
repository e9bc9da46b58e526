import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
out = torch.arange(24, dtype=torch.float64, device="cuda").reshape(1, 4, 6)
ref = out.clone()
dist.all_gather_into_tensor(out.view(-1), out[0].view(-1))
torch.cuda.synchronize()
print("in-place all_gather_into_tensor on RCCL, world 1:", torch.equal(out, ref))
dist.destroy_process_group()
