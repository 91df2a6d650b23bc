import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
os.environ["RANK"]="0"; os.environ["WORLD_SIZE"]="1"; os.environ["LOCAL_RANK"]="0"
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
y = torch.zeros((4, 8), dtype=torch.complex128, device="cuda:0")
dst = torch.empty((4, 8), dtype=torch.complex128, device="cuda:0")
dist.all_gather_into_tensor(torch.view_as_real(dst), torch.view_as_real(y))
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("nccl world-1 ok", t.item())
