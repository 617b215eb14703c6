import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
for n in (10_000_000, 60_000_000, 134_000_000, 139_000_000, 200_000_000):
    send = torch.randint(0, 2**62, (n, 2), dtype=torch.int64, device="cuda")
    recv = torch.zeros_like(send)
    dist.all_to_all_single(recv, send, output_split_sizes=[n], input_split_sizes=[n])
    torch.cuda.synchronize()
    bad = int((recv != send).any(dim=1).sum().item())
    first = int((recv != send).any(dim=1).nonzero()[0].item()) if bad else -1
    print(n, n*16/2**30, "GiB bad rows", bad, "first", first, flush=True)
dist.destroy_process_group()
