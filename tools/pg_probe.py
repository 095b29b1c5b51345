import os, torch, torch.distributed as dist, traceback
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.ones(4, device=dev); dist.all_reduce(x); torch.cuda.synchronize()
try:
    pg = dist.new_group(ranks=[0], backend="nccl", device_id=dev)
    be = pg._get_backend(dev)
    print("backend", type(be), "supports_splitting", be.supports_splitting)
    print("initialized after new_group:", be._is_initialized())
    if not be._is_initialized():
        try:
            be.eager_connect_single_device(dev); print("eager_connect ok ->", be._is_initialized())
        except Exception as e:
            print("eager_connect raised", type(e).__name__, e)
    # capture an all_reduce on it
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); y = torch.ones(8, device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        dist.all_reduce(y, group=pg)
    g.replay(); torch.cuda.synchronize(); print("captured all_reduce on the fresh group ok", y[:2].tolist())
except Exception:
    traceback.print_exc()
dist.destroy_process_group()
