"""RCCL on one GPU: latency of a barrier and of a one-element all-reduce in a process group of ONE rank, and whether an
initialised communicator slows ordinary launches (it does not; what it costs a handle is hardware-queue sharing)."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.zeros(1 << 20, device=dev)
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dist.barrier(); torch.cuda.synchronize(); print("barrier %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3))
t = torch.tensor([1.0], dtype=torch.float64, device=dev)
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dist.all_reduce(t, op=dist.ReduceOp.MAX); torch.cuda.synchronize(); print("all_reduce %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3))
# does an initialised communicator slow ordinary kernels down?
def work():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): x.add_(1.0)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
print("200 small kernels: %.3f ms, %.3f ms" % (work(), work()))
dist.destroy_process_group()
