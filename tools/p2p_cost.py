"""host cost of torch.distributed.batch_isend_irecv (RCCL, one-rank group, sends to self)"""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
plane = 512 * 512
buf = torch.zeros(plane * 66, dtype=torch.float64, device="cuda")
first, last, glo, ghi = buf[plane:2*plane], buf[64*plane:65*plane], buf[:plane], buf[65*plane:]
st = torch.zeros(16, dtype=torch.float64, device="cuda")
def halo():
    ops = [dist.P2POp(dist.isend, first, 0), dist.P2POp(dist.irecv, glo, 0), dist.P2POp(dist.isend, last, 0), dist.P2POp(dist.irecv, ghi, 0)]
    for r in dist.batch_isend_irecv(ops): r.wait()
for _ in range(5): halo(); dist.all_reduce(st[3:4])
torch.cuda.synchronize()
for name, fn in (("batch_isend_irecv(4 ops)+wait", halo), ("all_reduce(1 double)", lambda: dist.all_reduce(st[3:4]))):
    t0 = time.perf_counter()
    for _ in range(200): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: host enqueue {1e6*(t1-t0)/200:.1f} us per call, with device drain {1e6*(t2-t0)/200:.1f} us", flush=True)
dist.destroy_process_group()
