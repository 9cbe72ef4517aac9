import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544")
dev=torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from eogs2_amd.parallel import GradBucket
P=1<<20
ps=[torch.zeros(P,k,device=dev,requires_grad=True) for k in (3,5,1,3,4)]
b=GradBucket(ps, cols=[slice(0,3),slice(0,3),slice(0,1),slice(0,3),slice(0,4)])
def setg():
    for p in ps: p.grad=torch.randn_like(p)
def timeit(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
grads=[torch.randn_like(p) for p in ps]
def fresh():
    for p,g in zip(ps,grads): p.grad=g
print("pack ms", timeit(lambda:(fresh(), b.pack())))
print("allreduce ms", timeit(lambda: dist.all_reduce(b.flat)))
print("unpack ms", timeit(lambda:(fresh(), b.unpack())))
print("all ms", timeit(lambda:(fresh(), b.all_reduce())))
arena=torch.zeros(P*16,device=dev)
print("allreduce arena(64B) ms", timeit(lambda: dist.all_reduce(arena)))
dist.destroy_process_group()
