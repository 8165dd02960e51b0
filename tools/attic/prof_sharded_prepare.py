import sys, time
sys.path.insert(0, "/root/repo")
import torch
from trainer import synthetic
from trainer.hip_api import GloveHip
dev = torch.device("cuda:0")
hip = GloveHip(dev)
wl = synthetic.make_workload("zipf_v400k_d300", device=dev, work_device=dev)
V, d, B, W = wl["V"], wl["d"], 1048576, 8
row, col, w, y = (wl[k][:B].contiguous() for k in ("row", "col", "w", "y"))
def tm(name, f, n=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); print("%-28s %.3f ms" % (name, (time.perf_counter() - t0) * 1e3 / n)); return r
uc = tm("unique(col.long())", lambda: torch.unique(col.long()))
tm("unique_consecutive(sorted)", lambda: torch.unique_consecutive(torch.sort(col)[0]))
owner = uc % W
order = tm("argsort(owner, stable)", lambda: torch.argsort(owner, stable=True))
inv = torch.empty_like(order); inv[order] = torch.arange(order.numel(), device=dev)
compact = tm("searchsorted + inv", lambda: inv[torch.searchsorted(uc, col.long())].to(torch.int32))
def lut():
    t = torch.empty(V, dtype=torch.int32, device=dev)
    t[uc[order]] = torch.arange(order.numel(), device=dev, dtype=torch.int32)
    return t[col.long()]
c2 = tm("lut scatter + gather", lut)
assert torch.equal(compact, c2)
tm("bincount + tolist", lambda: torch.bincount(owner, minlength=W).tolist())
n_uc = int(uc.numel())
tm("build_plan compact", lambda: hip.build_plan(row, compact, w, y, max(n_uc, V // W), chunk_cap=32, compact=True, d=d, run_words=False))
tm("build_plan no compact", lambda: hip.build_plan(row, compact, w, y, max(n_uc, V // W), chunk_cap=32, records=True, links=False))
tm("build_plan compact, no links", lambda: hip.build_plan(row, compact, w, y, max(n_uc, V // W), chunk_cap=32, compact=True, d=d, run_words=False, links=False))
