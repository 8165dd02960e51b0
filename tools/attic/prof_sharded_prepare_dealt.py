import sys, time
sys.path.insert(0, "/root/repo")
import torch
from trainer import synthetic
from trainer.data_utils import NonzeroStream
from trainer.hip_api import DeviceTables, GloveHip, Pairs, PlanBlock
from trainer.stepper import HipBackend
dev = torch.device("cuda:0")
hip = GloveHip(dev)
wl = synthetic.make_workload("zipf_v400k_d300", device=dev, work_device=dev)
V, d, B, W = wl["V"], wl["d"], 1048576, 1
backend = HipBackend(dev); backend.hip = hip
stream = NonzeroStream({k: wl[k] for k in ("row", "col", "w", "y")}, B, V, backend, dev, seed=0, static_plans=False, cols_by_owner=1)
stream.reshuffle_in_place()
per = stream.col_per
rs0, cs0 = stream.epoch_sides()
torch.cuda.synchronize()
lut = torch.zeros(W * per, dtype=torch.int32, device=dev)
ws = torch.empty(max(hip.lib.glove_plan_sorted_workspace_bytes(B, 1), 256), dtype=torch.uint8, device=dev)
T = {}
def sec(name, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return r
for b in range(14):
    if b == 2:
        T.clear()        # (the first rounds warm the allocator and the kernels up)
    sl = slice(b * B, (b + 1) * B)
    uc0, run0 = sec("unique_consecutive", lambda: torch.unique_consecutive(cs0.id[sl], return_inverse=True))
    def runs():
        cid = cs0.id[sl]
        flags = torch.ones(B, dtype=torch.int32, device=dev)
        flags[1:] = cid[1:] != cid[:-1]
        return cid[flags.bool()], torch.cumsum(flags, 0, dtype=torch.int32) - 1
    uc, run = sec("flags + cumsum + select", runs)
    assert torch.equal(uc, uc0) and torch.equal(run.long(), run0)
    n_uc = int(uc.numel())
    def f():
        lut[uc.long()] = torch.arange(n_uc, dtype=torch.int32, device=dev)
        return lut[rs0.partner[sl].long()]
    compact = sec("lut scatter + gather", f)
    sec("bincount etc", lambda: (torch.bincount(uc.long() // per, minlength=W), (uc % per).to(torch.int32).contiguous(), run.to(torch.int32)))
    plan = sec("staging_plan alloc", lambda: hip.staging_plan(B, max(n_uc, V), 32, dev, records=True))
    block = sec("PlanBlock", lambda: PlanBlock([plan]))
    def side(src, ids, partner):
        p = Pairs.__new__(Pairs)
        p.n, p.id, p.partner, p.w, p.y, p._struct = B, ids.contiguous(), partner.contiguous(), src.w[sl], src.y[sl], None
        return p
    r2, c2 = side(rs0, rs0.id[sl], compact), side(cs0, run.to(torch.int32), cs0.partner[sl])
    sec("build_plans_sorted", lambda: hip.build_plans_sorted(r2, c2, 0, block, 1, max(n_uc, V), ws))
    sec("fetch+adopt", lambda: (block.fetch_counts(), torch.cuda.synchronize(), block.adopt_counts(1)))
T = {k: v for k, v in T.items()}
for k, v in T.items():
    print("%-24s %.3f ms" % (k, v / 12))
