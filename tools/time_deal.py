import sys, torch
sys.path.insert(0, "/root/repo")
from trainer.hip_api import GloveHip, Pairs
dev = torch.device("cuda:0")
hip = GloveHip(dev)
n, V, B = 25_000_000, 400_000, 1048576
g = torch.Generator(device=dev); g.manual_seed(0)
row = torch.randint(0, V, (n,), device=dev, dtype=torch.int32, generator=g)
col = torch.randint(0, V, (n,), device=dev, dtype=torch.int32, generator=g)
w = torch.rand(n, device=dev, generator=g); y = torch.rand(n, device=dev, generator=g)
m = hip.build_masters(row, col, w, y, V)
rs, cs = Pairs(n, dev), Pairs(n, dev)
ws = hip.deal_workspace(n, B, dev)
for k in range(3): hip.deal_epoch(m, B, 1234 + k, rs, cs, ws)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for k in range(10): hip.deal_epoch(m, B, 99 + k, rs, cs, ws)
b.record(); torch.cuda.synchronize()
print("deal of %d pairs into batches of %d: %.1f us" % (n, B, a.elapsed_time(b) * 100))
