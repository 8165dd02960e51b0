"""The epoch deal alone: us per epoch for (n pairs, batch size), one line per library given (default: the in-tree one).
usage: python tools/time_deal.py [lib.so ...]"""
import sys, torch
sys.path.insert(0, "/root/repo")
from trainer.hip_api import GloveHip, Pairs
dev = torch.device("cuda:0")
libs = sys.argv[1:] or [None]
for n, V, B in ((25_000_000, 400_000, 1048576), (1_187_978, 10_000, 131072), (1_187_978, 10_000, 1024)):
    g = torch.Generator(device=dev); g.manual_seed(0)
    row = torch.randint(0, V, (n,), device=dev, dtype=torch.int32, generator=g)
    col = torch.randint(0, V, (n,), device=dev, dtype=torch.int32, generator=g)
    w = torch.rand(n, device=dev, generator=g); y = torch.rand(n, device=dev, generator=g)
    outs = []
    for lib in libs:
        hip = GloveHip(dev, lib_path=lib, any_abi=True)
        m = hip.build_masters(row, col, w, y, V)
        rs, cs = Pairs(n, dev), Pairs(n, dev)
        ws = hip.deal_workspace(n, B, dev)
        for k in range(3): hip.deal_epoch(m, B, 1234 + k, rs, cs, ws)
        best = 1e9
        for rep in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for k in range(10): hip.deal_epoch(m, B, 99 + k, rs, cs, ws)
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 100)
        outs.append((rs.id.clone(), rs.partner.clone(), cs.id.clone(), cs.partner.clone()))
        print("deal of %d pairs into batches of %d: %.1f us per epoch  (%s)" % (n, B, best, lib or "in-tree"), flush=True)
    for o in outs[1:]:
        print("   same epoch as the first library's:", all(bool((x == y_).all()) for x, y_ in zip(o, outs[0])))
