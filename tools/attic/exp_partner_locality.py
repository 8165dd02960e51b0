"""Timing experiment only (results are garbage): how fast is the fused step when every partner gather hits L2?
The partner ids in the per-chunk records are folded into [0, M)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from trainer import synthetic
from trainer.hip_api import DeviceTables, GloveHip, make_hyper

dev = torch.device("cuda:0")
hip = GloveHip(dev)
name, B = sys.argv[1], int(sys.argv[2])
wl = synthetic.make_workload(name, device=dev, work_device=dev)
V, d = wl["V"], wl["d"]
tables = DeviceTables(V, d, "Adagrad", device=dev, seed=1)
tables.enable_twin()
nb = 8
plans = [hip.build_plan(*(wl[k][b * B:(b + 1) * B].contiguous() for k in ("row", "col", "w", "y")), V, chunk_cap=32, compact=True, d=tables.d)
         for b in range(nb)]
hyper = make_hyper(learning_rate=0.05, batch_size=B, step_form=4)
ws = torch.empty(max(hip.lib.glove_step_workspace_bytes(B, p.cap_chunks, tables.d) for p in plans), dtype=torch.uint8, device=dev)
loss = torch.zeros(4, device=dev)


def timeit(tag):
    for i in range(8):
        hip.step_adagrad(plans[i % nb], tables, hyper, loss, ws)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(40):
        hip.step_adagrad(plans[i % nb], tables, hyper, loss, ws)
    b.record()
    torch.cuda.synchronize()
    print(tag, "%.1f us/step" % (a.elapsed_time(b) * 1e3 / 40), flush=True)


timeit("real partners")
for M in (65536, 8192, 1024):
    for p in plans:
        capP = (p.chunk_cap + 7) // 8 * 8
        rd = p.rec_dwords                                     # in memory: line 0 = header | block 0 | pad, then the other blocks
        for rec, n in ((p.r_crec, p.host_counts[0]), (p.c_crec, p.host_counts[2])):
            raw = rec[:n * rd].view(n, rd)
            raw[:, 4:12] %= M                                   # the partner ids of block 0
            raw[:, 32:32 + 24 * (capP // 8 - 1)].reshape(n, capP // 8 - 1, 3, 8)[:, :, 0] %= M     # ... and of the other blocks
    timeit("partners folded into [0, %d)" % M)
