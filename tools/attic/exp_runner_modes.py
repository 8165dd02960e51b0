#!/usr/bin/env python3
"""Steps/s of the reshuffling runner (epochs dealt, index per segment on the side stream) with its steps replayed from hipGraphs
or launched eagerly: tools/exp_runner_modes.py [B] [optimizer] [segment: 0 = auto] [graphs,eager] [workload]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from trainer import synthetic  # noqa: E402
from trainer.data_utils import NonzeroStream  # noqa: E402
from trainer.hip_api import DeviceTables, GloveHip, make_hyper  # noqa: E402
from trainer.stepper import HipBackend, ReshufflingRunner  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
opt = sys.argv[2] if len(sys.argv) > 2 else "Adagrad"
segment = int(sys.argv[3]) if len(sys.argv) > 3 else 0
wl_name = sys.argv[5] if len(sys.argv) > 5 else ""
if wl_name:                                  # a bench workload (e.g. zipf_v400k_d300: the C4 shard)
    wl = synthetic.make_workload(wl_name, device="cuda:0", work_device="cuda:0")
    V, d = wl["V"], wl["d"]
    coo = {k: wl[k].cpu().numpy() for k in ("row", "col", "w", "y")}
    del wl
else:
    V, d = 10000, 64
    row, col, w, y = synthetic.text8_shaped(V=V, seed=0)
    coo = dict(row=row.numpy(), col=col.numpy(), w=w.numpy(), y=y.numpy())
backend = HipBackend("cuda:0")
hip = GloveHip("cuda:0")
hyper = make_hyper(learning_rate=0.05 if opt == "Adagrad" else 0.001, batch_size=B)
steps = max(200, min(20000, 40_000_000 // B)) if not wl_name else 96
modes = (("graphs", dict(graphs=True)), ("eager", dict(graphs=False)))
only = sys.argv[4].split(",") if len(sys.argv) > 4 else None
for name, kw in [m for m in modes if only is None or m[0].split()[0] in only]:
    stream = NonzeroStream(coo, B, V, backend, "cuda:0", seed=11, static_plans=False)
    tables = DeviceTables(V, d, opt, seed=4)
    runner = ReshufflingRunner(hip, stream, tables, hyper, burst=64, segment=segment, **kw)
    done = 0
    while done < steps // 4:
        done += runner.run(steps // 4 - done)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = 0
    while done < steps:
        done += runner.run(steps - done)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("B=%d %s segment=%d %-16s %8.0f steps/s  %.1f us/step  loss %.4f" % (B, opt, runner.S, name, steps / dt, dt / steps * 1e6, runner.read_loss()["loss"]), flush=True)
    runner.release_graphs()
