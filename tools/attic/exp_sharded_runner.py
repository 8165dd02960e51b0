#!/usr/bin/env python3
"""The reshuffling runner over a fully sharded stepper with one rank and the exchange exercised (every collective issued, a
process group of one): steps per second across epoch boundaries — what preparing an epoch's batches costs beside its steps.
Usage: tools/exp_sharded_runner.py [workload] [B] [steps] [--at-the-boundary] [--sorting-prepare]"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from trainer import synthetic  # noqa: E402
from trainer.data_utils import NonzeroStream  # noqa: E402
from trainer.hip_api import DeviceTables, GloveHip, make_hyper  # noqa: E402
from trainer.stepper import HipBackend, ReshufflingRunner, ShardedStepper  # noqa: E402

wl_name = sys.argv[1] if len(sys.argv) > 1 else "zipf_v400k_d300"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
dev = torch.device("cuda:0")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", device_id=dev)
hip = GloveHip(dev)
wl = synthetic.make_workload(wl_name, device=dev, work_device=dev)
V, d = wl["V"], wl["d"]
tables = DeviceTables(V, d, "Adagrad", device=dev, seed=1)
backend = HipBackend(dev)
backend.hip = hip
backend.row_floats = tables.d
stepper = ShardedStepper(backend, tables, dict(learning_rate=0.05), B, 1, 0, dist, collectives=True, exercise_exchange=True)
# (col ids numbered owner-major — the identity with one owner —: the batches' fetch lists and indexes come straight from the
# dealt order; --sorting-prepare: the general preparation, torch.unique + the sorting index builder)
stream = NonzeroStream({k: wl[k] for k in ("row", "col", "w", "y")}, B, V, backend, dev, seed=0, static_plans=False,
                       cols_by_owner=0 if "--sorting-prepare" in sys.argv else 1)
runner = ReshufflingRunner(hip, stream, tables, stepper.hyper, stepper=stepper, graphs=False)
if "--at-the-boundary" in sys.argv:      # as before: the whole epoch prepared when it begins
    runner._prepare_ahead = lambda: None
print("batches per epoch:", runner.nb)


def go(n):
    done = 0
    while done < n:
        done += runner.run(n - done)


go(runner.nb)                       # one epoch of warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
go(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%d steps across %.1f epochs: %.3f ms per step" % (steps, steps / runner.nb, dt * 1e3 / steps))
print("loss", runner.read_loss())
dist.destroy_process_group()
