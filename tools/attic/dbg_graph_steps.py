#!/usr/bin/env python3
"""Diagnostic (not a test): the body of test_multi_rank_steps_replayed_from_hipgraphs_with_their_rccl_collectives with a
progress line per stage and a traceback if a stage hangs."""
import faulthandler
import os
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
for p in (REPO, REPO / "tests", REPO / "oracle"):
    sys.path.insert(0, str(p))
import numpy as np
import torch
import torch.distributed as dist
import glove_ref as ref
from helpers import make_batch, tables_from_oracle, to_dev
from trainer.data_utils import NonzeroStream
from trainer.hip_api import DeviceTables, GloveHip, make_hyper
from trainer.stepper import HipBackend, ReshufflingRunner, RowShardedStepper, ShardedStepper, Stepper

faulthandler.dump_traceback_later(int(os.environ.get("DBG_TIMEOUT", "90")), exit=True)


def say(*a):
    print(*a, flush=True)


os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29573", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
hip = GloveHip("cuda:0")
B, V, d, nb, rounds = 6000, 700, 64, 3, 4
backend = HipBackend("cuda:0")
t = ref.Tables(V, d, "Adagrad", dtype=np.float32, seed=4).astype(np.float64)
kw = dict(l2_reg=0.01, reg_mult=2.0, learning_rate=0.05)
batches = [to_dev(*make_batch(40 + s, B, V)) for s in range(nb)]
plans = [backend.build_plan(*bt, V, 0).compact(hip.lib, d) for bt in batches]


def make(form):
    tabs = tables_from_oracle(t, DeviceTables)
    if form in ("dp rows", "dp dense"):
        st = Stepper(backend, tabs, kw, B, world=1, dist=dist, exchange=form.split()[1], collectives=True)
        st.prepare(plans)
        return tabs, st, plans
    if form.startswith("row-sharded"):
        st = RowShardedStepper(backend, tabs, kw, B, 1, dist, exchange=form.split()[1], collectives=True)
        st.prepare(plans)
        return tabs, st, plans
    st = ShardedStepper(backend, tabs, kw, B, 1, 0, dist, collectives=True, exercise_exchange=True)
    return tabs, st, [st.add_batch(*bt) for bt in batches]


forms = sys.argv[1:] or ["dp dense", "dp rows", "row-sharded rows", "row-sharded dense", "both tables sharded", "runners"]
for form in forms:
    if form == "runners":
        continue
    say("form", form)
    tb, sb, ib = make(form)
    sb.enable_graphs(after=1)
    for rnd in range(rounds):
        for k in range(nb):
            say("  round", rnd, "batch", k)
            sb.step(ib[k])
            torch.cuda.synchronize()
    say("form", form, "ok", sb.read_loss())
if "runners" in forms:
    Br = 1000
    coo = {k: v for k, v in zip(("row", "col", "w", "y"), make_batch(7, 5 * Br + 123, V))}
    for mode in ("single", "dp eager", "dp graphs", "row-sharded graphs"):
        say("runner", mode)
        tabs = tables_from_oracle(t, DeviceTables)
        stream = NonzeroStream(coo, Br, V, backend, "cuda:0", seed=3, static_plans=False)
        if mode == "single":
            runner = ReshufflingRunner(hip, stream, tabs, make_hyper(batch_size=Br, **kw), burst=4, segment=2)
        else:
            cls = RowShardedStepper if mode.startswith("row") else Stepper
            st = cls(backend, tabs, kw, Br, 1, dist, exchange="dense" if mode.startswith("dp") else "rows", collectives=True)
            st.prepare(batch_size=Br)
            runner = ReshufflingRunner(hip, stream, tabs, st.hyper, burst=4, segment=2, stepper=st, graphs=mode.endswith("graphs"))
        say("  constructed")
        done = 0
        while done < 23:
            n = runner.run(23 - done)
            done += n
            torch.cuda.synchronize()
            say("  ran", n, "->", done)
        say("runner", mode, "ok", runner.read_loss())
dist.destroy_process_group()
say("all ok")
