#!/usr/bin/env python3
"""Reshuffled steps (index build of batch i+ahead on side streams while step i runs) with the streams bound to disjoint
CU sets (hipExtStreamCreateWithCUMask): does the latency-bound build hide under the bandwidth-bound step when the step's
waves cannot take every CU?   tools/exp_cu_mask.py WORKLOAD B [build CUs per XCD: 0 = no masks] [ahead]"""
import ctypes
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from trainer import synthetic  # noqa: E402
from trainer.hip_api import DeviceTables, GloveHip, Plan, make_hyper, staging_records  # noqa: E402

wl_name, B = sys.argv[1], int(sys.argv[2])
share = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ahead = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
torch.cuda.init()
torch.zeros(1, device=dev)
hiplib = ctypes.CDLL("libamdhip64.so")


def masked_stream(keep):
    """A stream whose kernels run on the CUs i with keep(i) (256 CUs: eight 32-bit words)."""
    words = [0] * 8
    for i in range(256):
        if keep(i):
            words[i // 32] |= 1 << (i % 32)
    arr = (ctypes.c_uint32 * 8)(*words)
    s = ctypes.c_void_p()
    rc = hiplib.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


hip = GloveHip(dev)
wl = synthetic.make_workload(wl_name, device=dev, work_device=dev)
V, d, nb, cap = wl["V"], wl["d"], 6, 32
batches = [tuple(wl[k][b * B:(b + 1) * B].contiguous() for k in ("row", "col", "w", "y")) for b in range(nb)]
tables = DeviceTables(V, d, "Adagrad", device=dev, seed=1)
tables.maybe_enable_twin()
hyper = make_hyper(learning_rate=0.05, batch_size=B)
loss = torch.zeros(4, device=dev)
rec = staging_records(B, V, V, d)
ring = [Plan(B, V, cap, dev, records=rec) for _ in range(ahead)]
ring_ws = [torch.empty(hip.lib.glove_plan_workspace_bytes(B, V), dtype=torch.uint8, device=dev) for _ in range(ahead)]
ws = torch.empty(hip.lib.glove_step_workspace_bytes(B, B, tables.d), dtype=torch.uint8, device=dev)
if share:
    # mask bit i = CU i // 8 of XCD i % 8 (measured: a mask that leaves an XCD without CUs is ignored): the build streams
    # get the last `share` CUs of every XCD, the step's stream the others
    main = masked_stream(lambda i: i // 8 < 32 - share)
    sides = [masked_stream(lambda i: i // 8 >= 32 - share) for _ in range(ahead)]
else:
    prio = int(sys.argv[5]) if len(sys.argv) > 5 else 0        # -1: the build streams get the higher queue priority
    main = torch.cuda.Stream(device=dev)
    sides = [torch.cuda.Stream(device=dev, priority=prio) for _ in range(ahead)]
    print("stream priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?", "build streams:", prio)


def sweep(n_steps):
    built, stepped = [None] * n_steps, [None] * n_steps
    start = torch.cuda.Event()
    start.record(main)

    def launch_build(i):
        st = sides[i % ahead]
        st.wait_event(stepped[i - ahead] if i >= ahead else start)
        with torch.cuda.stream(st):
            hip.build_plan(*batches[i % nb], V, chunk_cap=cap, into=ring[i % ahead], ws=ring_ws[i % ahead])
            built[i] = torch.cuda.Event()
            built[i].record(st)
    for i in range(min(ahead, n_steps)):
        launch_build(i)
    with torch.cuda.stream(main):
        for i in range(n_steps):
            main.wait_event(built[i])
            hip.step_adagrad(ring[i % ahead], tables, hyper, loss, ws)
            stepped[i] = torch.cuda.Event()
            stepped[i].record(main)
            if i + ahead < n_steps:
                launch_build(i + ahead)


torch.cuda.synchronize()
sweep(8)
torch.cuda.synchronize()
import time
for rnd in range(3):
    t0 = time.perf_counter()
    sweep(40)
    torch.cuda.synchronize()
    print("%s B=%d ahead=%d build CUs per XCD %d: %.1f us per step (loss %.5f)" % (
        wl_name, B, ahead, share, (time.perf_counter() - t0) * 1e6 / 40, float(loss[0])), flush=True)
