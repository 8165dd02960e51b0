#!/usr/bin/env python3
"""Does hipExtStreamCreateWithCUMask bind kernels here?  The C4 step on a stream holding 1/8, 1/2, 7/8 and all of the CUs."""
import ctypes
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from trainer import synthetic  # noqa: E402
from trainer.hip_api import DeviceTables, GloveHip, make_hyper  # noqa: E402

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
hiplib = ctypes.CDLL("libamdhip64.so")


def masked_stream(keep):
    words = [0] * 8
    for i in range(256):
        if keep(i):
            words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hiplib.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, (ctypes.c_uint32 * 8)(*words))
    assert rc == 0, rc
    got = (ctypes.c_uint32 * 8)()
    rc2 = hiplib.hipExtStreamGetCUMask(s, 8, got)
    print("mask set", ["%08x" % w for w in words], "get rc", rc2, ["%08x" % w for w in got])
    return torch.cuda.ExternalStream(s.value, device=dev)


hip = GloveHip(dev)
wl = synthetic.make_workload("zipf_v400k_d300", device=dev, work_device=dev)
V, d, B = wl["V"], wl["d"], 1048576
plan = hip.build_plan(*(wl[k][:B].contiguous() for k in ("row", "col", "w", "y")), V, chunk_cap=32, compact=True, d=d)
tables = DeviceTables(V, d, "Adagrad", device=dev, seed=1)
tables.maybe_enable_twin()
hyper = make_hyper(learning_rate=0.05, batch_size=B)
loss = torch.zeros(4, device=dev)
ws = torch.empty(hip.lib.glove_step_workspace_bytes(B, plan.cap_chunks, tables.d), dtype=torch.uint8, device=dev)
for name, keep in (("all", None), ("7/8", lambda i: i % 8 >= 1), ("1/2", lambda i: i % 2 == 0), ("1/8", lambda i: i % 8 == 0),
                   ("first 32", lambda i: i < 32)):
    st = torch.cuda.Stream(device=dev) if keep is None else masked_stream(keep)
    with torch.cuda.stream(st):
        for i in range(3):
            hip.step_adagrad(plan, tables, hyper, loss, ws)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(10):
            hip.step_adagrad(plan, tables, hyper, loss, ws)
        b.record()
    torch.cuda.synchronize()
    print("CUs %-8s %.1f us per step" % (name, a.elapsed_time(b) * 100), flush=True)
