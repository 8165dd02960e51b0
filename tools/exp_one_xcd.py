#!/usr/bin/env python3
"""Timing experiment: the tagged chain of the reference's default shape (bs = 1,024, d = 64) replayed on streams that hold
one XCD's CUs, two XCDs', ... (hipExtStreamCreateWithCUMask), against all 256 CUs.  Rows written in step t by one XCD are
L2 hits for the same XCD in step t + 1; across XCDs they are a trip to the memory side.
    python tools/exp_one_xcd.py [B] [optimizer]"""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
opt = sys.argv[2] if len(sys.argv) > 2 else "Adagrad"
NP = 64


def masked_stream(keep):
    words = [0] * 8
    for i in range(256):
        if keep(i):
            words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hiplib.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, (ctypes.c_uint32 * 8)(*words))
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


hip = GloveHip(dev)
wl = synthetic.make_workload("text8_d64", device=dev, work_device=dev)
V, d = wl["V"], wl["d"]
plans = [hip.build_plan(*(wl[k][b * B:(b + 1) * B].contiguous() for k in ("row", "col", "w", "y")), V, records=True) for b in range(NP)]
ws = torch.empty(hip.lib.glove_step_workspace_bytes(B, B, d), dtype=torch.uint8, device=dev)
masks = (("all 256 CUs", None), ("CU i % 8 == 0 (32 CUs)", lambda i: i % 8 == 0), ("CU i < 32", lambda i: i < 32),
         ("CU i % 4 == 0 (64 CUs)", lambda i: i % 4 == 0), ("CU i < 64", lambda i: i < 64), ("CU i % 2 == 0 (128)", lambda i: i % 2 == 0),
         ("CU i < 128", lambda i: i < 128), ("CU i % 8 == 0 and i < 128 (16 CUs)", lambda i: i % 8 == 0 and i < 128),
         ("CU i < 16", lambda i: i < 16))
for name, keep in masks:
    t = DeviceTables(V, d, opt, seed=1)
    t.enable_tags()
    h = make_hyper(learning_rate=0.05 if opt == "Adagrad" else 0.001, batch_size=B, step_form=0)
    loss = torch.zeros(4, device=dev)
    G = hip.dense_grad_buffer(t) if opt == "Adam" else None
    st = torch.cuda.Stream(device=dev) if keep is None else masked_stream(keep)

    def chain():
        if opt == "Adam":
            hip.steps_adam(plans, t, h, G, loss, ws=ws)
        else:
            hip.steps_adagrad(plans, t, h, loss, ws=ws)
    with torch.cuda.stream(st):
        chain()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        chain()
    with torch.cuda.stream(st):
        for _ in range(3):
            g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            g.replay()
        b.record()
    torch.cuda.synchronize()
    print("%-40s %s B %d  %.2f us per step (graph)   loss %.6f" % (name, opt, B, a.elapsed_time(b) * 1e3 / (20 * NP), float(loss[0])), flush=True)
    # eager on the same stream (the launch path as well)
    with torch.cuda.stream(st):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            chain()
        b.record()
    torch.cuda.synchronize()
    print("%-40s %s B %d  %.2f us per step (one C call per 64 steps)" % (name, opt, B, a.elapsed_time(b) * 1e3 / (10 * NP)), flush=True)
    del g
