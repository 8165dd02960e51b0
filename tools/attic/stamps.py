#!/usr/bin/env python3
"""Diagnostic: per-wave wall-clock stamps (libglove_hip_diag.so, `make -C glove-tensorflow_amd/csrc diag`).
Shows when waves start, how long each phase of the dependent chain takes and when they end.
Shares of a diagnostic build only — never quote its run time (cdna guide §7)."""
import argparse
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from trainer import synthetic  # noqa: E402
from trainer.hip_api import PKG_DIR, DeviceTables, GloveHip, make_hyper  # noqa: E402


def report(name, st, nslots):
    st = st.reshape(-1, 8).astype(np.float64)
    live = st[:, 0] > 0
    st = st[live]
    if len(st) == 0:
        print("%s: no stamps in this kernel" % name)
        return
    t0 = st[:, 0].min()
    us = lambda x: (x - t0) / 100.0          # 100 MHz -> us
    q = lambda a: "p10 %.2f  p50 %.2f  p90 %.2f  max %.2f" % tuple(np.percentile(a, [10, 50, 90, 100]))
    print("%s: %d waves stamped" % (name, len(st)))
    print("   start            " + q(us(st[:, 0])))
    prev = 0
    for s in range(1, nslots):
        ok = st[:, s] > 0
        if ok.sum() == 0:
            continue
        print("   stamp %d - stamp %d " % (s, prev) + q((st[ok, s] - st[ok, prev]) / 100.0) + "   (n=%d)" % ok.sum())
        prev = s
    print("   end              " + q(us(st[:, prev])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="text8_d64")
    ap.add_argument("--batch-size", type=int, default=131072)
    ap.add_argument("--cap", type=int, default=32)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    hip = GloveHip(dev, lib_path=PKG_DIR / "lib" / "libglove_hip_diag.so")
    hip.lib.glove_debug_set_stamps.argtypes = [C.c_void_p]
    wl = synthetic.make_workload(args.workload, device=dev, work_device=dev)
    V, d, B = wl["V"], wl["d"], args.batch_size
    plans = [hip.build_plan(*(wl[k][b * B:(b + 1) * B].contiguous() for k in ("row", "col", "w", "y")), V,
                            chunk_cap=args.cap, compact=True) for b in range(2)]
    tables = DeviceTables(V, d, "Adagrad", device=dev, seed=1)
    hyper = make_hyper(learning_rate=0.05, batch_size=B)
    ws = torch.empty(hip.lib.glove_step_workspace_bytes(B, B, d), dtype=torch.uint8, device=dev)
    loss = torch.zeros(4, device=dev)
    stamps = torch.zeros(8192 * 8 * 4, dtype=torch.int64, device=dev)
    for _ in range(5):
        for p in plans:
            hip.step_adagrad(p, tables, hyper, loss, ws)
    torch.cuda.synchronize()
    assert hip.lib.glove_debug_set_stamps(stamps.data_ptr()) == 0
    for name, fn, ns in (("passes", lambda: hip.passes(plans[0], tables, hyper, ws), 6),
                         ("apply", lambda: hip.apply_adagrad(plans[0], tables, hyper, loss, ws), 7)):
        if name == "apply":
            hip.lib.glove_debug_set_stamps(None)
            hip.passes(plans[0], tables, hyper, ws)
            torch.cuda.synchronize()
            hip.lib.glove_debug_set_stamps(stamps.data_ptr())
        stamps.zero_()
        torch.cuda.synchronize()
        fn()
        torch.cuda.synchronize()
        report(name, stamps.cpu().numpy(), ns)


if __name__ == "__main__":
    main()
