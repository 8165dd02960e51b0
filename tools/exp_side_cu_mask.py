#!/usr/bin/env python3
"""The default mode's side stream (epoch deals and index builds beside the steps) on a stream that holds a share of the CUs
(hipExtStreamCreateWithCUMask, the same share of every XCD): do the steps — bound by bandwidth, hiding latency with every wave
the chip can hold — lose less when the side work keeps to a few CUs and takes its time?

Usage: python tools/exp_side_cu_mask.py [--workload zipf_v2m_d128] [--batch-size 1048576] [--steps 200]"""
import argparse
import ctypes
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from trainer import synthetic  # noqa: E402
from trainer.data_utils import NonzeroStream  # noqa: E402
from trainer.hip_api import DeviceTables, GloveHip, make_hyper  # noqa: E402
from trainer.stepper import HipBackend, ReshufflingRunner  # noqa: E402


def masked_stream(hiplib, dev, share):
    """1 / share of the CUs of every XCD, whichever way the 256 mask bits number them (XCD-major or interleaved)"""
    words = [0] * 8
    for i in range(256):
        if ((i % 8) + (i // 8)) % share == 0:
            words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p()
    rc = hiplib.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, (ctypes.c_uint32 * 8)(*words))
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="zipf_v2m_d128")
    ap.add_argument("--batch-size", type=int, default=1048576)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--shares", default="0,1,2,4,8")
    ap.add_argument("--lib", default="", help="another build of libglove_hip.so")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    hiplib = ctypes.CDLL("libamdhip64.so")
    hip = GloveHip(dev, lib_path=args.lib, any_abi=True) if args.lib else GloveHip(dev)
    print('lib:', args.lib or 'in-tree', flush=True)
    wl = synthetic.make_workload(args.workload, device=dev, work_device=dev)
    V, d, B = wl["V"], wl["d"], args.batch_size
    print("%s B=%d: us per step of the runner over %d steps (two timed runs each)" % (args.workload, B, args.steps), flush=True)
    for share in [int(x) for x in args.shares.split(",")]:
        backend = HipBackend(dev)
        backend.hip = hip
        tables = DeviceTables(V, d, "Adagrad", device=dev, seed=1)
        backend.row_floats = tables.d
        stream = NonzeroStream({k: wl[k] for k in ("row", "col", "w", "y")}, B, V, backend, dev, seed=0, static_plans=False)
        if share > 0:
            stream.side = masked_stream(hiplib, dev, share)
        elif share < 0:                 # no second stream at all: deals and index builds in line with the steps
            stream.side = torch.cuda.current_stream(dev)
        hyper = make_hyper(batch_size=B, learning_rate=0.05)
        runner = ReshufflingRunner(hip, stream, tables, hyper, burst=64)
        out = []
        for rnd in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            done = 0
            while done < args.steps:
                done += runner.run(args.steps - done)
            torch.cuda.synchronize()
            if rnd:
                out.append((time.perf_counter() - t0) * 1e6 / args.steps)
        print("  side stream on %-22s %s" % ("all CUs (torch stream)" if not share else "the steps' own stream" if share < 0 else "1/%d of the CUs" % share,
                                            "  ".join("%.1f" % x for x in out)), flush=True)
        del runner, stream, tables
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
