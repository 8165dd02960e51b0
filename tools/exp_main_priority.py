#!/usr/bin/env python3
"""The default mode with the steps on a HIGH-priority stream (the side stream of deals and index builds stays at normal priority),
and with the side stream at the LOWEST priority HIP offers (hipStreamCreateWithPriority), against the trainer's streams: runners in
one process, timed runs alternating (process-to-process spread is 2 - 3 %).
Usage: python tools/exp_main_priority.py [--workload zipf_v400k_d300] [--batch-size 1048576] [--steps 230]"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from trainer import synthetic  # noqa: E402
from trainer.data_utils import NonzeroStream  # noqa: E402
from trainer.hip_api import DeviceTables, GloveHip, make_hyper  # noqa: E402
from trainer.stepper import HipBackend, ReshufflingRunner  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="zipf_v400k_d300")
    ap.add_argument("--batch-size", type=int, default=1048576)
    ap.add_argument("--steps", type=int, default=230)
    ap.add_argument("--rounds", type=int, default=4)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    hip = GloveHip(dev)
    wl = synthetic.make_workload(args.workload, device=dev, work_device=dev)
    V, d, B = wl["V"], wl["d"], args.batch_size
    import ctypes
    hiplib = ctypes.CDLL("libamdhip64.so")
    least, greatest = ctypes.c_int(), ctypes.c_int()
    hiplib.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest))
    print("stream priorities: least %d, greatest %d" % (least.value, greatest.value), flush=True)
    runs = {}
    for name, main_stream in (("default stream", torch.cuda.current_stream(dev)), ("high-priority stream", torch.cuda.Stream(device=dev, priority=-1)),
                              ("default, side lowest", torch.cuda.current_stream(dev))):
        with torch.cuda.stream(main_stream):
            backend = HipBackend(dev)
            backend.hip = hip
            tables = DeviceTables(V, d, "Adagrad", device=dev, seed=1)
            backend.row_floats = tables.d
            stream = NonzeroStream({k: wl[k] for k in ("row", "col", "w", "y")}, B, V, backend, dev, seed=0, static_plans=False)
            if name.endswith("side lowest"):
                h = ctypes.c_void_p()
                assert hiplib.hipStreamCreateWithPriority(ctypes.byref(h), 1, least.value) == 0        # 1 = hipStreamNonBlocking
                stream.side = torch.cuda.ExternalStream(h.value, device=dev)
            runner = ReshufflingRunner(hip, stream, tables, make_hyper(batch_size=B, learning_rate=0.05), burst=64)
            torch.cuda.synchronize()
        runs[name] = (main_stream, runner, [])
    for rnd in range(args.rounds + 1):
        for name, (main_stream, runner, out) in runs.items():
            with torch.cuda.stream(main_stream):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                done = 0
                while done < args.steps:
                    done += runner.run(args.steps - done)
                torch.cuda.synchronize()
                if rnd:
                    out.append((time.perf_counter() - t0) * 1e6 / args.steps)
    print("%s B=%d: us per step of the runner over %d steps, alternating timed runs" % (args.workload, B, args.steps))
    for name, (_, _, out) in runs.items():
        print("  steps on the %-22s %s" % (name, "  ".join("%.1f" % x for x in out)))


if __name__ == "__main__":
    main()
