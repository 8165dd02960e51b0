#!/usr/bin/env python3
"""Diagnostic (not a test): which RCCL collectives survive hipGraph capture + replay in this torch / ROCm, one rank."""
import faulthandler
import os
import sys
import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(40, exit=True)          # a hang ends with a traceback instead of a silent box
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
dev = torch.device("cuda:0")


def say(*a):
    print(*a, flush=True)


def stage(name, fn, check):
    say("stage", name, ": eager")
    fn()
    torch.cuda.synchronize()
    say("stage", name, ": warm on a side stream")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    say("stage", name, ": capture")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    say("stage", name, ": replay")
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    say("stage", name, ": ok", check())


x = torch.ones(1000, device=dev)
stage("all_reduce", lambda: dist.all_reduce(x), lambda: float(x.sum()))
send, recv = torch.arange(4096, device=dev, dtype=torch.float32), torch.zeros(4096, device=dev)
stage("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(recv, send), lambda: float(recv.sum()))
a, b = torch.arange(999, device=dev, dtype=torch.float32), torch.zeros(999, device=dev)
stage("all_to_all_single split", lambda: dist.all_to_all_single(b, a, [999], [999]), lambda: float(b.sum()))


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from trainer.stepper import SideCollective          # noqa: E402

side_coll = SideCollective(dev)


def beside():
    side_coll.start(lambda a_: dist.all_to_all_single(b, a, [999], [999], async_op=a_))
    x.mul_(1.0)                      # work on the compute stream meanwhile
    side_coll.wait()
    b.add_(1.0)


stage("all_to_all beside the compute stream (side stream, events)", beside, lambda: float(b.sum()))


def gather_beside():
    side_coll.start(lambda a_: dist.all_gather_into_tensor(recv, send, async_op=a_))
    x.mul_(1.0)
    side_coll.wait()
    recv.add_(1.0)


stage("all_gather beside the compute stream", gather_beside, lambda: float(recv.sum()))
if "--async" in sys.argv:           # the form that crashes under capture (kept for the record)
    def async_pair():
        w = dist.all_to_all_single(b, a, [999], [999], async_op=True)
        x.mul_(1.0)
        w.wait()
        b.add_(1.0)
    stage("async_op all_to_all + wait", async_pair, lambda: float(b.sum()))
dist.destroy_process_group()
say("all stages ok")
