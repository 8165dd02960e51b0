#!/usr/bin/env python3
"""bench.py — co-occurrence nonzeros/sec of the GloVe training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload text8_d64] [--batch-size B]

A "step" is one optimizer step over one batch of B synthetic co-occurrence nonzeros that are already resident in HBM.
On one GPU the stream is stepped the way the trainer steps it by default (`--epoch-shuffle full`, what the reference's
make_csv_dataset(shuffle=True, num_epochs=None) does): every epoch is dealt anew from the sorted master orders and the
dedup index of every batch is numbered inside the timed region (trainer.stepper.ReshufflingRunner: epoch deals and index
builds on a side stream, steps replayed from hipGraphs).  `--static-index` times the other product mode instead
(`--epoch-shuffle static`: one permutation, the index of every batch built once at load, outside the clock).

N = 1: the sparse Adagrad step (fused forward+gradient pass kernel(s) + apply kernel; glove_step_adagrad_f32 picks the
form).  The headline is the workload BASELINE.json's metric is quoted on at 1, 2, 4 and 8 GPUs — config 4, synthetic
Zipf V = 400 k, d = 300 — as the one-GPU shard of its 200 M nonzeros (25 M, batches of 1 M): the configuration where
"achieved HBM GB/s vs peak" is about HBM (text8's tables live in the caches).  Unless `--single` is given the run also
times the other configurations (each a `[bench-config]` line on stderr): text8 d = 64 (BASELINE configs[1]: static index, index rebuilt every step, the
reference's batch size, Keras-legacy Adam), V = 50 k at d = 300 and V = 2 M at d = 128, each with its own roofline —
as far as the wall-clock budget (`--budget-seconds`) goes; what did not fit is named in `configs_skipped`.  `roofline.frac`
of every entry = algorithmic bytes per step / ms_per_step / 8 TB/s (the whole step as timed, index work included).

N > 1: `python bench.py --gpus N` starts N ranks itself (torch.distributed.run as a child process; under a launcher it
is a rank).  Every rank owns its own shard of nonzeros (global batch = N * B, weak scaling).  The headline is config 4
with both tables sharded (touched col rows fetched from / returned to their owners by all-to-all: the exchange follows
the batch, not the vocabulary) in the same mode as the one-GPU headline: every rank deals its shard anew every epoch,
fetch lists and indexes prepared beside the steps (what `python -m trainer.estimator --row-sharded --shard-cols` runs);
the other configurations (stderr lines, bench_configs.json): the same with a static index, config 4 data parallel as
BASELINE.json words it (dense-gradient all-reduce or touched-rows all-gather, whichever is the shorter payload), config 5
sharded in both modes.

Rank 0 prints ONE JSON line — the headline alone, a few KB, the last thing on stdout; every other configuration is a
`[bench-config] {...}` line on stderr and an entry of bench_configs.json (gpurun_out/ when it exists).  `roofline` is measured live with HIP events on the
launch stream in a second, instrumented pass over the same batches; `cpu_baseline` times the C port of the oracle
(oracle/glove_ref.c: all cores with OpenMP, and one core) on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import subprocess
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
T_START = time.perf_counter()
sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md "Chip-level parameters")
HBM_STREAM_GBS = 6290.0        # measured float4-copy ceiling of the same guide (BASELINE.md §3 asks for both)
WORKLOADS = ["text8_d64", "text8_v50k_d300", "zipf_v400k_d300", "zipf_v2m_d128"]
DEFAULT_BATCH = {"text8_d64": 131072, "text8_v50k_d300": 131072, "zipf_v400k_d300": 1048576, "zipf_v2m_d128": 1048576}
DATA_NOTE = {"text8_d64": "synthetic (text8-shaped Poisson model of the reference's data prep: 17 M-token Zipf corpus, "
                          "window 5, count >= 10; no text8 on disk)",
             "text8_v50k_d300": "synthetic (Zipf(1.0) ids over V = 50,000, SURVEY.md §8d generator)",
             "zipf_v400k_d300": "synthetic (Zipf(1.0) ids over V = 400,000, 25 M nonzeros = one GPU's shard of config 4)",
             "zipf_v2m_d128": "synthetic (Zipf(1.0) ids over V = 2,000,000, 25 M nonzeros = one GPU's shard of config 5)"}


def log(msg: str):
    """Progress on stderr (rank 0): a long run says where it is — and, should a run die, how far it got."""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %7.1f s] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="zipf_v400k_d300", choices=WORKLOADS)
    ap.add_argument("--row-sharded", action="store_true",
                    help="BASELINE config 5: both tables sharded by id %% N, nonzeros routed to row owners (all-to-all at "
                         "load), touched col rows fetched from / returned to their owners every step")
    ap.add_argument("--cols-replicated", action="store_true",
                    help="with --row-sharded: keep the col table replicated (RowShardedStepper: col side data parallel)")
    ap.add_argument("--batch-size", type=int, default=0, help="0 = the workload's own (DEFAULT_BATCH)")
    ap.add_argument("--chunk-cap", type=int, default=0, help="0 = auto (hip_api.auto_chunk_cap)")
    ap.add_argument("--force-dense", action="store_true", help="run the data-parallel form also on one GPU")
    ap.add_argument("--exchange", default="auto", choices=["auto", "dense", "rows"],
                    help="multi-rank gradient exchange: dense all-reduce, all-gather of touched-row lists, or the shorter payload")
    ap.add_argument("--optimizer", default="Adagrad", choices=["Adagrad", "Adam"],
                    help="Adam = Keras-legacy dense-decay Adam (config 1 of BASELINE.json), single GPU only")
    ap.add_argument("--learning-rate", type=float, default=0.05)
    ap.add_argument("--collectives", action="store_true",
                    help="with --row-sharded / --force-dense on one GPU: issue every collective through RCCL although there is one rank")
    ap.add_argument("--exercise-exchange", action="store_true",
                    help="with --row-sharded on one GPU: the whole serve / fetch / push / owner-apply sequence although every row is local")
    ap.add_argument("--step-form", type=int, default=0, help="glove_hyper.step_form: 0 auto, 1 two launches, 2 fused one pass, 3 fused three launches, 4 fused on a twinned row table, 5 tagged step (latency-bound regime)")
    ap.add_argument("--static-index", action="store_true",
                    help="one GPU: the trainer's --epoch-shuffle static (the index of every resident batch built at load, "
                         "outside the clock) instead of its default, epochs dealt and indexed inside the timed region")
    ap.add_argument("--index-segment", type=int, default=0,
                    help="dealt epochs: consecutive batches whose index one set of launches builds (0 = the runner's choice)")
    ap.add_argument("--no-graph", action="store_true", help="launch every step from Python instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-time budget of all CPU legs together")
    ap.add_argument("--budget-seconds", type=float, default=300.0,
                    help="wall-clock budget of the whole run: a configs[] entry is only started while the time spent so "
                         "far leaves room for it; the ones left out are listed in configs_skipped")
    ap.add_argument("--max-batches", type=int, default=16, help="resident batches to cycle through")
    ap.add_argument("--single", action="store_true", help="only the named workload (no configs[] array)")
    ap.add_argument("--with-configs", action="store_true", help="the configs[] array also under --rehearse-on-one-gpu")
    ap.add_argument("--min-timed-ms", type=float, default=20.0,
                    help="the timed region of --steps steps is repeated until this much time has been measured; the "
                         "median repeat is reported")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="control-flow rehearsal of the multi-rank path on a one-GPU box: every rank uses cuda:0 and the "
                         "collectives go through gloo; its numbers mean nothing")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def launcher_argv(n: int, argv: list, port: int) -> list:
    """The command that runs this file as n ranks of ONE node (the form the task contract names)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)


def free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv: list) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process (this process has not
    touched the GPU and never will), relay rank 0's JSON line, return the child's exit code."""
    proc = subprocess.Popen(launcher_argv(n, argv, free_port()), stdout=subprocess.PIPE, text=True, cwd=str(REPO))
    lines = 0
    for ln in proc.stdout:              # relayed as it comes: rank 0's line is out before the side configurations start
        ln = ln.rstrip("\n")
        if ln.startswith('{"metric"'):
            lines += 1
            print(ln, flush=True)
        else:
            print(ln, file=sys.stderr, flush=True)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print("expected one JSON line from rank 0, got %d" % lines, file=sys.stderr)
        return 1
    return rc


# ------------------------------------------------------------------------------------------------ figures
def algorithmic_bytes_adam(B, V, d):
    """Keras-legacy Adam sweeps W, m, v (read + write) of both tables and bias vectors
    every step, independent of the batch, plus the nonzero stream."""
    return 16 * B + 2 * 24 * V * (d + 1)


def algorithmic_bytes(B, d, u_row, u_col):
    """16 B of (row, col, weight, value) per nonzero + read W, read A, write W,
    write A for every distinct touched row and its bias."""
    return 16 * B + 16 * (d + 1) * (u_row + u_col)


# (workload, batch, index mode) -> the committed PMC summary of exactly that bench command (tools/pmc_traffic.py:
# separate `rocprofv3 --pmc` passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Looked up by key,
# never by name matching; a configuration without a row reports traffic null and says so in `traffic_source`.
TRAFFIC_PROFILES = {
    ("zipf_v400k_d300", 1048576, "dealt"): "r05_c4_v400k_d300_b1m_index_rebuilt_traffic.json",
    ("zipf_v400k_d300", 1048576, "static"): "r05_c4_v400k_d300_b1m_static_index_traffic.json",
    ("zipf_v2m_d128", 1048576, "dealt"): "r05_c5_v2m_d128_b1m_index_rebuilt_traffic.json",
    ("zipf_v2m_d128", 1048576, "static"): "r05_c5_v2m_d128_b1m_static_index_traffic.json",
    ("text8_v50k_d300", 131072, "dealt"): "r05_c3_v50k_d300_b131072_index_rebuilt_traffic.json",
    ("text8_v50k_d300", 131072, "static"): "r05_c3_v50k_d300_b131072_static_index_traffic.json",
    ("text8_d64", 131072, "dealt"): "r05_text8_d64_b131072_index_rebuilt_traffic.json",
    ("text8_d64", 131072, "static"): "r03_text8_d64_b131072_traffic.json",
    ("text8_d64", 1024, "dealt"): "r05_text8_d64_b1024_index_rebuilt_traffic.json",
    ("text8_d64", 1024, "static"): "r03_text8_d64_b1024_traffic.json",
}


def measured_traffic(workload, B, index="static"):
    """HBM-side bytes per step of this configuration from its committed PMC summary (TRAFFIC_PROFILES), with the file's
    name — or (None, the reason there is none).  The counters need their own rocprofv3 passes, so this is the profile of
    the same command, not a measurement of the run that prints it."""
    name = TRAFFIC_PROFILES.get((workload, int(B), index))
    if name is None:
        return None, "no PMC profile committed for (%s, B=%d, %s)" % (workload, B, index)
    try:
        j = json.load(open(REPO / "profiles" / name))
        return float(j["traffic_bytes_per_step"]), "profiles/" + name
    except (OSError, ValueError, KeyError) as exc:
        return None, "profiles/%s unreadable (%s)" % (name, type(exc).__name__)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def usable_cores() -> int:
    """Cores this process may actually use: the affinity mask, cut by the cgroup's CPU quota (a GPU box gives a
    one-GPU job a share of the host's cores; more OpenMP threads than that only queue behind each other)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def cpu_leg(wl, B, optimizer, lr, seconds, threads):
    """oracle/glove_ref.c on a bounded sample of the SAME batches: `threads` = 1 is the scalar port, otherwise the
    OpenMP form (ids grouped per batch at load like the GPU's resident index, outside the clock)."""
    sys.path.insert(0, str(REPO / "oracle"))
    import numpy as np
    import glove_ref as ref
    import glove_ref_c
    V, d = wl["V"], wl["d"]
    t = ref.Tables(V, d, optimizer, dtype=np.float32, seed=1)
    port = glove_ref_c.CPort(t, B)
    hp = ref.Hyper(learning_rate=lr)
    nb = max(1, min(4, wl["row"].numel() // B))
    host = {k: wl[k][:nb * B].cpu().numpy() for k in ("row", "col", "w", "y")}
    bt = [tuple(host[k][b * B:(b + 1) * B] for k in ("row", "col", "w", "y")) for b in range(nb)]
    mt = threads != 1
    idx = [glove_ref_c.BatchIndex(b[0], b[1], d) for b in bt] if mt else None
    n_threads = min(port.max_threads(), usable_cores()) if threads == 0 else threads

    def one(i):
        if mt:
            port.step_mt(idx[i % nb], *bt[i % nb], hp, threads=n_threads)
        else:
            port.step(*bt[i % nb], hp)
    one(0)                                                   # warm
    n, t0 = 0, time.perf_counter()
    while True:
        one(n)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 5000:
            break
    return {"value": n * B / el, "unit": "nonzeros/s", "steps_per_s": n / el, "cores": n_threads if mt else 1, "kind": "port",
            "sample": "%d %s steps of %d nonzeros in %.1f s (same batches; oracle/glove_ref.c, gcc -O2 fp32%s)" % (
                n, optimizer, B, el, ", OpenMP" if mt else ", scalar")}


def cpu_baseline(ctx, head_wl, head_B, lr, seconds):
    """CPU-2 = Adagrad at the GPU run's batch size and (V, d), all cores; CPU-1 = config 1
    (Adam dense-decay, batch 1,024, text8 shape), all cores; plus the one-core scalar figures for scale."""
    legs = {}
    budget = max(seconds, 0.9)
    try:
        legs["adagrad_at_gpu_batch"] = cpu_leg(head_wl, head_B, "Adagrad", lr, budget * 0.4, 0)
        legs["adagrad_at_gpu_batch_one_core"] = cpu_leg(head_wl, head_B, "Adagrad", lr, budget * 0.2, 1)
        c1 = ctx.workload("text8_d64")
        legs["c1_adam_bs1024"] = cpu_leg(c1, 1024, "Adam", 0.001, budget * 0.3, 0)
        legs["c1_adam_bs1024_one_core"] = cpu_leg(c1, 1024, "Adam", 0.001, budget * 0.1, 1)
    except Exception as exc:                     # the baseline is a side measurement: never lose the bench line to it
        legs["error"] = "%s: %s" % (type(exc).__name__, exc)
    main = legs.get("adagrad_at_gpu_batch", {"value": None, "unit": "nonzeros/s", "cores": 0, "kind": "port", "sample": "failed"})
    out = dict(main)
    out.update(host_cpus=os.cpu_count(), usable_cores=usable_cores(), cpu_model=cpu_model(), legs=legs,
               label="CPU restatement of yxtay/glove-tensorflow estimator step (TF 2.11 unavailable offline)")
    return out


# ------------------------------------------------------------------------------------------------ one configuration
class Ctx:
    def __init__(self, args, world, rank, dev, dist, hip):
        self.args, self.world, self.rank, self.dev, self.dist, self.hip = args, world, rank, dev, dist, hip
        self._wl = {}

    def workload(self, name):
        from trainer import synthetic
        if name not in self._wl:
            self._wl = {}                        # one resident workload at a time
            self._wl[name] = synthetic.make_workload(name, seed=self.rank, device=self.dev, work_device=self.dev)
        return self._wl[name]

    def barrier(self):
        if self.dist is not None and self.world > 1:
            self.dist.barrier()
        torch.cuda.synchronize()


def process_group_facts(ctx) -> dict:
    """What the ranks themselves see of the job (all-gathered), for the JSON line: the world size and backend
    torch.distributed reports, the RCCL version, and every rank's device — evidence that N ranks on N GPUs took part."""
    dist, dev = ctx.dist, ctx.dev
    if dist is None or not dist.is_initialized():
        return {"world_size": 1, "backend": None}
    import socket
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": dist.get_rank(), "cuda_device": dev.index, "device_name": props.name,
            "pci_bus_id": getattr(props, "pci_bus_id", None), "host": socket.gethostname(), "pid": os.getpid(),
            "visible_devices": torch.cuda.device_count()}
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, mine)
    try:
        rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:                       # a build without the binding
        rccl = None
    return {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": rccl,
            "distinct_devices": len({(r["host"], r["pci_bus_id"] if r["pci_bus_id"] is not None else r["cuda_device"]) for r in everyone}),
            "ranks": everyone}


COLLECTIVE_PHASES = {"all_reduce": "all_reduce", "all_gather": "all_gather", "fetch_all_to_all": "all_to_all",
                     "push_all_to_all": "all_to_all", "loss_tail": "all_reduce"}


def collectives_breakdown(kern: dict) -> dict:
    """The step's collectives apart (north_star: "with all-reduce time broken out"): milliseconds per step of every phase that
    is (or ends in) a collective, each timed to its end on its own; the overlapped ones run beside kernels in the real step."""
    per = {name: us / 1e3 for name, us in kern.items() if name in COLLECTIVE_PHASES}
    out = {"phases_ms_per_step": per}
    for kind in ("all_reduce", "all_gather", "all_to_all"):
        out[kind + "_ms"] = sum(ms for name, ms in per.items() if COLLECTIVE_PHASES[name] == kind)
    return out


def steps_per_graph(steps: int) -> int:
    """Largest divisor of `steps` that is at most 64: the timed region is then a whole number of graph replays."""
    for k in range(min(64, steps), 0, -1):
        if steps % k == 0:
            return k
    return 1


def timed_region(ctx, run, steps, warmup, min_timed_ms):
    """The contract's timed region around run(n_steps): warm-up, then exactly `steps` steps between barrier + synchronize on
    both sides, MAX over ranks; repeated at least three times and until at least min_timed_ms have been measured (a single
    transient cannot swing the figure), median reported.  Returns (median seconds, all)."""
    dev, world, dist = ctx.dev, ctx.world, ctx.dist
    torch.cuda.synchronize()
    log("  warm-up steps")
    run(warmup)
    # the first tens of milliseconds after the load phase run slow whatever the kernels are (B = 1 M at text8 scale:
    # 158 us/step in a first region of 200 steps, 47 in every later one; a graph's first replay also carries its
    # upload): the warm-up goes on, untimed, until the loop has run for 50 ms
    ctx.barrier()
    t_warm = time.perf_counter()
    while True:
        run(steps)
        ctx.barrier()
        warm = torch.tensor([time.perf_counter() - t_warm], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(warm, op=dist.ReduceOp.MAX)     # every rank takes the same number of rounds
        if float(warm.item()) >= 0.05:
            break
    log("  timed region")
    elapsed_all, total = [], 0.0
    while True:
        ctx.barrier()
        t0 = time.perf_counter()
        run(steps)
        ctx.barrier()
        el = time.perf_counter() - t0
        stop = torch.tensor([el, 0.0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(stop, op=dist.ReduceOp.MAX)
            el = float(stop[0].item())
        elapsed_all.append(el)
        total += el
        if (total * 1e3 >= min_timed_ms and len(elapsed_all) >= 3) or len(elapsed_all) >= 50:   # every rank sees the same MAX: same decision
            break
    return statistics.median(elapsed_all), elapsed_all


def event_us(fn, reps=3, stream=None):
    """Median GPU time of fn() in microseconds, HIP events on the stream the work is launched on."""
    st = stream or torch.cuda.current_stream()
    spans = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            a.record(st)
            fn()
            b.record(st)
        torch.cuda.synchronize()
        spans.append(a.elapsed_time(b) * 1e3)
    return sorted(spans)[len(spans) // 2]


def run_dealt_multi(ctx, workload, B, mode, steps=200, warmup=20, lr=0.05, chunk_cap=0, step_form=0, exchange="auto",
                    no_graph=False, min_timed_ms=20.0, segment=0):
    """The multi-rank forms in the trainer's default mode (--epoch-shuffle full), set up as trainer.estimator sets them up:
    every rank deals ITS shard anew every epoch (NonzeroStream + ReshufflingRunner over the form's stepper); with both tables
    sharded the epoch's fetch lists are agreed between the ranks beside the steps.  mode: "dp", "rowsharded", "sharded"."""
    from trainer.data_utils import NonzeroStream
    from trainer.hip_api import DeviceTables, make_hyper
    from trainer.stepper import HipBackend, ReshufflingRunner, RowShardedStepper, ShardedStepper, Stepper, owned_rows
    hip, dev, dist, world, rank = ctx.hip, ctx.dev, ctx.dist, ctx.world, ctx.rank
    log("%s B=%d Adagrad mode=%s, epochs dealt and indexed inside the timed region: generating the workload" % (workload, B, mode))
    wl = ctx.workload(workload)
    V, d = wl["V"], wl["d"]
    V_row = V_col = V
    if mode in ("sharded", "rowsharded"):
        V_row = owned_rows(V, world, rank)
        V_col = V_row if mode == "sharded" else V
    backend = HipBackend(dev)
    backend.hip = hip
    tables = DeviceTables(V, d, "Adagrad", device=dev, seed=1, V_row=V_row, V_col=V_col)
    backend.row_floats = tables.d
    backend.exchange = True
    if mode in ("sharded", "rowsharded"):
        backend.shard_rows = tables.V_row
    t0 = time.perf_counter()
    stream = NonzeroStream({k: wl[k] for k in ("row", "col", "w", "y")}, B, V, backend, dev, rank=rank, world=world, seed=0,
                           static_plans=False, route=dist if mode in ("sharded", "rowsharded") and world > 1 else None,
                           cols_by_owner=world if mode == "sharded" else 0, presharded=True)
    torch.cuda.synchronize()
    masters_ms = (time.perf_counter() - t0) * 1e3
    nnz = stream.nnz
    hyper_kw = dict(learning_rate=lr, step_form=step_form)
    if mode == "sharded":
        stepper = ShardedStepper(backend, tables, hyper_kw, B, world, rank, dist, collectives=ctx.args.collectives,
                                 exercise_exchange=ctx.args.exercise_exchange)
    elif mode == "rowsharded":
        stepper = RowShardedStepper(backend, tables, hyper_kw, B, world, dist, exchange=exchange, collectives=ctx.args.collectives)
        stepper.prepare(batch_size=B)
    else:
        stepper = Stepper(backend, tables, hyper_kw, B, world, dist, exchange=exchange, collectives=ctx.args.collectives)
        if world > 1 or ctx.args.collectives:
            stepper.prepare(batch_size=B)
        else:
            stepper.dense, stepper.G = True, backend.dense_grad_buffer(tables)
    hyper = make_hyper(batch_size=B * world, **hyper_kw)
    # (a captured multi-rank step issues its collectives in line; the trainer takes graphs there only with --multi-rank-graphs)
    runner = ReshufflingRunner(hip, stream, tables, hyper, chunk_cap=chunk_cap, burst=64, stepper=stepper, graphs=False, segment=segment)
    nb = runner.nb
    log("  masters in %.1f ms (once, at load); %d batches per epoch" % (masters_ms, nb))

    def run(n_steps):
        done = 0
        while done < n_steps:
            done += runner.run(n_steps - done)
    elapsed, elapsed_all = timed_region(ctx, run, steps, warmup, min_timed_ms)
    final_loss = float(stepper.loss_out[0].item())
    log("  %.4f ms per step (%d repeats); per-phase pass" % (elapsed / steps * 1e3, len(elapsed_all)))
    if not (final_loss == final_loss):
        raise SystemExit("loss is NaN")
    torch.cuda.synchronize()
    # ---- what the batches touched, and the phases of one step apart (each timed to its end, collectives included)
    if runner.sharded:
        items = [h for h in runner.handles if stepper.batches[h] is not None][:8]
        plans = [stepper.batches[h]["plan"] for h in items]
    else:
        run(1)                                  # (the current segment's plans are built and adopted)
        torch.cuda.synchronize()
        slot = runner.slots[(runner._g) % 2]
        plans = items = slot.plans[:max(1, min(8, runner.S))]
    counts = torch.stack([p.counts for p in plans]).double()
    counts = counts[counts[:, 0] > 0].mean(0).tolist()
    chunks, u_row, u_col, n_heavy = counts[0] + counts[2], counts[1], counts[3], counts[4]
    if dist is not None and world > 1:          # the ranks' batches differ: the job's bytes are the sum over ranks
        tot = torch.tensor([u_row, u_col], dtype=torch.float64, device=dev)
        dist.all_reduce(tot)
        u_row_all, u_col_all = float(tot[0].item()), float(tot[1].item())
    else:
        u_row_all, u_col_all = u_row, u_col
    calls = dict(stepper.phases())
    finish = getattr(stepper, "finish_async", None)
    if finish is not None:
        calls = {name: (lambda x, fn=fn: (fn(x), finish())) for name, fn in calls.items()}
    kern = {}
    reps = max(len(items), 8)
    for name, fn in calls.items():
        for i in range(min(len(items), 4)):
            fn(items[i % len(items)])
        torch.cuda.synchronize()
        spans = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(reps):
                fn(items[i % len(items)])
            b.record()
            torch.cuda.synchronize()
            spans.append(a.elapsed_time(b) * 1e3 / reps)
        kern[name] = sorted(spans)[1]
    step_us = sum(kern.values())
    # per GPU, like the peak it is held against: one rank's share of the job's step (the mean over the ranks' batches)
    alg = 16 * B + 16 * (d + 1) * (u_row_all + u_col_all) / world
    achieved = alg / (elapsed / steps) / 1e9
    rows = getattr(stepper, "rows", False)
    parallelism = {"dp": "dp%d, %s" % (world, "touched-rows all-gather" if rows else "dense-grad all-reduce"),
                   "sharded": "both tables sharded x%d, touched col rows by all-to-all" % world,
                   "rowsharded": "row table sharded x%d, col side %s" % (
                       world, "local (one rank)" if world == 1 else "touched-rows all-gather" if rows else "dense all-reduce")}[mode]
    peak = HBM_PEAK_GBS
    out = {
        "metric": "co-occurrence nonzeros/sec", "value": steps * B * world / elapsed, "unit": "nonzeros/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "repeats": len(elapsed_all), "ms_per_step_min_max": [min(elapsed_all) / steps * 1e3, max(elapsed_all) / steps * 1e3],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": DATA_NOTE.get(workload, "synthetic"),
        "config": {"workload": workload, "V": V, "d": d, "optimizer": "Adagrad",
                   "batch_size_per_gpu": B, "global_batch": B * world, "nnz_per_gpu": nnz, "batches_per_epoch": nb, "chunk_cap": runner.cap,
                   **({"rehearsal": "ranks share cuda:0 over gloo: control flow only, the numbers mean nothing"}
                      if ctx.args.rehearse_on_one_gpu else {}),
                   "index": "rebuilt every step: every rank deals its shard anew every epoch (trainer.stepper.ReshufflingRunner over the "
                            "form's stepper), indexes and fetch lists prepared beside the steps, inside the timed region",
                   "launch": "the trainer's runner, steps launched eagerly (collectives on the transport's stream)",
                   "parallelism": parallelism,
                   "exchange_floats_per_rank_per_step": getattr(stepper, "payload_floats", None)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": peak, "unit": "GB/s",
                     "frac": achieved / peak, "frac_of_measured_stream_ceiling": achieved / HBM_STREAM_GBS,
                     "stream_ceiling": HBM_STREAM_GBS, "traffic": None,
                     "traffic_source": "multi-rank form: not profiled with counters", "traffic_over_algorithmic": None,
                     "kernel": "per GPU; one step = " + " + ".join(kern),
                     "algorithmic_bytes_per_step": alg, "kernel_us": kern,
                     "step_kernels_alone_frac": alg / (step_us * 1e-6) / 1e9 / peak,
                     "heavy_ids_per_step": n_heavy, "uniq_rows_per_step": u_row, "uniq_cols_per_step": u_col, "chunks_per_step": chunks},
        "masters_build_ms_at_load": masters_ms, "final_loss": final_loss,
    }
    out["collectives"] = collectives_breakdown(kern)
    out["process_group"] = process_group_facts(ctx)
    runner.release_graphs()
    if hasattr(stepper, "release_graphs"):
        stepper.release_graphs()
    del runner, stream, tables, plans, items, stepper
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    log("  done: %.3g nonzeros/s, %.1f us per step of phases, roofline %.3f" % (out["value"], step_us, out["roofline"]["frac"]))
    return out


def run_dealt(ctx, workload, B, optimizer="Adagrad", steps=200, warmup=20, lr=0.05, chunk_cap=0, step_form=0,
              no_graph=False, min_timed_ms=20.0, segment=0):
    """One GPU, the trainer's default mode (--epoch-shuffle full): the stream object and the runner are the trainer's own
    (trainer.data_utils.NonzeroStream, trainer.stepper.ReshufflingRunner), timed from outside — epoch deals, index builds
    and steps all inside the timed region."""
    from trainer.data_utils import NonzeroStream
    from trainer.hip_api import DeviceTables, make_hyper
    from trainer.stepper import HipBackend, ReshufflingRunner
    hip, dev = ctx.hip, ctx.dev
    adam = optimizer == "Adam"
    log("%s B=%d %s, epochs dealt and indexed inside the timed region: generating the workload" % (workload, B, optimizer))
    wl = ctx.workload(workload)
    V, d = wl["V"], wl["d"]
    nnz = wl["row"].numel()
    if nnz < B:
        raise SystemExit("workload has %d nonzeros < batch size %d" % (nnz, B))
    backend = HipBackend(dev)
    backend.hip = hip
    tables = DeviceTables(V, d, optimizer, device=dev, seed=1)
    backend.row_floats = tables.d
    if step_form == 4:
        tables.enable_twin()
    if step_form == 5:
        tables.enable_tags()
    t0 = time.perf_counter()
    stream = NonzeroStream({k: wl[k] for k in ("row", "col", "w", "y")}, B, V, backend, dev, seed=0, static_plans=False)
    torch.cuda.synchronize()
    masters_ms = (time.perf_counter() - t0) * 1e3
    hyper = make_hyper(batch_size=B, learning_rate=lr, step_form=step_form)
    runner = ReshufflingRunner(hip, stream, tables, hyper, chunk_cap=chunk_cap, burst=64, graphs=False if no_graph else None, segment=segment)
    cap, S, nb = runner.cap, runner.S, runner.nb
    log("  masters in %.1f ms (once, at load); %d batches per epoch, index built %d batches at a time, chunk records: %s" % (
        masters_ms, nb, S, "run words instead (pair fields as dealt)" if getattr(runner, "run_words", False) else runner.records))

    def run(n_steps):
        done = 0
        while done < n_steps:
            done += runner.run(n_steps - done)
    elapsed, elapsed_all = timed_region(ctx, run, steps, warmup, min_timed_ms)
    final_loss = float(runner.loss_out[0].item())
    log("  %.4f ms per step (%d repeats); per-kernel pass" % (elapsed / steps * 1e3, len(elapsed_all)))
    if not (final_loss == final_loss):
        raise SystemExit("loss is NaN")
    # ---- what the batches touched (the staging plans of both slots, as the last builds left them)
    torch.cuda.synchronize()
    plans = [p for blk in runner.slots for p in blk.plans]
    counts = torch.stack([p.counts for p in plans]).double()
    counts = counts[counts[:, 0] > 0].mean(0).tolist()          # (a slot's tail plans are never built when the epoch's last segment is short)
    chunks, u_row, u_col, n_heavy = counts[0] + counts[2], counts[1], counts[3], counts[4]
    # ---- the pieces apart, each on its own with HIP events on its launch stream (the timed region overlaps them)
    n0 = min(S, nb)
    slot = runner.slots[0]
    rs, cs = stream.epoch_sides()
    with torch.cuda.stream(stream.side):        # slot 0 indexed afresh, its counts adopted as the runner does at a segment's start
        hip.build_plans_sorted(rs, cs, 0, slot, n0, V, runner.sorted_ws)
        if runner.host_counts:
            slot.fetch_counts()
    torch.cuda.synchronize()
    if runner.host_counts:
        slot.adopt_counts(n0)

    def steps_once():
        runner._steps(slot.plans[:n0])
    steps_once()
    torch.cuda.synchronize()
    kg = torch.cuda.CUDAGraph()               # replayed, so that kernels shorter than a host call are timed on the GPU's clock
    with torch.cuda.graph(kg):
        steps_once()
    kern = {"step": event_us(kg.replay) / n0}
    kern["index_build"] = event_us(lambda: hip.build_plans_sorted(rs, cs, 0, slot, n0, V, runner.sorted_ws), stream=stream.side) / n0
    spare = stream._sets[(stream.epoch + 1) % 2]
    kern["epoch_deal"] = event_us(lambda: hip.deal_epoch(stream.masters, B, 12345, spare[0], spare[1], stream._deal_ws),
                                  stream=stream.side) / nb
    alg = algorithmic_bytes_adam(B, V, d) if adam else algorithmic_bytes(B, d, u_row, u_col)
    achieved = alg / (elapsed / steps) / 1e9
    traffic, traffic_src = measured_traffic(workload, B, "dealt") if not adam else (None, "Adam: not profiled with counters")
    out = {
        "metric": "co-occurrence nonzeros/sec", "value": steps * B / elapsed, "unit": "nonzeros/s",
        "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "repeats": len(elapsed_all), "ms_per_step_min_max": [min(elapsed_all) / steps * 1e3, max(elapsed_all) / steps * 1e3],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": DATA_NOTE.get(workload, "synthetic"),
        "config": {"workload": workload, "V": V, "d": d, "optimizer": optimizer,
                   "batch_size_per_gpu": B, "global_batch": B, "nnz_per_gpu": nnz, "batches_per_epoch": nb, "chunk_cap": cap,
                   "index": "rebuilt every step: epochs dealt from the sorted master orders (one partition pass per epoch), the "
                            "index of %d consecutive batches numbered by %d launches on a side stream, inside the timed region%s" % (
                                S, 3 if runner.records else 2,
                                "" if runner.records else "; no chunk records: run words, pair fields %s the epoch's arrays" % (
                                    "borrowed from" if getattr(runner, "borrow", False) else "copied from")),
                   "launch": "the trainer's runner: steps replayed from hipGraphs of 2^k steps" if runner.graphs_on else
                             "the trainer's runner: every run of steps issued by one C call (glove_steps_adagrad_f32)",
                   "parallelism": "single GPU", "chunk_records": bool(runner.records), "chunk_run_words": bool(getattr(runner, "run_words", False))},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_stream_ceiling": achieved / HBM_STREAM_GBS,
                     "stream_ceiling": HBM_STREAM_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "traffic_over_algorithmic": (traffic / alg) if traffic else None,
                     "kernel": "one step as timed = step kernels, beside them on a side stream index_build and epoch_deal",
                     "algorithmic_bytes_per_step": alg, "kernel_us": kern,
                     "kernel_us_note": "per step, each piece alone on an idle GPU (HIP events on its launch stream); in the timed "
                                       "region index_build and epoch_deal overlap the steps",
                     "step_kernels_alone_frac": alg / (kern["step"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "heavy_ids_per_step": n_heavy, "uniq_rows_per_step": u_row, "uniq_cols_per_step": u_col, "chunks_per_step": chunks},
        "masters_build_ms_at_load": masters_ms, "final_loss": final_loss,
    }
    runner.release_graphs()
    del kg, runner, stream, tables, plans
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    log("  done: %.3g nonzeros/s, %.4f ms per step, roofline %.3f (step kernels alone %.1f us, index %.1f us, deal %.1f us per step)" % (
        out["value"], out["ms_per_step"], out["roofline"]["frac"], kern["step"], kern["index_build"], kern["epoch_deal"]))
    return out


def run_config(ctx, workload, B, optimizer="Adagrad", mode="auto", steps=200, warmup=20, lr=0.05, chunk_cap=0,
               step_form=0, exchange="auto", dynamic=False, no_graph=False, max_batches=64,
               min_timed_ms=20.0, segment=0):
    """mode: "auto" (single-GPU sparse step on one rank, data parallel on several), "dp" (data-parallel form also on
    one rank), "sharded" (both tables sharded), "rowsharded" (row table sharded, col side data parallel)."""
    from trainer.hip_api import FUSED_STEP_BYTES, DeviceTables, auto_chunk_cap, make_hyper, row_width
    from trainer.stepper import HipBackend, RowShardedStepper, ShardedStepper, Stepper, owned_rows, route_by_row_owner
    hip, dev, dist, world, rank = ctx.hip, ctx.dev, ctx.dist, ctx.world, ctx.rank
    adam = optimizer == "Adam"
    if mode == "auto":
        mode = "dp" if world > 1 else "single"
    if adam and mode != "single":
        raise SystemExit("--optimizer Adam is benchmarked on one GPU")
    if dynamic and mode == "single":
        return run_dealt(ctx, workload, B, optimizer, steps, warmup, lr, chunk_cap, step_form, no_graph, min_timed_ms, segment)
    if dynamic:
        return run_dealt_multi(ctx, workload, B, mode, steps, warmup, lr, chunk_cap, step_form, exchange, no_graph, min_timed_ms, segment)
    log("%s B=%d %s mode=%s: generating the workload" % (workload, B, optimizer, mode))
    wl = ctx.workload(workload)
    V, d = wl["V"], wl["d"]
    coo = {k: wl[k] for k in ("row", "col", "w", "y")}
    V_row = V_col = V
    if mode in ("sharded", "rowsharded"):
        V_row = owned_rows(V, world, rank)
        V_col = V_row if mode == "sharded" else V
        if world > 1:
            coo = route_by_row_owner(coo, world, rank, dist)
    cap = chunk_cap or auto_chunk_cap(B, V, row_width(V, d))
    nnz = coo["row"].numel()
    if nnz < B:
        raise SystemExit("workload has %d nonzeros < batch size %d" % (nnz, B))
    nb = min(max(1, nnz // B), max_batches)
    if dist is not None and world > 1:
        # every rank's shard has its own nonzero count: agree on the number of resident batches, so that all
        # ranks issue the same number of collectives in every loop below
        agreed = torch.tensor([nb], dtype=torch.int64, device=dev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        nb = int(agreed.item())

    tables = DeviceTables(V, d, optimizer, device=dev, seed=1, V_row=V_row, V_col=V_col)   # identical replicas on every rank
    if mode == "single" and step_form == 4:
        tables.enable_twin()
    if mode == "single" and step_form == 5:
        tables.enable_tags()
    backend = HipBackend(dev)
    backend.hip = hip
    backend.row_floats = tables.d
    backend.exchange = mode != "single"         # the multi-rank forms' packing passes read chunk records
    hyper_kw = dict(learning_rate=lr, step_form=step_form)

    # ---- load time (untimed): resident batches + their dedup index
    batches = [tuple(coo[k][b * B:(b + 1) * B].contiguous() for k in ("row", "col", "w", "y")) for b in range(nb)]
    log("  index of %d resident batches" % nb)
    t0 = time.perf_counter()
    stepper = None
    if mode == "sharded":
        stepper = ShardedStepper(backend, tables, hyper_kw, B, world, rank, dist, collectives=ctx.args.collectives,
                                 exercise_exchange=ctx.args.exercise_exchange)
        handles = [stepper.add_batch(*bt, cap) for bt in batches]
        plans = [stepper.batches[h]["plan"] for h in handles]
    else:
        plans = [hip.build_plan(*bt, V, chunk_cap=cap, compact=True, d=tables.d, V_row=V_row if V_row < V else 0,
                                run_words=None if mode == "single" else False)
                 for bt in batches]
        handles = plans
    torch.cuda.synchronize()
    plan_build_ms = (time.perf_counter() - t0) * 1e3 / nb
    counts = [p.host_counts for p in plans]
    n_heavy = sum(c[4] for c in counts) / nb
    u_row = sum(c[1] for c in counts) / nb
    u_col = sum(c[3] for c in counts) / nb
    chunks = sum(c[0] + c[2] for c in counts) / nb
    if mode == "rowsharded":
        stepper = RowShardedStepper(backend, tables, hyper_kw, B, world, dist, exchange=exchange, collectives=ctx.args.collectives)
        stepper.prepare(plans)
    elif mode == "dp":
        stepper = Stepper(backend, tables, hyper_kw, B, world, dist, exchange=exchange, collectives=ctx.args.collectives)
        if world == 1:
            stepper.dense, stepper.G = True, backend.dense_grad_buffer(tables)
        stepper.prepare(plans)

    if mode == "single" and step_form in (0, 5) and plans[0].r_crec is not None:
        tables.maybe_enable_tags(B)         # small batches on small tables: the tagged step (as Stepper does)
    if mode == "single" and step_form == 0 and not adam and plans[0].fusable and \
            (u_row + u_col) * tables.d * 16 >= FUSED_STEP_BYTES:
        tables.maybe_enable_twin()          # the library will take a fused form: give it the twinned row table (as Stepper does)
    hyper = make_hyper(batch_size=B * world, **hyper_kw)
    loss_out = stepper.loss_out if stepper is not None else torch.zeros(4, device=dev)
    ws = None
    G = hip.dense_grad_buffer(tables) if adam else None
    if mode == "single":
        ws = torch.empty(max(hip.lib.glove_step_workspace_bytes(B, p.cap_chunks, tables.d) for p in plans), dtype=torch.uint8, device=dev)

    def step(i):
        if stepper is not None:
            stepper.step(handles[i % nb])
            return
        plan = plans[i % nb]
        if adam:
            hip.step_adam(plan, tables, hyper, G, loss_out, ws)
        else:
            hip.step_adagrad(plan, tables, hyper, loss_out, ws)

    log("  %.1f ms per index; warm-up and capture" % plan_build_ms)
    # One hipGraph holds `spg` consecutive steps, a divisor of --steps: the timed region is a whole number of replays.
    # (a multi-rank step is captured with its collectives when the transport is RCCL; a gloo rehearsal launches eagerly)
    from trainer.stepper import transport_is_capturable
    use_graph = not no_graph and (mode == "single" or transport_is_capturable(dist, getattr(stepper, "_multi", False)))
    if use_graph and mode in ("sharded", "rowsharded") and getattr(stepper, "_multi", False) and B > 65536:
        # a captured step issues its collectives in line (SideCollective): a big batch gains more from the push / gather
        # overlapping the row side (eager launches, far ahead of a millisecond step) than from saving launches
        use_graph = False
    graph, spg = None, steps_per_graph(steps)
    if use_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(min(nb, 4)):
                step(i)                       # warm the launch path on the capture stream
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                if stepper is None and not adam:
                    # as the trainer's static mode issues them (Stepper.step_many): one host call; on step-tagged tables the
                    # library chains consecutive small batches, one launch per step
                    hip.steps_adagrad([plans[i % nb] for i in range(spg)], tables, hyper, loss_out, ws=ws)
                elif stepper is None:
                    hip.steps_adam([plans[i % nb] for i in range(spg)], tables, hyper, G, loss_out, ws=ws)
                else:
                    for i in range(spg):
                        step(i)
        except Exception as exc:             # a transport that refuses capture: the same steps, launched eagerly
            if mode == "single":
                raise
            log("  hipGraph capture of the multi-rank step failed (%s: %s): launching eagerly" % (type(exc).__name__, exc))
            graph = None
            torch.cuda.synchronize()

    def run(n_steps, first):
        done = 0
        if graph is not None:
            for _ in range(n_steps // spg):
                graph.replay()
            done = (n_steps // spg) * spg
        for i in range(done, n_steps):
            step(first + i)

    elapsed, elapsed_all = timed_region(ctx, lambda n: run(n, 0), steps, warmup, min_timed_ms)
    final_loss = float(loss_out[0].item())
    log("  %.4f ms per step (%d repeats); per-kernel pass" % (elapsed / steps * 1e3, len(elapsed_all)))
    if not (final_loss == final_loss):
        raise SystemExit("loss is NaN")

    # ---- instrumented pass: HIP events (on the launch stream) around `reps` back-to-back launches of
    # each piece of the step over the same resident batches; per-launch time = span / reps (it includes
    # the ~1-2 us launch-to-launch gap that rocprofv3's per-kernel durations exclude)
    reps = max(min(nb, 8), min(200, steps))
    if stepper is not None:
        calls = dict(stepper.phases())
        items = handles
    else:
        items = plans
        form = step_form
        calls = {}
        if adam:
            adam_fused = 2 * B <= V_row + V          # glove_step_adam_f32's own rule (include/glove_hip.h)
            one_launch = adam_fused and tables.R_tag is not None and B <= 2048 and plans[0].r_mark is not None and form in (0, 5)
            if one_launch:
                # twinned tables, plans with id bitmaps: the whole step is ONE kernel (+ a one-workgroup epilogue)
                calls["step_adam_one_launch"] = lambda p: hip.step_adam(p, tables, hyper, G, loss_out, ws)
            else:
                calls["passes"] = lambda p: hip.passes(p, tables, hyper, ws)
            if one_launch:
                pass
            elif adam_fused:
                # passes (+ id marks) and ONE fused apply/decay kernel; the second kernel has no entry point of its own,
                # so it is timed as the whole step minus the passes
                calls["step_adam"] = lambda p: hip.step_adam(p, tables, hyper, G, loss_out, ws)
            else:
                calls["dense_grad"] = lambda p: hip.dense_grad(p, tables, hyper, G, ws)
                calls["dense_adam"] = lambda p: hip.dense_adam(tables, hyper, G, loss_out)
        else:
            fused_auto = plans[0].fusable and (u_row + u_col) * tables.d * 16 >= FUSED_STEP_BYTES
            if form == 0:
                form = 5 if tables.R_tag is not None and B <= 2048 else (4 if tables.R_ver is not None else 3) if fused_auto else 1
            if form == 1 or not plans[0].fusable:
                calls["passes"] = lambda p: hip.passes(p, tables, hyper, ws)      # row side + col side, one launch
                calls["apply_adagrad"] = lambda p: hip.apply_adagrad(p, tables, hyper, loss_out, ws)
            else:
                # the fused forms have no entry points per launch: the step is timed whole (rocprofv3 splits it:
                # profiles/*_kernel_stats.txt)
                calls["step_fused_form_%d" % form] = lambda p: hip.step_adagrad(p, tables, hyper, loss_out, ws)
    kern = {}
    finish = getattr(stepper, "finish_async", None)
    if finish is not None:
        # a phase that only STARTS a collective (the sharded forms overlap it with the row side) is timed to its end here
        calls = {name: (lambda x, fn=fn: (fn(x), finish())) for name, fn in calls.items()}
    for name, fn in calls.items():
        for i in range(min(2 * nb, 8)):
            fn(items[i % nb])
        torch.cuda.synchronize()

        def burst(fn=fn):
            for i in range(reps):
                fn(items[i % nb])
        if mode == "single":                  # replayed from a hipGraph, so that a kernel shorter than the host's
            kg = torch.cuda.CUDAGraph()       # per-call cost (small batches) is still timed on the GPU's clock
            with torch.cuda.graph(kg):
                burst()
            burst = kg.replay
            burst()
        spans = []
        for _ in range(3):                    # median of three spans: robust against a transient on a fresh box
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            burst()
            b.record()
            torch.cuda.synchronize()
            spans.append(a.elapsed_time(b) * 1e3 / reps)
        kern[name] = sorted(spans)[1]
    if "step_adam" in kern:
        kern["adam_fused"] = kern.pop("step_adam") - kern["passes"]
    # attribution of the algorithmic bytes to the two kernels of the two-launch sparse step: the pass
    # kernel owns the nonzero stream and ONE read of every distinct row + bias; the apply kernel owns the accumulator
    # read and the two writes
    per_kernel = None
    if stepper is None and not adam and "apply_adagrad" in kern:
        attributed = {"passes": 16 * B + 4 * (d + 1) * (u_row + u_col), "apply_adagrad": 12 * (d + 1) * (u_row + u_col)}
        per_kernel = {k: {"algorithmic_bytes": attributed[k], "avg_us": kern[k],
                          "achieved_GBps": attributed[k] / (kern[k] * 1e-6) / 1e9,
                          "frac": attributed[k] / (kern[k] * 1e-6) / 1e9 / HBM_PEAK_GBS} for k in attributed}
    step_us = sum(kern.values())
    alg = algorithmic_bytes_adam(B, V, d) if adam else algorithmic_bytes(B, d, u_row, u_col)
    # the whole step as timed (contract: bytes / ms_per_step); the kernels alone, back to back, are reported beside it
    achieved = alg / (elapsed / steps) / 1e9
    traffic, traffic_src = measured_traffic(workload, B, "static") if mode == "single" and not adam else \
        (None, "Adam: not profiled with counters" if adam else "multi-rank form: not profiled with counters")
    rows = getattr(stepper, "rows", False)
    parallelism = {"single": "single GPU",
                   "dp": "dp%d, %s" % (world, "touched-rows all-gather" if rows else "dense-grad all-reduce"),
                   "sharded": "both tables sharded x%d, touched col rows by all-to-all" % world,
                   "rowsharded": "row table sharded x%d, col side %s" % (
                       world, "local (one rank)" if world == 1 else "touched-rows all-gather" if rows else "dense all-reduce")}[mode]
    out = {
        "metric": "co-occurrence nonzeros/sec", "value": steps * B * world / elapsed, "unit": "nonzeros/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
        "repeats": len(elapsed_all), "ms_per_step_min_max": [min(elapsed_all) / steps * 1e3, max(elapsed_all) / steps * 1e3],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": DATA_NOTE.get(workload, "synthetic"),
        "config": {"workload": workload, "V": V, "d": d, "optimizer": optimizer,
                   "batch_size_per_gpu": B, "global_batch": B * world, "nnz_per_gpu": nnz,
                   "resident_batches": nb, "chunk_cap": cap,
                   **({"rehearsal": "ranks share cuda:0 over gloo: control flow only, the numbers mean nothing"}
                      if ctx.args.rehearse_on_one_gpu else {}),
                   "index": "static, built at load (the trainer's --epoch-shuffle static)",
                   "launch": "hipGraph replay, %d steps per graph" % spg if graph is not None else "eager",
                   "parallelism": parallelism,
                   **({"exchange_floats_per_rank_per_step": getattr(stepper, "payload_floats", None)} if stepper is not None else {})},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_stream_ceiling": achieved / HBM_STREAM_GBS,
                     "stream_ceiling": HBM_STREAM_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "traffic_over_algorithmic": (traffic / alg) if traffic else None,
                     "kernel": "one step = " + " + ".join(kern),
                     "algorithmic_bytes_per_step": alg, "kernel_us": kern, "per_kernel": per_kernel,
                     "step_kernels_alone_frac": alg / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "heavy_ids_per_step": n_heavy, "uniq_rows_per_step": u_row, "uniq_cols_per_step": u_col, "chunks_per_step": chunks},
        "plan_build_ms_per_batch": plan_build_ms, "final_loss": final_loss,
    }
    if stepper is not None:
        out["collectives"] = collectives_breakdown(kern)
        out["process_group"] = process_group_facts(ctx)
    del tables, plans, handles, batches, stepper, graph
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    log("  done: %.3g nonzeros/s, %.1f us per step of kernels, roofline %.3f" % (out["value"], step_us, out["roofline"]["frac"]))
    return out


def brief(r: dict) -> dict:
    """A configuration as the side file (bench_configs.json) keeps it."""
    keep = ("name", "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "repeats", "dtype", "data", "config",
            "roofline", "final_loss", "collectives")
    return {k: r[k] for k in keep if k in r}


def config_line(r: dict) -> dict:
    """A configuration as its own short stderr line (`[bench-config] {...}`, well under 1 KB)."""
    rf, cf = r["roofline"], r["config"]
    out = {"name": r.get("name"), "workload": cf["workload"], "optimizer": cf["optimizer"], "B": cf["batch_size_per_gpu"],
           "n_gpus": r["n_gpus"], "index": "static" if cf["index"].startswith("static") else "dealt",
           "parallelism": cf["parallelism"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
           "frac": rf["frac"], "step_kernels_alone_frac": rf["step_kernels_alone_frac"],
           "traffic_over_algorithmic": rf["traffic_over_algorithmic"], "algorithmic_bytes_per_step": rf["algorithmic_bytes_per_step"],
           "kernel_us": {k: round(v, 2) for k, v in rf["kernel_us"].items()}}
    if "collectives" in r:
        out["collectives_ms"] = {k: round(v, 4) for k, v in r["collectives"].items() if k.endswith("_ms")}
    return out


def headline(out: dict) -> dict:
    """The ONE stdout line: the headline configuration only, a few KB at most (a driver keeps the tail of stdout — every
    other configuration goes to stderr as `[bench-config]` lines and, whole, to bench_configs.json)."""
    rf = out["roofline"]
    cfg = {k: out["config"][k] for k in ("workload", "V", "d", "optimizer", "batch_size_per_gpu", "global_batch", "nnz_per_gpu",
                                         "chunk_cap", "parallelism", "rehearsal", "same_workload_static_index",
                                         "exchange_floats_per_rank_per_step") if k in out["config"]}
    cfg["index"] = "static" if out["config"]["index"].startswith("static") else "dealt"
    cfg["index_note"] = out["config"]["index"][:160]
    cfg["launch"] = out["config"]["launch"][:120]
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data") if k in out}
    line["config"] = cfg
    line["roofline"] = {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                                           "traffic_over_algorithmic", "algorithmic_bytes_per_step", "kernel",
                                           "step_kernels_alone_frac", "frac_of_measured_stream_ceiling") if k in rf}
    line["roofline"]["kernel_us"] = {k: round(v, 2) for k, v in rf["kernel_us"].items()}
    cb = out.get("cpu_baseline")
    if cb is not None:
        line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "host_cpus", "usable_cores",
                                                   "cpu_model", "label") if k in cb}
        line["cpu_baseline"]["legs"] = {name: {"value": leg["value"], "cores": leg["cores"]}
                                        for name, leg in cb.get("legs", {}).items() if isinstance(leg, dict)}
        if "error" in cb.get("legs", {}):
            line["cpu_baseline"]["error"] = str(cb["legs"]["error"])[:200]
    for k in ("configs_skipped", "configs_failed", "configs_file", "final_loss", "wall_seconds"):
        if k in out:
            line[k] = out[k]
    if "configs" in out:
        line["configs_run"] = [r["name"] for r in out["configs"]]
    if "collectives" in out:        # the per-phase times are in roofline.kernel_us already
        line["collectives"] = {k: round(v, 4) for k, v in out["collectives"].items() if k.endswith("_ms")}
    if "process_group" in out:
        pg = out["process_group"]
        line["process_group"] = {k: pg[k] for k in ("world_size", "backend", "rccl_version", "distinct_devices") if k in pg}
        line["process_group"]["devices_by_rank"] = [r.get("pci_bus_id") or r.get("cuda_device") for r in pg.get("ranks", [])]
    return line


def write_configs_file(out: dict) -> str:
    """The full objects (headline + every configs[] entry) beside the run; returns the path written ('' if none could be)."""
    for d in (REPO / "gpurun_out", REPO, Path("/tmp")):
        try:
            d.mkdir(exist_ok=True)
            path = d / "bench_configs.json"
            with open(path, "w") as f:
                json.dump(out, f, indent=1)
            return str(path.relative_to(REPO)) if d != Path("/tmp") else str(path)
        except OSError:
            continue
    return ""


def emit_headline(out: dict, rank: int) -> None:
    """Rank 0: the headline as a `[bench-config]` stderr line, the side file, and THE stdout line (the only JSON line there)."""
    if rank != 0:
        return
    out["name"] = "headline"
    print("[bench-config] " + json.dumps(config_line(out)), file=sys.stderr, flush=True)
    out["configs_file"] = write_configs_file(out)
    sys.stderr.flush()
    print(json.dumps(headline(out)), flush=True)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            # no launcher: be one.  Nothing has touched the GPU in this process (device_count() does not)
            if not args.rehearse_on_one_gpu and torch.cuda.device_count() < args.gpus:
                raise SystemExit("--gpus %d but %d visible" % (args.gpus, torch.cuda.device_count()))
            raise SystemExit(launch_ranks(args.gpus, argv))
        args.gpus = world
    if world > 1 or args.row_sharded or args.force_dense:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # (trainer.estimator.multi_rank_queues: before the first GPU call)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # a rank that hangs (a collective nobody else entered) ends with a traceback instead of holding the node
    import faulthandler
    try:
        faulthandler.dump_traceback_later(max(900.0, 3.0 * args.budget_seconds), exit=True)
    except (OSError, ValueError):               # no usable stderr
        pass
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    mode = "auto"
    if args.row_sharded:
        mode = "rowsharded" if args.cols_replicated else "sharded"
    elif args.force_dense:
        mode = "dp"
    if world > 1 or mode != "auto":
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:      # a multi-rank form on a single GPU without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from trainer import hip_api
    if not hip_api.LIB_PATH.exists():
        # sources arrived without the built library: the first local rank compiles it (hipcc is part of the image),
        # the others wait for the file; a failing build ends the run
        if local_rank == 0 or args.rehearse_on_one_gpu and rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        for _ in range(600):
            if hip_api.LIB_PATH.exists():
                break
            time.sleep(1.0)
    hip = hip_api.GloveHip(dev)
    ctx = Ctx(args, world, rank, dev, dist, hip)
    common = dict(lr=args.learning_rate, chunk_cap=args.chunk_cap, step_form=args.step_form, exchange=args.exchange,
                  no_graph=args.no_graph, max_batches=args.max_batches, min_timed_ms=args.min_timed_ms)
    B_head = args.batch_size or DEFAULT_BATCH[args.workload]
    big = args.workload in ("zipf_v400k_d300", "zipf_v2m_d128")
    if mode == "auto" and world > 1 and big and args.exchange == "auto":
        mode = "sharded"        # the form whose exchange follows the batch, not the vocabulary (DESIGN.md "Multi-GPU")
    out = run_config(ctx, args.workload, B_head, args.optimizer, mode, args.steps, args.warmup,
                     dynamic=not args.static_index, segment=args.index_segment, **common)
    plain = not (args.single or args.static_index or args.index_segment or args.optimizer != "Adagrad" or args.row_sharded or args.force_dense or
                 args.step_form or args.chunk_cap or args.batch_size or args.workload != "zipf_v400k_d300" or
                 (args.rehearse_on_one_gpu and not args.with_configs))
    if rank == 0 and not args.no_cpu_baseline and world == 1 and mode == "auto":
        log("cpu baseline (%g s)" % args.cpu_seconds)
        out["cpu_baseline"] = cpu_baseline(ctx, ctx.workload(args.workload), B_head, args.learning_rate, args.cpu_seconds)
    if plain:
        # the other configurations of BASELINE.json / BASELINE.md §3 in the same line, the HBM-bound ones last (the tail of
        # a long line is what a log keeps).  (name, estimated seconds incl. load, run_config arguments)
        extra = dict(lr=args.learning_rate, max_batches=8, min_timed_ms=args.min_timed_ms, exchange=args.exchange)
        specs = ([  # the headline workload in the trainer's other mode (--epoch-shuffle static: index built once at load)
                  ("c4_zipf_v400k_d300_static_index", 12, dict(workload="zipf_v400k_d300", B=1048576, steps=24, warmup=4, max_batches=16)),
                  # BASELINE configs[0]'s shape (Keras-legacy Adam, the reference's default batch) as the trainer runs it, and static
                  ("c1_shape_adam_bs1024", 5, dict(workload="text8_d64", B=1024, optimizer="Adam", steps=2000, warmup=200, lr=0.001, dynamic=True)),
                  ("c1_shape_adam_bs1024_static_index", 4, dict(workload="text8_d64", B=1024, optimizer="Adam", steps=2000, warmup=200, lr=0.001, max_batches=1024)),
                  # BASELINE configs[1] at the reference's batch size, at 131,072 pairs and at (nearly) the whole stream per step
                  ("text8_d64_bs1024", 5, dict(workload="text8_d64", B=1024, steps=2000, warmup=200, dynamic=True)),
                  # (1,024 resident batches, most of an epoch, as the trainer's static mode keeps ALL of them: cycling through 8 leaves their
                  # records and rows cache-hot — 8.4 against 8.9 us per step, profiles/r05_exp_resident_batches_bs1024.txt)
                  ("text8_d64_bs1024_static_index", 4, dict(workload="text8_d64", B=1024, steps=2000, warmup=200, max_batches=1024)),
                  ("c2_text8_d64", 5, dict(workload="text8_d64", B=131072, steps=200, warmup=20, dynamic=True)),
                  ("c2_text8_d64_static_index", 4, dict(workload="text8_d64", B=131072, steps=200, warmup=20, max_batches=64)),
                  ("text8_d64_bs1048576_static_index", 4, dict(workload="text8_d64", B=1048576, steps=100, warmup=10)),
                  ("c3_text8_v50k_d300", 7, dict(workload="text8_v50k_d300", B=131072, steps=100, warmup=10, dynamic=True)),
                  ("c3_text8_v50k_d300_static_index", 6, dict(workload="text8_v50k_d300", B=131072, steps=100, warmup=10)),
                  ("c5_zipf_v2m_d128_one_gpu_shard", 15, dict(workload="zipf_v2m_d128", B=1048576, steps=24, warmup=4, dynamic=True)),
                  ("c5_zipf_v2m_d128_one_gpu_shard_static_index", 15, dict(workload="zipf_v2m_d128", B=1048576, steps=24, warmup=4))]
                 if world == 1 else
                 # config 4 as BASELINE.json words it (nonzeros sharded, gradient exchange per step), and config 5
                 [("c4_zipf_v400k_d300_static_index", 60, dict(workload="zipf_v400k_d300", B=1048576, steps=12, warmup=3, mode="sharded")),
                  ("c4_zipf_v400k_d300_data_parallel_static_index", 60, dict(workload="zipf_v400k_d300", B=1048576, steps=12, warmup=3, mode="dp")),
                  ("c5_zipf_v2m_d128_both_tables_sharded", 70, dict(workload="zipf_v2m_d128", B=1048576, steps=12, warmup=3, mode="sharded", dynamic=True)),
                  ("c5_zipf_v2m_d128_both_tables_sharded_static_index", 60, dict(workload="zipf_v2m_d128", B=1048576, steps=12, warmup=3, mode="sharded"))])
        if world > 1 and not args.with_configs:
            # more than one GPU: the static companion of the headline only; the other forms on request (--with-configs)
            specs = specs[:1]
        if world > 1:
            # no run of this path on more than one GPU exists: the headline is on stdout before any side
            # configuration starts, so nothing that happens in one of them can take it away
            out["wall_seconds"] = time.perf_counter() - T_START
            emit_headline(out, rank)
        configs, skipped, failed = [], [], []
        for name, est, spec in specs:
            # the decision is rank 0's (the ranks' clocks differ) and collective
            go = torch.tensor([1 if time.perf_counter() - T_START + est <= args.budget_seconds and not (failed and world > 1) else 0], device=dev)
            if world > 1:
                dist.broadcast(go, src=0)
            if not int(go.item()):
                skipped.append(name)
                continue
            kw = dict(extra)
            kw.update({k: v for k, v in spec.items() if k not in ("workload", "B")})
            try:
                r = run_config(ctx, spec["workload"], spec["B"], **kw)
            except Exception as exc:            # a side configuration never costs the run its headline
                import traceback
                traceback.print_exc()
                failed.append({"name": name, "error": ("%s: %s" % (type(exc).__name__, exc))[:200]})
                if world > 1:
                    break                       # (the ranks are no longer in step: no further collective decision)
                continue
            r["name"] = name
            r = brief(r)
            configs.append(r)
            if rank == 0:
                print("[bench-config] " + json.dumps(config_line(r)), file=sys.stderr, flush=True)
        out["configs_skipped"] = skipped
        if failed:
            out["configs_failed"] = failed
        out["configs"] = configs
        # the headline workload in the trainer's static mode, beside the headline
        for r in configs:
            if r["name"] == "c4_zipf_v400k_d300_static_index":
                out["config"]["same_workload_static_index"] = {
                    "nonzeros_per_s": r["value"], "ms_per_step": r["ms_per_step"], "index": r["config"]["index"]}
    out.setdefault("wall_seconds", time.perf_counter() - T_START)
    if world > 1 and plain:
        if rank == 0:
            write_configs_file(out)             # (the stdout line went out before the side configurations)
    else:
        emit_headline(out, rank)
    faulthandler.cancel_dump_traceback_later()
    if dist is not None:
        import gc
        gc.collect()                 # no captured collective may outlive the communicator: RCCL's teardown waits for them
        torch.cuda.synchronize()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
