#!/usr/bin/env python3
"""bench.py — co-occurrence nonzeros/sec of the GloVe training step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload text8_d64] [--batch-size B]

A "step" is one optimizer step (the fused forward+gradient pass kernel + the sparse Adagrad apply kernel) over one
batch of B synthetic co-occurrence nonzeros that are already resident in HBM together with
their dedup index (DESIGN.md "Data layout"; the index of a static stream is built once at load
time, `--dynamic` puts the index build of every batch inside the timed region instead).
N > 1 (launched by torch.distributed.run, one rank per GPU): every rank owns its own shard of
nonzeros, computes the summed gradients of its batch into a dense buffer, the buffers are
all-reduced over RCCL and every rank applies the identical dense Adagrad update
(global batch = N * B, weak scaling).

Rank 0 prints ONE JSON line; see the task contract for the fields.  `roofline` is measured live
with HIP events on the launch stream in a second, instrumented pass over the same batches;
`cpu_baseline` times the scalar C port of the oracle (oracle/glove_ref.c) on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md "Chip-level parameters")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="text8_d64", choices=["text8_d64", "text8_v50k_d300", "zipf_v400k_d300", "zipf_v2m_d128"])
    ap.add_argument("--row-sharded", action="store_true",
                    help="BASELINE config 5: row table sharded by id %% N, nonzeros routed to row owners (all-to-all at load), "
                         "col side data parallel")
    ap.add_argument("--batch-size", type=int, default=131072)
    ap.add_argument("--chunk-cap", type=int, default=0, help="0 = auto (hip_api.auto_chunk_cap)")
    ap.add_argument("--force-dense", action="store_true", help="run the data-parallel form (dense gradient buffer + all-reduce) also on one GPU")
    ap.add_argument("--optimizer", default="Adagrad", choices=["Adagrad", "Adam"],
                    help="Adam = Keras-legacy dense-decay Adam (config 1 of BASELINE.json), single GPU only")
    ap.add_argument("--learning-rate", type=float, default=0.05)
    ap.add_argument("--dynamic", action="store_true", help="rebuild the dedup index of every batch inside the timed region")
    ap.add_argument("--build-ahead", type=int, default=1,
                    help="with --dynamic: index builds in flight (each on its own stream and staging plan), the way an "
                         "input pipeline prefetches batches; 1 = build and step strictly alternate on one stream")
    ap.add_argument("--no-graph", action="store_true", help="launch every step from Python instead of replaying a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--max-batches", type=int, default=64, help="resident batches to cycle through")
    ap.add_argument("--step-form", type=int, default=0, help="glove_hyper.step_form: 0 auto, 1 two launches, 2 fused one pass, 3 fused three launches")
    ap.add_argument("--single", action="store_true", help="only the named workload (no configs[] array)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="control-flow rehearsal of the multi-rank path on a one-GPU box: every rank uses cuda:0 and the "
                         "collectives go through gloo; its numbers mean nothing")
    return ap.parse_args()


def algorithmic_bytes_adam(B, V, d):
    """Keras-legacy Adam sweeps W, m, v (read + write) of both tables and bias vectors
    every step, independent of the batch, plus the nonzero stream."""
    return 16 * B + 2 * 24 * V * (d + 1)


def algorithmic_bytes(B, d, u_row, u_col):
    """16 B of (row, col, weight, value) per nonzero + read W, read A, write W,
    write A for every distinct touched row and its bias."""
    return 16 * B + 16 * (d + 1) * (u_row + u_col)


def measured_traffic(workload, B, cap):
    """HBM-side bytes per step from the committed PMC summary of this exact configuration
    (profiles/*_traffic.json, produced by tools/pmc_traffic.py from separate `rocprofv3 --pmc`
    passes of this bench); None when no summary matches."""
    import glob
    for f in sorted(glob.glob(str(REPO / "profiles" / "*_traffic.json")), reverse=True):
        try:
            j = json.load(open(f))
            m = j.get("meta", {})
            if m.get("workload") == workload and int(m.get("batch", -1)) == B and int(m.get("chunk_cap", cap)) == cap:
                return float(j["traffic_bytes_per_step"]), os.path.basename(f)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def cpu_baseline(workload, B, hp_kwargs, seconds, optimizer="Adagrad"):
    sys.path.insert(0, str(REPO / "oracle"))
    import numpy as np
    import glove_ref as ref
    import glove_ref_c
    t = ref.Tables(workload["V"], workload["d"], optimizer, dtype=np.float32, seed=1)
    port = glove_ref_c.CPort(t, B)
    hp = ref.Hyper(**hp_kwargs)
    row, col = workload["row"].cpu().numpy(), workload["col"].cpu().numpy()
    w, y = workload["w"].cpu().numpy(), workload["y"].cpu().numpy()
    nb = max(1, len(row) // B)
    port.step(row[:B], col[:B], w[:B], y[:B], hp)            # warm
    n, t0 = 0, time.perf_counter()
    while True:
        b = n % nb
        s = slice(b * B, (b + 1) * B)
        port.step(row[s], col[s], w[s], y[s], hp)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 2000:
            break
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": n * B / el, "unit": "nonzeros/s", "cores": 1, "kind": "port",
            "sample": "%d %s steps of %d nonzeros (same batches, oracle/glove_ref.c, -O2 scalar fp32)" % (n, optimizer, B),
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                             % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    dense = world > 1 or args.force_dense or args.row_sharded
    adam = args.optimizer == "Adam"
    if adam and dense:
        raise SystemExit("--optimizer Adam is benchmarked on one GPU")
    if dense:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:      # --force-dense on a single GPU without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from trainer import hip_api, synthetic
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
    from trainer.hip_api import DeviceTables, GloveHip, make_hyper

    hip = GloveHip(dev)
    B = args.batch_size
    wl = synthetic.make_workload(args.workload, seed=rank, device=dev, work_device=dev)
    V, d = wl["V"], wl["d"]
    V_row = V
    if args.row_sharded:
        from trainer.stepper import owned_rows, route_by_row_owner
        V_row = owned_rows(V, world, rank)
        wl.update(route_by_row_owner({k: wl[k] for k in ("row", "col", "w", "y")}, world, rank, dist))
    from trainer.hip_api import auto_chunk_cap
    cap = args.chunk_cap or auto_chunk_cap(B, V)
    nnz = wl["row"].numel()
    nb = min(max(1, nnz // B), args.max_batches)
    if nnz < B:
        raise SystemExit("workload has %d nonzeros < batch size %d" % (nnz, B))
    if dist is not None and world > 1:
        # every rank's shard has its own nonzero count: agree on the number of resident batches, so that all
        # ranks issue the same number of collectives in every loop below
        agreed = torch.tensor([nb], dtype=torch.int64, device=dev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        nb = int(agreed.item())

    # ---- load time (untimed): resident batches + their dedup index
    batches, plans = [], []
    for b in range(nb):
        s = slice(b * B, (b + 1) * B)
        batches.append(tuple(wl[k][s].contiguous() for k in ("row", "col", "w", "y")))
    t0 = time.perf_counter()
    for bt in batches:
        plans.append(hip.build_plan(*bt, V, chunk_cap=cap, compact=True))
    torch.cuda.synchronize()
    plan_build_ms = (time.perf_counter() - t0) * 1e3 / nb
    counts = [p.counts.tolist() for p in plans]
    n_heavy = sum(c[4] for c in counts) / nb
    u_row = sum(c[1] for c in counts) / nb
    u_col = sum(c[3] for c in counts) / nb
    chunks = sum(c[0] + c[2] for c in counts) / nb

    tables = DeviceTables(V, d, args.optimizer, device=dev, seed=1, V_row=V_row)      # identical replicas on every rank
    hyper = make_hyper(learning_rate=args.learning_rate, batch_size=B * world, step_form=args.step_form)
    ws = torch.empty(max(hip.lib.glove_step_workspace_bytes(B, p.cap_chunks, d) for p in plans) if not args.dynamic
                     else hip.lib.glove_step_workspace_bytes(B, B, d), dtype=torch.uint8, device=dev)
    loss_out = torch.zeros(4, device=dev)
    G = hip.dense_grad_buffer(tables) if dense or adam else None
    if args.row_sharded:
        hyper_rows = make_hyper(learning_rate=args.learning_rate, batch_size=B * world, sides=1)
        hyper_cols = make_hyper(learning_rate=args.learning_rate, batch_size=B * world, sides=2)
        G_col_half = G[hip.grad_layout(tables)["G_C"]:]

    staging = hip.build_plan(*batches[0], V, chunk_cap=cap) if args.dynamic else None   # refilled every step
    ahead = max(1, args.build_ahead) if args.dynamic else 1
    if ahead > 1:
        # a ring of staging plans, scratch buffers and streams: the index of batch i is built on stream i % ahead
        # while earlier steps run; it may start once step i - ahead, the previous reader of its staging plan, is done
        ring = [staging] + [hip.build_plan(*batches[0], V, chunk_cap=cap) for _ in range(ahead - 1)]
        ring_ws = [torch.empty(hip.lib.glove_plan_workspace_bytes(B, V), dtype=torch.uint8, device=dev) for _ in range(ahead)]
        ring_streams = [torch.cuda.Stream() for _ in range(ahead)]

    def sweep_pipelined(n_steps):
        """n_steps dynamic steps with `ahead` index builds in flight (call inside a graph capture or eagerly)."""
        main = torch.cuda.current_stream()
        built, stepped = [None] * n_steps, [None] * n_steps
        start = torch.cuda.Event()
        start.record(main)

        def launch_build(i):
            st = ring_streams[i % ahead]
            st.wait_event(stepped[i - ahead] if i >= ahead else start)
            with torch.cuda.stream(st):
                hip.build_plan(*batches[i % nb], V, chunk_cap=cap, into=ring[i % ahead], ws=ring_ws[i % ahead])
                built[i] = torch.cuda.Event()
                built[i].record(st)
        for i in range(min(ahead, n_steps)):
            launch_build(i)
        for i in range(n_steps):
            main.wait_event(built[i])
            hip.step_adagrad(ring[i % ahead], tables, hyper, loss_out, ws)
            stepped[i] = torch.cuda.Event()
            stepped[i].record(main)
            if i + ahead < n_steps:
                launch_build(i + ahead)

    def step(i):
        bt = batches[i % nb]
        plan = hip.build_plan(*bt, V, chunk_cap=cap, into=staging) if args.dynamic else plans[i % nb]
        if adam:
            hip.step_adam(plan, tables, hyper, G, loss_out, ws)
        elif not dense:
            hip.step_adagrad(plan, tables, hyper, loss_out, ws)
        elif args.row_sharded:
            hip.passes(plan, tables, hyper, ws)
            hip.dense_grad(plan, tables, hyper_cols, G, ws)
            hip.apply_adagrad(plan, tables, hyper_rows, None, ws)
            dist.all_reduce(G_col_half)
            hip.dense_adagrad(tables, hyper_cols, G, loss_out)
        else:
            hip.passes(plan, tables, hyper, ws)
            hip.dense_grad(plan, tables, hyper, G, ws)
            dist.all_reduce(G)
            hip.dense_adagrad(tables, hyper, G, loss_out)

    def barrier():
        if dense:
            dist.barrier()
        torch.cuda.synchronize()

    # One hipGraph holds whole sweeps over the resident batches (spg steps); it is replayed
    # steps // spg times and the remainder runs eagerly, so exactly `steps` steps are timed.
    use_graph = not dense and not args.no_graph
    graph = None
    spg = nb * ((16 + nb - 1) // nb)          # steps per graph: whole sweeps, at least 16 steps per replay
    if args.dynamic and args.build_ahead > 1:
        spg = nb * ((96 + nb - 1) // nb)      # the build pipeline fills and drains once per replay: longer graphs
    if use_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(nb):
                step(i)                       # warm the launch path on the capture stream
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            if ahead > 1:
                sweep_pipelined(spg)
            else:
                for i in range(spg):
                    step(i)

    def run(n_steps, first):
        done = 0
        if graph is not None:
            for _ in range(n_steps // spg):
                graph.replay()
            done = (n_steps // spg) * spg
        for i in range(done, n_steps):
            step(first + i)

    run(args.warmup, 0)
    barrier()
    t0 = time.perf_counter()
    run(args.steps, args.warmup)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    final_loss = float(loss_out[0].item())
    if not (final_loss == final_loss):
        raise SystemExit("loss is NaN")

    # ---- instrumented pass: HIP events (on the launch stream) around `reps` back-to-back launches of
    # each kernel of the step over the same resident batches; per-launch time = span / reps (it includes
    # the ~1-2 us launch-to-launch gap that rocprofv3's per-kernel durations exclude)
    kern = {}
    reps = max(nb, min(200, args.steps))
    calls = {"passes": lambda p: hip.passes(p, tables, hyper, ws)}      # row side + col side, one launch
    adam_fused = adam and 2 * B <= V_row + V          # glove_step_adam_f32's own rule (include/glove_hip.h)
    if adam_fused:
        # passes (+ id marks) and ONE fused apply/decay kernel; the second kernel has no entry point of its own,
        # so it is timed as the whole step minus the passes
        calls["step_adam"] = lambda p: hip.step_adam(p, tables, hyper, G, loss_out, ws)
    elif adam:
        calls["dense_grad"] = lambda p: hip.dense_grad(p, tables, hyper, G, ws)
        calls["dense_adam"] = lambda p: hip.dense_adam(tables, hyper, G, loss_out)
    elif not dense:
        calls["apply_adagrad"] = lambda p: hip.apply_adagrad(p, tables, hyper, loss_out, ws)
    else:
        if args.row_sharded:
            calls["dense_grad_cols"] = lambda p: hip.dense_grad(p, tables, hyper_cols, G, ws)
            calls["apply_adagrad_rows"] = lambda p: hip.apply_adagrad(p, tables, hyper_rows, None, ws)
            calls["all_reduce"] = lambda p: dist.all_reduce(G_col_half)    # RCCL over xGMI, broken out
            calls["dense_adagrad_cols"] = lambda p: hip.dense_adagrad(tables, hyper_cols, G, loss_out)
        else:
            calls["dense_grad"] = lambda p: hip.dense_grad(p, tables, hyper, G, ws)
            calls["all_reduce"] = lambda p: dist.all_reduce(G)             # RCCL over xGMI, broken out
            calls["dense_adagrad"] = lambda p: hip.dense_adagrad(tables, hyper, G, loss_out)
    for name, fn in calls.items():
        for i in range(2 * nb):
            fn(plans[i % nb])
        torch.cuda.synchronize()

        def burst(fn=fn):
            for i in range(reps):
                fn(plans[i % nb])
        if not dense:                         # replayed from a hipGraph, so that a kernel shorter than the host's
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
    if adam_fused:
        kern["adam_fused"] = kern.pop("step_adam") - kern["passes"]
    if G is not None:
        G.zero_()
    # attribution of the algorithmic bytes to the two kernels of the sparse step: the pass kernel
    # owns the nonzero stream and ONE read of every distinct row + bias; the apply kernel owns the accumulator
    # read and the two writes
    per_kernel = None
    if not dense and not adam:
        attributed = {"passes": 16 * B + 4 * (d + 1) * (u_row + u_col), "apply_adagrad": 12 * (d + 1) * (u_row + u_col)}
        per_kernel = {k: {"algorithmic_bytes": attributed[k], "avg_us": kern[k],
                          "achieved_GBps": attributed[k] / (kern[k] * 1e-6) / 1e9,
                          "frac": attributed[k] / (kern[k] * 1e-6) / 1e9 / HBM_PEAK_GBS} for k in attributed}
    step_us = sum(kern.values())
    alg = algorithmic_bytes_adam(B, V, d) if adam else algorithmic_bytes(B, d, u_row, u_col)
    achieved = alg / (step_us * 1e-6) / 1e9

    traffic, traffic_src = measured_traffic(args.workload, B, cap) if not dense else (None, None)

    if rank == 0:
        out = {
            "metric": "co-occurrence nonzeros/sec", "value": args.steps * B * world / elapsed, "unit": "nonzeros/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "V": V, "d": d, "optimizer": args.optimizer,
                       "batch_size_per_gpu": B, "global_batch": B * world, "nnz_per_gpu": nnz,
                       "resident_batches": nb, "chunk_cap": cap,
                       **({"rehearsal": "ranks share cuda:0 over gloo: control flow only, the numbers mean nothing"}
                          if args.rehearse_on_one_gpu else {}),
                       "index": ("rebuilt every step, %d builds in flight" % ahead if ahead > 1 else
                                 "rebuilt every step") if args.dynamic else "static, built at load",
                       "launch": "hipGraph replay" if graph is not None else "eager",
                       "parallelism": ("row-sharded x%d + col all-reduce" % world if args.row_sharded else
                                       "dp%d dense-grad all-reduce" % world if dense else "single GPU")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "one step = " + " + ".join(kern),
                         "algorithmic_bytes_per_step": alg, "kernel_us": kern, "per_kernel": per_kernel,
                         "heavy_ids_per_step": n_heavy, "uniq_rows_per_step": u_row, "uniq_cols_per_step": u_col, "chunks_per_step": chunks},
            "plan_build_ms_per_batch": plan_build_ms, "final_loss": final_loss,
        }
        if not args.no_cpu_baseline and world == 1 and not args.force_dense:
            try:
                out["cpu_baseline"] = cpu_baseline(wl, B, dict(learning_rate=args.learning_rate), args.cpu_seconds,
                                                   args.optimizer)
            except Exception as exc:                     # the baseline is a side measurement: never lose the bench line to it
                out["cpu_baseline"] = {"value": None, "unit": "nonzeros/s", "cores": 1, "kind": "port",
                                       "sample": "failed: %s: %s" % (type(exc).__name__, exc)}
        print(json.dumps(out), flush=True)
    if dense:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
