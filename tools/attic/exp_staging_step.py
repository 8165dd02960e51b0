#!/usr/bin/env python3
"""Step time on a resident (compacted, counts known on the host) plan against the same batch's device-refilled staging plan
(full capacity, counts never read back), without any rebuild in the timed loop: what the unknown counts alone cost."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from trainer import synthetic  # noqa: E402
from trainer.hip_api import DeviceTables, GloveHip, Plan, make_hyper  # noqa: E402

wl_name, B = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("zipf_v400k_d300", 1048576)
dev = torch.device("cuda:0")
lib = sys.argv[3] if len(sys.argv) > 3 else None
hip = GloveHip(dev, lib_path=lib) if lib else GloveHip(dev)
print("library:", lib or "shipped")
wl = synthetic.make_workload(wl_name, device=dev, work_device=dev)
V, d, nb, cap = wl["V"], wl["d"], 6, 32
batches = [tuple(wl[k][b * B:(b + 1) * B].contiguous() for k in ("row", "col", "w", "y")) for b in range(nb)]
kinds = {}
kinds["resident"] = [hip.build_plan(*bt, V, chunk_cap=cap, compact=True, d=d) for bt in batches]
kinds["staging+records"] = [hip.build_plan(*bt, V, chunk_cap=cap, records=True) for bt in batches]
forget = [hip.build_plan(*bt, V, chunk_cap=cap, compact=True, d=d) for bt in batches]
for p in forget:
    p.host_counts = [-1] * 8
    p._struct = None
kinds["resident, counts forgotten"] = forget
most_only = [hip.build_plan(*bt, V, chunk_cap=cap, compact=True, d=d) for bt in batches]
for p in most_only:
    p.host_counts[6] = -1
    p._struct = None
kinds["resident, most-chunks forgotten"] = most_only
loss = torch.zeros(4, device=dev)
ws = torch.empty(hip.lib.glove_step_workspace_bytes(B, B, (d + 3) // 4 * 4), dtype=torch.uint8, device=dev)
for twin in (True,):
    tables = DeviceTables(V, d, "Adagrad", device=dev, seed=1)
    if twin:
        tables.enable_twin()
    hyper = make_hyper(learning_rate=0.05, batch_size=B)
    for rnd in range(3):
        for name, plans in kinds.items():
            for i in range(3):
                hip.step_adagrad(plans[i % nb], tables, hyper, loss, ws)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(18):
                hip.step_adagrad(plans[i % nb], tables, hyper, loss, ws)
            b.record()
            torch.cuda.synchronize()
            if rnd:
                print("twin=%s %-34s %.1f us/step" % (twin, name, a.elapsed_time(b) * 1e3 / 18), flush=True)
