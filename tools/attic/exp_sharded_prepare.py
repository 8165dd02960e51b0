import sys, time
sys.path.insert(0, "/root/repo")
import torch
from trainer import synthetic
from trainer.hip_api import DeviceTables, GloveHip
from trainer.stepper import HipBackend, ShardedStepper
wl_name = sys.argv[1] if len(sys.argv) > 1 else "zipf_v400k_d300"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
dev = torch.device("cuda:0")
hip = GloveHip(dev)
wl = synthetic.make_workload(wl_name, device=dev, work_device=dev)
V, d = wl["V"], wl["d"]
t = DeviceTables(V, d, "Adagrad", device=dev, seed=1)
backend = HipBackend(dev); backend.hip = hip; backend.row_floats = t.d
st = ShardedStepper(backend, t, dict(learning_rate=0.05), B, 1, 0, None, exercise_exchange=True)
bts = [tuple(wl[k][b * B:(b + 1) * B].contiguous() for k in ("row", "col", "w", "y")) for b in range(6)]
for rep in range(2):
    st.clear_batches()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hs = [st.add_batch(*bt, 0) for bt in bts]
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("add_batch: %.2f ms per batch" % ((t1 - t0) * 1e3 / len(bts)))
for h in hs[:2]:
    st.step(h)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    for h in hs:
        st.step(h)
torch.cuda.synchronize(); t1 = time.perf_counter()
print("step: %.2f ms" % ((t1 - t0) * 1e3 / (3 * len(hs))))
