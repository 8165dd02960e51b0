import sys, torch
sys.path.insert(0, '.')
from trainer import synthetic
from trainer.hip_api import DeviceTables, GloveHip, make_hyper
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wl = synthetic.make_workload("text8_d64", device="cuda:0", work_device="cuda:0")
V, d = wl["V"], wl["d"]
for name in [None]:
    hip = GloveHip("cuda:0", lib_path=name)
    for form, tags in ((1, False), (5, True)):
        t = DeviceTables(V, d, "Adagrad", seed=1)
        if tags: t.enable_tags()
        plans = [hip.build_plan(*(wl[k][b*B:(b+1)*B].contiguous() for k in ("row","col","w","y")), V, records=True) for b in range(NP)]
        h = make_hyper(learning_rate=0.05, batch_size=B, step_form=form)
        loss = torch.zeros(4, device="cuda:0")
        ws = torch.empty(hip.lib.glove_step_workspace_bytes(B, B, d), dtype=torch.uint8, device="cuda:0")
        for p in plans[:4]: hip.step_adagrad(p, t, h, loss, ws)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            hip.steps_adagrad(plans, t, h, loss, ws=ws)
        for _ in range(3): g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): g.replay()
        b.record(); torch.cuda.synchronize()
        print(name or "shipped", "form", form, "B", B, "%.2f us/step" % (a.elapsed_time(b) * 1e3 / (20 * 64)), "loss", float(loss[0]))
