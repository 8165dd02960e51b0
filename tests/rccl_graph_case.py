"""TEST CASE run in a process of its own by tests/test_gpu_trainer.py (a second RCCL process group inside the pytest process,
behind the one an earlier test made and destroyed, hung; and a hang here must fail one test, not the suite)."""
import faulthandler
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
for _p in (REPO, REPO / "oracle", REPO / "tests"):
    sys.path.insert(0, str(_p))
import numpy as np      # noqa: E402
import torch            # noqa: E402
import glove_ref as ref  # noqa: E402


def run(hip):

    import os
    import torch.distributed as dist
    from helpers import make_batch, tables_from_oracle, to_dev
    from trainer.data_utils import NonzeroStream
    from trainer.hip_api import DeviceTables, make_hyper
    from trainer.stepper import HipBackend, ReshufflingRunner, RowShardedStepper, ShardedStepper, Stepper
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29549", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    try:
        B, V, d, nb, rounds = 6000, 700, 64, 3, 4
        backend = HipBackend("cuda:0")
        t = ref.Tables(V, d, "Adagrad", dtype=np.float32, seed=4).astype(np.float64)
        kw = dict(l2_reg=0.01, reg_mult=2.0, learning_rate=0.05)
        batches = [to_dev(*make_batch(40 + s, B, V)) for s in range(nb)]
        plans = [backend.build_plan(*bt, V, 0).compact(hip.lib, d) for bt in batches]

        def make(form):
            tabs = tables_from_oracle(t, DeviceTables)
            if form == "dp rows" or form == "dp dense":
                st = Stepper(backend, tabs, kw, B, world=1, dist=dist, exchange=form.split()[1], collectives=True)
                st.prepare(plans)
                return tabs, st, plans
            if form.startswith("row-sharded"):
                st = RowShardedStepper(backend, tabs, kw, B, 1, dist, exchange=form.split()[1], collectives=True)
                st.prepare(plans)
                return tabs, st, plans
            st = ShardedStepper(backend, tabs, kw, B, 1, 0, dist, collectives=True, exercise_exchange=True)
            return tabs, st, [st.add_batch(*bt) for bt in batches]
        for form in ("dp dense", "dp rows", "row-sharded rows", "row-sharded dense", "both tables sharded"):
            (ta, sa, ia), (tb, sb, ib) = make(form), make(form)
            sb.enable_graphs(after=1)
            assert sb._graphs is not None, form
            for rnd in range(rounds):
                for k in range(nb):
                    sa.step(ia[k])
                    sb.step(ib[k])                      # round 0 eager, round 1 captures and replays, then replays
                if rnd == 1:
                    assert len(sb._graphs) == nb, form
            for n in ("R", "C", "br", "bc"):
                assert torch.equal(getattr(ta, n), getattr(tb, n)), (form, n)
                assert torch.equal(ta.s1[n], tb.s1[n]), (form, n)
            assert torch.equal(ta.scalars, tb.scalars) and ta.global_step == tb.global_step == rounds * nb, form
            assert torch.equal(sa.loss_out, sb.loss_out), form
        # ---- reshuffled epochs: single-GPU runner == data-parallel runner through RCCL, graphs on == graphs off
        Br = 1000
        coo = {k: v for k, v in zip(("row", "col", "w", "y"), make_batch(7, 5 * Br + 123, V))}
        outs, alive = [], []
        for mode in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("single", "dp eager", "dp graphs", "row-sharded graphs")):
            tabs = tables_from_oracle(t, DeviceTables)
            stream = NonzeroStream(coo, Br, V, backend, "cuda:0", seed=3, static_plans=False)
            if mode == "single":
                # (two-launch form: what the dense data-parallel form reproduces bit for bit on one rank; the single-GPU default
                # at this batch size, the tagged step, sums an id of more than heavy_chunks chunks in another order)
                runner = ReshufflingRunner(hip, stream, tabs, make_hyper(batch_size=Br, step_form=1, **kw), burst=4, segment=2)
            else:
                cls = RowShardedStepper if mode.startswith("row") else Stepper
                st = cls(backend, tabs, kw, Br, 1, dist, exchange="dense" if mode.startswith("dp") else "rows", collectives=True)
                st.prepare(batch_size=Br)
                runner = ReshufflingRunner(hip, stream, tabs, st.hyper, burst=4, segment=2, stepper=st, graphs=mode.endswith("graphs"))
                assert runner.graphs_on == mode.endswith("graphs")
            print("reshuffled epochs:", mode, flush=True)
            done = 0
            while done < 23:                             # four and a half epochs of five batches, bursts of up to four
                done += runner.run(23 - done)
            outs.append((mode, tabs, runner.read_loss()))
            # The runners are deliberately KEPT ALIVE with their graphs while the next ones are built and replayed.  (Round 3:
            # with the first runner's graphs alive — bursts that captured index builds on forked side streams — the replay
            # of the fourth runner's graph segfaulted inside hipGraphLaunch; a runner's graphs now hold steps on one stream
            # only, the builds are ordinary launches: DESIGN.md §7.)
            alive.append(runner)
        base = outs[0]
        for mode, tabs, loss in outs[1:]:
            exact = mode.startswith("dp")                # the dense data-parallel form on one rank sums in the sparse step's order
            for n in ("R", "C", "br", "bc"):
                if exact:
                    assert torch.equal(getattr(tabs, n), getattr(base[1], n)), (mode, n)
                else:
                    torch.testing.assert_close(getattr(tabs, n), getattr(base[1], n), rtol=5e-5, atol=5e-6)
            assert abs(loss["loss"] - base[2]["loss"]) <= 2e-5 * abs(base[2]["loss"]), (mode, loss, base[2])
        assert torch.equal(outs[1][1].R, outs[2][1].R) and torch.equal(outs[1][1].C, outs[2][1].C)      # graphs on == off
    finally:
        # RCCL does not finish tearing a communicator down while hipGraphs that captured its collectives exist
        import gc
        for obj in gc.get_objects():
            if isinstance(obj, (Stepper, RowShardedStepper, ShardedStepper, ReshufflingRunner)):
                obj.release_graphs()
        gc.collect()
        dist.destroy_process_group()


if __name__ == "__main__":
    faulthandler.dump_traceback_later(200, exit=True)
    from trainer.hip_api import GloveHip
    run(GloveHip("cuda:0"))
    print("rccl graph case ok", flush=True)
