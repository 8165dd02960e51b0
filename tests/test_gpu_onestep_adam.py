"""The one-launch Keras-legacy Adam step (glove_step_adam_f32 on twinned tables with a plan that carries id bitmaps: every row of
both tables moves from the current copy to the other one in ONE launch — the batch's rows by the lane group that holds the id's
first chunk, all others by the sweep) against the float64 oracle and against the passes + adam_fused_kernel form.

The reference step it replaces: session.run(train_op) of reference src/models/estimator.py:48-56 at its defaults (Adam, 1,024
pairs: configs/app.ini:39-53; tf.keras.optimizers.get("Adam"), train_utils.py:13-16, whose sparse apply decays every row).
Tolerances as everywhere: loss rtol 1e-5, parameters and slots rtol 1e-5 / atol 1e-6; swept rows and ids of up
to heavy_chunks chunks come out bit-identical to the two-launch form."""
import numpy as np
import pytest
import torch

import glove_ref as ref
from helpers import assert_tables_close, make_batch, oracle_tables, tables_from_oracle, to_dev

pytestmark = pytest.mark.gpu


def _hyper(hp, B, form=0):
    from trainer.hip_api import make_hyper
    return make_hyper(l2_reg=hp.l2_reg, reg_mult=hp.reg_mult, learning_rate=hp.learning_rate, epsilon=hp.epsilon, batch_size=B,
                      head=hp.head, neg_factor=hp.neg_factor, step_form=form)


def _mid_run(V, d, seed=5, step=17):
    """Adam tables in a mid-run state: m, v of every row non-zero, so that the sweep's decay shows."""
    t = oracle_tables(V, d, "Adam")
    rng = np.random.default_rng(seed)
    for n in ("R", "C", "br", "bc"):
        setattr(t, "M_" + n, rng.normal(0, 1e-3, getattr(t, n).shape).astype(np.float32).astype(np.float64))
        setattr(t, "V_" + n, (rng.uniform(0, 1e-5, getattr(t, n).shape)).astype(np.float32).astype(np.float64))
    t.step = step
    return t


def _twinned(t):
    from trainer.hip_api import DeviceTables
    dt = tables_from_oracle(t, DeviceTables)
    dt.enable_tags()
    assert dt.R_tag is not None and dt._R.shape[0] == 2 * dt.V_row
    return dt


CASES = [(1024, 10000, 64, 16), (1024, 2000, 50, 16), (512, 1000, 300, 32), (2048, 3000, 128, 32), (1, 10, 64, 16), (20, 50, 16, 2),
         (700, 1000, 16, 2), (100, 300, 1024, 7), (1024, 1024, 8, 16), (333, 40000, 32, 16)]


@pytest.mark.parametrize("B,V,d,cap", CASES)
def test_one_launch_adam_single_steps(hip, B, V, d, cap):
    """Two steps (the second reads the second copies) == the oracle; == the two-launch form bit for bit on every swept row and
    every id the apply sums chunk by chunk; loss scalars, global bias and global_step as the two-launch form leaves them."""
    from trainer.hip_api import DeviceTables
    hp = ref.Hyper(learning_rate=0.001)
    row, col, w, y = make_batch(B + V + d, B, V)
    if B >= 20:
        row[::3] = 3                                # one id of many chunks
    t = _mid_run(V, d)
    a, b = _twinned(t), tables_from_oracle(t, DeviceTables)
    plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap, records=True)
    assert plan.r_mark is not None
    Ga, Gb = hip.dense_grad_buffer(a), hip.dense_grad_buffer(b)
    la, lb = torch.zeros(4, device="cuda:0"), torch.zeros(4, device="cuda:0")
    for k in range(2):
        hip.step_adam(plan, a, _hyper(hp, B, 0), Ga, la)
        assert float(a.scalars[3]) == float((k + 1) % 2) and a._twin_dirty          # the form really ran: the tables flipped
        hip.step_adam(plan, b, _hyper(hp, B, 1), Gb, lb)
        loss, L, reg = ref.train_step(t, row, col, w, y, hp)
        np.testing.assert_allclose(la.cpu().numpy()[:3], [loss, L, reg], rtol=1e-5)
    hip.step_adam(plan, a, _hyper(hp, B, 0), Ga, la)                                  # a third one: read back from the second copies
    hip.step_adam(plan, b, _hyper(hp, B, 1), Gb, lb)
    ref.train_step(t, row, col, w, y, hp)
    assert float(a.scalars[3]) == 1.0
    assert_tables_close(a, t, 2e-5, 2e-6)                                             # (reading brings the rows home)
    assert float(a.scalars[3]) == 0.0 and int((a.R_tag != 0).sum()) == 0
    assert float(Ga.abs().max()) == 0.0
    want = ref.build_plan(row, col, cap, V=V)
    if int(want["r_uniq_rec"][:, 2].max()) == 1:
        np.testing.assert_array_equal(la.cpu().numpy(), lb.cpu().numpy())
        assert torch.equal(a.scalars, b.scalars)
    np.testing.assert_allclose(la.cpu().numpy(), lb.cpu().numpy(), rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(a.scalars.cpu().numpy(), b.scalars.cpu().numpy(), rtol=2e-6, atol=1e-9)
    assert a.global_step == b.global_step == 20
    for side, name, bias in (("r", "R", "br"), ("c", "C", "bc")):
        rec = want[side + "_uniq_rec"]
        same = torch.ones(V, dtype=torch.bool, device="cuda:0")
        same[torch.from_numpy(rec[rec[:, 2] > 8, 0].astype(np.int64)).cuda()] = False      # ids the two-launch form hands to its heavy path
        assert torch.equal(getattr(a, name)[same], getattr(b, name)[same]), name
        assert torch.equal(getattr(a, bias)[same], getattr(b, bias)[same]), bias
        assert torch.equal(a.s1[name][same], b.s1[name][same]) and torch.equal(a.s2[name][same], b.s2[name][same]), name
        assert torch.equal(a.s1[bias][same], b.s1[bias][same]) and torch.equal(a.s2[bias][same], b.s2[bias][same]), bias


@pytest.mark.parametrize("B,V,d,head", [(1024, 4000, 64, 0), (512, 600, 300, 0), (30, 64, 32, 0), (1024, 4000, 64, 1)])
def test_one_launch_adam_trajectory_and_repeatability(hip, B, V, d, head):
    """40 steps on fresh batches (AUTO picks the form on twinned tables): within tolerance of the oracle all along, bitwise
    repeatable run to run, and readable in between (a read of R brings the tables home; the next step flips them again)."""
    hp = ref.Hyper(learning_rate=0.01, head=head, neg_factor=0.7)
    runs = []
    for rep in range(2):
        t = oracle_tables(V, d, "Adam")
        a = _twinned(t)
        G = hip.dense_grad_buffer(a)
        la = torch.zeros(4, device="cuda:0")
        for s in range(40):
            row, col, w, y = make_batch(3000 + s, B, V)
            if head == 1:
                y = np.abs(y) * 0.1
            plan = hip.build_plan(*to_dev(row, col, w, y), V)
            assert plan.r_crec is not None and plan.c_mark is not None
            hip.step_adam(plan, a, _hyper(hp, B, 0), G, la)
            if rep == 0:
                loss, _, _ = ref.train_step(t, row, col, w, y, hp)
                np.testing.assert_allclose(la[0].item(), loss, rtol=5e-5, err_msg="step %d" % s)
            if s == 16:
                assert float(a.scalars[3]) == 1.0
                _ = a.R.sum().item()
                assert float(a.scalars[3]) == 0.0
        if rep == 0:
            assert_tables_close(a, t, 2e-4, 1e-5)
        runs.append(a)
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(runs[0], n), getattr(runs[1], n)), n
        assert torch.equal(runs[0].s1[n], runs[1].s1[n]) and torch.equal(runs[0].s2[n], runs[1].s2[n]), n
    assert torch.equal(runs[0].scalars, runs[1].scalars)


@pytest.mark.parametrize("B,V,d,n", [(1024, 3000, 64, 13), (256, 400, 16, 40), (2048, 3000, 300, 5), (1024, 10000, 64, 64)])
def test_chained_adam_steps_equal_single_ones(hip, B, V, d, n):
    """glove_steps_adam_f32 on twinned tables: n steps as a chain (one launch per step; every workgroup derives the step's
    global bias and its moments from the record the step before left; a workspace too small for n records makes several chains)
    == the same steps one by one bit for bit, == the two-launch form on plain tables within tolerance; loss scalars are the
    LAST step's, global_step advances by n, an odd n leaves the second copies current."""
    from trainer.hip_api import DeviceTables
    hp = ref.Hyper(learning_rate=0.01)
    t = _mid_run(V, d, seed=9, step=3)
    a, b, c = _twinned(t), _twinned(t), tables_from_oracle(t, DeviceTables)
    plans = [hip.build_plan(*to_dev(*make_batch(900 + k, B, V)), V) for k in range(n)]
    h = _hyper(hp, B, 0)
    G = hip.dense_grad_buffer(a)
    la, lb, lc = (torch.zeros(4, device="cuda:0") for _ in range(3))
    hip.steps_adam(plans, a, h, G, la)
    for p in plans:
        hip.step_adam(p, b, h, G, lb)
        hip.step_adam(p, c, _hyper(hp, B, 1), G, lc)
    assert float(a.scalars[3]) == float(n % 2) == float(b.scalars[3])
    assert a.global_step == b.global_step == c.global_step == 3 + n
    assert torch.equal(la, lb)
    for m in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(a, m), getattr(b, m)), m
        assert torch.equal(a.s1[m], b.s1[m]) and torch.equal(a.s2[m], b.s2[m]), m
    assert torch.equal(a.scalars, b.scalars)
    np.testing.assert_allclose(la.cpu().numpy(), lc.cpu().numpy(), rtol=1e-4)
    for m in ("R", "C", "br", "bc"):
        np.testing.assert_allclose(getattr(a, m).cpu().numpy(), getattr(c, m).cpu().numpy(), rtol=1e-4, atol=1e-6)
    # a small workspace: chains of a few steps each
    a2 = _twinned(t)
    small = torch.empty(3 * (8 + 4 * 64) * 4, dtype=torch.uint8, device="cuda:0")
    try:
        hip.steps_adam(plans, a2, h, G, la, ws=small)
    except Exception:
        small = torch.empty(3 * (8 + 4 * 1024) * 4, dtype=torch.uint8, device="cuda:0")
        a2 = _twinned(t)
        hip.steps_adam(plans, a2, h, G, la, ws=small)
    for m in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(a2, m), getattr(b, m)), m
    assert torch.equal(a2.scalars, b.scalars)


def test_one_launch_adam_steps_replayed_from_a_hipgraph(hip):
    """A chain of 8 one-launch Adam steps captured once and replayed 8 times == the same 64 steps launched one by one, bit
    for bit; global_step and the current copy (scalars[3]) carry over from replay to replay on the device."""
    B, V, d = 1024, 6000, 64
    hp = ref.Hyper(learning_rate=0.01)
    t = oracle_tables(V, d, "Adam")
    a, b = _twinned(t), _twinned(t)
    plans = [hip.build_plan(*to_dev(*make_batch(50 + k, B, V)), V) for k in range(9)]
    h = _hyper(hp, B, 0)
    G = hip.dense_grad_buffer(a)
    la, lb = torch.zeros(4, device="cuda:0"), torch.zeros(4, device="cuda:0")
    ws = torch.empty(max(hip.lib.glove_step_workspace_bytes(B, p.cap_chunks, d) for p in plans), dtype=torch.uint8, device="cuda:0")
    hip.step_adam(plans[8], a, h, G, la, ws)            # (an odd number of steps in front: the replays start on the second copies)
    hip.step_adam(plans[8], b, h, G, lb, ws)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        hip.steps_adam(plans[:8], a, h, G, la, ws=ws)
    for _ in range(8):
        g.replay()
    for _ in range(8):
        for k in range(8):
            hip.step_adam(plans[k], b, h, G, lb, ws)
    assert a.global_step == b.global_step == 65
    assert float(a.scalars[3]) == 1.0
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n
        assert torch.equal(a.s1[n], b.s1[n]) and torch.equal(a.s2[n], b.s2[n]), n
    assert torch.equal(la, lb)


def test_forms_that_do_not_fit_fall_back(hip):
    """A batch that touches most rows, a plan without bitmaps, step_form = two launches: the step brings the tables home and
    takes the older forms — same results as on plain tables, bit for bit."""
    from trainer.hip_api import DeviceTables
    hp = ref.Hyper(learning_rate=0.01)
    for B, V, form, strip in ((1024, 300, 0, False), (512, 4000, 1, False), (512, 4000, 0, True)):
        t = _mid_run(V, 64)
        a, b = _twinned(t), tables_from_oracle(t, DeviceTables)
        G = hip.dense_grad_buffer(a)
        la, lb = torch.zeros(4, device="cuda:0"), torch.zeros(4, device="cuda:0")
        for k in range(3):
            plan = hip.build_plan(*to_dev(*make_batch(70 + k, B, V)), V)
            if strip:
                plan.r_mark = plan.c_mark = None
                plan._struct = None
            hip.step_adam(plan, a, _hyper(hp, B, form), G, la)
            assert float(a.scalars[3]) == 0.0
            hip.step_adam(plan, b, _hyper(hp, B, 1), G, lb)
        for n in ("R", "C", "br", "bc"):
            assert torch.equal(getattr(a, n), getattr(b, n)), n
        assert torch.equal(la, lb)


def test_runner_steps_adam_in_one_launch(hip):
    """The reshuffling runner on the reference's default shape (Adam, 1,024 pairs, V = 10 k): twins the tables, chains the
    steps of a segment; two runs agree bit for bit, and with a runner forced to the two-launch form within tolerance."""
    from trainer.data_utils import NonzeroStream
    from trainer.hip_api import DeviceTables, make_hyper
    from trainer.stepper import HipBackend, ReshufflingRunner
    V, d, B, nnz = 10000, 64, 1024, 1024 * 24 + 77
    rng = np.random.default_rng(3)
    data = {"row": rng.integers(0, V, nnz).astype(np.int32), "col": rng.integers(0, V, nnz).astype(np.int32),
            "w": rng.uniform(0.1, 1, nnz).astype(np.float32), "y": rng.normal(0, 1, nnz).astype(np.float32)}
    dev = torch.device("cuda:0")
    out = []
    for form in (0, 0, 1):
        backend = HipBackend(dev)
        backend.hip = hip
        tables = DeviceTables(V, d, "Adam", device=dev, seed=1)
        stream = NonzeroStream(dict(data), B, V, backend, dev, seed=11, static_plans=False)
        runner = ReshufflingRunner(hip, stream, tables, make_hyper(batch_size=B, learning_rate=0.01, step_form=form))
        assert (tables.R_tag is not None) == (form == 0)
        done = 0
        while done < 60:
            done += runner.run(60 - done)
        torch.cuda.synchronize()
        out.append((tables, runner.read_loss()["loss"]))
        runner.release_graphs()
    (a, la), (b, lb), (c, lc) = out
    assert a.global_step == b.global_step == c.global_step == 60
    assert la == lb and abs(la - lc) <= 1e-4 * abs(lc)
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n
        np.testing.assert_allclose(getattr(a, n).cpu().numpy(), getattr(c, n).cpu().numpy(), rtol=1e-4, atol=1e-6)
