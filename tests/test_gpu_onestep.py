"""The tagged sparse Adagrad step (GLOVE_STEP_TAGGED: step-tagged twinned tables, one launch for all the row work + a
one-workgroup scalar epilogue; an id's chunks all done by the lane group that holds its first one) against the float64 oracle and against the two-launch form.

The reference step it replaces: session.run(train_op) of reference src/models/estimator.py:48-56 at the default batch of 1,024
pairs (configs/app.ini:39-53), Keras-legacy Adagrad (train_utils.py:13-16).  Tolerances as everywhere: loss rtol
1e-5, parameters and slots rtol 1e-5 / atol 1e-6; an id one chunk holds comes out bit-identical to the two-launch form."""
import numpy as np
import pytest
import torch

import glove_ref as ref
from helpers import assert_tables_close, make_batch, oracle_tables, tables_from_oracle, to_dev

pytestmark = pytest.mark.gpu


def _hyper(hp, B, form):
    from trainer.hip_api import make_hyper
    return make_hyper(l2_reg=hp.l2_reg, reg_mult=hp.reg_mult, learning_rate=hp.learning_rate, epsilon=hp.epsilon, batch_size=B,
                      head=hp.head, neg_factor=hp.neg_factor, step_form=form)


def _tagged(t):
    from trainer.hip_api import DeviceTables
    dt = tables_from_oracle(t, DeviceTables)
    dt.enable_tags()
    assert dt.R_tag is not None
    return dt


CASES = [(1024, 10000, 64, 16), (1024, 300, 64, 16), (4096, 12000, 64, 16), (700, 50, 16, 2), (3000, 40, 16, 5), (2048, 500, 128, 32),
         (512, 100, 300, 32), (1024, 2000, 50, 16), (1, 10, 64, 16), (4000, 7, 32, 16), (1024, 10000, 8, 16), (900, 77, 1024, 7)]


@pytest.mark.parametrize("B,V,d,cap", CASES)
def test_one_launch_single_step(hip, B, V, d, cap):
    """One step in one launch == the oracle; every id that one chunk holds == the two-launch form bit for bit (rows, biases,
    accumulators); loss scalars, global bias and global_step as the two-launch form leaves them."""
    from trainer.hip_api import DeviceTables
    hp = ref.Hyper(learning_rate=0.05)
    row, col, w, y = make_batch(B + V + d, B, V)
    t = oracle_tables(V, d, "Adagrad")
    a, b = _tagged(t), tables_from_oracle(t, DeviceTables)
    plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap, records=True)
    la, lb = torch.zeros(4, device="cuda:0"), torch.zeros(4, device="cuda:0")
    hip.step_adagrad(plan, a, _hyper(hp, B, 5), la)
    assert int((a.R_tag != 0).sum()) == int(plan.counts[1]) and int((a.C_tag != 0).sum()) == int(plan.counts[3])    # the form really ran
    hip.step_adagrad(plan, b, _hyper(hp, B, 1), lb)
    loss, L, reg = ref.train_step(t, row, col, w, y, hp)
    np.testing.assert_allclose(la.cpu().numpy()[:3], [loss, L, reg], rtol=1e-5)
    assert_tables_close(a, t, 1e-5, 1e-6)
    want = ref.build_plan(row, col, cap, V=V)
    if int(want["r_uniq_rec"][:, 2].max()) == 1:                             # every row id in one chunk: the loss partials add up in the same order
        np.testing.assert_array_equal(la.cpu().numpy(), lb.cpu().numpy())
        assert torch.equal(a.scalars, b.scalars)
    np.testing.assert_allclose(la.cpu().numpy(), lb.cpu().numpy(), rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(a.scalars.cpu().numpy(), b.scalars.cpu().numpy(), rtol=2e-6, atol=1e-9)
    assert a.global_step == b.global_step == 1
    for side, name, bias in (("r", "R", "br"), ("c", "C", "bc")):
        rec = want[side + "_uniq_rec"]
        one = torch.from_numpy(rec[rec[:, 2] <= 8, 0].astype(np.int64)).cuda()   # ids the two-launch form's apply sums chunk by chunk (up to heavy_chunks)
        assert torch.equal(getattr(a, name)[one], getattr(b, name)[one]), name
        assert torch.equal(getattr(a, bias)[one], getattr(b, bias)[one]) and torch.equal(a.s1[name][one], b.s1[name][one]), name
        untouched = torch.ones(V, dtype=torch.bool, device="cuda:0")
        untouched[torch.from_numpy(rec[:, 0].astype(np.int64)).cuda()] = False
        assert torch.equal(getattr(a, name)[untouched], getattr(b, name)[untouched])


@pytest.mark.parametrize("B,V,d,head", [(1024, 400, 64, 0), (512, 100, 300, 0), (2048, 30, 32, 0), (1024, 400, 64, 1)])
def test_one_launch_trajectory_and_repeatability(hip, B, V, d, head):
    """40 steps on fresh batches through the one-launch form (AUTO picks it on tagged tables), rows moving between their two
    copies: within tolerance of the oracle all along, bitwise repeatable run to run, and readable in between (an eval pass
    and a read of R bring the tables home; the next step tags them again)."""
    hp = ref.Hyper(learning_rate=0.05, head=head, neg_factor=0.7)
    runs = []
    for rep in range(2):
        t = oracle_tables(V, d, "Adagrad")
        a = _tagged(t)
        la = torch.zeros(4, device="cuda:0")
        for s in range(40):
            row, col, w, y = make_batch(3000 + s, B, V)
            if head == 1:
                y = np.abs(y) * 0.1
            plan = hip.build_plan(*to_dev(row, col, w, y), V)
            assert plan.r_crec is not None
            hip.step_adagrad(plan, a, _hyper(hp, B, 0), la)
            if rep == 0:
                loss, _, _ = ref.train_step(t, row, col, w, y, hp)
                np.testing.assert_allclose(la[0].item(), loss, rtol=2e-5)
                if s == 17:
                    assert getattr(a, "_twin_dirty", False)
                    _ = a.R.sum().item()                                    # reading brings the tables home (tags cleared)
                    assert int((a.R_tag != 0).sum()) == 0 and int((a.C_tag != 0).sum()) == 0
            elif s == 17:
                _ = a.R.sum().item()
        if rep == 0:
            assert_tables_close(a, t, 4e-5, 4e-6)
        runs.append(a)
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(runs[0], n), getattr(runs[1], n)), n
        assert torch.equal(runs[0].s1[n], runs[1].s1[n]), n
    assert torch.equal(runs[0].scalars, runs[1].scalars)


def test_one_launch_steps_replayed_from_a_hipgraph(hip):
    """64 one-launch steps captured once and replayed == the same steps launched one by one, bit for bit; global_step, the
    arrival counters and the tags carry over from replay to replay on the device."""
    from trainer.hip_api import DeviceTables
    B, V, d = 1024, 3000, 64
    hp = ref.Hyper(learning_rate=0.05)
    t = oracle_tables(V, d, "Adagrad")
    a, b = _tagged(t), _tagged(t)
    plans = [hip.build_plan(*to_dev(*make_batch(50 + k, B, V)), V) for k in range(8)]
    h = _hyper(hp, B, 0)
    la, lb = torch.zeros(4, device="cuda:0"), torch.zeros(4, device="cuda:0")
    ws = torch.empty(max(hip.lib.glove_step_workspace_bytes(B, p.cap_chunks, d) for p in plans), dtype=torch.uint8, device="cuda:0")
    hip.step_adagrad(plans[0], a, h, la, ws)
    hip.step_adagrad(plans[0], b, h, lb, ws)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        hip.steps_adagrad(plans, a, h, la, ws=ws)           # a chain: one launch per step + one epilogue
    for _ in range(8):
        g.replay()
    for _ in range(8):
        for k in range(8):
            hip.step_adagrad(plans[k], b, h, lb, ws)        # chains of one
    assert a.global_step == b.global_step == 65
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n
    assert torch.equal(la, lb)


@pytest.mark.parametrize("B,V,d,n", [(1024, 3000, 64, 13), (256, 200, 16, 40), (2048, 900, 300, 5)])
def test_chained_tagged_steps_equal_single_ones(hip, B, V, d, n):
    """glove_steps_adagrad_f32 on step-tagged tables: n steps as a chain (one launch per step: every workgroup derives the
    step's global bias from the record the step before left in the workspace; a workspace too small for n records makes several
    chains) == the same steps one by one, and == the two-launch form on plain tables within tolerance; the loss scalars are
    the LAST step's, global_step advances by n."""
    from trainer.hip_api import DeviceTables
    hp = ref.Hyper(learning_rate=0.05)
    t = oracle_tables(V, d, "Adagrad")
    a, b, c = _tagged(t), _tagged(t), tables_from_oracle(t, DeviceTables)
    plans = [hip.build_plan(*to_dev(*make_batch(80 + k, B, V)), V, records=True) for k in range(n)]
    h = _hyper(hp, B, 0)
    la, lb, lc = (torch.zeros(4, device="cuda:0") for _ in range(3))
    full = max(hip.lib.glove_step_workspace_bytes(B, p.cap_chunks, d) for p in plans)
    ws = torch.empty(full, dtype=torch.uint8, device="cuda:0")
    small = torch.empty(3 * (8 + 4 * ((B + 7) // 8 + 1)) * 4, dtype=torch.uint8, device="cuda:0")      # room for about three records
    hip.steps_adagrad(plans, a, h, la, ws=ws)
    hip.steps_adagrad(plans, b, h, lb, ws=small)
    for p in plans:
        hip.step_adagrad(p, c, _hyper(hp, B, 1), lc)
    assert a.global_step == b.global_step == c.global_step == n
    for name in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
        np.testing.assert_allclose(getattr(a, name).cpu().numpy(), getattr(c, name).cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg=name)
    assert torch.equal(a.scalars, b.scalars) and torch.equal(la, lb)
    np.testing.assert_allclose(la.cpu().numpy(), lc.cpu().numpy(), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(a.scalars.cpu().numpy(), c.scalars.cpu().numpy(), rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("optimizer", ["Adagrad", "Adam"])
def test_auto_leaves_a_skewed_batch_to_the_two_launch_form(hip, optimizer):
    """One token holding half of the batch's pairs (a frequent word): the one-launch forms would walk its 32 chunks in ONE lane
    group, one record trip after the other, and sum them in another order than the two-launch form.  GLOVE_STEP_AUTO therefore takes
    them only for batches without heavy ids (plan.host_counts[4] == 0 where the host knows it): here the step on tagged tables
    is the two-launch form's, bit for bit, and equals the oracle; a balanced batch beside it still takes one launch."""
    from trainer.hip_api import DeviceTables, make_hyper
    B, V, d, cap = 1024, 2000, 64, 16
    hp = ref.Hyper(learning_rate=0.05 if optimizer == "Adagrad" else 0.001)
    row, col, w, y = make_batch(77, B, V, zipf=False)
    row = row.copy()
    row[::2] = 5                                               # id 5 holds every second pair: 512 pairs = 32 chunks
    t = oracle_tables(V, d, optimizer)
    a, b = tables_from_oracle(t, DeviceTables), tables_from_oracle(t, DeviceTables)
    a.enable_tags()
    kw = dict(l2_reg=hp.l2_reg, reg_mult=hp.reg_mult, learning_rate=hp.learning_rate, epsilon=hp.epsilon, batch_size=B)
    plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap, records=True).compact(hip.lib, records=True)
    assert plan.host_counts[4] >= 1 and plan.host_counts[6] >= 32
    la, lb = torch.zeros(4, device="cuda:0"), torch.zeros(4, device="cuda:0")
    if optimizer == "Adagrad":
        hip.step_adagrad(plan, a, make_hyper(step_form=0, **kw), la)
        hip.step_adagrad(plan, b, make_hyper(step_form=1, **kw), lb)
    else:
        Ga, Gb = hip.dense_grad_buffer(a), hip.dense_grad_buffer(b)
        hip.step_adam(plan, a, make_hyper(step_form=0, **kw), Ga, la)
        hip.step_adam(plan, b, make_hyper(step_form=1, **kw), Gb, lb)
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n
        assert torch.equal(a.s1[n], b.s1[n]), n
    assert torch.equal(la, lb) and a.global_step == b.global_step == 1
    loss, L, reg = ref.train_step(t, row, col, w, y, hp)
    np.testing.assert_allclose(la.cpu().numpy()[:3], [loss, L, reg], rtol=1e-5)
    assert_tables_close(a, t, 1e-5, 1e-6)
    # a balanced batch on the same tables: one launch (rows are tagged / the twins flip)
    row2, col2, w2, y2 = make_batch(78, B, V, zipf=False)
    plan2 = hip.build_plan(*to_dev(row2, col2, w2, y2), V, chunk_cap=cap, records=True).compact(hip.lib, records=True)
    assert plan2.host_counts[4] == 0
    if optimizer == "Adagrad":
        hip.step_adagrad(plan2, a, make_hyper(step_form=0, **kw), la)
        assert int((a.R_tag != 0).sum()) == int(plan2.counts[1])
    else:
        before = float(a.scalars[3])
        hip.step_adam(plan2, a, make_hyper(step_form=0, **kw), Ga, la)
        assert float(a.scalars[3]) != before
    ref.train_step(t, row2, col2, w2, y2, hp)
    assert_tables_close(a, t, 2e-5, 2e-6)
