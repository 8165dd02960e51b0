"""CPU tests of the oracle itself: the float64 restatement against finite differences and
against its scalar-C twin, and the dedup-index construction against brute force."""
import math

import numpy as np
import pytest

import glove_ref as ref
from helpers import make_batch


def _total_loss(t, row, col, w, y, hp):
    L, reg, _ = ref.loss_terms(t, row, col, w, y, hp)
    return L + hp.reg_mult * (reg + hp.l2_reg * t.g * t.g)


def test_logistic_head_is_the_weighted_sigmoid_cross_entropy_of_both_labels():
    """logistic_matrix_factorisation.py:48-54 with the TF definition z = label, x = logit:
    max(x, 0) - x z + log(1 + exp(-|x|)), weighted by pos (z = 1) and neg_factor * neg (z = 0), divided by B."""
    p = np.array([-30.0, -2.0, 0.0, 0.7, 25.0])
    pos, neg = np.array([1.0, 0.5, 2.0, 0.0, 3.0]), np.array([0.2, 4.0, 1.0, 2.0, 0.0])
    hp = ref.Hyper(head=1, neg_factor=1.7)
    xent = lambda x, z: np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))
    want = (pos * xent(p, 1.0) + 1.7 * neg * xent(p, 0.0)) / len(p)
    got, e = ref.head_loss_and_error(p, pos, neg, hp, 1.0 / len(p))
    np.testing.assert_allclose(got, want, rtol=1e-13)
    eps = 1e-6
    num = (ref.head_loss_and_error(p + eps, pos, neg, hp, 0.2)[0] - ref.head_loss_and_error(p - eps, pos, neg, hp, 0.2)[0]) / (2 * eps)
    np.testing.assert_allclose(e, num, rtol=1e-6, atol=1e-10)


@pytest.mark.parametrize("m,head", [(1.0, 0), (2.0, 0), (2.0, 1)])
def test_gradients_match_finite_differences(m, head):
    B, V, d = 40, 9, 8
    row, col, w, y = make_batch(0, B, V)
    if head == 1:
        y = np.abs(y) * 0.3                      # negative weights are weights: non-negative
    hp = ref.Hyper(reg_mult=m, l2_reg=0.3, head=head, neg_factor=0.6)
    t = ref.Tables(V, d, "Adagrad", dtype=np.float64, seed=2)
    t.g = np.float64(0.2)
    gr = ref.gradients(t, row, col, w, y, hp)
    rng = np.random.default_rng(0)
    eps = 1e-6
    for name, G in (("R", gr["G_R"]), ("C", gr["G_C"]), ("br", gr["G_br"]), ("bc", gr["G_bc"])):
        W = getattr(t, name)
        for _ in range(12):
            idx = tuple(rng.integers(0, s) for s in W.shape)
            old = W[idx]
            W[idx] = old + eps
            lp = _total_loss(t, row, col, w, y, hp)
            W[idx] = old - eps
            lm = _total_loss(t, row, col, w, y, hp)
            W[idx] = old
            np.testing.assert_allclose(G[idx], (lp - lm) / (2 * eps), rtol=1e-5, atol=1e-9, err_msg=name)
    g0 = t.g
    t.g = g0 + eps
    lp = _total_loss(t, row, col, w, y, hp)
    t.g = g0 - eps
    lm = _total_loss(t, row, col, w, y, hp)
    t.g = g0
    np.testing.assert_allclose(gr["sum_e"] + gr["dg_reg"], (lp - lm) / (2 * eps), rtol=1e-5)


def test_adagrad_sums_duplicates_before_squaring():
    """Two pairs hitting the same row: A += (g1+g2)^2, not g1^2+g2^2."""
    t = ref.Tables(3, 4, "Adagrad", dtype=np.float64, seed=0)
    row, col = np.array([1, 1], np.int32), np.array([0, 2], np.int32)
    w, y = np.ones(2, np.float32), np.array([1.0, -2.0], np.float32)
    hp = ref.Hyper(l2_reg=0.0, learning_rate=0.1)
    gr = ref.gradients(t, row, col, w, y, hp)
    A0 = t.A_R.copy()
    ref.apply_update(t, gr, hp)
    np.testing.assert_allclose(t.A_R[1] - A0[1], gr["G_R"][1] ** 2)
    assert (t.A_R[[0, 2]] == A0[[0, 2]]).all() and t.step == 1


def test_adam_moves_untouched_rows():
    t = ref.Tables(6, 4, "Adam", dtype=np.float64, seed=0)
    t.M_R[:] = 0.01
    t.V_R[:] = 1e-4
    R0 = t.R.copy()
    row, col, w, y = make_batch(1, 3, 6)
    ref.train_step(t, row, col, w, y, ref.Hyper())
    untouched = np.setdiff1d(np.arange(6), row)
    assert len(untouched) and (t.R[untouched] != R0[untouched]).all()


@pytest.mark.parametrize("optimizer", ["Adagrad", "Adam"])
def test_c_port_tracks_numpy_restatement(optimizer):
    import glove_ref_c
    B, V, d = 512, 200, 32
    hp = ref.Hyper(learning_rate=0.05 if optimizer == "Adagrad" else 0.002)
    t64 = ref.Tables(V, d, optimizer, dtype=np.float32, seed=4).astype(np.float64)
    port = glove_ref_c.CPort(t64, B)
    for s in range(10):
        row, col, w, y = make_batch(50 + s, B, V)
        loss64, L64, reg64 = ref.train_step(t64, row, col, w, y, hp)
        loss, L, reg = port.step(row, col, w, y, hp)
        np.testing.assert_allclose([loss, L, reg], [loss64, L64, reg64], rtol=2e-5)
    np.testing.assert_allclose(port.arr["R"], t64.R, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(port.arr["bc"], t64.bc, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(port.g, t64.g, rtol=2e-4, atol=2e-6)
    assert port.st.step == 10
    assert not port.arr["G_R"].any() and not port.arr["mark_r"].any()


@pytest.mark.parametrize("optimizer", ["Adagrad", "Adam"])
@pytest.mark.parametrize("chunk,threads", [(100000, 1), (16, 3), (7, 0)])
def test_all_core_c_port_equals_the_scalar_port(optimizer, chunk, threads):
    """oracle/glove_ref.c glove_ref_step_mt_f32 (the all-core CPU baseline of bench.py) against the scalar port:
    when no id is cut into chunks an id's pairs are summed by one thread in batch order like the scalar port (only
    the global bias differs in its last bits: sum e is reduced in double over threads); heavy ids summed chunk by
    chunk agree within fp32 rounding."""
    import glove_ref_c
    B, V, d = 2000, 150, 20
    hp = ref.Hyper(learning_rate=0.05 if optimizer == "Adagrad" else 0.002)
    t = ref.Tables(V, d, optimizer, dtype=np.float32, seed=4)
    a, b = glove_ref_c.CPort(t, B), glove_ref_c.CPort(t, B)
    for s in range(4):
        row, col, w, y = make_batch(70 + s, B, V)
        la = a.step(row, col, w, y, hp)
        lb = b.step_mt(glove_ref_c.BatchIndex(row, col, d, chunk=chunk), row, col, w, y, hp, threads=threads)
        np.testing.assert_allclose(lb, la, rtol=1e-5)
    for n in ("R", "C", "br", "bc", "S1_R", "S1_C", "S1_bc"):
        if chunk >= B:
            np.testing.assert_allclose(b.arr[n], a.arr[n], rtol=1e-6, atol=1e-9, err_msg=n)
        else:
            np.testing.assert_allclose(b.arr[n], a.arr[n], rtol=1e-4, atol=1e-6, err_msg=n)
    np.testing.assert_allclose(b.g, a.g, rtol=1e-4, atol=1e-7)
    assert a.st.step == b.st.step == 4 and not b.arr["G_R"].any() and not b.arr["G_bc"].any()
    assert b.max_threads() >= 1


@pytest.mark.parametrize("B,V,cap", [(1, 3, 4), (50, 7, 3), (1000, 31, 8), (300, 1000, 32)])
def test_plan_partitions_the_batch(B, V, cap):
    row, col, _, _ = make_batch(B, B, V)
    p = ref.build_plan(row, col, cap)
    nc_r, nu_r, nc_c, nu_c = p["counts"][:4]
    assert nu_r == len(np.unique(row)) and nu_c == len(np.unique(col))
    # every pair is in exactly one chunk of each side; chunks respect the cap and hold one id
    for side, keys in (("r", row[p["perm_r"]]), ("c", col[p["perm_r"]][p["c_perm"]])):
        cs, cid = p[side + "_chunk_start"], p[side + "_chunk_id"]
        assert cs[0] == 0 and cs[-1] == B and (np.diff(cs) > 0).all() and (np.diff(cs) <= cap).all()
        for j in range(len(cid)):
            assert (keys[cs[j]:cs[j + 1]] == cid[j]).all()
        us = p[side + "_uniq_slot"]
        assert us[0] == 0 and us[-1] == len(cid)
        assert len(np.unique(cid[us[:-1]])) == len(us) - 1
    # what the library sizes a fused pass by when a plan's counts were never read back (glove_common.h most_chunks):
    # a side has at most one chunk per distinct id plus one per full chunk_cap pairs
    assert nc_r <= min(B, V) + B // cap + 1 and nc_c <= min(B, V) + B // cap + 1
    assert nc_r <= nu_r + B // cap and nc_c <= nu_c + B // cap
    # col side points back to the same pairs
    np.testing.assert_array_equal(p["c_partner"], row[p["perm_r"]][p["c_perm"]])
    # segment sums through the plan == np.add.at
    e = np.random.default_rng(0).normal(size=B)
    want = np.zeros(V)
    np.add.at(want, col, e)
    got = np.zeros(V)
    e_rs = e[p["perm_r"]]
    cs = p["c_chunk_start"]
    for j, cid in enumerate(p["c_chunk_id"]):
        got[cid] += e_rs[p["c_perm"][cs[j]:cs[j + 1]]].sum()
    np.testing.assert_allclose(got, want, atol=1e-12)


def test_eval_and_topk_reference_behaviour():
    t = ref.Tables(20, 8, "Adagrad", dtype=np.float64, seed=1)
    row, col, w, y = make_batch(3, 100, 20)
    m = ref.eval_metrics(t, row, col, w, y)
    p = ref.forward(t, row, col)
    np.testing.assert_allclose(m["average_loss"], np.average((p - y) ** 2, weights=w))
    sims, idx = ref.cosine_topk(t.R, np.array([4, 9]), 5)
    assert idx[0, 0] == 4 and idx[1, 0] == 9 and np.allclose(sims[:, 0], 1.0)
    assert (np.diff(sims, axis=1) <= 1e-15).all()


def test_plan_treats_out_of_range_ids_as_unknown_token():
    """estimator.py:26-28: the vocabulary lookup maps anything unknown to id 0."""
    row = np.array([3, 9, -1, 2, 3], np.int32)
    col = np.array([0, 1, 2, 77, 4], np.int32)
    p = ref.build_plan(row, col, 4, V=5)
    q = ref.build_plan(np.array([3, 0, 0, 2, 3]), np.array([0, 1, 2, 0, 4]), 4)
    assert p["counts"][5] == 3 and q["counts"][5] == 0
    for k in ("r_partner", "c_partner", "r_chunk_id", "c_chunk_id", "c_perm", "r_to_c"):
        np.testing.assert_array_equal(p[k], q[k])


@pytest.mark.parametrize("optimizer,lr,steps", [("Adagrad", 0.05, 4), ("Adam", 0.001, 6)])
def test_oracle_steps_match_an_independent_autodiff_and_optimizer(optimizer, lr, steps):
    """The restated step against machinery the oracle shares no code with: the loss of SURVEY.md Appendix A written as a
    torch expression (float64), gradients by torch.autograd (duplicate ids sum as Keras' dedup does), the update by
    torch.optim.  Adagrad(initial_accumulator_value=0.1, eps=1e-7) IS the Keras-legacy rule (sum, then square; eps outside
    the root; untouched rows do not move).  torch's Adam places eps differently (sqrt(v)/sqrt(1-b2^t) + eps where Keras has
    sqrt(v) + eps under lr_t = lr sqrt(1-b2^t)/(1-b1^t)); handing torch eps/sqrt(1-b2^t) at step t makes the two the same
    expression, and — like Keras' legacy sparse path, unlike a lazy Adam — torch's dense Adam moves EVERY row every step.  Not TensorFlow, so the step stays unpinned; it pins
    the restatement's calculus and update algebra."""
    import torch
    B, V, d = 64, 11, 6
    hp = ref.Hyper(learning_rate=lr, l2_reg=0.05, reg_mult=2.0)
    t = ref.Tables(V, d, optimizer, dtype=np.float64, seed=4)
    t.g = np.float64(0.1)
    P = {n: torch.tensor(np.array(getattr(t, n)), dtype=torch.float64, requires_grad=True) for n in ("R", "C", "br", "bc")}
    P["g"] = torch.tensor(float(t.g), dtype=torch.float64, requires_grad=True)
    params = list(P.values())
    f32 = lambda v: float(np.float32(v))              # Keras holds the hyper-parameters in float32 (oracle apply_update)
    opt = (torch.optim.Adagrad(params, lr=f32(lr), initial_accumulator_value=0.1, eps=f32(1e-7)) if optimizer == "Adagrad"
           else torch.optim.Adam(params, lr=f32(lr), betas=(f32(0.9), f32(0.999)), eps=f32(1e-7)))
    untouched_moved = False
    for s in range(steps):
        row, col, w, y = make_batch(30 + s, B, V)
        r_, c_ = torch.from_numpy(row).long(), torch.from_numpy(col).long()
        wt, yt = torch.from_numpy(w).double(), torch.from_numpy(y).double()
        r, c = P["R"][r_], P["C"][c_]
        p = (r * c).sum(-1) + P["br"][r_] + P["bc"][c_] + P["g"]
        L = (wt * (p - yt) ** 2).sum() / B
        reg = hp.l2_reg / (d * B) * ((r ** 2).sum() + (c ** 2).sum()) + hp.l2_reg / B * ((P["br"][r_] ** 2).sum() + (P["bc"][c_] ** 2).sum()) \
            + hp.l2_reg * P["g"] ** 2
        loss = L + hp.reg_mult * reg
        opt.zero_grad()
        loss.backward()
        if optimizer == "Adam":
            opt.param_groups[0]["eps"] = f32(1e-7) / math.sqrt(1.0 - f32(0.999) ** (s + 1))
        before = P["R"].detach().clone()
        opt.step()
        want_loss, want_L, want_reg = ref.train_step(t, row, col, w, y, hp)
        np.testing.assert_allclose(loss.item(), want_loss, rtol=1e-12)
        np.testing.assert_allclose(L.item(), want_L, rtol=1e-12)
        rest = np.setdiff1d(np.arange(V), row)
        if len(rest):
            moved = bool((P["R"].detach()[rest] != before[rest]).any())
            untouched_moved |= moved
            assert moved == (optimizer == "Adam" and s > 0)          # Keras-legacy Adam decays every row once it has momentum
        tol = dict(rtol=1e-10, atol=1e-13) if optimizer == "Adagrad" else dict(rtol=1e-9, atol=1e-13)
        for n in ("R", "C", "br", "bc"):
            np.testing.assert_allclose(P[n].detach().numpy(), getattr(t, n), err_msg="%s after step %d" % (n, s + 1), **tol)
        np.testing.assert_allclose(P["g"].item(), t.g, **tol)
    assert untouched_moved == (optimizer == "Adam")


@pytest.mark.parametrize("optimizer,kw", [("SGD", {}), ("SGD", {"momentum": 0.9}), ("RMSprop", {}), ("Adamax", {}), ("Adadelta", {}), ("Ftrl", {}), ("Nadam", {})])
def test_other_keras_optimizers_match_independent_updates(optimizer, kw):
    """The optimizers `tf.keras.optimizers.get(name)` resolves beyond Adagrad / Adam (train_utils.py:13-16), restated with their
    Keras-legacy sparse semantics, against machinery the oracle shares no code with: gradients by torch.autograd, the update by
    torch.optim.SGD (Keras' accum = accum m - lr g, var += accum is torch's buf = m buf + g, var -= lr buf for a constant lr)
    and torch.optim.RMSprop (alpha = rho; Keras decays the whole rms slot, rows without gradient do not move: the dense form).
    Adamax is lazy in Keras (touched rows only) and keeps eps in the denominator where torch has it inside the max: checked
    against a hand-written float64 update of the touched rows.  Adadelta: torch.optim.Adadelta's formulas are TensorFlow's
    (accumulators, delta = sqrt(acc_delta + eps) / sqrt(square_avg + eps) g), but Keras' sparse path leaves the accumulators
    of untouched rows alone where torch decays them: torch's functional single-tensor update is run on the touched rows only.
    Ftrl (no torch counterpart): the textbook FTRL-proximal closed form — z accumulates g - sigma w with sigma = (sqrt(n') -
    sqrt(n)) / lr, w = -z lr / sqrt(n') for l1 = l2 = 0 — written out per element in plain Python floats."""
    import torch
    B, V, d, lr = 48, 9, 5, 0.01
    hp = ref.Hyper(learning_rate=lr, l2_reg=0.05, reg_mult=2.0, **kw)
    t = ref.Tables(V, d, optimizer, dtype=np.float64, seed=4)
    t.g = np.float64(0.1)
    P = {n: torch.tensor(np.array(getattr(t, n)), dtype=torch.float64, requires_grad=True) for n in ("R", "C", "br", "bc")}
    P["g"] = torch.tensor(float(t.g), dtype=torch.float64, requires_grad=True)
    params = list(P.values())
    f32 = lambda v: float(np.float32(v))
    opt = None
    if optimizer == "SGD":
        opt = torch.optim.SGD(params, lr=f32(lr), momentum=f32(kw.get("momentum", 0.0)))
    elif optimizer == "RMSprop":
        opt = torch.optim.RMSprop(params, lr=f32(lr), alpha=f32(0.9), eps=f32(1e-7))
    elif optimizer == "Nadam":
        # torch.optim.NAdam has Keras' formulas (momentum_decay = schedule_decay = 0.004, base 0.96) and, fed dense gradients that
        # are zero on untouched rows, decays m and v everywhere as Keras' sparse path does; it also moves those rows, which
        # Keras does not: their values are put back after every step below
        opt = torch.optim.NAdam(params, lr=f32(lr), betas=(f32(0.9), f32(0.999)), eps=f32(1e-7), momentum_decay=0.004)
    state = {n: (np.zeros_like(getattr(t, n)), np.zeros_like(getattr(t, n))) for n in ("R", "C", "br", "bc")}
    if optimizer == "Ftrl":
        state = {n: (np.full_like(getattr(t, n), 0.1), np.zeros_like(getattr(t, n))) for n in ("R", "C", "br", "bc")}
    mg, vg = (0.1, 0.0) if optimizer == "Ftrl" else (0.0, 0.0)
    for s in range(5):
        row, col, w, y = make_batch(70 + s, B, V)
        row[row == 3] = 4                                       # row 3 is never touched: it must not move (nor its slots)
        r_, c_ = torch.from_numpy(row).long(), torch.from_numpy(col).long()
        wt, yt = torch.from_numpy(w).double(), torch.from_numpy(y).double()
        r, c = P["R"][r_], P["C"][c_]
        p = (r * c).sum(-1) + P["br"][r_] + P["bc"][c_] + P["g"]
        L = (wt * (p - yt) ** 2).sum() / B
        reg = hp.l2_reg / (d * B) * ((r ** 2).sum() + (c ** 2).sum()) + hp.l2_reg / B * ((P["br"][r_] ** 2).sum() + (P["bc"][c_] ** 2).sum()) \
            + hp.l2_reg * P["g"] ** 2
        loss = L + hp.reg_mult * reg
        for q in params:
            q.grad = None
        loss.backward()
        before = P["R"].detach().clone()
        if opt is not None:
            kept = {n: P[n].detach().clone() for n in ("R", "C", "br", "bc")}
            opt.step()
            if optimizer == "Nadam":
                with torch.no_grad():
                    for n, ids in (("R", row), ("C", col), ("br", row), ("bc", col)):
                        idle = torch.ones(V, dtype=torch.bool)
                        idle[torch.from_numpy(np.unique(ids))] = False
                        P[n][idle] = kept[n][idle]
        elif optimizer == "Adadelta":                            # torch's own update formulas, on the touched rows only
            from torch.optim.adadelta import adadelta as torch_adadelta
            rho, eps = f32(0.95), f32(1e-7)
            with torch.no_grad():
                for n, ids in (("R", row), ("C", col), ("br", row), ("bc", col)):
                    u = torch.from_numpy(np.unique(ids))
                    sq, acc = state[n]
                    pv, gv = P[n][u].clone(), P[n].grad[u].clone()
                    sv, av = torch.from_numpy(sq[u.numpy()]), torch.from_numpy(acc[u.numpy()])
                    torch_adadelta([pv], [gv], [sv], [av], [torch.tensor(0.0)], foreach=False, lr=f32(lr), rho=rho, eps=eps, weight_decay=0.0, maximize=False)
                    P[n][u] = pv
                    sq[u.numpy()], acc[u.numpy()] = sv.numpy(), av.numpy()
                gp, gg = P["g"].detach().clone().reshape(1), P["g"].grad.detach().clone().reshape(1)
                sv, av = torch.tensor([mg], dtype=torch.float64), torch.tensor([vg], dtype=torch.float64)
                torch_adadelta([gp], [gg], [sv], [av], [torch.tensor(0.0)], foreach=False, lr=f32(lr), rho=rho, eps=eps, weight_decay=0.0, maximize=False)
                P["g"].copy_(gp[0])
                mg, vg = sv.item(), av.item()
        elif optimizer == "Ftrl":                                # FTRL-proximal, element by element
            import math

            def ftrl_elem(wv, nv, zv, gv, lr_):
                n2 = nv + gv * gv
                sigma = (math.sqrt(n2) - math.sqrt(nv)) / lr_
                z2 = zv + gv - sigma * wv
                return (-z2 * lr_ / math.sqrt(n2) if z2 != 0 else 0.0), n2, z2
            with torch.no_grad():
                for n, ids in (("R", row), ("C", col), ("br", row), ("bc", col)):
                    acc, lin = state[n]
                    g = P[n].grad.numpy()
                    wv = P[n].detach().numpy().copy()
                    for uu in np.unique(ids):
                        for idx in np.ndindex(wv[uu].shape):
                            k = (uu,) + idx
                            wv[k], acc[k], lin[k] = ftrl_elem(float(wv[k]), float(acc[k]), float(lin[k]), float(g[k]), f32(lr))
                    P[n].copy_(torch.from_numpy(wv))
                gn, mg, vg = ftrl_elem(P["g"].item(), mg, vg, P["g"].grad.item(), f32(lr))
                P["g"].fill_(gn)
        else:                                                    # Adamax by hand, touched rows only
            b1, b2, eps = f32(0.9), f32(0.999), f32(1e-7)
            lr_t = f32(lr) / (1.0 - b1 ** (s + 1))
            with torch.no_grad():
                for n, ids in (("R", row), ("C", col), ("br", row), ("bc", col)):
                    g = P[n].grad.numpy()
                    m, v = state[n]
                    u = np.unique(ids)
                    m[u] = b1 * m[u] + (1 - b1) * g[u]
                    v[u] = np.maximum(b2 * v[u], np.abs(g[u]))
                    P[n][torch.from_numpy(u)] -= torch.from_numpy(lr_t * m[u] / (v[u] + eps))
                dg = P["g"].grad.item()
                mg = b1 * mg + (1 - b1) * dg
                vg = max(b2 * vg, abs(dg))
                P["g"] -= lr_t * mg / (vg + eps)
        want_loss, _, _ = ref.train_step(t, row, col, w, y, hp)
        np.testing.assert_allclose(loss.item(), want_loss, rtol=1e-7 if optimizer == "Nadam" else 1e-12)    # (parameters that carry torch's float32 noise)
        assert torch.equal(P["R"].detach()[3], before[3])       # the untouched row
        # (torch.optim.NAdam keeps its step count and the momentum product in float32 scalars: 1e-7 of relative noise)
        tol = dict(rtol=1e-6, atol=1e-10) if optimizer == "Nadam" else dict(rtol=1e-9, atol=1e-13)
        for n in ("R", "C", "br", "bc"):
            np.testing.assert_allclose(P[n].detach().numpy(), getattr(t, n), err_msg="%s after step %d" % (n, s + 1), **tol)
        np.testing.assert_allclose(P["g"].item(), t.g, **tol)
