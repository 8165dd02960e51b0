"""Parity of the HIP path (through the C ABI) against the float64 oracle on identical inputs.

Tolerances (fp32 kernels vs float64 restatement): loss rtol 1e-5; updated
parameters / optimizer slots rtol 1e-5, atol 1e-6; integer index work bit-exact.
"""
import os

import numpy as np
import pytest
import torch

import glove_ref as ref
from helpers import assert_tables_close, make_batch, oracle_tables, tables_from_oracle, to_dev, PLAN_ARRAYS, _poison, _assert_plan_equals_oracle

pytestmark = pytest.mark.gpu

from trainer.hip_api import FUSED_STEP_BYTES  # noqa: E402

LOSS_RTOL = 1e-5
PARAM_RTOL, PARAM_ATOL = 1e-5, 1e-6


def _hyper(hp: ref.Hyper, B, step_form=0):
    from trainer.hip_api import make_hyper
    return make_hyper(l2_reg=hp.l2_reg, reg_mult=hp.reg_mult, learning_rate=hp.learning_rate, epsilon=hp.epsilon,
                      beta1=hp.beta1, beta2=hp.beta2, batch_size=B, head=hp.head, neg_factor=hp.neg_factor,
                      step_form=step_form)


def _assert_tables_agree(a, b, rtol, atol, info=""):
    for n in ("R", "C", "br", "bc"):
        np.testing.assert_allclose(getattr(a, n).cpu().numpy(), getattr(b, n).cpu().numpy(), rtol=rtol, atol=atol, err_msg=n + " " + info)
        np.testing.assert_allclose(a.s1[n].cpu().numpy(), b.s1[n].cpu().numpy(), rtol=rtol, atol=atol, err_msg="slot1 " + n + " " + info)
    np.testing.assert_allclose(a.scalars.cpu().numpy(), b.scalars.cpu().numpy(), rtol=rtol, atol=atol, err_msg=info)
    assert a.global_step == b.global_step, info


def _assert_same_bits(a, b, info=""):
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n + " " + info
        assert torch.equal(a.s1[n], b.s1[n]), "slot1 " + n + " " + info
        if n in a.s2:
            assert torch.equal(a.s2[n], b.s2[n]), "slot2 " + n + " " + info
    assert torch.equal(a.scalars, b.scalars) and a.global_step == b.global_step, info


@pytest.mark.parametrize("B", [300, 9000])        # one-workgroup builder / tiled builder
def test_plan_maps_out_of_range_ids_to_zero(hip, B):
    """Ids outside [0, V) must never reach the tables: both builders treat them as id 0 (the reference's
    unknown-token id, estimator.py:26-28) and count them in counts[5]."""
    V = 40
    row, col, w, y = make_batch(B, B, V)
    row[::7] = V + 3
    col[5::11] = -2
    row[1] = np.iinfo(np.int32).max
    plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=8)
    want = ref.build_plan(row, col, 8, V=V)
    assert want["counts"][5] > 0
    np.testing.assert_array_equal(plan.counts.cpu().numpy(), want["counts"])
    nc_r, nu_r, nc_c, nu_c = want["counts"][:4]
    np.testing.assert_array_equal(plan.r_partner.cpu().numpy()[:B], want["r_partner"])
    np.testing.assert_array_equal(plan.c_partner.cpu().numpy()[:B], want["c_partner"])
    np.testing.assert_array_equal(plan.r_chunk_id.cpu().numpy()[:nc_r], want["r_chunk_id"])
    np.testing.assert_array_equal(plan.c_chunk_id.cpu().numpy()[:nc_c], want["c_chunk_id"])
    np.testing.assert_array_equal(plan.c_perm.cpu().numpy()[:B], want["c_perm"])
    # and a step over it stays inside the tables (equals the step on the cleaned batch)
    from trainer.hip_api import DeviceTables
    crow = np.where((row < 0) | (row >= V), 0, row).astype(np.int32)
    ccol = np.where((col < 0) | (col >= V), 0, col).astype(np.int32)
    out = []
    for r, c in ((row, col), (crow, ccol)):
        tables = DeviceTables(V, 16, "Adagrad", hip.device, seed=3)
        hip.step_adagrad(hip.build_plan(*to_dev(r, c, w, y), V, chunk_cap=8), tables, _hyper(ref.Hyper(), B))
        out.append(tables.R.cpu().numpy().copy())
    np.testing.assert_array_equal(out[0], out[1])


@pytest.mark.parametrize("B,V,cap", [(1, 5, 32), (7, 50, 32), (64, 50, 4), (1024, 300, 32), (5000, 97, 3),
                                     (20000, 2000, 32), (100000, 12000, 16),
                                     # the tiled builder (B > 4096, tiles of 2048 positions): exact tile multiples,
                                     # one position past, runs that span many tiles, one id only, chunks of one pair
                                     (4097, 300, 32), (6144, 50, 16), (8193, 3, 32), (10000, 1, 7), (30000, 40000, 1),
                                     (65536, 7, 5), (300000, 1000, 32)])
def test_plan_build_bit_exact(hip, B, V, cap):
    row, col, w, y = make_batch(B + V, B, V)
    drow, dcol, dw, dy = to_dev(row, col, w, y)
    plan = hip.build_plan(drow, dcol, dw, dy, V, chunk_cap=cap)
    want = ref.build_plan(row, col, cap)
    counts = plan.counts.cpu().numpy()
    np.testing.assert_array_equal(counts, want["counts"])
    nc_r, nu_r, nc_c, nu_c, n_heavy = counts[:5]
    np.testing.assert_array_equal(np.sort(plan.heavy.cpu().numpy()[:n_heavy]), want["heavy"])   # order is free
    np.testing.assert_array_equal(plan.r_partner.cpu().numpy()[:B], want["r_partner"])
    np.testing.assert_array_equal(plan.r_w.cpu().numpy()[:B], w[want["perm_r"]])
    np.testing.assert_array_equal(plan.r_y.cpu().numpy()[:B], y[want["perm_r"]])
    np.testing.assert_array_equal(plan.r_chunk_id.cpu().numpy()[:nc_r], want["r_chunk_id"])
    np.testing.assert_array_equal(plan.r_chunk_start.cpu().numpy()[:nc_r + 1], want["r_chunk_start"])
    np.testing.assert_array_equal(plan.r_uniq_slot.cpu().numpy()[:nu_r + 1], want["r_uniq_slot"])
    np.testing.assert_array_equal(plan.c_perm.cpu().numpy()[:B], want["c_perm"])
    np.testing.assert_array_equal(plan.r_to_c.cpu().numpy()[:B], want["r_to_c"])
    np.testing.assert_array_equal(plan.c_w.cpu().numpy()[:B], w[want["perm_r"]][want["c_perm"]])
    np.testing.assert_array_equal(plan.c_y.cpu().numpy()[:B], y[want["perm_r"]][want["c_perm"]])
    np.testing.assert_array_equal(plan.c_partner.cpu().numpy()[:B], want["c_partner"])
    np.testing.assert_array_equal(plan.c_chunk_id.cpu().numpy()[:nc_c], want["c_chunk_id"])
    np.testing.assert_array_equal(plan.c_chunk_start.cpu().numpy()[:nc_c + 1], want["c_chunk_start"])
    np.testing.assert_array_equal(plan.c_uniq_slot.cpu().numpy()[:nu_c + 1], want["c_uniq_slot"])
    np.testing.assert_array_equal(plan.r_uniq_rec.cpu().numpy()[:4 * nu_r].reshape(-1, 4), want["r_uniq_rec"])
    np.testing.assert_array_equal(plan.c_uniq_rec.cpu().numpy()[:4 * nu_c].reshape(-1, 4), want["c_uniq_rec"])
    # per-chunk records {id, n, id's position, flag | chunks behind | blocks of 8 pairs: partner, w, y} (small plans carry them from the build, big ones
    # get them when compacted): same content as the SoA arrays, padding slots weigh 0
    cpr = plan.compact(hip.lib)
    capP = (cap + 7) // 8 * 8
    rd = 4 + 3 * capP
    for side, nc, ids, starts, partner, ww, yy in (
            ("r", nc_r, want["r_chunk_id"], want["r_chunk_start"], want["r_partner"], w[want["perm_r"]], y[want["perm_r"]]),
            ("c", nc_c, want["c_chunk_id"], want["c_chunk_start"], want["c_partner"],
             w[want["perm_r"]][want["c_perm"]], y[want["perm_r"]][want["c_perm"]])):
        for p_ in [q_ for q_ in (plan, cpr) if q_.r_crec is not None]:
            rec = p_.records(side, nc).cpu().numpy()
            n = np.diff(starts)
            ids = np.asarray(ids)
            np.testing.assert_array_equal(rec[:, 0], ids)
            np.testing.assert_array_equal(rec[:, 1], n)
            # word 2: the id's position among the side's distinct ids; word 3: first-chunk-of-its-id flag in bit 31,
            # chunks of the same id behind this one below it
            first = np.r_[True, ids[1:] != ids[:-1]]
            run_id = np.cumsum(first) - 1
            np.testing.assert_array_equal(rec[:, 2], run_id)
            run_end = np.r_[np.flatnonzero(first)[1:], nc] - 1
            want_w3 = (run_end[run_id] - np.arange(nc)).astype(np.uint32) | (first.astype(np.uint32) << 31)
            np.testing.assert_array_equal(rec[:, 3].view(np.uint32), want_w3)
            # the pairs in blocks of 8: {partner[8] | w[8] | y[8]} per block
            blocks = rec[:, 4:].reshape(nc, capP // 8, 3, 8)
            for j in (0, nc // 2, nc - 1):
                sl = slice(starts[j], starts[j + 1])
                np.testing.assert_array_equal(blocks[j, :, 0].reshape(-1)[:n[j]], partner[sl])
                np.testing.assert_array_equal(blocks[j, :, 1].reshape(-1)[:n[j]].view(np.float32), ww[sl])
                np.testing.assert_array_equal(blocks[j, :, 2].reshape(-1)[:n[j]].view(np.float32), yy[sl])
                nb_ = (n[j] + 7) // 8               # the blocks a reader of this chunk touches (the others stay unwritten)
                assert (blocks[j, :nb_, 1].reshape(-1)[n[j]:].view(np.float32) == 0).all()
                assert ((blocks[j, :nb_, 0] >= 0) & (blocks[j, :nb_, 0] < V)).all()    # their padding slots hold valid ids
    assert (plan.r_crec is not None) == (B <= 4096)
    assert (cpr.r_crec is not None) == (B <= 2048 or 4 * B >= cap * max(nc_r, nc_c))     # small batches (the tagged step reads records only) and reasonably filled chunks
    # compacted copy describes the same index
    cp = plan.compact()
    assert cp.cap_chunks == max(nc_r, nc_c) and cp.cap_uniq == max(nu_r, nu_c)
    np.testing.assert_array_equal(cp.r_chunk_start.cpu().numpy()[:nc_r + 1], want["r_chunk_start"])


@pytest.mark.parametrize("B,V,cap", [(1000, 300, 8), (4096, 12000, 16), (9000, 97, 3), (131072, 10000, 16), (70000, 300000, 32),
                                     (300000, 2000000, 32)])
def test_consecutive_builds_into_a_poisoned_staging_plan(hip, plan_checker, B, V, cap):
    """Two consecutive glove_plan_build calls into the SAME staging plan, every plan array and the whole build workspace
    filled with 0xFF before each: the index must not depend on anything a build did not write itself (state zeroed at
    allocation only, leftovers of the previous batch).  Bit-exact against the oracle both times, and the device-side
    range check (every partner / perm / record slot a step kernel may read) finds nothing."""
    from trainer.hip_api import Plan
    staging = Plan(B, V, cap, "cuda:0", records=True, run_words=True)   # records filled by the build as well (small batches; big batches on big tables); run words beside them
    ws = torch.empty(hip.lib.glove_plan_workspace_bytes(B, V), dtype=torch.uint8, device="cuda:0")
    errors = torch.zeros(8, dtype=torch.int32, device="cuda:0")
    for k in range(2):
        row, col, w, y = make_batch(B + V + 31 * k, B, V, zipf=(k == 0))
        if k == 1:
            row[::9] = V + 5                             # ids outside the vocabulary count as id 0
            col[3::13] = -7
        _poison(staging, ws)
        hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap, into=staging, ws=ws)
        plan_checker(staging, V, errors)
        _assert_plan_equals_oracle(staging, ref.build_plan(row, col, cap, V=V), B, w, y)
        assert errors.tolist() == [0] * 8, errors.tolist()


def test_plan_checker_sees_what_it_should(hip, plan_checker):
    """The checker itself: a sound plan passes; a stale partner id, a broken permutation and a bad record slot are each reported."""
    B, V, cap = 9000, 500, 8
    row, col, w, y = make_batch(5, B, V)
    plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap).compact(hip.lib, records=True)
    errors = torch.zeros(8, dtype=torch.int32, device="cuda:0")
    plan_checker(plan, V, errors)
    assert errors.tolist() == [0] * 8
    plan.r_partner[17] = V + 1
    plan.c_perm[5], plan.c_perm[6] = plan.c_perm[6].clone(), plan.c_perm[6].clone()
    plan.c_crec[4 + 1] = -1                              # second partner slot of chunk 0 (a padding slot counts too)
    plan_checker(plan, V, errors)
    e = errors.tolist()
    assert e[0] == 1 and e[4] >= 1 and e[7] == 1 and e[1] == e[2] == e[3] == 0, e


def test_index_rebuilt_inside_a_replayed_hipgraph(hip, plan_checker):
    """What `bench.py --dynamic` and the reshuffling runner do: index build into a staging plan + step, captured once and
    REPLAYED — with the device-side plan check between build and step, and the staging plan and workspace poisoned
    between replays.  The trajectory equals stepping on statically built plans, bit for bit, and no check ever fires."""
    from trainer.hip_api import DeviceTables, Plan
    B, V, d, cap = 131072, 10000, 64, 16
    batches = [to_dev(*make_batch(900 + k, B, V)) for k in range(3)]
    a, b = DeviceTables(V, d, "Adagrad", seed=4), DeviceTables(V, d, "Adagrad", seed=4)
    h = _hyper(ref.Hyper(learning_rate=0.05), B, step_form=1)
    static = [hip.build_plan(*bt, V, chunk_cap=cap) for bt in batches]
    staging = Plan(B, V, cap, "cuda:0")
    ws = torch.empty(hip.lib.glove_plan_workspace_bytes(B, V), dtype=torch.uint8, device="cuda:0")
    step_ws = torch.empty(hip.lib.glove_step_workspace_bytes(B, B, a.d), dtype=torch.uint8, device="cuda:0")
    errors = torch.zeros(8, dtype=torch.int32, device="cuda:0")
    la, lb = torch.zeros(4, device="cuda:0"), torch.zeros(4, device="cuda:0")

    def burst():
        for bt in batches:
            hip.build_plan(*bt, V, chunk_cap=cap, into=staging, ws=ws)
            plan_checker(staging, V, errors)
            hip.step_adagrad(staging, b, h, lb, step_ws)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        burst()                                          # warm the launch paths outside the capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        burst()
    for rnd in range(4):                                 # the warm-up burst + 3 replays
        if rnd:
            _poison(staging, ws)
            graph.replay()
        for p in static:
            hip.step_adagrad(p, a, h, la, step_ws)
        torch.cuda.synchronize()
        assert errors.tolist() == [0] * 8, (rnd, errors.tolist())
        _assert_same_bits(a, b, "round %d" % rnd)
        assert torch.equal(la, lb)


@pytest.mark.parametrize("seed", range(int(os.environ.get("GLOVE_FUZZ_CASES", "40"))))
def test_randomized_plan_is_bit_exact(hip, seed):
    """Seeded random batches through both index builders (one workgroup / tiled): every array of the plan equals
    the oracle's, for uniform and Zipf ids, any chunk cap, with and without ids outside the vocabulary."""
    rng = np.random.default_rng(7000 + seed)
    V = int(rng.choice([1, 3, 50, 1000, 20000, 300000]))
    B = int(rng.choice([1, 2, 100, 4095, 4096, 4097, 6143, 6144, 6145, 20000, 70000]))
    cap = int(rng.integers(1, 33))
    if rng.uniform() < 0.5:
        pdf = np.arange(1, V + 1, dtype=np.float64) ** -1.2
        row, col = (rng.choice(V, B, p=pdf / pdf.sum()).astype(np.int32) for _ in range(2))
    else:
        row, col = (rng.integers(0, V, B).astype(np.int32) for _ in range(2))
    if rng.uniform() < 0.3:
        row[rng.integers(0, B, max(1, B // 50))] = V + int(rng.integers(0, 5))
        col[rng.integers(0, B, max(1, B // 70))] = -1 - int(rng.integers(0, 5))
    w, y = rng.uniform(size=B).astype(np.float32), rng.normal(size=B).astype(np.float32)
    plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap)
    want = ref.build_plan(row, col, cap, V=V)
    counts = plan.counts.cpu().numpy()
    np.testing.assert_array_equal(counts, want["counts"], err_msg=str((V, B, cap)))
    nc_r, nu_r, nc_c, nu_c, n_heavy = counts[:5]
    got = lambda name, n: getattr(plan, name).cpu().numpy()[:n]
    for name, n in (("r_partner", B), ("c_partner", B), ("c_perm", B), ("r_to_c", B), ("r_chunk_id", nc_r),
                    ("c_chunk_id", nc_c), ("r_chunk_start", nc_r + 1), ("c_chunk_start", nc_c + 1),
                    ("r_uniq_slot", nu_r + 1), ("c_uniq_slot", nu_c + 1)):
        np.testing.assert_array_equal(got(name, n), want[name], err_msg="%s %s" % (name, (V, B, cap)))
    np.testing.assert_array_equal(got("r_uniq_rec", 4 * nu_r).reshape(-1, 4), want["r_uniq_rec"])
    np.testing.assert_array_equal(got("c_uniq_rec", 4 * nu_c).reshape(-1, 4), want["c_uniq_rec"])
    np.testing.assert_array_equal(np.sort(got("heavy", n_heavy)), want["heavy"])
    np.testing.assert_array_equal(got("r_w", B), w[want["perm_r"]])
    np.testing.assert_array_equal(got("c_y", B), y[want["perm_r"]][want["c_perm"]])


def test_plan_empty_batch(hip):
    e = torch.empty(0, dtype=torch.int32, device="cuda:0")
    f = torch.empty(0, dtype=torch.float32, device="cuda:0")
    plan = hip.build_plan(e, e, f, f, 10)
    assert plan.counts.tolist() == [0] * 8
    # a step over it launches, moves no row and advances global_step (the reference never feeds one:
    # its training input repeats forever in full batches, data_utils.py:12-21)
    from trainer.hip_api import DeviceTables
    dt = DeviceTables(10, 8, "Adagrad", seed=0)
    before = dt.R.clone()
    loss_out = torch.ones(4, device="cuda:0")
    hip.step_adagrad(plan, dt, _hyper(ref.Hyper(), 1), loss_out)
    assert torch.equal(dt.R, before) and dt.global_step == 1 and loss_out[:3].tolist() == [0.0, 0.0, 0.0]


# (B, V, d, chunk_cap): d covers every (lanes-per-row, float4-per-lane) kernel shape
STEP_CASES = [
    (7, 50, 8, 32), (64, 50, 64, 4), (1024, 300, 64, 32), (1024, 50, 128, 32), (1024, 120, 300, 32),
    (3000, 40, 16, 2), (4096, 500, 256, 8), (2048, 64, 512, 32), (2048, 64, 1024, 32), (512, 64, 768, 16),
    (20000, 2000, 64, 32), (1024, 12000, 64, 32), (777, 33, 96, 5), (900, 70, 384, 7),
    # embedding sizes that are not a multiple of 4: rows are padded to 16 B, the reference's d enters l2/d
    (1024, 300, 50, 32), (513, 40, 2, 8), (2048, 90, 150, 16), (700, 60, 301, 32), (640, 77, 1, 4),
    # chunk caps below the 8 pair slots a trip of the d <= 32 shapes reads, with per-chunk records (B <= 4096):
    # the record fields are padded to 8 slots (glove_common.h rec_cap); (3000, 40, 16, 2) above is the case that faulted
    # while the records were being written
    (1000, 40, 16, 7), (1000, 40, 32, 3), (2500, 30, 8, 1), (4096, 64, 24, 5),
]


@pytest.mark.parametrize("B,V,d,cap", STEP_CASES)
def test_adagrad_single_step(hip, B, V, d, cap):
    from trainer.hip_api import DeviceTables
    row, col, w, y = make_batch(B * 3 + d, B, V)
    hp = ref.Hyper(learning_rate=0.05)
    t = oracle_tables(V, d, "Adagrad")
    dt = tables_from_oracle(t, DeviceTables)
    plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap)
    loss_out = torch.zeros(4, device="cuda:0")
    hip.step_adagrad(plan, dt, _hyper(hp, B), loss_out)
    loss, L, reg = ref.train_step(t, row, col, w, y, hp)
    got = loss_out.cpu().numpy()
    np.testing.assert_allclose(got[:3], [loss, L, reg], rtol=LOSS_RTOL)
    assert_tables_close(dt, t, PARAM_RTOL, PARAM_ATOL)


@pytest.mark.parametrize("optimizer", ["Adagrad", "Adam"])
@pytest.mark.parametrize("B,V,d,cap", [(1024, 300, 64, 32), (3000, 40, 16, 2), (2048, 90, 150, 16)])
def test_single_step_with_reg_multiplicity_one(hip, optimizer, B, V, d, cap):
    """--reg-multiplicity 1 (TF 2.1's `get_losses_for`): the regulariser list enters the loss once —
    kappa halves, loss = L + Reg — against the float64 oracle; and it is not the m = 2 result."""
    from trainer.hip_api import DeviceTables
    row, col, w, y = make_batch(B + d, B, V)
    res = {}
    for m in (1.0, 2.0):
        hp = ref.Hyper(learning_rate=0.05 if optimizer == "Adagrad" else 0.001, reg_mult=m, l2_reg=0.1)
        t = oracle_tables(V, d, optimizer)
        dt = tables_from_oracle(t, DeviceTables)
        plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap)
        loss_out = torch.zeros(4, device="cuda:0")
        if optimizer == "Adagrad":
            hip.step_adagrad(plan, dt, _hyper(hp, B), loss_out)
        else:
            hip.step_adam(plan, dt, _hyper(hp, B), hip.dense_grad_buffer(dt), loss_out)
        loss, L, reg = ref.train_step(t, row, col, w, y, hp)
        np.testing.assert_allclose(loss_out.cpu().numpy()[:3], [loss, L, reg], rtol=LOSS_RTOL)
        np.testing.assert_allclose(loss, L + m * reg, rtol=1e-12)
        assert_tables_close(dt, t, PARAM_RTOL, PARAM_ATOL)
        res[m] = (dt.R.clone(), float(loss_out[0]))
    assert not torch.equal(res[1.0][0], res[2.0][0]) and res[1.0][1] < res[2.0][1]


def test_per_pair_error_coefficients(hip):
    """e_i = 2 w_i (p_i - y_i)/B straight out of the rowpass workspace (e_i rtol 1e-5, atol 1e-7)."""
    from trainer.hip_api import DeviceTables
    B, V, d = 4096, 200, 64
    row, col, w, y = make_batch(5, B, V)
    hp = ref.Hyper()
    t = oracle_tables(V, d, "Adagrad")
    dt = tables_from_oracle(t, DeviceTables)
    plan = hip.build_plan(*to_dev(row, col, w, y), V)
    ws = hip.step_workspace(plan, d)
    hip.rowpass(plan, dt, _hyper(hp, B), ws)       # the standalone row pass is the one that stores e
    torch.cuda.synchronize()
    e_dev = ws[:4 * B].view(torch.float32).cpu().numpy()     # first array of the workspace, row-sorted order
    gr = ref.gradients(t, row, col, w, y, hp)
    want = ref.build_plan(row, col, 32)
    np.testing.assert_allclose(e_dev, gr["e"][want["perm_r"]], rtol=1e-5, atol=1e-7)
    assert dt.global_step == 1          # rowpass advances global_step


@pytest.mark.parametrize("optimizer", ["Adagrad", "Adam"])
@pytest.mark.parametrize("B,V,d", [(1024, 400, 64), (512, 100, 300)])
def test_trajectory(hip, optimizer, B, V, d):
    """30 consecutive steps on fresh batches: the state stays within tolerance of the oracle."""
    from trainer.hip_api import DeviceTables
    hp = ref.Hyper(learning_rate=0.01 if optimizer == "Adam" else 0.05)
    t = oracle_tables(V, d, optimizer)
    dt = tables_from_oracle(t, DeviceTables)
    G = hip.dense_grad_buffer(dt) if optimizer == "Adam" else None
    loss_out = torch.zeros(4, device="cuda:0")
    for s in range(30):
        row, col, w, y = make_batch(1000 + s, B, V)
        plan = hip.build_plan(*to_dev(row, col, w, y), V)
        if optimizer == "Adam":
            hip.step_adam(plan, dt, _hyper(hp, B), G, loss_out)
        else:
            hip.step_adagrad(plan, dt, _hyper(hp, B), loss_out)
        loss, _, _ = ref.train_step(t, row, col, w, y, hp)
        np.testing.assert_allclose(loss_out[0].item(), loss, rtol=5e-5, err_msg="step %d" % s)
    assert_tables_close(dt, t, rtol=2e-4, atol=1e-5)   # 30 steps of fp32 rounding
    if G is not None:
        assert float(G.abs().max()) == 0.0     # the dense apply leaves the gradient buffer zeroed


@pytest.mark.parametrize("form", [2, 3, 4])
@pytest.mark.parametrize("B,V,d,cap", [(8192, 3000, 64, 16), (6000, 500, 300, 8), (4000, 20000, 128, 4)])
def test_trajectory_in_the_fused_step_forms(hip, form, B, V, d, cap):
    """30 consecutive steps on fresh batches through a fused step form (pass kernel applies the ids it holds; slots,
    in place, or twinned row table with version flips carried from step to step): loss of every step and the final
    state within tolerance of the float64 oracle."""
    from trainer.hip_api import DeviceTables
    hp = ref.Hyper(learning_rate=0.05)
    t = oracle_tables(V, d, "Adagrad")
    dt = tables_from_oracle(t, DeviceTables)
    if form == 4:
        dt.enable_twin()
    loss_out = torch.zeros(4, device="cuda:0")
    for s in range(30):
        row, col, w, y = make_batch(3000 + s, B, V)
        plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap).compact(hip.lib, d=1 << 20)   # d: records whatever the fill
        assert plan.r_crec is not None
        hip.step_adagrad(plan, dt, _hyper(hp, B, step_form=form), loss_out)
        loss, _, _ = ref.train_step(t, row, col, w, y, hp)
        np.testing.assert_allclose(loss_out[0].item(), loss, rtol=5e-5, err_msg="step %d" % s)
    if form == 4:
        assert int(dt.R_ver.sum()) > 0          # rows live in both copies by now
    assert_tables_close(dt, t, rtol=2e-4, atol=1e-5)   # 30 steps of fp32 rounding


@pytest.mark.parametrize("B,V,d,cap", [(7, 50, 8, 32), (1024, 300, 64, 32), (1024, 120, 300, 32), (5000, 60, 64, 4),
                                       (1024, 200, 50, 32), (300, 31, 3, 8)])
def test_adam_single_step_dense_decay(hip, B, V, d, cap):
    """Keras-legacy Adam: every row moves, also untouched ones."""
    from trainer.hip_api import DeviceTables
    row, col, w, y = make_batch(B + 11, B, V)
    hp = ref.Hyper(learning_rate=0.001)
    t = oracle_tables(V, d, "Adam")
    # start from a mid-run state so that m, v of untouched rows are non-zero
    rng = np.random.default_rng(3)
    for n in ("R", "C", "br", "bc"):
        setattr(t, "M_" + n, rng.normal(0, 1e-3, getattr(t, n).shape).astype(np.float32).astype(np.float64))
        setattr(t, "V_" + n, (rng.uniform(0, 1e-5, getattr(t, n).shape)).astype(np.float32).astype(np.float64))
    t.step = 41
    dt = tables_from_oracle(t, DeviceTables)
    before = dt.R.clone()
    plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap)
    G = hip.dense_grad_buffer(dt)
    loss_out = torch.zeros(4, device="cuda:0")
    hip.step_adam(plan, dt, _hyper(hp, B), G, loss_out)
    loss, L, reg = ref.train_step(t, row, col, w, y, hp)
    np.testing.assert_allclose(loss_out.cpu().numpy()[:3], [loss, L, reg], rtol=LOSS_RTOL)
    assert_tables_close(dt, t, PARAM_RTOL, PARAM_ATOL)
    untouched = np.setdiff1d(np.arange(V), row)
    if len(untouched):
        assert (dt.R[untouched] != before[untouched]).any()


@pytest.mark.parametrize("B,V,d,cap", [(1024, 300, 64, 32), (6000, 80, 300, 4), (4096, 64, 128, 2), (3000, 150, 50, 16)])
def test_dense_path_equals_sparse_path_bitwise(hip, B, V, d, cap):
    """dense_grad + dense_adagrad (the data-parallel form) == sparse apply, bit for bit: both sum
    the same partials in the same order and G = 0 is an exact no-op for Adagrad."""
    from trainer.hip_api import DeviceTables
    row, col, w, y = make_batch(99, B, V)
    hp = ref.Hyper(learning_rate=0.05)
    t = oracle_tables(V, d, "Adagrad")
    a, b = tables_from_oracle(t, DeviceTables), tables_from_oracle(t, DeviceTables)
    plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap)
    h = _hyper(hp, B)
    la, lb = torch.zeros(4, device="cuda:0"), torch.zeros(4, device="cuda:0")
    hip.step_adagrad(plan, a, h, la)
    G = hip.dense_grad_buffer(b)
    hip.rowpass(plan, b, h)
    hip.colpass(plan, b, h)
    hip.dense_grad(plan, b, h, G)
    hip.dense_adagrad(b, h, G, lb)
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n
        assert torch.equal(a.s1[n], b.s1[n]), "slot " + n
    assert torch.equal(a.scalars, b.scalars)
    # the loss scalars are formed by two different kernels (fma contraction may differ by an ulp)
    np.testing.assert_allclose(la.cpu().numpy(), lb.cpu().numpy(), rtol=1e-6)
    assert float(G.abs().max()) == 0.0


@pytest.mark.parametrize("B,V,d,cap", [(200, 1000, 64, 32), (1024, 3000, 50, 16), (64, 500, 300, 32), (300, 5000, 8, 2),
                                       (400, 2000, 128, 2)])
def test_adam_fused_step_equals_the_dense_form_bitwise(hip, B, V, d, cap):
    """glove_step_adam_f32 on a batch that touches a minority of the rows (passes that mark the ids + ONE kernel
    that applies them and decays every other row) == passes + dense_grad + dense_adam, bit for bit; the
    gradient buffer is all zero again afterwards; both within tolerance of the oracle."""
    from trainer.hip_api import DeviceTables
    row, col, w, y = make_batch(B + d, B, V)
    row[::3] = 3                                   # one id far over the heavy threshold at small caps
    assert 2 * B <= 2 * V
    hp = ref.Hyper(learning_rate=0.001)
    t = oracle_tables(V, d, "Adam")
    rng = np.random.default_rng(5)
    for n in ("R", "C", "br", "bc"):
        setattr(t, "M_" + n, rng.normal(0, 1e-3, getattr(t, n).shape).astype(np.float32).astype(np.float64))
        setattr(t, "V_" + n, (rng.uniform(0, 1e-5, getattr(t, n).shape)).astype(np.float32).astype(np.float64))
    t.step = 17
    a, b = tables_from_oracle(t, DeviceTables), tables_from_oracle(t, DeviceTables)
    plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap)
    h = _hyper(hp, B)
    Ga, Gb = hip.dense_grad_buffer(a), hip.dense_grad_buffer(b)
    la, lb = torch.zeros(4, device="cuda:0"), torch.zeros(4, device="cuda:0")
    for _ in range(2):                             # twice: the marks of step 1 must be gone in step 2
        hip.step_adam(plan, a, h, Ga, la)
        hip.passes(plan, b, h)
        hip.dense_grad(plan, b, h, Gb)
        hip.dense_adam(b, h, Gb, lb)
        assert float(Ga.abs().max()) == 0.0
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n
        assert torch.equal(a.s1[n], b.s1[n]) and torch.equal(a.s2[n], b.s2[n]), "slots " + n
    assert torch.equal(a.scalars, b.scalars) and a.global_step == b.global_step == 19
    np.testing.assert_allclose(la.cpu().numpy(), lb.cpu().numpy(), rtol=1e-6)
    ref.train_step(t, row, col, w, y, hp)
    loss, L, reg = ref.train_step(t, row, col, w, y, hp)
    np.testing.assert_allclose(la.cpu().numpy()[:3], [loss, L, reg], rtol=LOSS_RTOL)
    assert_tables_close(a, t, PARAM_RTOL, PARAM_ATOL)


@pytest.mark.parametrize("optimizer", ["Adagrad", "Adam"])
@pytest.mark.parametrize("B,V,d,cap", [(1024, 300, 64, 32), (3000, 40, 16, 2), (512, 2000, 50, 16), (2048, 64, 300, 8)])
def test_logistic_head_single_step(hip, optimizer, B, V, d, cap):
    """The pos / neg logistic heads of logistic_matrix_factorisation.py:48-54 as an epilogue of the same kernels:
    w = positive weight, y = negative weight (non-negative), logits from large negative to large positive."""
    from trainer.hip_api import DeviceTables
    row, col, pos, neg = make_batch(B + 7 * d, B, V)
    neg = (np.abs(neg) * 0.5).astype(np.float32)
    hp = ref.Hyper(learning_rate=0.05 if optimizer == "Adagrad" else 0.001, head=1, neg_factor=0.8)
    t = oracle_tables(V, d, optimizer)
    t.R *= 30.0                                     # spread the logits: both softplus branches, saturated sigmoids
    t.g = t.dtype(0.3)
    dt = tables_from_oracle(t, DeviceTables)
    plan = hip.build_plan(*to_dev(row, col, pos, neg), V, chunk_cap=cap)
    loss_out = torch.zeros(4, device="cuda:0")
    if optimizer == "Adagrad":
        # the e_i of the head, then the step; the dense (data-parallel) form must agree bit for bit
        ws = hip.step_workspace(plan, dt.d)
        hip.rowpass(plan, dt, _hyper(hp, B), ws)
        e_dev = ws[:4 * B].view(torch.float32).cpu().numpy().copy()
        dt.step.fill_(t.step)
        gr = ref.gradients(t, row, col, pos, neg, hp)
        np.testing.assert_allclose(e_dev, gr["e"][ref.build_plan(row, col, cap)["perm_r"]], rtol=1e-5, atol=1e-7)
        twin = tables_from_oracle(t, DeviceTables)
        G = hip.dense_grad_buffer(twin)
        hip.passes(plan, twin, _hyper(hp, B))
        hip.dense_grad(plan, twin, _hyper(hp, B), G)
        hip.dense_adagrad(twin, _hyper(hp, B), G)
        hip.step_adagrad(plan, dt, _hyper(hp, B), loss_out)
        assert torch.equal(dt.R, twin.R) and torch.equal(dt.bc, twin.bc) and torch.equal(dt.scalars, twin.scalars)
    else:
        hip.step_adam(plan, dt, _hyper(hp, B), hip.dense_grad_buffer(dt), loss_out)
    loss, L, reg = ref.train_step(t, row, col, pos, neg, hp)
    np.testing.assert_allclose(loss_out.cpu().numpy()[:3], [loss, L, reg], rtol=LOSS_RTOL)
    assert_tables_close(dt, t, PARAM_RTOL, PARAM_ATOL)


def _random_case(seed):
    rng = np.random.default_rng(seed)
    V = int(rng.choice([1, 2, 7, 60, 333, 2000, 30000]))
    B = int(rng.choice([1, 5, 63, 64, 65, 700, 4096, 4097, 9000]))
    d = int(rng.choice([1, 4, 6, 8, 20, 50, 64, 100, 128, 200, 300, 520]))
    cap = int(rng.choice([1, 2, 5, 8, 16, 31, 32]))
    skew = float(rng.choice([0.0, 1.0, 1.6]))          # 0: uniform ids, else Zipf exponent (heavy duplicates)
    if skew:
        ranks = np.arange(1, V + 1, dtype=np.float64) ** -skew
        pdf = ranks / ranks.sum()
        row, col = rng.choice(V, B, p=pdf).astype(np.int32), rng.choice(V, B, p=pdf).astype(np.int32)
    else:
        row, col = rng.integers(0, V, B).astype(np.int32), rng.integers(0, V, B).astype(np.int32)
    w = rng.uniform(0, 1, B).astype(np.float32)
    w[rng.uniform(size=B) < 0.1] = 0.0                  # some pairs weigh nothing
    y = rng.normal(0, 2, B).astype(np.float32)
    return dict(V=V, B=B, d=d, cap=cap, row=row, col=col, w=w, y=y, optimizer=str(rng.choice(["Adagrad", "Adam"])),
                head=int(rng.integers(0, 2)), nf=float(rng.uniform(0.2, 2.0)), steps=int(rng.integers(1, 4)))


@pytest.mark.parametrize("seed", range(int(os.environ.get("GLOVE_FUZZ_CASES", "40"))))
def test_randomized_step_parity(hip, seed):
    """Seeded random shapes through the whole step: vocabularies from 1 id up, batches around the builder and tile
    boundaries, every kernel shape incl. padded embedding sizes, uniform and Zipf ids, zero weights, both
    optimizers and both heads, one to three consecutive steps on the same batch."""
    from trainer.hip_api import DeviceTables
    c = _random_case(1000 + seed)
    y = np.abs(c["y"]).astype(np.float32) if c["head"] else c["y"]
    hp = ref.Hyper(learning_rate=0.05 if c["optimizer"] == "Adagrad" else 0.001, head=c["head"], neg_factor=c["nf"])
    t = oracle_tables(c["V"], c["d"], c["optimizer"], seed=seed)
    dt = tables_from_oracle(t, DeviceTables)
    plan = hip.build_plan(*to_dev(c["row"], c["col"], c["w"], y), c["V"], chunk_cap=c["cap"])
    G = hip.dense_grad_buffer(dt) if c["optimizer"] == "Adam" else None
    loss_out = torch.zeros(4, device="cuda:0")
    for _ in range(c["steps"]):
        if G is None:
            hip.step_adagrad(plan, dt, _hyper(hp, c["B"]), loss_out)
        else:
            hip.step_adam(plan, dt, _hyper(hp, c["B"]), G, loss_out)
        loss, L, reg = ref.train_step(t, c["row"], c["col"], c["w"], y, hp)
    info = {k: c[k] for k in ("V", "B", "d", "cap", "optimizer", "head", "steps")}
    np.testing.assert_allclose(loss_out.cpu().numpy()[:3], [loss, L, reg], rtol=LOSS_RTOL, atol=1e-7, err_msg=str(info))
    assert_tables_close(dt, t, PARAM_RTOL * c["steps"], PARAM_ATOL * c["steps"])


@pytest.mark.parametrize("seed", range(int(os.environ.get("GLOVE_FUZZ_CASES", "40"))))
def test_randomized_forms_agree_bitwise(hip, seed):
    """The same random step through the two forms the product has of it must give the same bits: Adagrad sparse
    apply vs dense gradient + dense apply (the data-parallel form), Adam through glove_step_adam_f32 (two-launch
    form for small batches) vs passes + dense gradient + dense sweep."""
    from trainer.hip_api import DeviceTables
    c = _random_case(50000 + seed)
    y = np.abs(c["y"]).astype(np.float32) if c["head"] else c["y"]
    hp = ref.Hyper(learning_rate=0.05 if c["optimizer"] == "Adagrad" else 0.001, head=c["head"], neg_factor=c["nf"])
    t = oracle_tables(c["V"], c["d"], c["optimizer"], seed=seed)
    if c["optimizer"] == "Adam":                           # mid-run slots, so that decay of untouched rows shows
        rng = np.random.default_rng(seed)
        for n in ("R", "C", "br", "bc"):
            setattr(t, "M_" + n, rng.normal(0, 1e-3, getattr(t, n).shape).astype(np.float32).astype(np.float64))
            setattr(t, "V_" + n, rng.uniform(0, 1e-5, getattr(t, n).shape).astype(np.float32).astype(np.float64))
        t.step = int(rng.integers(0, 1000))
    a, b = tables_from_oracle(t, DeviceTables), tables_from_oracle(t, DeviceTables)
    plan = hip.build_plan(*to_dev(c["row"], c["col"], c["w"], y), c["V"], chunk_cap=c["cap"])
    if seed % 2:
        plan = plan.compact(hip.lib)           # resident form: host counts, per-chunk records when they pay
    h = _hyper(hp, c["B"], step_form=1)
    Ga, Gb = hip.dense_grad_buffer(a), hip.dense_grad_buffer(b)
    fused = [(tables_from_oracle(t, DeviceTables), _hyper(hp, c["B"], step_form=f)) for f in (2, 3, 4)] \
        if c["optimizer"] == "Adagrad" else []
    if fused:
        fused[2][0].enable_twin()              # form 4 steps on a twinned row table
    exch = None
    if fused and seed % 2:
        # the touched-rows exchange of the data-parallel form on one rank: packing passes + pack_rest -> count, combine,
        # apply from the list.  With chunk records it sums like the fused forms, without them like the two-launch step
        from trainer.stepper import HipBackend, Stepper
        et = tables_from_oracle(t, DeviceTables)
        exch = Stepper(HipBackend("cuda:0"), et, dict(l2_reg=hp.l2_reg, reg_mult=hp.reg_mult, learning_rate=hp.learning_rate,
                                                      epsilon=hp.epsilon, head=hp.head, neg_factor=hp.neg_factor),
                       c["B"], exchange="rows")
        exch.prepare([plan])
        assert exch.rows
    for _ in range(c["steps"]):
        for ft, fh in fused:                   # the forms whose single-chunk ids are applied by the pass kernel itself
            hip.step_adagrad(plan, ft, fh)
        if exch:
            exch.step(plan)
        if c["optimizer"] == "Adagrad":
            hip.step_adagrad(plan, a, h)
            hip.passes(plan, b, h)
            hip.dense_grad(plan, b, h, Gb)
            hip.dense_adagrad(b, h, Gb)
        else:
            hip.step_adam(plan, a, h, Ga)
            hip.passes(plan, b, h)
            hip.dense_grad(plan, b, h, Gb)
            hip.dense_adam(b, h, Gb)
    info = str({k: c[k] for k in ("V", "B", "d", "cap", "optimizer", "head", "steps")})
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n + " " + info
        assert torch.equal(a.s1[n], b.s1[n]), "slot1 " + n + " " + info
        if n in a.s2:
            assert torch.equal(a.s2[n], b.s2[n]), "slot2 " + n + " " + info
    assert torch.equal(a.scalars, b.scalars) and a.global_step == b.global_step, info
    assert float(Ga.abs().max()) == 0.0 and float(Gb.abs().max()) == 0.0, info
    if fused:        # the fused forms sum an id's pairs run by run instead of chunk by chunk: same bits among themselves
        _assert_same_bits(fused[0][0], fused[1][0], "step_form 2 vs 3 " + info)
        _assert_same_bits(fused[0][0], fused[2][0], "step_form 2 vs 4 " + info)
        _assert_tables_agree(a, fused[0][0], 2e-5 * c["steps"], 2e-6 * c["steps"], "step_form 2 vs 1 " + info)
    if exch:
        _assert_same_bits(fused[0][0] if plan.r_crec is not None else a, exch.tables, "rows exchange " + info)
        assert int(exch.bufs["mark"].abs().max()) == 0, info


@pytest.mark.parametrize("B,V,d,cap", STEP_CASES + [(4096, 4096, 64, 32), (9000, 20000, 300, 16), (30000, 300000, 128, 16),
                                                    (50000, 3000, 64, 8), (20000, 5, 32, 8)])
def test_step_forms_agree(hip, B, V, d, cap):
    """glove_hyper.step_form.  The fused forms (ids whose chunks one lane group holds are applied by the pass kernel:
    one-pass form with the new rows through the slots, three-launch form with the col side in place) give the same
    bits as each other and are bitwise repeatable; against the two-launch step they differ only in the order the
    pairs of a multi-chunk id are summed (fp32 rounding), and every form matches the float64 oracle."""
    from trainer.hip_api import DeviceTables
    row, col, w, y = make_batch(B * 3 + d, B, V)
    hp = ref.Hyper(learning_rate=0.05)
    t = oracle_tables(V, d, "Adagrad")
    plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap).compact(hip.lib)
    if plan.r_crec is None:                  # nearly empty chunks carry no records: the library then takes the two-launch form
        plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=max(1, min(cap, 2))).compact(hip.lib)
    runs = {}
    for form in (1, 2, 3, 2, 3):
        dt = tables_from_oracle(t, DeviceTables)
        loss_out = torch.zeros(4, device="cuda:0")
        for k in range(3):
            hip.step_adagrad(plan, dt, _hyper(hp, B, step_form=form), loss_out)
            if k == 0 and form not in runs:
                t1 = t.copy()
                want = ref.train_step(t1, row, col, w, y, hp)
                np.testing.assert_allclose(loss_out.cpu().numpy()[:3], want, rtol=LOSS_RTOL)
                assert_tables_close(dt, t1, PARAM_RTOL, PARAM_ATOL)
        if form in runs:                     # second run of a fused form: bitwise repeatable
            _assert_same_bits(runs[form][0], dt, "repeat of form %d" % form)
            assert torch.equal(runs[form][1], loss_out)
        runs[form] = (dt, loss_out)
    _assert_same_bits(runs[2][0], runs[3][0], "forms 2 and 3")
    assert torch.equal(runs[2][1], runs[3][1])
    _assert_tables_agree(runs[1][0], runs[2][0], 5e-5, 5e-6, "forms 1 and 2")
    np.testing.assert_allclose(runs[1][1].cpu().numpy(), runs[2][1].cpu().numpy(), rtol=1e-5)


def test_step_is_bitwise_repeatable(hip):
    from trainer.hip_api import DeviceTables
    B, V, d = 20000, 500, 64
    row, col, w, y = make_batch(17, B, V)
    hp = ref.Hyper(learning_rate=0.05)
    t = oracle_tables(V, d, "Adagrad")
    plan = hip.build_plan(*to_dev(row, col, w, y), V)
    outs = []
    for _ in range(2):
        dt = tables_from_oracle(t, DeviceTables)
        for _ in range(3):
            hip.step_adagrad(plan, dt, _hyper(hp, B))
        outs.append(dt)
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(outs[0], n), getattr(outs[1], n))
    assert torch.equal(outs[0].scalars, outs[1].scalars)


def test_heavy_ids_take_the_workgroup_path(hip):
    """One id owning most of the batch (Zipf head) -> hundreds of chunks -> LDS-queue path."""
    from trainer.hip_api import DeviceTables
    B, V, d = 30000, 40, 64
    rng = np.random.default_rng(2)
    row = np.where(rng.random(B) < 0.7, 3, rng.integers(0, V, B)).astype(np.int32)
    col = np.where(rng.random(B) < 0.5, 7, rng.integers(0, V, B)).astype(np.int32)
    col[row == col] = (col[row == col] + 1) % V
    _, _, w, y = make_batch(4, B, V)
    hp = ref.Hyper(learning_rate=0.05)
    t = oracle_tables(V, d, "Adagrad")
    dt = tables_from_oracle(t, DeviceTables)
    plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=8)
    loss_out = torch.zeros(4, device="cuda:0")
    hip.step_adagrad(plan, dt, _hyper(hp, B), loss_out)
    loss, _, _ = ref.train_step(t, row, col, w, y, hp)
    np.testing.assert_allclose(loss_out[0].item(), loss, rtol=LOSS_RTOL)
    assert_tables_close(dt, t, 2e-5, PARAM_ATOL)


def test_untouched_rows_do_not_move_under_adagrad(hip):
    from trainer.hip_api import DeviceTables
    B, V, d = 256, 5000, 64
    row, col, w, y = make_batch(8, B, V, zipf=False)
    dt = DeviceTables(V, d, "Adagrad", seed=3)
    R0, A0 = dt.R.clone(), dt.s1["R"].clone()
    plan = hip.build_plan(*to_dev(row, col, w, y), V)
    hip.step_adagrad(plan, dt, _hyper(ref.Hyper(learning_rate=0.05), B))
    untouched = torch.from_numpy(np.setdiff1d(np.arange(V), row)).cuda()
    assert torch.equal(dt.R[untouched], R0[untouched]) and torch.equal(dt.s1["R"][untouched], A0[untouched])
    touched = torch.from_numpy(np.unique(row)).cuda()
    assert (dt.R[touched] != R0[touched]).any(dim=1).all()


def test_eval_metrics(hip):
    from trainer.hip_api import DeviceTables
    B, V, d = 10000, 300, 64
    row, col, w, y = make_batch(21, B, V)
    t = oracle_tables(V, d, "Adagrad")
    t.g = np.float64(np.float32(0.3))
    dt = tables_from_oracle(t, DeviceTables)
    sums = hip.eval_sums(*to_dev(row, col, w, y), dt).cpu().numpy()
    want = ref.eval_metrics(t, row, col, w, y)
    np.testing.assert_allclose(sums[0] / sums[1], want["average_loss"], rtol=1e-5)
    np.testing.assert_allclose(sums[1], want["weight_sum"], rtol=1e-6)
    np.testing.assert_allclose(sums[2] / sums[1], want["prediction_mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sums[3] / sums[1], want["label_mean"], rtol=1e-6)
    assert dt.global_step == 0      # EVAL mode leaves the step alone


def test_eval_metrics_of_the_logistic_heads(hip):
    """glove_eval_logistic_f32: the BinaryClassHead sums of the two heads (logistic_matrix_factorisation.py:50-54) over a
    batch — sum pos xent(p, 1), sum pos, sum neg xent(p, 0), sum neg, sum pos sigmoid(p), sum neg sigmoid(p) — against float64."""
    from trainer.hip_api import DeviceTables
    B, V, d = 10000, 300, 64
    row, col, pos, neg = make_batch(23, B, V)
    neg = np.abs(neg).astype(np.float32)
    t = oracle_tables(V, d, "Adagrad")
    t.g = np.float64(np.float32(-0.2))
    dt = tables_from_oracle(t, DeviceTables)
    sums = hip.eval_sums_logistic(*to_dev(row, col, pos, neg), dt).cpu().numpy()
    R, Cm = np.asarray(t.R, np.float64), np.asarray(t.C, np.float64)
    p = (R[row] * Cm[col]).sum(1) + np.asarray(t.br, np.float64).reshape(-1)[row] + np.asarray(t.bc, np.float64).reshape(-1)[col] + float(t.g)
    softplus = lambda x: np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))
    sig = 1.0 / (1.0 + np.exp(-p))
    pos64, neg64 = pos.astype(np.float64), neg.astype(np.float64)
    want = [(pos64 * softplus(-p)).sum(), pos64.sum(), (neg64 * softplus(p)).sum(), neg64.sum(), (pos64 * sig).sum(), (neg64 * sig).sum()]
    np.testing.assert_allclose(sums, want, rtol=1e-5)
    assert dt.global_step == 0


@pytest.mark.parametrize("V,d,k", [(500, 64, 20), (1000, 300, 5), (64, 8, 64), (100000, 64, 20), (3000, 128, 33)])
def test_topk_cosine(hip, V, d, k):
    rng = np.random.default_rng(5)
    R = rng.normal(size=(V, d)).astype(np.float32)
    R[7] = R[3]                          # exact tie -> lower index first, like tf.math.top_k
    q = np.array([0, 3, 7, V - 1, 5], np.int32)
    sims, idx = hip.topk_cosine(*to_dev(R, q), k)
    want_s, want_i = ref.cosine_topk(R.astype(np.float64), q, k)
    np.testing.assert_allclose(sims.cpu().numpy(), want_s, rtol=1e-5, atol=1e-6)
    got_i = idx.cpu().numpy()
    # identical neighbour sets; order may only differ where similarities agree to rounding
    for a in range(len(q)):
        mism = got_i[a] != want_i[a]
        if mism.any():
            np.testing.assert_allclose(want_s[a][mism], np.sort(want_s[a][mism])[::-1])
            assert set(got_i[a]) == set(want_i[a]) or abs(want_s[a, -1] - ref.cosine_topk(
                R.astype(np.float64), q[a:a + 1], k + 1)[0][0, -1]) < 1e-6
    assert got_i[0, 0] == 0 and got_i[1, 0] == 3 and got_i[2, 0] == 3   # self first; tie -> index 3 before 7


def test_topk_cosine_many_queries_and_segments(hip):
    """More queries than one 128-row MFMA tile, a vocabulary cut into several top-k segments, ragged edges."""
    V, d, k, n = 40003, 52, 20, 301
    rng = np.random.default_rng(9)
    R = rng.normal(size=(V, d)).astype(np.float32)
    q = rng.integers(0, V, n).astype(np.int32)
    sims, idx = hip.topk_cosine(*to_dev(R, q), k)
    want_s, want_i = ref.cosine_topk(R.astype(np.float64), q, k)
    np.testing.assert_allclose(sims.cpu().numpy(), want_s, rtol=1e-5, atol=1e-6)
    got_i = idx.cpu().numpy()
    assert (got_i[:, 0] == q).all()                      # every token is its own nearest neighbour
    assert (got_i == want_i).mean() > 0.999              # continuous data: ties only by rounding


def test_argument_errors_are_reported_not_swallowed(hip):
    from trainer.hip_api import DeviceTables, GloveHipError
    with pytest.raises(ValueError):
        DeviceTables(10, 0, "Adagrad")
    with pytest.raises(ValueError):
        DeviceTables(10, 8, "Lion")                # not a Keras 2.11 optimizer name
    row, col, w, y = make_batch(1, 64, 10)
    dt = DeviceTables(10, 8, "Adagrad", seed=0)
    bad = DeviceTables(10, 8, "Adagrad", seed=0)
    bad.struct().d_model = 9                       # more model columns than the row stride holds
    with pytest.raises(GloveHipError, match="BADARG"):
        hip.step_adagrad(hip.build_plan(*to_dev(row, col, w, y), 10), bad, _hyper(ref.Hyper(), 64))
    plan = hip.build_plan(*to_dev(row, col, w, y), 10)
    tiny = torch.empty(16, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(GloveHipError, match="WORKSPACE"):
        hip.step_adagrad(plan, dt, _hyper(ref.Hyper(), 64), ws=tiny)
    with pytest.raises(GloveHipError):
        hip.build_plan(*[a.cpu() for a in to_dev(row, col, w, y)], 10)


@pytest.mark.parametrize("B,V,d,W", [(3000, 101, 64, 2), (2000, 77, 50, 3), (5000, 403, 128, 4), (900, 31, 300, 8)])
def test_row_sharded_pieces_on_one_gpu(hip, B, V, d, W):
    """BASELINE config 5 kernels on one GPU: two virtual ranks, each with half of the row table (V_row < V, local
    row ids) and a replica of the col table; the col halves of their gradient buffers are summed by hand where
    the all-reduce would run.  Result == oracle step on the union batch; and with a single shard the mixed
    form (row side sparse, col side dense) is bit-identical to the plain sparse step."""
    from trainer.hip_api import DeviceTables, make_hyper
    from trainer.stepper import owned_rows
    # odd V: unequal shards, unaligned bias sections; d = 50: padded rows; W up to 8 virtual ranks
    hp = ref.Hyper(learning_rate=0.05)
    t = oracle_tables(V, d, "Adagrad")
    full = tables_from_oracle(t, DeviceTables)
    batches = [make_batch(70 + r, B, V) for r in range(W)]
    joint = [np.concatenate([b[i] for b in batches]) for i in range(4)]
    kw = dict(l2_reg=hp.l2_reg, reg_mult=hp.reg_mult, learning_rate=hp.learning_rate, batch_size=W * B)
    ranks = []
    for r in range(W):
        rows = np.arange(r, V, W)
        dt = DeviceTables(V, d, "Adagrad", seed=0, V_row=owned_rows(V, W, r))
        assert dt.R.shape[0] == len(rows)
        idx = torch.from_numpy(rows).cuda()
        dt.R.copy_(full.R[idx]); dt.br.copy_(full.br[idx]); dt.s1["R"].copy_(full.s1["R"][idx]); dt.s1["br"].copy_(full.s1["br"][idx])   # padded widths agree
        dt.C.copy_(full.C); dt.bc.copy_(full.bc); dt.s1["C"].copy_(full.s1["C"]); dt.s1["bc"].copy_(full.s1["bc"])
        dt.scalars.copy_(full.scalars)
        mine = joint[0] % W == r                         # route the union batch by row owner
        plan = hip.build_plan(*to_dev(joint[0][mine] // W, joint[1][mine], joint[2][mine], joint[3][mine]), V)
        G = hip.dense_grad_buffer(dt)
        hip.rowpass(plan, dt, make_hyper(**kw)); hip.colpass(plan, dt, make_hyper(**kw))
        hip.dense_grad(plan, dt, make_hyper(sides=2, **kw), G)
        hip.apply_adagrad(plan, dt, make_hyper(sides=1, **kw))
        ranks.append((dt, G, rows))
    lay = hip.grad_layout(ranks[0][0])
    assert lay["G_C"] % 4 == 0 and lay["tail"] % 4 == 0
    assert float(ranks[0][1][:lay["G_C"]].abs().max()) == 0.0           # sides = 2 left the row half untouched
    total = sum(G[hip.grad_layout(dt)["G_C"]:] for dt, G, _ in ranks)            # the "all-reduce"
    loss_out = torch.zeros(4, device="cuda:0")
    for dt, G, _ in ranks:
        G[hip.grad_layout(dt)["G_C"]:] = total
        hip.dense_adagrad(dt, make_hyper(sides=2, **kw), G, loss_out)
        assert float(G.abs().max()) == 0.0
    loss, L, reg = ref.train_step(t, *joint, hp)
    np.testing.assert_allclose(loss_out.cpu().numpy()[:3], [loss, L, reg], rtol=LOSS_RTOL)
    for dt, _, rows in ranks:
        np.testing.assert_allclose(dt.embeddings("R").cpu().numpy(), t.R[rows], rtol=PARAM_RTOL, atol=PARAM_ATOL)
        np.testing.assert_allclose(dt.br.cpu().numpy(), t.br[rows], rtol=PARAM_RTOL, atol=PARAM_ATOL)
        np.testing.assert_allclose(dt.s1["R"][:, :d].cpu().numpy(), t.A_R[rows], rtol=PARAM_RTOL, atol=PARAM_ATOL)
        np.testing.assert_allclose(dt.embeddings("C").cpu().numpy(), t.C, rtol=PARAM_RTOL, atol=PARAM_ATOL)
        np.testing.assert_allclose(dt.bc.cpu().numpy(), t.bc, rtol=PARAM_RTOL, atol=PARAM_ATOL)
        np.testing.assert_allclose(dt.scalars[0].item(), t.g, rtol=PARAM_RTOL, atol=PARAM_ATOL)
    assert torch.equal(ranks[0][0].C, ranks[1][0].C)

    # single shard: mixed form == sparse step, bit for bit
    t2 = oracle_tables(V, d, "Adagrad")
    a, b = tables_from_oracle(t2, DeviceTables), tables_from_oracle(t2, DeviceTables)
    plan = hip.build_plan(*to_dev(*batches[0]), V)
    kw1 = dict(kw, batch_size=B)
    hip.step_adagrad(plan, a, make_hyper(**kw1))
    G = hip.dense_grad_buffer(b)
    hip.rowpass(plan, b, make_hyper(**kw1)); hip.colpass(plan, b, make_hyper(**kw1))
    hip.dense_grad(plan, b, make_hyper(sides=2, **kw1), G)
    hip.apply_adagrad(plan, b, make_hyper(sides=1, **kw1))
    hip.dense_adagrad(b, make_hyper(sides=2, **kw1), G)
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n
    assert torch.equal(a.scalars, b.scalars)


@pytest.mark.parametrize("exchange,workload", [("dense", "zipf_v2m_d128"), ("rows", "zipf_v2m_d128"), ("rows", "zipf_v400k_d300")])
def test_full_size_row_sharded_two_virtual_ranks(hip, exchange, workload):
    """BASELINE config 5 at its workload (V = 2 M, d = 128, 1 M nonzeros per step) — and config 4's (V = 400 k, d = 300),
    which `bench.py --gpus N` also runs in the sharded forms — as two virtual ranks on one GPU:
    each holds half of the row table (V_row = 1 M < V, local row ids: the union batch routed by row owner) and a
    replica of the col table; the col side is exchanged by hand where the collective would run — the summed dense
    halves, or the two packed lists combined in rank order.  Result == the single-GPU step on the union batch
    (itself checked against the float64 restatement in test_full_size_spot_check_against_oracle) within the fp32
    rounding of summing two ranks' col gradients."""
    from trainer import synthetic
    from trainer.hip_api import DeviceTables, make_hyper
    W, B = 2, 1048576
    wl = synthetic.make_workload(workload, seed=4, device="cuda:0", work_device="cuda:0")
    V, d = wl["V"], wl["d"]
    row, col, w, y = (wl[k][:B].contiguous() for k in ("row", "col", "w", "y"))
    full = DeviceTables(V, d, "Adagrad", seed=6)
    kw = dict(learning_rate=0.05, batch_size=B)
    ranks = []
    for r in range(W):
        dt = DeviceTables(V, d, "Adagrad", seed=0, V_row=V // W)
        dt.R.copy_(full.R[r::W]); dt.br.copy_(full.br[r::W])
        dt.C.copy_(full.C); dt.bc.copy_(full.bc)
        mine = row % W == r
        plan = hip.build_plan((row[mine] // W).contiguous(), col[mine].contiguous(), w[mine].contiguous(), y[mine].contiguous(),
                              V, chunk_cap=0, compact=True, V_row=V // W)
        assert int(plan.counts[5]) == 0                                       # no local row id outside the shard
        ranks.append((dt, plan))
    loss_out = torch.zeros(4, device="cuda:0")
    if exchange == "dense":
        Gs = []
        for dt, plan in ranks:                   # (the virtual ranks share one step workspace: a rank's passes and their
            G = hip.dense_grad_buffer(dt)        # consumers run back to back)
            hip.passes(plan, dt, make_hyper(**kw))
            hip.dense_grad(plan, dt, make_hyper(sides=2, **kw), G)
            hip.apply_adagrad(plan, dt, make_hyper(sides=1, **kw))
            Gs.append(G)
        off = hip.grad_layout(ranks[0][0])["G_C"]
        total = Gs[0][off:] + Gs[1][off:]
        for (dt, _), G in zip(ranks, Gs):
            G[off:] = total
            hip.dense_adagrad(dt, make_hyper(sides=2, **kw), G, loss_out)
    else:
        cap = 1 + max(p.host_counts[3] for _, p in ranks)
        lists = torch.zeros(W, cap, ranks[0][0].d + 4, device="cuda:0")     # (the stored row width: hip_api.row_width)
        for r, (dt, plan) in enumerate(ranks):
            hip.passes(plan, dt, make_hyper(**kw))
            hip.pack_grad(plan, dt, make_hyper(sides=2, **kw), lists[r])
            hip.apply_adagrad(plan, dt, make_hyper(sides=1, **kw))
        for dt, _ in ranks:
            G, mark = hip.dense_grad_buffer(dt), torch.zeros(dt.V_row + V, dtype=torch.int32, device="cuda:0")
            G.fill_(float("nan"))                                              # never read before it is written
            pl = [hip.packed_list(lists[r]) for r in range(W)]
            for r in range(W):
                hip.combine_packed(pl[r], r, dt, G, mark, cap)
            hip.apply_packed(pl, dt, make_hyper(sides=2, **kw), G, mark, None, loss_out, cap)
            assert int(mark.abs().max()) == 0
            del G, mark
    plan = hip.build_plan(row, col, w, y, V, chunk_cap=0, compact=True)
    want_loss = torch.zeros(4, device="cuda:0")
    hip.step_adagrad(plan, full, make_hyper(step_form=1, **kw), want_loss)
    np.testing.assert_allclose(loss_out.cpu().numpy()[:3], want_loss.cpu().numpy()[:3], rtol=2e-5)
    for r, (dt, _) in enumerate(ranks):
        assert torch.allclose(dt.R, full.R[r::W], rtol=2e-5, atol=2e-6) and torch.allclose(dt.br, full.br[r::W], rtol=2e-5, atol=2e-6)
        assert torch.equal(dt.R, full.R[r::W])                                # the row side has no cross-rank sum: same bits
        assert torch.allclose(dt.C, full.C, rtol=2e-5, atol=2e-6) and torch.allclose(dt.bc, full.bc, rtol=2e-5, atol=2e-6)
        assert abs(dt.global_bias - full.global_bias) <= 2e-5 * abs(full.global_bias) + 1e-7
    assert torch.equal(ranks[0][0].C, ranks[1][0].C)


@pytest.mark.parametrize("B,V,d,cap", STEP_CASES[:14:2] + [(9000, 20000, 300, 16), (30000, 300000, 128, 16), (50000, 3000, 64, 8),
                                                          (20000, 5, 32, 8), (700, 60, 301, 32)])
def test_twinned_row_table_step_equals_the_three_launch_form_bitwise(hip, B, V, d, cap):
    """GLOVE_STEP_FUSED_TWIN: the row side's new rows go into the other copy of a twinned row table and the apply launch
    flips versions — the arithmetic of the three-launch form, so the tables agree bit for bit after every step, whatever
    mixture of copies is current; reading R / br (or any other entry point) first restores the plain form."""
    from trainer.hip_api import DeviceTables
    t = oracle_tables(V, d, "Adagrad")
    a, b = tables_from_oracle(t, DeviceTables), tables_from_oracle(t, DeviceTables)
    b.enable_twin()
    assert b.R_ver is not None and b._R.shape[0] == 2 * V and b.R.shape[0] == V
    hp = ref.Hyper(learning_rate=0.05)
    batches = [make_batch(B * 3 + d + k, B, V) for k in range(3)]
    plans = []
    for bt in batches:
        pl = hip.build_plan(*to_dev(*bt), V, chunk_cap=cap).compact(hip.lib)
        if pl.r_crec is None:
            pl = hip.build_plan(*to_dev(*bt), V, chunk_cap=max(1, min(cap, 2))).compact(hip.lib)
        plans.append(pl)
    la, lb = torch.zeros(4, device="cuda:0"), torch.zeros(4, device="cuda:0")
    for k in (0, 1, 2, 1, 0, 2, 2):
        hip.step_adagrad(plans[k], a, _hyper(hp, B, step_form=3), la)
        hip.step_adagrad(plans[k], b, _hyper(hp, B, step_form=4), lb)
        assert torch.equal(la, lb)
        if k == 1:                            # look at the tables mid-run: canonicalises, stepping goes on from there
            _assert_same_bits(a, b, "mid-run")
    flipped = int(b.R_ver.sum())              # raw state: some rows live in the second copy ...
    assert plans[0].r_crec is None or V < 8 or flipped > 0
    _assert_same_bits(a, b)                   # ... reading R / br brings them home
    assert int(b.R_ver.sum()) == 0
    # an oracle step from here (state_dict of the twinned tables is the plain model)
    sd = b.state_dict()
    assert sd["R"].shape == (V, d) and torch.equal(sd["R"], a.state_dict()["R"])
    # without a twin the form is refused, not silently replaced
    from trainer.hip_api import GloveHipError
    with pytest.raises(GloveHipError):
        hip.step_adagrad(plans[0], a, _hyper(hp, B, step_form=4), la)


def test_twin_steps_replayed_from_a_hipgraph_stay_readable(hip):
    """A captured burst of twin-form steps is replayed without any Python call per step, so the host cannot know which
    copies are current: once such a step has been issued every reader of R / br canonicalises first — reading the
    tables between replays, and stepping on from there, gives exactly what eager three-launch steps give."""
    from trainer.hip_api import DeviceTables
    B, V, d, cap = 9000, 20000, 64, 8
    t = oracle_tables(V, d, "Adagrad")
    a, b = tables_from_oracle(t, DeviceTables), tables_from_oracle(t, DeviceTables)
    b.enable_twin()
    hp = ref.Hyper(learning_rate=0.05)
    plans = [hip.build_plan(*to_dev(*make_batch(40 + k, B, V)), V, chunk_cap=cap).compact(hip.lib) for k in range(3)]
    assert plans[0].r_crec is not None
    h3, h4 = _hyper(hp, B, step_form=3), _hyper(hp, B, step_form=4)
    ws = hip.step_workspace(plans[0], b.d)
    ws = torch.empty(max(hip.lib.glove_step_workspace_bytes(B, p.cap_chunks, b.d) for p in plans), dtype=torch.uint8, device="cuda:0")
    lb = torch.zeros(4, device="cuda:0")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        hip.step_adagrad(plans[0], b, h4, lb, ws)            # warm the launch path outside the capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    hip.step_adagrad(plans[0], a, h3)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for p in plans:
            hip.step_adagrad(p, b, h4, lb, ws)
    for rnd in range(3):
        g.replay()
        for p in plans:
            hip.step_adagrad(p, a, h3)
        _assert_same_bits(a, b, "after replay %d" % rnd)       # reads R, br of the twinned tables: canonicalises
        assert int(b.R_ver.sum()) == 0


@pytest.mark.parametrize("B,V,d,cap", [(1024, 300, 64, 32), (6000, 80, 300, 4), (4096, 64, 128, 2), (3000, 150, 50, 16),
                                       (2000, 5000, 20, 8), (9000, 40000, 64, 16)])
def test_touched_rows_exchange_equals_dense_and_sparse_bitwise(hip, B, V, d, cap):
    """The packed-list exchange on one rank (pack -> combine -> apply) == dense gradient + dense apply == sparse
    apply, bit for bit: all three sum the same partials in the same order; the dense buffer behind the lists is
    never zeroed and the marks come back all zero."""
    from trainer.hip_api import DeviceTables
    from trainer.stepper import HipBackend, Stepper
    row, col, w, y = make_batch(77, B, V)
    t = oracle_tables(V, d, "Adagrad")
    tabs = [tables_from_oracle(t, DeviceTables) for _ in range(3)]
    backend = HipBackend("cuda:0")
    plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap, compact=True)
    kw = dict(l2_reg=0.01, reg_mult=2.0, learning_rate=0.05, step_form=1)
    steppers = [Stepper(backend, tabs[0], kw, B, exchange="rows"), Stepper(backend, tabs[1], kw, B, exchange="dense"),
                Stepper(backend, tabs[2], kw, B, exchange="dense")]
    steppers[0].prepare([plan])
    steppers[1].dense, steppers[1].G = True, backend.dense_grad_buffer(tabs[1])        # the data-parallel form on one rank
    assert steppers[0].rows and [n for n, _ in steppers[0].phases()] == ["passes", "pack_grad", "all_gather", "combine_apply"]
    assert [n for n, _ in steppers[1].phases()] == ["passes", "dense_grad", "dense_apply"] and not steppers[2].dense
    steppers[0].G.fill_(float("nan"))                # whatever the buffer holds: first touches store, they do not add
    for _ in range(3):
        for st in steppers:
            st.step(plan)
    _assert_same_bits(tabs[1], tabs[2])
    if plan.r_crec is None or plan.host_counts[6] == 1:
        _assert_same_bits(tabs[0], tabs[1])
        assert torch.equal(steppers[0].loss_out[:3], steppers[1].loss_out[:3])
    else:
        # with chunk records the stepper's passes write the entries of the ids they sum completely themselves
        # (glove_passes_packing_f32): ids of several chunks are then summed pair by pair instead of chunk by chunk
        for n in ("R", "C", "br", "bc"):
            torch.testing.assert_close(getattr(tabs[0], n), getattr(tabs[1], n), rtol=2e-5, atol=2e-6)
            torch.testing.assert_close(tabs[0].s1[n], tabs[1].s1[n], rtol=2e-5, atol=2e-6)
        torch.testing.assert_close(steppers[0].loss_out[:3], steppers[1].loss_out[:3], rtol=1e-6, atol=0)
    np.testing.assert_allclose(steppers[0].loss_out.cpu().numpy(), steppers[2].loss_out.cpu().numpy(), rtol=1e-6)
    assert int(steppers[0].bufs["mark"].abs().max()) == 0
    n_r, n_c = plan.host_counts[1], plan.host_counts[3]
    head = steppers[0].bufs["send"][0, :2].view(torch.int32).tolist()
    assert head == [n_r, n_c]
    ids = steppers[0].bufs["send"][1:1 + n_r + n_c, tabs[0].d + 1].view(torch.int32).cpu().numpy()
    np.testing.assert_array_equal(ids, np.r_[np.unique(row), np.unique(col)])


@pytest.mark.parametrize("B,V,d,cap,sides", [(3000, 101, 64, 16, 3), (3000, 101, 64, 16, 2), (20000, 5000, 128, 4, 3),
                                             (9000, 700, 300, 8, 2), (9000, 700, 300, 8, 1), (6000, 50, 52, 32, 3),
                                             (40000, 30000, 128, 16, 3), (200000, 3000, 64, 16, 3),
                                             (600000, 20000, 64, 8, 3), (600000, 20000, 128, 8, 2)])   # several chunks per lane group
def test_packing_passes_write_the_same_list(hip, B, V, d, cap, sides):
    """glove_passes_packing_f32 + glove_pack_rest_f32 == glove_passes_f32 + glove_pack_grad_f32: same header, same ids
    in the same places; gradient rows of single-chunk ids bit for bit, the others (summed pair by pair instead of chunk by
    chunk) within fp32 rounding; heavy ids and ids split between lane groups come from pack_rest.  Without chunk
    records the pair IS the second pair of calls."""
    import os
    from trainer.hip_api import DeviceTables, make_hyper
    row, col, w, y = make_batch(B + d, B, V)
    t = tables_from_oracle(oracle_tables(V, d, "Adagrad"), DeviceTables)
    h = make_hyper(learning_rate=0.05, batch_size=B, l2_reg=0.01, reg_mult=2.0)
    h.sides = sides
    raw = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap)
    for records in (True, False):
        plan = raw.compact(hip.lib, t.d, records=records)
        assert (plan.r_crec is not None) == records
        n = 1 + plan.host_counts[1] + plan.host_counts[3]
        a = torch.full((n, t.d + 4), float("nan"), device="cuda:0")
        b = torch.full((n, t.d + 4), float("nan"), device="cuda:0")
        if sides == 2:      # the sharded forms: col pass, row side's step (folds the loss partials), then the pack
            hr = make_hyper(learning_rate=0.0, batch_size=B, l2_reg=0.01, reg_mult=2.0)     # lr 0: the tables stay as they are
            hr.sides = 1
            hip.passes_packing(plan, t, h, a)
            hip.rowside_step(plan, t, hr)
            hip.pack_rest(plan, t, h, a)
            hip.colpass(plan, t, h)
            hip.rowside_step(plan, t, hr)
            hip.pack_grad(plan, t, h, b)
        else:
            hip.passes_packing(plan, t, h, a)
            hip.pack_rest(plan, t, h, a)
            hip.passes(plan, t, h)
            hip.pack_grad(plan, t, h, b)
        nr = plan.host_counts[1] if sides & 1 else 0
        nc = plan.host_counts[3] if sides & 2 else 0
        a, b = a[:1 + nr + nc].cpu(), b[:1 + nr + nc].cpu()
        assert not torch.isnan(a[1:]).any() and not torch.isnan(b[1:]).any()        # every entry was written (the header: 8 floats)
        assert torch.equal(a[0, :2].view(torch.int32), b[0, :2].view(torch.int32))
        torch.testing.assert_close(a[0, 2:6], b[0, 2:6], rtol=2e-6, atol=0)
        assert torch.equal(a[1:, t.d + 1:], b[1:, t.d + 1:])                       # id, side, 0
        if records == "0":
            assert torch.equal(a[1:], b[1:])
            continue
        chunks = np.r_[plan.r_uniq_rec.cpu().numpy().reshape(-1, 4)[:nr, 2], plan.c_uniq_rec.cpu().numpy().reshape(-1, 4)[:nc, 2]]
        single = torch.from_numpy(chunks == 1)
        assert single.any() and (chunks > 1).any()
        assert torch.equal(a[1:, :t.d][single], b[1:, :t.d][single])         # (their bias gradients: a last-bit difference on a few)
        torch.testing.assert_close(a[1:], b[1:], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("count_first", [False, True])
@pytest.mark.parametrize("B,V,d,W", [(3000, 101, 64, 2), (2000, 5000, 50, 3), (5000, 403, 128, 4), (900, 31, 300, 8),
                                     (700, 300, 64, 11)])
def test_lists_of_several_ranks_sum_in_rank_order(hip, B, V, d, W, count_first):
    """W virtual ranks on one GPU, each with its own batch: their packed lists combined in rank order == the dense
    buffer they would have all-reduced (dense_grad of every plan into one G, in rank order), bit for bit, and both
    match the float64 oracle on the joint batch.  `count_first`: glove_count_packed_f32 ahead of the combines (ids that one
    list alone touches are then applied straight from their entry and skip the dense buffer) — the same bits."""
    from trainer.hip_api import DeviceTables, make_hyper
    t = oracle_tables(V, d, "Adagrad")
    a, b = tables_from_oracle(t, DeviceTables), tables_from_oracle(t, DeviceTables)
    batches = [make_batch(500 + r, B, V) for r in range(W)]
    plans = [hip.build_plan(*to_dev(*bt), V, chunk_cap=16, compact=True) for bt in batches]
    h = make_hyper(learning_rate=0.05, batch_size=W * B)
    cap = 1 + max(p.host_counts[1] + p.host_counts[3] for p in plans)
    recv = torch.zeros(W, cap, a.d + 4, device="cuda:0")
    mark = torch.zeros(2 * V, dtype=torch.int32, device="cuda:0")
    Ga, Gb = hip.dense_grad_buffer(a), hip.dense_grad_buffer(b)
    la, lb = torch.zeros(4, device="cuda:0"), torch.zeros(4, device="cuda:0")
    for _ in range(2):
        for r, p in enumerate(plans):
            hip.passes(p, a, h)
            hip.pack_grad(p, a, h, recv[r])
        lists = [hip.packed_list(recv[r]) for r in range(W)]
        if count_first:
            hip.count_packed(lists, a, Ga, mark, cap)
        for r, lst in enumerate(lists):
            hip.combine_packed(lst, r, a, Ga, mark, cap)
        tail = None
        if W > 8:           # more lists than one apply launch takes: the loss partials are handed over summed (rank order)
            tail = torch.zeros(4, device="cuda:0")
            for r in range(W):
                tail += recv[r, 0, 2:6]
        hip.apply_packed(lists, a, h, Ga, mark, tail, la, cap)
        for p in plans:
            hip.passes(p, b, h)
            hip.dense_grad(p, b, h, Gb)
        hip.dense_adagrad(b, h, Gb, lb)
        # every rank's passes advanced global_step: one step happened
        a.step.fill_(a.global_step - (W - 1)); b.step.fill_(b.global_step - (W - 1))
        ref.train_step(t, *[np.concatenate([bt[i] for bt in batches]) for i in range(4)], ref.Hyper(learning_rate=0.05))
    _assert_same_bits(a, b)
    np.testing.assert_allclose(la.cpu().numpy()[:3], lb.cpu().numpy()[:3], rtol=1e-6)
    assert int(mark.abs().max()) == 0
    assert_tables_close(a, t, 2 * PARAM_RTOL, 2 * PARAM_ATOL)


@pytest.mark.parametrize("B,V,d", [(3000, 101, 64), (2000, 5000, 52), (20000, 40000, 300)])
def test_sharded_stepper_on_one_rank_equals_the_plain_step(hip, B, V, d):
    """trainer.stepper.ShardedStepper with world = 1 (every all-to-all is a copy): fetching the batch's col rows,
    stepping on renumbered col ids (col pass, then the row side applied in place), returning the packed col gradients
    and the owner-side apply give the plain sparse step — bit for bit where the batches have no chunk records (the
    row side then runs pass + apply as the plain step does), within the fp32 rounding of the fused row side's
    pair-by-pair sums where they have."""
    from trainer.hip_api import DeviceTables, make_hyper
    from trainer.stepper import HipBackend, ShardedStepper
    t = oracle_tables(V, d, "Adagrad")
    a, b = tables_from_oracle(t, DeviceTables), tables_from_oracle(t, DeviceTables)
    backend = HipBackend("cuda:0")
    st = ShardedStepper(backend, a, dict(l2_reg=0.01, reg_mult=2.0, learning_rate=0.05), B, 1, 0, None, exercise_exchange=True)
    batches = [to_dev(*make_batch(900 + k, B, V)) for k in range(3)]
    handles = [st.add_batch(*bt, 16) for bt in batches]
    plans = [hip.build_plan(*bt, V, chunk_cap=16, compact=True) for bt in batches]
    h = make_hyper(learning_rate=0.05, batch_size=B, step_form=1)
    lb = torch.zeros(4, device="cuda:0")
    for k in (0, 1, 2, 0, 1):
        st.step(handles[k])
        hip.step_adagrad(plans[k], b, h, lb)
    if all(st.batches[h_]["plan"].r_crec is None for h_ in handles):
        _assert_same_bits(a, b)
    else:
        _assert_tables_agree(a, b, 5e-5, 5e-6)
    np.testing.assert_allclose(st.loss_out.cpu().numpy()[:3], lb.cpu().numpy()[:3], rtol=1e-5)


@pytest.mark.parametrize("B,V,d,twin", [(3000, 500, 64, False), (20000, 6000, 300, True)])
def test_sharded_stepper_alone_in_the_world_is_the_plain_step(hip, B, V, d, twin):
    """world = 1 without `exercise_exchange`: one rank owns every row, nobody else contributes to a col id, nothing waits for an
    exchange — ShardedStepper.step is glove_step_adagrad_f32 on the batch's own ids, bit for bit (also in the fused twin form)."""
    from trainer.hip_api import DeviceTables, make_hyper
    from trainer.stepper import HipBackend, ShardedStepper
    t = oracle_tables(V, d, "Adagrad")
    a, b = tables_from_oracle(t, DeviceTables), tables_from_oracle(t, DeviceTables)
    form = 4 if twin else 0
    if twin:
        a.enable_twin(); b.enable_twin()
    backend = HipBackend("cuda:0")
    backend.row_floats = a.d
    st = ShardedStepper(backend, a, dict(l2_reg=0.01, reg_mult=2.0, learning_rate=0.05, step_form=form), B, 1, 0, None)
    assert st.local_only
    batches = [to_dev(*make_batch(700 + k, B, V)) for k in range(2)]
    handles = [st.add_batch(*bt, 16) for bt in batches]
    h = make_hyper(learning_rate=0.05, batch_size=B, step_form=form)
    lb = torch.zeros(4, device="cuda:0")
    for k in (0, 1, 0):
        st.step(handles[k])
        hip.step_adagrad(hip.build_plan(*batches[k], V, chunk_cap=16, compact=True, d=b.d), b, h, lb)
    _assert_same_bits(a, b)
    assert torch.equal(st.loss_out, lb)


@pytest.mark.parametrize("workload", ["zipf_v400k_d300", "zipf_v2m_d128"])
def test_full_size_sharded_stepper_on_one_rank(hip, workload):
    """trainer.stepper.ShardedStepper (both tables sharded: what `bench.py --gpus N` times at configs 4 and 5) at the
    full size of those configs with world = 1: == the plain sparse step within the fp32 rounding of the pair-by-pair sums
    of its fused row side, loss included; and bitwise repeatable."""
    from trainer import synthetic
    from trainer.hip_api import DeviceTables, make_hyper
    from trainer.stepper import HipBackend, ShardedStepper
    B = 1048576
    wl = synthetic.make_workload(workload, seed=4, device="cuda:0", work_device="cuda:0")
    V, d = wl["V"], wl["d"]
    bt = tuple(wl[k][:B].contiguous() for k in ("row", "col", "w", "y"))
    runs = []
    for _ in range(2):
        a = DeviceTables(V, d, "Adagrad", seed=6)
        backend = HipBackend("cuda:0")
        backend.row_floats = a.d
        st = ShardedStepper(backend, a, dict(learning_rate=0.05), B, 1, 0, None, exercise_exchange=True)
        h = st.add_batch(*bt, 0)
        assert st.batches[h]["plan"].r_crec is not None           # the fused row side and the packing col pass
        st.step(h)
        st.step(h)
        runs.append((a, st.loss_out.clone()))
        del st, backend
    _assert_same_bits(runs[0][0], runs[1][0], "repeat")
    assert torch.equal(runs[0][1], runs[1][1])
    b = DeviceTables(V, d, "Adagrad", seed=6)
    plan = hip.build_plan(*bt, V, chunk_cap=0, compact=True)
    lb = torch.zeros(4, device="cuda:0")
    for _ in range(2):
        hip.step_adagrad(plan, b, make_hyper(learning_rate=0.05, batch_size=B, step_form=1), lb)
    _assert_tables_agree(runs[0][0], b, 5e-5, 5e-6)
    np.testing.assert_allclose(runs[0][1].cpu().numpy()[:3], lb.cpu().numpy()[:3], rtol=1e-5)


@pytest.mark.parametrize("B,V,d,cap", [(9000, 20000, 64, 8), (30000, 300000, 128, 16), (6000, 700, 300, 8)])
def test_step_forms_mixed_on_a_twinned_table_without_reads_in_between(hip, B, V, d, cap):
    """A twin-form step leaves rows current in the second copy.  Any step of another form behind it (one C call after
    the other, or one list through glove_steps_adagrad_f32; nothing reads R in between, so the Python side never
    canonicalises) must see those rows: the library brings a twinned table home before every form but the twin one.
    Compared bit for bit with the same sequence on plain tables (form 4 there is form 3: same arithmetic)."""
    import ctypes as C
    from trainer.hip_api import DeviceTables, GlovePlan
    t = oracle_tables(V, d, "Adagrad")
    hp = ref.Hyper(learning_rate=0.05)
    plans = [hip.build_plan(*to_dev(*make_batch(70 + k, B, V)), V, chunk_cap=cap).compact(hip.lib, records=True) for k in range(3)]
    seq = [(0, 4), (1, 1), (2, 4), (0, 3), (1, 4), (2, 2), (0, 4), (1, 4), (2, 1)]       # (plan, form)
    plain, twin = tables_from_oracle(t, DeviceTables), tables_from_oracle(t, DeviceTables)
    twin.enable_twin()
    ws = torch.empty(max(hip.lib.glove_step_workspace_bytes(B, p.cap_chunks, twin.d) for p in plans), dtype=torch.uint8, device="cuda:0")
    la, lb = torch.zeros(4, device="cuda:0"), torch.zeros(4, device="cuda:0")
    ts = twin.struct(twin_ok=True)            # the raw struct: no accessor of the Python side runs between the steps
    for k, form in seq:
        hip.step_adagrad(plans[k], plain, _hyper(hp, B, step_form=3 if form == 4 else form), la, ws)
        rc = hip.lib.glove_step_adagrad_f32(C.byref(plans[k].struct()), C.byref(ts), C.byref(_hyper(hp, B, step_form=form)),
                                            ws.data_ptr(), ws.numel(), lb.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        assert torch.equal(la, lb), (k, form)
    twin._twin_dirty = True
    _assert_same_bits(plain, twin, "forms mixed call by call")


def test_auto_step_form_straddling_the_fused_threshold_on_a_twinned_table(hip):
    """GLOVE_STEP_AUTO picks the form per plan: a stream whose batches straddle glove_fused_step_bytes() mixes the twin
    form and the two-launch form inside ONE glove_steps_adagrad_f32 call.  Same bits as the same call on plain tables
    (where auto takes the three-launch form for the big batches: the twin form's arithmetic)."""
    import ctypes as C
    from trainer import synthetic
    from trainer.hip_api import DeviceTables, GlovePlan, make_hyper
    wl = synthetic.make_workload("zipf_v400k_d300", seed=2, device="cuda:0", work_device="cuda:0")
    V, d = wl["V"], wl["d"]
    plain, twin = DeviceTables(V, d, "Adagrad", seed=5), DeviceTables(V, d, "Adagrad", seed=5)
    twin.enable_twin()

    def build(first, n):
        return hip.build_plan(*(wl[k][first:first + n].contiguous() for k in ("row", "col", "w", "y")), V,
                              chunk_cap=32).compact(hip.lib, plain.d, records=True)
    big = [build(0, 262144), build(262144, 262144)]
    small = [build(600000 + 4096 * k, 4096) for k in range(3)]
    thr = FUSED_STEP_BYTES
    assert all((p.host_counts[1] + p.host_counts[3]) * plain.d * 16 >= thr for p in big)
    assert all((p.host_counts[1] + p.host_counts[3]) * plain.d * 16 < thr for p in small)
    lists = [small[0], big[0], small[1], big[1], small[2], big[0], big[1], small[0]]
    arr = (C.POINTER(GlovePlan) * len(lists))(*[C.pointer(p.struct()) for p in lists])
    ws = torch.empty(max(hip.lib.glove_step_workspace_bytes(p.B, p.cap_chunks, plain.d) for p in lists), dtype=torch.uint8, device="cuda:0")
    h = make_hyper(learning_rate=0.05, batch_size=262144)
    losses = []
    for tables in (plain, twin):
        st = tables.struct(twin_ok=True) if tables is twin else tables.struct()
        loss = torch.zeros(4, device="cuda:0")
        for _ in range(2):
            rc = hip.lib.glove_steps_adagrad_f32(arr, len(lists), C.byref(st), C.byref(h), ws.data_ptr(), ws.numel(),
                                                 loss.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        losses.append(loss)
    assert torch.equal(losses[0], losses[1])
    twin._twin_dirty = True
    _assert_same_bits(plain, twin)


def test_more_than_eight_lists_without_a_summed_tail(hip):
    """glove_apply_packed_adagrad_f32 with more than eight lists and tail = NULL (a data-parallel rows exchange over
    more than eight ranks): the loss partials of ALL headers are summed in list order — same result as handing the
    summed tail over, bit for bit, and the scratch floats in G_flat's tail are zero again afterwards."""
    from trainer.hip_api import DeviceTables, make_hyper
    B, V, d, W = 700, 300, 64, 11
    t = oracle_tables(V, d, "Adagrad")
    a, b = tables_from_oracle(t, DeviceTables), tables_from_oracle(t, DeviceTables)
    plans = [hip.build_plan(*to_dev(*make_batch(500 + r, B, V)), V, chunk_cap=16, compact=True) for r in range(W)]
    h = make_hyper(learning_rate=0.05, batch_size=W * B)
    cap = 1 + max(p.host_counts[1] + p.host_counts[3] for p in plans)
    out = []
    for tables, with_tail in ((a, True), (b, False)):
        recv = torch.zeros(W, cap, tables.d + 4, device="cuda:0")
        mark = torch.zeros(2 * V, dtype=torch.int32, device="cuda:0")
        G, loss = hip.dense_grad_buffer(tables), torch.zeros(4, device="cuda:0")
        for r, p in enumerate(plans):
            hip.passes(p, tables, h)
            hip.pack_grad(p, tables, h, recv[r])
        lists = [hip.packed_list(recv[r]) for r in range(W)]
        for r, lst in enumerate(lists):
            hip.combine_packed(lst, r, tables, G, mark, cap)
        tail = None
        if with_tail:
            tail = torch.zeros(4, device="cuda:0")
            for r in range(W):
                tail += recv[r, 0, 2:6]
        hip.apply_packed(lists, tables, h, G, mark, tail, loss, cap)
        assert float(G[hip.grad_layout(tables)["tail"]:].abs().max()) == 0.0
        out.append(loss)
    _assert_same_bits(a, b)
    assert torch.equal(out[0], out[1])


EDGE_CASES = [
    # (B, V, d, cap)  — degenerate shapes the reference's data can produce
    (1, 2, 4, 1), (5, 2, 4, 32), (64, 3, 4, 1), (1000, 7, 12, 1), (4096, 4096, 64, 32), (2000, 2, 64, 16),
    (513, 1000, 20, 7),
]


@pytest.mark.parametrize("B,V,d,cap", EDGE_CASES)
def test_adagrad_edge_shapes(hip, B, V, d, cap):
    from trainer.hip_api import DeviceTables
    row, col, w, y = make_batch(B + 5 * V + d, B, V, zipf=(V > 3))
    if V == 4096:                                       # every id exactly once on each side
        row = np.random.default_rng(0).permutation(V).astype(np.int32)
        col = ((row.astype(np.int64) * 7 + 1) % V).astype(np.int32)
    hp = ref.Hyper(learning_rate=0.05)
    t = oracle_tables(V, d, "Adagrad")
    dt = tables_from_oracle(t, DeviceTables)
    plan = hip.build_plan(*to_dev(row, col, w, y), V, chunk_cap=cap)
    want = ref.build_plan(row, col, cap)
    np.testing.assert_array_equal(plan.counts.cpu().numpy(), want["counts"])
    loss_out = torch.zeros(4, device="cuda:0")
    for _ in range(2):
        hip.step_adagrad(plan, dt, _hyper(hp, B), loss_out)
        loss, _, _ = ref.train_step(t, row, col, w, y, hp)
        np.testing.assert_allclose(loss_out[0].item(), loss, rtol=2e-5)
    assert_tables_close(dt, t, 2e-5, 2e-6)


@pytest.mark.parametrize("workload,B", [("text8_v50k_d300", 65536), ("zipf_v400k_d300", 1048576)])
def test_fused_step_on_a_device_refilled_plan(hip, plan_checker, workload, B):
    """A reshuffled epoch on big tables: the staging plan is refilled on the device every step (its counts are never read
    back: host_counts stay -1) and carries chunk records from the build, of which only the blocks a chunk needs are
    written (the plan is poisoned with 0xFF before each build).  The library judges such a plan by the most ids its batch
    can hold and takes the fused form — the twin form on a twinned row table; both equal the two-launch step on a
    resident plan of the same batch within the fp32 tolerance, and each other bit for bit."""
    from trainer import synthetic
    from trainer.hip_api import DeviceTables, Plan, make_hyper, staging_records
    wl = synthetic.make_workload(workload, seed=3, device="cuda:0", work_device="cuda:0")
    V, d = wl["V"], wl["d"]
    assert staging_records(B, V, V, d) is True
    cap = 32
    staging = Plan(B, V, cap, "cuda:0", records=True, links=False)      # as the trainer's runner allocates them
    ws = torch.empty(hip.lib.glove_plan_workspace_bytes(B, V), dtype=torch.uint8, device="cuda:0")
    errors = torch.zeros(8, dtype=torch.int32, device="cuda:0")
    lr = 0.05
    for k in range(2):
        row, col, w, y = (wl[n][k * B:(k + 1) * B].contiguous() for n in ("row", "col", "w", "y"))
        a = DeviceTables(V, d, "Adagrad", seed=5)
        hip.step_adagrad(hip.build_plan(row, col, w, y, V, chunk_cap=cap, compact=True), a,
                         make_hyper(learning_rate=lr, batch_size=B, step_form=1))
        _poison(staging, ws)
        hip.build_plan(row, col, w, y, V, chunk_cap=cap, into=staging, ws=ws)
        assert staging.host_counts[1] == -1 and staging.r_crec is not None
        plan_checker(staging, V, errors)
        got = []
        for twin in (True, False):
            c = DeviceTables(V, d, "Adagrad", seed=5)
            if twin:
                c.enable_twin()
            hip.step_adagrad(staging, c, make_hyper(learning_rate=lr, batch_size=B))
            if twin:
                assert int(c.R_ver.sum()) > 0, "the twin form was not taken"
            _assert_tables_agree(a, c, 2e-5, 2e-6, "batch %d twin %s" % (k, twin))
            got.append(c)
        _assert_same_bits(got[0], got[1], "twin form vs three launches, batch %d" % k)
        assert errors.tolist() == [0] * 8, errors.tolist()
        del a, got, c


@pytest.mark.parametrize("workload,B", [("text8_d64", 131072), ("text8_v50k_d300", 131072), ("zipf_v400k_d300", 1048576),
                                        ("zipf_v2m_d128", 1048576)])
def test_full_size_spot_check_against_oracle(hip, workload, B):
    """BASELINE-size batches: a full float64 oracle step is out of reach (2.4 GB of gathers), so
    (1) the sparse and the dense (data-parallel) paths must agree bit for bit, (2) rows no pair touches must
    not move, and (3) for a sample of ids the float64 restatement of exactly their pairs must match."""
    from trainer import synthetic
    from trainer.hip_api import DeviceTables, make_hyper
    wl = synthetic.make_workload(workload, seed=3, device="cuda:0", work_device="cuda:0")
    V, d = wl["V"], wl["d"]
    row, col, w, y = (wl[k][:B].contiguous() for k in ("row", "col", "w", "y"))
    a = DeviceTables(V, d, "Adagrad", seed=5)
    b = DeviceTables(V, d, "Adagrad", seed=5)
    R0, C0, br0, bc0 = a.R.clone(), a.C.clone(), a.br.clone(), a.bc.clone()
    lr, l2, m = 0.05, 0.01, 2.0
    h = make_hyper(learning_rate=lr, batch_size=B, step_form=1)
    plan = hip.build_plan(row, col, w, y, V, chunk_cap=0, compact=True)
    hip.step_adagrad(plan, a, h)
    G = hip.dense_grad_buffer(b)
    hip.rowpass(plan, b, h); hip.colpass(plan, b, h); hip.dense_grad(plan, b, h, G); hip.dense_adagrad(b, h, G)
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n                       # (1)
    del G
    prev = None
    for form in (0, 2, 3):                                   # (1b) the library's own choice and the fused forms of the step
        for n, x0 in (("R", R0), ("C", C0), ("br", br0), ("bc", bc0)):
            getattr(b, n).copy_(x0)
            b.s1[n].fill_(0.1)
        b.scalars.zero_(); b.scalars[1] = 0.1; b.step.zero_()
        hip.step_adagrad(plan, b, make_hyper(learning_rate=lr, batch_size=B, step_form=form))
        _assert_tables_agree(a, b, 2e-5, 2e-6, "step_form %d" % form)
        if form == 3 and plan.r_crec is not None:            # the two fused forms sum in the same order
            for n in ("R", "C", "br", "bc"):
                assert torch.equal(getattr(b, n), prev[n]), n
        prev = {n: getattr(b, n).clone() for n in ("R", "C", "br", "bc")} if form == 2 else prev
    if plan.r_crec is not None:
        # (1c) the form bench.py times at the HBM-bound sizes: the fused step on a TWINNED row table (form 4, and the
        # library's own choice on such a table) == the three-launch form, bit for bit, at this size
        fused_auto = (plan.host_counts[1] + plan.host_counts[3]) * a.d * 16 >= FUSED_STEP_BYTES
        for form in (4, 0):
            c = DeviceTables(V, d, "Adagrad", seed=5)
            c.enable_twin()
            hip.step_adagrad(plan, c, make_hyper(learning_rate=lr, batch_size=B, step_form=form))
            flipped = int(c.R_ver.sum())                       # raw state, before any accessor canonicalises
            if form == 4 or fused_auto:
                assert flipped > 0, "step_form %d did not take the twin form" % form
                for n in ("R", "C", "br", "bc"):
                    assert torch.equal(getattr(c, n), prev[n]), (form, n)          # prev: form 2 == form 3
            else:
                assert flipped == 0                            # below the fused regime auto stays with two launches
                for n in ("R", "C", "br", "bc"):
                    assert torch.equal(getattr(c, n), getattr(a, n)), (form, n)
            assert int(c.R_ver.sum()) == 0                     # reading R brought the table home
            del c
    # (1d) the same batch as a resident plan of the fused regime keeps it: run words + pair arrays instead of records (the
    # trainer's plans at these sizes, static and dealt): the fused forms read the same pairs in the same order, bit for bit
    plan_w = hip.build_plan(row, col, w, y, V, chunk_cap=plan.chunk_cap, compact=True, d=a.d, run_words=True)
    if plan.r_crec is not None and plan_w.r_chunk_hw is not None:
        assert plan_w.r_crec is None and plan_w.fusable
        for form in (3, 4):
            c = DeviceTables(V, d, "Adagrad", seed=5)
            if form == 4:
                c.enable_twin()
            hip.step_adagrad(plan_w, c, make_hyper(learning_rate=lr, batch_size=B, step_form=form))
            for n in ("R", "C", "br", "bc"):
                assert torch.equal(getattr(c, n), prev[n]), ("run words", form, n)
            del c
    del plan_w
    touched = torch.zeros(V, dtype=torch.bool, device="cuda:0")
    touched[row.long()] = True
    assert torch.equal(a.R[~touched], R0[~touched])                               # (2)
    assert bool((a.R[touched] != R0[touched]).any(dim=1).all())
    # (3) float64 restatement for sampled ids of BOTH sides (heaviest, singleton and strided ones)
    g = 0.0
    kappa, kappa_b = 2 * m * l2 / d / B, 2 * m * l2 / B
    for side, own_id, other_id, own0, other0, ownb0, otherb0, got, gotb in (
            ("row", row, col, R0, C0, br0, bc0, a.R, a.br), ("col", col, row, C0, R0, bc0, br0, a.C, a.bc)):
        oid = own_id.long()
        cnt = torch.bincount(oid, minlength=V)
        ids = torch.cat([cnt.argsort(descending=True)[:4], torch.nonzero(cnt == 1)[:4, 0],
                         torch.nonzero(cnt > 0)[:: max(1, int((cnt > 0).sum()) // 24), 0]]).unique()
        for u in ids.tolist():
            sel = oid == u
            partner = other0[other_id[sel].long()].double()
            own = own0[u].double()
            p = partner @ own + ownb0[u].double() + otherb0[other_id[sel].long()].double() + g
            e = 2.0 * w[sel].double() * (p - y[sel].double()) / B
            n = int(sel.sum())
            G_u = (e[:, None] * partner).sum(0) + kappa * n * own
            A = 0.1 + G_u ** 2
            want = own - lr * G_u / (A.sqrt() + 1e-7)
            np.testing.assert_allclose(got[u].cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=1e-6, err_msg="%s %d" % (side, u))
            gb = e.sum() + kappa_b * n * ownb0[u].double()
            want_b = ownb0[u].double() - lr * gb / ((0.1 + gb ** 2).sqrt() + 1e-7)
            np.testing.assert_allclose(gotb[u].item(), want_b.item(), rtol=2e-5, atol=1e-6, err_msg="%s bias %d" % (side, u))


@pytest.mark.parametrize("V,d", [(300, 64), (200, 300), (400, 128)])
def test_solid_workgroups_of_heavy_ids(hip, V, d):
    """From 131,072 chunks a side the run-merged passes add up, per workgroup, the sums of lane groups that all hold the same id
    ("solid" workgroups: the head of a Zipf batch), and the apply launch enumerates one partial row per solid workgroup (Slots).
    Forced here with one pair per chunk (chunk_cap = 1: 140,000 chunks a side, the head ids fill dozens of workgroups): every fused
    form, with chunk records and with run words, against the float64 oracle; the forms agree bit for bit among themselves."""
    from trainer.hip_api import DeviceTables
    B = 140000
    row, col, w, y = make_batch(77, B, V)
    hp = ref.Hyper(learning_rate=0.05)
    t = oracle_tables(V, d, "Adagrad")
    t0 = tables_from_oracle(t, DeviceTables)
    start = {n: getattr(t0, n).clone() for n in ("R", "C", "br", "bc")}
    loss, _, _ = ref.train_step(t, row, col, w, y, hp)
    dev = to_dev(row, col, w, y)
    plans = {"records": hip.build_plan(*dev, V, chunk_cap=1).compact(hip.lib, d=1 << 20),
             "run words": hip.build_plan(*dev, V, chunk_cap=1, compact=True, d=t0.d, run_words=True)}
    assert plans["records"].r_crec is not None and plans["records"].host_counts[0] == B
    prev = None
    for kind, plan in plans.items():
        if kind == "run words" and plan.r_chunk_hw is None:
            continue
        for form in (2, 3, 4):
            if kind == "run words" and form == 2:
                continue
            dt = DeviceTables(V, d, "Adagrad", seed=0)
            for n, x in start.items():
                getattr(dt, n).copy_(x)
            if form == 4:
                dt.enable_twin()
            loss_out = torch.zeros(4, device="cuda:0")
            hip.step_adagrad(plan, dt, _hyper(hp, B, step_form=form), loss_out)
            np.testing.assert_allclose(loss_out[0].item(), loss, rtol=5e-5, err_msg="%s form %d" % (kind, form))
            assert_tables_close(dt, t, rtol=2e-5, atol=2e-6)
            now = {n: getattr(dt, n).clone() for n in ("R", "C", "br", "bc")}
            if prev is not None:
                for n in now:
                    assert torch.equal(now[n], prev[n]), (kind, form, n)
            prev = now
