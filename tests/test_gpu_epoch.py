"""Epochs dealt from id-sorted master orders (glove_masters_build, glove_epoch_deal, glove_plan_build_sorted) against
oracle/glove_ref.py: build_masters / deal_epoch / build_plan.  Integer work: bit-exact.

What stands behind it in the reference: make_csv_dataset(shuffle=True, num_epochs=None) reshuffles the file every epoch
(src/models/data_utils.py:12-21) and Keras' OptimizerV2 dedups every step's sparse gradients (train_utils.py:13-16).
"""
import numpy as np
import pytest
import torch

import glove_ref as ref
from helpers import _assert_plan_equals_oracle, _poison, make_batch, to_dev

pytestmark = pytest.mark.gpu

KEY = 0x0123456789abcdef_fedcba9876543210


def _stream(seed, n, V, bad=False):
    row, col, w, y = make_batch(seed, n, V)
    if bad:
        row[::17] = V + 3                  # ids outside the vocabulary count as id 0
        col[5::29] = -2
    return row, col, w, y


def _epoch_buffers(masters):
    from trainer.hip_api import Pairs
    return Pairs(masters.n, "cuda:0"), Pairs(masters.n, "cuda:0")


@pytest.mark.parametrize("n,V,V_row,bad", [(5000, 97, 0, False), (70001, 10000, 0, True), (300000, 50000, 0, False), (40000, 3000, 1200, True)])
def test_masters_are_the_two_sorted_orders(hip, n, V, V_row, bad):
    """Row-major = sorted by (row id, col id, stream index), col-major = sorted by (col id, row id, stream index), link =
    the row-major position of every col-major pair; ids outside their table are 0 before anything is sorted."""
    row, col, w, y = _stream(n + V, n, V, bad)
    if V_row:
        row = (row % (V_row + 5)).astype(np.int32)           # some local row ids outside the shard as well
    m = hip.build_masters(*to_dev(row, col, w, y), V, V_row)
    want = ref.build_masters(row, col, V, V_row or None)
    assert m.mapped == int(((row < 0) | (row >= (V_row or V))).sum() + ((col < 0) | (col >= V)).sum())
    for pairs, perm, own, other in ((m.row_major, want["perm_r"], want["row"], want["col"]),
                                    (m.col_major, want["perm_c"], want["col"], want["row"])):
        np.testing.assert_array_equal(pairs.id.cpu().numpy()[:n], own[perm])
        np.testing.assert_array_equal(pairs.partner.cpu().numpy()[:n], other[perm])
        np.testing.assert_array_equal(pairs.w.cpu().numpy()[:n], w[perm])
        np.testing.assert_array_equal(pairs.y.cpu().numpy()[:n], y[perm])
    np.testing.assert_array_equal(m.link.cpu().numpy()[:n], want["link"])


@pytest.mark.parametrize("n,V,B", [(5000, 97, 128), (70001, 10000, 1024), (300000, 50000, 100), (300000, 50000, 131072),
                                   (9000, 300, 9000), (9000, 300, 5000)])
def test_deal_is_bit_exact(hip, n, V, B):
    """One epoch: both orders partitioned by the batch number of the keyed bijection — one counting-sort pass up to 2,048
    batches (n = 300,000 at B = 100: two passes) — pair for pair what oracle.deal_epoch gives; a second key gives another
    epoch from the same masters."""
    row, col, w, y = _stream(n + B, n, V)
    m = hip.build_masters(*to_dev(row, col, w, y), V)
    want_m = ref.build_masters(row, col, V)
    rs, cs = _epoch_buffers(m)
    ws = hip.deal_workspace(n, B, "cuda:0")
    for key in (KEY, KEY ^ 0x5555_0000_ffff_1234_5678):
        ws.fill_(0xFF)
        for p in (rs, cs):
            for t in (p.id, p.partner, p.w, p.y):
                t.view(torch.uint8).fill_(0xFF)
        hip.deal_epoch(m, B, key, rs, cs, ws)
        ir, ic = ref.deal_epoch(want_m, B, key)
        np.testing.assert_array_equal(rs.id.cpu().numpy()[:n], want_m["row"][ir])
        np.testing.assert_array_equal(rs.partner.cpu().numpy()[:n], want_m["col"][ir])
        np.testing.assert_array_equal(rs.w.cpu().numpy()[:n], w[ir])
        np.testing.assert_array_equal(rs.y.cpu().numpy()[:n], y[ir])
        np.testing.assert_array_equal(cs.id.cpu().numpy()[:n], want_m["col"][ic])
        np.testing.assert_array_equal(cs.partner.cpu().numpy()[:n], want_m["row"][ic])
        np.testing.assert_array_equal(cs.w.cpu().numpy()[:n], w[ic])
        np.testing.assert_array_equal(cs.y.cpu().numpy()[:n], y[ic])


@pytest.mark.parametrize("n,V,B,cap,records", [(20000, 300, 1024, 16, True), (20000, 300, 1024, 7, False), (150000, 10000, 16384, 16, True),
                                               (150000, 10000, 16384, 32, False), (600000, 200000, 131072, 32, True),
                                               (30000, 12000, 4096, 16, True)])
def test_sorted_build_equals_the_oracle_index(hip, plan_checker, n, V, B, cap, records):
    """glove_plan_build_sorted on every batch of a dealt epoch, into poisoned staging plans (with chunk records and no pair
    arrays of their own, or with pair arrays and no records): the index oracle.build_plan gives for the batch handed over
    in row-major order — which is what glove_plan_build gives for it, too (checked on one batch) — and the device-side range
    check of every slot a step kernel may read finds nothing."""
    from trainer.hip_api import PlanBlock
    row, col, w, y = _stream(n + cap, n, V)
    m = hip.build_masters(*to_dev(row, col, w, y), V)
    rs, cs = _epoch_buffers(m)
    hip.deal_epoch(m, B, KEY, rs, cs, hip.deal_workspace(n, B, "cuda:0"))
    nb = n // B
    block = PlanBlock([hip.staging_plan(B, V, cap, "cuda:0", records=records, run_words=not records) for _ in range(nb)])
    assert (block.plans[0].r_chunk_hw is not None) == (not records)        # (arrays-only plans carry the run words the fused forms need)
    ws = torch.empty(hip.lib.glove_plan_sorted_workspace_bytes(B, nb), dtype=torch.uint8, device="cuda:0")
    errors = torch.zeros(8, dtype=torch.int32, device="cuda:0")
    for first, count in ((0, nb), (nb // 2, nb - nb // 2)):          # the whole epoch, then a run from the middle into plans 0 ..
        for p in block.plans:
            _poison(p, ws)
        hip.build_plans_sorted(rs, cs, first, block, count, V, ws)
        for j in range(count):
            b = first + j
            r_, c_, w_, y_ = (t.cpu().numpy() for t in rs.arrays(b * B, (b + 1) * B))
            want = ref.build_plan(r_, c_, cap, V=V)
            assert (want["perm_r"] == np.arange(B)).all()           # the batch arrives sorted by row id
            want = {k: v for k, v in want.items() if k not in ("c_perm_unused",)}
            plan = block.plans[j]
            assert plan.c_perm is None and (plan.r_partner is None) == records
            plan_checker(plan, V, errors)
            _assert_plan_equals_oracle(plan, want, B, w_, y_)
    assert errors.tolist() == [0] * 8, errors.tolist()
    # the same index from the sorting builder on the same arrival order
    r_, c_, w_, y_ = rs.arrays(0, B)
    hip.build_plans_sorted(rs, cs, 0, block, 1, V, ws)
    other = hip.build_plan(r_.contiguous(), c_.contiguous(), w_.contiguous(), y_.contiguous(), V, chunk_cap=cap, records=records, links=False)
    got, exp = block.plans[0], other
    assert got.counts.tolist() == exp.counts.tolist()
    nc_r, nu_r, nc_c, nu_c = got.counts.tolist()[:4]
    for name, k in (("r_chunk_id", nc_r), ("r_chunk_start", nc_r + 1), ("r_uniq_slot", nu_r + 1), ("c_chunk_id", nc_c),
                    ("c_chunk_start", nc_c + 1), ("c_uniq_slot", nu_c + 1)):
        assert torch.equal(getattr(got, name)[:k], getattr(exp, name)[:k]), name
    if records:
        assert torch.equal(got.records("r", nc_r)[:, :4], exp.records("r", nc_r)[:, :4])
        assert torch.equal(got.records("c", nc_c)[:, :4], exp.records("c", nc_c)[:, :4])


@pytest.mark.parametrize("V,d,B,cap,records,form", [(300, 64, 1024, 16, True, 0), (10000, 64, 16384, 16, False, 0), (3000, 300, 8192, 32, True, 3),
                                                     (3000, 128, 8192, 32, True, 4),
                                                     # arrays + run words (the staging plans of big batches) against RECORDS on the other side:
                                                     # the fused forms read the same pairs in the same order either way
                                                     (3000, 300, 8192, 32, "words", 3), (3000, 128, 8192, 32, "words", 4), (500, 64, 8192, 16, "words", 3),
                                                     (2000, 16, 4096, 4, "words", 2), (40, 1024, 3000, 7, "words", 4), (60000, 32, 65536, 2, "words", 3),
                                                     # 12 chunks per lane group of 8 lanes (the descriptors of a group arrive in two rounds)
                                                     (150000, 32, 300000, 2, "words", 3), (150000, 64, 300000, 2, "words", 4)])
def test_steps_on_dealt_batches_equal_steps_on_sorted_batches(hip, V, d, B, cap, records, form):
    """Training on the staging plans of a dealt epoch == training on plans glove_plan_build makes of the same batches in
    the same (row-major) arrival order, bit for bit, over a few steps — records only (no pair arrays), arrays only, and
    arrays + run words (a fused step without records: the descriptors and pair fields of a lane group's chunks staged in
    LDS, ranges too long for the stage read chunk by chunk) against plans WITH records."""
    words = records == "words"
    records = False if words else records
    from trainer.hip_api import DeviceTables, PlanBlock, make_hyper
    n = 5 * B + 77
    row, col, w, y = _stream(V + d, n, V)
    m = hip.build_masters(*to_dev(row, col, w, y), V)
    rs, cs = _epoch_buffers(m)
    hip.deal_epoch(m, B, KEY, rs, cs, hip.deal_workspace(n, B, "cuda:0"))
    nb = n // B
    block = PlanBlock([hip.staging_plan(B, V, cap, "cuda:0", records=records, run_words=words) for _ in range(nb)])
    ws = torch.empty(hip.lib.glove_plan_sorted_workspace_bytes(B, nb), dtype=torch.uint8, device="cuda:0")
    hip.build_plans_sorted(rs, cs, 0, block, nb, V, ws)
    assert block.plans[0].fusable == (records or words)
    a, b = DeviceTables(V, d, "Adagrad", seed=3), DeviceTables(V, d, "Adagrad", seed=3)
    if form == 4:
        a.enable_twin(); b.enable_twin()
    h = make_hyper(learning_rate=0.05, batch_size=B, step_form=form)
    la, lb = torch.zeros(4, device="cuda:0"), torch.zeros(4, device="cuda:0")
    for k in range(nb):
        hip.step_adagrad(block.plans[k], a, h, la)
        other = hip.build_plan(*(t.contiguous() for t in rs.arrays(k * B, (k + 1) * B)), V, chunk_cap=cap, records=True if words else (records or None),
                               links=False, run_words=False)
        assert (other.r_crec is not None) == bool(words or records) and other.r_chunk_hw is None
        hip.step_adagrad(other, b, h, lb)
        assert torch.equal(la, lb), (k, la.tolist(), lb.tolist())
    for name in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
        assert torch.equal(a.s1[name], b.s1[name]), name


@pytest.mark.parametrize("workload,B", [("text8_d64", 1024), ("text8_d64", 131072), ("zipf_v400k_d300", 1048576)])
def test_full_size_deal_properties(hip, workload, B):
    """At BASELINE.json's sizes (text8 shape: 1.2 M pairs; config 4's one-GPU shard: 25 M pairs), through properties that do
    not need the oracle to finish: every full batch is sorted by its id on both sides and holds the same multiset of pairs
    on both sides (per-batch checksums of (row, col, w, y)); the epoch is a permutation of the masters (checksum of
    checksums); two keys give different epochs; the deal is repeatable bit for bit."""
    from trainer import synthetic
    wl = synthetic.make_workload(workload, seed=2, device="cuda:0", work_device="cuda:0")
    V = wl["V"]
    n = wl["row"].numel()
    m = hip.build_masters(wl["row"], wl["col"], wl["w"], wl["y"], V)
    assert m.mapped == 0
    # the masters: sorted by (id, partner), and the same multiset as the stream
    def checksum(r, c, w, y):
        h = (r.long() * 1000003 + c.long()) * 998244353 + w.view(torch.int32).long() * 7919 + y.view(torch.int32).long()
        return h
    total = int(checksum(wl["row"], wl["col"], wl["w"], wl["y"]).sum().item())
    rm, cm = m.row_major, m.col_major
    kr = rm.id[:n].long() << 32 | rm.partner[:n].long()
    kc = cm.id[:n].long() << 32 | cm.partner[:n].long()
    assert bool((kr[1:] >= kr[:-1]).all()) and bool((kc[1:] >= kc[:-1]).all())
    assert int(checksum(rm.id[:n], rm.partner[:n], rm.w[:n], rm.y[:n]).sum().item()) == total
    assert int(checksum(cm.partner[:n], cm.id[:n], cm.w[:n], cm.y[:n]).sum().item()) == total
    link = m.link[:n].long()
    assert torch.equal(rm.id[:n][link], cm.partner[:n]) and torch.equal(rm.partner[:n][link], cm.id[:n])
    del kr, kc
    rs, cs = _epoch_buffers(m)
    ws = hip.deal_workspace(n, B, "cuda:0")
    nb = n // B
    seen = []
    for key in (KEY, KEY + 1, KEY):
        hip.deal_epoch(m, B, key, rs, cs, ws)
        hr = checksum(rs.id[:n], rs.partner[:n], rs.w[:n], rs.y[:n])
        hc = checksum(cs.partner[:n], cs.id[:n], cs.w[:n], cs.y[:n])
        assert int(hr.sum().item()) == total and int(hc.sum().item()) == total
        assert torch.equal(hr[:nb * B].view(nb, B).sum(1), hc[:nb * B].view(nb, B).sum(1))          # same pairs per batch on both sides
        for p in (rs, cs):
            ids = p.id[:nb * B].view(nb, B)
            assert bool((ids[:, 1:] >= ids[:, :-1]).all())
        seen.append((rs.id[:n].clone(), cs.partner[:n].clone()))
    assert not torch.equal(seen[0][0], seen[1][0])
    assert torch.equal(seen[0][0], seen[2][0]) and torch.equal(seen[0][1], seen[2][1])
