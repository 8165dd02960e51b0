"""Shared test helpers: synthetic batches and oracle <-> device table transfer."""
import numpy as np
import torch

import glove_ref as ref


def zipf_ids(rng, n, V, s=1.0):
    p = 1.0 / np.arange(1, V + 1) ** s
    p /= p.sum()
    return rng.choice(V, size=n, p=p).astype(np.int32)


def make_batch(seed, B, V, zipf=True):
    """(row, col, weight, value) like the reference's input_fn delivers after the vocab lookup
    (reference data_utils.py:4-26, estimator.py:26-28); row != col as text8.py:92 guarantees."""
    rng = np.random.default_rng(seed)
    if zipf:
        row, col = zipf_ids(rng, B, V), zipf_ids(rng, B, V)
    else:
        row = rng.integers(0, V, B).astype(np.int32)
        col = rng.integers(0, V, B).astype(np.int32)
    clash = row == col
    col[clash] = (col[clash] + 1) % V
    count = 10 + np.floor(rng.pareto(1.2, B)).clip(0, 1e5)
    w = ref.glove_weight(count).astype(np.float32)
    y = np.log(count * rng.uniform(0.35, 0.6, B)).astype(np.float32)
    return row, col, w, y


def to_dev(*arrays, device="cuda:0"):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in arrays]


def tables_from_oracle(t: "ref.Tables", DeviceTables, device="cuda:0"):
    """Device tables holding exactly the (fp32-rounded) oracle state."""
    dt = DeviceTables(t.V, t.d, t.optimizer, device=device, seed=0)
    f = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(device)
    dm = dt.d_model                    # device rows are padded to a multiple of 4 floats; the padding stays zero

    def put(dst, a):
        (dst[:, :dm] if dst.dim() == 2 else dst).copy_(f(a))
    for n in ("R", "C", "br", "bc"):
        put(getattr(dt, n), getattr(t, n))
        if t.optimizer == "Adagrad":
            put(dt.s1[n], getattr(t, "A_" + n))
        else:
            put(dt.s1[n], getattr(t, "M_" + n))
            put(dt.s2[n], getattr(t, "V_" + n))
    sc = np.zeros(8, np.float32)
    sc[0] = t.g
    if t.optimizer == "Adagrad":
        sc[1] = t.A_g
    else:
        sc[1], sc[2] = t.M_g, t.V_g
    dt.scalars.copy_(torch.from_numpy(sc))
    dt.step.fill_(t.step)
    return dt


def oracle_tables(V, d, optimizer, seed=1):
    """float64 oracle tables whose values are exactly representable in fp32, so the device
    starts from bit-identical parameters."""
    t = ref.Tables(V, d, optimizer, dtype=np.float32, seed=seed)
    return t.astype(np.float64)


def assert_tables_close(dt, t, rtol=1e-5, atol=1e-6):
    dm = dt.d_model
    got = lambda x: (x[:, :dm] if x.dim() == 2 else x).cpu().numpy()
    for n in ("R", "C", "br", "bc"):
        np.testing.assert_allclose(got(getattr(dt, n)), getattr(t, n), rtol=rtol, atol=atol, err_msg=n)
        slot = "A_" if t.optimizer == "Adagrad" else "M_"
        np.testing.assert_allclose(got(dt.s1[n]), getattr(t, slot + n), rtol=rtol, atol=atol, err_msg=slot + n)
        if t.optimizer == "Adam":
            np.testing.assert_allclose(got(dt.s2[n]), getattr(t, "V_" + n), rtol=rtol, atol=1e-9, err_msg="V_" + n)
        if getattr(dt, n).dim() == 2 and dt.d > dm:      # alignment padding must stay exactly zero
            assert float(getattr(dt, n)[:, dm:].abs().max()) == 0.0, n + " padding moved"
    np.testing.assert_allclose(dt.scalars[0].item(), t.g, rtol=rtol, atol=atol, err_msg="global_bias")
    assert dt.global_step == t.step


def free_port() -> int:
    """A TCP port nobody listens on right now (for a torch.distributed rendezvous on 127.0.0.1): fixed numbers derived from the
    pid fell into the ephemeral range and collided, once in a while, with a connection of an earlier test."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]



# ---- dedup index ("plan") against oracle/glove_ref.py:build_plan ------------------------------------------------------
PLAN_ARRAYS = ("r_partner", "r_w", "r_y", "r_to_c", "r_chunk_id", "r_chunk_start", "r_uniq_slot", "r_uniq_rec", "c_partner",
               "c_perm", "c_w", "c_y", "c_chunk_id", "c_chunk_start", "c_uniq_slot", "c_uniq_rec", "heavy", "counts")


def _poison(plan, ws):
    """0xFF into every array of a plan and into the build workspace: -1 as an id or position, NaN as a float — whatever a
    build leaves unwritten, or expects zeroed from allocation time, shows."""
    for n in PLAN_ARRAYS + ("r_crec", "c_crec", "r_mark", "c_mark", "r_chunk_hw", "c_chunk_hw"):
        t = getattr(plan, n)
        if t is not None:
            t.view(torch.uint8).fill_(0xFF)
    ws.fill_(0xFF)


def _assert_plan_equals_oracle(plan, want, B, w, y):
    counts = plan.counts.cpu().numpy()
    np.testing.assert_array_equal(counts, want["counts"])
    nc_r, nu_r, nc_c, nu_c, n_heavy = counts[:5]
    np.testing.assert_array_equal(np.sort(plan.heavy.cpu().numpy()[:n_heavy]), want["heavy"])
    for name, exp, n in (("r_partner", want["r_partner"], B), ("c_partner", want["c_partner"], B), ("c_perm", want["c_perm"], B),
                         ("r_to_c", want["r_to_c"], B), ("r_chunk_id", want["r_chunk_id"], nc_r),
                         ("r_chunk_start", want["r_chunk_start"], nc_r + 1), ("r_uniq_slot", want["r_uniq_slot"], nu_r + 1),
                         ("c_chunk_id", want["c_chunk_id"], nc_c), ("c_chunk_start", want["c_chunk_start"], nc_c + 1),
                         ("c_uniq_slot", want["c_uniq_slot"], nu_c + 1)):
        if getattr(plan, name) is not None:                     # (c_perm / r_to_c are optional: Plan(links=False))
            np.testing.assert_array_equal(getattr(plan, name).cpu().numpy()[:n], exp, err_msg=name)
    for side in ("r", "c"):                                     # the id bitmaps (small batches): exactly the side's distinct ids
        mark = getattr(plan, side + "_mark", None)
        if mark is not None:
            bits = np.unpackbits(mark.cpu().numpy().view(np.uint8), bitorder="little")
            np.testing.assert_array_equal(np.flatnonzero(bits), np.unique(want[side + "_uniq_rec"][:, 0]), err_msg=side + "_mark")
    np.testing.assert_array_equal(plan.r_uniq_rec.cpu().numpy()[:4 * nu_r].reshape(-1, 4), want["r_uniq_rec"])
    np.testing.assert_array_equal(plan.c_uniq_rec.cpu().numpy()[:4 * nu_c].reshape(-1, 4), want["c_uniq_rec"])
    if plan.r_w is not None:                                    # (a plan of a dealt epoch may keep its pair fields in its records only)
        np.testing.assert_array_equal(plan.r_w.cpu().numpy()[:B], w[want["perm_r"]])
        np.testing.assert_array_equal(plan.c_y.cpu().numpy()[:B], y[want["perm_r"]][want["c_perm"]])
    for side, nc in (("r", nc_r), ("c", nc_c)):                 # the run words (glove_plan.r_chunk_hw): word 3 of a record header on its own
        hw = getattr(plan, side + "_chunk_hw", None)
        if hw is not None:
            ids = np.asarray(want[side + "_chunk_id"])
            first = np.r_[True, ids[1:] != ids[:-1]]
            run_id = np.cumsum(first) - 1
            run_end = np.r_[np.flatnonzero(first)[1:], nc] - 1
            np.testing.assert_array_equal(hw.cpu().numpy()[:nc].view(np.uint32), (run_end[run_id] - np.arange(nc)).astype(np.uint32) | (first.astype(np.uint32) << 31),
                                          err_msg=side + "_chunk_hw")
    if plan.r_crec is None:
        return
    # per-chunk records: every header, every pair slot, and the padding of the blocks a reader touches (weight 0, valid id)
    capP = (plan.chunk_cap + 7) // 8 * 8
    rd = 4 + 3 * capP
    wr, yr = w[want["perm_r"]], y[want["perm_r"]]
    for side, nc, partner, ww, yy in (("r", nc_r, want["r_partner"], wr, yr), ("c", nc_c, want["c_partner"], wr[want["c_perm"]], yr[want["c_perm"]])):
        ids, starts = np.asarray(want[side + "_chunk_id"]), np.asarray(want[side + "_chunk_start"])
        rec = plan.records(side, nc).cpu().numpy()
        n = np.diff(starts)
        first = np.r_[True, ids[1:] != ids[:-1]]
        run_id = np.cumsum(first) - 1
        run_end = np.r_[np.flatnonzero(first)[1:], nc] - 1
        np.testing.assert_array_equal(rec[:, 0], ids, err_msg=side + " record id")
        np.testing.assert_array_equal(rec[:, 1], n, err_msg=side + " record pairs")
        np.testing.assert_array_equal(rec[:, 2], run_id, err_msg=side + " record id position")
        np.testing.assert_array_equal(rec[:, 3].view(np.uint32), (run_end[run_id] - np.arange(nc)).astype(np.uint32) | (first.astype(np.uint32) << 31),
                                      err_msg=side + " record chunks-behind word")
        blocks = rec[:, 4:].reshape(nc, capP // 8, 3, 8)
        slot = np.arange(capP)[None, :]
        used, padding = slot < n[:, None], (slot >= n[:, None]) & (slot < ((n + 7) // 8 * 8)[:, None])
        fields = [blocks[:, :, f, :].reshape(nc, capP) for f in range(3)]
        np.testing.assert_array_equal(fields[0][used], partner, err_msg=side + " record partners")
        np.testing.assert_array_equal(fields[1][used].view(np.float32), ww, err_msg=side + " record weights")
        np.testing.assert_array_equal(fields[2][used].view(np.float32), yy, err_msg=side + " record values")
        assert (fields[1][padding].view(np.float32) == 0).all() and ((fields[0][padding] >= 0) & (fields[0][padding] < plan.V)).all()


