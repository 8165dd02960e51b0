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

