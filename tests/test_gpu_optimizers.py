"""The Keras optimizers beyond Adagrad / Adam that `tf.keras.optimizers.get(name)` resolves (reference
src/models/train_utils.py:13-16 hands over the name and the learning rate; everything else keeps its Keras-legacy default):
SGD (plain, momentum, Nesterov), RMSprop, Adamax, Adadelta, Ftrl, Nadam (all eight names of Keras 2.11 with Adagrad and Adam) through glove_step_sparse_f32 against oracle/glove_ref.py (float64).
Tolerances as for the other optimizers: loss rtol 1e-5, parameters and slots rtol 1e-5 / atol 1e-6."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

import glove_ref as ref
from helpers import make_batch, oracle_tables, to_dev

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"

CASES = [("SGD", {}, 0.05), ("SGD", {"momentum": 0.9}, 0.02), ("SGD", {"momentum": 0.9, "nesterov": True}, 0.02),
         ("RMSprop", {}, 0.001), ("Adamax", {}, 0.002), ("Adadelta", {}, 1.0), ("Ftrl", {}, 0.05), ("Nadam", {}, 0.002)]
SLOTS = {"SGD": ("A_",), "RMSprop": ("A_",), "Adamax": ("M_", "V_"), "Adadelta": ("A_", "U_"), "Ftrl": ("A_", "Z_"), "Nadam": ("M_", "V_")}


def _device_tables(t):
    from trainer.hip_api import DeviceTables
    dt = DeviceTables(t.V, t.d, t.optimizer, device="cuda:0", seed=0)
    f = lambda a: torch.from_numpy(np.asarray(a, np.float32)).cuda()
    for n in ("R", "C", "br", "bc"):
        dst = getattr(dt, n)
        (dst[:, :dt.d_model] if dst.dim() == 2 else dst).copy_(f(getattr(t, n)))
    dt.scalars[0] = float(t.g)
    return dt


def _check(dt, t, rtol, atol):
    got = lambda x: (x[:, :dt.d_model] if x.dim() == 2 else x).cpu().numpy()
    slots = SLOTS[t.optimizer]
    for n in ("R", "C", "br", "bc"):
        np.testing.assert_allclose(got(getattr(dt, n)), getattr(t, n), rtol=rtol, atol=atol, err_msg=n)
        np.testing.assert_allclose(got(dt.s1[n]), getattr(t, slots[0] + n), rtol=rtol, atol=atol, err_msg=slots[0] + n)
        if len(slots) > 1:
            np.testing.assert_allclose(got(dt.s2[n]), getattr(t, slots[1] + n), rtol=rtol, atol=atol, err_msg=slots[1] + n)
    np.testing.assert_allclose(dt.scalars[0].item(), t.g, rtol=rtol, atol=atol)
    assert dt.global_step == t.step


@pytest.mark.parametrize("optimizer,kw,lr", CASES)
@pytest.mark.parametrize("B,V,d", [(1024, 400, 64), (3000, 60, 16), (512, 100, 300)])
def test_single_step_and_trajectory(hip, optimizer, kw, lr, B, V, d):
    """One step, then 30 on fresh batches: loss, the five variables and the optimizer's slots stay within tolerance of the
    oracle; rows no batch touched never move (RMSprop: their rms slot decays, they do not)."""
    from trainer.hip_api import make_hyper
    hp = ref.Hyper(learning_rate=lr, **kw)
    t = oracle_tables(V, d, optimizer)
    dt = _device_tables(t)
    h = make_hyper(l2_reg=hp.l2_reg, reg_mult=hp.reg_mult, learning_rate=lr, batch_size=B, optimizer=optimizer, **kw)
    G = hip.dense_grad_buffer(dt) if optimizer in ("RMSprop", "Nadam") else None
    loss_out = torch.zeros(4, device="cuda:0")
    keep = V - 1                                            # an id no batch contains
    r0 = dt.R[keep].clone()
    for s in range(31):
        row, col, w, y = make_batch(500 + s, B, V - 1)
        plan = hip.build_plan(*to_dev(row, col, w, y), V)
        hip.step_sparse(plan, dt, h, G, loss_out)
        loss, L, reg = ref.train_step(t, row, col, w, y, hp)
        np.testing.assert_allclose(loss_out.cpu().numpy()[:3], [loss, L, reg], rtol=2e-5)
        if s == 0:
            _check(dt, t, 1e-5, 1e-6)
    _check(dt, t, 5e-5, 5e-6)
    assert torch.equal(dt.R[keep], r0)
    if optimizer == "Nadam":
        np.testing.assert_allclose(dt.scalars[4 + t.step % 2].item(), t.m_cache, rtol=1e-5)     # the momentum cache, from the device
    if G is not None:
        assert float(G.abs().max()) == 0.0                  # the dense buffer is all zero again


@pytest.mark.parametrize("name,shuffle", [("sgd", "full"), ("RMSprop", "full"), ("adamax", "static"), ("SGD", "static"),
                                          ("adadelta", "full"), ("FTRL", "static"), ("nadam", "full")])
def test_cli_with_other_keras_optimizers(hip, tmp_path, name, shuffle):
    """`--optimizer` takes the Keras names case-insensitively, as tf.keras.optimizers.get does; both epoch modes train (the
    eval loss over the whole file falls) and checkpoints carry the slots."""
    from trainer import estimator
    csv, vocab = GOLDEN / "text8_cov90_ctx5_interaction.csv", GOLDEN / "text8_cov90_ctx5_vocab.txt"
    job = tmp_path / "job"
    lr = {"sgd": "0.5", "rmsprop": "0.01", "adamax": "0.02", "adadelta": "5.0", "ftrl": "0.5", "nadam": "0.02"}[name.lower()]
    estimator.main(["--train-csv", str(csv), "--vocab-txt", str(vocab), "--job-dir", str(job), "--disable-datetime-path",
                    "--embedding-size", "16", "--optimizer", name, "--learning-rate", lr, "--batch-size", "64",
                    "--train-steps", "200", "--log-every", "50", "--seed", "3", "--epoch-shuffle", shuffle,
                    "--save-checkpoints-secs", "0"])
    ev = [json.loads(l) for l in (job / "eval" / "eval_log.jsonl").read_text().splitlines()]
    assert ev[-1]["global_step"] == 200 and ev[-1]["average_loss"] < ev[0]["average_loss"]
    blob = torch.load(job / "model.ckpt-200.pt", weights_only=False)["tables"]
    assert blob["optimizer"].lower() == name.lower() and "slot1_R" in blob and ("slot2_R" in blob) == (name.lower() in ("adamax", "adadelta", "ftrl", "nadam"))


def test_unknown_optimizer_is_rejected():
    from trainer.train_utils import get_optimizer
    with pytest.raises(ValueError, match="no HIP kernel"):
        get_optimizer("Lion", learning_rate=0.1)               # not a Keras 2.11 name
