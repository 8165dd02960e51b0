"""TEST INFRASTRUCTURE ONLY — CPU oracle for the GloVe training step.

This module is the checker for the HIP path.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import it; nothing under `glove-tensorflow_amd/`
or `trainer/` does, and the product path fails loudly when the HIP library is missing.

PARITY UNPINNED (training step).  The reference's hot path runs inside TensorFlow 2.11 /
Keras 2.11 / tensorflow-estimator 2.11 (reference `requirements.txt:15,38,40`), which is
not present under /root/reference, is not installed here and cannot be fetched; the
reference ships no tests or golden vectors for the step.  This file is
therefore a restatement of the published algorithm written from the reference's call sites:

  * parameters / init / activity regularisers .... src/models/model_utils.py:7-15,31-39
  * forward  p = r.c + br + bc + g ................ src/models/model_utils.py:41-54
  * mean activity loss (divide by batch) .......... src/models/model_utils.py:18-21,52
  * weighted MSE head, regularisation list ........ src/models/estimator.py:48-56
  * optimizer by Keras name, lr only .............. src/models/train_utils.py:13-16
  * step counter == optimizer.iterations .......... src/models/estimator.py:45
  * eval metrics of RegressionHead ................ src/models/estimator.py:48 (tf-estimator 2.11)
  * predict: cosine over ROW embeddings + top_k ... src/models/model_utils.py:81-110, utils.py:12-19

and of the pinned third-party semantics SURVEY.md §8a marks with a warning sign
(SUM_OVER_BATCH_SIZE loss reduction, `get_losses_for` multiplicity m, OptimizerV2
duplicate-index dedup = sum first then square, Keras-legacy Adagrad
initial_accumulator_value=0.1 / epsilon=1e-7 with epsilon outside the sqrt, Keras-legacy
Adam whose sparse path decays the WHOLE table every step).

The parts of the reference that ARE importable here (`src/data/text8.py`) pin the data-side
functions at the bottom of this file through committed fixtures: see
`tests/golden/make_text8_golden.py` and `tests/test_oracle_golden.py`.

Everything computes in the dtype of the arrays handed in: float64 tables give the
high-precision oracle the fp32 HIP kernels are compared against; float32 tables give the
"CPU port" that `bench.py` times.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

# Keras-legacy defaults
ADAGRAD_INIT_ACC = 0.1
KERAS_EPSILON = 1e-7
ADAM_BETA1 = 0.9
ADAM_BETA2 = 0.999
FTRL_INIT_ACC = 0.1  # Keras-legacy Ftrl(initial_accumulator_value=0.1)
INIT_RANGE = 0.05  # Keras Embedding default initializer "uniform" = U(-0.05, 0.05)


@dataclasses.dataclass
class Hyper:
    """Hyper-parameters of one step (reference defaults: configs/app.ini:39-53)."""
    l2_reg: float = 0.01
    reg_mult: float = 2.0          # m: estimator.py:55 adds the loss list twice under keras>=2.4
    learning_rate: float = 0.001
    epsilon: float = KERAS_EPSILON
    beta1: float = ADAM_BETA1
    beta2: float = ADAM_BETA2
    # loss head: 0 = RegressionHead(weight_column) of the GloVe estimator (estimator.py:48-56);
    # 1 = MultiHead([BinaryClassHead(pos), BinaryClassHead(neg)], [1, neg_factor]) of
    # logistic_matrix_factorisation.py:50-54 (then w = positive weight, y = negative weight)
    head: int = 0
    neg_factor: float = 1.0
    # Keras-legacy defaults of the other optimizers `tf.keras.optimizers.get(name)` resolves (train_utils.py:13-16 hands over
    # the name and the learning rate only): SGD(momentum=0.0, nesterov=False), RMSprop(rho=0.9, momentum=0.0, centered=False)
    momentum: float = 0.0
    nesterov: bool = False
    rho: float | None = None       # None: the optimizer's own Keras default (RMSprop 0.9, Adadelta 0.95)


class Tables:
    """The five trainable variables + optimizer slots (model_utils.py:31-39)."""

    def __init__(self, vocab_size, dim, optimizer="Adagrad", dtype=np.float64, seed=1):
        rng = np.random.default_rng(seed)
        u = lambda *s: rng.uniform(-INIT_RANGE, INIT_RANGE, size=s).astype(dtype)
        self.V, self.d, self.dtype = int(vocab_size), int(dim), dtype
        self.optimizer = optimizer
        self.R, self.C = u(vocab_size, dim), u(vocab_size, dim)
        self.br, self.bc = u(vocab_size), u(vocab_size)
        self.g = dtype(0.0)
        self.step = 0
        names = ("R", "C", "br", "bc")
        if optimizer == "Adagrad":
            for n in names:
                setattr(self, "A_" + n, np.full_like(getattr(self, n), ADAGRAD_INIT_ACC))
            self.A_g = dtype(ADAGRAD_INIT_ACC)
        elif optimizer == "Adam":
            for n in names:
                setattr(self, "M_" + n, np.zeros_like(getattr(self, n)))
                setattr(self, "V_" + n, np.zeros_like(getattr(self, n)))
            self.M_g = dtype(0.0)
            self.V_g = dtype(0.0)
        elif optimizer in ("SGD", "RMSprop"):
            # SGD: slot "momentum" (zeros; only used with momentum > 0); RMSprop: slot "rms" (zeros)
            for n in names:
                setattr(self, "A_" + n, np.zeros_like(getattr(self, n)))
            self.A_g = dtype(0.0)
        elif optimizer == "Adamax":
            for n in names:
                setattr(self, "M_" + n, np.zeros_like(getattr(self, n)))
                setattr(self, "V_" + n, np.zeros_like(getattr(self, n)))
            self.M_g = dtype(0.0)
            self.V_g = dtype(0.0)
        elif optimizer == "Nadam":
            # slots m, v (zeros) and the optimizer's scalar weight "momentum_cache" (ones): the running product of the momentum schedule
            for n in names:
                setattr(self, "M_" + n, np.zeros_like(getattr(self, n)))
                setattr(self, "V_" + n, np.zeros_like(getattr(self, n)))
            self.M_g = dtype(0.0)
            self.V_g = dtype(0.0)
            self.m_cache = dtype(1.0)
        elif optimizer == "Adadelta":
            # slots "accum_grad" (A_) and "accum_var" (U_), zeros
            for n in names:
                setattr(self, "A_" + n, np.zeros_like(getattr(self, n)))
                setattr(self, "U_" + n, np.zeros_like(getattr(self, n)))
            self.A_g = dtype(0.0)
            self.U_g = dtype(0.0)
        elif optimizer == "Ftrl":
            # slots "accumulator" (A_, initial_accumulator_value 0.1) and "linear" (Z_, zeros)
            for n in names:
                setattr(self, "A_" + n, np.full_like(getattr(self, n), FTRL_INIT_ACC))
                setattr(self, "Z_" + n, np.zeros_like(getattr(self, n)))
            self.A_g = dtype(FTRL_INIT_ACC)
            self.Z_g = dtype(0.0)
        else:
            raise ValueError("optimizer must be Adagrad, Adam, SGD, RMSprop, Adamax, Nadam, Adadelta or Ftrl, got %r" % (optimizer,))

    def astype(self, dtype):
        out = Tables.__new__(Tables)
        out.V, out.d, out.dtype, out.optimizer, out.step = self.V, self.d, dtype, self.optimizer, self.step
        for k, v in self.__dict__.items():
            if isinstance(v, np.ndarray):
                setattr(out, k, v.astype(dtype))
            elif isinstance(v, np.floating):
                setattr(out, k, dtype(v))
        return out

    def copy(self):
        return self.astype(self.dtype)


def forward(t: Tables, row, col):
    """logits p_i = r_i . c_i + br_i + bc_i + g  (model_utils.py:41-54)."""
    r, c = t.R[row], t.C[col]
    return (r * c).sum(-1) + t.br[row] + t.bc[col] + t.g


def _softplus(x):
    """log(1 + e^x) without overflow: the form of tf.nn.sigmoid_cross_entropy_with_logits."""
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def head_loss_and_error(p, w, y, hp: Hyper, ib):
    """Per-pair loss of the head (already divided by the batch size) and e_i = d(sum of them)/d p_i.

    head 0 (estimator.py:48-51, SUM_OVER_BATCH_SIZE):  l_i = w_i (p_i - y_i)^2 / B
    head 1 (logistic_matrix_factorisation.py:48-54): both BinaryClassHeads see the same logit p, labels 1 (pos)
        and 0 (neg), weights w = features[pos_name], y = features[neg_name]; sigmoid cross-entropy
        SUM_OVER_BATCH_SIZE per head, MultiHead sums the heads with weights [1, neg_factor]:
        l_i = (w_i softplus(-p_i) + neg_factor y_i softplus(p_i)) / B
    """
    dt = p.dtype.type
    w, y = w.astype(dt), y.astype(dt)
    if hp.head == 0:
        diff = p - y
        return w * diff * diff * ib, dt(2.0) * w * diff * ib
    if hp.head != 1:
        raise ValueError("unknown head %r" % (hp.head,))
    nf = dt(hp.neg_factor)
    s = dt(1.0) / (dt(1.0) + np.exp(-p))
    return (w * _softplus(-p) + nf * y * _softplus(p)) * ib, (w * (s - dt(1.0)) + nf * y * s) * ib


def loss_terms(t: Tables, row, col, w, y, hp: Hyper, inv_batch=None):
    """(head loss L, Reg, per-pair e) with the reference's reductions.

    L   = sum_i w_i (p_i - y_i)^2 / B                     (estimator.py:48-51, SUM_OVER_BATCH_SIZE)
    Reg = l2/(d B) sum_i(|r_i|^2+|c_i|^2) + l2/B sum_i(br_i^2+bc_i^2) + l2 g^2
          (model_utils.py:8,18-21,32-38,52: activity regularisers on the gathered batch)
    `inv_batch` overrides 1/B (data-parallel shards use 1/B_global).
    """
    dt = t.dtype
    B = len(row)
    ib = dt(1.0 / B) if inv_batch is None else dt(inv_batch)
    r, c = t.R[row], t.C[col]
    p = (r * c).sum(-1) + t.br[row] + t.bc[col] + t.g
    per_pair, e = head_loss_and_error(p, w, y, hp, ib)
    L = per_pair.sum()
    lam = dt(hp.l2_reg)
    reg = lam / dt(t.d) * ib * ((r * r).sum() + (c * c).sum()) \
        + lam * ib * ((t.br[row] ** 2).sum() + (t.bc[col] ** 2).sum())
    return L, reg, e


def gradients(t: Tables, row, col, w, y, hp: Hyper, inv_batch=None):
    """Deduplicated gradients of TotalLoss = L + m*Reg  (SURVEY.md §8a a8-a9).

    Returns dict with dense [V,d]/[V] summed gradients (zero on untouched rows), the scalar
    global-bias gradient WITHOUT its own regulariser split out, per-pair e_i, and the loss.
    In a data-parallel shard the l2*g^2 term (which does not depend on the batch) is added
    once by the caller, so it is reported separately as `reg_g`/`dg_reg`.
    """
    dt = t.dtype
    B = len(row)
    ib = dt(1.0 / B) if inv_batch is None else dt(inv_batch)
    lam, m, d = dt(hp.l2_reg), dt(hp.reg_mult), dt(t.d)
    L, reg, e = loss_terms(t, row, col, w, y, hp, inv_batch)
    kappa = dt(2.0) * m * lam / d * ib
    kappa_b = dt(2.0) * m * lam * ib
    r, c = t.R[row], t.C[col]
    G_R = np.zeros_like(t.R)
    G_C = np.zeros_like(t.C)
    G_br = np.zeros_like(t.br)
    G_bc = np.zeros_like(t.bc)
    # OptimizerV2 dedup: Unique + UnsortedSegmentSum == np.add.at
    np.add.at(G_R, row, e[:, None] * c + kappa * r)
    np.add.at(G_C, col, e[:, None] * r + kappa * c)
    np.add.at(G_br, row, e + kappa_b * t.br[row])
    np.add.at(G_bc, col, e + kappa_b * t.bc[col])
    touched_r = np.zeros(len(t.R), bool)      # len(R) < V when the row table is a shard (config 5)
    touched_c = np.zeros(len(t.C), bool)
    touched_r[row] = True
    touched_c[col] = True
    return dict(G_R=G_R, G_C=G_C, G_br=G_br, G_bc=G_bc, sum_e=e.sum(), e=e,
                L=L, reg=reg, reg_g=lam * t.g * t.g, dg_reg=dt(2.0) * m * lam * t.g,
                touched_r=touched_r, touched_c=touched_c)


def _adagrad(W, A, G, touched, lr, eps):
    """Keras-legacy Adagrad, sparse: only touched rows move."""
    A[touched] += G[touched] ** 2
    W[touched] -= lr * G[touched] / (np.sqrt(A[touched]) + eps)


def _adam_dense_decay(W, M, Vv, G, lr_t, b1, b2, eps):
    """Keras-legacy Adam `_resource_apply_sparse`: m,v decay over the WHOLE variable, the
    scaled gradient is scatter-added on touched rows, and every row moves (a11)."""
    M *= b1
    M += (1 - b1) * G
    Vv *= b2
    Vv += (1 - b2) * G * G
    W -= lr_t * M / (np.sqrt(Vv) + eps)


def _sgd(W, A, G, touched, lr, mom, nesterov):
    """Keras-legacy SGD on an IndexedSlices gradient (optimizer_v2/gradient_descent.py, pinned keras 2.11, stated from knowledge
    of that source like the other third-party semantics): momentum == 0: scatter-add of -lr g on the touched rows; otherwise
    ResourceSparseApplyKerasMomentum on the deduplicated rows: accum = accum momentum - lr g; var += accum (nesterov: var +=
    accum momentum - lr g).  Untouched rows (and their accumulators) do not move."""
    if mom == 0:
        W[touched] -= lr * G[touched]
        return
    A[touched] = A[touched] * mom - lr * G[touched]
    W[touched] += (A[touched] * mom - lr * G[touched]) if nesterov else A[touched]


def _rmsprop_dense_decay(W, A, G, lr, rho, eps):
    """Keras-legacy RMSprop `_resource_apply_sparse` (optimizer_v2/rmsprop.py; momentum 0, not centered): the WHOLE rms slot
    decays by rho, (1 - rho) g^2 is scatter-added on the touched rows, and only those rows move:
    var -= lr g / (sqrt(rms) + eps).  With g = 0 elsewhere the dense form below is the same."""
    A *= rho
    A += (1 - rho) * G * G
    W -= lr * G / (np.sqrt(A) + eps)


def _adamax(W, M, Vv, G, touched, lr_t, b1, b2, eps):
    """Keras-legacy Adamax `_resource_apply_sparse` (optimizer_v2/adamax.py): lazy — m, v and var of the touched rows only:
    m = b1 m + (1 - b1) g; v = max(b2 v, |g|); var -= lr_t m / (v + eps), lr_t = lr / (1 - b1^t)."""
    M[touched] = b1 * M[touched] + (1 - b1) * G[touched]
    Vv[touched] = np.maximum(b2 * Vv[touched], np.abs(G[touched]))
    W[touched] -= lr_t * M[touched] / (Vv[touched] + eps)


NADAM_SCHEDULE_DECAY = 0.004    # Keras-legacy Nadam(schedule_decay=0.004), base 0.96


def nadam_coefficients(step_after, m_cache, lr, b1, b2, dt=np.float64):
    """Keras-legacy Nadam `_prepare_local` (optimizer_v2/nadam.py) for the step that ends with iterations == step_after:
    u_t = beta1 (1 - 0.5 0.96^(0.004 t)), u_{t+1} likewise, the momentum cache (product of all u_i) advanced by u_t;
    the learning rate is NOT decayed (the `decay` the class hands to OptimizerV2 only parameterises this schedule)."""
    t = dt(step_after)
    u_t = b1 * (dt(1.0) - dt(0.5) * dt(0.96) ** (dt(NADAM_SCHEDULE_DECAY) * t))
    u_t1 = b1 * (dt(1.0) - dt(0.5) * dt(0.96) ** (dt(NADAM_SCHEDULE_DECAY) * (t + 1)))
    sched_new = m_cache * u_t
    sched_next = sched_new * u_t1
    return dict(lr=lr, b1=b1, b2=b2, one_minus_u_t=1 - u_t, u_t1=u_t1, om_new=1 - sched_new, om_next=1 - sched_next,
                v_den=1 - b2 ** t, sched_new=sched_new)


def _nadam(W, M, Vv, G, touched, k, eps):
    """Keras-legacy Nadam `_resource_apply_sparse` (optimizer_v2/nadam.py): m and v decay over the WHOLE variable (like the
    legacy Adam), the scaled gradient is scatter-added on the touched rows, and ONLY the touched rows move:
    m_bar = (1 - u_t) g / (1 - prod u_1..t) + u_{t+1} m / (1 - prod u_1..t+1); var -= lr m_bar / (sqrt(v / (1 - beta2^t)) + eps)."""
    M *= k["b1"]
    M += (1 - k["b1"]) * G
    Vv *= k["b2"]
    Vv += (1 - k["b2"]) * G * G
    m_bar = k["one_minus_u_t"] * (G[touched] / k["om_new"]) + k["u_t1"] * (M[touched] / k["om_next"])
    W[touched] -= k["lr"] * m_bar / (np.sqrt(Vv[touched] / k["v_den"]) + eps)


def _adadelta(W, A, U, G, touched, lr, rho, eps):
    """Keras-legacy Adadelta `_resource_apply_sparse` (optimizer_v2/adadelta.py -> ResourceSparseApplyAdadelta, kernel
    SparseApplyAdadelta of tensorflow/core/kernels/training_ops.cc), touched rows only:
    accum = rho accum + (1 - rho) g^2; update = sqrt(accum_update + eps) / sqrt(accum + eps) g; var -= lr update;
    accum_update = rho accum_update + (1 - rho) update^2.  Defaults: rho 0.95, epsilon 1e-7."""
    g = G[touched]
    a = rho * A[touched] + (1 - rho) * g * g
    upd = np.sqrt(U[touched] + eps) / np.sqrt(a + eps) * g
    A[touched] = a
    W[touched] -= lr * upd
    U[touched] = rho * U[touched] + (1 - rho) * upd * upd


def _ftrl(W, A, Z, G, touched, lr):
    """Keras-legacy Ftrl `_resource_apply_sparse` at its defaults (optimizer_v2/ftrl.py: learning_rate_power -0.5, l1 = l2 =
    l2_shrinkage = beta = 0 -> ResourceSparseApplyFtrl, kernel FtrlCompute of training_ops.cc), touched rows only:
    new_accum = accum + g^2; linear += g - (sqrt(new_accum) - sqrt(accum)) / lr var;
    var = -linear / (sqrt(new_accum) / lr) where |linear| > l1 = 0, else 0; accum = new_accum."""
    g, a, w = G[touched], A[touched], W[touched]
    na = a + g * g
    z = Z[touched] + g - (np.sqrt(na) - np.sqrt(a)) / lr * w
    Z[touched] = z
    W[touched] = np.where(np.abs(z) > 0, -z / (np.sqrt(na) / lr), 0)
    A[touched] = na


def apply_update(t: Tables, gr, hp: Hyper):
    """Optimizer update of the five variables from summed gradients `gr`; step += 1."""
    dt = t.dtype
    lr, eps = dt(np.float32(hp.learning_rate)), dt(np.float32(hp.epsilon))     # cast to the variable dtype, as Keras does
    dg = gr["sum_e"] + gr["dg_reg"]
    if t.optimizer == "Adagrad":
        _adagrad(t.R, t.A_R, gr["G_R"], gr["touched_r"], lr, eps)
        _adagrad(t.C, t.A_C, gr["G_C"], gr["touched_c"], lr, eps)
        _adagrad(t.br, t.A_br, gr["G_br"], gr["touched_r"], lr, eps)
        _adagrad(t.bc, t.A_bc, gr["G_bc"], gr["touched_c"], lr, eps)
        t.A_g = t.A_g + dg * dg
        t.g = t.g - lr * dg / (np.sqrt(t.A_g) + eps)
    elif t.optimizer == "SGD":
        mom = dt(np.float32(hp.momentum))
        for n, side in (("R", "r"), ("C", "c"), ("br", "r"), ("bc", "c")):
            _sgd(getattr(t, n), getattr(t, "A_" + n), gr["G_" + n], gr["touched_" + side], lr, mom, hp.nesterov)
        if mom == 0:
            t.g = t.g - lr * dg
        else:
            t.A_g = t.A_g * mom - lr * dg
            t.g = t.g + ((t.A_g * mom - lr * dg) if hp.nesterov else t.A_g)
    elif t.optimizer == "RMSprop":
        rho = dt(np.float32(0.9 if hp.rho is None else hp.rho))
        for n in ("R", "C", "br", "bc"):
            _rmsprop_dense_decay(getattr(t, n), getattr(t, "A_" + n), gr["G_" + n], lr, rho, eps)
        t.A_g = rho * t.A_g + (1 - rho) * dg * dg
        t.g = t.g - lr * dg / (np.sqrt(t.A_g) + eps)
    elif t.optimizer == "Nadam":
        b1, b2 = dt(np.float32(hp.beta1)), dt(np.float32(hp.beta2))
        k = nadam_coefficients(t.step + 1, t.m_cache, lr, b1, b2, dt)
        for n, side in (("R", "r"), ("C", "c"), ("br", "r"), ("bc", "c")):
            _nadam(getattr(t, n), getattr(t, "M_" + n), getattr(t, "V_" + n), gr["G_" + n], gr["touched_" + side], k, eps)
        t.M_g = b1 * t.M_g + (1 - b1) * dg                     # a dense variable: `_resource_apply_dense`, same formulas
        t.V_g = b2 * t.V_g + (1 - b2) * dg * dg
        m_bar = k["one_minus_u_t"] * (dg / k["om_new"]) + k["u_t1"] * (t.M_g / k["om_next"])
        t.g = t.g - lr * m_bar / (np.sqrt(t.V_g / k["v_den"]) + eps)
        t.m_cache = dt(k["sched_new"])
    elif t.optimizer == "Adadelta":
        rho = dt(np.float32(0.95 if hp.rho is None else hp.rho))
        for n, side in (("R", "r"), ("C", "c"), ("br", "r"), ("bc", "c")):
            _adadelta(getattr(t, n), getattr(t, "A_" + n), getattr(t, "U_" + n), gr["G_" + n], gr["touched_" + side], lr, rho, eps)
        t.A_g = rho * t.A_g + (1 - rho) * dg * dg
        upd = np.sqrt(t.U_g + eps) / np.sqrt(t.A_g + eps) * dg
        t.g = t.g - lr * upd
        t.U_g = rho * t.U_g + (1 - rho) * upd * upd
    elif t.optimizer == "Ftrl":
        for n, side in (("R", "r"), ("C", "c"), ("br", "r"), ("bc", "c")):
            _ftrl(getattr(t, n), getattr(t, "A_" + n), getattr(t, "Z_" + n), gr["G_" + n], gr["touched_" + side], lr)
        na = t.A_g + dg * dg
        t.Z_g = t.Z_g + dg - (np.sqrt(na) - np.sqrt(t.A_g)) / lr * t.g
        t.g = dt(-t.Z_g / (np.sqrt(na) / lr)) if abs(t.Z_g) > 0 else dt(0.0)
        t.A_g = na
    elif t.optimizer == "Adamax":
        b1, b2 = dt(np.float32(hp.beta1)), dt(np.float32(hp.beta2))
        lr_t = dt(float(lr) / (1.0 - float(b1) ** (t.step + 1)))
        for n, side in (("R", "r"), ("C", "c"), ("br", "r"), ("bc", "c")):
            _adamax(getattr(t, n), getattr(t, "M_" + n), getattr(t, "V_" + n), gr["G_" + n], gr["touched_" + side], lr_t, b1, b2, eps)
        t.M_g = b1 * t.M_g + (1 - b1) * dg
        t.V_g = max(b2 * t.V_g, abs(dg))
        t.g = t.g - lr_t * t.M_g / (t.V_g + eps)
    else:
        # Keras casts the hyper-parameters to the variable dtype before use (OptimizerV2._get_hyper(name, var_dtype)):
        # beta_2 = float32(0.999) = 0.99900001287..., so 1 - beta_2 is 1.3e-5 (relative) away from 0.001
        b1, b2 = dt(np.float32(hp.beta1)), dt(np.float32(hp.beta2))
        tt = t.step + 1
        lr_t = dt(float(lr) * math.sqrt(1.0 - float(b2) ** tt) / (1.0 - float(b1) ** tt))
        _adam_dense_decay(t.R, t.M_R, t.V_R, gr["G_R"], lr_t, b1, b2, eps)
        _adam_dense_decay(t.C, t.M_C, t.V_C, gr["G_C"], lr_t, b1, b2, eps)
        _adam_dense_decay(t.br, t.M_br, t.V_br, gr["G_br"], lr_t, b1, b2, eps)
        _adam_dense_decay(t.bc, t.M_bc, t.V_bc, gr["G_bc"], lr_t, b1, b2, eps)
        t.M_g = b1 * t.M_g + (1 - b1) * dg
        t.V_g = b2 * t.V_g + (1 - b2) * dg * dg
        t.g = t.g - lr_t * t.M_g / (np.sqrt(t.V_g) + eps)
    t.step += 1


def train_step(t: Tables, row, col, w, y, hp: Hyper):
    """One `session.run(train_op)`.  Returns (loss, L, Reg)."""
    gr = gradients(t, row, col, w, y, hp)
    m = t.dtype(hp.reg_mult)
    reg = gr["reg"] + gr["reg_g"]
    loss = gr["L"] + m * reg
    apply_update(t, gr, hp)
    return loss, gr["L"], reg


def eval_metrics(t: Tables, row, col, w, y):
    """RegressionHead eval metrics over one pass (tf-estimator 2.11 `_eval_metric_ops`):
    average_loss = sum w l / sum w, prediction/mean, label/mean (both weighted means)."""
    p = forward(t, row, col)
    wd = w.astype(t.dtype)
    yd = y.astype(t.dtype)
    sw = wd.sum()
    return dict(average_loss=(wd * (p - yd) ** 2).sum() / sw,
                prediction_mean=(wd * p).sum() / sw, label_mean=(wd * yd).sum() / sw,
                weight_sum=sw)


def cosine_topk(R, query_ids, k):
    """get_predictions (model_utils.py:81-110) with utils.cosine_similarity (utils.py:12-19):
    l2-normalise rows (tf.math.l2_normalize: x * rsqrt(max(sum x^2, 1e-12))), matmul,
    top_k sorted descending (ties -> lower index first, as tf.math.top_k)."""
    n = R / np.sqrt(np.maximum((R * R).sum(-1, keepdims=True), 1e-12))
    sim = n[query_ids] @ n.T
    idx = np.argsort(-sim, axis=-1, kind="stable")[:, :k]
    return np.take_along_axis(sim, idx, -1), idx


# --------------------------------------------------------------------------------------
# Dedup index ("plan") — integer work, bit-exact target for glove_plan_build.
# --------------------------------------------------------------------------------------
def build_plan(row, col, chunk_cap, heavy_chunks=8, V=None):
    """Reference construction of the per-batch dedup index the HIP library builds on device.

    Row side: pairs stably sorted by row id; each run of equal ids is cut into chunks of at
    most `chunk_cap` pairs.  Col side: the pairs (in the order they arrive, like the row side) stably sorted by
    col id, `c_perm` giving the row-sorted position of each and `r_to_c` its inverse.  See DESIGN.md "Data layout".
    With `V` given, ids outside [0, V) count as id 0 — what the reference's vocabulary lookup returns for
    an unknown token (src/models/estimator.py:26-28) — and counts[5] says how many there were.
    """
    row = np.asarray(row, np.int64)
    col = np.asarray(col, np.int64)
    B = len(row)
    mapped = 0
    if V is not None:
        bad_r, bad_c = (row < 0) | (row >= V), (col < 0) | (col >= V)
        mapped = int(bad_r.sum() + bad_c.sum())
        row, col = np.where(bad_r, 0, row), np.where(bad_c, 0, col)

    def side(keys):
        starts = np.flatnonzero(np.r_[True, keys[1:] != keys[:-1]]) if B else np.zeros(0, np.int64)
        ends = np.r_[starts[1:], B] if B else starts
        chunk_id, chunk_start, uniq_slot = [], [], []
        for s, e_ in zip(starts, ends):
            uniq_slot.append(len(chunk_id))
            for cs in range(s, e_, chunk_cap):
                chunk_id.append(int(keys[s]))
                chunk_start.append(cs)
        chunk_start.append(B)
        uniq_slot.append(len(chunk_id))
        rec = [[chunk_id[a], a, b - a, chunk_start[b] - chunk_start[a]] for a, b in zip(uniq_slot[:-1], uniq_slot[1:])]
        return (np.asarray(chunk_id, np.int32), np.asarray(chunk_start, np.int32),
                np.asarray(uniq_slot, np.int32), np.asarray(rec, np.int32).reshape(-1, 4))

    perm_r = np.argsort(row, kind="stable")
    s_row, s_col = row[perm_r], col[perm_r]
    r_chunk_id, r_chunk_start, r_uniq_slot, r_uniq_rec = side(s_row)
    # the col side sorts the batch as it arrives, independently of the row side (so both sorts can share their launches
    # on the device); what links the sides is the row-sorted position of every pair
    rpos = np.empty(B, np.int64)
    rpos[perm_r] = np.arange(B)
    perm_c = rpos[np.argsort(col, kind="stable")]
    c_chunk_id, c_chunk_start, c_uniq_slot, c_uniq_rec = side(s_col[perm_c])
    r_to_c = np.empty(B, np.int64)
    r_to_c[perm_c] = np.arange(B)
    heavy = sorted([int(q) for q in np.flatnonzero(r_uniq_rec[:, 2] > heavy_chunks)] +
                   [(1 << 30) | int(q) for q in np.flatnonzero(c_uniq_rec[:, 2] > heavy_chunks)]) if B else []
    return dict(perm_r=perm_r.astype(np.int32), r_partner=s_col.astype(np.int32), r_to_c=r_to_c.astype(np.int32),
                heavy=np.asarray(heavy, np.int32),
                r_chunk_id=r_chunk_id, r_chunk_start=r_chunk_start, r_uniq_slot=r_uniq_slot,
                r_uniq_rec=r_uniq_rec, c_uniq_rec=c_uniq_rec,
                c_perm=perm_c.astype(np.int32), c_partner=s_row[perm_c].astype(np.int32),
                c_chunk_id=c_chunk_id, c_chunk_start=c_chunk_start, c_uniq_slot=c_uniq_slot,
                counts=np.asarray([len(r_chunk_id), len(r_uniq_slot) - 1,
                                   len(c_chunk_id), len(c_uniq_slot) - 1, len(heavy), mapped, 0, 0], np.int32))


# --------------------------------------------------------------------------------------
# Epochs of a resident stream (glove_masters_build / glove_epoch_deal) — integer work, bit-exact target.
# The reference reshuffles the file every epoch (src/models/data_utils.py:12-21: make_csv_dataset(shuffle=True,
# num_epochs=None)); WHICH permutation it draws is unseeded and irrelevant — what is restated here is the build's own
# keyed bijection, so that the device deal can be checked pair for pair.
# --------------------------------------------------------------------------------------
def _feistel_mix(x, k):
    """glove_common.h feistel_mix on uint32 arrays (wrap-around arithmetic)."""
    x = ((x.astype(np.uint64) + np.uint64(k)) & np.uint64(0xffffffff)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x.astype(np.uint64) * np.uint64(0x7feb352d)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    return x


def feistel_walk(x, n, key):
    """The keyed bijection of [0, n) of glove_common.h: a Feistel network over b = ceil(log2 n) bits (halves of b // 2 high and
    b - b // 2 low bits, four rounds alternately rewriting the high half from the low one and back), cycle walking.
    `key`: 128-bit int (round keys = its four 32-bit words, lowest first)."""
    b = 2
    while (1 << b) < n:
        b += 1
    hb = b // 2
    lb = b - hb
    hmask, lmask = np.uint32((1 << hb) - 1), np.uint32((1 << lb) - 1)
    ks = [(key >> (32 * r)) & 0xffffffff for r in range(4)]
    x = np.asarray(x, np.uint64).copy()
    todo = np.ones(x.shape, bool)
    while todo.any():
        v = x[todo]
        H, L = (v >> np.uint64(lb)).astype(np.uint32), (v & np.uint64(lmask)).astype(np.uint32)
        H = H ^ (_feistel_mix(L, ks[0]) & hmask)
        L = L ^ (_feistel_mix(H, ks[1]) & lmask)
        H = H ^ (_feistel_mix(L, ks[2]) & hmask)
        L = L ^ (_feistel_mix(H, ks[3]) & lmask)
        v = (H.astype(np.uint64) << np.uint64(lb)) | L.astype(np.uint64)
        x[todo] = v
        todo[todo] = v >= np.uint64(n)
    return x.astype(np.int64)


def build_masters(row, col, V, V_row=None):
    """The two master orders of a rank's nonzeros: perm_r = positions sorted by (row id, col id, stream index), perm_c by
    (col id, row id, stream index); link[q] = row-major position of the pair at col-major position q.  Ids outside their
    table count as id 0 (the unknown token, src/models/estimator.py:26-28) before anything is sorted."""
    row, col = np.asarray(row, np.int64), np.asarray(col, np.int64)
    Vr = V if not V_row else V_row
    row = np.where((row < 0) | (row >= Vr), 0, row)
    col = np.where((col < 0) | (col >= V), 0, col)
    perm_r = np.lexsort((np.arange(len(row)), col, row))
    perm_c = np.lexsort((np.arange(len(row)), row, col))
    inv_r = np.empty(len(row), np.int64)
    inv_r[perm_r] = np.arange(len(row))
    return dict(row=row, col=col, perm_r=perm_r, perm_c=perm_c, link=inv_r[perm_c])


def deal_epoch(masters, B, key):
    """One epoch: the pair at row-major position p takes seat feistel_walk(p) and belongs to batch seat // B; both orders
    are stably partitioned by batch number.  Returns the stream indices of the pairs in the row side's and the col side's
    epoch order: batch k = positions [k B, (k + 1) B) of either, sorted by row id / by col id."""
    n = len(masters["perm_r"])
    batch_r = feistel_walk(np.arange(n), n, key) // B              # by row-major position
    order_r = np.argsort(batch_r, kind="stable")
    order_c = np.argsort(batch_r[masters["link"]], kind="stable")
    return masters["perm_r"][order_r], masters["perm_c"][order_c]


# --------------------------------------------------------------------------------------
# Data-side functions (pinned by the importable reference module src/data/text8.py).
# --------------------------------------------------------------------------------------
def glove_weight(count, alpha=0.75, x_max=100):
    """src/data/text8.py:138-139."""
    return np.clip(np.power(np.asarray(count, np.float64) / x_max, alpha), 0, 1)


def cooccurrence(token_ids, context_size):
    """Right-context co-occurrence, symmetrised (src/data/text8.py:84-108).

    For every position p and offset k in 1..context_size with ids a=tok[p], b=tok[p+k], a!=b:
    count(a,b)+=1, value(a,b)+=1/k; then the union with the swapped table is summed.
    Returns (row, col, count, value) sorted by (row, col).
    """
    tok = np.asarray(token_ids, np.int64)
    n = len(tok)
    V = int(tok.max()) + 1 if n else 0
    keys, vals = [], []
    for k in range(1, context_size + 1):
        a, b = tok[:n - k], tok[k:]
        keep = a != b
        keys.append(a[keep] * V + b[keep])
        vals.append(np.full(int(keep.sum()), 1.0 / k))
    keys = np.concatenate(keys) if keys else np.zeros(0, np.int64)
    vals = np.concatenate(vals) if vals else np.zeros(0)
    uk, inv = np.unique(keys, return_inverse=True)
    cnt = np.bincount(inv, minlength=len(uk))
    val = np.bincount(inv, weights=vals, minlength=len(uk))
    r, c = uk // V, uk % V
    # union swap + sum (text8.py:103-108)
    k2 = np.concatenate([r * V + c, c * V + r])
    uk2, inv2 = np.unique(k2, return_inverse=True)
    cnt2 = np.bincount(inv2, weights=np.concatenate([cnt, cnt]), minlength=len(uk2)).astype(np.int64)
    val2 = np.bincount(inv2, weights=np.concatenate([val, val]), minlength=len(uk2))
    return uk2 // V, uk2 % V, cnt2, val2
