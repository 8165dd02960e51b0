"""TEST-ONLY kernel provider: the numpy oracle behind the interface `trainer.stepper.Stepper` /
`RowShardedStepper` expect, so the multi-rank host logic (sharding, routing, flat-buffer all-reduce,
1/(world*B) scaling, side selection) can be exercised with gloo on CPU.  Never imported by the product."""
import numpy as np
import torch

import glove_ref as ref


class OracleTables:
    def __init__(self, t: "ref.Tables"):
        self.t = t
        self.device = torch.device("cpu")
        self.optimizer = t.optimizer
        self.V, self.d, self.V_row = len(t.C), t.d, len(t.R)

    @property
    def global_step(self):
        return self.t.step


class OracleBackend:
    def build_plan(self, row, col, w, y, V, chunk_cap):
        return tuple(np.asarray(a) for a in (row, col, w, y))

    def make_hyper(self, batch_size, l2_reg=0.01, reg_mult=2.0, learning_rate=0.001, sides=0, head=0, neg_factor=1.0):
        return dict(hp=ref.Hyper(l2_reg=l2_reg, reg_mult=reg_mult, learning_rate=learning_rate, head=head,
                                 neg_factor=neg_factor),
                    inv_batch=1.0 / batch_size, sides=sides or 3)

    # flat layout [G_R | G_br | G_C | G_bc | tail 8], as the product's (without alignment padding)
    def _views(self, tables, G):
        t, g = tables.t, G.numpy()
        nr, nc, d = len(t.R), len(t.C), t.d
        o = np.cumsum([0, nr * d, nr, nc * d, nc])
        return (g[o[0]:o[1]].reshape(nr, d), g[o[1]:o[2]], g[o[2]:o[3]].reshape(nc, d), g[o[3]:o[4]], g[o[4]:o[4] + 8])

    def dense_grad_buffer(self, tables):
        t = tables.t
        return torch.zeros((len(t.R) + len(t.C)) * (t.d + 1) + 8, dtype=torch.float64)

    def col_half(self, tables, G):
        t = tables.t
        return G[len(t.R) * (t.d + 1):]

    def passes(self, plan, tables, hyper):
        row, col, w, y = plan
        self._gr = ref.gradients(tables.t, row, col, w, y, hyper["hp"], inv_batch=hyper["inv_batch"])

    def dense_grad(self, plan, tables, hyper, G):
        gr = self._gr
        G_R, G_br, G_C, G_bc, tail = self._views(tables, G)
        if hyper["sides"] & 1:
            G_R += gr["G_R"]; G_br += gr["G_br"]
        if hyper["sides"] & 2:
            G_C += gr["G_C"]; G_bc += gr["G_bc"]
            tail[0] += gr["sum_e"]
            tail[1] += gr["L"] / hyper["inv_batch"]

    def local_dense_grad(self, plan, tables, hyper, G):
        self.passes(plan, tables, hyper)
        self.dense_grad(plan, tables, hyper, G)

    def apply_sparse(self, plan, tables, hyper):
        t, gr, hp = tables.t, self._gr, hyper["hp"]
        assert hyper["sides"] == 1 and t.optimizer == "Adagrad"
        lr, eps = t.dtype(np.float32(hp.learning_rate)), t.dtype(np.float32(hp.epsilon))     # as ref.apply_update
        ref._adagrad(t.R, t.A_R, gr["G_R"], gr["touched_r"], lr, eps)
        ref._adagrad(t.br, t.A_br, gr["G_br"], gr["touched_r"], lr, eps)

    def apply_dense(self, tables, hyper, G, loss_out):
        t, hp, sides = tables.t, hyper["hp"], hyper["sides"]
        G_R, G_br, G_C, G_bc, tail = self._views(tables, G)
        zero_r, zero_c = np.zeros_like(t.R), np.zeros_like(t.C)
        gr = dict(G_R=G_R.copy() if sides & 1 else zero_r, G_br=G_br.copy() if sides & 1 else zero_r[:, 0] * 0,
                  G_C=G_C.copy() if sides & 2 else zero_c, G_bc=G_bc.copy() if sides & 2 else zero_c[:, 0] * 0,
                  sum_e=tail[0], dg_reg=2.0 * hp.reg_mult * hp.l2_reg * t.g)
        gr["touched_r"] = (gr["G_R"] != 0).any(1) | (gr["G_br"] != 0)
        gr["touched_c"] = (gr["G_C"] != 0).any(1) | (gr["G_bc"] != 0)
        assert sides & 2, "the scalar work goes with the col side"
        loss_out[1] = tail[1] * hyper["inv_batch"]
        if t.optimizer == "Adam" and not sides & 1:
            raise AssertionError("Adam has no side-restricted form")
        ref.apply_update(t, gr, hp)
        G.zero_()

    def step_sparse_adagrad(self, plan, tables, hyper, loss_out):
        row, col, w, y = plan
        loss, L, reg = ref.train_step(tables.t, row, col, w, y, hyper["hp"])
        loss_out[0], loss_out[1], loss_out[2] = loss, L, reg
