"""TEST-ONLY kernel provider: the numpy oracle behind the interface `trainer.stepper.Stepper`
expects, so the data-parallel host logic (sharding, flat-buffer all-reduce, 1/(world*B) scaling)
can be exercised with gloo on CPU.  Never imported by the product."""
import numpy as np
import torch

import glove_ref as ref


class OracleTables:
    def __init__(self, t: "ref.Tables"):
        self.t = t
        self.device = torch.device("cpu")
        self.optimizer = t.optimizer
        self.V, self.d = t.V, t.d

    @property
    def global_step(self):
        return self.t.step


class OracleBackend:
    def build_plan(self, row, col, w, y, V, chunk_cap):
        return tuple(np.asarray(a) for a in (row, col, w, y))

    def make_hyper(self, batch_size, l2_reg=0.01, reg_mult=2.0, learning_rate=0.001):
        return dict(hp=ref.Hyper(l2_reg=l2_reg, reg_mult=reg_mult, learning_rate=learning_rate),
                    inv_batch=1.0 / batch_size)

    def dense_grad_buffer(self, tables):
        return torch.zeros(2 * tables.V * tables.d + 2 * tables.V + 8, dtype=torch.float64)

    def local_dense_grad(self, plan, tables, hyper, G):
        t = tables.t
        row, col, w, y = plan
        gr = ref.gradients(t, row, col, w, y, hyper["hp"], inv_batch=hyper["inv_batch"])
        Vd = t.V * t.d
        g = G.numpy()
        g[:Vd] += gr["G_R"].ravel()
        g[Vd:2 * Vd] += gr["G_C"].ravel()
        g[2 * Vd:2 * Vd + t.V] += gr["G_br"]
        g[2 * Vd + t.V:2 * Vd + 2 * t.V] += gr["G_bc"]
        tail = g[2 * Vd + 2 * t.V:]
        tail[0] += gr["sum_e"]
        tail[1] += gr["L"] / hyper["inv_batch"]

    def apply_dense(self, tables, hyper, G, loss_out):
        t = tables.t
        Vd = t.V * t.d
        g = G.numpy()
        G_R, G_C = g[:Vd].reshape(t.V, t.d), g[Vd:2 * Vd].reshape(t.V, t.d)
        G_br, G_bc = g[2 * Vd:2 * Vd + t.V], g[2 * Vd + t.V:2 * Vd + 2 * t.V]
        tail = g[2 * Vd + 2 * t.V:]
        hp = hyper["hp"]
        gr = dict(G_R=G_R.copy(), G_C=G_C.copy(), G_br=G_br.copy(), G_bc=G_bc.copy(), sum_e=tail[0],
                  dg_reg=2.0 * hp.reg_mult * hp.l2_reg * t.g,
                  touched_r=(G_R != 0).any(1) | (G_br != 0), touched_c=(G_C != 0).any(1) | (G_bc != 0))
        loss_out[1] = tail[1] * hyper["inv_batch"]
        ref.apply_update(t, gr, hp)
        G.zero_()

    def step_sparse_adagrad(self, plan, tables, hyper, loss_out):
        row, col, w, y = plan
        loss, L, reg = ref.train_step(tables.t, row, col, w, y, hyper["hp"])
        loss_out[0], loss_out[1], loss_out[2] = loss, L, reg
