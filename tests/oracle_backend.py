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

    def make_hyper(self, batch_size, l2_reg=0.01, reg_mult=2.0, learning_rate=0.001, sides=0, head=0, neg_factor=1.0, step_form=0,
                   optimizer="Adagrad", momentum=0.0, nesterov=False, rho=None):
        return dict(hp=ref.Hyper(l2_reg=l2_reg, reg_mult=reg_mult, learning_rate=learning_rate, head=head,
                                 neg_factor=neg_factor, momentum=momentum, nesterov=nesterov, rho=rho),
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
        if getattr(tables, "_base", None) is not None:      # a col_view: the global bias lives in the real tables
            tables.t.g = tables._base.t.g
        self._gr = ref.gradients(tables.t, row, col, w, y, hyper["hp"], inv_batch=hyper["inv_batch"])
        self._inv_batch = hyper["inv_batch"]

    def colpass(self, plan, tables, hyper):
        self.passes(plan, tables, hyper)             # the oracle forms every gradient at once, from the pre-step tables

    def rowside_step(self, plan, tables, hyper):
        self.apply_sparse(plan, tables, hyper)

    def passes_packing(self, plan, tables, hyper, send):
        self.passes(plan, tables, hyper)             # the list is written in one go by pack_rest

    def pack_rest(self, plan, tables, hyper, send):
        self.pack_grad(plan, tables, hyper, send)

    tail_dtype = torch.float64

    def loss_partials(self, plan, tables, out4):
        out4.zero_()
        out4[0] = float(self._gr["sum_e"])
        out4[1] = float(self._gr["L"]) / self._inv_batch

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

    # ---- touched-rows exchange (float64 lists; ids travel as float values)
    def id_counts(self, plan):
        return len(np.unique(plan[0])), len(np.unique(plan[1]))

    def exchange_buffers(self, tables, capacity, world):
        d = tables.t.d
        return dict(send=torch.zeros(capacity, d + 4, dtype=torch.float64),
                    recv=torch.zeros(world, capacity, d + 4, dtype=torch.float64), mark=None, capacity=capacity)

    def pack_grad(self, plan, tables, hyper, send):
        gr, d, out = self._gr, tables.t.d, send.numpy()
        out[:] = 0
        k = 1
        n = [0, 0]
        for side, bit, G, Gb, touched in ((0, 1, "G_R", "G_br", "touched_r"), (1, 2, "G_C", "G_bc", "touched_c")):
            if not hyper["sides"] & bit:
                continue
            ids = np.flatnonzero(gr[touched])
            n[side] = len(ids)
            out[k:k + len(ids), :d] = gr[G][ids]
            out[k:k + len(ids), d] = gr[Gb][ids]
            out[k:k + len(ids), d + 1] = ids
            out[k:k + len(ids), d + 2] = side
            k += len(ids)
        out[0, :2] = n
        if hyper["sides"] & 2:
            out[0, 2] = gr["sum_e"]
            out[0, 3] = gr["L"] / hyper["inv_batch"]

    def _apply_lists(self, t, hp, lists, sides, tail, loss_out, inv_batch):
        """lists: [(entries [n, d+4], ids or None, fixed side or None)] in rank order: sum in that order, then apply."""
        d = t.d
        G = {0: {}, 1: {}}
        for entries, ids, side in lists:
            for i, e in enumerate(entries):
                key = int(ids[i]) if ids is not None else int(e[d + 1])
                sd = side if side is not None else int(e[d + 2])
                if key in G[sd]:
                    G[sd][key] = (G[sd][key][0] + e[:d], G[sd][key][1] + e[d])
                else:
                    G[sd][key] = (e[:d].copy(), e[d])
        if t.optimizer != "Adagrad":
            # the per-row Keras optimizers on the touched-rows exchange (both sides in one step): the lists' sums as dense
            # gradients + the union of the ranks' ids, then the optimizer as one rank applies it to the joint batch
            assert sides == 3, "side-restricted applies exist for Adagrad only"
            gr = dict(G_R=np.zeros_like(t.R), G_C=np.zeros_like(t.C), G_br=np.zeros_like(t.br), G_bc=np.zeros_like(t.bc),
                      touched_r=np.zeros(len(t.R), bool), touched_c=np.zeros(len(t.C), bool),
                      sum_e=tail[0], dg_reg=2.0 * hp.reg_mult * hp.l2_reg * t.g)
            for sd, GW, Gb, touched in ((0, "G_R", "G_br", "touched_r"), (1, "G_C", "G_bc", "touched_c")):
                for key, (g, gb) in G[sd].items():
                    gr[GW][key], gr[Gb][key], gr[touched][key] = g, gb, True
            loss_out[1] = tail[1] * inv_batch
            ref.apply_update(t, gr, hp)
            return
        lr, eps = t.dtype(np.float32(hp.learning_rate)), t.dtype(np.float32(hp.epsilon))
        for sd, W, A, b, Ab in ((0, t.R, t.A_R, t.br, t.A_br), (1, t.C, t.A_C, t.bc, t.A_bc)):
            for key, (g, gb) in G[sd].items():
                A[key] += g * g
                W[key] -= lr * g / (np.sqrt(A[key]) + eps)
                Ab[key] += gb * gb
                b[key] -= lr * gb / (np.sqrt(Ab[key]) + eps)
        if sides & 2:
            dg = tail[0] + 2.0 * hp.reg_mult * hp.l2_reg * t.g
            t.A_g = t.A_g + dg * dg
            t.g = t.g - lr * dg / (np.sqrt(t.A_g) + eps)
            loss_out[1] = tail[1] * inv_batch
            t.step += 1

    def apply_gathered(self, bufs, world, tables, hyper, G, loss_out, tail_in=None):
        recv = bufs["recv"].numpy()
        lists, tail = [], np.zeros(2)
        for r in range(world):
            n = int(recv[r, 0, 0] + recv[r, 0, 1])
            lists.append((recv[r, 1:1 + n], None, None))
            tail += recv[r, 0, 2:4]
        if tail_in is not None:                 # the loss partials were handed over apart from the lists
            tail = tail_in.numpy()[:2].copy()
        self._apply_lists(tables.t, hyper["hp"], lists, hyper["sides"], tail, loss_out, hyper["inv_batch"])

    # ---- both tables sharded
    def gather_rows(self, tables, idx, rows, biases):
        i = idx.numpy()
        rows.numpy()[:len(i)] = tables.t.C[i]
        biases.numpy()[:len(i)] = tables.t.bc[i]

    def fetch_buffers(self, tables, capacity, serve_capacity):
        d, f64 = tables.t.d, dict(dtype=torch.float64)
        return dict(C=torch.zeros(max(capacity, 1), d, **f64), bc=torch.zeros(max(capacity, 1), **f64),
                    send_rows=torch.zeros(max(serve_capacity, 1), d, **f64), send_bias=torch.zeros(max(serve_capacity, 1), **f64),
                    packed=torch.zeros(1 + capacity, d + 4, **f64), recv=torch.zeros(max(serve_capacity, 1), d + 4, **f64))

    def col_view(self, tables, bufs, capacity):
        import copy
        v = copy.copy(tables.t)                       # shares R, br and their accumulators with the real shard
        v.C, v.bc = bufs["C"].numpy(), bufs["bc"].numpy()
        view = OracleTables.__new__(OracleTables)
        view.t, view.device, view.optimizer = v, tables.device, tables.optimizer
        view.V, view.d, view.V_row = len(v.C), v.d, len(v.R)
        view._base = tables
        return view

    def owner_apply(self, tables, state, recv, ids, counts, hyper, tail, loss_out):
        r, i, lists, off = recv.numpy(), ids.numpy(), [], 0
        for n in counts:
            lists.append((r[off:off + n], i[off:off + n], 1))
            off += n
        self._apply_lists(tables.t, hyper["hp"], lists, 3, tail.numpy(), loss_out, hyper["inv_batch"])
