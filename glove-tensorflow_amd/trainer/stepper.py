"""One optimizer step = `session.run(train_op)` of the reference (estimator.py:49-56), on one
GPU or data-parallel over the GPUs of a node.

Data-parallel form (new: the reference is single-process).  Every rank holds its
own shard of the nonzero stream and a full replica of the five variables and their slots.
Per step every rank runs the forward+gradient passes over its batch of B nonzeros with
inv_batch = 1 / (world * B), adds the summed gradients into one flat dense buffer
[G_R | G_C | G_br | G_bc | tail], the buffers are summed with ONE all-reduce (RCCL over xGMI when
the backend is "nccl") and every rank applies the identical dense update.  The result equals a
single-GPU step at batch size world * B up to fp32 summation order.

`backend` is the kernel provider: `HipBackend` (below) is the only product implementation and
drives libglove_hip.so; there is no CPU implementation in the product.  Tests inject their own
provider to exercise the sharding / collective logic with gloo on CPU.
"""
from __future__ import annotations

import torch


class HipBackend:
    """Kernel provider on top of the C ABI (trainer.hip_api.GloveHip)."""

    def __init__(self, device):
        from trainer.hip_api import GloveHip
        self.hip = GloveHip(device)
        self.device = torch.device(device)
        self.row_floats = None      # floats per table row, once the tables exist: lets resident plans carry what the fused step needs

    def build_plan(self, row, col, w, y, V, chunk_cap):
        return self.hip.build_plan(row.contiguous(), col.contiguous(), w.contiguous(), y.contiguous(), V,
                                   chunk_cap=chunk_cap, compact=True, d=self.row_floats)

    def make_hyper(self, **kw):
        from trainer.hip_api import make_hyper
        return make_hyper(**kw)

    def dense_grad_buffer(self, tables):
        return self.hip.dense_grad_buffer(tables)

    def step_sparse_adagrad(self, plan, tables, hyper, loss_out):
        self.hip.step_adagrad(plan, tables, hyper, loss_out)

    def steps_sparse_adagrad(self, plans, tables, hyper, loss_out):
        self.hip.steps_adagrad(plans, tables, hyper, loss_out)

    def steps_dense_adam(self, plans, tables, hyper, G, loss_out):
        self.hip.steps_adam(plans, tables, hyper, G, loss_out)

    def local_dense_grad(self, plan, tables, hyper, G):
        self.hip.passes(plan, tables, hyper)
        self.hip.dense_grad(plan, tables, hyper, G)

    def apply_dense(self, tables, hyper, G, loss_out):
        if tables.optimizer == "Adagrad":
            self.hip.dense_adagrad(tables, hyper, G, loss_out)
        else:
            self.hip.dense_adam(tables, hyper, G, loss_out)

    # ---- pieces of the row-sharded step (hyper.sides selects the side)
    def passes(self, plan, tables, hyper):
        self.hip.passes(plan, tables, hyper)

    def apply_sparse(self, plan, tables, hyper):
        self.hip.apply_adagrad(plan, tables, hyper)

    def dense_grad(self, plan, tables, hyper, G):
        self.hip.dense_grad(plan, tables, hyper, G)

    def col_half(self, tables, G):
        """The contiguous [G_C | G_bc | tail] part of the flat buffer."""
        return G[self.hip.grad_layout(tables)["G_C"]:]

    def eval_sums(self, row, col, w, y, tables, sums):
        return self.hip.eval_sums(row.contiguous(), col.contiguous(), w.contiguous(), y.contiguous(), tables, sums)

    def eval_sums_logistic(self, row, col, pos, neg, tables, sums):
        return self.hip.eval_sums_logistic(row.contiguous(), col.contiguous(), pos.contiguous(), neg.contiguous(),
                                           tables, sums)

    def topk_cosine(self, R, query_ids, k):
        return self.hip.topk_cosine(R, query_ids, k)


class Stepper:
    def __init__(self, backend, tables, hyper_kwargs: dict, batch_size: int, world=1, dist=None):
        self.backend, self.tables, self.world, self.dist = backend, tables, int(world), dist
        if self.world > 1 and dist is None:
            raise ValueError("world > 1 needs an initialised torch.distributed module")
        self.hyper = backend.make_hyper(batch_size=batch_size * self.world, **hyper_kwargs)
        self.loss_out = torch.zeros(4, dtype=torch.float32, device=tables.device)
        # Adam (dense whole-table decay) and every multi-rank step go through the dense buffer
        self.dense = self.world > 1 or tables.optimizer != "Adagrad"
        self.G = backend.dense_grad_buffer(tables) if self.dense else None

    def step(self, plan):
        if not self.dense:
            self.backend.step_sparse_adagrad(plan, self.tables, self.hyper, self.loss_out)
            return
        self.backend.local_dense_grad(plan, self.tables, self.hyper, self.G)
        if self.world > 1:
            self.dist.all_reduce(self.G)          # sum over ranks; the tail carries the loss partials
        self.backend.apply_dense(self.tables, self.hyper, self.G, self.loss_out)

    def step_many(self, plans):
        """Several consecutive steps; on one GPU they are issued by one C call."""
        if not self.dense and hasattr(self.backend, "steps_sparse_adagrad"):
            self.backend.steps_sparse_adagrad(plans, self.tables, self.hyper, self.loss_out)
        elif (self.world == 1 and self.tables.optimizer == "Adam" and hasattr(self.backend, "steps_dense_adam")):
            self.backend.steps_dense_adam(plans, self.tables, self.hyper, self.G, self.loss_out)
        else:
            for plan in plans:
                self.step(plan)

    def read_loss(self) -> dict:
        """Host read of the last step's scalars (synchronises; call at the logging cadence only)."""
        loss, L, reg, _ = self.loss_out.tolist()
        return {"loss": loss, "weighted_mse": L, "regularization_loss": reg}


def owned_rows(V: int, world: int, rank: int) -> int:
    """Rows of the row table held by `rank` when row id u lives on rank u % world at local index u // world."""
    return (V - rank + world - 1) // world


def route_by_row_owner(coo: dict, world: int, rank: int, dist) -> dict:
    """All-to-all of a rank's nonzeros to the owners of their rows (BASELINE config 5: "row-embedding table
    sharded across 8 GPUs with all-to-all token-id routing").  `coo`: dict of equally long 1-D tensors
    row/col (int32) and w/y (float32) on the device the process group works on.  Returns the nonzeros this
    rank owns, with `row` rewritten to the LOCAL row index (row // world).  For a static stream this runs
    once at load time; a caller with fresh batches every step calls it per batch."""
    owner = (coo["row"] % world).long()
    order = torch.argsort(owner, stable=True)
    send_counts = torch.bincount(owner, minlength=world)
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts)
    s_list, r_list = send_counts.tolist(), recv_counts.tolist()
    out = {}
    for k in ("row", "col", "w", "y"):
        src = coo[k][order].contiguous()
        dst = torch.empty(int(sum(r_list)), dtype=src.dtype, device=src.device)
        dist.all_to_all_single(dst, src, output_split_sizes=r_list, input_split_sizes=s_list)
        out[k] = dst
    if not bool((out["row"] % world == rank).all()):
        raise RuntimeError("all-to-all routing delivered a nonzero to a rank that does not own its row")
    out["row"] = (out["row"] // world).to(coo["row"].dtype)
    return out


class RowShardedStepper:
    """Model-parallel form of BASELINE config 5.  The row table R / br (and their Adagrad accumulators) are
    sharded by row id % world; the col table, the global bias and their slots are replicated.  Every rank
    steps on nonzeros whose rows it owns (see route_by_row_owner), so

      * the row side is completely local: rowpass / colpass, then a sparse Adagrad apply restricted to the
        row side (hyper.sides = 1) — no communication;
      * the col side is data parallel: the rank's summed col gradients (hyper.sides = 2) go into the
        contiguous [G_C | G_bc | tail] half of the flat buffer, ONE all-reduce sums it over the ranks, and
        every rank applies the identical dense update of C, bc and the global bias.

    With inv_batch = 1 / (world * B) the result equals a single-GPU step on the union of the ranks'
    batches (tests/test_dp_gloo.py)."""

    def __init__(self, backend, tables, hyper_kwargs: dict, batch_size: int, world: int, dist):
        if tables.optimizer != "Adagrad":
            raise ValueError("the row-sharded step is implemented for Adagrad (Keras Adam has no sparse form)")
        self.backend, self.tables, self.world, self.dist = backend, tables, int(world), dist
        gb = batch_size * self.world
        self.hyper_rows = backend.make_hyper(batch_size=gb, sides=1, **hyper_kwargs)
        self.hyper_cols = backend.make_hyper(batch_size=gb, sides=2, **hyper_kwargs)
        self.loss_out = torch.zeros(4, dtype=torch.float32, device=tables.device)
        self.G = backend.dense_grad_buffer(tables)

    def step(self, plan):
        b, t = self.backend, self.tables
        b.passes(plan, t, self.hyper_cols)
        b.dense_grad(plan, t, self.hyper_cols, self.G)        # reads C (activity-L2 term): before any update
        b.apply_sparse(plan, t, self.hyper_rows)              # R, br: local, no communication
        if self.world > 1:
            self.dist.all_reduce(b.col_half(t, self.G))
        b.apply_dense(t, self.hyper_cols, self.G, self.loss_out)

    def step_many(self, plans):
        for plan in plans:
            self.step(plan)

    def read_loss(self) -> dict:
        loss, L, reg, _ = self.loss_out.tolist()
        return {"loss": loss, "weighted_mse": L, "regularization_loss": reg}


class ReshufflingRunner:
    """Single-GPU training over a stream whose pairs are re-permuted every epoch (`--epoch-shuffle full`): the
    batches are new every epoch, so their dedup index is built when they are used — like an input pipeline that
    prefetches batches, `ahead` index builds are in flight on their own streams and staging plans while earlier
    steps run.  A burst of consecutive batches [first, first + count) is captured ONCE as a hipGraph (builds,
    steps and their cross-stream dependencies) and replayed in every later epoch: the graph reads the batch
    positions of the stream's buffers, which `NonzeroStream.reshuffle_in_place` refills.
    """

    def __init__(self, hip, stream, tables, hyper, chunk_cap=0, ahead=4, burst=128):
        from trainer.hip_api import auto_chunk_cap
        self.hip, self.stream, self.tables, self.hyper = hip, stream, tables, hyper
        self.cap = chunk_cap or auto_chunk_cap(stream.B, stream.V)
        self.ahead, self.burst = max(1, int(ahead)), max(1, int(burst))
        dev, B, V = tables.device, stream.B, stream.V
        self.ring = [hip.build_plan(*stream.batch(0), V, chunk_cap=self.cap) for _ in range(self.ahead)]
        self.ring_ws = [torch.empty(hip.lib.glove_plan_workspace_bytes(B, V), dtype=torch.uint8, device=dev)
                        for _ in range(self.ahead)]
        self.ring_streams = [torch.cuda.Stream(device=dev) for _ in range(self.ahead)]
        self.step_ws = torch.empty(hip.lib.glove_step_workspace_bytes(B, B, tables.d), dtype=torch.uint8, device=dev)
        self.G = hip.dense_grad_buffer(tables) if tables.optimizer == "Adam" else None
        self.loss_out = torch.zeros(4, dtype=torch.float32, device=dev)
        self.graphs = {}
        self.position = 0                      # next batch of the current epoch
        stream.reshuffle_in_place()
        # every kernel of the sequence is launched once outside any capture (on throw-away tables of the same shape)
        from trainer.hip_api import DeviceTables
        real = self.tables
        self.tables = DeviceTables(real.V, real.d_model, real.optimizer, device=dev, seed=0, V_row=real.V_row)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._issue(0, min(self.ahead + 1, stream.batches_per_epoch))
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.tables = real
        if self.G is not None:
            self.G.zero_()

    def _step(self, plan):
        if self.G is None:
            self.hip.step_adagrad(plan, self.tables, self.hyper, self.loss_out, self.step_ws)
        else:
            self.hip.step_adam(plan, self.tables, self.hyper, self.G, self.loss_out, self.step_ws)

    def _issue(self, first, count):
        """`count` steps over batches first..first+count-1 with `ahead` index builds in flight."""
        main = torch.cuda.current_stream()
        built, stepped = [None] * count, [None] * count
        start = torch.cuda.Event()
        start.record(main)

        def launch_build(i):
            st = self.ring_streams[i % self.ahead]
            st.wait_event(stepped[i - self.ahead] if i >= self.ahead else start)
            with torch.cuda.stream(st):
                self.hip.build_plan(*self.stream.batch(first + i), self.stream.V, chunk_cap=self.cap,
                                    into=self.ring[i % self.ahead], ws=self.ring_ws[i % self.ahead])
                built[i] = torch.cuda.Event()
                built[i].record(st)
        for i in range(min(self.ahead, count)):
            launch_build(i)
        for i in range(count):
            main.wait_event(built[i])
            self._step(self.ring[i % self.ahead])
            stepped[i] = torch.cuda.Event()
            stepped[i].record(main)
            if i + self.ahead < count:
                launch_build(i + self.ahead)

    def run(self, n_steps: int) -> int:
        """Up to `n_steps` steps, never across an epoch boundary or a burst boundary; returns the number done."""
        nb = self.stream.batches_per_epoch
        if self.position >= nb:
            self.stream.reshuffle_in_place()
            self.position = 0
        first = self.position
        count = min(n_steps, nb - first, self.burst - first % self.burst)
        key = (first, count)
        if key not in self.graphs and len(self.graphs) < 256:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._issue(first, count)
            self.graphs[key] = graph
        if key in self.graphs:
            self.graphs[key].replay()
        else:
            self._issue(first, count)              # cache full: same sequence, launched eagerly
        self.position += count
        return count

    def read_loss(self) -> dict:
        """Host read of the last step's scalars (synchronises; call at the logging cadence only)."""
        loss, L, reg, _ = self.loss_out.tolist()
        return {"loss": loss, "weighted_mse": L, "regularization_loss": reg}
