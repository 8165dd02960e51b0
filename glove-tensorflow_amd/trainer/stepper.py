"""One optimizer step = `session.run(train_op)` of the reference (estimator.py:49-56), on one
GPU or data-parallel over the GPUs of a node.

Data-parallel form (new: the reference is single-process).  Every rank holds its
own shard of the nonzero stream and a full replica of the five variables and their slots.
Per step every rank runs the forward+gradient passes over its batch of B nonzeros with
inv_batch = 1 / (world * B), adds the summed gradients into one flat dense buffer
[G_R | G_C | G_br | G_bc | tail], the buffers are summed with ONE all-reduce (RCCL over xGMI when
the backend is "nccl") and every rank applies the identical dense update.  The result equals a
single-GPU step at batch size world * B up to fp32 summation order.

Touched-rows exchange.  The dense buffer has 2 V (d+1) floats whatever the batch touches.  When the ranks' lists of
(id, summed gradient row) are together shorter than that, every rank instead packs its list, the lists are
all-gathered, and every rank adds them into the (never zeroed) dense buffer in rank order and applies Adagrad to the
touched rows only: the same sum, then the same apply, with a payload that follows the batch and not the vocabulary
(`exchange="auto"` decides once from the id counts of the resident plans of all ranks).

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
        self.shard_rows = 0         # rows of this rank's row-table shard when the row ids handed in are shard-local (else 0)
        self.exchange = False       # the plans feed a multi-rank step: its packing passes read chunk records, not run words

    def build_plan(self, row, col, w, y, V, chunk_cap):
        return self.hip.build_plan(row.contiguous(), col.contiguous(), w.contiguous(), y.contiguous(), V,
                                   chunk_cap=chunk_cap, compact=True, d=self.row_floats, V_row=self.shard_rows,
                                   run_words=False if self.exchange else None)

    def make_hyper(self, **kw):
        from trainer.hip_api import make_hyper
        return make_hyper(**kw)

    def dense_grad_buffer(self, tables):
        return self.hip.dense_grad_buffer(tables)

    def step_sparse_adagrad(self, plan, tables, hyper, loss_out):
        if tables.optimizer == "Adagrad":
            self.hip.step_adagrad(plan, tables, hyper, loss_out)
        else:                                   # the other Keras names: passes + their apply epilogue (glove_step_sparse_f32)
            self.hip.step_sparse(plan, tables, hyper, None, loss_out)

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
        elif tables.optimizer in ("Adam", "RMSprop"):    # the dense-decay optimizers: every row's slots move every step
            from trainer.hip_api import OPTIMIZER_CODES
            hyper.optimizer = OPTIMIZER_CODES[tables.optimizer]
            self.hip.dense_adam(tables, hyper, G, loss_out)
        else:
            raise ValueError("no dense apply for %s" % tables.optimizer)

    # ---- pieces of the row-sharded step (hyper.sides selects the side)
    def passes(self, plan, tables, hyper):
        self.hip.passes(plan, tables, hyper)

    def apply_sparse(self, plan, tables, hyper):
        self.hip.apply_adagrad(plan, tables, hyper)

    def colpass(self, plan, tables, hyper):
        self.hip.colpass(plan, tables, hyper)

    def rowside_step(self, plan, tables, hyper):
        self.hip.rowside_step(plan, tables, hyper)

    def dense_grad(self, plan, tables, hyper, G):
        self.hip.dense_grad(plan, tables, hyper, G)

    def col_half(self, tables, G):
        """The contiguous [G_C | G_bc | tail] part of the flat buffer."""
        return G[self.hip.grad_layout(tables)["G_C"]:]

    # ---- touched-rows exchange
    def id_counts(self, plan):
        """(distinct row ids, distinct col ids) of a resident plan."""
        return plan.host_counts[1], plan.host_counts[3]

    def exchange_buffers(self, tables, capacity: int, world: int):
        """send [capacity, d + 4], recv [world, capacity, d + 4], mark int32 [V_row + V] (all zero between steps)."""
        f32 = dict(dtype=torch.float32, device=tables.device)
        return dict(send=torch.zeros(capacity, tables.d + 4, **f32), recv=torch.zeros(world, capacity, tables.d + 4, **f32),
                    mark=torch.zeros(tables.V_row + tables.V, dtype=torch.int32, device=tables.device), capacity=capacity)

    def pack_grad(self, plan, tables, hyper, send):
        self.hip.pack_grad(plan, tables, hyper, send)

    def passes_packing(self, plan, tables, hyper, send):
        """The passes of hyper.sides, writing the list entries of the ids a lane group holds completely on the way."""
        self.hip.passes_packing(plan, tables, hyper, send)

    def loss_partials(self, plan, tables, out4):
        """out4 = {sum e, sum w diff^2, sum |r|^2+|c|^2, sum b^2} of the plan's last row pass (floats 2..5 of a list header)."""
        self.hip.loss_partials(plan, tables, out4)

    def pack_rest(self, plan, tables, hyper, send):
        """The rest of the list passes_packing started (the other ids and the header)."""
        self.hip.pack_rest(plan, tables, hyper, send)

    def apply_gathered(self, bufs, world, tables, hyper, G, loss_out, tail=None):
        """Adds the ranks' lists into G in rank order, then Adagrad on every touched id (G needs no zeroing).
        tail: the loss partials summed over the ranks when the lists' headers do not carry them."""
        cap = bufs["capacity"]
        lists = bufs.get("_lists")
        if lists is None:
            lists = bufs["_lists"] = [self.hip.packed_list(bufs["recv"][r]) for r in range(world)]
        self.hip.count_packed(lists, tables, G, bufs["mark"], cap)       # ids one rank alone touched skip the dense buffer
        for r, lst in enumerate(lists):
            self.hip.combine_packed(lst, r, tables, G, bufs["mark"], cap)
        self.hip.apply_packed(lists, tables, hyper, G, bufs["mark"], tail, loss_out, cap)

    # ---- both tables sharded (ShardedStepper)
    def gather_rows(self, tables, idx, rows, biases):
        """rows[i] = C[idx[i]], biases[i] = bc[idx[i]] of this rank's col shard."""
        self.hip.gather_rows(tables.C, tables.bc, idx, rows, biases)

    def fetch_buffers(self, tables, capacity: int, serve_capacity: int):
        f32 = dict(dtype=torch.float32, device=tables.device)
        d = tables.d
        return dict(C=torch.zeros(max(capacity, 1), d, **f32), bc=torch.zeros(max(capacity, 1), **f32),
                    send_rows=torch.zeros(max(serve_capacity, 1), d, **f32), send_bias=torch.zeros(max(serve_capacity, 1), **f32),
                    packed=torch.zeros(1 + capacity, d + 4, **f32), recv=torch.zeros(max(serve_capacity, 1), d + 4, **f32))

    def col_view(self, tables, bufs, capacity: int):
        """The tables the passes of one batch see: this rank's row shard and, as the col table, the fetched rows
        (compact col ids index it); the col side's slots are never touched through this view."""
        from trainer.hip_api import TablesView
        return TablesView(tables, C=bufs["C"], bc=bufs["bc"], s1_C=bufs["C"], s1_bc=bufs["bc"],
                          V=max(capacity, tables.V_row), V_row=tables.V_row)

    def owner_apply(self, tables, state, recv, ids, counts, hyper, tail, loss_out):
        """The owner's half of the col side: `recv` holds, rank after rank, the summed gradient rows the ranks computed
        for this rank's col rows `ids` (owner-local indices); they are added in rank order and Adagrad is applied."""
        from trainer.hip_api import TablesView
        if "view" not in state:
            f32 = dict(dtype=torch.float32, device=tables.device)
            dummy, dummy_b = torch.zeros(4, tables.d, **f32), torch.zeros(4, **f32)
            # the col shard sits on the ROW side of this view (entries of side 0); its col side is a 4-row dummy
            view = TablesView(tables, R=tables.C, br=tables.bc, s1_R=tables.s1["C"], s1_br=tables.s1["bc"],
                              C=dummy, bc=dummy_b, s1_C=dummy, s1_bc=dummy_b, V=4, V_row=tables.C.shape[0],
                              keep=(dummy, dummy_b))
            state.update(view=view, G=self.hip.dense_grad_buffer(view),
                         mark=torch.zeros(view.V_row + view.V, dtype=torch.int32, device=tables.device))
        view, G, mark = state["view"], state["G"], state["mark"]
        lists, off = [], 0
        for n in counts:                      # a rank whose batch touches none of this owner's rows sends nothing
            if n:
                lists.append(self.hip.packed_list(recv[off:off + n], with_header=False, ids=ids[off:off + n], n=n, side=0))
            off += n
        if not lists:                         # nothing to apply: the scalar work (global bias, loss) still has to happen
            lists = [self.hip.packed_list(recv[0:1], with_header=False, ids=None, n=0, side=0)]
        self.hip.count_packed(lists, view, G, mark, 0)                    # ids one rank alone touched skip the dense buffer
        for k, lst in enumerate(lists):       # tag = position among the non-empty lists, in rank order
            if lst.n and len(lists) > 1:      # a single list: every id is its alone, nothing to combine
                self.hip.combine_packed(lst, k, view, G, mark, 0)
        self.hip.apply_packed(lists, view, hyper, G, mark, tail, loss_out, 0)

    def eval_sums(self, row, col, w, y, tables, sums):
        return self.hip.eval_sums(row.contiguous(), col.contiguous(), w.contiguous(), y.contiguous(), tables, sums)

    def eval_sums_logistic(self, row, col, pos, neg, tables, sums):
        return self.hip.eval_sums_logistic(row.contiguous(), col.contiguous(), pos.contiguous(), neg.contiguous(),
                                           tables, sums)

    def topk_cosine(self, R, query_ids, k):
        return self.hip.topk_cosine(R, query_ids, k)


def all_gather_rows(dist, recv, send, async_op=False):
    """recv[r] = rank r's `send` (equal shapes).  async_op: returns the work handle (wait() before reading recv)."""
    try:
        return dist.all_gather_into_tensor(recv.view(-1), send.view(-1), async_op=async_op)
    except (RuntimeError, NotImplementedError):          # a transport without the flat form
        return dist.all_gather([recv[r] for r in range(recv.shape[0])], send, async_op=async_op)


class SideCollective:
    """A collective that runs BESIDE the compute stream: started after what the compute stream has enqueued so far, awaited
    where its result is needed.  On a GPU it is issued — as an ordinary call, blocking on its stream only — from a side
    stream forked off and joined back with events.  INSIDE a hipGraph capture it is issued in line on the capturing stream
    instead (no overlap in a replayed step): in this torch / RCCL a collective on a stream forked off a capturing one, like a
    work handle awaited under capture, crashes the process (tools/dbg_rccl_graph.py), while collectives on the capturing
    stream itself capture and replay correctly.  On a CPU transport (the gloo tests) it is the transport's asynchronous form."""

    def __init__(self, device):
        # (high priority: HIP keeps the hardware queues of each priority apart — at normal priority this stream has been seen
        # sharing a queue with the compute stream, and a queue runs in order: no overlap at all)
        self.stream = torch.cuda.Stream(device=device, priority=-1) if torch.device(device).type == "cuda" else None
        self.done, self.work = None, None

    def start(self, issue):
        """issue(async_op) launches the collective."""
        if self.stream is None:
            self.work = issue(True)
            return
        if torch.cuda.is_current_stream_capturing():
            issue(False)
            return
        fork = torch.cuda.Event()
        fork.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(fork)
            issue(False)
            self.done = torch.cuda.Event()
            self.done.record(self.stream)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.done is not None:
            torch.cuda.current_stream().wait_event(self.done)
            self.done = None


# Captures use capture_error_mode="thread_local": RCCL's watchdog thread polls the events of earlier collectives, which a
# capture in the default (global) mode turns into an error in that thread ("operation not permitted when stream is capturing").
def transport_is_capturable(dist, multi: bool) -> bool:
    """Can a step be captured into a hipGraph?  Its kernels always; its collectives only on RCCL ("nccl" backend)."""
    return not multi or (dist is not None and dist.get_backend() == "nccl")


class GraphedSteps:
    """Steps of a static stream replayed from hipGraphs: ONE graph per resident batch holds every launch of its step —
    the kernels and, on RCCL, the collectives (issued in line on the capturing stream: SideCollective) — so a multi-rank step costs one graph launch instead of six or more launches plus a
    collective from Python.  A batch's step is captured the third time the batch comes round (short runs never pay for
    captures they do not replay); until then, and on a transport that cannot be captured (gloo), steps run eagerly.
    Same launches in the same order either way: same bits."""

    def enable_graphs(self, limit=4096, after=2):
        if self.tables.device.type == "cuda" and transport_is_capturable(self.dist, self._multi):
            self._graphs, self._seen, self._graph_limit, self._graph_after = {}, {}, int(limit), int(after)

    def step(self, item):
        graphs = getattr(self, "_graphs", None)
        if graphs is None:
            return self.step_eager(item)
        key = item if isinstance(item, int) else self._graph_key(item)
        # the captured launches hold raw pointers into the backend's shared, lazily growing step workspace: a plan with a
        # larger one reallocates it — every graph captured before that replays into freed memory and has to go
        gen = getattr(getattr(self.backend, "hip", None), "ws_generation", 0)
        g = graphs.get(key)
        if g is not None and g[2] != gen:
            torch.cuda.synchronize()
            del graphs[key]
            self._seen.pop(key, None)        # counted afresh: a workspace that keeps growing does not re-capture at every growth
            g = None
        if g is None:
            n = self._seen.get(key, 0)
            self._seen[key] = n + 1
            if n < self._graph_after or len(graphs) >= self._graph_limit:
                return self.step_eager(item)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self.step_eager(item)
            except Exception as exc:             # a transport that refuses capture: the same launches, eagerly, from now on
                if not self._multi:
                    raise
                import logging
                logging.getLogger(__name__).warning("hipGraph capture of the multi-rank step failed (%s: %s): launching eagerly",
                                                    type(exc).__name__, exc)
                torch.cuda.synchronize()
                self._graphs = None
                return self.step_eager(item)
            # (the graph holds raw pointers into the plan: keep the plan alive; and the workspace generation it saw at its end)
            graphs[key] = (g, item, getattr(getattr(self.backend, "hip", None), "ws_generation", 0))
            g = graphs[key]
        g[0].replay()

    _graph_serials = iter(range(1 << 62))

    @classmethod
    def _graph_key(cls, item):
        """A serial number given to the plan at first sight (kept on the object): unlike id(), never handed to a later plan
        that happens to be allocated at a dropped one's address."""
        key = getattr(item, "_graph_serial", None)
        if key is None:
            key = ("plan", next(cls._graph_serials))
            try:
                item._graph_serial = key
            except AttributeError:             # an object without attributes of its own: its address, as before
                key = ("id", id(item))
        return key

    def step_eager(self, item):
        for _, fn in self.phases():
            fn(item)

    def release_graphs(self):
        """Drops the captured steps.  Call before torch.distributed.destroy_process_group(): RCCL does not finish tearing a
        communicator down while hipGraphs that captured its collectives exist (the call hangs)."""
        if getattr(self, "_graphs", None) is not None:
            torch.cuda.synchronize()
            self._graphs, self._seen = {}, {}

    def __del__(self):           # graphs before the streams they were captured on (see ReshufflingRunner.__del__)
        try:
            if getattr(self, "_graphs", None):
                self.release_graphs()
        except Exception:
            pass


class Stepper(GraphedSteps):
    """`exchange`: "dense" = all-reduce of the flat dense gradient buffer; "rows" = all-gather of packed touched-row
    lists; "auto" = rows when `prepare(plans)` finds the ranks' lists together shorter than the dense buffer, else dense.
    Which optimizers (`tf.keras.optimizers.get`, reference train_utils.py:13-16) run on several ranks, and how:
    Adagrad either way; the per-row ones (SGD, Adamax, Adadelta, Ftrl: only touched rows move) on the touched-rows
    exchange, whose apply takes their epilogue; the dense-decay ones (Adam, RMSprop: every row's slots move every step) on
    the dense all-reduce.  Nadam (m, v decay everywhere, touched rows move) rides the touched-rows exchange too: the lists
    ARE the union of the ranks' ids — the rows no list names decay (a sweep in front of the apply), the named ones move."""

    ROWS_ONLY = ("SGD", "Adamax", "Adadelta", "Ftrl", "Nadam")       # touched-rows exchange (the lists carry the union of the ids)
    DENSE_ONLY = ("Adam", "RMSprop")                        # dense-decay optimizers: dense all-reduce

    def __init__(self, backend, tables, hyper_kwargs: dict, batch_size: int, world=1, dist=None, exchange="auto",
                 collectives=False):
        """collectives: go through the transport even with one rank (tests: RCCL is really called on a one-GPU box)."""
        self.backend, self.tables, self.world, self.dist = backend, tables, int(world), dist
        self._multi = self.world > 1 or bool(collectives)
        if self._multi and hasattr(backend, "exchange"):
            backend.exchange = True         # plans built from here on keep chunk records (the packing passes read them)
        if self._multi and dist is None:
            raise ValueError("world > 1 needs an initialised torch.distributed module")
        if exchange not in ("auto", "dense", "rows"):
            raise ValueError("exchange must be auto, dense or rows")
        if exchange == "rows" and tables.optimizer in self.DENSE_ONLY:
            raise ValueError("the touched-rows exchange is for optimizers that move touched rows only (Keras' %s moves every row "
                             "every step)" % tables.optimizer)
        if exchange == "dense" and self._multi and tables.optimizer in self.ROWS_ONLY:
            raise ValueError("%s runs on the touched-rows exchange (its dense form would need the ranks' id marks)" % tables.optimizer)
        if self._multi and tables.optimizer not in ("Adagrad",) + self.ROWS_ONLY + self.DENSE_ONLY:
            raise ValueError("the data-parallel form takes the eight Keras names (Adagrad, SGD, Adamax, Adadelta, Ftrl, Nadam, Adam, RMSprop), got %s" % tables.optimizer)
        if self._multi and tables.optimizer in self.ROWS_ONLY:
            exchange = "rows"
        self.hyper = backend.make_hyper(batch_size=batch_size * self.world, **hyper_kwargs)
        self.loss_out = torch.zeros(4, dtype=torch.float32, device=tables.device)
        # Adam (dense whole-table decay) and every multi-rank step go through the dense buffer; RMSprop (whole-slot decay) too,
        # inside its own entry point
        self.dense = self._multi or tables.optimizer == "Adam"
        self._rms_G = backend.dense_grad_buffer(tables) if tables.optimizer in ("RMSprop", "Nadam") and not self._multi else None
        form = int(hyper_kwargs.get("step_form", 0) or 0)
        if not self.dense and hasattr(tables, "maybe_enable_twin"):
            if form in (0, 5):
                tables.maybe_enable_tags(batch_size)   # small batches on small tables: the tagged step
            if form in (0, 4):
                tables.maybe_enable_twin()      # big tables: the fused step writes new rows beside the old ones
        elif not self._multi and tables.optimizer == "Adam" and form in (0, 5) and hasattr(tables, "maybe_enable_tags"):
            tables.maybe_enable_tags(batch_size)   # small batches on small tables: Adam's one-launch step on twinned tables
        self.G = backend.dense_grad_buffer(tables) if self.dense else None
        self.exchange, self.rows, self.bufs = exchange, False, None
        self.payload_floats = int(self.G.numel()) if self.G is not None else 0

    def prepare(self, plans=None, force_world=None, batch_size=None):
        """Agree (collectively) on the exchange.  Static stream: from the id counts of every rank's resident plans.
        Stream whose batches are indexed as they are used (`plans` None, `batch_size` given: reshuffled epochs): from the
        most ids a batch of that size can touch — the lists' capacity has to hold any batch."""
        world = self.world if force_world is None else force_world
        if self.tables.optimizer in self.DENSE_ONLY or self.exchange == "dense" or (
                world == 1 and not self._multi and self.exchange != "rows"):
            return
        if self.G is None:
            self.G, self.dense = self.backend.dense_grad_buffer(self.tables), True
        if plans is None:
            most = min(int(batch_size), self.tables.V_row) + min(int(batch_size), self.tables.V)
        else:
            most = max(sum(self.backend.id_counts(p)) for p in plans)
        if self._multi:
            t = torch.tensor([most], dtype=torch.int64, device=self.tables.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            most = int(t.item())
        capacity = 1 + most
        listed = self.world * capacity * (self.tables.d + 4)
        if self.exchange == "rows" or listed <= self.G.numel():
            self.rows = True
            self.bufs = self.backend.exchange_buffers(self.tables, capacity, self.world)
            self.payload_floats = listed

    def phases(self):
        """The step as named pieces [(name, fn(plan))]: what step() runs, in order (bench.py times them apart)."""
        b, t, h = self.backend, self.tables, self.hyper
        if not self.dense:
            if self._rms_G is not None:
                return [("step", lambda p: b.hip.step_sparse(p, t, h, self._rms_G, self.loss_out))]
            return [("step", lambda p: b.step_sparse_adagrad(p, t, h, self.loss_out))]
        if self.rows:
            ph = [("passes", lambda p: b.passes_packing(p, t, h, self.bufs["send"])),
                  ("pack_grad", lambda p: b.pack_rest(p, t, h, self.bufs["send"]))]
            if self._multi:
                ph.append(("all_gather", lambda p: all_gather_rows(self.dist, self.bufs["recv"], self.bufs["send"])))
            else:
                ph.append(("all_gather", lambda p: self.bufs["recv"][0].copy_(self.bufs["send"])))
            ph.append(("combine_apply", lambda p: b.apply_gathered(self.bufs, self.world, t, h, self.G, self.loss_out)))
            return ph
        ph = [("passes", lambda p: b.passes(p, t, h)), ("dense_grad", lambda p: b.dense_grad(p, t, h, self.G))]
        if self._multi:
            ph.append(("all_reduce", lambda p: self.dist.all_reduce(self.G)))     # sum over ranks; the tail carries the loss partials
        ph.append(("dense_apply", lambda p: b.apply_dense(t, h, self.G, self.loss_out)))
        return ph

    def step_many(self, plans):
        """Several consecutive steps; on one GPU they are issued by one C call."""
        if not self.dense and self.tables.optimizer == "Adagrad" and hasattr(self.backend, "steps_sparse_adagrad"):
            self.backend.steps_sparse_adagrad(plans, self.tables, self.hyper, self.loss_out)
        elif (not self._multi and self.tables.optimizer == "Adam" and not self.rows and hasattr(self.backend, "steps_dense_adam")):
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


class RowShardedStepper(GraphedSteps):
    """Model-parallel form of BASELINE config 5.  The row table R / br (and their Adagrad accumulators) are
    sharded by row id % world; the col table, the global bias and their slots are replicated.  Every rank
    steps on nonzeros whose rows it owns (see route_by_row_owner), so

      * the row side is completely local: rowpass / colpass, then a sparse Adagrad apply restricted to the
        row side (hyper.sides = 1) — no communication;
      * the col side is data parallel: either the rank's summed col gradients (hyper.sides = 2) go into the
        contiguous [G_C | G_bc | tail] half of the flat buffer, ONE all-reduce sums it over the ranks and every rank
        applies the identical dense update of C, bc and the global bias (`exchange="dense"`); or the ranks all-gather
        their packed lists of touched col rows and apply those ("rows": payload ~ distinct col ids of the batches
        instead of V; "auto" picks it in `prepare(plans)` when the lists are the shorter payload).

    With inv_batch = 1 / (world * B) the result equals a single-GPU step on the union of the ranks'
    batches (tests/test_dp_gloo.py).  On one rank nothing is exchanged and the step is the plain sparse one."""

    def __init__(self, backend, tables, hyper_kwargs: dict, batch_size: int, world: int, dist, exchange="auto",
                 collectives=False):
        if tables.optimizer != "Adagrad":
            raise ValueError("the row-sharded step is implemented for Adagrad (Keras Adam has no sparse form)")
        if exchange not in ("auto", "dense", "rows"):
            raise ValueError("exchange must be auto, dense or rows")
        self.backend, self.tables, self.world, self.dist = backend, tables, int(world), dist
        self._multi = self.world > 1 or bool(collectives)      # collectives: the transport even with one rank (tests)
        if self._multi and hasattr(backend, "exchange"):
            backend.exchange = True         # plans built from here on keep chunk records (the packing passes read them)
        gb = batch_size * self.world
        self.hyper = backend.make_hyper(batch_size=gb, **hyper_kwargs)
        self.hyper_rows = backend.make_hyper(batch_size=gb, sides=1, **hyper_kwargs)
        self.hyper_cols = backend.make_hyper(batch_size=gb, sides=2, **hyper_kwargs)
        self.loss_out = torch.zeros(4, dtype=torch.float32, device=tables.device)
        self.tail = torch.zeros(4, dtype=getattr(backend, "tail_dtype", torch.float32), device=tables.device)   # loss partials over the ranks
        self._gather = SideCollective(tables.device)        # the lists' all-gather while it is in flight
        self.G = backend.dense_grad_buffer(tables) if self._multi else None
        self.exchange, self.rows, self.bufs = exchange, False, None
        self.payload_floats = int(backend.col_half(tables, self.G).numel()) if self.G is not None else 0

    def prepare(self, plans=None, batch_size=None):
        """Agree (collectively) on the col-side exchange: from the col id counts of all resident plans (static stream) or,
        for batches indexed as they are used (`plans` None), from the most col ids a batch of `batch_size` pairs can touch."""
        if not self._multi or self.exchange == "dense":
            return
        most = min(int(batch_size), self.tables.V) if plans is None else max(self.backend.id_counts(p)[1] for p in plans)
        t = torch.tensor([most], dtype=torch.int64, device=self.tables.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        capacity = 1 + int(t.item())
        listed = self.world * capacity * (self.tables.d + 4)
        if self.exchange == "rows" or listed <= self.backend.col_half(self.tables, self.G).numel():
            self.rows = True
            self.bufs = self.backend.exchange_buffers(self.tables, capacity, self.world)
            self.payload_floats = listed

    def phases(self):
        b, t = self.backend, self.tables
        if not self._multi:
            return [("step", lambda p: b.step_sparse_adagrad(p, t, self.hyper, self.loss_out))]
        # the col pass first (it gathers the old rows of R), then the whole row side, applied in place by its pass where a
        # lane group holds an id completely: R, br are local, nothing else reads them in this step
        if self.rows:       # the col pass already writes the list entries of the col ids it sums completely
            ph = [("colpass", lambda p: b.passes_packing(p, t, self.hyper_cols, self.bufs["send"]))]
        else:
            ph = [("colpass", lambda p: b.colpass(p, t, self.hyper_cols))]
        if self.rows:
            # the lists travel while the row side runs: the all-gather is started (it waits for what this stream has
            # enqueued — the list is complete) and awaited after the row side; the loss partials, which the row pass
            # leaves, follow in a 4-float all-reduce
            def gather(p):
                self._gather.start(lambda a: all_gather_rows(self.dist, self.bufs["recv"], self.bufs["send"], async_op=a))

            def loss_tail(p):
                self._gather.wait()
                b.loss_partials(p, t, self.tail)
                self.dist.all_reduce(self.tail)

            ph += [("pack_grad_cols", lambda p: b.pack_rest(p, t, self.hyper_cols, self.bufs["send"])),      # reads C: before its update
                   ("all_gather", gather),
                   ("rowside_step", lambda p: b.rowside_step(p, t, self.hyper_rows)),
                   ("loss_tail", loss_tail),
                   ("combine_apply_cols", lambda p: b.apply_gathered(self.bufs, self.world, t, self.hyper_cols, self.G,
                                                                       self.loss_out, self.tail))]
        else:
            ph.append(("rowside_step", lambda p: b.rowside_step(p, t, self.hyper_rows)))
            ph += [("dense_grad_cols", lambda p: b.dense_grad(p, t, self.hyper_cols, self.G)),   # reads C (activity-L2 term): before its update
                   ("all_reduce", lambda p: self.dist.all_reduce(b.col_half(t, self.G))),
                   ("dense_adagrad_cols", lambda p: b.apply_dense(t, self.hyper_cols, self.G, self.loss_out))]
        return ph

    def finish_async(self):
        """Waits for the collective a phase started (for callers that time the phases one by one)."""
        self._gather.wait()

    def step_many(self, plans):
        for plan in plans:
            self.step(plan)

    def read_loss(self) -> dict:
        loss, L, reg, _ = self.loss_out.tolist()
        return {"loss": loss, "weighted_mse": L, "regularization_loss": reg}


class ShardedStepper(GraphedSteps):
    """BASELINE config 5 with BOTH tables sharded ("row-embedding table sharded across 8 GPUs with all-to-all token-id
    routing"): row u and col v live on ranks u % world and v % world at local index // world, with their
    Adagrad accumulators; only the global bias is replicated.  Nonzeros are routed to the owners of their ROWS
    (route_by_row_owner), so the row side of a step is local.  For the col side a rank

      1. receives the col rows its batch touches from their owners (all-to-all; the owners gather them),
      2. runs the col pass against that fetched block — the batch's col ids are renumbered 0 .. n-1 in fetch order —
         and then the whole row side (pass + Adagrad, in place where a lane group holds an id completely),
      3. returns the summed gradient of every fetched row to its owner (all-to-all of the packed list),
      4. and, as an owner, adds what the ranks returned for its rows in rank order and applies Adagrad to them.

    Per step and rank that moves 2 x (distinct col ids of its batch) x (d+1) floats, instead of the 2 V (d+1) of the
    dense all-reduce; the loss partials travel in one 4-float all-reduce.  The result equals a single-GPU step on the
    union of the ranks' batches.  The stream is static: `add_batch` (collective) prepares a batch once — the fetch
    lists, what this rank serves, the dedup index on the renumbered ids — and `step` replays it."""

    def __init__(self, backend, tables, hyper_kwargs: dict, batch_size: int, world: int, rank: int, dist, collectives=False,
                 exercise_exchange=False):
        """One rank owns every row: nobody else can contribute to a col id, so nothing has to wait for an exchange and the
        step is the plain single-GPU step (fused in place / on the twinned row table), bit for bit — what the sharded form
        costs at world 1 is what the plain step costs.  (With more ranks an own col id cannot be applied early: Keras sums
        the duplicates of ALL ranks' slices before it squares, a9 / a10, so its gradient waits for the other ranks' lists
        like everybody else's.)  `exercise_exchange`: run the full serve / fetch / push / owner-apply sequence on one rank
        all the same (tests: every collective of the form really goes through the transport on a one-GPU box)."""
        if tables.optimizer != "Adagrad":
            raise ValueError("the sharded step is implemented for Adagrad (Keras Adam has no sparse form)")
        self.backend, self.tables, self.world, self.rank, self.dist = backend, tables, int(world), int(rank), dist
        self._multi = self.world > 1 or bool(collectives)      # collectives: the transport even with one rank (tests)
        if self._multi and hasattr(backend, "exchange"):
            backend.exchange = True         # plans built from here on keep chunk records (the packing passes read them)
        self.local_only = self.world == 1 and not exercise_exchange
        self.col_per = 0        # > 0: the col ids handed in are numbered owner-major, ceil(V / world) per owner (the runner sets it from its stream)
        if hasattr(backend, "exchange"):
            backend.exchange = not self.local_only      # (alone in the world the step is the plain one: run words will do)
        if self.local_only and hasattr(tables, "maybe_enable_twin"):
            tables.maybe_enable_twin()
        gb = batch_size * self.world
        self.hyper = backend.make_hyper(batch_size=gb, **hyper_kwargs)
        self.hyper_rows = backend.make_hyper(batch_size=gb, sides=1, **hyper_kwargs)
        self.hyper_cols = backend.make_hyper(batch_size=gb, sides=2, **hyper_kwargs)
        self.loss_out = torch.zeros(4, dtype=torch.float32, device=tables.device)
        self.batches, self.bufs, self.view, self.owner_state, self._caps, self._dirty = [], None, None, {}, (0, 0), True
        self.tail = torch.zeros(4, dtype=getattr(backend, "tail_dtype", torch.float32), device=tables.device)   # loss partials, summed over ranks
        self._push = SideCollective(tables.device)          # the col gradients' all-to-all while it is in flight
        self._prep_pg = None                                # the communicator of the prepare's collectives (prepare_group)
        self._spare = {}                                    # staging plans of dropped batches, by (B, id bound, chunk cap)
        if hasattr(backend, "shard_rows"):
            backend.shard_rows = tables.V_row       # local row ids: anything outside the shard counts as id 0, like a bad col id

    def prepare_group(self):
        """The communicator the fetch lists' sizes and indices travel on — one of their own.  A transport runs the collectives of
        ONE communicator in the order they were issued: on the steps' communicator a batch prepared beside the steps (a dealt
        epoch: ReshufflingRunner._prepare_ahead) had its two small all-to-alls queued behind the fetch / push of every step
        already launched, and the host read of its sizes waited for all of them — steps and prepares took turns (V = 400 k,
        d = 300, B = 1 M, one rank with the exchange exercised: 1.89 ms per step against 1.34 on a static stream)."""
        if not self._multi:
            return None
        if self._prep_pg is None:
            self._prep_pg = self.dist.new_group()           # (collective: every rank prepares its first batch at the same point)
        return self._prep_pg

    def add_batch(self, row, col, w, y, chunk_cap=0) -> int:
        """row: this rank's LOCAL row indices; col: global col ids.  Collective; returns the batch's handle."""
        W, dist = self.world, self.dist
        pg = self.prepare_group()
        if self.local_only:                                             # local index = id: the plain step's plan
            plan = self.backend.build_plan(row, col, w, y, self.tables.V, chunk_cap)
            self.batches.append(dict(plan=plan, want=[0], serve=[0], serve_idx=None, n=0, ns=0))
            return len(self.batches) - 1
        uc = torch.unique(col.long())                                  # ascending
        per = getattr(self, "col_per", 0)                              # > 0: col ids numbered owner-major (NonzeroStream(cols_by_owner=))
        owner = uc // per if per else uc % W
        order = torch.argsort(owner, stable=True)                      # (owner, id) order = fetch order
        inv = torch.empty_like(order)
        inv[order] = torch.arange(order.numel(), device=order.device)
        compact = inv[torch.searchsorted(uc, col.long())].to(torch.int32)
        want = torch.bincount(owner, minlength=W)
        req = ((uc[order] % per) if per else (uc[order] // W)).to(torch.int32).contiguous()        # the owners' local indices
        serve = torch.empty_like(want)
        if self._multi:
            dist.all_to_all_single(serve, want, group=pg)
        else:
            serve.copy_(want)
        want_l, serve_l = [int(x) for x in want.tolist()], [int(x) for x in serve.tolist()]
        serve_idx = torch.empty(sum(serve_l), dtype=torch.int32, device=req.device)
        if self._multi:
            dist.all_to_all_single(serve_idx, req, serve_l, want_l, group=pg)
        else:
            serve_idx.copy_(req)
        n_uc = int(uc.numel())
        plan = self.backend.build_plan(row, compact, w, y, max(n_uc, self.tables.V_row), chunk_cap)
        self.batches.append(dict(plan=plan, want=want_l, serve=serve_l, serve_idx=serve_idx, n=n_uc, ns=sum(serve_l)))
        self._dirty = True                                               # capacities may have grown
        return len(self.batches) - 1

    def add_batch_dealt(self, row_side, col_side, first: int, B: int, chunk_cap: int) -> int:
        """The batch at positions [first, first + B) of a dealt epoch whose col ids are numbered owner-major (`col_per`): it lies
        sorted by local row id on the row side and by col id — i.e. in fetch order: owner after owner, local index ascending —
        on the col side, so nothing is sorted here: the distinct col ids are the runs of the col side, a pair's compact col id is
        the number of its run, and the index is numbered by glove_plan_build_sorted.  Collective; returns the batch's handle.
        Same fetch lists and the same plan as add_batch gives for the batch in row-major arrival order."""
        return self.add_batch_dealt_finish(self.add_batch_dealt_begin(row_side, col_side, first, B, chunk_cap))

    def add_batch_dealt_begin(self, row_side, col_side, first: int, B: int, chunk_cap: int) -> dict:
        """First half of add_batch_dealt: everything the device can do without the host knowing a size — the runs of the col
        side, the fetch counts per owner and their all-to-all, the index — issued on the current stream with NO host read; the
        sizes (distinct col ids, ids wanted from / served to every rank, the index's counts) start their way to one pinned
        block behind it.  add_batch_dealt_finish reads them.  A runner keeps a batch or two between the halves
        (ReshufflingRunner._prepare_one): by the time it reads, the copy has long landed, and the host never waits for a
        prepare launch queued behind the running step's kernels (six such waits per batch made a prepare 1.7 ms beside
        1.3 ms steps: the host fell behind and the steps waited for their fetch lists at every epoch's end).  Collective."""
        from trainer.hip_api import Pairs, PlanBlock, auto_chunk_cap
        hip, W, dist, per = self.backend.hip, self.world, self.dist, self.col_per
        pg = self.prepare_group()
        sl = slice(first, first + B)
        ids = col_side.id[sl]
        dev = ids.device
        i32 = dict(dtype=torch.int32, device=dev)
        new = torch.ones(B, **i32)                                       # 1 where a run of equal col ids starts
        new[1:] = (ids[1:] != ids[:-1]).to(torch.int32)
        upto = torch.cumsum(new, 0, dtype=torch.int32)                  # runs that have started up to and including a position
        run = upto - 1                                                   # the pair's compact col id = the number of its run
        if getattr(self, "_lut", None) is None:
            self._lut = torch.zeros(W * per, **i32)
            self._edges = torch.arange(0, (W + 1) * per, per, **i32)     # the owners' first col ids
        self._lut[ids] = run                                             # (the pairs of a run all write its number)
        req = torch.zeros(B, **i32)                                      # the owners' local indices, in fetch order; [:n_uc] is used
        req[run] = ids % per
        # ids wanted from every owner = runs that start inside its range of the (sorted) ids: W + 1 binary searches and the
        # running count at their positions (an index_add_ of the million flags onto W counters was 213 us of atomics on one
        # address at world 1)
        pos = torch.searchsorted(ids, self._edges)
        before = torch.where(pos > 0, upto[(pos - 1).clamp(min=0)], torch.zeros_like(upto[:1]))
        want = (before[1:] - before[:-1]).long()
        serve = torch.empty_like(want)
        if self._multi:
            dist.all_to_all_single(serve, want, group=pg)
        else:
            serve.copy_(want)
        # the index over compact col ids, while the counts travel.  Its capacities come from a bound of the distinct col ids
        # (the host does not know their number yet): no more than pairs, no more than columns
        def side(src, sid, partner):
            p = Pairs.__new__(Pairs)
            p.n, p.id, p.partner, p.w, p.y, p._struct = B, sid.contiguous(), partner.contiguous(), src.w[sl], src.y[sl], None
            return p
        rs = side(row_side, row_side.id[sl], self._lut[row_side.partner[sl]])
        cs = side(col_side, run, col_side.partner[sl])
        V_plan = max(min(B, W * per), self.tables.V_row)
        cap = chunk_cap or auto_chunk_cap(B, V_plan, self.tables.d)
        # the staging plan, its struct block and the pinned block of its sizes come back from the batches of epochs past
        # (drop_batches): allocating them anew — two dozen tensors, two pinned blocks, a pageable copy of the structs, each
        # of the last three a wait for the stream — was most of a prepare's host time
        key = (B, V_plan, cap)
        pool = self._spare.setdefault(key, [])
        if pool:
            plan, block, sizes_host = pool.pop()
        else:
            plan = hip.staging_plan(B, V_plan, cap, dev, V_row=self.tables.V_row, records=True)
            block = PlanBlock([plan])
            sizes_host = torch.empty(2 * W + 1, dtype=torch.int64)
            if dev.type == "cuda":
                sizes_host = sizes_host.pin_memory()
        if getattr(self, "_sorted_ws", None) is None or self._sorted_ws_B != B:
            self._sorted_ws = torch.empty(max(hip.lib.glove_plan_sorted_workspace_bytes(B, 1), 256), dtype=torch.uint8, device=dev)
            self._sorted_ws_B = B
        hip.build_plans_sorted(rs, cs, 0, block, 1, V_plan, self._sorted_ws)
        block.fetch_counts()
        sizes = torch.cat([want, serve, run[-1:].long() + 1])
        sizes_host.copy_(sizes, non_blocking=True)
        landed = None
        if dev.type == "cuda":
            landed = torch.cuda.Event()
            landed.record()
        return dict(plan=plan, block=block, req=req, sizes=sizes_host, landed=landed, keep=(sizes, rs, cs), spare_key=key)

    def add_batch_dealt_finish(self, half: dict) -> int:
        """Second half: the sizes on the host (the only wait of a prepare, for a copy issued a batch or two ago), the ids this
        rank serves by all-to-all.  Collective; returns the batch's handle."""
        W, dist = self.world, self.dist
        if half["landed"] is not None:
            half["landed"].synchronize()
        got = [int(x) for x in half["sizes"].tolist()]
        want_l, serve_l, n_uc = got[:W], got[W:2 * W], got[2 * W]
        half["block"].adopt_counts(1)
        req = half["req"][:n_uc]
        serve_idx = torch.empty(sum(serve_l), dtype=torch.int32, device=req.device)
        if self._multi:
            dist.all_to_all_single(serve_idx, req, serve_l, want_l, group=self.prepare_group())
        else:
            serve_idx.copy_(req)
        self.batches.append(dict(plan=half["plan"], want=want_l, serve=serve_l, serve_idx=serve_idx, n=n_uc, ns=sum(serve_l),
                                 block=half["block"], spare=(half["spare_key"], half["sizes"])))
        self._dirty = True
        return len(self.batches) - 1

    def clear_batches(self):
        """Forgets the prepared batches (a reshuffled epoch prepares its own); the fetch buffers are kept while they fit."""
        self.batches = []
        if getattr(self, "_graphs", None) is not None:
            self._graphs, self._seen = {}, {}       # graphs of the old batches point at their plans

    def drop_batches(self, handles):
        """Forgets these prepared batches only (the handles of the others stay valid): an epoch that has been stepped through,
        while the next one's batches — prepared beside its steps — stay."""
        for h in handles:
            bt = self.batches[h]
            if bt is not None and "spare" in bt:
                # (the caller orders the next refill behind the steps that still read the plan: ReshufflingRunner._prepare_epoch)
                self._spare.setdefault(bt["spare"][0], []).append((bt["plan"], bt["block"], bt["spare"][1]))
            self.batches[h] = None
            if getattr(self, "_graphs", None):
                self._graphs.pop(h, None)
                self._seen.pop(h, None)

    def _ready(self):
        if not self._dirty:
            return
        self._dirty = False
        live = [b for b in self.batches if b is not None]
        cap, scap = max(b["n"] for b in live), max(b["ns"] for b in live)
        if self.bufs is None or cap > self._caps[0] or scap > self._caps[1]:
            self._caps = (cap, scap)
            if getattr(self, "_graphs", None) is not None:
                self._graphs = {}                   # captured steps point into the old buffers
            self.bufs = self.backend.fetch_buffers(self.tables, cap, scap)
            self.view = self.backend.col_view(self.tables, self.bufs, cap)
            self.payload_floats = 2 * cap * (self.tables.d + 1)

    def phases(self):
        """[(name, fn(batch handle))]."""
        b, t, dist, W = self.backend, self.tables, self.dist, self.world
        if self.local_only:
            bt = self.batches
            ph = [("step", lambda i: b.step_sparse_adagrad(bt[i]["plan"], t, self.hyper, self.loss_out))]
            if self._multi:         # `collectives`: the loss scalars still travel through the transport (a sum over one rank)
                ph.append(("loss_tail", lambda i: dist.all_reduce(self.loss_out)))
            return ph
        self._ready()
        f, bt = self.bufs, self.batches

        def a2a(out, inp, out_split, in_split):
            if self._multi:
                dist.all_to_all_single(out, inp, out_split, in_split)
            else:
                out.copy_(inp)

        def fetch(i):
            n, ns = bt[i]["n"], bt[i]["ns"]
            a2a(f["C"][:n], f["send_rows"][:ns], bt[i]["want"], bt[i]["serve"])
            a2a(f["bc"][:n], f["send_bias"][:ns], bt[i]["want"], bt[i]["serve"])

        def push(i):
            # started, not awaited: the row side below runs while the gradients travel (the collective waits for what
            # this stream has enqueued so far — the packed list is complete — and runs on the transport's own stream)
            n, ns = bt[i]["n"], bt[i]["ns"]
            if self._multi:
                self._push.start(lambda a: dist.all_to_all_single(f["recv"][:ns], f["packed"][1:1 + n], bt[i]["serve"],
                                                                  bt[i]["want"], async_op=a))
            else:
                f["recv"][:ns].copy_(f["packed"][1:1 + n])

        def loss_tail(i):
            self._push.wait()
            # the loss partials come from the row pass, which ran after the list left: handed to the owners' apply apart
            b.loss_partials(bt[i]["plan"], self.view, self.tail)
            if self._multi:
                dist.all_reduce(self.tail)

        return [("serve_rows", lambda i: b.gather_rows(t, bt[i]["serve_idx"], f["send_rows"], f["send_bias"])),
                ("fetch_all_to_all", fetch),
                # the col pass first: it gathers the OLD rows of R, which the row side then updates in place
                # (it also writes the list entries of the col ids it sums completely)
                ("colpass", lambda i: b.passes_packing(bt[i]["plan"], self.view, self.hyper_cols, f["packed"])),
                ("pack_grad_cols", lambda i: b.pack_rest(bt[i]["plan"], self.view, self.hyper_cols, f["packed"])),
                ("push_all_to_all", push),
                ("rowside_step", lambda i: b.rowside_step(bt[i]["plan"], self.view, self.hyper_rows)),
                ("loss_tail", loss_tail),
                ("owner_apply_cols", lambda i: b.owner_apply(t, self.owner_state, f["recv"], bt[i]["serve_idx"], bt[i]["serve"],
                                                             self.hyper, self.tail, self.loss_out))]

    def finish_async(self):
        """Waits for the collective a phase started (for callers that time the phases one by one)."""
        self._push.wait()

    def read_loss(self) -> dict:
        loss, L, reg, _ = self.loss_out.tolist()
        return {"loss": loss, "weighted_mse": L, "regularization_loss": reg}


class ReshufflingRunner:
    """Training over a stream whose pairs are re-permuted every epoch (`--epoch-shuffle full`, the reference's
    `make_csv_dataset(shuffle=True, num_epochs=None)`, data_utils.py:12-21): the batches are new every epoch, so Keras'
    Unique + UnsortedSegmentSum (a9) — here the dedup index of a batch — cannot be prepared at load.  One GPU, or any of the
    multi-GPU forms through its `stepper`.

    On the HIP backend the stream is DEALT (trainer.data_utils.NonzeroStream: the rank's pairs were sorted once, into a
    row-major and a col-major master order; an epoch is one stable partition pass of both by batch number under a keyed
    bijection, drawn and written on the stream's side stream while the epoch before trains), so every batch arrives sorted on
    both sides and nothing is sorted per step.  What is left of the index — numbering chunks and ids, writing the chunk
    records — is done a SEGMENT of consecutive batches at a time: three launches (glove_plan_build_sorted) refill the
    staging plans of one of two slots on the side stream while the steps of the segment before run from the other slot.  The
    builds are ordinary launches ordered by events; only the steps are replayed from hipGraphs: a graph holds 2^k
    consecutive steps of one slot on ONE stream (kernels and, on RCCL, the collectives).

      * single GPU / data parallel / row-sharded (`Stepper`, `RowShardedStepper`): every rank deals ITS shard (no pair
        changes rank);
      * both tables sharded (`ShardedStepper`): a batch also needs its fetch lists agreed between the ranks
        (`add_batch`, collective), so the epoch's batches are prepared together when the epoch starts.

    With a test backend (no `hip`: the gloo tests on CPU) the same sequence runs eagerly, one synchronous build per step.
    On one rank every form gives exactly what the single-GPU runner gives.
    """

    GRAPH_MAX_BATCH = 8192

    def __init__(self, hip, stream, tables, hyper, chunk_cap=0, burst=64, stepper=None, graphs=None, segment=0,
                 slot_bytes=16 << 30, max_graphs=256):
        """`segment`: batches per index build (0: up to 64 — more would push the staging plans of small batches out of the caches —, as the epoch and `slot_bytes` of staging plans per slot allow);
        `burst`: most steps per graph replay; `graphs`: None = replay the steps from hipGraphs up to GRAPH_MAX_BATCH pairs per
        batch (the latency-bound regime: a step is one to three short launches) and issue them from one C call per run beyond
        (measured, graphs / C loop, us per step: B = 16,384 17.5 / 17.2; text8 B = 131,072 32.0 / 29.4; V = 50 k, d = 300
        120 / 117; C4 674 / 687), True / False force it."""
        if graphs is None:
            graphs = stream.B <= self.GRAPH_MAX_BATCH
        from trainer.hip_api import auto_chunk_cap
        self.hip, self.stream, self.tables, self.hyper, self.stepper = hip, stream, tables, hyper, stepper
        self.cap = chunk_cap or auto_chunk_cap(stream.B, stream.V, tables.d)
        self.burst = max(1, int(burst))
        self.graphs_on = bool(graphs) and hip is not None and (
            stepper is None or transport_is_capturable(stepper.dist, stepper._multi))
        self.sharded = isinstance(stepper, ShardedStepper)
        self.loss_out = stepper.loss_out if stepper is not None else torch.zeros(4, dtype=torch.float32, device=tables.device)
        self.graphs, self.max_graphs = {}, int(max_graphs)
        self.position = 0                      # next batch of the current epoch
        self.handles = None                    # both tables sharded: the epoch's prepared batches
        self._ahead = []                       # ... and those of the next epoch prepared so far (dealt streams)
        self._begun = []                       # ... and those begun, not finished (_prepare_one)
        # (high priority: short launches that should not queue behind a step's workgroups — and HIP keeps the hardware queues
        # of each priority apart, so this stream shares none with the steps' streams)
        self._prep = torch.cuda.Stream(device=tables.device, priority=-1) if tables.device.type == "cuda" else None
        # batches per epoch: the ranks' shards differ in length (by one pair data parallel, by the ownership of the rows
        # when routed), epochs end together: everybody steps through as many batches as the shortest shard has — the
        # pairs behind them wait for the next permutation, like the `nnz mod B` behind a rank's last full batch
        self.nb = stream.batches_per_epoch
        if stepper is not None and stepper._multi:
            t = torch.tensor([self.nb], dtype=torch.int64, device=tables.device)
            stepper.dist.all_reduce(t, op=stepper.dist.ReduceOp.MIN)
            self.nb = int(t.item())
        stream.reshuffle_in_place()
        if self.sharded:
            stepper.col_per = int(getattr(stream, "col_per", 0) or 0)
            self._prepare_epoch()
            return
        if hip is None:
            return
        if getattr(stream, "masters", None) is None:
            raise ValueError("the HIP runner steps over a dealt stream: NonzeroStream(..., static_plans=False) on the HIP backend")
        dev, B, V = tables.device, stream.B, stream.V
        shard_rows = tables.V_row if tables.V_row < tables.V else 0          # row-sharded: the stream carries shard-local row ids
        single = stepper is None
        # ---- what the staging plans carry, decided once from the first batch (one sync, here at set-up): chunk records where
        # the library takes a fused step (big batches on big tables, one GPU, Adagrad: it judges a device-refilled plan by the
        # most ids its batch can hold) or where the chunks are reasonably filled (the rule of Plan.compact); otherwise pair
        # arrays of their own.  A plan with records keeps no pair arrays: the records hold the pair fields.
        from trainer.hip_api import FUSED_STEP_BYTES, PlanBlock, staging_records
        form = getattr(hyper, "step_form", 0)
        probe = hip.build_plan(*(t.contiguous() for t in stream.batch(0)), V, chunk_cap=self.cap, V_row=shard_rows, links=False)
        counts = probe.counts.tolist()
        del probe
        fused = single and tables.optimizer == "Adagrad" and (
            form in (2, 3, 4) or (form == 0 and bool(staging_records(B, tables.V_row, V, tables.d)) and
                                  (counts[1] + counts[3]) * tables.d * 16 >= FUSED_STEP_BYTES))
        from trainer.hip_api import RECORDS_AT_BUILD_MAX
        records = fused or B <= RECORDS_AT_BUILD_MAX or 4 * B >= self.cap * max(counts[0], counts[2], 1)
        if fused and form == 0 and hasattr(tables, "maybe_enable_twin"):
            tables.maybe_enable_twin()      # as Stepper does: the fused step writes new rows beside the old ones
        # a fused step on a batch indexed every step: no records at all — their 128-byte lines would be written once and read
        # once — but the pair fields as they were dealt (12 B per pair) + a run word per chunk (glove_plan.r_chunk_hw)
        self.run_words = bool(fused) and form != 2
        if self.run_words:
            records = False
        self.records = records
        if single and records and tables.optimizer in ("Adagrad", "Adam") and form in (0, 5) and hasattr(tables, "maybe_enable_tags"):
            tables.maybe_enable_tags(B)     # small batches on small tables: the tagged step (one launch for all the row work)
        # ... and the pair fields are not even copied: the plans point into the epoch's arrays (the steps are issued by C calls
        # here, not replayed from graphs that would hold last epoch's addresses); a deal then waits for the steps that read
        # the buffer set it overwrites
        # (copied instead: +10 ... 16 us per C4 step, A/B on one box)
        self.borrow = self.run_words and not self.graphs_on
        stream.main_reads_epochs = self.borrow  # otherwise only the side stream's builds read the epoch buffers from here on
        staging = lambda: hip.staging_plan(B, V, self.cap, dev, V_row=shard_rows, records=records, run_words=self.run_words,
                                           borrow=self.borrow)
        first = staging()
        per_plan = max(first.nbytes(), 1)
        self.S = int(segment) if segment else max(1, min(64, self.nb, int(slot_bytes) // per_plan))
        self.S = max(1, min(self.S, self.nb))
        plans = [first] + [staging() for _ in range(2 * self.S - 1)]
        self.slots = [PlanBlock(plans[:self.S]), PlanBlock(plans[self.S:])]
        self.sorted_ws = torch.empty(max(hip.lib.glove_plan_sorted_workspace_bytes(B, self.S), 256), dtype=torch.uint8, device=dev)
        self.step_ws = torch.empty(hip.lib.glove_step_workspace_bytes(B, first.cap_chunks, tables.d), dtype=torch.uint8, device=dev)
        self.G = hip.dense_grad_buffer(tables) if single and tables.optimizer in ("Adam", "RMSprop", "Nadam") else None
        # ---- segments: `_g` = the segment the next step belongs to (counted over all epochs), slot = segment % 2
        self._g, self._issued, self._entered = 0, 0, -1
        self._cursor = (stream.epoch, 0)       # (epoch, segment of the epoch) the next build takes
        self._built, self._built_n, self._freed = {}, {}, {}
        # One GPU, steps issued by C calls (no graphs: big batches): the counts of every indexed batch come back to the host
        # with the build, so that a staging plan is stepped like a resident one — exact grids, the form picked by the ids the
        # batch really holds (V = 50 k, d = 300: 107 -> 99 us per step; V = 2 M, d = 128: 537 -> 517).  A captured graph bakes
        # its grids in: there the plans keep their counts on the device.
        self.host_counts = single and not self.graphs_on
        self._issue_build()                    # segment 0: needed now anyway
        # every kernel (and collective) of a step runs once outside any capture, on throw-away tables of the same shape
        from trainer.hip_api import DeviceTables
        real = self.tables
        scratch = DeviceTables(real.V, real.d_model, real.optimizer, device=dev, seed=0, V_row=real.V_row)
        if getattr(real, "R_ver", None) is not None:
            scratch.enable_twin()           # the same step forms are legal on the scratch tables
        if getattr(real, "R_tag", None) is not None:
            scratch.enable_tags()
        self._swap_tables(scratch)
        torch.cuda.current_stream().wait_event(self._built[0])
        if self.host_counts:
            self._built[0].synchronize()
            self.slots[0].adopt_counts(self._built_n[0])
        self._step(self.slots[0].plans[0])
        torch.cuda.synchronize()
        self._swap_tables(real)
        if self.G is not None:
            self.G.zero_()
        if stepper is not None and getattr(stepper, "G", None) is not None:
            stepper.G.zero_()

    def _swap_tables(self, tables):
        self.tables = tables
        if self.stepper is not None:
            self.stepper.tables = tables

    def _step(self, plan):
        if self.stepper is not None:
            self.stepper.step(plan)
        elif self.tables.optimizer == "Adagrad":
            self.hip.step_adagrad(plan, self.tables, self.hyper, self.loss_out, self.step_ws)
        elif self.tables.optimizer == "Adam":
            self.hip.step_adam(plan, self.tables, self.hyper, self.G, self.loss_out, self.step_ws)
        else:                                   # the other Keras names (glove_step_sparse_f32)
            self.hip.step_sparse(plan, self.tables, self.hyper, self.G, self.loss_out, self.step_ws)

    def _segments_per_epoch(self) -> int:
        return (self.nb + self.S - 1) // self.S

    def _issue_build(self):
        """The index of the next segment — three launches on the stream's side stream, behind the deal of its epoch and the
        last step that read the slot it refills."""
        g, slot = self._issued, self._issued % 2
        epoch, seg = self._cursor
        first = seg * self.S
        n = min(self.S, self.nb - first)
        rs, cs = self.stream.epoch_sides(epoch)             # (issues the epoch's deal if nobody has yet)
        side = self.stream.side
        if slot in self._freed:
            side.wait_event(self._freed[slot])
        with torch.cuda.stream(side):
            self.hip.build_plans_sorted(rs, cs, first, self.slots[slot], n, self.stream.V, self.sorted_ws)
            if self.host_counts:
                self.slots[slot].fetch_counts()
            ev = torch.cuda.Event()
            ev.record(side)
        self._built[g] = ev
        self._built_n[g] = n
        self._issued += 1
        self._cursor = (epoch, seg + 1) if seg + 1 < self._segments_per_epoch() else (epoch + 1, 0)

    def _prepare_epoch(self):
        """Both tables sharded: the fetch lists and indexes of all batches of the epoch (collective).  On a dealt stream the
        next epoch exists while this one trains: its batches are prepared one per step, on a stream of their own, beside the
        steps (`_prepare_ahead`); what is still missing at the boundary is prepared here."""
        if getattr(self.stream, "masters", None) is None:
            self.stepper.clear_batches()
            self.handles = [self.stepper.add_batch(*(t.contiguous() for t in self.stream.batch(b)), self.cap) for b in range(self.nb)]
            return
        old = self.handles or []
        self.handles, self._ahead = self._ahead, []
        while len(self.handles) + len(self._begun) < self.nb:      # (the first epoch; an epoch shorter than the steps that were run in it)
            self._prepare_one(self.stream.epoch, self.handles)
        self._finish_begun(self.handles)
        main = torch.cuda.current_stream()
        main.wait_stream(self._prep)                    # the plans and fetch lists are complete before a step reads them
        self.stepper.drop_batches(old)
        # (their memory goes back to the prepare stream's pool: what that stream does from here on — it may reuse the blocks —
        # waits for the steps of the old epoch that are still running)
        self._prep.wait_stream(main)

    PREPARES_IN_FLIGHT = 2

    def _prepare_one(self, epoch: int, into: list):
        """The next batch of `epoch` (batch len(into) + those begun and not finished): its col ids' fetch lists agreed between
        the ranks and its index, issued on the prepare stream — the steps on the compute stream keep running.  On a dealt
        stream a prepare is begun here and finished PREPARES_IN_FLIGHT calls later (ShardedStepper.add_batch_dealt_begin /
        _finish): its one host read finds the sizes already there."""
        b = len(into) + len(self._begun)
        rs, cs = self.stream.epoch_sides(epoch)
        with torch.cuda.stream(self._prep):
            if b == 0:
                self._prep.wait_event(self.stream.dealt_event(epoch))
            B = self.stream.B
            if self.stepper.col_per and not self.stepper.local_only:        # (col ids numbered owner-major: the batch arrives sorted for this form too)
                self._begun.append(self.stepper.add_batch_dealt_begin(rs, cs, b * B, B, self.cap))
                if len(self._begun) > self.PREPARES_IN_FLIGHT:
                    into.append(self.stepper.add_batch_dealt_finish(self._begun.pop(0)))
            else:
                into.append(self.stepper.add_batch(*(t.contiguous() for t in rs.arrays(b * B, (b + 1) * B)), self.cap))

    def _finish_begun(self, into: list):
        with torch.cuda.stream(self._prep):
            while self._begun:
                into.append(self.stepper.add_batch_dealt_finish(self._begun.pop(0)))

    def _prepare_ahead(self):
        """One batch of the NEXT epoch per step of this one (every rank the same sequence: the collectives inside line up)."""
        if getattr(self.stream, "masters", None) is None:
            return
        if len(self._ahead) + len(self._begun) < self.nb:
            self._prepare_one(self.stream.epoch + 1, self._ahead)
        elif self._begun:
            with torch.cuda.stream(self._prep):
                self._ahead.append(self.stepper.add_batch_dealt_finish(self._begun.pop(0)))

    def _launch(self, slot: int, off: int, count: int):
        """Steps off .. off + count - 1 of a slot: replayed from the hipGraph of that run (captured the first time it is
        asked for), or launched one by one."""
        plans = self.slots[slot].plans
        key = (slot, off, count)
        graph = self.graphs.get(key) if self.graphs_on else None
        if self.graphs_on and graph is None and len(self.graphs) < self.max_graphs:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    self._steps(plans[off:off + count])
            except Exception as exc:                # a transport that refuses capture: the same launches, eagerly, from now on
                if self.stepper is None:
                    raise
                import logging
                logging.getLogger(__name__).warning("hipGraph capture of the multi-rank step failed (%s: %s): launching eagerly",
                                                    type(exc).__name__, exc)
                torch.cuda.synchronize()
                self.graphs_on, self.graphs, graph = False, {}, None
            else:
                self.graphs[key] = graph
        if graph is not None:
            graph.replay()
            return
        self._steps(plans[off:off + count])

    def _steps(self, plans):
        """Consecutive steps.  One GPU, Adagrad: one host call (on step-tagged tables the library chains them: one launch per
        step, the global bias handed on through the workspace)."""
        if self.stepper is None and self.tables.optimizer == "Adagrad":
            self.hip.steps_adagrad(plans, self.tables, self.hyper, self.loss_out, ws=self.step_ws)
        elif self.stepper is None and self.tables.optimizer == "Adam":
            self.hip.steps_adam(plans, self.tables, self.hyper, self.G, self.loss_out, ws=self.step_ws)
        else:
            for plan in plans:
                self._step(plan)

    def run(self, n_steps: int) -> int:
        """Up to `n_steps` steps, never across a segment's or the epoch's end; returns the number done (the caller asks again)."""
        if n_steps <= 0:
            return 0
        nb = self.nb
        if self.position >= nb:
            self.stream.reshuffle_in_place()
            self.position = 0
            if self.sharded:
                self._prepare_epoch()
        first = self.position
        if self.sharded:
            count = min(n_steps, nb - first, self.burst)
            # a step, then a batch of the next epoch, in turn: a prepare reads its sizes PREPARES_IN_FLIGHT batches late, so the
            # host stays that far ahead of the prepare stream and never waits for the steps.  (All steps of the run first, then
            # the prepares, left the prepares waiting for the whole run where their stream shared a hardware queue with the
            # stream of the steps' push: a HIP process has four queues by default, a queue runs in order, and every push in it
            # waits for its step — 1.7 - 1.9 ms per step against 1.25 on a static stream, rocprofv3 queue ids.)
            for b in range(first, first + count):
                self.stepper.step(self.handles[b])
                self._prepare_ahead()
        elif self.hip is None:                     # a test backend: one synchronous build per step
            count = min(n_steps, nb - first, self.burst)
            for b in range(first, first + count):
                self.stepper.step(self.stepper.backend.build_plan(*self.stream.batch(b), self.stream.V, self.cap))
        else:
            S = self.S
            seg, off = first // S, first % S
            seg_len = min(S, nb - seg * S)
            g, slot = self._g, self._g % 2
            main = torch.cuda.current_stream()
            if self._entered != g:                  # the segment's first step: its index, then the next segment's behind it
                while self._issued <= g:
                    self._issue_build()
                ev = self._built.pop(g)
                main.wait_event(ev)
                if self.host_counts:
                    # the counts of the segment's batches, read back: its steps size their grids and pick their form by what
                    # the batches hold, as on resident plans (the build was issued a segment ago: the wait is short, and the
                    # host stays at most one segment ahead of the GPU)
                    ev.synchronize()
                    self.slots[slot].adopt_counts(self._built_n.pop(g))
                self._entered = g
                if self._issued == g + 1:
                    self._issue_build()            # into the other slot, free since the segment before this one ended
            count = min(n_steps, seg_len - off, self.burst)
            if self.graphs_on:
                count = 1 << (count.bit_length() - 1)      # the largest power of two that fits: the caller comes back for the rest
            self._launch(slot, off, count)
            if off + count == seg_len:
                ev = torch.cuda.Event()
                ev.record(main)
                self._freed[slot] = ev
                self._g += 1
        self.position += count
        return count

    def release_graphs(self):
        """Drops the captured runs of steps (before the process group is destroyed: see GraphedSteps.release_graphs)."""
        if self.graphs:
            torch.cuda.synchronize()
            self.graphs = {}

    def __del__(self):
        try:
            self.release_graphs()
        except Exception:
            pass

    def read_loss(self) -> dict:
        """Host read of the last step's scalars (synchronises; call at the logging cadence only)."""
        loss, L, reg, _ = self.loss_out.tolist()
        return {"loss": loss, "weighted_mse": L, "regularization_loss": reg}
