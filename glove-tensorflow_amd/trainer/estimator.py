"""`python -m trainer.estimator` — the GloVe trainer (reference src/models/estimator.py).

Same command line, same job_dir layout and the same three modes as the reference's `model_fn`
(TRAIN / EVAL / PREDICT, estimator.py:13-56), but the graph the reference hands to
`tf.estimator.train_and_evaluate` (estimator.py:95) is replaced by a host loop that calls the
hand-written HIP kernels of libglove_hip.so through the C ABI.  One process per GPU; launch with
`python -m torch.distributed.run --nproc-per-node N -m trainer.estimator ...` for data parallel.
"""
from __future__ import annotations

import json
import logging
import math
import os
import sys
import time

import torch

from trainer.config_utils import parse_args
from trainer.data_utils import NonzeroStream, file_lines, get_id_string_table, load_interaction_csv
from trainer.model_utils import MatrixFactorisation, get_predictions, logged_biases, summary_histograms, summary_values
from trainer.stepper import HipBackend, Stepper
from trainer.train_utils import CheckpointManager, get_optimizer

logger = logging.getLogger(__name__)


def _dist_env():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return world, int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


class Estimator:
    """The slice of `tf.estimator.Estimator` the reference uses (train_utils.py:30-36):
    model_dir = job_dir, periodic checkpoints, auto-resume, train / evaluate / predict."""

    def __init__(self, params: dict, backend=None, dist=None, device=None):
        self.params = params
        self.world, self.rank, local_rank = _dist_env() if dist is not None or backend is None else (1, 0, 0)
        self.dist = dist
        if backend is None:
            if not torch.cuda.is_available():
                raise RuntimeError("trainer.estimator needs an MI355X: the HIP path has no CPU fallback")
            torch.cuda.set_device(local_rank)
            device = torch.device("cuda", local_rank)
            if self.world > 1 and dist is None:
                import torch.distributed as dist
                dist.init_process_group("nccl", device_id=device)
                self.dist = dist
            backend = HipBackend(device)
        self.backend, self.device = backend, torch.device(device)
        self.vocab_size = file_lines(params["vocab_txt"])
        opt = get_optimizer(params["optimizer"], learning_rate=params["learning_rate"])
        self.optimizer_name = opt["class_name"]
        seed = params.get("seed")
        if self.world > 1 and seed is None:
            # the ranks cut ONE permutation of the stream into shards: they must agree on it even when the
            # run is unseeded (the reference's default), so rank 0 draws the seed
            box = [int(torch.seed() % (1 << 31)) if self.rank == 0 else None]
            self.dist.broadcast_object_list(box, src=0)
            seed = self.params["seed"] = box[0]
        self.model = MatrixFactorisation(self.vocab_size, params["embedding_size"], params["l2_reg"],
                                         optimizer=self.optimizer_name, device=self.device,
                                         seed=None if seed is None else seed + 1)
        if hasattr(self.backend, "row_floats"):
            self.backend.row_floats = self.model.tables.d
        if hasattr(self.backend, "exchange"):
            self.backend.exchange = self.world > 1
        if self.world > 1:      # identical replicas: rank 0's init wins
            t = self.model.tables
            for buf in (t.R, t.C, t.br, t.bc):
                self.dist.broadcast(buf, src=0)
        self.row_sharded = bool(params.get("row_sharded")) and self.world > 1
        # --shard-cols: the col table sharded by id % ranks as well (trainer.stepper.ShardedStepper)
        self.both_sharded = self.row_sharded and bool(params.get("shard_cols"))
        if params.get("shard_cols") and not params.get("row_sharded"):
            raise ValueError("--shard-cols goes with --row-sharded")
        if self.row_sharded:
            if self.optimizer_name != "Adagrad":
                raise ValueError("--row-sharded is implemented for Adagrad (Keras' Adam has no sparse form)")
            # keep this rank's rows (u % world == rank) of the row side; the col side stays replicated, or is cut the same way
            from trainer.hip_api import DeviceTables
            from trainer.stepper import owned_rows
            whole = self.model.tables
            mine = owned_rows(whole.V, self.world, self.rank)
            shard = DeviceTables(whole.V, whole.d_model, whole.optimizer, device=self.device, seed=0,
                                 V_row=mine, V_col=mine if self.both_sharded else None)
            shard.load_whole_state_dict(whole.state_dict(), self.world, self.rank)
            self.model.tables = shard
            if hasattr(self.backend, "shard_rows"):
                self.backend.shard_rows = shard.V_row        # the routed stream carries shard-local row ids
        self.ckpt = CheckpointManager(params["job_dir"], params.get("save_checkpoints_secs", 300.0),
                                      params.get("keep_checkpoint_max", 5))
        self.ckpt.restore(self.model.tables, shard=(self.world, self.rank) if self.row_sharded else None)
        self._stream = None
        self._events = {}
        self.reshuffling = params.get("epoch_shuffle", "full") == "full"
        if params.get("epoch_shuffle", "full") not in ("static", "full"):
            raise ValueError("--epoch-shuffle must be static or full, got %r" % (params["epoch_shuffle"],))
        self.logistic = params.get("head", "regression") == "logistic"

    # ---- input_fn
    def stream(self) -> NonzeroStream:
        if self._stream is None:
            a = self.params["input_fn_args"]
            row, col, weight, target = a["select_columns"]
            coo = load_interaction_csv(a["file_pattern"], self.params["vocab_txt"], row, col, weight, target,
                                       cache_dir=self.params["job_dir"] if self.rank == 0 else None)
            self._stream = NonzeroStream(coo, a["batch_size"], self.vocab_size, self.backend, self.device,
                                         rank=self.rank, world=self.world, seed=self.params.get("seed"),
                                         chunk_cap=self.params.get("chunk_cap", 0),
                                         static_plans=not self.reshuffling,
                                         route=self.dist if self.row_sharded else None,
                                         cols_by_owner=self.world if self.both_sharded else 0)
        return self._stream

    def _log(self, name, record, histograms=None):
        """One line of <job_dir>/<name> (JSON) plus the same scalars as a TensorBoard event in that directory,
        where the reference's Estimator leaves its summaries (job_dir for training, job_dir/eval for eval)."""
        if self.rank != 0:
            return
        path = os.path.join(self.params["job_dir"], name)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps(record) + "\n")
        logdir = os.path.dirname(path)
        if logdir not in self._events:
            from trainer.event_writer import EventWriter
            self._events[logdir] = EventWriter(logdir)
        scalars = {k: v for k, v in record.items() if k != "global_step" and isinstance(v, (int, float))}
        if "steps_per_sec" in scalars:
            scalars["global_step/sec"] = scalars.pop("steps_per_sec")          # the Estimator's tag
        self._events[logdir].scalars(record["global_step"], scalars, histograms)

    # ---- TRAIN
    def train(self, max_steps: int):
        """Runs until global_step == max_steps (absolute, train_utils.py:39-40)."""
        p, tables = self.params, self.model.tables
        step = tables.global_step
        if step >= max_steps:
            logger.info("global_step %d >= max_steps %d: nothing to train", step, max_steps)
            return
        stream = self.stream()
        hyper_kwargs = dict(l2_reg=p["l2_reg"], reg_mult=p.get("reg_multiplicity", 2.0),
                            learning_rate=p["learning_rate"], optimizer=self.optimizer_name)
        if p.get("step_form"):
            hyper_kwargs["step_form"] = int(p["step_form"])
            if int(p["step_form"]) == 4 and self.world == 1:
                tables.enable_twin()        # the form needs the second copy of the row table
            if int(p["step_form"]) == 5 and self.world == 1:
                tables.enable_tags()        # the tagged form needs both tables twinned and step-tagged
        if self.logistic:       # logistic_matrix_factorisation.py:50-54: the stream's (w, y) are (pos, neg) weights
            hyper_kwargs.update(head=1, neg_factor=p.get("neg_factor", 1.0))
        log_every = max(1, int(p.get("log_every", 100)))
        # a reshuffled stream has no resident plans: the exchange is agreed from the batch size (prepare(batch_size=...))
        plans_or_size = dict(batch_size=p["batch_size"]) if self.reshuffling else dict(plans=stream.plans)
        if self.both_sharded:
            if not self.reshuffling:
                raise ValueError("--shard-cols runs with --epoch-shuffle full (a batch's fetch lists are prepared per epoch)")
            from trainer.stepper import ShardedStepper
            stepper = ShardedStepper(self.backend, tables, hyper_kwargs, p["batch_size"], self.world, self.rank, self.dist)
        elif self.row_sharded:
            from trainer.stepper import RowShardedStepper
            stepper = RowShardedStepper(self.backend, tables, hyper_kwargs, p["batch_size"], self.world, self.dist,
                                        exchange=p.get("exchange", "auto"))
            stepper.prepare(**plans_or_size)        # collective: the ranks agree on the col-side exchange
        elif self.world > 1 or not self.reshuffling or getattr(self.backend, "hip", None) is None:
            # (a test backend without the library steps through a Stepper also on one rank)
            stepper = Stepper(self.backend, tables, hyper_kwargs, p["batch_size"], self.world, self.dist,
                              exchange=p.get("exchange", "auto"))
            if self.world > 1:
                stepper.prepare(**plans_or_size)    # collective: dense all-reduce or touched-rows all-gather
        else:
            stepper = None
        # hipGraphs: always on one GPU; with several ranks a captured step holds RCCL collectives, which has run through RCCL
        # with ONE rank only so far (one-GPU boxes): opt-in there (--multi-rank-graphs) until a node has shown replay == eager
        graphs = not p.get("no_graphs", False) and (self.world == 1 or bool(p.get("multi_rank_graphs", False)))
        if stepper is not None and not self.reshuffling and graphs:
            stepper.enable_graphs()                 # a resident batch's step (kernels + RCCL collectives) replayed from a hipGraph
        if self.reshuffling:
            # every rank re-permutes ITS shard of the stream each epoch (reference data_utils.py:12-21 reshuffles every
            # epoch); on one GPU the runner steps by itself, on several through the stepper above
            from trainer.stepper import ReshufflingRunner
            stepper = ReshufflingRunner(getattr(self.backend, "hip", None), stream, tables,
                                        self.backend.make_hyper(batch_size=p["batch_size"] * self.world, **hyper_kwargs),
                                        chunk_cap=p.get("chunk_cap", 0), burst=64, segment=p.get("index_segment", 0),
                                        stepper=stepper, graphs=None if graphs else False)
        fresh = self.ckpt.latest() is None
        if self.world > 1:                  # saving may be collective (row-sharded): rank 0's view of job_dir decides
            flag = torch.tensor([1 if fresh else 0], device=self.device)
            self.dist.broadcast(flag, src=0)
            fresh = bool(flag.item())
        if fresh:
            self._save_checkpoint()         # Estimator saves at step 0 too
        t_last, s_last = time.perf_counter(), step
        try:
            self._train_loop(stepper, stream, step, max_steps, log_every, t_last, s_last)
        finally:
            # captured RCCL collectives must be gone before the process group is (RCCL's teardown waits for them) — also
            # when the loop ends in an exception (the NaN stop)
            if self.device.type == "cuda":
                torch.cuda.synchronize()
            for obj in (stepper, getattr(stepper, "stepper", None)):
                if hasattr(obj, "release_graphs"):
                    obj.release_graphs()

    def _train_loop(self, stepper, stream, step, max_steps, log_every, t_last, s_last):
        """Steps up to the next logging point go out in one call (launch loop in C or replayed graphs, no Python per step).  A
        logging point does not drain the GPU: the loss scalars and the bias vectors are copied to pinned host memory behind
        the point's last step, the next run of steps is issued, and only then the host waits for THAT copy, checks the loss
        (NanTensorHook) and hands the record to the writer thread (histograms, event file, JSON line)."""
        p = self.params
        self._clock = (t_last, s_last)
        pending = None
        try:
            while step < max_steps:
                upto = min(max_steps, (step // log_every + 1) * log_every)
                if self.reshuffling:
                    # a call ends early at the epoch's end and at a burst's (the runner replays graphs of 2^k steps): the loop
                    # simply asks again
                    step += stepper.run(upto - step)
                else:
                    stepper.step_many([stream.next_plan() for _ in range(upto - step)])
                    step = upto
                if pending is not None:
                    self._finish_log_point(pending)
                    pending = None
                if step % log_every == 0 or step == max_steps:
                    pending = self._start_log_point(stepper, step)
                due = self.ckpt.due() or step == max_steps
                if self.world > 1:               # the eval pass below is collective: rank 0's clock decides for all
                    flag = torch.tensor([1 if due else 0], device=self.device)
                    self.dist.broadcast(flag, src=0)
                    due = bool(flag.item())
                if due:
                    if pending is not None:
                        self._finish_log_point(pending)
                        pending = None
                    self._drain_logs()
                    self._save_checkpoint()
                    if not p.get("skip_eval"):
                        self.evaluate()
            if pending is not None:
                self._finish_log_point(pending)
        except BaseException:
            # the loop is unwinding with its own error (the NaN stop, a GloveHipError): stop the writer, but a writer error
            # must not replace the cause
            try:
                self._drain_logs(stop=True)
            except Exception as log_exc:
                logger.error("log writer failed while the training loop was stopping: %r", log_exc)
            raise
        self._drain_logs(stop=True)

    # ---- logging points
    HOST_BIASES_MAX = 1 << 18       # bias vectors up to this length are logged from a host copy (one 1 MB copy at most)

    def _start_log_point(self, stepper, step):
        """Behind the last step issued: the loss scalars, the global bias and (small vocabularies) the bias vectors on their
        way to pinned host memory; an event marks the copies' end."""
        tables = self.model.tables
        br, bc = tables.br, tables.bc                      # (brings twinned tables home, in stream order)
        src = {"loss": stepper.loss_out, "scalars": tables.scalars}
        if max(br.numel(), bc.numel()) <= self.HOST_BIASES_MAX:
            src.update(br=br, bc=bc)
        if self.device.type != "cuda":
            return {"step": step, "host": {k: v.clone() for k, v in src.items()}, "event": None, "dev": None}
        bufs = getattr(self, "_log_bufs", None)
        if bufs is None or any(bufs[k].shape != v.shape for k, v in src.items()) or set(bufs) != set(src):
            bufs = self._log_bufs = {k: torch.empty(v.shape, dtype=v.dtype).pin_memory() for k, v in src.items()}
        for k, v in src.items():
            bufs[k].copy_(v, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        # big vocabularies: statistics and histograms on the device, at the finish — of copies taken HERE, in stream order behind
        # the point's last step (the steps issued meanwhile flip rows of a twinned table to their other copy and move on: a view
        # read at the finish would mix old and new rows under this point's global_step); 8 B per id
        dev = None if "br" in src else (br.clone(), bc.clone())
        return {"step": step, "host": bufs, "event": ev, "dev": dev}

    def _finish_log_point(self, pt):
        if pt["event"] is not None:
            pt["event"].synchronize()                      # the copies of THIS point; later steps keep running
        host, step = pt["host"], pt["step"]
        loss, L, reg, _ = host["loss"].tolist()
        if not math.isfinite(loss):
            raise FloatingPointError("loss is %r at global_step %d" % (loss, step))   # NanTensorHook
        now = time.perf_counter()
        t_last, s_last = self._clock
        rate = (step - s_last) / max(now - t_last, 1e-9)
        self._clock = (now, step)
        if pt["dev"] is not None:
            biases = pt["dev"]                             # (the device copies taken at the point)
        else:
            biases = (host["br"].clone(), host["bc"].clone())
        rec = {"loss": loss, "weighted_mse": L, "regularization_loss": reg, "global_step": step, "steps_per_sec": rate,
               "nonzeros_per_sec": rate * self.params["batch_size"] * self.world}
        rec.update(summary_values(self.model, biases, global_bias=float(host["scalars"][0])))
        logger.info("global_step %d: loss = %.6f (%.1f steps/s)", step, loss, rate)
        # a row-sharded run logs the histogram of rank 0's row-bias shard (every world-th row)
        if self.rank == 0:
            if pt["dev"] is not None:
                self._log("train_log.jsonl", rec, summary_histograms(self.model, biases))
            else:
                self._log_later("train_log.jsonl", rec, biases)

    def _log_later(self, name, rec, biases):
        """The record's histograms, its event and its JSON line, on the writer thread (started at the first record)."""
        import queue
        import threading
        if getattr(self, "_log_queue", None) is None:
            self._log_queue, self._log_error = queue.Queue(), None

            def work():
                while True:
                    item = self._log_queue.get()
                    try:
                        if item is None:
                            return
                        if self._log_error is None:
                            self._log(item[0], item[1], summary_histograms(self.model, item[2]))
                    except BaseException as exc:           # surfaces at the next drain
                        self._log_error = exc
                    finally:
                        self._log_queue.task_done()
            self._log_thread = threading.Thread(target=work, name="glove-log-writer", daemon=True)
            self._log_thread.start()
        self._log_queue.put((name, rec, biases))

    def _drain_logs(self, stop=False):
        """Everything handed to the writer thread is on disk when this returns (before a checkpoint, an eval pass, the end);
        `stop`: the thread ends as well (it holds the estimator — tables, stream — alive while it runs)."""
        q = getattr(self, "_log_queue", None)
        if q is None:
            return
        q.join()
        if stop:
            q.put(None)
            self._log_thread.join()
            self._log_queue = self._log_thread = None
        if self._log_error is not None:
            err, self._log_error = self._log_error, None
            raise err

    def _save_checkpoint(self):
        """Rank 0 writes; a row-sharded run first gathers the whole model (collective: every rank calls this)."""
        tables = self.model.tables
        state = tables.gathered_state_dict(self.dist, self.world) if self.row_sharded else None
        if self.rank == 0:
            self.ckpt.save(tables, state=state)

    # ---- EVAL
    def evaluate(self) -> dict:
        """One pass over the CSV (the reference evaluates on the training file, estimator.py:86-87).
        RegressionHead metrics: average_loss = sum w l / sum w, loss = mean over batches of the
        batch-mean weighted loss, prediction/mean, label/mean."""
        tables, stream = self.model.tables, self.stream()
        if self.both_sharded:
            # the eval pass reads the col row of every pair: the whole col side is gathered for its duration (collective;
            # V x d x 4 B per rank, every few minutes — the steps themselves never hold it)
            from trainer.hip_api import TablesView
            # (rank after rank: the stream's col ids are numbered owner-major, NonzeroStream(cols_by_owner=))
            C_all = tables.gather_by_owner(tables.C, self.dist, self.world).contiguous()
            bc_all = tables.gather_by_owner(tables.bc, self.dist, self.world).contiguous()
            tables = TablesView(tables, C_all.shape[0], tables.V_row, keep=(C_all, bc_all), C=C_all, bc=bc_all)
        k = 6 if self.logistic else 4
        buf = torch.zeros(k + 1, dtype=torch.float64, device=self.device)       # the metric sums + this rank's nonzero count
        sums = buf[:k]
        buf[k] = stream.nnz
        for row, col, w, y in stream.eval_batches():
            if self.logistic:
                self.backend.eval_sums_logistic(row, col, w, y, tables, sums)
            else:
                self.backend.eval_sums(row, col, w, y, tables, sums)
        if self.world > 1:
            self.dist.all_reduce(buf)
        s = sums.tolist()
        n = int(buf[k].item())                  # every nonzero of the file, whatever the ranks' shares
        if self.logistic:
            # BinaryClassHead metrics of the two heads ("pos": label 1, "neg": label 0) and MultiHead's merged loss
            nf = self.params.get("neg_factor", 1.0)
            rec = {"global_step": self.model.tables.global_step, "loss": (s[0] + nf * s[2]) / n,
                   "average_loss/pos": s[0] / max(s[1], 1e-300), "average_loss/neg": s[2] / max(s[3], 1e-300),
                   "prediction/mean/pos": s[4] / max(s[1], 1e-300), "prediction/mean/neg": s[5] / max(s[3], 1e-300),
                   "label/mean/pos": 1.0, "label/mean/neg": 0.0}
            rec["average_loss"] = rec["average_loss/pos"] + nf * rec["average_loss/neg"]
        else:
            rec = {"global_step": self.model.tables.global_step, "average_loss": s[0] / s[1], "loss": s[0] / n,
                   "prediction/mean": s[2] / s[1], "label/mean": s[3] / s[1]}
        self._log(os.path.join("eval", "eval_log.jsonl"), rec)
        logger.info("eval at global_step %d: average_loss = %.6f", rec["global_step"], rec["average_loss"])
        return rec

    # ---- PREDICT
    def predict(self, batch=256):
        """Every vocab line as the query token (estimator.py:59-76); yields one dict per token."""
        id_string = get_id_string_table(self.params["vocab_txt"])
        k = min(self.params.get("top_k", 20), self.vocab_size)
        for s in range(0, self.vocab_size, batch):
            ids = torch.arange(s, min(s + batch, self.vocab_size), dtype=torch.int32)
            out = get_predictions(self.backend, self.model, ids, id_string, k)
            for i in range(len(ids)):
                yield {"input_string": out["input_string"][i], "input_embedding": out["input_embedding"][i].numpy(),
                       "top_k_similarity": out["top_k_similarity"][i].numpy(),
                       "top_k_string": out["top_k_string"][i]}


def estimator_predict(params):
    """Reference estimator.py:72-76."""
    return Estimator(params).predict()


def multi_rank_queues():
    """A HIP process has four hardware queues by default and runs a queue's packets in order; a multi-rank process has more
    streams than that (compute, the epoch deals, the prepares, the push all-to-all, RCCL's and torch's own), and two that
    share a queue take turns whatever their events say (rocprofv3 queue ids: profiles/r05_exp_sharded_prepare_hw_queues.txt).
    Eight queues, unless the caller has chosen; read by the HIP runtime when it starts, so call before the first GPU call."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main(argv=None, adapt_params=None):
    """`adapt_params(params)`: hook for the sibling entry points that share this loop
    (trainer.logistic_matrix_factorisation)."""
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(name)s: %(message)s")
    world, rank, _ = _dist_env()
    if world > 1:
        multi_rank_queues()
    params = parse_args(argv) if rank == 0 or world == 1 else None
    if params is not None and adapt_params is not None:
        adapt_params(params)
    if world > 1:
        # every rank must agree on the (time-stamped) job_dir: rank 0 decides
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
        box = [params]
        dist.broadcast_object_list(box, src=0)
        params = box[0]
        estimator = Estimator(params, dist=dist)
    else:
        estimator = Estimator(params)
    try:
        estimator.train(params["train_steps"])
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except KeyboardInterrupt:
        pass
