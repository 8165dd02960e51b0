"""End-to-end trainer on the GPU: CLI, job_dir layout, resume, export, and the trajectory parity
gates of SURVEY.md §8d (same batches + same init for 1,024 steps at bs = 1,024: 100-step-smoothed
loss within 1 % of the CPU restatement, top-20 neighbour overlap >= 0.9 on probe tokens)."""
import json
import os
from pathlib import Path

import numpy as np
import pytest
import torch

import glove_ref as ref
from helpers import free_port

pytestmark = pytest.mark.gpu
GOLDEN = Path(__file__).resolve().parent / "golden"


def test_cli_train_resume_export(hip, tmp_path):
    from trainer import estimator, export_embeddings
    csv, vocab = GOLDEN / "text8_cov90_ctx5_interaction.csv", GOLDEN / "text8_cov90_ctx5_vocab.txt"
    job = tmp_path / "job"
    argv = ["--train-csv", str(csv), "--vocab-txt", str(vocab), "--job-dir", str(job), "--disable-datetime-path",
            "--embedding-size", "50", "--optimizer", "Adagrad", "--learning-rate", "0.05", "--batch-size", "64",
            "--train-steps", "60", "--log-every", "20", "--seed", "7"]      # 50: not a multiple of 4 (padded rows)
    estimator.main(argv)
    assert (job / "params.json").exists() and (job / vocab.name).exists()
    assert (job / "checkpoint").read_text().startswith('model_checkpoint_path: "model.ckpt-60"')
    assert (job / "model.ckpt-0.pt").exists() and (job / "model.ckpt-60.pt").exists()
    log = [json.loads(l) for l in (job / "train_log.jsonl").read_text().splitlines()]
    assert [r["global_step"] for r in log] == [20, 40, 60]
    assert all(np.isfinite(r["loss"]) for r in log)         # (a logged loss is ONE batch of 64 pairs: too noisy to order)
    ev = [json.loads(l) for l in (job / "eval" / "eval_log.jsonl").read_text().splitlines()]
    assert ev[-1]["global_step"] == 60 and ev[-1]["average_loss"] > 0
    # the same scalars as TensorBoard event files, where the reference's Estimator leaves them
    from trainer.event_writer import read_events
    (train_events,), (eval_events,) = list(job.glob("events.out.tfevents.*")), list((job / "eval").glob("events.out.tfevents.*"))
    recs = [(step, sc) for _, step, sc in read_events(train_events) if sc]
    assert [step for step, _ in recs] == [20, 40, 60]
    assert abs(recs[-1][1]["loss"] - log[-1]["loss"]) < 1e-6 * abs(log[-1]["loss"]) and "global_step/sec" in recs[0][1]
    assert "mf/global_bias" in recs[0][1] and [sc["average_loss"] > 0 for _, _, sc in read_events(eval_events) if sc]
    # resume: max_steps is absolute
    estimator.main(argv[:-6] + ["--train-steps", "100", "--log-every", "20", "--seed", "7"])
    log = [json.loads(l) for l in (job / "train_log.jsonl").read_text().splitlines()]
    assert log[-1]["global_step"] == 100 and (job / "model.ckpt-100.pt").exists()
    ev2 = [json.loads(l) for l in (job / "eval" / "eval_log.jsonl").read_text().splitlines()]
    assert ev2[-1]["global_step"] == 100 and ev2[-1]["average_loss"] < ev[-1]["average_loss"]       # the whole file's weighted loss falls
    estimator.main(argv[:-6] + ["--train-steps", "100", "--log-every", "20"])      # nothing left to do
    assert len((job / "train_log.jsonl").read_text().splitlines()) == len(log)
    # export (PREDICT mode over the vocabulary)
    out = tmp_path / "embeddings.json"
    export_embeddings.main(job_dir=str(job), embeddings_json=str(out))
    emb = json.loads(out.read_text())
    tokens = vocab.read_text().split("\n")
    assert set(emb) == set(tokens) - {"<UNK>"} and "nan" in emb
    assert emb["the"]["item_id"] == "the" and len(emb["the"]["item_embedding"]) == 50
    # predictions: the query token is its own nearest neighbour (reference README.md:284)
    params = json.loads((job / "params.json").read_text())
    first = next(iter(estimator.Estimator(params).predict()))
    assert first["top_k_string"][0] == first["input_string"] and abs(first["top_k_similarity"][0] - 1) < 1e-5
    assert len(first["top_k_string"]) == 20


@pytest.mark.parametrize("optimizer,lr", [("Adam", 0.001), ("Adagrad", 0.05)])
def test_loss_curve_and_neighbours_match_cpu_restatement(hip, optimizer, lr):
    from helpers import tables_from_oracle
    from trainer import synthetic
    from trainer.data_utils import NonzeroStream
    from trainer.hip_api import DeviceTables
    from trainer.stepper import HipBackend, Stepper
    V, d, B, steps = 600, 64, 1024, 1024
    row, col, w, y = synthetic.text8_shaped(V=V, n_tokens=400_000, seed=5)
    coo = dict(row=row.numpy(), col=col.numpy(), w=w.numpy(), y=y.numpy())
    backend = HipBackend("cuda:0")
    stream = NonzeroStream(coo, B, V, backend, "cuda:0", seed=11)
    assert stream.batches_per_epoch >= 8
    t = ref.Tables(V, d, optimizer, dtype=np.float32, seed=1).astype(np.float64)
    dt = tables_from_oracle(t, DeviceTables)
    hp = ref.Hyper(learning_rate=lr)
    stepper = Stepper(backend, dt, dict(l2_reg=hp.l2_reg, reg_mult=hp.reg_mult, learning_rate=lr), B)
    gpu_loss, cpu_loss = [], []
    host = {k: getattr(stream, k).cpu().numpy() for k in ("row", "col", "w", "y")}
    for s in range(steps):
        stepper.step(stream.next_plan())
        gpu_loss.append(stepper.loss_out[0].item())
        sl = slice(stream.last_batch * B, (stream.last_batch + 1) * B)
        cpu_loss.append(ref.train_step(t, host["row"][sl], host["col"][sl], host["w"][sl], host["y"][sl], hp)[0])
    gpu_loss, cpu_loss = np.array(gpu_loss), np.array(cpu_loss)
    smooth = lambda x: np.convolve(x, np.ones(100) / 100, mode="valid")
    np.testing.assert_allclose(smooth(gpu_loss), smooth(cpu_loss), rtol=0.01)
    assert cpu_loss[-100:].mean() < cpu_loss[:100].mean()
    assert dt.global_step == steps
    # top-20 neighbours of probe tokens by cosine over ROW embeddings
    probes = np.array([1, 2, 5, 17, 100], np.int32)
    _, idx = backend.topk_cosine(dt.R, torch.from_numpy(probes).cuda(), 20)
    _, want = ref.cosine_topk(t.R, probes, 20)
    overlap = [len(set(a) & set(b)) / 20 for a, b in zip(idx.cpu().numpy().tolist(), want.tolist())]
    assert min(overlap) >= 0.9, overlap


def test_data_parallel_form_on_one_gpu_matches_sparse_step(hip):
    """The N > 1 code path (dense gradient buffer -> nccl all-reduce -> dense apply) with world size 1:
    RCCL is really called, and the result is bit-identical to the single-GPU sparse step."""
    import os
    import torch.distributed as dist
    from helpers import make_batch, tables_from_oracle, to_dev
    from trainer.hip_api import DeviceTables
    from trainer.stepper import HipBackend, Stepper
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    try:
        B, V, d = 4096, 300, 64
        backend = HipBackend("cuda:0")
        t = ref.Tables(V, d, "Adagrad", dtype=np.float32, seed=2).astype(np.float64)
        a, b = tables_from_oracle(t, DeviceTables), tables_from_oracle(t, DeviceTables)
        kw = dict(l2_reg=0.01, reg_mult=2.0, learning_rate=0.05)
        sparse = Stepper(backend, a, kw, B)
        dp = Stepper(backend, b, kw, B, world=1, dist=dist)
        dp.dense, dp.world, dp.G = True, 2, backend.dense_grad_buffer(b)      # force the collective path
        dp.hyper = backend.make_hyper(batch_size=B, **kw)
        for s in range(5):
            plan = backend.build_plan(*to_dev(*make_batch(s, B, V)), V, 0)
            sparse.step(plan)
            dp.step(plan)
        for n in ("R", "C", "br", "bc"):
            assert torch.equal(getattr(a, n), getattr(b, n)), n
        assert torch.equal(a.scalars, b.scalars) and a.global_step == b.global_step == 5
    finally:
        dist.destroy_process_group()


def test_exchange_forms_through_rccl_with_one_rank(hip):
    """Every collective of the multi-GPU forms issued through RCCL on this one GPU (a process group of one rank,
    `collectives=True`): the all-gather of packed lists (data parallel, and row-sharded started asynchronously), the
    all-to-alls with split sizes and the asynchronous push of the fully sharded form, the 4-float all-reduces.  Each
    form must equal the plain single-GPU step on the same batches."""
    import os
    import torch.distributed as dist
    from helpers import make_batch, tables_from_oracle, to_dev
    from trainer.hip_api import DeviceTables
    from trainer.stepper import HipBackend, RowShardedStepper, ShardedStepper, Stepper
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    try:
        B, V, d, steps = 6000, 700, 64, 4
        backend = HipBackend("cuda:0")
        t = ref.Tables(V, d, "Adagrad", dtype=np.float32, seed=4).astype(np.float64)
        kw = dict(l2_reg=0.01, reg_mult=2.0, learning_rate=0.05)
        batches = [to_dev(*make_batch(40 + s, B, V)) for s in range(steps)]
        plans = [backend.build_plan(*bt, V, 0).compact(hip.lib, d) for bt in batches]
        plain_t = tables_from_oracle(t, DeviceTables)
        plain = Stepper(backend, plain_t, kw, B)
        runs = {}
        tabs = tables_from_oracle(t, DeviceTables)
        st = Stepper(backend, tabs, kw, B, world=1, dist=dist, exchange="rows", collectives=True)
        st.prepare(plans)
        assert st.rows and "all_gather" in dict(st.phases())
        runs["data parallel, rows"] = (tabs, st, plans)
        tabs = tables_from_oracle(t, DeviceTables)
        st = RowShardedStepper(backend, tabs, kw, B, 1, dist, exchange="rows", collectives=True)
        st.prepare(plans)
        assert st.rows and "loss_tail" in dict(st.phases())
        runs["row-sharded, rows"] = (tabs, st, plans)
        tabs = tables_from_oracle(t, DeviceTables)
        st = RowShardedStepper(backend, tabs, kw, B, 1, dist, exchange="dense", collectives=True)
        st.prepare(plans)
        assert not st.rows and "all_reduce" in dict(st.phases())
        runs["row-sharded, dense"] = (tabs, st, plans)
        tabs = tables_from_oracle(t, DeviceTables)
        st = ShardedStepper(backend, tabs, kw, B, 1, 0, dist, collectives=True, exercise_exchange=True)
        handles = [st.add_batch(*bt) for bt in batches]
        runs["both tables sharded"] = (tabs, st, handles)
        for s in range(steps):
            plain.step(plans[s])
            for tabs, st, items in runs.values():
                st.step(items[s])
        want = plain.read_loss()
        for name, (tabs, st, _) in runs.items():
            for n in ("R", "C", "br", "bc"):
                torch.testing.assert_close(getattr(tabs, n), getattr(plain_t, n), rtol=2e-5, atol=2e-6, msg=lambda m: name + " " + n + ": " + m)
            torch.testing.assert_close(tabs.scalars, plain_t.scalars, rtol=2e-5, atol=2e-6)
            assert tabs.global_step == plain_t.global_step == steps, name
            got = st.read_loss()
            assert abs(got["loss"] - want["loss"]) <= 2e-5 * abs(want["loss"]), (name, got, want)
    finally:
        dist.destroy_process_group()


def test_multi_rank_steps_replayed_from_hipgraphs_with_their_rccl_collectives(hip):
    """A multi-rank step captured ONCE as a hipGraph — its kernels AND its collectives, issued through RCCL on this one GPU
    (a process group of one rank, `collectives=True`): the dense all-reduce, the all-gather of packed lists, the
    asynchronous all-gather / all-to-all that overlap the row side (captured on their side stream), the 4-float
    all-reduces — and replayed == the same steps launched eagerly from Python, bit for bit; likewise the reshuffling
    runner (index builds + steps + collectives in one graph per burst) against its eager form and the single-GPU runner."""
    import subprocess
    import sys
    # (in a process of its own: tests/rccl_graph_case.py says why)
    proc = subprocess.run([sys.executable, str(Path(__file__).resolve().parent / "rccl_graph_case.py")], capture_output=True,
                          text=True, timeout=280)
    assert proc.returncode == 0 and "rccl graph case ok" in proc.stdout, proc.stdout[-2000:] + proc.stderr[-4000:]


def test_whole_pipeline_recovers_planted_topics(hip, tmp_path):
    """corpus -> trainer.text8 (GPU co-occurrence) -> trainer.estimator (CLI) -> PREDICT.  The corpus is built from
    8 topics of 30 words; sentences stay inside one topic, so a word's nearest neighbours by cosine over the
    trained ROW embeddings must be words of its own topic."""
    from trainer import estimator, text8
    rng = np.random.default_rng(0)
    topics, words_per = 8, 30
    vocab = [["t%dw%02d" % (t, i) for i in range(words_per)] for t in range(topics)]
    tokens = []
    for _ in range(4000):
        t = rng.integers(topics)
        tokens.extend(rng.choice(vocab[t], size=rng.integers(30, 60)))        # long sentences: few cross-topic windows
    data_dir = tmp_path / "data"
    data_dir.mkdir()
    (data_dir / "text8").write_text(" ".join(tokens))
    text8.main(url="", dest=str(data_dir), vocab_size=None, coverage=0.999, context_size=5, seed=1)
    assert (data_dir / "interaction.csv").exists() and (data_dir / "vocab.txt").exists()
    job = tmp_path / "job"
    estimator.main(["--train-csv", str(data_dir / "interaction.csv"), "--vocab-txt", str(data_dir / "vocab.txt"),
                    "--job-dir", str(job), "--disable-datetime-path", "--embedding-size", "32", "--optimizer", "Adagrad",
                    "--learning-rate", "0.1", "--batch-size", "2048", "--train-steps", "3000", "--log-every", "50",
                    "--seed", "3", "--skip-eval"])
    log = [json.loads(l) for l in (job / "train_log.jsonl").read_text().splitlines()]
    assert log[-1]["global_step"] == 3000 and log[-1]["loss"] < 0.8 * log[0]["loss"]      # step 50 vs step 3000
    params = json.loads((job / "params.json").read_text())
    params["top_k"] = 11
    same, total = 0, 0
    for pred in estimator.Estimator(params).predict():
        word = pred["input_string"]
        if word == "<UNK>":
            continue
        assert pred["top_k_string"][0] == word
        neighbours = [n for n in pred["top_k_string"][1:] if n != "<UNK>"]
        same += sum(n[:2] == word[:2] for n in neighbours)
        total += len(neighbours)
    assert total >= topics * words_per * 9 and same / total > 0.6, (same, total)       # chance level: 0.12; 0.76 observed


def test_cli_on_a_twinned_row_table_equals_the_three_launch_form(hip, tmp_path):
    """--step-form 4 end to end: training on a twinned row table (new rows written beside the old ones, versions
    flipped per step), with logging, eval, checkpoints, resume and export in between — every reader first brings the
    table back to its plain form — gives the tables of --step-form 3, bit for bit (the forms share their arithmetic)."""
    from trainer import estimator, export_embeddings
    csv, vocab = GOLDEN / "text8_cov90_ctx5_interaction.csv", GOLDEN / "text8_cov90_ctx5_vocab.txt"
    state = {}
    for form in (3, 4):
        job = tmp_path / ("job%d" % form)
        argv = ["--train-csv", str(csv), "--vocab-txt", str(vocab), "--job-dir", str(job), "--disable-datetime-path",
                "--embedding-size", "50", "--optimizer", "Adagrad", "--learning-rate", "0.05", "--batch-size", "64",
                "--step-form", str(form), "--chunk-cap", "2", "--train-steps", "45", "--log-every", "15", "--seed", "7"]
        estimator.main(argv)
        estimator.main(argv[:-6] + ["--train-steps", "90", "--log-every", "15", "--seed", "7"])      # resume from the checkpoint
        state[form] = torch.load(job / "model.ckpt-90.pt")
        out = tmp_path / ("emb%d.json" % form)
        export_embeddings.main(job_dir=str(job), embeddings_json=str(out))
        state[str(form)] = out.read_text()
        log = [json.loads(l) for l in (job / "train_log.jsonl").read_text().splitlines()]
        assert log[-1]["global_step"] == 90 and log[-1]["loss"] < log[0]["loss"]
    tensors = 0
    for k, v in state[3]["tables"].items():
        if torch.is_tensor(v):
            assert torch.equal(v, state[4]["tables"][k]), k
            tensors += 1
    assert tensors >= 9 and state["3"] == state["4"]


@pytest.mark.parametrize("shuffle", ["full", "static"])
def test_cli_with_adam_in_one_launch_equals_the_two_launch_form(hip, tmp_path, shuffle):
    """The reference's default optimizer end to end on twinned tables (AUTO: Keras-legacy Adam in one launch per step, the
    tables flipping as a whole every step), with logging points, eval, checkpoints, resume and export in between — every
    reader first brings the tables home — against --step-form 1 (passes + the fused apply / decay kernel on plain tables):
    same checkpoints within fp32 rounding of a few multi-chunk ids' sums, same exported neighbours."""
    from trainer import estimator, export_embeddings
    csv, vocab = GOLDEN / "text8_cov90_ctx5_interaction.csv", GOLDEN / "text8_cov90_ctx5_vocab.txt"
    state = {}
    for form in (0, 1):
        job = tmp_path / ("job%d" % form)
        argv = ["--train-csv", str(csv), "--vocab-txt", str(vocab), "--job-dir", str(job), "--disable-datetime-path",
                "--embedding-size", "50", "--optimizer", "Adam", "--learning-rate", "0.01", "--batch-size", "16",
                "--epoch-shuffle", shuffle, "--step-form", str(form), "--train-steps", "45", "--log-every", "15", "--seed", "7"]
        estimator.main(argv)
        params = json.loads((job / "params.json").read_text())
        params["train_steps"] = 91
        est = estimator.Estimator(params)                      # resume from the checkpoint, in process: an odd number of steps more
        est.train(91)
        assert (est.model.tables.R_tag is not None) == (form == 0)              # the one-launch form really ran
        if form == 0:
            assert est.model.tables._twin_dirty
        state[form] = torch.load(job / "model.ckpt-91.pt", weights_only=False)
        assert float(state[form]["tables"]["scalars"][3]) == 0.0                # checkpoints hold the plain form
        out = tmp_path / ("emb%d.json" % form)
        export_embeddings.main(job_dir=str(job), embeddings_json=str(out))
        state[str(form)] = json.loads(out.read_text())
        ev = [json.loads(l) for l in (job / "eval" / "eval_log.jsonl").read_text().splitlines()]
        assert ev[-1]["global_step"] == 91 and ev[-1]["average_loss"] < ev[0]["average_loss"]
    tensors = 0
    for k, v in state[0]["tables"].items():
        if torch.is_tensor(v) and v.is_floating_point():
            np.testing.assert_allclose(v.numpy(), state[1]["tables"][k].numpy(), rtol=2e-4, atol=2e-6, err_msg=k)
            tensors += 1
    assert tensors >= 13
    for token, rec in state["0"].items():
        np.testing.assert_allclose(rec["item_embedding"], state["1"][token]["item_embedding"], rtol=2e-4, atol=2e-6)


def _two_rank_trainer_sorting_prepare(rank, port, argv, out_dir):
    """As _two_rank_trainer, but the sharded stepper prepares every batch the general way (torch.unique, owner order, the
    sorting index builder) instead of reading it off the dealt order: the reference for add_batch_dealt with two owners."""
    import sys
    here = Path(__file__).resolve().parent
    for p in (here.parent, here):
        sys.path.insert(0, str(p))
    from trainer.stepper import ShardedStepper
    # (the runner prepares in two halves: the first only remembers the batch, the second does everything the general way)
    ShardedStepper.add_batch_dealt_begin = lambda self, rs, cs, first, B, cap: dict(rs=rs, first=first, B=B, cap=cap)
    ShardedStepper.add_batch_dealt_finish = lambda self, half: self.add_batch(
        *(t.contiguous() for t in half["rs"].arrays(half["first"], half["first"] + half["B"])), half["cap"])
    _two_rank_trainer(rank, port, argv, out_dir)


def _two_rank_trainer(rank, port, argv, out_dir):
    """One rank of `python -m trainer.estimator` under a launcher, both ranks on the box's one GPU: the HIP
    kernels are the product's, only the transport of the collectives is gloo instead of RCCL."""
    import os
    import sys
    import torch
    import torch.distributed as dist
    here = Path(__file__).resolve().parent
    for p in (here.parent, here):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0")
    from trainer import estimator
    from trainer.config_utils import parse_args
    from trainer.stepper import HipBackend
    dist.init_process_group("gloo", rank=rank, world_size=2)
    box = [parse_args(argv) if rank == 0 else None]          # as trainer.estimator.main: rank 0 decides job_dir
    dist.broadcast_object_list(box, src=0)
    params = box[0]
    est = estimator.Estimator(params, backend=HipBackend("cuda:0"), dist=dist, device="cuda:0")
    est.train(params["train_steps"])
    t = est.model.tables
    torch.save({"R": t.embeddings("R").cpu(), "C": t.embeddings("C").cpu(), "br": t.br.cpu(), "bc": t.bc.cpu(), "g": t.global_bias,
                "step": t.global_step}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_data_parallel_trainer_on_one_gpu(hip, tmp_path):
    """The multi-rank host loop end to end (sharded stream, agreed seed, dense-gradient all-reduce, rank-0
    checkpoints, collective eval): replicas stay bit-identical, the loss falls, only rank 0 writes."""
    import torch.multiprocessing as mp
    csv, vocab = GOLDEN / "text8_cov90_ctx5_interaction.csv", GOLDEN / "text8_cov90_ctx5_vocab.txt"
    job = tmp_path / "job"
    argv = ["--train-csv", str(csv), "--vocab-txt", str(vocab), "--job-dir", str(job), "--disable-datetime-path",
            "--embedding-size", "32", "--optimizer", "Adagrad", "--learning-rate", "0.05", "--batch-size", "64",
            "--train-steps", "60", "--log-every", "10", "--seed", "21", "--save-checkpoints-secs", "0"]    # (an eval pass at every logging point)
    mp.spawn(_two_rank_trainer, args=(free_port(), argv, str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(tmp_path / ("rank%d.pt" % r)) for r in range(2))
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(a[n], b[n]), n
    assert a["g"] == b["g"] and a["step"] == b["step"] == 60
    log = [json.loads(l) for l in (job / "train_log.jsonl").read_text().splitlines()]
    assert [r["global_step"] for r in log] == [10, 20, 30, 40, 50, 60]
    # (a log line is the loss of ONE batch of 128 pairs: too noisy to order; the eval pass weighs the whole file)
    assert all(np.isfinite(r["loss"]) for r in log)
    assert (job / "model.ckpt-60.pt").exists()
    ev = [json.loads(l) for l in (job / "eval" / "eval_log.jsonl").read_text().splitlines()]
    assert ev[0]["global_step"] <= 10 and ev[-1]["global_step"] == 60 and 0 < ev[-1]["average_loss"] < ev[0]["average_loss"]


@pytest.mark.parametrize("name", ["sgd", "Adamax", "rmsprop"])
def test_two_rank_trainer_with_other_keras_optimizers_on_one_gpu(hip, tmp_path, name):
    """`python -m trainer.estimator --optimizer <any Keras name but Nadam>` under a two-rank launcher (both ranks on the one GPU,
    gloo transport): the per-row optimizers on the touched-rows exchange, RMSprop on the dense all-reduce; replicas stay
    bit-identical, the eval loss over the whole file falls, rank 0's checkpoint carries the slots."""
    import torch.multiprocessing as mp
    csv, vocab = GOLDEN / "text8_cov90_ctx5_interaction.csv", GOLDEN / "text8_cov90_ctx5_vocab.txt"
    job = tmp_path / "job"
    lr = {"sgd": "0.5", "Adamax": "0.01", "rmsprop": "0.002"}[name]
    argv = ["--train-csv", str(csv), "--vocab-txt", str(vocab), "--job-dir", str(job), "--disable-datetime-path",
            "--embedding-size", "32", "--optimizer", name, "--learning-rate", lr, "--batch-size", "64",
            "--train-steps", "60", "--log-every", "20", "--seed", "23", "--save-checkpoints-secs", "0"]
    mp.spawn(_two_rank_trainer, args=(free_port(), argv, str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(tmp_path / ("rank%d.pt" % r)) for r in range(2))
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(a[n], b[n]), n
    assert a["g"] == b["g"] and a["step"] == b["step"] == 60
    ev = [json.loads(l) for l in (job / "eval" / "eval_log.jsonl").read_text().splitlines()]
    assert ev[-1]["global_step"] == 60 and 0 < ev[-1]["average_loss"] < ev[0]["average_loss"]
    ck = torch.load(job / "model.ckpt-60.pt")["tables"]
    assert "slot1_R" in ck and (name != "Adamax" or "slot2_R" in ck)


def test_two_rank_trainer_with_touched_rows_exchange_on_one_gpu(hip, tmp_path):
    """--exchange rows: the same run with the all-gather of packed touched-row lists instead of the dense all-reduce
    gives the same tables as --exchange dense, bit for bit (two ranks: g0 + g1 either way), seeded."""
    import torch.multiprocessing as mp
    csv, vocab = GOLDEN / "text8_cov90_ctx5_interaction.csv", GOLDEN / "text8_cov90_ctx5_vocab.txt"
    res = {}
    for k, exchange in enumerate(("rows", "dense")):
        out = tmp_path / exchange
        out.mkdir()
        argv = ["--train-csv", str(csv), "--vocab-txt", str(vocab), "--job-dir", str(out / "job"), "--disable-datetime-path",
                "--embedding-size", "32", "--optimizer", "Adagrad", "--learning-rate", "0.05", "--batch-size", "64",
                "--train-steps", "40", "--log-every", "20", "--seed", "5", "--exchange", exchange, "--skip-eval"]
        mp.spawn(_two_rank_trainer, args=(free_port(), argv, str(out)), nprocs=2, join=True)
        res[exchange] = [torch.load(out / ("rank%d.pt" % r)) for r in range(2)]
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(res["rows"][0][n], res["rows"][1][n]), n            # replicas identical
        assert torch.equal(res["rows"][0][n], res["dense"][0][n]), n           # and equal to the dense exchange
    assert res["rows"][0]["g"] == res["dense"][0]["g"] and res["rows"][0]["step"] == 40


def _two_rank_sharded(rank, port, out_dir, V, d, B, steps):
    """Two ranks on the one GPU over gloo, BOTH tables sharded (trainer.stepper.ShardedStepper)."""
    import os
    import sys
    import torch
    import torch.distributed as dist
    here = Path(__file__).resolve().parent
    for p in (here.parent, here.parent / "oracle", here):
        sys.path.insert(0, str(p))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0")
    import numpy as np
    from helpers import make_batch, to_dev
    from trainer.hip_api import DeviceTables
    from trainer.stepper import HipBackend, ShardedStepper, owned_rows, route_by_row_owner
    dist.init_process_group("gloo", rank=rank, world_size=2)
    whole = DeviceTables(V, d, "Adagrad", device="cuda:0", seed=7)
    own = owned_rows(V, 2, rank)
    shard = DeviceTables(V, d, "Adagrad", device="cuda:0", seed=0, V_row=own, V_col=own)
    for n in ("R", "C", "br", "bc"):
        getattr(shard, n).copy_(getattr(whole, n)[rank::2])
    st = ShardedStepper(HipBackend("cuda:0"), shard, dict(l2_reg=0.01, reg_mult=2.0, learning_rate=0.05), B, 2, rank, dist)
    handles = []
    for k in range(steps):
        row, col, w, y = make_batch(4000 + 10 * k + rank, B, V)
        if k == steps - 1:                # a step whose col ids all belong to rank 0 (even ids): rank 1 serves nothing
            col = (col // 2 * 2) % V
            col[col == row] = (col[col == row] + 2) % V
        row, col, w, y = to_dev(row, col.astype(np.int32), w, y)
        got = route_by_row_owner(dict(row=row, col=col, w=w, y=y), 2, rank, dist)
        handles.append(st.add_batch(got["row"], got["col"], got["w"], got["y"], 16))
    for h in handles:
        st.step(h)
    torch.save({n: getattr(shard, n).cpu() for n in ("R", "C", "br", "bc")} | {"g": shard.global_bias, "loss": st.read_loss()},
               os.path.join(out_dir, "shard%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_fully_sharded_step_on_one_gpu(hip, tmp_path):
    """BASELINE config 5 with both tables sharded, two ranks sharing the box's GPU (gloo transport, HIP kernels):
    equals the single-GPU step on the union of the ranks' batches within fp32 rounding of the sum over ranks."""
    import torch.multiprocessing as mp
    from helpers import make_batch, to_dev
    from trainer.hip_api import DeviceTables, make_hyper
    V, d, B, steps = 2001, 64, 3000, 4
    mp.spawn(_two_rank_sharded, args=(30300 + os.getpid() % 200, str(tmp_path), V, d, B, steps), nprocs=2, join=True)
    ref_t = DeviceTables(V, d, "Adagrad", device="cuda:0", seed=7)
    h = make_hyper(learning_rate=0.05, batch_size=2 * B, step_form=1)
    loss_out = torch.zeros(4, device="cuda:0")
    for k in range(steps):
        parts = [list(make_batch(4000 + 10 * k + r, B, V)) for r in range(2)]
        if k == steps - 1:
            for p_ in parts:
                p_[1] = ((p_[1] // 2 * 2) % V).astype(np.int32)
                p_[1][p_[1] == p_[0]] = (p_[1][p_[1] == p_[0]] + 2) % V
        joint = [np.concatenate(x) for x in zip(*parts)]
        hip.step_adagrad(hip.build_plan(*to_dev(*joint), V, chunk_cap=16), ref_t, h, loss_out)
    shards = [torch.load(tmp_path / ("shard%d.pt" % r)) for r in range(2)]
    for r, s_ in enumerate(shards):
        for n in ("R", "C", "br", "bc"):
            np.testing.assert_allclose(s_[n].numpy(), getattr(ref_t, n)[r::2].cpu().numpy(), rtol=2e-5, atol=2e-6, err_msg=n)
        np.testing.assert_allclose(s_["g"], ref_t.global_bias, rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(s_["loss"]["loss"], loss_out[0].item(), rtol=1e-5)


def test_logistic_matrix_factorisation_cli(hip, tmp_path):
    """`python -m trainer.logistic_matrix_factorisation` on the reference-made CSV (columns `value` / `neg_weight`):
    trains, logs a falling merged loss, evaluates both heads, and its first steps match the oracle."""
    from helpers import tables_from_oracle                                  # noqa: F401  (path side effect)
    from trainer import logistic_matrix_factorisation as lmf
    csv, vocab = GOLDEN / "text8_cov90_ctx5_interaction.csv", GOLDEN / "text8_cov90_ctx5_vocab.txt"
    job = tmp_path / "job"
    argv = ["--train-csv", str(csv), "--vocab-txt", str(vocab), "--job-dir", str(job), "--disable-datetime-path",
            "--embedding-size", "16", "--optimizer", "Adagrad", "--learning-rate", "0.05", "--batch-size", "64",
            "--neg-factor", "0.5", "--train-steps", "80", "--log-every", "20", "--seed", "3"]
    lmf.main(argv)
    params = json.loads((job / "params.json").read_text())
    assert params["head"] == "logistic" and params["input_fn_args"]["select_columns"][2:] == ["value", "neg_weight"]
    log = [json.loads(l) for l in (job / "train_log.jsonl").read_text().splitlines()]
    assert [r["global_step"] for r in log] == [20, 40, 60, 80]
    assert all(np.isfinite(r["loss"]) for r in log) and log[-1]["loss"] < log[0]["loss"]
    ev = [json.loads(l) for l in (job / "eval" / "eval_log.jsonl").read_text().splitlines()][-1]
    assert ev["global_step"] == 80 and 0.0 < ev["prediction/mean/neg"] < 1.0 and ev["average_loss/pos"] > 0
    np.testing.assert_allclose(ev["average_loss"], ev["average_loss/pos"] + 0.5 * ev["average_loss/neg"], rtol=1e-12)


def test_non_finite_loss_stops_training(hip, tmp_path):
    """The Estimator's NanTensorHook: a non-finite loss ends the run with an error instead of training on
    (reference: tf.estimator's default hooks under train_and_evaluate, estimator.py:95)."""
    import pandas as pd
    from trainer import estimator
    csv, vocab = GOLDEN / "text8_cov90_ctx5_interaction.csv", GOLDEN / "text8_cov90_ctx5_vocab.txt"
    df = pd.read_csv(csv, keep_default_na=False, na_filter=False)
    df.loc[5, "glove_value"] = float("inf")
    bad = tmp_path / "interaction.csv"
    df.to_csv(bad, index=False)
    argv = ["--train-csv", str(bad), "--vocab-txt", str(vocab), "--job-dir", str(tmp_path / "job"),
            "--disable-datetime-path", "--embedding-size", "8", "--optimizer", "Adagrad", "--batch-size", "64",
            "--train-steps", "200", "--log-every", "10", "--seed", "1"]
    with pytest.raises(FloatingPointError, match="global_step"):
        estimator.main(argv)


@pytest.mark.parametrize("optimizer,lr,B,graphs,segment", [("Adagrad", 0.05, 256, True, 0), ("Adam", 0.001, 256, True, 5), ("Adagrad", 0.05, 5000, True, 2),
                                                           ("Adagrad", 0.05, 256, False, 3), ("Adam", 0.001, 256, False, 0), ("Adagrad", 0.05, 5000, False, 0),
                                                           ("Adagrad", 0.05, 1024, True, 1)])
def test_reshuffling_runner_equals_plain_dynamic_stepping(hip, optimizer, lr, B, graphs, segment):
    """--epoch-shuffle full: epochs dealt from the sorted master orders, the index of a segment of batches numbered by three
    launches on the side stream while the segment before steps (replayed from hipGraphs of 2^k steps, or launched one by
    one) give bit for bit what sorting and stepping batch after batch of the same epochs gives, across segment and epoch
    boundaries."""
    from trainer import synthetic
    from trainer.data_utils import NonzeroStream
    from trainer.hip_api import DeviceTables, make_hyper
    from trainer.stepper import HipBackend, ReshufflingRunner
    V, d = 300, 32                                           # B = 5000: the tiled (multi-launch) index builder on the side streams
    row, col, w, y = synthetic.text8_shaped(V=V, n_tokens=60_000, seed=2)
    coo = dict(row=row.numpy(), col=col.numpy(), w=w.numpy(), y=y.numpy())
    backend = HipBackend("cuda:0")
    hyper = make_hyper(learning_rate=lr, batch_size=B)
    steps = None
    results = []
    for mode in ("runner", "plain"):
        stream = NonzeroStream(coo, B, V, backend, "cuda:0", seed=11, static_plans=False)
        tables = DeviceTables(V, d, optimizer, seed=4)
        nb = stream.batches_per_epoch
        steps = 2 * nb + 7                                  # two full epochs and a bit
        if mode == "runner":
            runner = ReshufflingRunner(hip, stream, tables, hyper, burst=16, graphs=graphs, segment=segment)
            assert runner.graphs_on == graphs and runner.S == min(segment or 64, nb)
            done = 0
            while done < steps:
                done += runner.run(min(steps - done, 11))    # (logging points fall anywhere inside a segment)
            loss = runner.read_loss()["loss"]
            assert bool(runner.graphs) == graphs
            runner.release_graphs()
        else:
            G = hip.dense_grad_buffer(tables) if optimizer == "Adam" else None
            loss_out = torch.zeros(4, device="cuda:0")
            pos = nb                                          # forces the first reshuffle, as the runner's constructor does
            for _ in range(steps):
                if pos >= nb:
                    stream.reshuffle_in_place()
                    pos = 0
                plan = hip.build_plan(*stream.batch(pos), V, chunk_cap=0)
                if G is None:
                    hip.step_adagrad(plan, tables, hyper, loss_out)
                else:
                    hip.step_adam(plan, tables, hyper, G, loss_out)
                pos += 1
            loss = float(loss_out[0])
        results.append((tables, loss))
    (a, la), (b, lb) = results
    assert a.global_step == b.global_step == steps and la == lb
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n


@pytest.mark.parametrize("graphs", [False, True])
def test_reshuffling_runner_on_big_tables_takes_the_fused_step(hip, graphs):
    """--epoch-shuffle full at a scale where the step is fused (V = 60 k, d = 300, B = 131,072: the staging plans carry run
    words and take their pair fields from the epoch's arrays — borrowed when the steps are issued by C calls, copied when they are
    replayed from graphs —, the row table is twinned): equal,
    within the fp32 tolerance of summing a heavy id's pairs in another order, to building and stepping batch after batch in
    two launches, across an epoch boundary; the twin form really ran."""
    from trainer import synthetic
    from trainer.data_utils import NonzeroStream
    from trainer.hip_api import DeviceTables, make_hyper, staging_records
    from trainer.stepper import HipBackend, ReshufflingRunner
    V, d, B = 60000, 300, 131072
    assert staging_records(B, V, V, d) is True
    wl = synthetic.make_workload("text8_v50k_d300", seed=2, device="cuda:0", work_device="cuda:0")
    n = 3 * B + 1234
    g = torch.Generator(device="cpu").manual_seed(1)
    # (Zipf rows, uniform cols: a batch touches ~75 k ids — the runner looks at its first batch before it goes fused)
    coo = dict(row=wl["row"][:n].cpu().numpy(), col=torch.randint(0, V, (n,), generator=g).int().numpy(),
               w=wl["w"][:n].cpu().numpy(), y=wl["y"][:n].cpu().numpy())
    backend = HipBackend("cuda:0")
    hyper = make_hyper(learning_rate=0.05, batch_size=B)
    results = []
    for mode in ("runner", "plain"):
        stream = NonzeroStream(coo, B, V, backend, "cuda:0", seed=11, static_plans=False)
        tables = DeviceTables(V, d, "Adagrad", seed=4)
        nb = stream.batches_per_epoch
        steps = nb + 2                                       # across an epoch boundary
        if mode == "runner":
            tables.enable_twin()      # (the policy twins row tables of 128 MB and more; this one has 73 MB: asked for here)
            runner = ReshufflingRunner(hip, stream, tables, hyper, burst=4, graphs=graphs, segment=2)
            p0 = runner.slots[0].plans[0]       # a fused step on batches indexed every step: run words + the pair fields as dealt, no records
            assert tables.R_ver is not None and p0.r_crec is None and p0.r_chunk_hw is not None and p0.fusable
            assert p0.borrows == (not graphs) and (p0.r_partner is None) == p0.borrows      # (a captured step cannot follow the epochs' arrays)
            done = 0
            while done < steps:
                done += runner.run(steps - done)
            assert getattr(tables, "_twin_dirty", False)      # a step of the twin form was issued
            loss = runner.read_loss()["loss"]
            runner.release_graphs()
        else:
            loss_out = torch.zeros(4, device="cuda:0")
            pos = nb
            for _ in range(steps):
                if pos >= nb:
                    stream.reshuffle_in_place()
                    pos = 0
                hip.step_adagrad(hip.build_plan(*stream.batch(pos), V, chunk_cap=0), tables, hyper, loss_out)
                pos += 1
            loss = float(loss_out[0])
        results.append((tables, loss))
    (a, la), (b, lb) = results
    assert a.global_step == b.global_step == steps
    np.testing.assert_allclose(la, lb, rtol=1e-5)
    for n_ in ("R", "C", "br", "bc"):
        np.testing.assert_allclose(getattr(a, n_).cpu().numpy(), getattr(b, n_).cpu().numpy(), rtol=5e-5, atol=5e-6, err_msg=n_)


def test_sharded_runner_prepares_the_next_epoch_beside_the_steps(hip):
    """Reshuffled epochs with both tables sharded (one rank, the exchange exercised): the next epoch's batches — fetch lists,
    indexes — are prepared one per step on a stream of their own while this epoch trains; across four epoch boundaries the
    result is bit for bit what preparing every epoch at its boundary gives, and within tolerance what the plain runner gives."""
    from trainer.data_utils import NonzeroStream
    from trainer.hip_api import DeviceTables, make_hyper
    from trainer.stepper import HipBackend, ReshufflingRunner, ShardedStepper
    V, d, B, nnz = 3000, 32, 2048, 2048 * 6 + 100
    rng = np.random.default_rng(8)
    data = {"row": rng.integers(0, V, nnz).astype(np.int32), "col": rng.integers(0, V, nnz).astype(np.int32),
            "w": rng.uniform(0.1, 1, nnz).astype(np.float32), "y": rng.normal(0, 1, nnz).astype(np.float32)}
    dev = torch.device("cuda:0")
    out = []
    for mode in ("ahead", "boundary", "plain", "dealt"):
        backend = HipBackend(dev)
        backend.hip = hip
        tables = DeviceTables(V, d, "Adagrad", device=dev, seed=2)
        backend.row_floats = tables.d
        stepper = None if mode == "plain" else ShardedStepper(backend, tables, dict(learning_rate=0.05), B, 1, 0, None, exercise_exchange=True)
        # "dealt": col ids numbered owner-major (the identity with one owner) — the batches' fetch lists and indexes then come
        # straight from the dealt order (add_batch_dealt: no sort), the same lists and plans bit for bit
        stream = NonzeroStream(dict(data), B, V, backend, dev, seed=21, static_plans=False, cols_by_owner=1 if mode == "dealt" else 0)
        hyper = make_hyper(batch_size=B, learning_rate=0.05)
        runner = ReshufflingRunner(hip, stream, tables, stepper.hyper if stepper else hyper, stepper=stepper, graphs=False, burst=4)
        if mode == "boundary":
            runner._prepare_ahead = lambda: None
        steps, done = 6 * 4 + 3, 0
        while done < steps:
            done += runner.run(min(5, steps - done))
            if mode == "dealt":
                assert stepper.col_per == V and "block" in stepper.batches[runner.handles[0]]     # the dealt path really prepared them
            if mode in ("ahead", "dealt") and 0 < runner.position < runner.nb:
                assert 0 < len(runner._ahead) <= runner.nb                  # batches of the next epoch are being prepared
        torch.cuda.synchronize()
        out.append((tables, runner.read_loss()["loss"]))
        if stepper is not None:
            assert sum(b is not None for b in stepper.batches) <= 2 * runner.nb      # finished epochs are dropped
    (a, la), (b, lb), (c, lc), (e, le) = out
    assert a.global_step == b.global_step == c.global_step == e.global_step == 27 and la == lb == le
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(a, n), getattr(e, n)), ("dealt", n)
    np.testing.assert_allclose(la, lc, rtol=1e-4)
    for n in ("R", "C", "br", "bc"):
        assert torch.equal(getattr(a, n), getattr(b, n)), n
        np.testing.assert_allclose(getattr(a, n).cpu().numpy(), getattr(c, n).cpu().numpy(), rtol=1e-4, atol=1e-6, err_msg=n)


def test_cli_with_full_epoch_shuffle(hip, tmp_path):
    from trainer import estimator
    csv, vocab = GOLDEN / "text8_cov90_ctx5_interaction.csv", GOLDEN / "text8_cov90_ctx5_vocab.txt"
    job = tmp_path / "job"
    estimator.main(["--train-csv", str(csv), "--vocab-txt", str(vocab), "--job-dir", str(job), "--disable-datetime-path",
                    "--embedding-size", "16", "--optimizer", "Adagrad", "--learning-rate", "0.05", "--batch-size", "64",
                    "--train-steps", "150", "--log-every", "25", "--seed", "5", "--epoch-shuffle", "full",
                    "--index-segment", "3", "--save-checkpoints-secs", "0"])
    log = [json.loads(l) for l in (job / "train_log.jsonl").read_text().splitlines()]
    ev = [json.loads(l) for l in (job / "eval" / "eval_log.jsonl").read_text().splitlines()]
    assert log[-1]["global_step"] == 150 and ev[-1]["global_step"] == 150 and ev[-1]["average_loss"] < ev[0]["average_loss"]
    assert all(b["global_step"] - a["global_step"] < 50 for a, b in zip(log, log[1:]))     # a line per crossed multiple of --log-every
    assert (job / "model.ckpt-150.pt").exists()


def test_two_rank_row_sharded_trainer_on_one_gpu(hip, tmp_path):
    """--row-sharded (BASELINE config 5) end to end with two ranks on the box's one GPU (gloo transport): every
    nonzero is routed to the owner of its row, the row side trains locally, the col side through one all-reduce;
    the checkpoint rank 0 writes holds the WHOLE model again, and a single process can resume / export from it."""
    import torch.multiprocessing as mp
    from trainer import estimator, export_embeddings
    csv, vocab = GOLDEN / "text8_cov90_ctx5_interaction.csv", GOLDEN / "text8_cov90_ctx5_vocab.txt"
    job = tmp_path / "job"
    argv = ["--train-csv", str(csv), "--vocab-txt", str(vocab), "--job-dir", str(job), "--disable-datetime-path",
            "--embedding-size", "24", "--optimizer", "Adagrad", "--learning-rate", "0.05", "--batch-size", "48",
            "--train-steps", "60", "--log-every", "20", "--seed", "9", "--row-sharded"]
    mp.spawn(_two_rank_trainer, args=(free_port(), argv, str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(tmp_path / ("rank%d.pt" % r)) for r in range(2))
    V = len(vocab.read_text().split("\n"))
    assert a["R"].shape[0] + b["R"].shape[0] == V and a["R"].shape[0] == (V + 1) // 2      # disjoint row shards
    assert torch.equal(a["C"], b["C"]) and torch.equal(a["bc"], b["bc"]) and a["g"] == b["g"]   # replicated col side
    log = [json.loads(l) for l in (job / "train_log.jsonl").read_text().splitlines()]
    assert [r["global_step"] for r in log] == [20, 40, 60] and log[-1]["loss"] < log[0]["loss"]
    # the checkpoint is a whole-model checkpoint: rows interleave back (row u on rank u % 2 at index u // 2)
    blob = torch.load(job / "model.ckpt-60.pt", weights_only=False)["tables"]
    assert blob["R"].shape == (V, 24) and blob["V_row"] == V
    assert torch.equal(blob["R"][0::2], a["R"]) and torch.equal(blob["R"][1::2], b["R"])
    assert torch.equal(blob["br"][1::2], b["br"]) and torch.equal(blob["C"], a["C"])
    out = tmp_path / "embeddings.json"
    export_embeddings.main(job_dir=str(job), embeddings_json=str(out))      # one process, whole model
    emb = json.loads(out.read_text())
    assert len(emb["the"]["item_embedding"]) == 24
    # a single process resumes from it as an ordinary (unsharded) run
    estimator.main([x for x in argv if x != "--row-sharded"][:-6] + ["--train-steps", "70", "--log-every", "10", "--seed", "9"])
    assert (job / "model.ckpt-70.pt").exists()


def test_two_rank_trainer_with_both_tables_sharded_on_one_gpu(hip, tmp_path):
    """--row-sharded --shard-cols end to end with two ranks on the box's one GPU (gloo transport): rows AND cols live on
    id % 2, a step fetches the col rows its batch touches and returns their gradients (the form `bench.py --gpus N` times at
    configs 4 / 5), epochs reshuffled with the next epoch's batches prepared beside the steps; eval passes gather the col
    side; the checkpoint holds the WHOLE model; same seed, same routing as --row-sharded alone: the same model within the
    fp32 rounding of where the col gradients are summed."""
    import torch.multiprocessing as mp
    from trainer import estimator
    csv, vocab = GOLDEN / "text8_cov90_ctx5_interaction.csv", GOLDEN / "text8_cov90_ctx5_vocab.txt"
    V = len(vocab.read_text().split("\n"))
    blobs = {}
    for name, extra, entry in (("both", ["--shard-cols"], _two_rank_trainer), ("sorted", ["--shard-cols"], _two_rank_trainer_sorting_prepare),
                               ("rows", [], _two_rank_trainer)):
        out = tmp_path / name
        out.mkdir()
        job = out / "job"
        argv = ["--train-csv", str(csv), "--vocab-txt", str(vocab), "--job-dir", str(job), "--disable-datetime-path",
                "--embedding-size", "24", "--optimizer", "Adagrad", "--learning-rate", "0.05", "--batch-size", "48",
                "--train-steps", "90", "--log-every", "30", "--seed", "9", "--row-sharded"] + extra
        mp.spawn(entry, args=(free_port(), argv, str(out)), nprocs=2, join=True)
        a, b = (torch.load(out / ("rank%d.pt" % r)) for r in range(2))
        assert a["R"].shape[0] + b["R"].shape[0] == V
        if extra:
            assert a["C"].shape[0] + b["C"].shape[0] == V and a["C"].shape[0] == (V + 1) // 2      # disjoint col shards too
        else:
            assert torch.equal(a["C"], b["C"])
        assert a["g"] == b["g"] and a["step"] == b["step"] == 90
        log = [json.loads(l) for l in (job / "train_log.jsonl").read_text().splitlines()]
        assert [r["global_step"] for r in log] == [30, 60, 90]
        ev = [json.loads(l) for l in (job / "eval" / "eval_log.jsonl").read_text().splitlines()]
        assert ev[-1]["global_step"] == 90 and 0 < ev[-1]["average_loss"] < 10          # (the eval pass saw whole col rows)
        blob = torch.load(job / "model.ckpt-90.pt", weights_only=False)["tables"]
        assert blob["R"].shape == (V, 24) and blob["C"].shape == (V, 24) and blob["V_row"] == V
        if extra:
            assert torch.equal(blob["C"][0::2], a["C"]) and torch.equal(blob["C"][1::2], b["C"]) and torch.equal(blob["bc"][1::2], b["bc"])
        blobs[name] = (blob, ev[-1]["average_loss"], job, argv)
    # the batches read off the dealt order (col ids numbered owner-major: fetch order = sorted order) give the fetch lists and
    # indexes the general preparation gives: the same model bit for bit
    for k in ("R", "C", "br", "bc", "slot1_R", "slot1_C"):
        assert torch.equal(blobs["both"][0][k], blobs["sorted"][0][k]), k
    # (--row-sharded alone deals other batches — its masters are sorted by the plain col ids —: the same loss level, not the same model)
    np.testing.assert_allclose(blobs["both"][1], blobs["rows"][1], rtol=0.05)
    # one process, plain tables, the whole-model checkpoint: the eval pass gives the loss the two ranks computed over their routed
    # pairs against the owner-major gathered col side (the renumbering is consistent with where the owners keep their rows)
    _, loss2, job, argv = blobs["both"]
    params = json.loads((job / "params.json").read_text())
    params.update(row_sharded=False, shard_cols=False)
    np.testing.assert_allclose(estimator.Estimator(params).evaluate()["average_loss"], loss2, rtol=1e-5)
    # a single process resumes from the whole-model checkpoint as an ordinary run
    estimator.main([x for x in argv if x not in ("--row-sharded", "--shard-cols")][:-6] + ["--train-steps", "100", "--log-every", "10", "--seed", "9"])
    assert (job / "model.ckpt-100.pt").exists()


def test_train_then_train_more_in_one_interpreter(hip, tmp_path):
    """A process that trains, keeps the first Estimator (stream, tables, staging plans) alive and trains on with a second one
    — a notebook, or train -> evaluate -> train: the second reshuffling runner captures and replays its hipGraphs beside
    whatever the first left behind, resumes from the first's checkpoint and reaches the absolute step count.  A third runner
    is built and stepped while the second one's graphs are still alive (round 3: a replay segfaulted in that situation when
    graphs held index builds on forked side streams)."""
    from trainer import estimator
    from trainer.config_utils import parse_args
    from trainer.data_utils import NonzeroStream
    from trainer.hip_api import DeviceTables, make_hyper
    from trainer.stepper import HipBackend, ReshufflingRunner
    csv, vocab = GOLDEN / "text8_cov90_ctx5_interaction.csv", GOLDEN / "text8_cov90_ctx5_vocab.txt"
    job = tmp_path / "job"
    argv = ["--train-csv", str(csv), "--vocab-txt", str(vocab), "--job-dir", str(job), "--disable-datetime-path",
            "--embedding-size", "16", "--optimizer", "Adagrad", "--learning-rate", "0.05", "--batch-size", "32",
            "--log-every", "40", "--seed", "2", "--skip-eval"]
    first = estimator.Estimator(parse_args(argv + ["--train-steps", "80"]))
    first.train(80)
    second = estimator.Estimator(parse_args(argv + ["--train-steps", "200"]))
    assert second.model.tables.global_step == 80                  # resumed from the first one's checkpoint
    second.train(200)
    assert second.model.tables.global_step == 200 and first.model.tables.global_step == 80
    # runners by hand, graphs kept alive across each other
    backend = HipBackend("cuda:0")
    coo = {k: getattr(first.stream(), k).cpu().numpy() for k in ("row", "col", "w", "y")}
    stream_a, stream_b = (NonzeroStream(coo, 32, first.vocab_size, backend, "cuda:0", seed=s, static_plans=False) for s in (1, 2))
    ta, tb = DeviceTables(first.vocab_size, 16, "Adagrad", seed=1), DeviceTables(first.vocab_size, 16, "Adagrad", seed=1)
    ra = ReshufflingRunner(hip, stream_a, ta, make_hyper(batch_size=32, learning_rate=0.05), burst=8, segment=4)
    done = 0
    while done < 50:
        done += ra.run(50 - done)
    assert ra.graphs
    rb = ReshufflingRunner(hip, stream_b, tb, make_hyper(batch_size=32, learning_rate=0.05), burst=8, segment=4)
    for _ in range(3):
        for r in (rb, ra):                                          # interleaved replays of both runners' graphs
            done = 0
            while done < 30:
                done += r.run(30 - done)
    assert ta.global_step == 140 and tb.global_step == 90
    assert np.isfinite(ra.read_loss()["loss"]) and np.isfinite(rb.read_loss()["loss"])


def test_other_optimizers_on_the_data_parallel_form_through_rccl_with_one_rank(hip):
    """The other Keras names `tf.keras.optimizers.get` resolves (reference train_utils.py:13-16) on the multi-rank data-parallel
    form, every collective issued through RCCL on this one GPU (`collectives=True`): the per-row optimizers (SGD with and
    without momentum, Adamax, Adadelta, Ftrl, and Nadam, whose untouched rows' m and v decay in a sweep in front of the apply) ride the
    touched-rows all-gather — glove_apply_packed_adagrad_f32 takes their epilogue from glove_hyper.optimizer —, RMSprop the dense all-reduce (glove_dense_adam_f32's RMSprop sweep).  Three steps ==
    the float64 oracle and == the single-GPU step (glove_step_sparse_f32) on the same batches."""
    import os
    import torch.distributed as dist
    from helpers import make_batch, oracle_tables, to_dev
    from test_gpu_optimizers import _check, _device_tables
    from trainer.stepper import HipBackend, Stepper
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29549", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    try:
        B, V, d, steps = 5000, 600, 64, 3
        backend = HipBackend("cuda:0")
        batches = [make_batch(60 + s, B, V) for s in range(steps)]
        for optimizer, extra in (("SGD", {}), ("SGD", dict(momentum=0.9, nesterov=True)), ("Adamax", {}), ("Adadelta", {}),
                                 ("Ftrl", {}), ("RMSprop", {}), ("Nadam", {})):
            lr = {"Adadelta": 1.0, "Ftrl": 0.05, "Nadam": 0.002}.get(optimizer, 0.01)
            hp = ref.Hyper(learning_rate=lr, **extra)
            t = oracle_tables(V, d, optimizer)
            multi_t, single_t = _device_tables(t), _device_tables(t)        # (fresh tables: the slots at their Keras initial values)
            kw = dict(l2_reg=hp.l2_reg, reg_mult=hp.reg_mult, learning_rate=lr, optimizer=optimizer, **extra)
            backend.row_floats = multi_t.d
            multi = Stepper(backend, multi_t, kw, B, world=1, dist=dist, exchange="auto", collectives=True)
            plans = [backend.build_plan(*to_dev(*bt), V, 0).compact(hip.lib, multi_t.d) for bt in batches]
            multi.prepare(plans)
            names = [n for n, _ in multi.phases()]
            if optimizer == "RMSprop":
                assert not multi.rows and "all_reduce" in names and "dense_apply" in names
            else:
                assert multi.rows and "all_gather" in names and "combine_apply" in names
            single = Stepper(backend, single_t, kw, B)
            for s in range(steps):
                multi.step(plans[s])
                single.step(plans[s])
                ref.train_step(t, *batches[s], hp)
            info = "%s %s" % (optimizer, extra)
            _check(multi_t, t, 5e-5, 5e-6)          # (the tolerance of the single-GPU trajectories of these optimizers)
            for n in ("R", "C", "br", "bc"):
                torch.testing.assert_close(getattr(multi_t, n), getattr(single_t, n), rtol=5e-5, atol=5e-6, msg=lambda m: info + " " + n + ": " + m)
                torch.testing.assert_close(multi_t.s1[n], single_t.s1[n], rtol=5e-5, atol=5e-6, msg=lambda m: info + " slot1 " + n + ": " + m)
                if n in multi_t.s2:
                    torch.testing.assert_close(multi_t.s2[n], single_t.s2[n], rtol=5e-5, atol=5e-6, msg=lambda m: info + " slot2 " + n + ": " + m)
            torch.testing.assert_close(multi_t.scalars[:3], single_t.scalars[:3], rtol=5e-5, atol=5e-6)
            assert multi_t.global_step == single_t.global_step == steps, info
            got, want = multi.read_loss(), single.read_loss()
            assert abs(got["loss"] - want["loss"]) <= 2e-5 * abs(want["loss"]), (info, got, want)
    finally:
        dist.destroy_process_group()
