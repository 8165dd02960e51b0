"""Host-side mirror of the reference interface: defaults, flags, job_dir layout, checkpoints."""
import json
import os
import re
from pathlib import Path

import numpy as np
import pytest
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"


def test_defaults_match_reference_app_ini():
    """reference configs/app.ini:39-53 under ENVIRONMENT=dev."""
    from trainer import config
    assert (config.EMBEDDING_SIZE, config.L2_REG, config.OPTIMIZER, config.LEARNING_RATE, config.BATCH_SIZE,
            config.TOP_K) == (64, 0.01, "Adam", 0.001, 1024, 20)
    assert config.TRAIN_STEPS == 1024                      # [dev] override (app.ini:49-50)
    assert config.load_settings(environment="prod")["TRAIN_STEPS"] == 65536
    assert (config.ROW_NAME, config.COL_NAME, config.TARGET_NAME, config.WEIGHT_NAME) == (
        "row_token", "col_token", "glove_value", "glove_weight")
    assert config.TRAIN_CSV == "data/interaction.csv" and config.JOB_DIR == "checkpoints/estimator"


def test_a_users_app_ini_overrides_the_built_in_defaults(tmp_path):
    """The reference's configuration file (configs/app.ini, sections picked by $ENVIRONMENT) still works."""
    from trainer import config
    ini = tmp_path / "app.ini"
    ini.write_text("[DEFAULT]\nCHECKPOINTS_DIR = ckpt\nMODEL_NAME = mf\nJOB_DIR = %(CHECKPOINTS_DIR)s/%(MODEL_NAME)s-x\n"
                   "BATCH_SIZE = 4096\nDATA_DIR = corpus\n[dev]\nTRAIN_STEPS = 77\n[prod]\nOPTIMIZER = Adagrad\n")
    dev = config.load_settings("dev", ini)
    assert (dev["BATCH_SIZE"], dev["TRAIN_STEPS"], dev["JOB_DIR"], dev["OPTIMIZER"]) == (4096, 77, "ckpt/mf-x", "Adam")
    assert dev["TRAIN_CSV"] == "corpus/interaction.csv" and dev["EMBEDDINGS_JSON"] == "ckpt/embeddings.json"
    prod = config.load_settings("prod", ini)
    assert prod["OPTIMIZER"] == "Adagrad" and prod["TRAIN_STEPS"] == 65536
    with pytest.raises(KeyError):
        config.load_settings("staging", ini)


def test_cli_flags_are_the_reference_flags():
    from trainer.config_utils import build_parser
    flags = {a.option_strings[0] for a in build_parser()._actions if a.option_strings}
    reference = {"--train-csv", "--vocab-txt", "--row-name", "--col-name", "--target-name", "--weight-name",
                 "--pos-name", "--neg-name", "--job-dir", "--disable-datetime-path", "--embedding-size", "--l2-reg",
                 "--neg-factor", "--optimizer", "--learning-rate", "--batch-size", "--train-steps",
                 "--steps-per-epoch", "--top-k"}                       # config_utils.py:78-180
    assert reference <= flags
    ns = build_parser().parse_args(["--embedding-size", "300", "--optimizer", "Adagrad", "--learning-rate", "0.05"])
    assert ns.embedding_size == 300 and ns.optimizer == "Adagrad" and ns.learning_rate == 0.05
    assert ns.reg_multiplicity == 2.0


def test_init_params_builds_the_job_dir(tmp_path):
    from trainer.config_utils import parse_args
    vocab = GOLDEN / "text8_cov90_ctx5_vocab.txt"
    params = parse_args(["--job-dir", str(tmp_path / "job"), "--vocab-txt", str(vocab), "--train-csv", "x.csv"])
    assert re.fullmatch(r".*job-\d{8}-\d{6}", params["job_dir"])          # config_utils.py:57-61
    job = Path(params["job_dir"])
    assert (job / vocab.name).read_text() == vocab.read_text()             # copied (config_utils.py:63-66)
    assert params["vocab_txt"] == str(job / vocab.name)                   # repointed (config_utils.py:63-66)
    saved = json.loads((job / "params.json").read_text())
    assert saved["input_fn_args"] == {"file_pattern": "x.csv", "batch_size": 1024,
                                      "select_columns": ["row_token", "col_token", "glove_weight", "glove_value"],
                                      "target_names": ["glove_value"]}
    assert saved["serving_input_fn_args"] == {"string_features": ["row_token", "col_token"]}
    fixed = parse_args(["--job-dir", str(tmp_path / "fixed"), "--disable-datetime-path", "--vocab-txt", str(vocab)])
    assert fixed["job_dir"] == str(tmp_path / "fixed")


def test_get_optimizer_by_keras_name():
    from trainer.train_utils import get_optimizer
    assert get_optimizer("adagrad", learning_rate=0.1) == {
        "class_name": "Adagrad", "config": {"initial_accumulator_value": 0.1, "epsilon": 1e-7, "learning_rate": 0.1}}
    assert get_optimizer("Adam")["config"]["beta_2"] == 0.999
    # every name tf.keras.optimizers.get resolves in Keras 2.11 (reference train_utils.py:13-16 accepts any of them)
    for name in ("sgd", "rmsprop", "adamax", "nadam", "adadelta", "ftrl"):
        assert get_optimizer(name)["class_name"].lower() == name
    with pytest.raises(ValueError, match="no HIP kernel"):
        get_optimizer("Lion")


def test_checkpoint_layout_and_resume(tmp_path):
    from trainer.hip_api import DeviceTables
    from trainer.train_utils import CheckpointManager
    t = DeviceTables(12, 8, "Adam", device="cpu", seed=0)
    mgr = CheckpointManager(str(tmp_path), save_secs=300, keep_max=3)
    for step in (0, 10, 20, 30, 40):
        t.step.fill_(step)
        t.R += 1.0
        mgr.save(t)
    state = (tmp_path / "checkpoint").read_text()
    assert state.splitlines()[0] == 'model_checkpoint_path: "model.ckpt-40"'
    assert sorted(p.name for p in tmp_path.glob("model.ckpt-*")) == [
        "model.ckpt-20.pt", "model.ckpt-30.pt", "model.ckpt-40.pt"]            # keep_checkpoint_max
    fresh = DeviceTables(12, 8, "Adam", device="cpu", seed=1)
    assert CheckpointManager(str(tmp_path)).restore(fresh)
    assert fresh.global_step == 40 and torch.equal(fresh.R, t.R) and torch.equal(fresh.s2["C"], t.s2["C"])
    with pytest.raises(ValueError, match="checkpoint is for"):
        CheckpointManager(str(tmp_path)).restore(DeviceTables(12, 16, "Adam", device="cpu", seed=1))
    assert not CheckpointManager(str(tmp_path / "empty")).restore(fresh)


def test_device_tables_follow_keras_defaults():
    from trainer.hip_api import DeviceTables
    t = DeviceTables(1000, 64, "Adagrad", device="cpu", seed=0)
    for w in (t.R, t.C, t.br, t.bc):
        assert -0.05 <= float(w.min()) and float(w.max()) <= 0.05 and abs(float(w.mean())) < 5e-3   # U(-0.05, 0.05)
    assert float(t.scalars[0]) == 0.0 and float(t.scalars[1]) == pytest.approx(0.1)               # g = 0, acc = 0.1
    assert all(float(s.min()) == pytest.approx(0.1) for s in t.s1.values())
    a = DeviceTables(10, 8, "Adam", device="cpu", seed=0)
    assert all(float(s.abs().max()) == 0.0 for s in list(a.s1.values()) + list(a.s2.values()))
    assert not torch.equal(DeviceTables(10, 8, "Adam", device="cpu").R, DeviceTables(10, 8, "Adam", device="cpu").R)


def test_synthetic_workloads_are_well_formed():
    from trainer import synthetic
    row, col, w, y = synthetic.text8_shaped(V=600, n_tokens=400_000, seed=3)
    assert row.dtype == torch.int32 and w.dtype == torch.float32 and (row != col).all()
    assert int(row.max()) < 600 and float(w.min()) > 0 and float(w.max()) <= 1.0
    pairs = set(zip(row.tolist(), col.tolist()))
    assert len(pairs) == row.numel() and all((c, r) in pairs for r, c in list(pairs)[:500])   # unique, symmetric
    r2, c2, w2, y2 = synthetic.zipf_sampled(1000, 20000, seed=1)
    assert (r2 != c2).all() and int(r2.max()) < 1000 and torch.isfinite(y2).all()
    assert np.bincount(r2.numpy(), minlength=1000)[0] > np.bincount(r2.numpy(), minlength=1000)[500]


def test_event_files_are_valid_tfrecords_of_event_protos(tmp_path):
    """trainer.event_writer against its known answers: the CRC-32C check value, and protobuf's own decoder
    reading the hand-encoded Event / Summary messages."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    from trainer.event_writer import EventWriter, crc32c, encode_event, read_events
    assert crc32c(b"123456789") == 0xE3069283                       # the standard CRC-32C (Castagnoli) check value
    fdp = descriptor_pb2.FileDescriptorProto(name="ev.proto", package="t", syntax="proto3")
    val = fdp.message_type.add(name="Value")
    val.field.add(name="tag", number=1, type=9, label=1)
    val.field.add(name="simple_value", number=2, type=2, label=1)
    val.field.add(name="histo", number=5, type=11, label=1, type_name=".t.HistogramProto")
    hp = fdp.message_type.add(name="HistogramProto")
    for i, n in enumerate(("min", "max", "num", "sum", "sum_squares"), 1):
        hp.field.add(name=n, number=i, type=1, label=1)
    hp.field.add(name="bucket_limit", number=6, type=1, label=3)
    hp.field.add(name="bucket", number=7, type=1, label=3)
    fdp.message_type.add(name="Summary").field.add(name="value", number=1, type=11, label=3, type_name=".t.Value")
    ev = fdp.message_type.add(name="Event")
    ev.field.add(name="wall_time", number=1, type=1, label=1)
    ev.field.add(name="step", number=2, type=3, label=1)
    ev.field.add(name="file_version", number=3, type=9, label=1)
    ev.field.add(name="summary", number=5, type=11, label=1, type_name=".t.Summary")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fdp)
    msg = message_factory.GetMessageClass(pool.FindMessageTypeByName("t.Event"))()
    msg.ParseFromString(encode_event(12.5, 1 << 40, {"loss": 0.125, "mf/global_bias": -2.0}))
    assert (msg.wall_time, msg.step) == (12.5, 1 << 40)
    assert [(v.tag, v.simple_value) for v in msg.summary.value] == [("loss", 0.125), ("mf/global_bias", -2.0)]
    msg.ParseFromString(encode_event(1.0, file_version="brain.Event:2"))
    assert msg.file_version == "brain.Event:2"
    w = EventWriter(tmp_path)
    w.scalars(100, {"loss": 0.5, "note": "text is skipped"})
    w.scalars(200, {"loss": 0.25})
    assert [(s, sc) for _, s, sc in read_events(w.path)] == [(0, {}), (100, {"loss": 0.5}), (200, {"loss": 0.25})]
    # histograms (the reference's summary.histogram of the biases): HistogramProto over TensorFlow's default buckets
    from trainer.event_writer import default_bucket_limits, histogram_of
    lim = default_bucket_limits()
    assert len(lim) == 2 * 775 + 1 and lim[len(lim) // 2] == 0.0 and lim[-1] == 1.7976931348623157e308 and lim[0] == -lim[-1]
    assert abs(lim[len(lim) // 2 + 1] - 1e-12) < 1e-25 and abs(lim[len(lim) // 2 + 2] / lim[len(lim) // 2 + 1] - 1.1) < 1e-12
    x = np.array([-0.03, -0.03, 0.0, 0.01, 0.0105, 0.04, 2.5], np.float32)
    h = histogram_of(x)
    assert h["num"] == 7 and abs(h["sum"] - float(x.astype(np.float64).sum())) < 1e-12 and h["min"] == float(x.min())
    assert sum(h["bucket"]) == 7 and len(h["bucket"]) == len(h["bucket_limit"]) < 20       # empty runs are merged
    edges = np.asarray(h["bucket_limit"])
    for v in x.astype(np.float64):                       # every value lies in a bucket that counted it: limit[i-1] <= v < limit[i]
        i = int(np.searchsorted(edges, v, side="right"))
        assert h["bucket"][i] > 0 and (i == 0 or edges[i - 1] <= v) and v < edges[i]
    msg.ParseFromString(encode_event(3.0, 7, {"loss": 1.0}, histograms={"mf/row_biases": h}))
    got = msg.summary.value[1]
    assert got.tag == "mf/row_biases" and got.histo.num == 7 and list(got.histo.bucket) == h["bucket"]
    assert list(got.histo.bucket_limit) == h["bucket_limit"] and got.histo.sum_squares == h["sum_squares"]
    w.scalars(300, {"loss": 0.1}, {"mf/col_biases": h})
    last = list(read_events(w.path))[-1]
    assert last[1] == 300 and last[2]["loss"] == np.float32(0.1) and last[2]["mf/col_biases"] == h


def test_histogram_encoding_matches_the_bucket_by_bucket_walk():
    """histogram_of (vectorised: searchsorted over TensorFlow's default bucket limits, empty runs merged with array masks)
    against the plain walk of Histogram::EncodeToProto — a run of empty buckets becomes one entry with its last limit —
    on random data of every scale, constants, a single value and the empty tensor; tensors and arrays alike."""
    import bisect
    import torch
    from trainer.event_writer import default_bucket_limits, histogram_of
    limits = default_bucket_limits()

    def walk(values):
        counts = [0] * len(limits)
        for v in values:
            counts[min(bisect.bisect_right(limits, float(v)), len(limits) - 1)] += 1
        out_l, out_c, i = [], [], 0
        while i < len(counts):
            c, end = counts[i], limits[i]
            i += 1
            if c <= 0:
                while i < len(counts) and counts[i] <= 0:
                    end = limits[i]
                    i += 1
            out_l.append(end)
            out_c.append(float(c))
        return out_l, out_c
    rng = np.random.default_rng(0)
    cases = [rng.normal(0, s, 500).astype(np.float32) for s in (1e-9, 1e-3, 0.05, 1.0, 1e6)]
    cases += [np.zeros(7, np.float32), np.array([3.25], np.float32), np.array([], np.float32),
              np.array([-1e30, 1e30, 0.0, -0.0, 1e-13, -1e-13], np.float64), rng.uniform(-1, 1, 2000)]
    for x in cases:
        for values in (x, torch.from_numpy(np.ascontiguousarray(x))):
            h = histogram_of(values)
            want_l, want_c = walk(x)
            assert h["bucket_limit"] == want_l and h["bucket"] == want_c
            assert h["num"] == len(x) and sum(h["bucket"]) == len(x)
            if len(x):
                assert h["min"] == float(x.min()) and h["max"] == float(x.max())
                np.testing.assert_allclose(h["sum"], float(x.astype(np.float64).sum()), rtol=1e-12, atol=1e-300)
                np.testing.assert_allclose(h["sum_squares"], float((x.astype(np.float64) ** 2).sum()), rtol=1e-12)


def test_bench_gpus_n_starts_the_ranks_itself(monkeypatch, capsys):
    """`python bench.py --gpus N` without a launcher (how the driver calls it): bench.py starts N ranks as a child
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same args>`, relays rank 0's one
    JSON line and exits with the child's code — before anything touches the GPU in this process."""
    import subprocess
    import sys
    import types
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    seen = {}

    def fake_child(code, text):
        def popen(cmd, **kw):
            seen["cmd"], seen["kw"] = cmd, kw
            return types.SimpleNamespace(stdout=iter(text.splitlines(keepends=True)), wait=lambda: code)
        return popen
    # (rank 0's line is relayed as it comes — at N > 1 it is printed before the side configurations start)
    monkeypatch.setattr(subprocess, "Popen", fake_child(0, 'rank 1 says hello\n{"metric": "co-occurrence nonzeros/sec", "n_gpus": 4}\n'))
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("the parent must not touch the GPU")))
    with pytest.raises(SystemExit) as exc:
        bench.main(["--gpus", "4", "--steps", "20", "--warmup", "5"])
    assert exc.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    script = cmd.index(str(Path(bench.__file__).resolve()))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]           # the ranks get the same arguments
    out = capsys.readouterr()
    assert out.out.strip() == '{"metric": "co-occurrence nonzeros/sec", "n_gpus": 4}' and "rank 1 says hello" in out.err
    # a failing child fails the bench; more GPUs asked for than visible is refused before any launch
    monkeypatch.setattr(subprocess, "Popen", fake_child(3, ""))
    with pytest.raises(SystemExit) as exc:
        bench.main(["--gpus", "2"])
    assert exc.value.code == 3
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit, match="visible"):
        bench.main(["--gpus", "2"])
    # under a launcher (WORLD_SIZE set) the process is a rank: no second launch
    assert bench.steps_per_graph(20) == 20 and bench.steps_per_graph(200) == 50 and bench.steps_per_graph(97) == 1


def test_reference_module_names_are_aliases():
    """The reference's own command lines (Makefile:16-36,84,95,140: `python -m src.config NAME`, `python -m
    src.models.estimator`, `python -m src.data.text8`, `python -m src.models.export_embeddings`) reach this build."""
    import subprocess
    import sys
    root = str(Path(__file__).resolve().parent.parent)
    run = lambda *a: subprocess.run([sys.executable, "-m", *a], cwd=root, capture_output=True, text=True, timeout=120)
    assert run("src.config", "BATCH_SIZE").stdout.strip() == "1024" and run("src.config", "OPTIMIZER").stdout.strip() == "Adam"
    for mod, flag in (("src.models.estimator", "--embedding-size"), ("src.models.logistic_matrix_factorisation", "--neg-factor"),
                      ("src.data.text8", "--context-size"), ("src.models.export_embeddings", "--job-dir")):
        out = run(mod, "--help")
        assert out.returncode == 0 and flag in out.stdout, (mod, out.stderr[-300:])


def _fake_bench_result(name, world=1):
    """What bench.run_config / run_dealt return, with the longest strings and the most keys they produce."""
    kern = {"step": 585.123456, "index_build": 60.123456, "epoch_deal": 12.3456789, "fetch_all_to_all": 207.0, "push_all_to_all": 219.0,
            "loss_tail": 24.0, "all_reduce": 100.0}
    r = {"name": name, "metric": "co-occurrence nonzeros/sec", "value": 1.54e9, "unit": "nonzeros/s", "n_gpus": world, "steps": 200,
         "warmup": 20, "ms_per_step": 0.6805, "repeats": 3, "ms_per_step_min_max": [0.67, 0.69], "higher_is_better": True,
         "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic " + "x" * 150,
         "config": {"workload": "zipf_v400k_d300", "V": 400000, "d": 300, "optimizer": "Adagrad", "batch_size_per_gpu": 1048576,
                    "global_batch": 1048576 * world, "nnz_per_gpu": 25000000, "batches_per_epoch": 23, "chunk_cap": 32,
                    "index": "rebuilt every step: " + "y" * 400, "launch": "the trainer's runner: " + "z" * 200,
                    "parallelism": "both tables sharded x8, touched col rows by all-to-all", "chunk_records": False,
                    "chunk_run_words": True, "exchange_floats_per_rank_per_step": 123456789},
         "roofline": {"bound": "hbm", "achieved": 2400.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.2995,
                      "frac_of_measured_stream_ceiling": 0.38, "stream_ceiling": 6290.0, "traffic": 3.478e9,
                      "traffic_source": "profiles/r04_c4_v400k_d300_b1m_index_rebuilt_traffic.json", "traffic_over_algorithmic": 2.13,
                      "kernel": "one step as timed = step kernels, beside them on a side stream index_build and epoch_deal",
                      "algorithmic_bytes_per_step": 1630400000, "kernel_us": kern, "kernel_us_note": "n" * 300, "per_kernel": None,
                      "step_kernels_alone_frac": 0.348, "heavy_ids_per_step": 100.0, "uniq_rows_per_step": 167438.0,
                      "uniq_cols_per_step": 167610.0, "chunks_per_step": 400000.0},
         "masters_build_ms_at_load": 100.0, "final_loss": 1.25}
    if world > 1:
        r["collectives"] = {"phases_ms_per_step": {k: v / 1e3 for k, v in kern.items()}, "all_reduce_ms": 0.124, "all_gather_ms": 0.0, "all_to_all_ms": 0.426}
        r["process_group"] = {"world_size": world, "backend": "nccl", "rccl_version": "2.26.6", "distinct_devices": world,
                              "ranks": [{"rank": i, "cuda_device": i, "device_name": "AMD Instinct MI355X", "pci_bus_id": "0000:%02x:00.0" % i,
                                         "host": "h" * 40, "pid": 100000 + i, "visible_devices": 8} for i in range(world)]}
    return r


@pytest.mark.parametrize("world", [1, 8])
def test_bench_headline_is_one_short_line(world):
    """The driver keeps the tail of stdout: the ONE JSON line is the headline alone and stays under 4 KB whatever the run
    carried (twelve configs[] entries, eight ranks, the longest notes), with the contract's fields, `roofline` and
    `cpu_baseline` in it; every configs[] entry has its own stderr line under 1 KB."""
    import json
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench
    out = _fake_bench_result("headline", world)
    leg = {"value": 1.07e7, "unit": "nonzeros/s", "steps_per_s": 10.2, "cores": 16, "kind": "port", "sample": "s" * 150}
    out["cpu_baseline"] = dict(leg, host_cpus=256, usable_cores=16, cpu_model="AMD EPYC 9575F 64-Core Processor",
                               label="CPU restatement of yxtay/glove-tensorflow estimator step (TF 2.11 unavailable offline)",
                               legs={"adagrad_at_gpu_batch": leg, "adagrad_at_gpu_batch_one_core": dict(leg, cores=1),
                                     "c1_adam_bs1024": leg, "c1_adam_bs1024_one_core": dict(leg, cores=1)})
    out["configs"] = [bench.brief(_fake_bench_result("config_number_%d_with_a_long_name_static_index" % i, world)) for i in range(12)]
    out["configs_skipped"] = ["another_config_with_a_long_name_%d" % i for i in range(12)]
    out["config"]["same_workload_static_index"] = {"nonzeros_per_s": 1.67e9, "ms_per_step": 0.628, "index": "static, built at load (the trainer's --epoch-shuffle static)"}
    out["wall_seconds"], out["configs_file"] = 240.0, "gpurun_out/bench_configs.json"
    line = json.dumps(bench.headline(out))
    assert len(line) < 4096 and "\n" not in line
    j = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["config"]["workload"] == "zipf_v400k_d300" and "model" not in j["config"] and j["config"]["index"] == "dealt"
    assert j["config"]["same_workload_static_index"]["ms_per_step"] == 0.628
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_over_algorithmic",
              "algorithmic_bytes_per_step", "kernel_us"):
        assert k in j["roofline"], k
    assert j["roofline"]["frac"] == 0.2995 and j["cpu_baseline"]["value"] == 1.07e7 and j["cpu_baseline"]["cores"] == 16
    assert set(j["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and len(j["cpu_baseline"]["legs"]) == 4
    assert "configs" not in j and len(j["configs_run"]) == 12 and len(j["configs_skipped"]) == 12
    if world > 1:
        assert j["process_group"]["world_size"] == world and len(j["process_group"]["devices_by_rank"]) == world
        assert j["collectives"]["all_to_all_ms"] == 0.426
    for r in out["configs"]:
        assert len(json.dumps(bench.config_line(r))) < 1024


def test_bench_traffic_table_points_at_committed_profiles():
    """bench.TRAFFIC_PROFILES: every row names a committed PMC summary of that very configuration (workload, batch, index
    mode in the file's own meta), and a configuration without a row says why its traffic is null."""
    import json
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench
    for (workload, B, index), name in bench.TRAFFIC_PROFILES.items():
        meta = json.load(open(Path(bench.REPO) / "profiles" / name))["meta"]
        assert meta["workload"] == workload and int(meta["batch"]) == B and meta.get("index", "static") == index, name
        traffic, src = bench.measured_traffic(workload, B, index)
        assert traffic > 0 and src == "profiles/" + name
    traffic, why = bench.measured_traffic("text8_d64", 12345, "dealt")
    assert traffic is None and "no PMC profile" in why


def test_row_width_puts_rows_on_line_boundaries():
    """trainer.hip_api.row_width: the stored row stride — 16-byte rows at least; 64-byte boundaries when the padding costs at
    most d / 12 floats; 128-byte lines for tables beyond the Infinity Cache (>= 128 MB) — decided by the whole vocabulary, so that
    every rank of a sharded run takes the same stride."""
    from trainer.hip_api import row_width
    assert row_width(400000, 300) == 320 and row_width(50000, 300) == 304            # C4: 1,280-byte rows; C3: 1,216
    assert row_width(10000, 64) == 64 and row_width(2000000, 128) == 128              # already aligned
    assert row_width(1000, 50) == 52 and row_width(10 ** 7, 50) == 52                 # padding beyond d / 12: 16-byte rows only
    assert row_width(1000, 301) == 304 and row_width(1000, 1) == 4 and row_width(1000, 200) == 208
    for d in (1, 7, 8, 63, 100, 129, 300, 1023):
        for rows in (10, 10 ** 5, 10 ** 7):
            w = row_width(rows, d)
            assert w >= d and w % 4 == 0 and w - d <= max(3, d // 12 + 0)
