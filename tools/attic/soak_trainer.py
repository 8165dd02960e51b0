#!/usr/bin/env python3
"""Soak run of the real trainer loop: hundreds of thousands of steps per optimizer / head with periodic
checkpoints and eval passes, watching the loss (finite, not rising at the end) and the process's device memory."""
import json
import sys
import tempfile
from pathlib import Path

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from trainer import estimator, logistic_matrix_factorisation, synthetic  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
tmp = Path(tempfile.mkdtemp())
V = 10000
row, col, w, y = synthetic.text8_shaped(V=V, seed=0)
vocab = ["<UNK>"] + ["w%d" % i for i in range(1, V)]
(tmp / "vocab.txt").write_text("\n".join(vocab))
tok = np.asarray(vocab, dtype=object)
neg = (np.random.default_rng(0).random(len(row)) * 5).astype(np.float32)
pd.DataFrame({"row_token": tok[row.numpy()], "col_token": tok[col.numpy()], "glove_weight": w.numpy(),
              "glove_value": y.numpy(), "value": np.exp(y.numpy()), "neg_weight": neg}).to_csv(tmp / "interaction.csv", index=False)
runs = (("Adagrad", "0.05", "1024", "static", steps), ("Adam", "0.001", "1024", "static", steps),
        ("Adagrad", "0.05", "131072", "static", steps // 20), ("Adagrad", "0.05", "131072", "full", steps // 20),
        ("Adagrad", "0.05", "1024", logistic_matrix_factorisation.main, steps // 4),
        ("Adagrad", "0.05", "1024", "full", steps), ("Adam", "0.001", "4096", "full", steps // 4),
        # the fused step forms forced on (the library would take two launches at this scale): slots and twinned row table,
        # with the checkpoints and eval passes of the loop reading the tables in between
        ("Adagrad", "0.05", "131072", "form3", steps // 20), ("Adagrad", "0.05", "131072", "form4", steps // 20),
        ("Adagrad", "0.05", "4096", "form4", steps // 4))
for opt, lr, bs, entry, n in runs:
    extra = []
    if isinstance(entry, str) and entry.startswith("form"):
        entry, extra = estimator.main, ["--step-form", entry[4:]]
    if entry in ("full", "static"):       # full (the default): a new permutation every epoch, indexes prefetched, bursts from cached graphs
        entry, extra = estimator.main, ["--epoch-shuffle", entry]
    job = tmp / ("job_%s_%s_%s%s" % (opt, bs, entry.__module__.split(".")[-1], "_".join([""] + extra).replace("-", "")))
    torch.cuda.reset_peak_memory_stats()
    entry(["--train-csv", str(tmp / "interaction.csv"), "--vocab-txt", str(tmp / "vocab.txt"), "--job-dir", str(job),
           "--disable-datetime-path", "--optimizer", opt, "--learning-rate", lr, "--batch-size", bs, "--train-steps", str(n),
           "--log-every", str(max(n // 20, 1)), "--save-checkpoints-secs", "5", "--seed", "1"] + extra)
    log = [json.loads(l) for l in (job / "train_log.jsonl").read_text().splitlines()]
    losses = [r["loss"] for r in log]
    assert all(np.isfinite(losses)), losses
    print("%s%s %s bs=%s: %d steps, loss %.5f -> %.5f (min %.5f), %d checkpoints, %d evals, peak device memory %.0f MB" % (
        entry.__module__, " " + " ".join(extra) if extra else "", opt, bs, n, losses[0], losses[-1], min(losses), len(list(job.glob("model.ckpt-*.pt"))),
        len((job / "eval" / "eval_log.jsonl").read_text().splitlines()), torch.cuda.max_memory_allocated() / 1e6), flush=True)
