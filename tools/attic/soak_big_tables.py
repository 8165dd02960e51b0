#!/usr/bin/env python3
"""Soak of the real trainer loop at a scale where the step is fused (V = 60 k, d = 300, B = 262,144: twinned row table; with
--epoch-shuffle full the index is rebuilt every step into staging plans that carry chunk records): thousands of steps in
both epoch modes with checkpoints and eval passes reading the tables in between; losses finite and falling, memory flat."""
import json
import sys
import tempfile
from pathlib import Path

import numpy as np
import pandas as pd
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from trainer import estimator, synthetic  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
tmp = Path(tempfile.mkdtemp())
V = 60000
row, col, w, y = synthetic.zipf_sampled(V, 1_400_000, seed=0)
vocab = ["<UNK>"] + ["w%d" % i for i in range(1, V)]
(tmp / "vocab.txt").write_text("\n".join(vocab))
tok = np.asarray(vocab, dtype=object)
pd.DataFrame({"row_token": tok[row.numpy()], "col_token": tok[col.numpy()], "glove_weight": w.numpy(),
              "glove_value": y.numpy()}).to_csv(tmp / "interaction.csv", index=False)
for mode, extra in (("full", []), ("static", []), ("full", ["--no-graphs"]), ("full", ["--step-form", "3"]), ("full", ["--batch-size", "16384", "--step-form", "4"])):
    job = tmp / ("job_%s%s" % (mode, "_".join([""] + extra).replace("-", "")))
    torch.cuda.reset_peak_memory_stats()
    estimator.main(["--train-csv", str(tmp / "interaction.csv"), "--vocab-txt", str(tmp / "vocab.txt"), "--job-dir", str(job),
                    "--disable-datetime-path", "--optimizer", "Adagrad", "--learning-rate", "0.05", "--batch-size", "262144",
                    "--embedding-size", "300", "--train-steps", str(steps), "--log-every", str(max(steps // 20, 1)),
                    "--save-checkpoints-secs", "2", "--seed", "1", "--epoch-shuffle", mode] + extra)
    log = [json.loads(l) for l in (job / "train_log.jsonl").read_text().splitlines()]
    losses = [r["loss"] for r in log]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    print("%s %s: %d steps, loss %.5f -> %.5f, %.0f steps/s, %d checkpoints, %d evals, peak device memory %.0f MB" % (
        mode, " ".join(extra), steps, losses[0], losses[-1], np.median([r["steps_per_sec"] for r in log[1:]]),
        len(list(job.glob("model.ckpt-*.pt"))), len((job / "eval" / "eval_log.jsonl").read_text().splitlines()),
        torch.cuda.max_memory_allocated() / 1e6), flush=True)
