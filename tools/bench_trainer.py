#!/usr/bin/env python3
"""Steps/s of the real trainer host loop (Estimator.train) at the reference's default batch size.
Usage: tools/bench_trainer.py [log_every (default 1000; the CLI's own default is 100)]"""
import json
import sys
import tempfile
from pathlib import Path

import numpy as np
import pandas as pd

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from trainer import estimator, synthetic  # noqa: E402

LOG_EVERY = sys.argv[1] if len(sys.argv) > 1 else "1000"
tmp = Path(tempfile.mkdtemp())
V = 10000
row, col, w, y = synthetic.text8_shaped(V=V, seed=0)
vocab = ["<UNK>"] + ["w%d" % i for i in range(1, V)]
(tmp / "vocab.txt").write_text("\n".join(vocab))
tok = np.asarray(vocab, dtype=object)
pd.DataFrame({"row_token": tok[row.numpy()], "col_token": tok[col.numpy()], "glove_weight": w.numpy(),
              "glove_value": y.numpy()}).to_csv(tmp / "interaction.csv", index=False)
for opt, lr, extra in (("Adagrad", "0.05", ["--epoch-shuffle", "static"]), ("Adam", "0.001", ["--epoch-shuffle", "static"]),
                       ("Adagrad", "0.05", ["--epoch-shuffle", "full"]), ("Adam", "0.001", ["--epoch-shuffle", "full"]),
                       ("Adagrad", "0.05", ["--epoch-shuffle", "full", "--step-form", "1"])):
    job = tmp / ("job_" + opt + "_".join(extra) + LOG_EVERY)
    estimator.main(["--train-csv", str(tmp / "interaction.csv"), "--vocab-txt", str(tmp / "vocab.txt"), "--job-dir", str(job),
                    "--disable-datetime-path", "--optimizer", opt, "--learning-rate", lr, "--train-steps", "20000",
                    "--log-every", LOG_EVERY, "--skip-eval", "--seed", "1"] + extra)
    log = [json.loads(l) for l in (job / "train_log.jsonl").read_text().splitlines()]
    print(opt, " ".join(extra), "bs=1024, a log point every " + LOG_EVERY + " steps: %.0f steps/s, %.3g nonzeros/s, loss %.4f -> %.4f" % (
        np.median([r["steps_per_sec"] for r in log[1:]]), np.median([r["nonzeros_per_sec"] for r in log[1:]]),
        log[0]["loss"], log[-1]["loss"]))
