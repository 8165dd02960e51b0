#!/usr/bin/env python3
"""cProfile of the trainer's host loop (Estimator.train, Adam, the reference's default batch size): where the host spends its time
between the launches."""
import cProfile
import pstats
import sys
import tempfile
from pathlib import Path

import numpy as np
import pandas as pd

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from trainer import estimator, synthetic
tmp = Path(tempfile.mkdtemp())
V = 10000
row, col, w, y = synthetic.text8_shaped(V=V, seed=0)
vocab = ["<UNK>"] + ["w%d" % i for i in range(1, V)]
(tmp / "vocab.txt").write_text("\n".join(vocab))
tok = np.asarray(vocab, dtype=object)
pd.DataFrame({"row_token": tok[row.numpy()], "col_token": tok[col.numpy()], "glove_weight": w.numpy(), "glove_value": y.numpy()}).to_csv(tmp / "interaction.csv", index=False)
args = ["--train-csv", str(tmp / "interaction.csv"), "--vocab-txt", str(tmp / "vocab.txt"), "--job-dir", str(tmp / "job"), "--disable-datetime-path",
        "--optimizer", "Adam", "--learning-rate", "0.001", "--train-steps", "20000", "--log-every", "1000", "--skip-eval", "--seed", "1"]
pr = cProfile.Profile()
pr.enable()
estimator.main(args)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
