#!/usr/bin/env python3
"""The whole pipeline at text8's real size on a stand-in corpus (there is no text8 on disk and no network): a
17,005,207-token Zipf corpus over 253,854 word types (text8's counts) written to <dir>/text8, then
`python -m trainer.text8` (GPU co-occurrence) -> `python -m trainer.estimator` -> `python -m trainer.export_embeddings`
with the reference's defaults (coverage 0.9, context 5, d = 64, batch 1,024).  Prints the wall time of every stage."""
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from trainer import estimator, export_embeddings, text8  # noqa: E402

N_TOKENS, N_TYPES = 17_005_207, 253_854
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
tmp = Path(tempfile.mkdtemp())
t0 = time.perf_counter()
rng = np.random.default_rng(0)
p = 1.0 / np.arange(1, N_TYPES + 1) ** 1.07          # text8's rank-frequency slope is close to 1
ids = rng.choice(N_TYPES, size=N_TOKENS, p=p / p.sum())
words = np.array(["w%d" % i for i in range(N_TYPES)], dtype=object)
(tmp / "text8").write_text(" ".join(words[ids]))
print("corpus: %d tokens, %d types used, %.0f MB, %.1f s" % (N_TOKENS, len(np.unique(ids)), (tmp / "text8").stat().st_size / 1e6,
                                                            time.perf_counter() - t0), flush=True)
t0 = time.perf_counter()
text8.main(url=text8.TEXT8_URL, dest=str(tmp), vocab_size=None, coverage=0.9, context_size=5)
n_vocab = len((tmp / "vocab.txt").read_text().split("\n"))
n_rows = sum(1 for _ in open(tmp / "interaction.csv")) - 1
print("trainer.text8: vocab %d, interaction.csv %d rows (%.0f MB), %.1f s" % (
    n_vocab, n_rows, (tmp / "interaction.csv").stat().st_size / 1e6, time.perf_counter() - t0), flush=True)
job = tmp / "job"
for opt, lr in (("Adam", "0.001"), ("Adagrad", "0.05")):
    t0 = time.perf_counter()
    estimator.main(["--train-csv", str(tmp / "interaction.csv"), "--vocab-txt", str(tmp / "vocab.txt"), "--job-dir", str(job / opt),
                    "--disable-datetime-path", "--optimizer", opt, "--learning-rate", lr, "--train-steps", str(steps),
                    "--log-every", str(steps // 10), "--seed", "1"])
    log = [json.loads(l) for l in (job / opt / "train_log.jsonl").read_text().splitlines()]
    ev = [json.loads(l) for l in (job / opt / "eval" / "eval_log.jsonl").read_text().splitlines()]
    print("trainer.estimator %s bs=1024: %d steps in %.1f s incl. CSV load, index build, checkpoints and %d eval passes; "
          "%.0f steps/s in the loop; loss %.4f -> %.4f; eval average_loss %.4f" % (
              opt, steps, time.perf_counter() - t0, len(ev), np.median([r["steps_per_sec"] for r in log[1:]]), log[0]["loss"],
              log[-1]["loss"], ev[-1]["average_loss"]), flush=True)
t0 = time.perf_counter()
export_embeddings.main(job_dir=str(job / "Adagrad"), embeddings_json=str(tmp / "embeddings.json"))
print("trainer.export_embeddings: %.0f MB, %.1f s" % ((tmp / "embeddings.json").stat().st_size / 1e6, time.perf_counter() - t0))
