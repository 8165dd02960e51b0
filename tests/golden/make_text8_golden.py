"""Generates the committed golden fixtures by IMPORTING the reference's TF-free data module
(`/root/reference/src/data/text8.py`) in this container.  The reference itself never travels:
only the inputs and expected outputs written here (plain data) are committed.

Recipe: `src.config` reads `configs/app.ini` relative to the cwd and the
reference logger writes `main.log` into the cwd, so run from a scratch directory holding a
`configs` symlink:

    python tests/golden/make_text8_golden.py        # rewrites tests/golden/text8_*.{json,csv,txt}

Environment: pandas 2.3.3 / numpy 2.2.6 here (the reference pins pandas 1.1.5 / numpy 1.21.6).
PYTHONHASHSEED is irrelevant to the fixtures: rows are re-sorted by (row_token_id, col_token_id)
before writing because the reference orders them by Python's per-process `hash()` (text8.py:118-123).
"""
import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REFERENCE = Path("/root/reference")


def synthetic_corpus(n_tokens=5000, n_types=60, seed=0):
    rng = np.random.default_rng(seed)
    p = 1.0 / np.arange(1, n_types + 1)
    p /= p.sum()
    # includes the tokens pandas would parse as NA: the loader must keep them as strings
    words = ["nan", "null", "na", "the", "of"] + ["w%02d" % i for i in range(n_types - 5)]
    return [words[i] for i in rng.choice(n_types, size=n_tokens, p=p)]


def main():
    scratch = Path(tempfile.mkdtemp(prefix="glove_golden_"))
    os.symlink(REFERENCE / "configs", scratch / "configs")
    os.chdir(scratch)
    sys.path.insert(0, str(REFERENCE))
    from src.data import text8  # noqa: E402  (the reference module, imported read-only)

    # 1. glove_weight on a count grid (text8.py:138-139)
    counts = [1, 2, 5, 9, 10, 11, 12, 18, 19, 24, 25, 35, 50, 99, 100, 101, 136, 176, 1000, 2312, 100000]
    weights = text8.glove_weight(np.asarray(counts, dtype=np.int64)).tolist()
    (HERE / "text8_glove_weight.json").write_text(json.dumps({"count": counts, "glove_weight": weights}, indent=1))

    # 2. vocabulary + interaction frames of a small corpus (text8.py:46-58)
    tokens = synthetic_corpus()
    for tag, kwargs in (("cov90_ctx5", dict(coverage=0.9, context_size=5)),
                        ("cov100_ctx2", dict(coverage=0.999, context_size=2))):
        data = text8.process_data(" ".join(tokens), vocab_size=None, **kwargs)
        df_vocab, df = data["vocabulary"], data["interaction"]
        df = df.sort_values(["row_token_id", "col_token_id"]).reset_index(drop=True)
        (HERE / ("text8_%s_vocab.txt" % tag)).write_text("\n".join(df_vocab["token"]))
        df_vocab.to_csv(HERE / ("text8_%s_vocab.csv" % tag), index=False)
        df.to_csv(HERE / ("text8_%s_interaction.csv" % tag), index=False)
        # the un-thresholded co-occurrence table (before count>=10) pins the windowing itself
        df_all = text8.create_interaction_dataframe(tokens, df_vocab, kwargs["context_size"])
        df_all = df_all.sort_values(["row_token_id", "col_token_id"]).reset_index(drop=True)
        df_all[["row_token_id", "col_token_id", "count", "value"]].to_csv(
            HERE / ("text8_%s_cooccurrence.csv" % tag), index=False)
    (HERE / "text8_corpus.txt").write_text(" ".join(tokens))
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
