"""`python -m trainer.text8` — data prep: corpus -> vocab.txt + interaction.csv.

Same outputs and flags as the reference's `python -m src.data.text8`
(reference src/data/text8.py:162-197: --url --dest --vocab-size --coverage --context-size; files
`vocab.csv`, `vocab.txt`, `interaction.csv` with the nine columns of text8.py:100-133,156), but the
windowed co-occurrence table — an 85 M-row pandas self-join in the reference (text8.py:90-108) — is
counted on the GPU by `glove_cooccurrence_i32` (csrc/glove_cooc.hip).  Differences by design:
`--vocab-size` is parsed as int (the reference passes the string through), the row
shuffle is a seeded permutation instead of Python's per-process `hash()` (text8.py:118-123), and the
corpus is never downloaded implicitly when the machine is offline.
"""
from __future__ import annotations

import logging
import sys
from argparse import ArgumentParser
from collections import Counter
from pathlib import Path

import numpy as np

from trainer.config import CONTEXT_SIZE, COVERAGE, DATA_DIR, VOCAB_SIZE

logger = logging.getLogger(__name__)
TEXT8_URL = "http://mattmahoney.net/dc/text8.zip"     # reference configs/app.ini:26


def load_data(src_dir=DATA_DIR) -> str:
    path = Path(src_dir, "text8")
    if not path.exists():
        zipped = Path(src_dir, "text8.zip")
        if zipped.exists():
            from zipfile import ZipFile
            with ZipFile(zipped) as zf:
                zf.extractall(src_dir)
        else:
            raise FileNotFoundError(
                "%s not found. Fetch %s on a connected machine and unzip it into %s." % (path, TEXT8_URL, src_dir))
    return path.read_text()


def create_vocabulary(text_tokens, vocab_size=VOCAB_SIZE, coverage=COVERAGE):
    """Tokens whose count reaches the count at which the cumulative token mass hits `coverage`
    (at most `vocab_size` of them), plus "<UNK>" carrying the rest; sorted by count, descending
    (reference text8.py:61-81).  Returns (tokens, counts, proportions) as lists/arrays in id order."""
    counter = Counter(text_tokens)
    by_count = np.sort(np.fromiter(counter.values(), dtype=np.int64))[::-1]
    total = int(by_count.sum())
    cutoff = by_count[np.searchsorted(np.cumsum(by_count) / total, coverage)]
    kept = [(tok, c) for tok, c in counter.most_common(vocab_size) if c >= cutoff]
    tokens = ["<UNK>"] + [t for t, _ in kept]
    counts = np.asarray([total - sum(c for _, c in kept)] + [c for _, c in kept], dtype=np.int64)
    order = np.argsort(-counts, kind="stable")        # ties keep most_common order
    tokens = [tokens[i] for i in order]
    counts = counts[order]
    return tokens, counts, counts / total


def token_ids(text_tokens, vocab_tokens) -> np.ndarray:
    """id = position in the vocabulary, out-of-vocabulary -> 0 (text8.py:85-86)."""
    lookup = {tok: i for i, tok in enumerate(vocab_tokens)}
    return np.fromiter((lookup.get(t, 0) for t in text_tokens), dtype=np.int32, count=len(text_tokens))


def glove_weight(values, alpha=0.75, x_max=100):
    """text8.py:138-139."""
    return np.clip(np.power(np.asarray(values, dtype=np.float64) / x_max, alpha), 0, 1)


def create_interaction_table(ids: np.ndarray, vocab, context_size=CONTEXT_SIZE, hip=None, device="cuda:0", seed=0):
    """Symmetrised window co-occurrence table with the vocabulary columns joined in
    (text8.py:84-126), rows in a seeded random order.  Returns a dict of numpy columns."""
    import torch
    if hip is None:
        from trainer.hip_api import GloveHip
        hip = GloveHip(device)
    tokens, counts, proportions = vocab
    row, col, cnt, val = hip.cooccurrence(torch.from_numpy(ids).to(hip.device), len(tokens), context_size)
    row, col = row.cpu().numpy(), col.cpu().numpy()
    cnt, val = cnt.cpu().numpy(), val.cpu().numpy()
    perm = np.random.default_rng(seed).permutation(len(row))
    row, col, cnt, val = row[perm], col[perm], cnt[perm], val[perm]
    tok = np.asarray(tokens, dtype=object)
    return {"row_token_id": row.astype(np.int64), "col_token_id": col.astype(np.int64), "count": cnt, "value": val,
            "row_token": tok[row], "col_token": tok[col],
            "neg_weight": counts[row] * proportions[col]}          # text8.py:115


def create_glove_table(table: dict, count_minimum=10) -> dict:
    """count >= 10, glove_weight from the COUNT, glove_value = ln(VALUE) (text8.py:129-135)."""
    keep = table["count"] >= count_minimum
    out = {k: v[keep] for k, v in table.items()}
    out["glove_weight"] = glove_weight(out["count"])
    out["glove_value"] = np.log(out["value"])
    return out


def process_data(text8: str, vocab_size=VOCAB_SIZE, coverage=COVERAGE, context_size=CONTEXT_SIZE, hip=None, seed=0):
    text_tokens = text8.split()
    vocab = create_vocabulary(text_tokens, vocab_size, coverage)
    logger.info("vocab created, size: %s.", len(vocab[0]))
    table = create_interaction_table(token_ids(text_tokens, vocab[0]), vocab, context_size, hip=hip, seed=seed)
    logger.info("interaction table: %d pairs.", len(table["count"]))
    table = create_glove_table(table)
    logger.info("after the count filter: %d pairs.", len(table["count"]))
    return {"vocabulary": vocab, "interaction": table}


def save_data(data, save_dir=DATA_DIR):
    """vocab.csv / vocab.txt / interaction.csv exactly as text8.py:142-159 lays them out."""
    import pandas as pd
    save_dir = Path(save_dir)
    save_dir.mkdir(parents=True, exist_ok=True)
    tokens, counts, proportions = data["vocabulary"]
    pd.DataFrame({"token": tokens, "count": counts, "proportion": proportions}).to_csv(save_dir / "vocab.csv", index=False)
    (save_dir / "vocab.txt").write_text("\n".join(tokens))
    columns = ["row_token_id", "col_token_id", "count", "value", "row_token", "col_token", "neg_weight",
               "glove_weight", "glove_value"]
    pd.DataFrame({c: data["interaction"][c] for c in columns}).to_csv(save_dir / "interaction.csv", index=False)
    logger.info("saved vocab.csv, vocab.txt, interaction.csv to %s", save_dir)
    return data


def main(url, dest, vocab_size, coverage, context_size, seed=0, **kwargs):
    text8 = load_data(dest)
    save_data(process_data(text8, vocab_size, coverage, context_size, seed=seed), dest)


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    parser = ArgumentParser(description="Prepare text8 data (vocabulary + co-occurrence) on the GPU.")
    d = " (default: %(default)s)"
    parser.add_argument("--url", default=TEXT8_URL, help="url of text8 data, informational only" + d)
    parser.add_argument("--dest", default=DATA_DIR, help="directory holding `text8` and receiving the outputs" + d)
    parser.add_argument("--vocab-size", type=int, default=VOCAB_SIZE, help="maximum size of vocab" + d)
    parser.add_argument("--coverage", type=float, default=COVERAGE, help="token coverage to set token count cutoff" + d)
    parser.add_argument("--context-size", type=int, default=CONTEXT_SIZE, help="size of context window" + d)
    parser.add_argument("--seed", type=int, default=0, help="seed of the row shuffle" + d)
    args = parser.parse_args()
    logger.info("call: %s.", " ".join(sys.argv))
    try:
        main(**args.__dict__)
    except KeyboardInterrupt:
        pass
