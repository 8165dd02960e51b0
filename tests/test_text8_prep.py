"""Data prep: vocabulary on the host, co-occurrence on the GPU, checked against the
fixtures produced by the reference's own `src/data/text8.py` (tests/golden/make_text8_golden.py)."""
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

import glove_ref as ref

GOLDEN = Path(__file__).resolve().parent / "golden"
CASES = [("cov90_ctx5", 0.9, 5), ("cov100_ctx2", 0.999, 2)]


@pytest.mark.parametrize("tag,coverage,context", CASES)
def test_vocabulary_matches_reference(tag, coverage, context):
    from trainer import text8
    tokens = (GOLDEN / "text8_corpus.txt").read_text().split()
    vocab, counts, props = text8.create_vocabulary(tokens, None, coverage)
    want = pd.read_csv(GOLDEN / ("text8_%s_vocab.csv" % tag), keep_default_na=False, na_filter=False)
    assert list(counts) == list(want["count"])                       # same cutoff, same <UNK> mass
    np.testing.assert_allclose(props, want["proportion"].astype(float), rtol=1e-15)
    # same tokens; the order may only differ inside groups of equal count (pandas' unstable sort, text8.py:80)
    assert sorted(vocab) == sorted(want["token"])
    for c in np.unique(counts):
        assert set(np.asarray(vocab, object)[counts == c]) == set(want["token"][want["count"] == c])
    ids = text8.token_ids(tokens, vocab)
    assert ids.dtype == np.int32 and ids.min() >= 0 and ids.max() < len(vocab)


@pytest.mark.gpu
@pytest.mark.parametrize("n,V,context", [(0, 5, 3), (1, 5, 3), (50, 4, 1), (5000, 60, 5), (20000, 1000, 5),
                                         (3000, 7, 9), (200000, 50000, 5)])
def test_cooccurrence_kernel_vs_oracle(hip, n, V, context):
    import torch
    rng = np.random.default_rng(n + V)
    p = 1.0 / np.arange(1, V + 1)
    tok = rng.choice(V, size=n, p=p / p.sum()).astype(np.int32)
    row, col, cnt, val = hip.cooccurrence(torch.from_numpy(tok).cuda(), V, context)
    if n == 0:
        assert row.numel() == 0
        return
    w_row, w_col, w_cnt, w_val = ref.cooccurrence(tok, context)
    np.testing.assert_array_equal(row.cpu().numpy(), w_row)            # integer work: bit-exact
    np.testing.assert_array_equal(col.cpu().numpy(), w_col)
    np.testing.assert_array_equal(cnt.cpu().numpy(), w_cnt)
    np.testing.assert_allclose(val.cpu().numpy(), w_val, rtol=1e-12)
    assert (row != col).all()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,coverage,context", CASES)
def test_prep_end_to_end_matches_reference_files(hip, tag, coverage, context, tmp_path):
    """corpus -> vocab.txt + interaction.csv, compared with the files the reference module wrote."""
    from trainer import text8
    from trainer.data_utils import load_interaction_csv
    corpus = (GOLDEN / "text8_corpus.txt").read_text()
    want_vocab = (GOLDEN / ("text8_%s_vocab.txt" % tag)).read_text().split("\n")
    data = text8.process_data(corpus, None, coverage, context, hip=hip, seed=3)
    # pin the id assignment to the reference's (ties inside equal counts are its unstable sort's choice)
    remap = np.asarray([want_vocab.index(t) for t in data["vocabulary"][0]])
    text8.save_data(data, tmp_path)
    got = pd.read_csv(tmp_path / "interaction.csv", keep_default_na=False, na_filter=False)
    want = pd.read_csv(GOLDEN / ("text8_%s_interaction.csv" % tag), keep_default_na=False, na_filter=False)
    assert list(got.columns) == list(want.columns)
    got["row_token_id"], got["col_token_id"] = remap[got["row_token_id"]], remap[got["col_token_id"]]
    got = got.sort_values(["row_token_id", "col_token_id"]).reset_index(drop=True)
    assert len(got) == len(want)
    for c in ("row_token_id", "col_token_id", "count", "row_token", "col_token"):
        assert (got[c].astype(str) == want[c].astype(str)).all(), c
    for c in ("value", "neg_weight", "glove_weight", "glove_value"):
        np.testing.assert_allclose(got[c].astype(float), want[c].astype(float), rtol=1e-12, err_msg=c)
    # and the trainer's loader accepts the files as written
    coo = load_interaction_csv(str(tmp_path / "interaction.csv"), str(tmp_path / "vocab.txt"))
    assert len(coo["row"]) == len(want) and coo["row"].max() < len(want_vocab)
    assert (tmp_path / "vocab.csv").exists()
