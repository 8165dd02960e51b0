"""Pins the oracle's data-side functions and the CSV loader against outputs of the REFERENCE's
own importable module (src/data/text8.py, run by tests/golden/make_text8_golden.py) and against
the known-answer rows printed in the reference README."""
import json
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

import glove_ref as ref

GOLDEN = Path(__file__).resolve().parent / "golden"

# reference README.md:48-59 (count, value, glove_weight, glove_value)
README_ROWS = [
    (24, 16.9500, 0.3428, 2.83027), (176, 74.1000, 1.0000, 4.30542), (19, 5.4500, 0.2877, 1.69562),
    (12, 5.9000, 0.2038, 1.77495), (25, 11.1667, 0.3535, 2.41293), (2312, 723.2000, 1.0000, 6.58369),
    (136, 46.5833, 1.0000, 3.84124), (18, 9.0500, 0.2763, 2.20276), (12, 5.2000, 0.2038, 1.64866),
    (35, 20.6333, 0.4550, 3.02691),
]


def test_glove_weight_matches_reference_grid():
    g = json.loads((GOLDEN / "text8_glove_weight.json").read_text())
    np.testing.assert_allclose(ref.glove_weight(g["count"]), g["glove_weight"], rtol=1e-15)


def test_readme_known_answers():
    for count, value, weight, target in README_ROWS:
        assert abs(ref.glove_weight(count) - weight) <= 1e-4       # README prints 4 digits
        assert abs(np.log(value) - target) <= 1e-4


@pytest.mark.parametrize("tag,context", [("cov90_ctx5", 5), ("cov100_ctx2", 2)])
def test_cooccurrence_matches_reference(tag, context):
    tokens = (GOLDEN / "text8_corpus.txt").read_text().split()
    vocab = (GOLDEN / ("text8_%s_vocab.txt" % tag)).read_text().split("\n")
    token2id = {t: i for i, t in enumerate(vocab)}
    ids = [token2id.get(t, 0) for t in tokens]                    # OOV -> 0 (text8.py:86)
    row, col, cnt, val = ref.cooccurrence(ids, context)
    want = pd.read_csv(GOLDEN / ("text8_%s_cooccurrence.csv" % tag))
    np.testing.assert_array_equal(row, want["row_token_id"])
    np.testing.assert_array_equal(col, want["col_token_id"])
    np.testing.assert_array_equal(cnt, want["count"])
    np.testing.assert_allclose(val, want["value"], rtol=1e-12)
    # thresholded + transformed frame (text8.py:129-135)
    keep = cnt >= 10
    inter = pd.read_csv(GOLDEN / ("text8_%s_interaction.csv" % tag), keep_default_na=False, na_filter=False)
    np.testing.assert_array_equal(row[keep], inter["row_token_id"])
    np.testing.assert_allclose(ref.glove_weight(cnt[keep]), inter["glove_weight"].astype(float), rtol=1e-12)
    np.testing.assert_allclose(np.log(val[keep]), inter["glove_value"].astype(float), rtol=1e-12)


@pytest.mark.parametrize("tag", ["cov90_ctx5", "cov100_ctx2"])
def test_csv_loader_reproduces_reference_ids(tag, tmp_path):
    """The loader's string->id lookup gives back the ids the reference wrote next to the tokens,
    also for the tokens pandas would turn into NaN ("nan", "null", "na")."""
    from trainer.data_utils import get_string_id_table, load_interaction_csv, read_vocab
    csv, vocab = GOLDEN / ("text8_%s_interaction.csv" % tag), GOLDEN / ("text8_%s_vocab.txt" % tag)
    coo = load_interaction_csv(str(csv), str(vocab), cache_dir=str(tmp_path))
    want = pd.read_csv(csv, keep_default_na=False, na_filter=False)
    np.testing.assert_array_equal(coo["row"], want["row_token_id"])
    np.testing.assert_array_equal(coo["col"], want["col_token_id"])
    np.testing.assert_allclose(coo["w"], want["glove_weight"].astype(np.float32))
    np.testing.assert_allclose(coo["y"], want["glove_value"].astype(np.float32))
    assert coo["row"].dtype == np.int32 and coo["w"].dtype == np.float32
    toks = read_vocab(vocab)
    assert {"nan", "null", "na"} <= set(toks) and len(toks) == len(set(toks))
    table = get_string_id_table(vocab)
    assert table.get("definitely-not-a-token", 0) == 0
    # second call is served from the binary COO cache
    again = load_interaction_csv(str(csv), str(vocab), cache_dir=str(tmp_path))
    assert list(tmp_path.glob("interaction-*.coo.npz")) and (again["row"] == coo["row"]).all()
    # the file is read in slabs (bounded host memory for 10^8-row files): slab boundaries change nothing
    from trainer import data_utils
    monkey = data_utils.CSV_SLAB_ROWS
    try:
        data_utils.CSV_SLAB_ROWS = 97
        slabs = load_interaction_csv(str(csv), str(vocab))
    finally:
        data_utils.CSV_SLAB_ROWS = monkey
    for k in ("row", "col", "w", "y"):
        np.testing.assert_array_equal(slabs[k], coo[k])
