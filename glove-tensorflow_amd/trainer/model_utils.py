"""Model-side mirror of reference src/models/model_utils.py.

`MatrixFactorisation` owns the five variables (row/col embeddings [V,d], row/col biases [V],
global bias) and their optimizer slots as device buffers; the arithmetic of `call`, of the
activity regularisers and of the optimizers lives in the HIP kernels (csrc/glove_step.hip).
"""
from __future__ import annotations

import torch

from trainer.config import EMBEDDING_SIZE, L2_REG, TOP_K


class MatrixFactorisation:
    """Reference model_utils.py:24-63 (Keras layer) as a plain parameter container."""

    def __init__(self, vocab_size, embedding_size=EMBEDDING_SIZE, l2_reg=L2_REG, optimizer="Adam", device="cuda:0",
                 seed=None, name="matrix_factorisation"):
        from trainer.hip_api import DeviceTables
        self.vocab_size, self.embedding_size, self.l2_reg, self.name = vocab_size, embedding_size, l2_reg, name
        self.tables = DeviceTables(vocab_size, embedding_size, optimizer, device=device, seed=seed)

    def get_config(self):
        return {"vocab_size": self.vocab_size, "embedding_size": self.embedding_size, "l2_reg": self.l2_reg,
                "name": self.name}


def get_named_variables(model: MatrixFactorisation) -> dict:
    """Reference model_utils.py:66-78 (the tensors, not Keras layers)."""
    t = model.tables
    return {"global_bias": t.scalars[0], "row_biases": t.br, "row_embeddings": t.embeddings("R"), "col_biases": t.bc,
            "col_embeddings": t.embeddings("C")}


def get_predictions(hip, model: MatrixFactorisation, input_ids: torch.Tensor, id_string_table: list, top_k=TOP_K):
    """PREDICT mode (reference model_utils.py:81-110): cosine similarity of the query ROW
    embeddings against all row embeddings + top_k, ids mapped back to tokens.
    Returns dict(input_string, input_embedding, top_k_similarity, top_k_string)."""
    R = model.tables.R
    ids = input_ids.to(R.device, torch.int32)
    sims, idx = hip.topk_cosine(R, ids, top_k)
    lookup = lambda i: id_string_table[i] if 0 <= i < len(id_string_table) else "<UNK>"
    return {
        "input_string": [lookup(i) for i in ids.tolist()],
        "input_embedding": model.tables.embeddings("R")[ids.long()].cpu(),
        "top_k_similarity": sims.cpu(),
        "top_k_string": [[lookup(i) for i in r] for r in idx.tolist()],
    }


def logged_biases(model: MatrixFactorisation):
    """The two bias vectors as a log point reads them: vocabularies up to 2^18 come to the host in one copy each (a handful
    of tiny device reductions, each with its own sync, costs more than the 40 KB copy), bigger ones stay on the device."""
    t = model.tables
    br, bc = t.br, t.bc
    if br.is_cuda and max(br.numel(), bc.numel()) <= 1 << 18:
        br, bc = br.cpu(), bc.cpu()
    return br, bc


def summary_values(model: MatrixFactorisation, biases=None, global_bias=None) -> dict:
    """What add_summary logs (reference model_utils.py:113-118): global bias scalar and, for the log line, the
    min/mean/max/std of the two bias vectors (their histograms go to the event file: summary_histograms)."""
    br, bc = biases if biases is not None else logged_biases(model)

    def stat(x):
        x = x.double()
        return {"min": float(x.min()), "mean": float(x.mean()), "max": float(x.max()), "std": float(x.std()) if x.numel() > 1 else 0.0}
    return {"mf/global_bias": model.tables.global_bias if global_bias is None else global_bias,
            "mf/row_biases": stat(br), "mf/col_biases": stat(bc)}


def summary_histograms(model: MatrixFactorisation, biases=None) -> dict:
    """`summary.histogram("row_biases" / "col_biases")` of add_summary (reference model_utils.py:116-117) as
    HistogramProto fields over TensorFlow's default buckets."""
    from trainer.event_writer import histogram_of
    br, bc = biases if biases is not None else logged_biases(model)
    return {"mf/row_biases": histogram_of(br), "mf/col_biases": histogram_of(bc)}
