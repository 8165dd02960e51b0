"""Command line -> `params` dict -> job directory.

Interface kept from the reference (src/models/config_utils.py:20-186): the 19 flag names and their defaults,
the `-YYYYmmdd-HHMMSS` suffix on --job-dir unless --disable-datetime-path, the copy of vocab.txt into the job
directory, and `params.json` with the derived `input_fn_args` / `dataset_args` / `serving_input_fn_args` blocks
that `trainer.export_embeddings` and a resumed run read back.  The flags are declared once in `FLAGS`; the last
group exists only in this build.
"""
from __future__ import annotations

import argparse
import json
import logging
import shutil
import sys
import time
from pathlib import Path
from typing import NamedTuple

from trainer import config

logger = logging.getLogger(__name__)


class Flag(NamedTuple):
    name: str                 # "--train-csv" without the dashes
    kind: type | None         # None: boolean switch
    default: object
    text: str


FLAGS = (
    # ---- inputs
    Flag("train-csv", str, config.TRAIN_CSV, "interaction CSV (co-occurrence nonzeros) to train and evaluate on"),
    Flag("vocab-txt", str, config.VOCAB_TXT, "vocabulary, one token per line; the line number is the id"),
    Flag("row-name", str, config.ROW_NAME, "CSV column holding the row token"),
    Flag("col-name", str, config.COL_NAME, "CSV column holding the column token"),
    Flag("target-name", str, config.TARGET_NAME, "CSV column regressed on (GloVe: log co-occurrence)"),
    Flag("weight-name", str, config.WEIGHT_NAME, "CSV column weighting the squared error"),
    Flag("pos-name", str, config.POS_NAME, "CSV column weighting the positive label (logistic heads)"),
    Flag("neg-name", str, config.NEG_NAME, "CSV column weighting the negative label (logistic heads)"),
    # ---- outputs
    Flag("job-dir", str, config.JOB_DIR, "where params.json, vocab.txt, checkpoints and logs go"),
    Flag("disable-datetime-path", None, False, "use --job-dir as given instead of appending -YYYYmmdd-HHMMSS"),
    # ---- model / optimisation
    Flag("embedding-size", int, config.EMBEDDING_SIZE, "columns of the row and column embedding tables"),
    Flag("l2-reg", float, config.L2_REG, "activity-L2 coefficient"),
    Flag("neg-factor", float, config.NEG_FACTOR, "weight of the negative head's loss (logistic heads)"),
    Flag("optimizer", str, config.OPTIMIZER, "Keras optimizer name: Adagrad or Adam"),
    Flag("learning-rate", float, config.LEARNING_RATE, "optimizer step size"),
    Flag("batch-size", int, config.BATCH_SIZE, "nonzeros per step and rank"),
    Flag("train-steps", int, config.TRAIN_STEPS, "absolute global_step to stop at"),
    Flag("steps-per-epoch", int, config.STEPS_PER_EPOCH, "accepted for compatibility; checkpoints are timed"),
    Flag("top-k", int, config.TOP_K, "neighbours returned per query token in PREDICT mode"),
    # ---- this build only
    Flag("reg-multiplicity", float, 2.0, "how often the regulariser list enters the loss: 2 under the pinned "
                                         "Keras 2.11 (`get_losses_for` returns all losses twice), 1 under TF 2.1"),
    Flag("seed", int, None, "seed of the parameter init and the stream permutation (the reference is unseeded)"),
    Flag("chunk-cap", int, 0, "nonzeros per dedup-index chunk; 0 picks it from batch and vocabulary size"),
    Flag("log-every", int, 100, "steps between two lines of train_log.jsonl"),
    Flag("save-checkpoints-secs", float, 300.0, "seconds between checkpoints (each followed by an eval pass)"),
    Flag("keep-checkpoint-max", int, 5, "checkpoints kept in the job directory"),
    Flag("skip-eval", None, False, "checkpoint without the eval pass"),
    Flag("epoch-shuffle", str, "full", "full (default, what the reference's input_fn does — make_csv_dataset reshuffles every "
                                       "epoch): every rank draws a new permutation of its pairs every epoch, the dedup "
                                       "pairs were sorted once at load, an epoch is one partition pass of both sorted orders, the index of a batch is numbered when the batch is used; static: the pairs are permuted "
                                       "once, cut into batches whose index is built at load, and every epoch visits the "
                                       "same batches in a new order (faster: no index build per step)"),
    Flag("no-graphs", None, False, "launch every step from the host instead of replaying captured hipGraphs"),
    Flag("index-segment", int, 0, "with --epoch-shuffle full: consecutive batches whose dedup index one set of launches "
                                  "builds (0: up to 64, as the epoch and 6 GB of staging plans allow)"),
    Flag("multi-rank-graphs", None, False, "replay multi-rank steps (kernels + RCCL collectives) from hipGraphs as well; "
                                           "one GPU always does"),
    Flag("row-sharded", None, False, "multi-GPU: shard the row table (and its slots) by row id % ranks and route every "
                                     "nonzero to the rank that owns its row, instead of replicating all tables "
                                     "(BASELINE config 5; Adagrad only)"),
    Flag("shard-cols", None, False, "with --row-sharded: shard the col table as well (rows and cols on id % ranks; a step fetches "
                                    "the col rows its batch touches from their owners and returns their gradients: two "
                                    "all-to-alls whose size follows the batch, not the vocabulary); --epoch-shuffle full only"),
    Flag("step-form", int, 0, "form of the single-GPU sparse Adagrad step (glove_hyper.step_form): 0 the library chooses, "
                              "1 two launches, 2 / 3 fused forms, 4 fused on a twinned row table"),
    Flag("exchange", str, "auto", "multi-GPU gradient exchange: dense (all-reduce of the [V,d] gradient buffer), rows "
                                  "(all-gather of the touched rows' summed gradients), auto (rows when that is the "
                                  "shorter payload for the resident batches)"),
)


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    for flag in FLAGS:
        if flag.kind is None:
            parser.add_argument("--" + flag.name, action="store_true", help=flag.text)
        else:
            parser.add_argument("--" + flag.name, type=flag.kind, default=flag.default, help=flag.text)
    return parser


def derived_blocks(params: dict) -> dict:
    """The three argument blocks the reference stores in params.json (config_utils.py:25-45): which CSV columns
    the input pipeline selects, what the dataset needs on top, and the string features of the serving signature."""
    tokens = [params["row_name"], params["col_name"]]
    reader = {"file_pattern": params["train_csv"], "batch_size": params["batch_size"],
              "select_columns": tokens + [params["weight_name"], params["target_name"]],
              "target_names": [params["target_name"]]}
    dataset = dict(row_col_names=tokens, vocab_txt=params["vocab_txt"], **reader, weight_names=[params["weight_name"]])
    return {"input_fn_args": reader, "dataset_args": dataset, "serving_input_fn_args": {"string_features": tokens}}


def save_params(params: dict, name: str = "params.json") -> None:
    Path(params["job_dir"], name).write_text(json.dumps(params, indent=2))


def init_params(params: dict, write: bool = True) -> dict:
    """Stamps the job directory, puts the vocabulary next to the checkpoints and records everything."""
    if not params["disable_datetime_path"]:
        params["job_dir"] = params["job_dir"] + time.strftime("-%Y%m%d-%H%M%S")
    job_dir, vocab = Path(params["job_dir"]), Path(params["vocab_txt"])
    inside = job_dir / vocab.name
    if write:
        job_dir.mkdir(parents=True, exist_ok=True)
        if vocab.resolve() != inside.resolve():
            shutil.copyfile(vocab, inside)
    params["vocab_txt"] = str(inside)            # resumed runs and the exporter read the copy
    params.update(derived_blocks(params))
    if write:
        save_params(params)
    return params


def parse_args(argv=None) -> dict:
    namespace = build_parser().parse_args(argv)
    logger.info("command line: %s", " ".join(sys.argv if argv is None else argv))
    return init_params(vars(namespace).copy())
