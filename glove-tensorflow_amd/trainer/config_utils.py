"""Command line -> params dict -> job_dir initialisation.

Mirrors reference src/models/config_utils.py:20-186: the same 19 flags with the same defaults,
the `-%Y%m%d-%H%M%S` job_dir suffix, the vocab.txt copy into job_dir, the derived
`input_fn_args` / `dataset_args` / `serving_input_fn_args` dicts and `params.json`.
Flags after the "MI355X path" comment are additions of this build (all optional).
"""
import json
import logging
import os
import shutil
import sys
from argparse import ArgumentParser
from datetime import datetime

from trainer.config import (
    BATCH_SIZE, COL_NAME, EMBEDDING_SIZE, JOB_DIR, L2_REG, LEARNING_RATE, NEG_FACTOR, NEG_NAME, OPTIMIZER, POS_NAME,
    ROW_NAME, STEPS_PER_EPOCH, TARGET_NAME, TOP_K, TRAIN_CSV, TRAIN_STEPS, VOCAB_TXT, WEIGHT_NAME,
)

logger = logging.getLogger(__name__)


def get_function_args(params):
    row_name, col_name = params["row_name"], params["col_name"]
    target_name, weight_name = params["target_name"], params["weight_name"]
    input_fn_args = {
        "file_pattern": params["train_csv"],
        "batch_size": params["batch_size"],
        "select_columns": [row_name, col_name, weight_name, target_name],
        "target_names": [target_name],
    }
    dataset_args = {
        "row_col_names": [row_name, col_name],
        "vocab_txt": params["vocab_txt"],
        **input_fn_args,
        "weight_names": [weight_name],
    }
    serving_input_fn_args = {"string_features": [row_name, col_name]}
    return {"input_fn_args": input_fn_args, "dataset_args": dataset_args,
            "serving_input_fn_args": serving_input_fn_args}


def save_params(params, params_json="params.json"):
    with open(os.path.join(params["job_dir"], params_json), "w") as f:
        json.dump(params, f, indent=2)


def init_params(params, write=True):
    # job_dir
    if not params["disable_datetime_path"]:
        params["job_dir"] = "{job_dir}-{datetime:%Y%m%d-%H%M%S}".format(job_dir=params["job_dir"],
                                                                      datetime=datetime.now())
    if write:
        os.makedirs(params["job_dir"], exist_ok=True)
    # vocab_txt
    output_vocab_txt = os.path.join(params["job_dir"], os.path.basename(params["vocab_txt"]))
    if write and os.path.abspath(params["vocab_txt"]) != os.path.abspath(output_vocab_txt):
        shutil.copyfile(params["vocab_txt"], output_vocab_txt)
    params["vocab_txt"] = output_vocab_txt
    params.update(get_function_args(params))
    if write:
        save_params(params)
    return params


def build_parser():
    parser = ArgumentParser()
    d = " (default: %(default)s)"
    parser.add_argument("--train-csv", default=TRAIN_CSV, help="path to the training csv data" + d)
    parser.add_argument("--vocab-txt", default=VOCAB_TXT, help="path to the vocab txt" + d)
    parser.add_argument("--row-name", default=ROW_NAME, help="row id name" + d)
    parser.add_argument("--col-name", default=COL_NAME, help="column id name" + d)
    parser.add_argument("--target-name", default=TARGET_NAME, help="target name" + d)
    parser.add_argument("--weight-name", default=WEIGHT_NAME, help="weight name" + d)
    parser.add_argument("--pos-name", default=POS_NAME, help="positive name" + d)
    parser.add_argument("--neg-name", default=NEG_NAME, help="negative name" + d)
    parser.add_argument("--job-dir", default=JOB_DIR, help="job directory" + d)
    parser.add_argument("--disable-datetime-path", action="store_true",
                        help="flag whether to disable appending datetime in job_dir path" + d)
    parser.add_argument("--embedding-size", type=int, default=EMBEDDING_SIZE, help="embedding size" + d)
    parser.add_argument("--l2-reg", type=float, default=L2_REG, help="scale of l2 regularisation" + d)
    parser.add_argument("--neg-factor", type=float, default=NEG_FACTOR, help="negative loss factor" + d)
    parser.add_argument("--optimizer", default=OPTIMIZER, help="name of optimzer" + d)
    parser.add_argument("--learning-rate", type=float, default=LEARNING_RATE, help="learning rate" + d)
    parser.add_argument("--batch-size", type=int, default=BATCH_SIZE, help="batch size" + d)
    parser.add_argument("--train-steps", type=int, default=TRAIN_STEPS, help="number of training steps" + d)
    parser.add_argument("--steps-per-epoch", type=int, default=STEPS_PER_EPOCH,
                        help="number of steps per checkpoint" + d)
    parser.add_argument("--top-k", type=int, default=TOP_K, help="number of similar items" + d)
    # ---- MI355X path (not in the reference)
    parser.add_argument("--reg-multiplicity", type=float, default=2.0,
                        help="times the activity-L2 list enters the loss: 2 = keras>=2.4 `get_losses_for` "
                             "behaviour of the pinned TF 2.11, 1 = TF 2.1" + d)
    parser.add_argument("--seed", type=int, default=None, help="seed of the init and of the batch shuffle "
                        "(the reference is unseeded)" + d)
    parser.add_argument("--chunk-cap", type=int, default=0, help="max nonzeros per dedup-index chunk, 0 = choose from batch size / vocab size" + d)
    parser.add_argument("--log-every", type=int, default=100, help="steps between loss log lines" + d)
    parser.add_argument("--save-checkpoints-secs", type=float, default=300.0, help="checkpoint cadence" + d)
    parser.add_argument("--keep-checkpoint-max", type=int, default=5, help="checkpoints kept" + d)
    parser.add_argument("--skip-eval", action="store_true", help="do not run the eval pass at checkpoints" + d)
    return parser


def parse_args(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    logger.info("call: %s.", " ".join(sys.argv))
    logger.info("ArgumentParser: %s.", args.__dict__)
    return init_params(dict(args.__dict__))
