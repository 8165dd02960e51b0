"""Static configuration: INI `[DEFAULT]` + the section named by $ENVIRONMENT (default "dev").

Mirrors reference src/config.py:8-54: same constant names, same defaults (configs/app.ini).
A `configs/app.ini` in the current directory wins (the reference reads it relative to the cwd,
config.py:10); otherwise the copy shipped with this package is used.
"""
import os
import sys
from argparse import ArgumentParser
from configparser import ConfigParser
from pathlib import Path

PACKAGE_INI = Path(__file__).resolve().parent.parent / "configs" / "app.ini"


def read_config(ini_file="app.ini", environment=None):
    environment = environment or os.environ.get("ENVIRONMENT", "dev")
    parser = ConfigParser()
    cwd_ini = Path("configs", ini_file)
    parser.read([PACKAGE_INI, cwd_ini] if cwd_ini.exists() else [PACKAGE_INI])
    if environment not in parser:
        raise KeyError("ENVIRONMENT=%r is not a section of %s" % (environment, ini_file))
    return parser[environment]


CONFIG = read_config()

# paths
JOB_DIR = CONFIG["JOB_DIR"]

# files
TRAIN_CSV = CONFIG["TRAIN_CSV"]
VOCAB_TXT = CONFIG["VOCAB_TXT"]
EMBEDDINGS_JSON = CONFIG["EMBEDDINGS_JSON"]

# preprocess
DATA_DIR = CONFIG["DATA_DIR"]
VOCAB_SIZE = None
COVERAGE = CONFIG.getfloat("COVERAGE")
CONTEXT_SIZE = CONFIG.getint("CONTEXT_SIZE")

# data
ROW_NAME = CONFIG["ROW_NAME"]
COL_NAME = CONFIG["COL_NAME"]
TARGET_NAME = CONFIG["TARGET_NAME"]
WEIGHT_NAME = CONFIG["WEIGHT_NAME"]
POS_NAME = CONFIG["POS_NAME"]
NEG_NAME = CONFIG["NEG_NAME"]

# model
EMBEDDING_SIZE = CONFIG.getint("EMBEDDING_SIZE")
L2_REG = CONFIG.getfloat("L2_REG")
NEG_FACTOR = CONFIG.getfloat("NEG_FACTOR")
OPTIMIZER = CONFIG["OPTIMIZER"]
LEARNING_RATE = CONFIG.getfloat("LEARNING_RATE")   # the reference keeps the string and lets argparse convert it
BATCH_SIZE = CONFIG.getint("BATCH_SIZE")
TRAIN_STEPS = CONFIG.getint("TRAIN_STEPS")
STEPS_PER_EPOCH = CONFIG.getint("STEPS_PER_EPOCH")
TOP_K = CONFIG.getint("TOP_K")

if __name__ == "__main__":
    parser = ArgumentParser()
    parser.add_argument("key", help="key name to get value")
    args = parser.parse_args()
    sys.stdout.write(CONFIG[args.key])
    sys.stdout.flush()
