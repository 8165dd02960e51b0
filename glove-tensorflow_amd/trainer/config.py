"""Defaults of the command lines, with the reference's optional INI override.

The reference keeps every default in `configs/app.ini` and picks a section with $ENVIRONMENT
(reference src/config.py:8-54, configs/app.ini).  Here the defaults live in this module (one typed table);
an INI file is only read if the user supplies one, with the reference's conventions, so that an existing
deployment keeps working:

  * `configs/app.ini` relative to the current directory (or the file named by $GLOVE_APP_INI),
  * section = $ENVIRONMENT (default "dev") on top of `[DEFAULT]`,
  * keys = the upper-case names below; unknown keys are ignored.

Every entry is exported as a module constant (`trainer.config.BATCH_SIZE` …) and through `SETTINGS`.
`python -m trainer.config KEY` prints one value, as the reference's module does.
"""
from __future__ import annotations

import configparser
import os
import sys
from pathlib import Path

# name -> (type, built-in value).  Path-like defaults are derived below so that overriding a directory moves them.
_TABLE = {
    "DATA_DIR": (str, "data"),
    "CHECKPOINTS_DIR": (str, "checkpoints"),
    "MODEL_NAME": (str, "estimator"),
    # data prep (trainer.text8)
    "COVERAGE": (float, 0.9),
    "CONTEXT_SIZE": (int, 5),
    # CSV columns
    "ROW_NAME": (str, "row_token"),
    "COL_NAME": (str, "col_token"),
    "TARGET_NAME": (str, "glove_value"),
    "WEIGHT_NAME": (str, "glove_weight"),
    "POS_NAME": (str, "value"),
    "NEG_NAME": (str, "neg_weight"),
    # model and optimisation
    "EMBEDDING_SIZE": (int, 64),
    "L2_REG": (float, 0.01),
    "NEG_FACTOR": (float, 1.0),
    "OPTIMIZER": (str, "Adam"),
    "LEARNING_RATE": (float, 0.001),
    "BATCH_SIZE": (int, 1024),
    "TRAIN_STEPS": (int, 16384),
    "STEPS_PER_EPOCH": (int, 16384),
    "TOP_K": (int, 20),
}
# what the reference's [dev] / [prod] sections change (configs/app.ini:55-59)
_ENVIRONMENTS = {"dev": {"TRAIN_STEPS": 1024}, "prod": {"TRAIN_STEPS": 65536}}
_DERIVED = {
    "JOB_DIR": lambda s: os.path.join(s["CHECKPOINTS_DIR"], s["MODEL_NAME"]),
    "TRAIN_CSV": lambda s: os.path.join(s["DATA_DIR"], "interaction.csv"),
    "VOCAB_TXT": lambda s: os.path.join(s["DATA_DIR"], "vocab.txt"),
    "EMBEDDINGS_JSON": lambda s: os.path.join(s["CHECKPOINTS_DIR"], "embeddings.json"),
}


def load_settings(environment: str | None = None, ini_path: str | os.PathLike | None = None) -> dict:
    environment = environment or os.environ.get("ENVIRONMENT", "dev")
    values = {name: default for name, (_, default) in _TABLE.items()}
    values.update(_ENVIRONMENTS.get(environment, {}))
    ini = Path(ini_path or os.environ.get("GLOVE_APP_INI", Path("configs", "app.ini")))
    explicit = set()
    if ini.is_file():
        parser = configparser.ConfigParser()
        parser.read(ini)
        if environment not in parser and environment not in _ENVIRONMENTS:
            raise KeyError("ENVIRONMENT=%r is neither a section of %s nor a built-in environment" % (environment, ini))
        section = parser[environment] if environment in parser else parser["DEFAULT"]
        for name in list(_TABLE) + list(_DERIVED):
            if name in section:
                kind = _TABLE[name][0] if name in _TABLE else str
                values[name] = kind(section[name])
                explicit.add(name)
    elif environment not in _ENVIRONMENTS:
        raise KeyError("ENVIRONMENT=%r: no such built-in environment and no INI file at %s" % (environment, ini))
    for name, rule in _DERIVED.items():
        if name not in explicit:
            values[name] = rule(values)
    values["VOCAB_SIZE"] = None          # trainer.text8: no cap unless --vocab-size is given
    return values


SETTINGS = load_settings()
globals().update(SETTINGS)               # JOB_DIR, TRAIN_CSV, BATCH_SIZE, ... as module constants

if __name__ == "__main__":
    if len(sys.argv) != 2 or sys.argv[1] not in SETTINGS:
        sys.exit("usage: python -m trainer.config KEY   (one of: %s)" % ", ".join(sorted(SETTINGS)))
    sys.stdout.write(str(SETTINGS[sys.argv[1]]))
    sys.stdout.flush()
