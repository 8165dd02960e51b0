"""`python -m trainer.export_embeddings --job-dir J [--embeddings-json F]`

Reads J/params.json, restores the newest checkpoint and writes the row embeddings as JSON
`{token: {"item_id": token, "item_embedding": [...]}}`, leaving out "<UNK>" — the file the reference's exporter
produces from its PREDICT pass (reference src/models/export_embeddings.py:13-39).
"""
from __future__ import annotations

import argparse
import json
import logging
from pathlib import Path

from trainer import config

logger = logging.getLogger(__name__)
SKIPPED_TOKEN = "<UNK>"


def format_predictions(predictions) -> dict:
    """PREDICT-mode records (`input_string`, `input_embedding`, ...) -> the exported mapping."""
    table = {}
    for record in predictions:
        token = record["input_string"]
        token = token.decode() if isinstance(token, bytes) else token
        if token == SKIPPED_TOKEN:
            continue
        table[token] = {"item_id": token, "item_embedding": [float(x) for x in record["input_embedding"]]}
    logger.info("%d embeddings exported", len(table))
    return table


def main(job_dir=config.JOB_DIR, embeddings_json=config.EMBEDDINGS_JSON, **_):
    from trainer.estimator import estimator_predict
    params = json.loads(Path(job_dir, "params.json").read_text())
    target = Path(embeddings_json)
    target.parent.mkdir(parents=True, exist_ok=True)
    target.write_text(json.dumps(format_predictions(estimator_predict(params))))


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    cli = argparse.ArgumentParser(description=__doc__.splitlines()[0], formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    cli.add_argument("--job-dir", default=config.JOB_DIR, help="job directory of a finished or running training")
    cli.add_argument("--embeddings-json", default=config.EMBEDDINGS_JSON, help="file to write")
    try:
        main(**vars(cli.parse_args()))
    except KeyboardInterrupt:
        pass
