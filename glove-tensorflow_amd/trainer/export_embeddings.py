"""`python -m trainer.export_embeddings --job-dir J` (reference src/models/export_embeddings.py):
load J/params.json, restore the latest checkpoint, run PREDICT over the vocabulary and write
{token: {"item_id", "item_embedding"}} skipping "<UNK>" (export_embeddings.py:13-26,39)."""
import json
import logging
import os
from argparse import ArgumentParser

from trainer.config import EMBEDDINGS_JSON, JOB_DIR

logger = logging.getLogger(__name__)


def format_predictions(predictions):
    embeddings = {}
    for instance in predictions:
        item_id = instance["input_string"]
        if isinstance(item_id, bytes):
            item_id = item_id.decode()
        if item_id != "<UNK>":
            embeddings[item_id] = {"item_id": item_id, "item_embedding": instance["input_embedding"].tolist()}
    logger.info("embedding dict size: %s.", len(embeddings))
    return embeddings


def main(job_dir=JOB_DIR, embeddings_json=EMBEDDINGS_JSON, **kwargs):
    from trainer.estimator import estimator_predict
    with open(os.path.join(job_dir, "params.json")) as f:
        params = json.load(f)
    embeddings = format_predictions(estimator_predict(params))
    os.makedirs(os.path.dirname(os.path.abspath(embeddings_json)), exist_ok=True)
    with open(embeddings_json, "w") as f:
        json.dump(embeddings, f)


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    parser = ArgumentParser()
    parser.add_argument("--job-dir", default=JOB_DIR, help="job directory (default: %(default)s)")
    parser.add_argument("--embeddings-json", default=EMBEDDINGS_JSON,
                        help="path to the embeddings json (default: %(default)s)")
    args = parser.parse_args()
    try:
        main(**args.__dict__)
    except KeyboardInterrupt:
        pass
