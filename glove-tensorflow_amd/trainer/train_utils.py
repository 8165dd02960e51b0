"""Training plumbing: optimizer selection and the job_dir checkpoint layout.

Mirrors reference src/models/train_utils.py: `get_optimizer` picks the optimizer by its Keras
name with only the learning rate set (:13-16); RunConfig(save_checkpoints_secs=300) (:26-27);
TrainSpec(max_steps) is an ABSOLUTE global_step (:39-40).  Checkpoints keep the Estimator's file
naming (`checkpoint` state file + `model.ckpt-<global_step>`), the payload is a torch file
because the TF bundle format cannot be written without TensorFlow.
"""
from __future__ import annotations

import logging
import os
import re
import time

import torch

logger = logging.getLogger(__name__)
EVAL_INTERVAL = 300

# Keras-legacy defaults of the optimizers the HIP path implements
OPTIMIZERS = {
    "Adagrad": {"initial_accumulator_value": 0.1, "epsilon": 1e-7},
    "Adam": {"beta_1": 0.9, "beta_2": 0.999, "epsilon": 1e-7},
    "SGD": {"momentum": 0.0, "nesterov": False},
    "RMSprop": {"rho": 0.9, "momentum": 0.0, "epsilon": 1e-7, "centered": False},
    "Adamax": {"beta_1": 0.9, "beta_2": 0.999, "epsilon": 1e-7},
    "Nadam": {"beta_1": 0.9, "beta_2": 0.999, "epsilon": 1e-7, "schedule_decay": 0.004},
    "Adadelta": {"rho": 0.95, "epsilon": 1e-7},
    "Ftrl": {"learning_rate_power": -0.5, "initial_accumulator_value": 0.1, "l1_regularization_strength": 0.0,
             "l2_regularization_strength": 0.0, "l2_shrinkage_regularization_strength": 0.0, "beta": 0.0},
}


def get_optimizer(optimizer_name="Adam", **kwargs) -> dict:
    """{"class_name", "config"} like `tf.keras.optimizers.get` takes (train_utils.py:13-16).
    Only the optimizers with a HIP apply kernel are accepted; the name match is Keras'
    (case-insensitive)."""
    for known, defaults in OPTIMIZERS.items():
        if optimizer_name.lower() == known.lower():
            return {"class_name": known, "config": {**defaults, **kwargs}}
    raise ValueError("optimizer %r has no HIP kernel (implemented: %s)" % (optimizer_name, ", ".join(OPTIMIZERS)))


class CheckpointManager:
    """`model_dir` layout of tf.estimator: `checkpoint` (text state file) + `model.ckpt-<step>`."""

    PREFIX = "model.ckpt-"

    def __init__(self, model_dir, save_secs=EVAL_INTERVAL, keep_max=5):
        self.model_dir, self.save_secs, self.keep_max = model_dir, min(save_secs, 300), keep_max
        self._last_save = time.monotonic()

    def _state_file(self):
        return os.path.join(self.model_dir, "checkpoint")

    def all_checkpoints(self) -> list:
        path = self._state_file()
        if not os.path.exists(path):
            return []
        names = re.findall(r'all_model_checkpoint_paths: "([^"]+)"', open(path).read())
        return [n for n in names if os.path.exists(os.path.join(self.model_dir, n + ".pt"))]

    def latest(self):
        path = self._state_file()
        if not os.path.exists(path):
            return None
        m = re.search(r'^model_checkpoint_path: "([^"]+)"', open(path).read(), re.M)
        if not m:
            return None
        f = os.path.join(self.model_dir, m.group(1) + ".pt")
        return f if os.path.exists(f) else None

    def due(self) -> bool:
        return time.monotonic() - self._last_save >= self.save_secs

    def save(self, tables, extra=None, state=None) -> str:
        """`state`: the state dict to store instead of tables.state_dict() (a row-sharded run hands over the
        gathered whole-model dict, so the file looks like any other checkpoint)."""
        step = tables.global_step
        name = "%s%d" % (self.PREFIX, step)
        os.makedirs(self.model_dir, exist_ok=True)
        tmp = os.path.join(self.model_dir, name + ".pt.tmp")
        torch.save({"tables": state if state is not None else tables.state_dict(), "extra": extra or {}}, tmp)
        os.replace(tmp, os.path.join(self.model_dir, name + ".pt"))
        names = [n for n in self.all_checkpoints() if n != name] + [name]
        for old in names[:-self.keep_max]:
            try:
                os.remove(os.path.join(self.model_dir, old + ".pt"))
            except OSError:
                pass
        names = names[-self.keep_max:]
        with open(self._state_file() + ".tmp", "w") as f:
            f.write('model_checkpoint_path: "%s"\n' % name)
            for n in names:
                f.write('all_model_checkpoint_paths: "%s"\n' % n)
        os.replace(self._state_file() + ".tmp", self._state_file())
        self._last_save = time.monotonic()
        logger.info("saved checkpoint %s", name)
        return name

    def restore(self, tables, shard=None) -> bool:
        """`shard = (world, rank)`: the tables hold one rank's rows of a row-sharded run; checkpoints always hold
        the whole model."""
        f = self.latest()
        if f is None:
            return False
        blob = torch.load(f, map_location="cpu", weights_only=False)
        if shard is None:
            tables.load_state_dict(blob["tables"])
        else:
            tables.load_whole_state_dict(blob["tables"], *shard)
        logger.info("restored %s (global_step %d)", f, tables.global_step)
        return True
