"""`python -m src.models.estimator` (reference Makefile:95 with MODEL_NAME = estimator, README.md:92): alias of `trainer.estimator`."""
if __name__ == "__main__":
    import runpy
    runpy.run_module("trainer.estimator", run_name="__main__")
else:
    from trainer.estimator import *  # noqa: F401,F403
