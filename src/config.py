"""`python -m src.config` (reference Makefile:16-36): alias of `trainer.config`."""
if __name__ == "__main__":
    import runpy
    runpy.run_module("trainer.config", run_name="__main__")
else:
    from trainer.config import *  # noqa: F401,F403
