"""`python -m src.data.text8` (reference Makefile:84, README.md:33): alias of `trainer.text8`."""
if __name__ == "__main__":
    import runpy
    runpy.run_module("trainer.text8", run_name="__main__")
else:
    from trainer.text8 import *  # noqa: F401,F403
