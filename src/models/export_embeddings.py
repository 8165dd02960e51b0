"""`python -m src.models.export_embeddings` (reference Makefile:140): alias of `trainer.export_embeddings`."""
if __name__ == "__main__":
    import runpy
    runpy.run_module("trainer.export_embeddings", run_name="__main__")
else:
    from trainer.export_embeddings import *  # noqa: F401,F403
