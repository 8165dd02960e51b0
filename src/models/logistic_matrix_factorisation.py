"""`python -m src.models.logistic_matrix_factorisation` (reference Makefile:95 with that MODEL_NAME): alias of `trainer.logistic_matrix_factorisation`."""
if __name__ == "__main__":
    import runpy
    runpy.run_module("trainer.logistic_matrix_factorisation", run_name="__main__")
else:
    from trainer.logistic_matrix_factorisation import *  # noqa: F401,F403
