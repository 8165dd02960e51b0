"""Module names of the reference (`python -m src.models.estimator`, `python -m src.data.text8`, `python -m src.config
NAME`: reference Makefile:16-36,84,95,140) as aliases of the `trainer` package, so that the reference's own command lines
and Makefile recipes run this build unchanged."""
