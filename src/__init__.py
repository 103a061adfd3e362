"""Alias so the reference's drivers' ``from src import LOG_DIR, MODELS_DIR, DATA_DIR`` and
``from src import REPOSITORY_DIR`` resolve to this build's configuration."""
from deeplocalproteindocking_amd import DATA_DIR, LOG_DIR, MODELS_DIR, REPOSITORY_DIR, RESULTS_DIR  # noqa: F401
