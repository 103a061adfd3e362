"""MI355X-native exhaustive rotation x translation docking search.

Package layout mirrors the reference's ``src/`` tree so that its drivers import unchanged
(``from Docker import Docker``, ``from Models import ...``, ``from src import LOG_DIR, ...``):
put this directory on ``sys.path``.

Replaces /root/reference/src/__init__.py:11-71, whose hard-coded cluster paths, mkdirs and
``assert os.path.exists(DATA_DIR)`` make ``import src`` fail anywhere else: directories come from
the environment, are created lazily, and nothing is asserted at import.
"""
import os

REPOSITORY_DIR = os.path.abspath(os.path.join(os.path.dirname(os.path.realpath(__file__)), os.pardir))
_storage = os.environ.get("DLPD_STORAGE_DIR", os.path.join(REPOSITORY_DIR, "storage"))
DATA_DIR = os.environ.get("DLPD_DATA_DIR", os.path.join(_storage, "data"))
MODELS_DIR = os.environ.get("DLPD_MODELS_DIR", os.path.join(_storage, "Models"))
LOG_DIR = os.environ.get("DLPD_LOG_DIR", os.path.join(_storage, "Experiments"))
RESULTS_DIR = os.environ.get("DLPD_RESULTS_DIR", os.path.join(REPOSITORY_DIR, "results"))

__all__ = ["REPOSITORY_DIR", "DATA_DIR", "MODELS_DIR", "LOG_DIR", "RESULTS_DIR"]
