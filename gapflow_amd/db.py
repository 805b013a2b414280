"""`from GaPFlow.db import Database` (db.py:46) -> the in-memory training database of gapflow_amd.gp."""
from .gp import Database  # noqa: F401
