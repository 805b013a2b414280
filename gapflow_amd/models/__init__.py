"""`from GaPFlow.models import Pressure, WallStress, BulkStress` (models/__init__.py:24-26): the device-backed views."""
from ..stress import Pressure, WallStress, BulkStress  # noqa: F401
