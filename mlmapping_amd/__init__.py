"""mlmapping_amd — MI355X-native map-update path of MLMapping."""
from .config import MapConfig, PRESETS, S1, S3, SDEF, S1_SIGMA0  # noqa: F401
