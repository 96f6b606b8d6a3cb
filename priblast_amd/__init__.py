"""pRIblast `ris` hot path, MI355X-native: thin Python plumbing around libpriblast_hip.so."""
from . import capi  # noqa: F401
