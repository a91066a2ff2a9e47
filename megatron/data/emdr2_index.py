"""megatron/data/emdr2_index.py names (reference :16-305)."""
from emdr2_amd.data.emdr2_index import OpenRetreivalDataStore, DistributedBruteForceIndex, FaissMIPSIndex, detach  # noqa: F401
