"""megatron/indexer_emdr2.py names (reference :77-114)."""
from emdr2_amd.indexer_emdr2 import IndexBuilder  # noqa: F401
