"""megatron/model/search_strategy.py names (reference :185-240)."""
from emdr2_amd.model.search_strategy import *  # noqa: F401,F403
