"""megatron/global_vars.py names (reference :35-110)."""
from emdr2_amd.global_vars import get_args, get_tokenizer, get_t5_tokenizer, set_args  # noqa: F401
