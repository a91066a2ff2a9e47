"""megatron/arguments.py names (reference :24-148)."""
from emdr2_amd.arguments import get_parser, parse_args  # noqa: F401
