"""`megatron` -- drop-in names for the reference's package (megatron/__init__.py:16-41): the EMDR2 hot path of emdr2_amd behind the import
paths the reference's task code uses (`from megatron import get_args`, `from megatron.model import EMDR2Model`, ...).  Thin aliases only:
everything lives in emdr2_amd/."""
from emdr2_amd.global_vars import get_args, get_tokenizer, get_t5_tokenizer, set_args  # noqa: F401
from emdr2_amd.tasks.openqa.e2eqa.train_e2eqa import print_rank_0  # noqa: F401
