"""megatron/checkpointing.py names (reference :74-340)."""
from emdr2_amd.checkpointing import (get_checkpoint_name, get_checkpoint_tracker_filename, save_checkpoint, load_checkpoint,  # noqa: F401
                                     load_t5_checkpoint, load_dualencoder_checkpoint)
