"""megatron/training.py names the EMDR2 task imports (reference :130-230)."""
from emdr2_amd.tasks.openqa.e2eqa.train_e2eqa import setup_model_and_optimizer, train_step  # noqa: F401
