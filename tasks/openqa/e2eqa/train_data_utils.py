"""tasks/openqa/e2eqa/train_data_utils.py names (reference :27-173)."""
from emdr2_amd.tasks.openqa.e2eqa.train_data_utils import *  # noqa: F401,F403
