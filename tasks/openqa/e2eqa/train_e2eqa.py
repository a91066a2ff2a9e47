"""tasks/openqa/e2eqa/train_e2eqa.py names (reference :72-641)."""
from emdr2_amd.tasks.openqa.e2eqa.train_e2eqa import *  # noqa: F401,F403
from emdr2_amd.tasks.openqa.e2eqa.train_e2eqa import _cross_entropy_forward_step, train, train_step  # noqa: F401
