"""tasks/openqa/dense_retriever/evaluation/evaluate.py names (reference :42-134)."""
from emdr2_amd.tasks.openqa.dense_retriever.evaluation.evaluate import *  # noqa: F401,F403
