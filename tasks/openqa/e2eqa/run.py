"""tasks/openqa/e2eqa/run.py names (reference :9-72)."""
from emdr2_amd.tasks.openqa.e2eqa.run import main, model_provider, open_retrieval_generative_qa  # noqa: F401
