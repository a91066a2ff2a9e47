"""tasks/openqa/e2eqa/async_indexer.py names (reference :84-144)."""
from emdr2_amd.tasks.openqa.e2eqa.async_indexer import AsyncIndexBuilder  # noqa: F401
