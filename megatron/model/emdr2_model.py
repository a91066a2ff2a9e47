"""megatron/model/emdr2_model.py names (reference :31-470)."""
from emdr2_amd.model.emdr2_model import EMDR2Model, PreComputedEvidenceDocsRetriever, OneContextLogits, emdr2_loss  # noqa: F401
