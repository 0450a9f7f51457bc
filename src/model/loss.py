"""`src.model.loss` import path of the reference -> graphsage_amd.loss."""
from graphsage_amd.loss import SkipGramLoss, create_loss  # noqa: F401
