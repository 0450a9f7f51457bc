"""Drop-in for the reference package `src.model` (its __init__ re-exports these names at
src/model/__init__.py:34-40): same import paths, MI355X kernels underneath."""
from graphsage_amd.graphsage import GraphSAGE, SAGEConv, create_model
from graphsage_amd.loss import SkipGramLoss, create_loss

__all__ = ["GraphSAGE", "SAGEConv", "create_model", "SkipGramLoss", "create_loss"]
