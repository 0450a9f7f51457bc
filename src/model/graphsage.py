"""`src.model.graphsage` import path of the reference -> graphsage_amd.graphsage."""
from graphsage_amd.graphsage import GraphSAGE, SAGEConv, create_model  # noqa: F401
