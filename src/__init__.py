"""Import-path shim: a tree that did `from src.model import GraphSAGE, SkipGramLoss`
(the reference layout) gets the MI355X drop-ins from graphsage_amd."""
