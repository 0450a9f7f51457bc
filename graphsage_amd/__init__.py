"""graphsage_amd: the GraphSAGE training hot path of celestice106/graphsage, written
for MI355X (gfx950).  Host side = the reference's Python interfaces; device side =
hand-written HIP kernels behind the C ABI in include/gsage.h (libgsage.so)."""
from .graphsage import GraphSAGE, SAGEConv, create_model
from .loss import SkipGramLoss, create_loss
from .graph import GraphCSR, GraphCache, build_csr, default_cache
from .train_step import TrainStep, FlatParams, shard_bounds
from .sampler import DegreeBiasedNegativeSampler, UniformNegativeSampler
from .encoder import StructuralEncoder, model_from_checkpoint
from .walks import RandomWalkGenerator, CooccurrencePairSampler, estimate_pair_count
from .trainer import (GraphSAGETrainer, BatchGenerator, StreamingBatchGenerator, Batch,
                      split_pairs_train_val, train_graphsage)
from .datafiles import load_training_data, save_training_data, load_walks

__all__ = ["GraphSAGE", "SAGEConv", "create_model", "SkipGramLoss", "create_loss", "GraphCSR",
           "GraphCache", "build_csr", "default_cache", "TrainStep", "FlatParams", "shard_bounds", "DegreeBiasedNegativeSampler",
           "UniformNegativeSampler", "GraphSAGETrainer", "BatchGenerator", "Batch",
           "split_pairs_train_val", "StructuralEncoder", "model_from_checkpoint",
           "RandomWalkGenerator", "CooccurrencePairSampler", "estimate_pair_count",
           "StreamingBatchGenerator", "train_graphsage", "load_training_data", "save_training_data", "load_walks"]
__version__ = "0.1.0"
