"""Training harness: counterpart of ``GraphSAGETrainer`` (``src/training/trainer.py:40-405``)
and of ``BatchGenerator`` / ``split_pairs_train_val`` (``src/training/batch_generator.py``),
kept on the device.

The reference holds the positive pairs as a Python list, shuffles a Python index list per
epoch and builds every batch with two list comprehensions and ``torch.tensor(...)``
(``batch_generator.py:147-194``), then pays seven ``.item()`` syncs per step for its running
means (``trainer.py:243-245``).  Here the pairs are two resident int64 tensors, an epoch is one
device permutation, a batch is two index_selects, and the running sums stay on the device
until the epoch ends (one 32-byte copy per epoch).
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass
from typing import Any, Dict, Iterator, List, Optional, Sequence, Tuple, Union

import torch

from .loss import DETAIL_KEYS
from .train_step import TrainStep

Pairs = Union[Sequence[Tuple[int, int]], Tuple[torch.Tensor, torch.Tensor], torch.Tensor]


def pairs_to_tensors(pairs: Pairs, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """Accepts the reference's list of (target, context) tuples, a [P,2] / [2,P] tensor, or a
    (targets, contexts) pair of tensors (the layout of ``pairs.pt``, generate_data.py:190-193)."""
    if isinstance(pairs, dict):
        t, c = pairs["targets"], pairs["contexts"]
    elif isinstance(pairs, tuple) and len(pairs) == 2 and torch.is_tensor(pairs[0]):
        t, c = pairs
    elif torch.is_tensor(pairs):
        t, c = (pairs[:, 0], pairs[:, 1]) if pairs.size(-1) == 2 else (pairs[0], pairs[1])
    else:
        arr = torch.as_tensor(list(pairs), dtype=torch.int64).reshape(-1, 2)
        t, c = arr[:, 0], arr[:, 1]
    return (t.to(device=device, dtype=torch.int64).contiguous(),
            c.to(device=device, dtype=torch.int64).contiguous())


def split_pairs_train_val(targets: torch.Tensor, contexts: torch.Tensor, val_fraction: float = 0.1,
                          seed: int = 42):
    """Shuffle once with a seeded generator and cut at ``int(P * (1 - val_fraction))``
    (``batch_generator.py:312-336``; the permutation comes from torch, not Python's random)."""
    n = targets.numel()
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    perm = torch.randperm(n, generator=g).to(targets.device)
    split = int(n * (1 - val_fraction))
    tr, va = perm[:split], perm[split:]
    return (targets[tr], contexts[tr]), (targets[va], contexts[va])


@dataclass
class Batch:
    """``batch_generator.py:44-67``."""
    targets: torch.Tensor
    contexts: torch.Tensor
    negatives: torch.Tensor
    size: int

    def to(self, device) -> "Batch":
        return Batch(self.targets.to(device), self.contexts.to(device), self.negatives.to(device),
                     self.size)


class BatchGenerator:
    """Device-side counterpart of ``BatchGenerator`` (``batch_generator.py:73-202``): every
    pair once per epoch in a fresh random order, fresh negatives per batch, last batch smaller
    unless ``drop_last``."""

    def __init__(self, positive_pairs: Pairs, negative_sampler, batch_size: int = 512,
                 num_negatives: int = 5, device=None, drop_last: bool = False,
                 shuffle: bool = True):
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        self.targets, self.contexts = pairs_to_tensors(positive_pairs, self.device)
        self.negative_sampler = negative_sampler
        self.batch_size = batch_size
        self.num_negatives = num_negatives
        self.drop_last = drop_last
        self.shuffle = shuffle

    def __len__(self) -> int:
        n = self.targets.numel()
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Batch]:
        n = self.targets.numel()
        order = torch.randperm(n, device=self.device) if self.shuffle else None
        for lo in range(0, n, self.batch_size):
            hi = min(lo + self.batch_size, n)
            if self.drop_last and hi - lo < self.batch_size:
                continue
            if order is not None:
                idx = order[lo:hi]
                t, c = self.targets[idx], self.contexts[idx]
            else:
                t, c = self.targets[lo:hi], self.contexts[lo:hi]
            neg = self.negative_sampler.sample(hi - lo, self.num_negatives)
            if neg.device != self.device:
                neg = neg.to(self.device)
            yield Batch(t, c, neg, hi - lo)


class EarlyStopping:
    """The stopping rule of the reference's ``EarlyStopping`` in 'min' mode
    (``src/training/callbacks.py:18-84``): a value counts as an improvement only when it beats
    the rule's OWN best by more than ``min_delta``, and only then does that best (and
    ``best_epoch``) move -- so a slow steady descent adds up and eventually registers, where a
    comparison against the running minimum would never see it."""

    def __init__(self, patience: int = 10, min_delta: float = 0.0001):
        self.patience, self.min_delta = patience, min_delta
        self.best_value = float("inf")
        self.counter = self.best_epoch = self.stopped_epoch = 0

    def __call__(self, value: float, epoch: int = 0) -> bool:
        if value < self.best_value - self.min_delta:
            self.best_value, self.best_epoch, self.counter = value, epoch, 0
            return False
        self.counter += 1
        if self.counter >= self.patience:
            self.stopped_epoch = epoch
            return True
        return False


class GraphSAGETrainer:
    """Same constructor and methods as the reference trainer; the step itself is
    :class:`TrainStep` (flat gradient buffer, clip, Adam, optional RCCL all-reduce)."""

    def __init__(self, model, features: torch.Tensor, edge_index: torch.Tensor,
                 positive_pairs: Pairs, negative_sampler, config: Dict[str, Any],
                 val_pairs: Optional[Pairs] = None, verbose: bool = False):
        self.device = next(model.parameters()).device
        assert self.device.type == "cuda", "Model must be on GPU for training"
        assert features.device == self.device, "Features must be on same device as model"
        assert edge_index.device == self.device, "Edge index must be on same device as model"
        self.model, self.features, self.edge_index, self.config = model, features, edge_index, config
        tc = config.get("training", {})
        self.learning_rate = tc.get("learning_rate", 0.001)
        self.weight_decay = tc.get("weight_decay", 0.0)
        self.batch_size = tc.get("batch_size", 512)
        self.gradient_clip = tc.get("gradient_clip", 1.0)
        self.num_negatives = config.get("negatives", {}).get("per_positive", 5)
        self.patience = tc.get("early_stopping_patience", 10)
        self.min_delta = tc.get("min_delta", 0.0001)
        self.checkpoint_every = tc.get("checkpoint_every", 10)
        self.checkpoint_dir = config.get("paths", {}).get("checkpoints")   # None: keep nothing
        val_fraction = config.get("validation", {}).get("val_fraction", 0.1)

        t, c = pairs_to_tensors(positive_pairs, self.device)
        if val_pairs is None and val_fraction > 0:
            (t, c), (vt, vc) = split_pairs_train_val(t, c, val_fraction, seed=tc.get("seed", 42))
            val = (vt, vc)
        else:
            val = pairs_to_tensors(val_pairs, self.device) if val_pairs is not None else None
        self.train_pairs, self.val_pairs = (t, c), val
        if hasattr(negative_sampler, "to"):
            negative_sampler = negative_sampler.to(self.device)
        self.batch_generator = BatchGenerator((t, c), negative_sampler, self.batch_size,
                                              self.num_negatives, self.device, drop_last=False)
        self.val_batch_generator = None
        if val is not None and val[0].numel() > 0:
            self.val_batch_generator = BatchGenerator(val, negative_sampler, self.batch_size * 2,
                                                      self.num_negatives, self.device)
        self.stepper = TrainStep(model, features, edge_index, self.learning_rate, self.weight_decay,
                                 self.gradient_clip,
                                 receptive_field=bool(tc.get("receptive_field", False)))
        # full batches replay ONE captured HIP graph when the step is launch-bound (small graphs:
        # the reference's default 500-node graph steps in 0.23 ms instead of 0.63 ms); "auto"
        # keeps the eager step where the per-layer mask tensors of the captured form would be
        # large or the step is long enough to hide its launches
        hg = tc.get("hip_graph", "auto")
        self._use_graph = (hg is True or (hg == "auto" and features.size(0) <= 200_000)) and \
            len(self.batch_generator) > 1 and \
            self.train_pairs[0].numel() >= self.batch_size and not self.stepper.receptive_field
        self._graph_ready = False
        # the reference builds its stopping rule once (trainer.py:166): patience counter and best
        # value carry over when train() is called again
        self.early_stopping = EarlyStopping(self.patience, self.min_delta)
        # the sampler's device draw stream hangs on the stepper's base seed (which rank 0 chose
        # for everybody), one independent stream per rank: each rank draws its own shard's
        # negatives, and no rank's stream depends on what its CPU generator was used for
        if hasattr(negative_sampler, "reseed") and getattr(negative_sampler, "_seed", None) is None:
            from .graphsage import mix_seed
            negative_sampler.reseed(mix_seed(self.stepper.base_seed, 0x5EED, self.stepper.rank))
        self.optimizer = self.stepper.optimizer
        self.loss_fn = self.stepper.loss_fn
        self.best_loss = float("inf")
        self.best_epoch = 0
        self.current_epoch = 0
        self.history: List[Dict[str, Any]] = []
        self.verbose = verbose

    def train_epoch(self) -> Dict[str, float]:
        """``trainer.py:199-255``: means over the epoch's batches of total/pos/neg loss."""
        sums = torch.zeros(8, device=self.device)
        nb = 0
        if self._use_graph and not self._graph_ready:
            self.stepper.capture_step(self.batch_size, self.num_negatives, preserve_state=True)
            self._graph_ready = True
        for batch in self.batch_generator:
            if self._use_graph and batch.size == self.batch_size:
                sums += self.stepper.step_captured(batch.targets, batch.contexts, batch.negatives)
            else:
                sums += self.stepper.step(batch.targets, batch.contexts, batch.negatives,
                                          global_batch=batch.size * self.stepper.world)
            nb += 1
        host = (sums / max(nb, 1)).tolist()       # the epoch's only device-to-host copy
        if host[7] != 0.0:
            raise IndexError("pair/negative ids outside the embedding table")
        return {"loss": host[0], "pos_loss": host[1], "neg_loss": host[2]}

    @torch.no_grad()
    def validate(self) -> Dict[str, float]:
        """``trainer.py:257-297``: one forward, loss over validation batches of 2x batch size."""
        if self.val_batch_generator is None:
            return {}
        self.model.eval()
        emb = self.model(self.features, self.edge_index)
        sums = torch.zeros(8, device=self.device)
        nb = 0
        for batch in self.val_batch_generator:
            _, stats = self.loss_fn.forward_stats(emb, batch.targets, batch.contexts, batch.negatives)
            sums += stats
            nb += 1
        host = (sums / max(nb, 1)).tolist()
        d = dict(zip(DETAIL_KEYS, host))
        return {"loss": d["total_loss"], "pos_prob": d["pos_prob"], "neg_prob": d["neg_prob"]}

    def train(self, num_epochs: int) -> float:
        """``trainer.py:299-364``: epochs with validation, best-model checkpoint, early stop."""
        for epoch in range(num_epochs):
            self.current_epoch = epoch
            t0 = time.time()
            train_metrics = self.train_epoch()
            val_metrics = self.validate()
            monitor = val_metrics.get("loss", train_metrics["loss"])
            self.history.append({"epoch": epoch, "train": train_metrics, "val": val_metrics,
                                 "seconds": time.time() - t0})
            if self.verbose:
                print(f"epoch {epoch}: train {train_metrics['loss']:.4f} val {monitor:.4f}")
            if monitor < self.best_loss:          # plain minimum, trainer.py:345-346
                self.best_loss = monitor
                if self.checkpoint_dir:
                    os.makedirs(self.checkpoint_dir, exist_ok=True)
                    self.save_checkpoint(os.path.join(self.checkpoint_dir, "model_best.pt"),
                                         value=monitor, train_metrics=train_metrics,
                                         val_metrics=val_metrics)
            if self.checkpoint_dir and self.checkpoint_every and (epoch + 1) % self.checkpoint_every == 0:
                self.save_checkpoint(os.path.join(self.checkpoint_dir, f"model_epoch{epoch + 1}.pt"),
                                     value=monitor)
            stop = self.early_stopping(monitor, epoch)          # trainer.py:349
            self.best_epoch = self.early_stopping.best_epoch    # as the reference logs it (:358)
            if stop:
                break
        return self.best_loss

    def get_embeddings(self) -> torch.Tensor:
        return self.stepper.get_embeddings()

    def save_checkpoint(self, path: str, **extra) -> None:
        """Same keys as ``trainer.py:385-391`` / ``callbacks.py:220-233``; the optimiser state
        is written per parameter tensor, as ``Adam(model.parameters())`` would hold it."""
        state = {"epoch": self.current_epoch, "model_state_dict": self.model.state_dict(),
                 "optimizer_state_dict": self.stepper.optimizer_state_dict(),
                 "best_loss": self.best_loss, "best_value": self.best_loss,
                 "best_epoch": self.best_epoch, "config": self.config,
                 # not in the reference: where the device-side random streams stand
                 "random_streams": {"stepper": self.stepper.state_dict(),
                                    "sampler": self.batch_generator.negative_sampler.state_dict()
                                    if hasattr(self.batch_generator.negative_sampler, "state_dict") else None}}
        state.update(extra)
        torch.save(state, path)

    def load_checkpoint(self, path: str) -> None:
        ckpt = torch.load(path, map_location=self.device, weights_only=True)
        self.model.load_state_dict(ckpt["model_state_dict"])
        self.stepper.load_optimizer_state_dict(ckpt["optimizer_state_dict"])
        self._graph_ready = False        # the optimiser state tensors may be new objects: recapture
        self.current_epoch = ckpt.get("epoch", 0)
        self.best_loss = ckpt.get("best_loss", float("inf"))
        streams = ckpt.get("random_streams")
        if streams:
            self.stepper.load_state_dict(streams["stepper"])
            sampler = self.batch_generator.negative_sampler
            if streams.get("sampler") is not None and hasattr(sampler, "load_state_dict"):
                sampler.load_state_dict(streams["sampler"])


class StreamingBatchGenerator:
    """Counterpart of ``StreamingBatchGenerator`` (``batch_generator.py:205-309``): batches of
    pairs streamed out of the walks instead of a materialised pair list -- every pair of every
    walk exactly once per epoch, walks visited in a fresh random order, pairs of a walk in the
    reference's order (position i, then window slot j), batches of exactly ``batch_size`` pairs
    except the last.  The reference runs a triple Python loop with list buffers; here the walks
    are the device tensor of :class:`RandomWalkGenerator`, a batch is pair numbers
    ``[b*B, (b+1)*B)`` of the permuted walks decoded by one kernel (``gsage_walk_pairs_gather``)
    and the negatives come from one more (``gsage_alias_sample``)."""

    def __init__(self, walks, context_window: int, negative_sampler, batch_size: int = 512,
                 num_negatives: int = 5, device=None, lengths: Optional[torch.Tensor] = None,
                 shuffle: bool = True):
        """``walks``: int32 ``[W, L]`` tensor padded with -1 plus ``lengths`` (what
        ``RandomWalkGenerator.generate_all_walks`` returns), or the reference's list of lists."""
        from .walks import CooccurrencePairSampler
        self.device = torch.device(device) if device is not None else torch.device("cuda")
        if not torch.is_tensor(walks):
            L = max((len(w) for w in walks), default=1)
            lengths = torch.tensor([len(w) for w in walks], dtype=torch.int32)
            walks = torch.tensor([list(w) + [-1] * (L - len(w)) for w in walks],
                                 dtype=torch.int32).reshape(len(lengths), L)
        elif lengths is None:
            lengths = (walks >= 0).sum(1).to(torch.int32)
        self.walks = walks.to(self.device, torch.int32).contiguous()
        self.lengths = lengths.to(self.device, torch.int32).contiguous()
        self.context_window = context_window
        self.negative_sampler = negative_sampler
        self.batch_size, self.num_negatives, self.shuffle = batch_size, num_negatives, shuffle
        self._pairs = CooccurrencePairSampler(context_window)
        self.walk_order: Optional[torch.Tensor] = None   # a fixed visiting order (tests)
        # batch_generator.py:243-251: walk_len * min(2*window, walk_len - 1), an upper estimate
        n = self.lengths.to(torch.int64)
        self.estimated_pairs = int((n * torch.clamp(n - 1, max=2 * context_window)).sum().item())

    def __iter__(self) -> Iterator[Batch]:
        W = self.lengths.numel()
        if W == 0:
            return
        walks, lengths = self.walks, self.lengths
        if self.walk_order is not None or self.shuffle:   # batch_generator.py:259-261
            perm = self.walk_order.to(self.device) if self.walk_order is not None else \
                torch.randperm(W, device=self.device)
            walks, lengths = walks[perm].contiguous(), lengths[perm].contiguous()
        offsets = self._pairs.index(walks, lengths)
        total = int(offsets[-1].item())
        for lo in range(0, total, self.batch_size):
            size = min(self.batch_size, total - lo)
            t, c = self._pairs._gather(walks, lengths, offsets, None, lo, size)
            neg = self.negative_sampler.sample(size, self.num_negatives)
            if neg.device != self.device:
                neg = neg.to(self.device)
            yield Batch(t, c, neg, size)


def train_graphsage(dataset, walks, config: Dict[str, Any], device=None):
    """Counterpart of ``train_graphsage`` (``trainer.py:408-486``), "the recommended entry point
    for training": pairs out of the walks, degree-biased negative sampler, model from the config,
    trainer, ``train(epochs)`` -> ``(trained_model, embeddings)``.

    ``dataset``: anything with ``get_data(device=)`` returning an object with ``x``,
    ``edge_index`` and ``num_nodes`` (the reference's ``GraphSAGEDataset``), or a plain
    ``(features, edge_index)`` pair.  ``walks``: the reference's list of lists, or the
    ``(walks, lengths)`` tensors of :class:`RandomWalkGenerator`; the pairs are extracted on the
    device in the reference's order either way."""
    from .graphsage import GraphSAGE
    from .sampler import DegreeBiasedNegativeSampler
    from .walks import CooccurrencePairSampler
    device = torch.device(device) if device is not None else torch.device("cuda")
    assert device.type == "cuda", "GPU required for training"
    if hasattr(dataset, "get_data"):
        data = dataset.get_data(device=device)
        features, edge_index, num_nodes = data.x, data.edge_index, int(data.num_nodes)
    else:
        features, edge_index = dataset
        features, edge_index = features.to(device), edge_index.to(device)
        num_nodes = int(features.size(0))
    window = config.get("walks", {}).get("context_window", 10)
    if isinstance(walks, (tuple, list)) and len(walks) == 2 and torch.is_tensor(walks[0]):
        wt, wl = walks[0].to(device, torch.int32), walks[1].to(device, torch.int32)
    else:
        L = max((len(w) for w in walks), default=1)
        wl = torch.tensor([len(w) for w in walks], dtype=torch.int32, device=device)
        wt = torch.tensor([list(w) + [-1] * (L - len(w)) for w in walks], dtype=torch.int32,
                          device=device).reshape(len(walks), L)
    pairs = CooccurrencePairSampler(window).extract_pairs_tensor(wt.contiguous(), wl.contiguous())
    neg_sampler = DegreeBiasedNegativeSampler(edge_index, num_nodes,
                                              exponent=config.get("negatives", {}).get("exponent", 0.75),
                                              device=device)
    mc = config.get("model", {})
    model = GraphSAGE(in_channels=config.get("features", {}).get("dimensions", 7),
                      hidden_channels=mc.get("hidden_dim", 64), out_channels=mc.get("output_dim", 64),
                      num_layers=mc.get("num_layers", 2), dropout=mc.get("dropout", 0.3),
                      normalize_output=mc.get("normalize_output", True)).to(device)
    trainer = GraphSAGETrainer(model, features, edge_index, pairs, neg_sampler, config)
    trainer.train(num_epochs=config.get("training", {}).get("epochs", 100))
    return model, trainer.get_embeddings()
