"""Minimal training-step harness: the loop body of the reference trainer
(``src/training/trainer.py:213-246``) around the MI355X hot path, plus the pair-sharded
data-parallel exchange the reference does not have.

Per step: zero_grad -> full-graph forward -> skip-gram loss on this rank's pair shard ->
backward -> [one all-reduce of the flat gradient buffer over RCCL] -> clip_grad_norm_ ->
Adam.  Graph, features and weights are replicated on every GPU; only pairs are sharded.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from .loss import SkipGramLoss, stats_to_details


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of ``total`` pairs for ``rank``; sizes differ by at most 1."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class FlatParams:
    """Re-homes a module's parameters and gradients into two flat fp32 buffers, so the
    gradient exchange is ONE collective and clip/Adam touch one tensor.  ``extra`` floats
    are appended to the gradient buffer for piggy-backed statistics."""

    def __init__(self, module: nn.Module, extra: int = 8):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("module has no trainable parameters")
        dev, dt = params[0].device, params[0].dtype
        n = sum(p.numel() for p in params)
        self.numel = n
        self.flat = torch.empty(n, dtype=dt, device=dev)
        self.buf = torch.zeros(n + extra, dtype=dt, device=dev)
        self.flat_grad = self.buf[:n]
        self.extra = self.buf[n:]
        off = 0
        for p in params:
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)
            p.grad = self.flat_grad[off:off + k].view_as(p)
            off += k
        self.params = params
        self.flat_param = nn.Parameter(self.flat, requires_grad=True)
        self.flat_param.grad = self.flat_grad

    def zero_grad(self) -> None:
        self.buf.zero_()


class TrainStep:
    def __init__(self, model: nn.Module, features: torch.Tensor, edge_index: torch.Tensor,
                 learning_rate: float = 1e-3, weight_decay: float = 0.0,
                 gradient_clip: float = 1.0, scale_factor: float = 5.0,
                 process_group: Optional[dist.ProcessGroup] = None,
                 loss_fn: Optional[nn.Module] = None, receptive_field: bool = False,
                 seed: Optional[int] = None, distributed: Optional[bool] = None):
        """``distributed=False``: ignore an initialised process group (a single-process stepper
        next to a sharded one, as the parity checks build).  ``seed``: base seed of the step's device-side random streams (the dropout masks; the
        trainer derives its negative sampler's from it too).  None draws one from torch's CPU
        generator (so ``torch.manual_seed`` reproduces a run).  With a process group the value of
        rank 0 is broadcast once, here: replicas then draw identical masks for step s whatever
        else each process does with its generators afterwards."""
        self.model = model
        self.features = features
        self.edge_index = edge_index
        self.gradient_clip = gradient_clip
        self.loss_fn = loss_fn if loss_fn is not None else SkipGramLoss(scale_factor)
        if loss_fn is None and features.is_cuda:
            # the embedding gradient of a step has one consumer (the model's last layer) and is
            # not kept: build it in a persistent table that is cleared row-wise, not zero-filled
            from .graphsage import GraphSAGE
            from .ops import GradTable
            self.loss_fn.grad_table = GradTable(keep_rows=isinstance(model, GraphSAGE))
        self.flat = FlatParams(model)
        # capturable: the step counter lives on the device, so the optimiser can sit inside a
        # captured HIP graph (capture_step below) as well as run eagerly
        on_gpu = self.flat.flat.is_cuda
        self.optimizer = torch.optim.Adam([self.flat.flat_param], lr=learning_rate,
                                          weight_decay=weight_decay, capturable=on_gpu,
                                          foreach=False if not on_gpu else None)
        self._graph = None
        self._graph_tail = None          # (data-parallel capture: clip + Adam behind the all-reduce)
        self._held: list = []            # what a captured graph's addresses point into
        self._captured_versions = None
        # opt-in (graphsage_amd/receptive.py): compute only the 2-hop closure of the batch
        self.receptive_field = bool(receptive_field)
        self.distributed = dist.is_available() and dist.is_initialized() and distributed is not False
        self.group = process_group
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        self.rank = dist.get_rank(process_group) if self.distributed else 0
        if seed is None:
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
        self.base_seed = self._agree_on_seed(int(seed))
        self.steps_done = 0              # index of the next step in the dropout stream

    def _agree_on_seed(self, seed: int) -> int:
        """Rank 0's base seed for everybody (one 8-byte broadcast at construction)."""
        if self.world <= 1:
            return seed
        on_host = dist.get_backend(self.group) == "gloo"
        dev = torch.device("cpu") if on_host or not self.features.is_cuda else self.features.device
        t = torch.tensor([seed & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=dev)
        dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                       group=self.group)
        return int(t.item())

    def _arm_dropout(self) -> None:
        """Masks of this step = f(base seed, step number, layer): no generator state involved."""
        arm = getattr(self.model, "set_dropout_stream", None)
        if arm is not None:
            arm(self.base_seed, self.steps_done)
        self.steps_done += 1

    def state_dict(self) -> Dict[str, int]:
        """What a resumed run needs to continue the same random streams."""
        return {"base_seed": self.base_seed, "steps_done": self.steps_done}

    def load_state_dict(self, sd: Dict[str, int]) -> None:
        self.base_seed, self.steps_done = int(sd["base_seed"]), int(sd["steps_done"])

    def step(self, targets: torch.Tensor, contexts: torch.Tensor, negatives: torch.Tensor,
             global_batch: Optional[int] = None) -> torch.Tensor:
        """One optimisation step on this rank's shard.  ``global_batch`` = pairs of the whole
        step over all ranks (defaults to shard size x world).  Returns the device stats[8]
        tensor of the global batch (after the all-reduce); no host sync."""
        if global_batch is None:
            global_batch = targets.numel() * self.world
        self.model.train()
        self._arm_dropout()
        self.flat.zero_grad()
        if self.receptive_field:
            from .graph import default_cache
            from .receptive import receptive_forward
            graph = default_cache().get(self.edge_index, self.features.size(0))
            emb, targets, contexts, negatives = receptive_forward(
                self.model, self.features, graph, targets, contexts, negatives)
        else:
            emb = self.model(self.features, self.edge_index)
        loss, stats = self.loss_fn.forward_stats(emb, targets, contexts, negatives,
                                                 global_batch=global_batch)
        loss.backward()
        self._release_grad_table()
        self.flat.extra.copy_(stats)
        if self.world > 1:
            self._all_reduce_flat()
        if self.gradient_clip > 0:
            torch.nn.utils.clip_grad_norm_([self.flat.flat_param], self.gradient_clip)
        self.optimizer.step()
        return self.flat.extra

    def _all_reduce_flat(self) -> None:
        """SUM of [gradients | stats] over the ranks: RCCL on the device buffer; a gloo group
        (CPU rehearsals, several ranks on one GPU) gets the 36 KB buffer staged through host
        memory, because gloo's own device path was seen to hang with ranks sharing a GPU."""
        buf = self.flat.buf
        if buf.is_cuda and dist.get_backend(self.group) == "gloo":
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            buf.copy_(host)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)

    # ------------------------------------------------------------- whole step as one HIP graph
    def capture_step(self, batch_size: int, num_negatives: int, warmup: int = 3,
                     preserve_state: bool = False) -> None:
        """Capture zero_grad -> forward -> loss -> backward -> clip -> Adam into ONE HIP graph
        for batches of exactly (batch_size, num_negatives) -- per rank, when a process group is
        in use: the step is then TWO graphs, [zero_grad .. backward] and [clip, Adam], with the one
        all-reduce of the flat buffer issued between their replays.  Worth it when the graph is
        small and the step is launch-bound (the reference's default B = 1024 on its 500-node
        graph: ~60 launches of a few microseconds each).  Dropout masks are then drawn by a
        graph-safe Philox generator of the stepper's own, seeded with the base seed every replica
        shares, and handed to the projection kernel as keep-masks (a seed passed by value would
        be frozen into the graph).  The warm-up and the capture run real steps on an all-zero
        batch; ``preserve_state`` puts parameters and Adam state back as they were afterwards
        (what a trainer wants when it captures in the middle of a run)."""
        dev = self.features.device
        saved = None
        if preserve_state:
            st = self.optimizer.state.get(self.flat.flat_param, {})
            saved = (self.flat.flat.detach().clone(),
                     {k: v.detach().clone() for k, v in st.items() if torch.is_tensor(v)})
        self._gt = torch.zeros(batch_size, dtype=torch.int64, device=dev)
        self._gc = torch.zeros(batch_size, dtype=torch.int64, device=dev)
        self._gn = torch.zeros(batch_size, num_negatives, dtype=torch.int64, device=dev)
        # masks: the same stream on every replica (same seed, same number of draws: warm-up,
        # capture and replays run in lockstep by construction), untouched by whatever else the
        # process draws from torch's default generators
        self._mask_gen = torch.Generator(device=dev)
        self._mask_gen.manual_seed(self.base_seed & 0x7FFFFFFFFFFFFFFF)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):          # builds the CSR cache, sets kernel attributes
                self._graph_front(batch_size)
                if self.world > 1:
                    self._all_reduce_flat()
                self._graph_back()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # the warm-up steps above moved the parameters: that is the caller's to know
        self._graph = torch.cuda.CUDAGraph()
        self._graph.register_generator_state(self._mask_gen)
        self._graph_tail = None
        if self.world == 1:
            with torch.cuda.graph(self._graph):
                self._graph_front(batch_size)
                self._graph_back()
        else:
            with torch.cuda.graph(self._graph):
                self._graph_front(batch_size)
            self._all_reduce_flat()          # (keeps the ranks' collectives paired up)
            self._graph_tail = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph_tail, pool=self._graph.pool()):
                self._graph_back()
        # The captured kernels hold raw device addresses of tensors that only the process-wide
        # GraphCache owns (the CSR built during warm-up, the 7 -> 8 padded features, a cached
        # input aggregate).  Keep them alive here for as long as the graph exists, whatever the
        # cache evicts later, and remember which versions of the inputs they were made from.
        from .graph import default_cache
        self._held = default_cache().held_for(self.features, self.edge_index)
        self._captured_versions = (self.features._version, self.edge_index._version)
        if saved is not None:
            with torch.no_grad():
                self.flat.flat.copy_(saved[0])
                st = self.optimizer.state.get(self.flat.flat_param, {})
                for k, v in st.items():
                    if torch.is_tensor(v):      # in place: the graph holds these addresses
                        v.copy_(saved[1][k]) if k in saved[1] else v.zero_()

    def _graph_front(self, batch_size: int) -> None:
        """zero_grad -> forward -> loss on this rank's (static) batch -> backward -> stats."""
        model = self.model
        model.train()
        self.flat.zero_grad()
        masks = None
        p = float(getattr(model, "dropout_rate", 0.0))
        if p > 0.0:
            n = self.features.size(0)
            masks = [(torch.rand(n, conv.out_channels, device=self.features.device,
                                 generator=self._mask_gen) >= p).to(torch.uint8)
                     for conv in model.convs[:-1]]
        emb = model(self.features, self.edge_index, keep_masks=masks)
        loss, stats = self.loss_fn.forward_stats(emb, self._gt, self._gc, self._gn,
                                                 global_batch=batch_size * self.world)
        loss.backward()
        self._release_grad_table()
        self.flat.extra.copy_(stats)

    def _graph_back(self) -> None:
        """clip -> Adam on the (all-reduced) flat gradient."""
        if self.gradient_clip > 0:
            torch.nn.utils.clip_grad_norm_([self.flat.flat_param], self.gradient_clip)
        self.optimizer.step()

    def _release_grad_table(self) -> None:
        rel = getattr(self.loss_fn, "release_grad_table", None)
        if rel is not None:
            rel()

    def step_captured(self, targets, contexts, negatives) -> torch.Tensor:
        """Replay the captured step on a new batch (copied into the graph's static inputs)."""
        if self._graph is None:
            raise RuntimeError("call capture_step() first")
        if (self.features._version, self.edge_index._version) != self._captured_versions:
            # an in-place change of the graph or the features: the captured kernels would go on
            # reading the CSR / padded copy made from the old contents
            self._graph, self._graph_tail, self._held = None, None, []
            raise RuntimeError("features or edge_index were modified in place after capture_step(); "
                               "capture again")
        self._gt.copy_(targets)
        self._gc.copy_(contexts)
        self._gn.copy_(negatives)
        self._graph.replay()
        if self._graph_tail is not None:
            self._all_reduce_flat()
            self._graph_tail.replay()
        return self.flat.extra

    # ----------------------------------------------- optimiser state in the reference's layout
    def optimizer_state_dict(self) -> dict:
        """Adam state as ``Adam(model.parameters())`` of the reference would hold it (one entry
        per parameter tensor), so ``optimizer_state_dict`` in checkpoints stays interchangeable
        (``trainer.py:385-391``, ``callbacks.py:220-233``)."""
        st = self.optimizer.state.get(self.flat.flat_param, {})
        group = {k: v for k, v in self.optimizer.param_groups[0].items() if k != "params"}
        state, off = {}, 0
        for i, p in enumerate(self.flat.params):
            k = p.numel()
            if st:
                state[i] = {"step": st["step"].detach().clone().cpu().to(torch.float32),
                            "exp_avg": st["exp_avg"][off:off + k].view_as(p).clone(),
                            "exp_avg_sq": st["exp_avg_sq"][off:off + k].view_as(p).clone()}
            off += k
        group["params"] = list(range(len(self.flat.params)))
        group["capturable"] = False      # a property of this harness, not of the checkpoint
        group["foreach"] = None
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd: dict) -> None:
        if not sd.get("state"):
            return
        fp = self.flat.flat_param
        exp_avg, exp_avg_sq = torch.zeros_like(fp.data), torch.zeros_like(fp.data)
        off, step = 0, None
        for i, p in enumerate(self.flat.params):
            k = p.numel()
            ent = sd["state"][i]
            exp_avg[off:off + k].copy_(ent["exp_avg"].reshape(-1))
            exp_avg_sq[off:off + k].copy_(ent["exp_avg_sq"].reshape(-1))
            step = ent["step"]
            off += k
        step_t = torch.as_tensor(float(step), dtype=torch.float32,
                                 device=fp.device if self.optimizer.defaults.get("capturable") else "cpu")
        self.optimizer.state[fp] = {"step": step_t, "exp_avg": exp_avg, "exp_avg_sq": exp_avg_sq}
        g = sd["param_groups"][0]
        for key in ("lr", "betas", "eps", "weight_decay"):
            if key in g:
                self.optimizer.param_groups[0][key] = g[key]

    def step_with_details(self, targets, contexts, negatives,
                          global_batch: Optional[int] = None) -> Dict[str, float]:
        return stats_to_details(self.step(targets, contexts, negatives, global_batch).clone())

    @torch.no_grad()
    def get_embeddings(self) -> torch.Tensor:
        """Counterpart of ``GraphSAGETrainer.get_embeddings`` (trainer.py:366-376)."""
        self.model.eval()
        return self.model(self.features, self.edge_index)
