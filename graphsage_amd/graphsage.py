"""Drop-in ``GraphSAGE`` for the reference's ``src/model/graphsage.py``.

Same constructor, attributes, ``forward(x, edge_index)`` contract and ``state_dict``
layout as the reference class (``src/model/graphsage.py:68-157``), whose layers are
``torch_geometric.nn.SAGEConv`` (parameters ``convs.{i}.lin_l.weight [out,in]``,
``convs.{i}.lin_l.bias [out]``, ``convs.{i}.lin_r.weight [out,in]``).  Every layer runs as
CSR segment-mean (K2) + fused projection (K3) on the MI355X; ReLU/dropout of the hidden
layers (``:134-135``) and the final ``F.normalize`` (``:141-142``) are epilogues of K3.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .graph import GraphCSR, default_cache
from . import ops
from .ops import SAGELayerFunction, aggregate_mean


def _f32(x: torch.Tensor) -> torch.Tensor:
    return x.contiguous()


_VEC_WIDTHS = (4, 8, 16, 32, 64, 128, 256)


def vector_width(f: int) -> int:
    """Feature width the vectorised gather kernels take for an ``f``-column input: the next
    of 4,8,...,256 if that costs at most a third more columns, else ``f`` itself."""
    for w in _VEC_WIDTHS:
        if f <= w:
            return w if w * 3 <= f * 4 else f
    return f


_M64 = (1 << 64) - 1


def mix_seed(base: int, step: int, layer: int = 0) -> int:
    """64-bit seed for (base seed, step number, layer): splitmix64 of their combination, pure
    integer arithmetic on the host (no generator state involved)."""
    z = (int(base) + 0x9E3779B97F4A7C15 * (int(step) * 64 + int(layer) + 1)) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return (z ^ (z >> 31)) & _M64


class SAGEConv(nn.Module):
    """Mean-aggregating SAGEConv: ``lin_l(mean_{j->i} x_j) + lin_r(x_i)``.

    Stands in for ``torch_geometric.nn.SAGEConv(in, out, aggr='mean')`` with PyG's
    defaults (root_weight=True, bias=True, normalize=False, project=False) and PyG's
    parameter names; initialisation is ``nn.Linear``'s (Kaiming-uniform, a=sqrt(5)).
    """

    def __init__(self, in_channels: int, out_channels: int, aggr: str = "mean"):
        super().__init__()
        if aggr != "mean":
            raise ValueError(f"Unknown aggregator: {aggr!r} (the MI355X hot path implements 'mean')")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.aggr = aggr
        self.lin_l = nn.Linear(in_channels, out_channels, bias=True)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False)
        # opt-in: reuse mean_j x_j across calls while x (no grad) and the graph stay the same
        self.cache_aggregation = False

    def reset_parameters(self) -> None:
        self.lin_l.reset_parameters()
        self.lin_r.reset_parameters()

    def run(self, x: torch.Tensor, graph: GraphCSR, epilogue: int = _lib.EPI_NONE, p: float = 0.0,
            seed: int = 0, keep_mask: Optional[torch.Tensor] = None,
            force_generic: bool = False, input_layer: bool = True,
            agg: Optional[torch.Tensor] = None, stash: Optional[dict] = None,
            amax_in: Optional[torch.Tensor] = None,
            amax_out: Optional[torch.Tensor] = None, upstream: Optional[dict] = None) -> torch.Tensor:
        """``input_layer``: x is the model's input (constant node features), so its padded copy is
        worth caching; hidden activations are temporaries and are padded without caching.
        ``agg``: the layer's neighbour mean computed elsewhere (no gradient wanted for it);
        ``stash``: filled with this layer's operands for :func:`ops.aggregate_hidden1_from`;
        ``amax_in`` / ``amax_out``: the range of the activations handed from layer to layer
        (one int32 word each, float bits of max |x| / max |out|);
        ``upstream``: the stash of the layer in front, given by a caller that knows this layer to be
        the only consumer of that layer's output (the two backwards may then meet in one kernel)."""
        w_l, w_r = self.lin_l.weight, self.lin_r.weight
        x_key = x
        width = vector_width(x.size(1))
        if width != x.size(1) and not force_generic:
            # e.g. the 7 memory features -> 8 columns: 32-byte rows that the 16-byte-per-lane
            # gather (and its hub-row pass) can take; zero columns change no result
            pad = width - x.size(1)
            x = default_cache().padded(x, width, keep=input_layer)
            w_l, w_r = F.pad(w_l, (0, pad)), F.pad(w_r, (0, pad))
        if agg is None and self.cache_aggregation and input_layer and not x_key.requires_grad:
            agg = default_cache().aggregated(x_key, x, graph, aggregate_mean)
        return SAGELayerFunction.apply(x, w_l, self.lin_l.bias, w_r, graph, epilogue, p, seed,
                                       keep_mask, force_generic, agg, stash, amax_in, amax_out, upstream)

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
        graph = default_cache().get(edge_index, x.size(0))
        return self.run(x, graph)

    def extra_repr(self) -> str:
        return f"{self.in_channels}, {self.out_channels}, aggr={self.aggr}"


class GraphSAGE(nn.Module):
    """GraphSAGE encoder (reference ``GraphSAGE``, ``src/model/graphsage.py:33-157``)."""

    def __init__(self, in_channels: int = 7, hidden_channels: int = 64, out_channels: int = 64,
                 num_layers: int = 2, dropout: float = 0.3, aggregator: str = "mean",
                 normalize_output: bool = True, cache_input_aggregation: bool = False):
        """``cache_input_aggregation`` (not in the reference): keep the first layer's
        ``mean_j x_j`` between calls while the feature tensor and the graph are unchanged, as
        they are across training steps.  Same results; off by default because the reference
        recomputes it, and so does the benchmark."""
        super().__init__()
        self.in_channels = in_channels
        self.hidden_channels = hidden_channels
        self.out_channels = out_channels
        self.num_layers = num_layers
        self.dropout_rate = dropout
        self.normalize_output = normalize_output

        self.convs = nn.ModuleList()
        self.convs.append(SAGEConv(in_channels, hidden_channels, aggr=aggregator))
        for _ in range(num_layers - 2):
            self.convs.append(SAGEConv(hidden_channels, hidden_channels, aggr=aggregator))
        if num_layers > 1:
            self.convs.append(SAGEConv(hidden_channels, out_channels, aggr=aggregator))
        # kept for API parity (reference attribute `dropout`); the mask itself is drawn
        # inside the K3 epilogue, not by this module
        self.dropout = nn.Dropout(dropout)
        self._force_generic = False
        self.convs[0].cache_aggregation = bool(cache_input_aggregation)
        # (base seed, step number) set by an owner that counts steps itself (TrainStep): the
        # in-kernel dropout hash is then seeded from those two numbers and the layer index alone,
        # so replicas that were given the same base seed draw the same masks whatever else their
        # processes did with torch's generators.  None: one draw from the CPU generator per layer.
        self._dropout_stream: Optional[tuple] = None

    def set_dropout_stream(self, base_seed: Optional[int], step: int = 0) -> None:
        """Seed the dropout masks of the next training-mode forward from ``(base_seed, step,
        layer)`` instead of torch's CPU generator; ``None`` goes back to the generator."""
        self._dropout_stream = None if base_seed is None else (int(base_seed), int(step))

    def _dropout_seed(self, layer: int = 0) -> int:
        if self._dropout_stream is not None:
            base, step = self._dropout_stream
            return mix_seed(base, step, layer)
        # CPU generator: reproducible under torch.manual_seed, no device sync
        return int(torch.empty((), dtype=torch.int64).random_().item())

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor,
                keep_masks: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
        """x: fp32 [N, in_channels]; edge_index: int64 [2, E] -> fp32 [N, out_channels].

        ``keep_masks`` (optional, one uint8 [N, hidden] tensor per hidden layer) injects the
        dropout masks instead of drawing them in the kernel (parity tests).
        """
        graph = default_cache().get(edge_index, x.size(0))
        if self._blocked_inference_applies(x):
            return self._forward_blocked(x, graph)
        p = float(self.dropout_rate)
        training = self.training and p > 0.0
        # the input layer leaves its operands behind: the layer after it can recompute the rows it
        # aggregates from them instead of gathering the stored activation (ops.aggregate_hidden1)
        stash = {} if (len(self.convs) > 1 and not self._force_generic and ops.HIDDEN1_RECOMPUTE) else None
        if stash is not None and ops.hidden1_will_apply(self.in_channels, self.hidden_channels,
                                                        graph.num_nodes, training, graph.num_edges):
            stash["want_u"] = True       # the input layer's kernel writes the gather's rows itself

        def agg_for(i):
            if i == 1 and ops.hidden1_applies(stash, self.in_channels, graph.num_nodes, graph.num_edges):
                return ops.aggregate_hidden1_from(graph, stash)
            return None

        def upstream_of(i):      # layer 1 is the only consumer of the input layer's output
            return stash if i == 1 and stash is not None and stash.get("u") is not None else None

        # the range of the hidden activations travels with them (one word per layer, written on the
        # device): a 64 -> 64 layer that is told max |x| runs on the fp16 two-part kernel
        amax = self._range_words(x)

        def amax_of(i):          # (word the layer in front of layer i published, word layer i publishes)
            if amax is None:
                return None, None
            return (amax[i - 1:i] if i > 0 else None), (amax[i:i + 1] if i < len(self.convs) - 1 else None)

        for i, conv in enumerate(self.convs[:-1]):
            a_in, a_out = amax_of(i)
            if training:
                km = keep_masks[i] if keep_masks is not None else None
                seed = 0 if km is not None else self._dropout_seed(i)
                x = conv.run(x, graph, _lib.EPI_RELU_DROPOUT, p, seed, km, self._force_generic,
                             input_layer=i == 0, agg=agg_for(i), stash=stash if i == 0 else None,
                             amax_in=a_in, amax_out=a_out, upstream=upstream_of(i))
            else:
                x = conv.run(x, graph, _lib.EPI_RELU, 0.0, 0, None, self._force_generic,
                             input_layer=i == 0, agg=agg_for(i), stash=stash if i == 0 else None,
                             amax_in=a_in, amax_out=a_out)
        epi = _lib.EPI_L2NORM if self.normalize_output else _lib.EPI_NONE
        last = len(self.convs) - 1
        return self.convs[-1].run(x, graph, epi, 0.0, 0, None, self._force_generic,
                                  input_layer=last == 0, agg=agg_for(last), amax_in=amax_of(last)[0],
                                  upstream=upstream_of(last))

    def _range_words(self, x: torch.Tensor) -> Optional[torch.Tensor]:
        """One int32 word per hidden layer for max |activation| (float bits), when every layer can
        publish / use it: the 7 (<= 8) input features into 64-wide layers all the way, 16-byte aligned
        fp32 on the GPU (the kernels behind gsage_project_fwd_range_supported)."""
        if self._force_generic or len(self.convs) < 2 or not x.is_cuda or x.dtype != torch.float32:
            return None
        if vector_width(x.size(1)) != 8 or self.hidden_channels != 64:
            return None
        if self.convs[-1].out_channels != 64 or not ops.RANGE_WORDS:
            return None
        if not _lib.load().gsage_project_fwd_range_supported(64, 64, _lib.EPI_RELU):   # GSAGE_PROJ64=f32
            return None
        return torch.empty(len(self.convs) - 1, dtype=torch.int32, device=x.device)

    # ---- wide models without autograd: hidden activations kept column-blocked (ops.py)
    def _blocked_inference_applies(self, x: torch.Tensor) -> bool:
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            return False
        if self.training and float(self.dropout_rate) > 0.0:
            return False
        h = self.hidden_channels
        if self._force_generic or len(self.convs) < 2 or h not in (128, 256) or not x.is_cuda:
            return False
        if vector_width(x.size(1)) != 8 or x.dtype != torch.float32:
            return False
        if x.size(0) * h * 4 < ops.BLOCKED_MIN_BYTES:
            return False
        lib = _lib.load()
        last = self.convs[-1]
        epi = _lib.EPI_L2NORM if self.normalize_output else _lib.EPI_NONE
        ok = lib.gsage_project_fwd_blocked_supported(8, h, 0, 1, _lib.EPI_RELU)
        ok = ok and all(lib.gsage_project_fwd_blocked_supported(h, h, 1, 1, _lib.EPI_RELU)
                        for _ in self.convs[1:-1])
        return bool(ok and lib.gsage_project_fwd_blocked_supported(h, last.out_channels, 1, 0, epi))

    def _forward_blocked(self, x: torch.Tensor, graph: GraphCSR) -> torch.Tensor:
        """Same arithmetic as the plain path, layer by layer; only the memory layout of the
        hidden activations differs (F/64 tables of [N,64]), so every gather runs on a table a
        quarter of the size."""
        first = self.convs[0]
        x8 = default_cache().padded(_f32(x), 8, keep=True)
        pad = 8 - first.in_channels
        w_l, w_r = F.pad(first.lin_l.weight, (0, pad)), F.pad(first.lin_r.weight, (0, pad))
        agg1 = agg = ops.aggregate_mean(graph, x8)
        # every hidden layer leaves max |h| behind (one word, float bits): the bound the next
        # layer's fp16-split GEMM scales its operands by (the mean of rows of h cannot exceed it),
        # so that GEMM need not make two passes over its 20 GB of input to find it
        amax = torch.empty(len(self.convs) - 1, dtype=torch.int32, device=x.device)
        h, _ = ops.project_fwd_blocked(agg, x8, w_l, first.lin_l.bias, w_r, _lib.EPI_RELU, False, True,
                                       amax_out=amax[0:1])
        del agg
        # the first wide aggregation: recomputed from [agg1 | x8] (64 bytes per neighbour instead of
        # the activation's F/64 rows of 256 bytes)
        stash = dict(agg=agg1, x=x8, w_l=w_l, b_l=first.lin_l.bias, w_r=w_r, sign=None,
                     epilogue=_lib.EPI_RELU, p=0.0)
        recompute = ops.hidden1_applies(stash, first.in_channels, graph.num_nodes, graph.num_edges)

        def aggregate(i, h):
            if i == 0 and recompute:
                return ops.aggregate_hidden1_from(graph, stash, out_blocked=True)
            return ops.aggregate_mean_blocked(graph, h)

        for i, conv in enumerate(self.convs[1:-1]):
            agg = aggregate(i, h)
            h, _ = ops.project_fwd_blocked(agg, h, conv.lin_l.weight, conv.lin_l.bias,
                                           conv.lin_r.weight, _lib.EPI_RELU, True, True,
                                           amax_in=amax[i:i + 1], amax_out=amax[i + 1:i + 2])
            del agg
        last = self.convs[-1]
        epi = _lib.EPI_L2NORM if self.normalize_output else _lib.EPI_NONE
        agg = aggregate(len(self.convs) - 2, h)
        out, _ = ops.project_fwd_blocked(agg, h, last.lin_l.weight, last.lin_l.bias,
                                         last.lin_r.weight, epi, True, False, amax_in=amax[-1:])
        return out

    def get_embedding_dim(self) -> int:
        return self.out_channels

    def reset_parameters(self) -> None:
        for conv in self.convs:
            conv.reset_parameters()

    def count_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


def create_model(config: dict, device: Optional[torch.device] = None) -> GraphSAGE:
    """Counterpart of ``create_model`` (``src/model/graphsage.py:299-325``)."""
    model_cfg = config.get("model", {})
    feat_cfg = config.get("features", {})
    model = GraphSAGE(
        in_channels=feat_cfg.get("dimensions", 7),
        hidden_channels=model_cfg.get("hidden_dim", 64),
        out_channels=model_cfg.get("output_dim", 64),
        num_layers=model_cfg.get("num_layers", 2),
        dropout=model_cfg.get("dropout", 0.3),
        aggregator=model_cfg.get("aggregator", "mean"),
        normalize_output=model_cfg.get("normalize_output", True),
    )
    if device is not None:
        model = model.to(device)
    return model
