"""Thin Python wrappers over the C ABI plus the two ``autograd.Function``s.

PyTorch owns every buffer (inputs, outputs, saved tensors, workspaces) and supplies
the stream; all arithmetic happens in libgsage.so.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import EPI_L2NORM, EPI_NONE
from .graph import GraphCSR


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32, got {t.dtype}")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (graphsage_amd has no CPU path)")
    return t.contiguous()


def _i64c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t.dtype != torch.int64:
        raise ValueError(f"{name} must be int64, got {t.dtype}")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (graphsage_amd has no CPU path)")
    return t.contiguous()


# ---------------------------------------------------------------- optional kernel timing
class KernelTimer:
    """Brackets selected C-ABI calls with HIP events on the stream they are launched on
    (torch's current stream), for bench.py's live per-kernel durations."""

    def __init__(self):
        self.records = {}
        self.enabled = True

    def bracket(self, name):
        start = torch.cuda.Event(enable_timing=True)
        end = torch.cuda.Event(enable_timing=True)
        self.records.setdefault(name, []).append((start, end))
        return start, end

    def summary(self):
        """name -> (launches, mean milliseconds); call after a device synchronize."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v))
                for k, v in self.records.items() if v}


_timer: Optional[KernelTimer] = None


class _Bracket:
    """``with _bracket(tag):`` -- HIP events around the launches inside, if a timer is set."""
    __slots__ = ("ev",)

    def __init__(self, tag):
        self.ev = _timer.bracket(tag) if _timer is not None and _timer.enabled else None

    def __enter__(self):
        if self.ev:
            self.ev[0].record()

    def __exit__(self, *exc):
        if self.ev:
            self.ev[1].record()
        return False


_bracket = _Bracket


def set_kernel_timer(timer: Optional[KernelTimer]) -> None:
    global _timer
    _timer = timer


# ----------------------------------------------------------------------------- raw ops
def segment_gather(rowptr, col, x, row_scale=None, edge_weight=None, add_to=None, out=None,
                   long_rows=None, n_long=None, src_nz_mask=None, live_mask=None) -> torch.Tensor:
    """out[i] = add_to[i] + row_scale[i] * sum_p edge_weight[col[p]] * x[col[p]] (K2/K7).

    ``src_nz_mask`` (int32 words, one bit per row of ``x``; a clear bit promises a zero row)
    lets the kernel skip the rows it need not fetch."""
    lib = _lib.load()
    x = _f32c(x, "x")
    F = x.size(1)
    N = rowptr.numel() - 1            # output rows; x may hold more rows than that (row subsets)
    if out is None:
        out = torch.empty(N, F, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        # calls that carry a zero-row mask move fewer bytes than the plain gather: timed apart
        tag = f"segment_gather_F{F}" + ("_skip" if src_nz_mask is not None or live_mask is not None
                                        else "")
        ev = _timer.bracket(tag) if _timer is not None and _timer.enabled else None
        if ev:
            ev[0].record()
        if live_mask is not None:
            # rows of x AND add_to outside the mask are zero by promise and may hold anything
            if live_mask.dtype != torch.int32 or live_mask.numel() * 32 < max(x.size(0), N):
                raise ValueError("live_mask must be int32 with one bit per row of x and add_to")
            if add_to is None:
                raise ValueError("live_mask needs add_to")
            rc = lib.gsage_segment_gather_live(rowptr.data_ptr(), col.data_ptr(),
                                               _lib.ptr(row_scale), _lib.ptr(edge_weight),
                                               x.data_ptr(), add_to.data_ptr(), N, F,
                                               out.data_ptr(), _lib.ptr(long_rows),
                                               _lib.ptr(n_long), live_mask.data_ptr(),
                                               _lib.stream_handle())
        elif src_nz_mask is not None:
            if src_nz_mask.dtype != torch.int32 or src_nz_mask.numel() * 32 < x.size(0):
                raise ValueError("src_nz_mask must be int32 with one bit per row of x")
            rc = lib.gsage_segment_gather_skip(rowptr.data_ptr(), col.data_ptr(),
                                               _lib.ptr(row_scale), _lib.ptr(edge_weight),
                                               x.data_ptr(), _lib.ptr(add_to), N, F,
                                               out.data_ptr(), _lib.ptr(long_rows),
                                               _lib.ptr(n_long), src_nz_mask.data_ptr(),
                                               _lib.stream_handle())
        else:
            rc = lib.gsage_segment_gather(rowptr.data_ptr(), col.data_ptr(), _lib.ptr(row_scale),
                                          _lib.ptr(edge_weight), x.data_ptr(), _lib.ptr(add_to),
                                          N, F, out.data_ptr(), _lib.ptr(long_rows),
                                          _lib.ptr(n_long), _lib.stream_handle())
        if ev:
            ev[1].record()
    _lib.check(rc, "gsage_segment_gather")
    return out


def aggregate_mean(graph: GraphCSR, x: torch.Tensor) -> torch.Tensor:
    """SAGEConv mean aggregation (forward, K2)."""
    return segment_gather(graph.rowptr, graph.col, x, row_scale=graph.inv_deg,
                          long_rows=graph.long_rows, n_long=graph.n_long[0:1])


# The layer behind the input layer aggregates rows that are functions of 14 numbers each: fetch
# those and redo the projection on the matrix cores instead of gathering the stored activation
# (gsage_segment_gather_hidden1).  GSAGE_HIDDEN1_RECOMPUTE=0 keeps the plain gather.
HIDDEN1_RECOMPUTE = os.environ.get("GSAGE_HIDDEN1_RECOMPUTE", "1") != "0"
# ... from the size on where it wins: below it the plain gather reads the stored activation out of
# the caches.  What decides is the bytes the plain gather would fetch, E x 4 Fo: measured break-even
# (tools/hidden_gather_probe.py, tools/hidden_gather_shapes_probe.py; pack kernels included) at
# 400 k nodes x 64 columns, 150 k x 128, 100 k x 256 on the benchmark's degree distribution (mean
# 20), and for 64 columns at ~1.5 M nodes with mean degree 5, ~130 k with mean degree 60 -- 1.5 to
# 2.05 GB of gathered rows every time, whatever the tail exponent (2.1 / 2.5 / 3.5, or degrees
# capped at 40).  Round 2 switched on the size of the activation (96 MiB), which is the same
# thing at mean degree 20 only.
HIDDEN1_MIN_BYTES = int(os.environ.get("GSAGE_HIDDEN1_MIN_BYTES", str(7 << 28)))     # 1.75 GiB


def hidden1_applies(stash: Optional[dict], in_channels: int, num_nodes: int,
                    num_edges: Optional[int] = None) -> bool:
    """Can the aggregation of the layer whose forward filled ``stash`` be recomputed?  Its input
    must be the 7 (or fewer) features padded to 8 columns and its epilogue one the kernel has."""
    if not HIDDEN1_RECOMPUTE or not stash or in_channels > 7 or num_nodes >= (1 << 26):
        return False
    x, w_l = stash["x"], stash["w_l"]
    if x.size(1) != 8 or w_l.size(1) != 8 or not x.is_cuda:
        return False
    training = stash["epilogue"] == _lib.EPI_RELU_DROPOUT
    edges = num_edges if num_edges is not None else 20 * num_nodes
    if edges * w_l.size(0) * 4 < HIDDEN1_MIN_BYTES:
        return False
    # the kernel addresses its row table with 32-bit byte offsets: 64-byte rows, or 128-byte rows
    # when the dropout masks of more than 64 columns ride behind them
    if num_nodes * (128 if training and w_l.size(0) > 64 else 64) > (1 << 32):
        return False
    if training and stash.get("u") is None and stash["sign"] is None and stash.get("out") is None:
        return False          # nothing to take the stored activation's zero pattern from
    return bool(_lib.load().gsage_segment_gather_hidden1_supported(w_l.size(0), int(stash["epilogue"])))


def hidden1_will_apply(in_channels: int, hidden: int, num_nodes: int, training: bool,
                       num_edges: Optional[int] = None) -> bool:
    """The size / shape part of :func:`hidden1_applies`, answerable before the input layer runs."""
    if not HIDDEN1_RECOMPUTE or in_channels > 7 or num_nodes >= (1 << 26) or hidden not in (64, 128, 256):
        return False
    edges = num_edges if num_edges is not None else 20 * num_nodes
    if edges * hidden * 4 < HIDDEN1_MIN_BYTES:
        return False
    return num_nodes * (128 if training and hidden > 64 else 64) <= (1 << 32)


HIDDEN1_STATS = {"used": 0}      # tests / diagnostics

# 64-wide models hand max |activation| from layer to layer, and a 64 -> 64 layer that is told the
# range of its input runs on the fp16 two-part kernel (proj64h_fwd_k) instead of the fp32 MFMA one.
# GSAGE_RANGE_WORDS=0 turns the hand-over off (every 64 -> 64 forward on the fp32 kernel).
RANGE_WORDS = os.environ.get("GSAGE_RANGE_WORDS", "1") != "0"


def aggregate_hidden1_from(graph: GraphCSR, stash: dict, out_blocked: bool = False) -> torch.Tensor:
    HIDDEN1_STATS["used"] += 1
    if stash.get("u") is not None:
        with torch.no_grad():
            return aggregate_hidden1_u(graph, stash["u"], stash["w_l"].detach().contiguous(),
                                       stash["b_l"].detach(), stash["w_r"].detach().contiguous(),
                                       stash["epilogue"], p=stash["p"], out_blocked=out_blocked)
    with torch.no_grad():
        return aggregate_hidden1(graph, stash["agg"], stash["x"], stash["w_l"].detach().contiguous(),
                                 stash["b_l"].detach(), stash["w_r"].detach().contiguous(),
                                 stash["epilogue"], p=stash["p"], relu_sign=stash["sign"],
                                 # training: the stored activation too, which the kernel reads for
                                 # the zero pattern when the forward kernel that ran left no sign
                                 # bits (unaligned operands or mask: the flag word behind the bits)
                                 h1=stash.get("out"), out_blocked=out_blocked)


def aggregate_hidden1(graph: GraphCSR, agg1, x8, w_l, b_l, w_r, epilogue, p: float = 0.0,
                      h1: Optional[torch.Tensor] = None, relu_sign: Optional[torch.Tensor] = None,
                      out_blocked: bool = False) -> torch.Tensor:
    """Mean aggregation of the first hidden layer recomputed from the input layer's operands
    (gsage_segment_gather_hidden1): the same [N, Fo] (or blocked) tensor ``aggregate_mean`` gives
    on the stored activation, up to fp32 rounding, for a quarter of the line traffic."""
    lib = _lib.load()
    N, Fo = x8.size(0), w_l.size(0)
    dev = x8.device
    for t, name in ((agg1, "agg1"), (x8, "x8"), (w_l, "w_l"), (w_r, "w_r")):
        if t.dtype != torch.float32 or t.size(-1) != 8 or not t.is_contiguous():
            raise ValueError(f"{name} must be contiguous float32 with 8 columns (7 features padded)")
    with torch.cuda.device(dev):
        out = torch.empty((Fo // 64, N, 64) if out_blocked else (N, Fo), dtype=torch.float32, device=dev)
        ws_bytes = lib.gsage_segment_gather_hidden1_workspace_bytes(N, graph.num_edges, Fo, int(epilogue))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        ev = _timer.bracket(f"segment_gather_hidden1_F{Fo}") if _timer is not None and _timer.enabled else None
        if ev:
            ev[0].record()
        rc = lib.gsage_segment_gather_hidden1(graph.hidden1_groups().data_ptr(), graph.num_edges,
                                              graph.col.data_ptr(), graph.inv_deg.data_ptr(), agg1.data_ptr(), x8.data_ptr(),
                                              w_l.data_ptr(), b_l.data_ptr(), w_r.data_ptr(), N, Fo,
                                              int(epilogue), float(p), _lib.ptr(h1), _lib.ptr(relu_sign),
                                              out.data_ptr(), int(out_blocked), ws.data_ptr(), ws_bytes,
                                              _lib.stream_handle())
        if ev:
            ev[1].record()
    _lib.check(rc, "gsage_segment_gather_hidden1")
    return out


def aggregate_hidden1_u(graph: GraphCSR, u_rows, w_l, b_l, w_r, epilogue, p: float = 0.0,
                        out_blocked: bool = False) -> torch.Tensor:
    """:func:`aggregate_hidden1` on a row table the input layer's kernel already wrote
    (gsage_segment_gather_hidden1_u): no pack pass."""
    lib = _lib.load()
    N, Fo = u_rows.size(0) - 1, w_l.size(0)        # (the table's last row holds max |u|)
    dev = u_rows.device
    with torch.cuda.device(dev):
        out = torch.empty((Fo // 64, N, 64) if out_blocked else (N, Fo), dtype=torch.float32, device=dev)
        ws_bytes = lib.gsage_segment_gather_hidden1_u_workspace_bytes(graph.num_edges, Fo)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _bracket(f"segment_gather_hidden1_F{Fo}"):
            rc = lib.gsage_segment_gather_hidden1_u(graph.hidden1_groups().data_ptr(), graph.num_edges,
                                                    graph.col.data_ptr(), graph.inv_deg.data_ptr(),
                                                    u_rows.data_ptr(), w_l.data_ptr(), b_l.data_ptr(),
                                                    w_r.data_ptr(), N, Fo, int(epilogue), float(p),
                                                    out.data_ptr(), int(out_blocked), ws.data_ptr(),
                                                    ws_bytes, _lib.stream_handle())
    _lib.check(rc, "gsage_segment_gather_hidden1_u")
    return out


# ------------------------------------------------ column-blocked activations (wide inference)
# A blocked [N, F] activation is F/64 separate [N, 64] tables: tensor shape [F/64, N, 64].  The
# 256-wide gather over one 10 GB table runs 15 % below the same bytes gathered from 2.56 GB
# tables, so in no-grad forwards of the 128/256-wide models the hidden layers hand their output
# on in this form (GraphSAGE.forward decides; GSAGE_BLOCKED_MIN_BYTES sets the table size from
# which it does, default 4 GiB; the layouts agree to fp32 rounding: the 64-wide gather adds the
# neighbours of a row in another order than the 256-wide one).
BLOCKED_MIN_BYTES = int(os.environ.get("GSAGE_BLOCKED_MIN_BYTES", str(4 << 30)))


def aggregate_mean_blocked(graph: GraphCSR, xb: torch.Tensor) -> torch.Tensor:
    """Mean aggregation of a blocked activation [F/64, N, 64] -> blocked [F/64, N, 64]."""
    global _timer
    nb = xb.size(0)
    out = torch.empty_like(xb)
    timer, ev = _timer, None
    if timer is not None and timer.enabled:
        ev = timer.bracket(f"segment_gather_F64x{nb}_blocked")
        ev[0].record()
    _timer = None                      # one bracket around the F/64 launches, not one each
    try:
        for b in range(nb):
            segment_gather(graph.rowptr, graph.col, xb[b], row_scale=graph.inv_deg, out=out[b],
                           long_rows=graph.long_rows, n_long=graph.n_long[0:1])
    finally:
        _timer = timer
    if ev:
        ev[1].record()
    return out


def project_fwd_blocked(agg, x, w_l, b_l, w_r, epilogue, in_blocked: bool, out_blocked: bool,
                        amax_in: Optional[torch.Tensor] = None,
                        amax_out: Optional[torch.Tensor] = None):
    """K3 forward with column-blocked operands (gsage_project_fwd_range) -> (out, rnorm).
    ``amax_in`` / ``amax_out``: one-word int32 tensors carrying max |activation| as float bits
    from a layer to the next (the fp16-split GEMM scales its operands by it)."""
    lib = _lib.load()
    for t in (amax_in, amax_out):
        if t is not None and (t.dtype != torch.int32 or t.numel() != 1 or not t.is_cuda):
            raise ValueError("amax_in / amax_out must be one-element int32 tensors on the GPU")
    Fo, Fi = w_l.shape
    N = x.size(1) if in_blocked else x.size(0)
    dev = x.device
    with torch.cuda.device(dev):
        out = torch.empty((Fo // 64, N, 64) if out_blocked else (N, Fo), dtype=torch.float32, device=dev)
        rnorm = torch.empty(N, dtype=torch.float32, device=dev) if epilogue == EPI_L2NORM else None
        wpack = torch.empty(lib.gsage_project_workspace_bytes(Fi, Fo), dtype=torch.uint8, device=dev)
        ev = _timer.bracket(f"project_fwd_{Fi}x{Fo}") if _timer is not None and _timer.enabled else None
        if ev:
            ev[0].record()
        rc = lib.gsage_project_fwd_range(agg.data_ptr(), x.data_ptr(), w_l.data_ptr(),
                                         b_l.data_ptr(), w_r.data_ptr(), N, Fi, Fo, int(epilogue),
                                         out.data_ptr(), _lib.ptr(rnorm), wpack.data_ptr(),
                                         int(in_blocked), int(out_blocked), _lib.ptr(amax_in),
                                         _lib.ptr(amax_out), _lib.stream_handle())
        if ev:
            ev[1].record()
    _lib.check(rc, "gsage_project_fwd_range")
    return out, rnorm


def aggregate_mean_bwd(graph: GraphCSR, grad_agg: torch.Tensor, prescaled: bool = False,
                       add_to: Optional[torch.Tensor] = None,
                       nz_mask: Optional[torch.Tensor] = None,
                       live_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """grad_x[j] = add_to[j] + sum_{i in out(j)} grad_agg[i] * inv_deg[i] (K7).
    ``nz_mask``: the zero-row bits of ``grad_agg`` from :func:`project_bwd`, if any;
    ``live_mask``: the rows of grad_agg AND add_to that were written at all
    (:func:`project_bwd_rows`)."""
    out = add_to if add_to is not None else None
    return segment_gather(graph.rowptr_t, graph.col_t, grad_agg,
                          edge_weight=None if prescaled else graph.inv_deg, add_to=add_to,
                          out=out, long_rows=graph.long_rows_t, n_long=graph.n_long[1:2],
                          src_nz_mask=nz_mask, live_mask=live_mask)


def project_fwd(agg, x, w_l, b_l, w_r, epilogue=EPI_NONE, keep_mask=None, seed=0, p=0.0,
                force_generic=False, want_relu_sign: bool = False, alloc_relu_sign: bool = True,
                u_out: Optional[dict] = None, amax_in: Optional[torch.Tensor] = None,
                amax_out: Optional[torch.Tensor] = None):
    """K3 forward -> (out, rnorm).  With ``want_relu_sign`` a third value is returned: the buffer
    of ``out > 0`` bits the first-layer kernel leaves for :func:`project_bwd` (None for shapes
    and epilogues that have no use for it, or when ``alloc_relu_sign`` is off).  ``u_out`` (a
    dict): asks the first-layer kernel to leave the rows the next layer's recompute gather fetches
    (gsage_project_fwd_u); ``u_out["u"]`` is the table if the kernel that ran wrote it, else None.
    ``amax_in`` / ``amax_out``: one-word int32 tensors carrying max |activation| (float bits) from
    layer to layer; a 64 -> 64 layer given ``amax_in`` runs on the fp16 two-part kernel."""
    lib = _lib.load()
    N, Fi = x.shape
    Fo = w_l.shape[0]
    dev = x.device
    with torch.cuda.device(dev):
        out = torch.empty(N, Fo, dtype=torch.float32, device=dev)
        rnorm = torch.empty(N, dtype=torch.float32, device=dev) if epilogue == EPI_L2NORM else None
        wpack = torch.empty(lib.gsage_project_workspace_bytes(Fi, Fo), dtype=torch.uint8, device=dev)
        if keep_mask is not None:
            if keep_mask.dtype != torch.uint8 or tuple(keep_mask.shape) != (N, Fo):
                raise ValueError("keep_mask must be uint8 [N, Fo]")
            keep_mask = keep_mask.contiguous()
        sign = None
        if want_relu_sign and alloc_relu_sign and Fi <= 8 and Fo % 64 == 0 and epilogue in (_lib.EPI_RELU,
                                                                       _lib.EPI_RELU_DROPOUT):
            sign = torch.empty(lib.gsage_relu_sign_bytes(N, Fo) // 8, dtype=torch.int64, device=dev)
        for t in (amax_in, amax_out):
            if t is not None and (t.dtype != torch.int32 or t.numel() != 1 or not t.is_cuda):
                raise ValueError("amax_in / amax_out must be one-element int32 tensors on the GPU")
        u_rows, wrote = None, None
        if u_out is not None and Fi == 8 and Fo in (64, 128, 256) and not force_generic and \
                epilogue in (_lib.EPI_RELU, _lib.EPI_RELU_DROPOUT):
            import ctypes
            u_rows = torch.empty(lib.gsage_hidden1_u_rows(N), lib.gsage_hidden1_u_row_bytes(Fo, int(epilogue)) // 4,
                                 dtype=torch.float32, device=dev)
            wrote = ctypes.c_int(0)
        with _bracket(f"project_fwd_{Fi}x{Fo}"):
            if u_rows is not None or amax_in is not None or amax_out is not None:
                import ctypes
                if wrote is None:
                    wrote = ctypes.c_int(0)

                def run(a_out):
                    return lib.gsage_project_fwd_u(
                        agg.data_ptr(), x.data_ptr(), w_l.data_ptr(), b_l.data_ptr(), w_r.data_ptr(), N, Fi, Fo,
                        int(epilogue), _lib.ptr(keep_mask), int(seed) & 0xFFFFFFFFFFFFFFFF, float(p),
                        out.data_ptr(), _lib.ptr(rnorm), wpack.data_ptr(), int(force_generic), _lib.ptr(sign),
                        _lib.ptr(u_rows), ctypes.byref(wrote), _lib.ptr(amax_in), _lib.ptr(a_out),
                        _lib.stream_handle())
                rc = run(amax_out)
                if rc == -2 and amax_out is not None:     # GSAGE_ERR_UNSUPPORTED
                    # the kernel this call lands on (odd addresses, a shape without a publishing
                    # kernel) does not leave max |out|: take it from the result instead
                    rc = run(None)
                    if rc == 0:
                        amax_out.copy_((out.abs().max() if N else out.new_zeros(())).view(torch.int32))
            else:
                rc = lib.gsage_project_fwd_ex(agg.data_ptr(), x.data_ptr(), w_l.data_ptr(), b_l.data_ptr(),
                                              w_r.data_ptr(), N, Fi, Fo, int(epilogue), _lib.ptr(keep_mask),
                                              int(seed) & 0xFFFFFFFFFFFFFFFF, float(p), out.data_ptr(),
                                              _lib.ptr(rnorm), wpack.data_ptr(), int(force_generic),
                                              _lib.ptr(sign), _lib.stream_handle())
        if u_out is not None:
            u_out["u"] = u_rows if (u_rows is not None and wrote is not None and wrote.value == 1) else None
    _lib.check(rc, "gsage_project_fwd")
    if want_relu_sign:
        return out, rnorm, sign
    return out, rnorm


# The gradient that reaches the last layer is non-zero only on the rows the pair batch touched
# (about half of a 10 M-node graph at B = 2^20, K = 5).  The fused 64 -> 64 backward notes the zero
# rows in a bit mask and the transposed gather skips them.  GSAGE_SKIP_ZERO_ROWS=0 turns it off.
SKIP_ZERO_ROWS = os.environ.get("GSAGE_SKIP_ZERO_ROWS", "1") != "0"


def project_bwd_rows(grad_out, out, rnorm, agg, x, w_l, w_r, agg_scale, epilogue, p,
                     live: LiveRows, gmax: Optional[torch.Tensor] = None):
    """Backward of K3 over the rows in ``live`` only (gsage_project_bwd_rows).  The rows of
    grad_agg / grad_x outside the list are NOT written: hand them on with ``live.mask``.
    ``gmax`` (one int32 word on the device): receives max |grad_agg|, |grad_x| as float bits
    (gsage_project_bwd_rows_range), which :func:`hidden1_wgrad` scales its operands by."""
    lib = _lib.load()
    N, Fi = x.shape
    Fo = w_l.shape[0]
    dev = x.device
    with torch.cuda.device(dev):
        f32 = dict(dtype=torch.float32, device=dev)
        grad_agg, grad_x = torch.empty(N, Fi, **f32), torch.empty(N, Fi, **f32)
        gwl, gwr, gb = torch.empty(Fo, Fi, **f32), torch.empty(Fo, Fi, **f32), torch.empty(Fo, **f32)
        ws_bytes = lib.gsage_project_bwd_workspace_bytes(N, Fi, Fo)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _bracket(f"project_bwd_rows_{Fi}x{Fo}"):
            args = (grad_out.data_ptr(), _lib.ptr(out), _lib.ptr(rnorm),
                    agg.data_ptr(), x.data_ptr(), w_l.data_ptr(), w_r.data_ptr(),
                    _lib.ptr(agg_scale), N, Fi, Fo, int(epilogue), float(p),
                    grad_agg.data_ptr(), grad_x.data_ptr(), gwl.data_ptr(),
                    gb.data_ptr(), gwr.data_ptr(), ws.data_ptr(), ws_bytes,
                    live.rows.data_ptr(), live.n_rows.data_ptr(), live.cap)
            if gmax is not None:
                rc = lib.gsage_project_bwd_rows_range(*args, gmax.data_ptr(), _lib.stream_handle())
            else:
                rc = lib.gsage_project_bwd_rows(*args, _lib.stream_handle())
    _lib.check(rc, "gsage_project_bwd_rows")
    return grad_agg, grad_x, gwl, gb, gwr


# The input layer wants only its weight gradients, and the layer behind it knows everything they
# are made of: with the 64-byte row table of the forward's recompute gather at hand, the second
# layer's backward hands (grad_agg, grad_x, the row list) to gsage_hidden1_wgrad instead of
# scattering grad_agg to the neighbours (K7) and pushing an [N, 64] gradient through the first layer.
# GSAGE_HIDDEN1_WGRAD=0 keeps the transposed gather + dW kernel.
HIDDEN1_WGRAD = os.environ.get("GSAGE_HIDDEN1_WGRAD", "1") != "0"
WGRAD_STATS = {"used": 0}      # tests / diagnostics


def hidden1_wgrad(graph: GraphCSR, u_rows, grad_agg, grad_x, gmax, live: LiveRows, epilogue, p: float):
    """gsage_hidden1_wgrad -> (grad_w_l [64, 8], grad_b [64], grad_w_r [64, 8]) of the input layer."""
    lib = _lib.load()
    N, Fo = grad_agg.shape
    dev = grad_agg.device
    with torch.cuda.device(dev):
        f32 = dict(dtype=torch.float32, device=dev)
        gwl, gb, gwr = torch.empty(Fo, 8, **f32), torch.empty(Fo, **f32), torch.empty(Fo, 8, **f32)
        ws_bytes = lib.gsage_hidden1_wgrad_workspace_bytes(live.cap, graph.num_edges)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _bracket(f"hidden1_wgrad_F{Fo}"):
            rc = lib.gsage_hidden1_wgrad(graph.rowptr.data_ptr(), graph.col.data_ptr(), N, graph.num_edges,
                                         u_rows.data_ptr(), grad_agg.data_ptr(), grad_x.data_ptr(),
                                         gmax.data_ptr(), live.rows.data_ptr(), live.n_rows.data_ptr(),
                                         live.cap, Fo, int(epilogue), float(p), gwl.data_ptr(),
                                         gb.data_ptr(), gwr.data_ptr(), ws.data_ptr(), ws_bytes,
                                         _lib.stream_handle())
    _lib.check(rc, "gsage_hidden1_wgrad")
    return gwl, gb, gwr


def project_bwd(grad_out, out, rnorm, agg, x, w_l, w_r, agg_scale, epilogue, p,
                need_input_grad: bool, force_generic=False, want_nz_mask: bool = False,
                relu_sign: Optional[torch.Tensor] = None):
    """Backward of K3.  With ``want_nz_mask`` a sixth value is returned: the int32 bit mask of the
    possibly non-zero rows of grad_agg / grad_x (None when the shape has no kernel that finds it)."""
    lib = _lib.load()
    N, Fi = x.shape
    Fo = w_l.shape[0]
    dev = x.device
    with torch.cuda.device(dev):
        f32 = dict(dtype=torch.float32, device=dev)
        dpre = torch.empty(N, Fo, **f32) if epilogue != EPI_NONE else None
        grad_agg = torch.empty(N, Fi, **f32) if need_input_grad else None
        grad_x = torch.empty(N, Fi, **f32) if need_input_grad else None
        gwl = torch.empty(Fo, Fi, **f32)
        gwr = torch.empty(Fo, Fi, **f32)
        gb = torch.empty(Fo, **f32)
        ws_bytes = lib.gsage_project_bwd_workspace_bytes(N, Fi, Fo)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        nz = None
        if want_nz_mask and need_input_grad and Fi == 64 and Fo == 64:
            nz = torch.empty((N + 31) // 32, dtype=torch.int32, device=dev)
        if relu_sign is not None and relu_sign.numel() * 8 != lib.gsage_relu_sign_bytes(N, Fo):
            raise ValueError("relu_sign does not belong to a forward of this shape")
        with _bracket(f"project_bwd_{Fi}x{Fo}" + ("" if need_input_grad else "_weights_only")):
            rc = lib.gsage_project_bwd_ex(grad_out.data_ptr(), _lib.ptr(out), _lib.ptr(rnorm),
                                          agg.data_ptr(), x.data_ptr(), w_l.data_ptr(), w_r.data_ptr(),
                                          _lib.ptr(agg_scale), N, Fi, Fo, int(epilogue), float(p),
                                          _lib.ptr(dpre), _lib.ptr(grad_agg), _lib.ptr(grad_x),
                                          gwl.data_ptr(), gb.data_ptr(), gwr.data_ptr(), ws.data_ptr(),
                                          ws_bytes, int(force_generic), _lib.ptr(nz),
                                          _lib.ptr(relu_sign), _lib.stream_handle())
    _lib.check(rc, "gsage_project_bwd")
    if want_nz_mask:
        return grad_agg, grad_x, gwl, gb, gwr, nz
    return grad_agg, grad_x, gwl, gb, gwr


ROWWISE_KEEPS_ROWS = os.environ.get("GSAGE_ROWWISE_KEEPS_ROWS", "1") != "0"


class LiveRows:
    """The rows of an [N, D] gradient that may be non-zero: bit mask + ascending list + count
    (all on the device), and the version counter of the tensor they describe."""
    __slots__ = ("mask", "rows", "n_rows", "cap", "version", "num_rows", "word_prefix", "consumed",
                 "rowwise", "outside_undefined")

    def __init__(self, mask, rows, n_rows, cap, version, num_rows, word_prefix=None):
        self.mask, self.rows, self.n_rows, self.cap = mask, rows, n_rows, cap
        self.version, self.num_rows = version, num_rows
        self.word_prefix = word_prefix     # place in `rows` of each mask word's first set bit
        self.consumed = False              # set by a consumer that read the listed rows only
        self.rowwise = False               # the listed rows were written in full (row-wise backward)
        self.outside_undefined = False     # rows outside the list hold a former step's values


# The gradient of the loss is non-zero only on the rows the pair batch names.  The loss backward
# says which (LiveRows, attached to the gradient tensor it returns); the 64 -> 64 layer behind it
# then walks those rows only and the transposed gather masks the rest.  GSAGE_LIVE_ROWS=0 turns
# the hand-over off (the dense kernels run, with the zero-row ballot of round 1).
LIVE_ROWS = os.environ.get("GSAGE_LIVE_ROWS", "1") != "0"
LIVE_STATS = {"made": 0, "used": 0}      # tests / diagnostics


def touched_rows(targets, contexts, negatives, N: int) -> LiveRows:
    """gsage_touched_rows: which rows of an N-row table the batch names."""
    lib = _lib.load()
    B = targets.numel()
    K = negatives.shape[1] if negatives.dim() == 2 else 0
    dev = targets.device
    cap = max(1, min(N, (2 + K) * B))
    with torch.cuda.device(dev):
        mask = torch.empty((N + 31) // 32, dtype=torch.int32, device=dev)
        wpre = torch.empty((N + 31) // 32, dtype=torch.int32, device=dev)
        rows = torch.empty(cap, dtype=torch.int32, device=dev)
        n_rows = torch.empty(1, dtype=torch.int32, device=dev)
        ws_bytes = lib.gsage_touched_rows_workspace_bytes(N)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _bracket("touched_rows"):
            rc = lib.gsage_touched_rows_ex(targets.data_ptr(), contexts.data_ptr(),
                                           negatives.data_ptr() if K > 0 else None, B, K, N,
                                           mask.data_ptr(), rows.data_ptr(), cap, n_rows.data_ptr(),
                                           wpre.data_ptr(), ws.data_ptr(), ws_bytes,
                                           _lib.stream_handle())
    _lib.check(rc, "gsage_touched_rows")
    return LiveRows(mask, rows, n_rows, cap, 0, N, wpre)


class GradTable:
    """A persistent [N, D] buffer for d(loss)/d(emb) that is ALL ZERO between steps: the loss
    backward adds into the rows its batch names, the consumer reads it, ``release`` clears
    exactly those rows again -- instead of zero-filling N x D floats every step (2.56 GB at 10 M
    x 64).  Only for an owner that knows the gradient has one consumer and is not kept
    (TrainStep): a leaf's ``.grad`` would alias the buffer."""

    def __init__(self, keep_rows: bool = False):
        """``keep_rows``: the owner vouches that the gradient's consumer is a layer of this
        package (which reads through the row list, or repairs the rest itself: SAGELayerFunction),
        so a row-wise loss backward may leave its rows in the table instead of clearing them."""
        self.buf: Optional[torch.Tensor] = None
        self.live: Optional[LiveRows] = None
        self.dirty = False
        self.keep_rows = bool(keep_rows) and ROWWISE_KEEPS_ROWS
        # rows outside the current list may hold a former step's values (the row-wise loss
        # backward overwrites its listed rows and a consumer that reads listed rows only needs
        # nothing cleared): a user that needs zeros elsewhere gets a full zero-fill first
        self.stale = False

    def take(self, like: torch.Tensor, overwrites_rows: bool = False) -> torch.Tensor:
        """``overwrites_rows``: the caller writes every listed row in full and its consumer will
        read listed rows only (gsage_skipgram_bwd_rows), so nothing has to be zero."""
        if self.buf is None or self.buf.shape != like.shape or self.buf.device != like.device:
            self.buf = torch.zeros_like(like)
            self.stale = False
        elif (self.dirty or self.stale) and not overwrites_rows:
            self.buf.zero_()             # the last user never released it / left rows behind
            self.stale = False
        self.dirty, self.live = True, None
        return self.buf

    def release(self) -> None:
        if self.buf is None or not self.dirty:
            return
        if self.live is not None and self.live.consumed and self.live.rowwise and self.keep_rows:
            # the listed rows were written in full and read through the list: leave them, the next
            # row-wise step overwrites its own rows and never looks at these
            self.stale = True
        elif self.live is not None:
            lib = _lib.load()
            with torch.cuda.device(self.buf.device), _bracket("zero_rows"):
                rc = lib.gsage_zero_rows(self.live.rows.data_ptr(), self.live.n_rows.data_ptr(),
                                         self.live.cap, self.buf.size(1), self.buf.data_ptr(),
                                         _lib.stream_handle())
            _lib.check(rc, "gsage_zero_rows")
        else:
            self.buf.zero_()
        self.dirty, self.live = False, None




def live_rows_worthwhile(N: int, D: int, B: int, K: int) -> bool:
    """Build the row list only where a consumer exists (the fused 64 -> 64 backward) and the
    batch cannot touch nearly every row anyway."""
    return LIVE_ROWS and D == 64 and (2 + K) * B <= 2 * N


def skipgram_fwd(emb, targets, contexts, negatives, scale, norm_b=None, norm_bk=None):
    lib = _lib.load()
    N, D = emb.shape
    B = targets.numel()
    K = negatives.shape[1] if negatives.dim() == 2 else 0
    dev = emb.device
    if norm_b is None:
        norm_b = float(B)
    if norm_bk is None:
        norm_bk = float(B * K) if K > 0 else 1.0
    with torch.cuda.device(dev):
        coef = torch.empty(B, 1 + K, dtype=torch.float32, device=dev)
        stats = torch.empty(8, dtype=torch.float32, device=dev)
        ws_bytes = lib.gsage_skipgram_workspace_bytes(B)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        with _bracket("skipgram_fwd"):
            rc = lib.gsage_skipgram_fwd(emb.data_ptr(), N, D, targets.data_ptr(), contexts.data_ptr(),
                                        negatives.data_ptr() if K > 0 else None, B, K, float(scale),
                                        float(norm_b), float(norm_bk), coef.data_ptr(),
                                        stats.data_ptr(), ws.data_ptr(), ws_bytes,
                                        _lib.stream_handle())
    _lib.check(rc, "gsage_skipgram_fwd")
    return coef, stats


# Loss backward: a float-atomic scatter, or the sort-based form (bitwise reproducible, no
# zero-fill).  Same values up to summation order.  Which is faster depends on how many of the
# 2(1+K)B terms land on each of the N rows: at B = 2^20, K = 5 the sorted form wins at 100 k and
# 1 M nodes (2.25 vs 3.35 and 9.6 vs 10.9 ms per step) and loses at 10 M (31.9 vs 30.1), and for
# small batches its dozen launches cost more than they save.  GSAGE_LOSS_BWD=atomic|det forces
# one, SkipGramLoss(deterministic=True/False) does the same per module; default "auto".
LOSS_BWD_MODE = os.environ.get("GSAGE_LOSS_BWD", "auto")
if LOSS_BWD_MODE not in ("auto", "atomic", "det"):
    raise ValueError(f"GSAGE_LOSS_BWD must be auto, atomic or det, not {LOSS_BWD_MODE!r}")
# A third form where the touched-row list exists (live_rows_worthwhile): sorted and summed over
# the LISTED rows only (gsage_skipgram_bwd_rows) -- deterministic like "det", without its costs
# that scale with N.  "auto" takes it whenever the list is built; GSAGE_LOSS_BWD_ROWS=0 keeps
# the atomics there.
LOSS_BWD_ROWS = os.environ.get("GSAGE_LOSS_BWD_ROWS", "1") != "0"


def loss_bwd_rowwise(N: int, D: int, B: int, K: int, deterministic: Optional[bool] = None) -> bool:
    """Does the loss backward of this shape run row-wise over the touched rows?"""
    if not live_rows_worthwhile(N, D, B, K) or D % 4 != 0:
        return False
    if deterministic is False or (deterministic is None and LOSS_BWD_MODE == "atomic"):
        return False
    if not LOSS_BWD_ROWS and deterministic is None and LOSS_BWD_MODE == "auto":
        return False
    return bool(_lib.load().gsage_skipgram_bwd_rows_supported(D))


def skipgram_bwd_rows(emb, targets, contexts, negatives, coef, grad_loss, grad_emb, live: LiveRows):
    """gsage_skipgram_bwd_rows: writes the listed rows of ``grad_emb`` (each in full, fixed term
    order), touches no other row."""
    lib = _lib.load()
    N, D = emb.shape
    B = targets.numel()
    K = coef.shape[1] - 1
    with torch.cuda.device(emb.device):
        ws_bytes = lib.gsage_skipgram_bwd_rows_workspace_bytes(live.cap, B, K)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=emb.device)
        with _bracket("skipgram_bwd_rows"):
            rc = lib.gsage_skipgram_bwd_rows(emb.data_ptr(), N, D, targets.data_ptr(),
                                             contexts.data_ptr(),
                                             negatives.data_ptr() if K > 0 else None, B, K,
                                             coef.data_ptr(), grad_loss.data_ptr(), grad_emb.data_ptr(),
                                             live.mask.data_ptr(), live.word_prefix.data_ptr(),
                                             live.rows.data_ptr(), live.n_rows.data_ptr(), live.cap,
                                             ws.data_ptr(), ws_bytes, _lib.stream_handle())
    _lib.check(rc, "gsage_skipgram_bwd_rows")
    return grad_emb


def loss_bwd_deterministic(N: int, B: int, K: int) -> bool:
    """The form skipgram_bwd takes when the caller does not say."""
    if LOSS_BWD_MODE != "auto":
        return LOSS_BWD_MODE == "det"
    terms = 2 * (1 + K) * B
    return terms >= (1 << 20) and terms >= 4 * N


def skipgram_bwd(emb, targets, contexts, negatives, coef, grad_loss, grad_emb=None,
                 deterministic: Optional[bool] = None):
    lib = _lib.load()
    N, D = emb.shape
    B = targets.numel()
    K = coef.shape[1] - 1
    if deterministic is None:
        deterministic = loss_bwd_deterministic(N, B, K)
    if deterministic and grad_emb is None:
        with torch.cuda.device(emb.device):
            grad_emb = torch.empty_like(emb)
            ws_bytes = lib.gsage_skipgram_bwd_det_workspace_bytes(N, B, K)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=emb.device)
            with _bracket("skipgram_bwd_det"):
                rc = lib.gsage_skipgram_bwd_det(emb.data_ptr(), N, D, targets.data_ptr(),
                                                contexts.data_ptr(),
                                                negatives.data_ptr() if K > 0 else None, B, K,
                                                coef.data_ptr(), grad_loss.data_ptr(),
                                                grad_emb.data_ptr(), ws.data_ptr(), ws_bytes,
                                                _lib.stream_handle())
        _lib.check(rc, "gsage_skipgram_bwd_det")
        return grad_emb
    with torch.cuda.device(emb.device):
        if grad_emb is None:
            grad_emb = torch.zeros_like(emb)
        with _bracket("skipgram_bwd_atomic"):
            rc = lib.gsage_skipgram_bwd(emb.data_ptr(), N, D, targets.data_ptr(), contexts.data_ptr(),
                                        negatives.data_ptr() if K > 0 else None, B, K, coef.data_ptr(),
                                        grad_loss.data_ptr(), grad_emb.data_ptr(),
                                        _lib.stream_handle())
    _lib.check(rc, "gsage_skipgram_bwd")
    return grad_emb


# -------------------------------------------------------------------------- autograd glue
class SAGELayerFunction(torch.autograd.Function):
    """One SAGEConv layer with a fused epilogue: K2 + K3 forward, K3' + K7 backward."""

    @staticmethod
    def forward(ctx, x, w_l, b_l, w_r, graph: GraphCSR, epilogue: int, p: float, seed: int,
                keep_mask, force_generic: bool, agg_cached=None, stash: Optional[dict] = None,
                amax_in=None, amax_out=None, upstream: Optional[dict] = None):
        x = _f32c(x, "x")
        w_l, b_l, w_r = _f32c(w_l, "lin_l.weight"), _f32c(b_l, "lin_l.bias"), _f32c(w_r, "lin_r.weight")
        if x.dim() != 2 or x.size(0) != graph.num_nodes:
            raise ValueError(f"x must be [num_nodes={graph.num_nodes}, F], got {tuple(x.shape)}")
        if w_l.shape != w_r.shape or w_l.size(1) != x.size(1) or b_l.numel() != w_l.size(0):
            raise ValueError("SAGEConv weight shapes do not match the input width")
        agg = agg_cached if agg_cached is not None else aggregate_mean(graph, x)
        # sign bits of the output for the backward, wanted only where the backward stops at this
        # layer's weights (the first layer): then ReLU'/dropout' is applied inside the dW kernel
        want_sign = not (ctx.needs_input_grad[0] or ctx.needs_input_grad[10])
        # a stash that asks for it ("want_u": the caller will run the next layer's aggregation on the
        # recompute gather) gets the 64-byte rows written by this layer's own kernel
        u_box = {} if (stash is not None and stash.get("want_u")) else None
        out, rnorm, sign = project_fwd(agg, x, w_l, b_l, w_r, epilogue, keep_mask, seed, p,
                                       force_generic, want_relu_sign=True,
                                       alloc_relu_sign=want_sign, u_out=u_box, amax_in=amax_in,
                                       amax_out=amax_out)
        ctx.graph, ctx.epilogue, ctx.p, ctx.force_generic = graph, epilogue, p, force_generic
        ctx.save_for_backward(x, agg, out, rnorm, w_l, w_r, sign)
        # ``stash`` (this layer's operands, for the NEXT layer) / ``upstream`` (the stash of the layer
        # in FRONT, handed in by a caller that vouches this layer is the only consumer of that
        # layer's output): the way the two backwards meet in gsage_hidden1_wgrad
        ctx.stash, ctx.upstream = stash, upstream
        if stash is not None:
            # what the next layer's aggregation can be recomputed from (aggregate_hidden1)
            stash.update(agg=agg, x=x, w_l=w_l, b_l=b_l, w_r=w_r, sign=sign, epilogue=epilogue, p=p,
                         out=out if epilogue == _lib.EPI_RELU_DROPOUT else None,
                         u=u_box.get("u") if u_box is not None else None,
                         weights_only=want_sign)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, agg, out, rnorm, w_l, w_r, sign = ctx.saved_tensors
        graph = ctx.graph
        need_x = ctx.needs_input_grad[0]
        deferred = ctx.stash.pop("deferred", None) if ctx.stash else None
        if deferred is not None:
            # the layer behind this one left what its backward knows instead of a gradient: this
            # layer's weight gradients come from there in one pass over the live edges
            grad_agg2, grad_x2, gmax, live2 = deferred
            gwl, gb, gwr = hidden1_wgrad(graph, ctx.stash["u"], grad_agg2, grad_x2, gmax, live2,
                                         ctx.epilogue, ctx.p)
            if grad_out.stride(0) != 0:
                # ... and somebody else used this layer's output too: that part the usual way
                _, _, gwl2, gb2, gwr2 = project_bwd(_f32c(grad_out, "grad_out"), out, rnorm, agg, x, w_l, w_r,
                                                    graph.inv_deg, ctx.epilogue, ctx.p, False,
                                                    ctx.force_generic, relu_sign=sign)
                gwl, gb, gwr = gwl + gwl2, gb + gb2, gwr + gwr2
            return None, gwl, gb, gwr, None, None, None, None, None, None, None, None, None, None, None
        grad_out = _f32c(grad_out, "grad_out")
        if ctx.needs_input_grad[10]:
            # the aggregation came from outside and wants its own gradient (row-subset layers,
            # graphsage_amd/receptive.py): hand back d/d agg unscaled, no transposed gather here
            grad_agg, grad_x, gwl, gb, gwr = project_bwd(
                grad_out, out, rnorm, agg, x, w_l, w_r, None, ctx.epilogue, ctx.p, True,
                ctx.force_generic)
            return (grad_x if need_x else None), gwl, gb, gwr, None, None, None, None, None, None, grad_agg, None, None, None, None
        live = getattr(grad_out, "_gsage_live", None)
        if (live is not None and need_x and not ctx.force_generic and SKIP_ZERO_ROWS
                and live.version == grad_out._version and live.num_rows == x.size(0)
                and _lib.load().gsage_project_bwd_rows_supported(x.size(1), w_l.size(0))):
            # the loss backward listed the rows of grad_out that can be non-zero: walk only those
            LIVE_STATS["used"] += 1
            live.consumed = True
            up = ctx.upstream
            if (HIDDEN1_WGRAD and up is not None and up.get("u") is not None and up.get("weights_only")
                    and live.rows is not None and live.cap <= x.size(0)
                    and _lib.load().gsage_hidden1_wgrad_supported(x.size(1), int(up["epilogue"]))):
                # the layer in front is the input layer and wants only its weight gradients: leave
                # it (grad_agg, grad_x, the row list) and hand autograd a gradient without storage
                WGRAD_STATS["used"] += 1
                gmax = torch.empty(1, dtype=torch.int32, device=x.device)
                grad_agg, grad_x, gwl, gb, gwr = project_bwd_rows(
                    grad_out, out, rnorm, agg, x, w_l, w_r, graph.inv_deg, ctx.epilogue, ctx.p, live,
                    gmax=gmax)
                up["deferred"] = (grad_agg, grad_x, gmax, live)
                return (grad_out.new_zeros(()).expand(x.shape), gwl, gb, gwr, None, None, None, None, None,
                        None, None, None, None, None, None)
            grad_agg, grad_x, gwl, gb, gwr = project_bwd_rows(
                grad_out, out, rnorm, agg, x, w_l, w_r, graph.inv_deg, ctx.epilogue, ctx.p, live)
            grad_x = aggregate_mean_bwd(graph, grad_agg, prescaled=True, add_to=grad_x,
                                        live_mask=live.mask)
            return grad_x, gwl, gb, gwr, None, None, None, None, None, None, None, None, None, None, None
        if live is not None and live.outside_undefined:
            # a persistent gradient table whose unlisted rows were left from an earlier step, and
            # this backward cannot go through the row list: rebuild the dense gradient (rare path)
            n_live = int(live.n_rows.item())
            idx = live.rows[:min(n_live, live.cap)].long()
            dense = torch.zeros_like(grad_out)
            dense[idx] = grad_out[idx]
            grad_out = dense
        grad_agg, grad_x, gwl, gb, gwr, nz = project_bwd(
            grad_out, out, rnorm, agg, x, w_l, w_r, graph.inv_deg, ctx.epilogue, ctx.p,
            need_x, ctx.force_generic, want_nz_mask=True, relu_sign=sign)
        if need_x:
            # grad_agg arrives pre-scaled by inv_deg[row]; gather it over the transposed CSR
            grad_x = aggregate_mean_bwd(graph, grad_agg, prescaled=True, add_to=grad_x,
                                        nz_mask=nz if SKIP_ZERO_ROWS else None)
        return grad_x, gwl, gb, gwr, None, None, None, None, None, None, None, None, None, None, None


class SkipGramFunction(torch.autograd.Function):
    """K4 forward / K5 backward.  Returns (loss 0-dim, stats[8])."""

    @staticmethod
    def forward(ctx, emb, targets, contexts, negatives, scale, norm_b, norm_bk, deterministic=None,
                grad_table: Optional[GradTable] = None):
        emb = _f32c(emb, "embeddings")
        targets, contexts = _i64c(targets, "targets"), _i64c(contexts, "contexts")
        negatives = _i64c(negatives, "negatives")
        if emb.dim() != 2:
            raise ValueError("embeddings must be [num_nodes, dim]")
        if targets.dim() != 1 or contexts.shape != targets.shape:
            raise ValueError("targets and contexts must be 1-D tensors of equal length")
        if negatives.dim() != 2 or negatives.size(0) != targets.numel():
            raise ValueError("negatives must be [batch_size, num_negatives]")
        coef, stats = skipgram_fwd(emb, targets, contexts, negatives, scale, norm_b, norm_bk)
        ctx.save_for_backward(emb, targets, contexts, negatives, coef)
        ctx.deterministic = deterministic
        ctx.grad_table = grad_table
        loss = stats[0].clone()
        ctx.mark_non_differentiable(stats)
        return loss, stats

    @staticmethod
    def backward(ctx, grad_loss, _grad_stats):
        emb, targets, contexts, negatives, coef = ctx.saved_tensors
        g = grad_loss.to(torch.float32).contiguous()
        K = coef.shape[1] - 1
        N, B = emb.size(0), targets.numel()
        want_live = live_rows_worthwhile(N, emb.size(1), B, K)
        rowwise = loss_bwd_rowwise(N, emb.size(1), B, K, ctx.deterministic)
        det = ctx.deterministic if ctx.deterministic is not None else loss_bwd_deterministic(N, B, K)
        live = touched_rows(targets, contexts, negatives, N) if want_live else None
        if rowwise:
            # sorted sums over the listed rows: every listed row written once and in full
            table = ctx.grad_table
            if table is not None:
                live.outside_undefined = table.stale and table.keep_rows
                grad_emb = table.take(emb, overwrites_rows=table.keep_rows)
            else:
                grad_emb = torch.zeros_like(emb)
            skipgram_bwd_rows(emb, targets, contexts, negatives, coef, g, grad_emb, live)
            live.rowwise = True
        else:
            table = ctx.grad_table if (want_live and not det and emb.size(1) % 4 == 0) else None
            grad_emb = skipgram_bwd(emb, targets, contexts, negatives, coef, g,
                                    grad_emb=table.take(emb) if table is not None else None,
                                    deterministic=det)
        if want_live:
            # rows outside the batch are exactly zero in grad_emb: tell the consumer which
            live.version = grad_emb._version
            try:                         # a fresh attribute each step on the persistent buffer
                grad_emb._gsage_live = live
            except AttributeError:       # pragma: no cover
                pass
            if table is not None:
                table.live = live
            LIVE_STATS["made"] += 1
        elif hasattr(grad_emb, "_gsage_live"):
            del grad_emb._gsage_live
        return grad_emb, None, None, None, None, None, None, None, None


class RowSubsetMean(torch.autograd.Function):
    """agg[r] = mean of x over the in-neighbours of the r-th row of a row subset, where ``graph``
    is the compact CSR of those edges (targets r < n_rows, sources index x).  Backward is the
    transposed gather with the 1/degree of the target row as the weight: deterministic."""

    @staticmethod
    def forward(ctx, x, graph: GraphCSR, n_rows: int):
        x = _f32c(x, "x")
        ctx.graph, ctx.n_src = graph, x.size(0)
        return segment_gather(graph.rowptr[:n_rows + 1], graph.col, x, row_scale=graph.inv_deg,
                              long_rows=graph.long_rows, n_long=graph.n_long[0:1])

    @staticmethod
    def backward(ctx, grad_agg):
        g = ctx.graph
        grad_x = segment_gather(g.rowptr_t, g.col_t, _f32c(grad_agg, "grad_agg"),
                                edge_weight=g.inv_deg, long_rows=g.long_rows_t,
                                n_long=g.n_long[1:2])
        return grad_x, None, None
