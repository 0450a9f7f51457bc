"""GPU: inference encode (BASELINE config 5).  A checkpoint in the reference's
layout (src/training/trainer.py:385-391: epoch, model_state_dict with PyG's key names,
optimizer_state_dict, best_loss, config) written from the CPU oracle model -- i.e. by "the
reference side" -- is loaded by StructuralEncoder(model_path=...) with weights_only, and
encode_tensors (src/inference/encoder.py:181-185) is compared with the oracle forward."""
import pytest
import torch

from oracle import sage_oracle as O
from util import assert_close

pytestmark = pytest.mark.gpu


def _reference_layout_checkpoint(path, hidden, out, layers, seed=0):
    torch.manual_seed(seed)
    ref = O.OracleGraphSAGE(7, hidden, out, layers, 0.3)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    cfg = {"features": {"dimensions": 7},
           "model": {"hidden_dim": hidden, "output_dim": out, "num_layers": layers, "dropout": 0.3,
                     "aggregator": "mean", "normalize_output": True},
           "training": {"batch_size": 1024, "learning_rate": 1e-3}}
    torch.save({"epoch": 3, "model_state_dict": ref.state_dict(),
                "optimizer_state_dict": opt.state_dict(), "best_loss": 1.25, "config": cfg}, path)
    return ref.eval()


@pytest.mark.parametrize("hidden,out,layers,n", [(64, 64, 2, 3000), (256, 256, 3, 1200)])
def test_encoder_loads_reference_layout_checkpoint_and_matches_oracle(tmp_path, hidden, out, layers, n):
    from graphsage_amd import StructuralEncoder
    from graphsage_amd.synth import memory_features, powerlaw_graph
    path = str(tmp_path / "model_best.pt")
    ref = _reference_layout_checkpoint(path, hidden, out, layers)
    enc = StructuralEncoder(model_path=path, device="cuda")
    assert enc.model.training is False and enc.model.dropout_rate == 0.0     # encoder.py:133
    assert (enc.model.hidden_channels, enc.model.out_channels, enc.model.num_layers) == (hidden, out, layers)
    assert enc.get_embedding_dim() == out
    ei = powerlaw_graph(n, 12, seed=7)
    x = memory_features(n, seed=7)
    emb = enc.encode_tensors(x, ei)                 # CPU tensors in, as encoder.py:181-182 moves them
    assert emb.is_cuda and emb.shape == (n, out) and not emb.requires_grad
    with torch.no_grad():
        want = ref(x, ei)
    assert_close(emb, want, what=f"{layers}x{hidden}")
    assert_close(emb.norm(dim=1).cpu(), torch.ones(n), rtol=1e-5, floor=1e-5)
    # same tensors again: same bits (deterministic kernels), and no cache growth from temporaries
    from graphsage_amd import default_cache
    xg, eg = x.cuda(), ei.cuda()
    a = enc.encode_tensors(xg, eg)
    pads = len(default_cache()._pads)
    b = enc.encode_tensors(xg, eg)
    assert torch.equal(a, b) and len(default_cache()._pads) == pads
    assert_close(a, want)


def test_encoder_falls_back_to_its_config_and_rejects_wrong_keys(tmp_path):
    from graphsage_amd import StructuralEncoder
    torch.manual_seed(1)
    ref = O.OracleGraphSAGE(7, 32, 16, 2, 0.0)
    p1 = str(tmp_path / "noconfig.pt")
    torch.save({"model_state_dict": ref.state_dict()}, p1)       # no 'config' key: encoder.py:121-123
    enc = StructuralEncoder(model_path=p1, device="cuda",
                            config={"model": {"hidden_dim": 32, "output_dim": 16}})
    assert enc.model.out_channels == 16
    with pytest.raises(RuntimeError):                            # strict load: shapes must fit
        StructuralEncoder(model_path=p1, device="cuda")
    with pytest.raises(ValueError):
        StructuralEncoder()


def test_encoder_benchmark_protocol():
    from graphsage_amd import GraphSAGE, StructuralEncoder
    from graphsage_amd.synth import memory_features, powerlaw_graph
    enc = StructuralEncoder(model=GraphSAGE(7, 64, 64, 2, 0.0), device="cuda")
    r = enc.benchmark(memory_features(2000, seed=1), powerlaw_graph(2000, 8, seed=1), num_runs=5, warmup=1)
    assert r["num_nodes"] == 2000 and 0 < r["min_ms"] <= r["mean_ms"] <= r["max_ms"]


def test_full_size_c5_encode_properties():
    """BASELINE config 5 at its stated size (10 M nodes, ~200 M edges, 7->256->256->256), beyond
    the CPU oracle: every row has unit norm, everything is finite, a second encode gives the same
    bits, and rows of isolated nodes equal the closed form normalize(W_r-path only), computed with
    torch on the GPU in float64."""
    from graphsage_amd import GraphSAGE, StructuralEncoder, default_cache
    from graphsage_amd.synth import memory_features, powerlaw_graph
    n = 10_000_000
    ei = powerlaw_graph(n, 20, seed=42, device="cuda")
    x = memory_features(n, seed=42, device="cuda")
    torch.manual_seed(0)
    enc = StructuralEncoder(model=GraphSAGE(7, 256, 256, 3, 0.0), device="cuda")
    emb = enc.encode_tensors(x, ei)
    assert emb.shape == (n, 256)
    assert bool(torch.isfinite(emb).all())
    norms = emb.norm(dim=1)
    assert float((norms - 1).abs().max()) <= 1e-5
    chk = emb.double().sum(0)
    del norms
    again = enc.encode_tensors(x, ei)
    assert torch.equal(again, emb)
    del again
    # isolated nodes (no in-edges): every layer reduces to b_l + W_r h, a closed form per node
    g = default_cache().get(ei, n)
    iso = (g.in_degree() == 0).nonzero().flatten()[:4096]
    assert iso.numel() > 0
    h = x[iso].double()
    convs = enc.model.convs
    for i, conv in enumerate(convs):
        h = h @ conv.lin_r.weight.double().t() + conv.lin_l.bias.double()
        if i < len(convs) - 1:
            h = h.relu()
    h = torch.nn.functional.normalize(h, dim=1)
    assert float((emb[iso].double() - h).abs().max()) <= 2e-6
    assert bool(torch.isfinite(chk).all())


@pytest.mark.parametrize("hidden,out,layers,n", [(256, 256, 3, 5000), (128, 128, 2, 7001), (256, 128, 4, 3000),
                                                 (128, 256, 3, 2500)])
def test_blocked_inference_layout_matches_plain_layout_and_oracle(hidden, out, layers, n):
    """Wide models without autograd keep their hidden activations column-blocked ([F/64, N, 64])
    from a table size on (ops.BLOCKED_MIN_BYTES).  Same arithmetic; the neighbour sums are taken
    by the 64-wide gather instead of the 128/256-wide one, whose lanes add the neighbours in a
    different order, so the two layouts agree to fp32 rounding, each is repeatable bit for bit,
    and both match the oracle."""
    from graphsage_amd import GraphSAGE, ops
    from graphsage_amd.synth import memory_features, powerlaw_graph
    ei = powerlaw_graph(n, 14, seed=3, max_degree=2000)
    x = memory_features(n, seed=3)
    torch.manual_seed(2)
    ref = O.OracleGraphSAGE(7, hidden, out, layers, 0.0).eval()
    m = GraphSAGE(7, hidden, out, layers, 0.0)
    m.load_state_dict(ref.state_dict())
    m = m.cuda().eval()
    xg, eg = x.cuda(), ei.cuda()
    saved = ops.BLOCKED_MIN_BYTES
    try:
        with torch.no_grad():
            ops.BLOCKED_MIN_BYTES = 1 << 60
            assert not m._blocked_inference_applies(xg)
            plain = m(xg, eg)
            ops.BLOCKED_MIN_BYTES = 0
            assert m._blocked_inference_applies(xg)
            blocked = m(xg, eg)
            assert torch.equal(blocked, m(xg, eg))           # deterministic
        assert not m._blocked_inference_applies(xg)          # autograd on: the plain path
        m.train()
    finally:
        ops.BLOCKED_MIN_BYTES = saved
    assert blocked.shape == (n, out)
    assert_close(blocked, plain, what="blocked vs plain layout")
    with torch.no_grad():
        want = ref(x, ei)
    assert_close(blocked, want, what="blocked vs oracle")
    assert_close(plain, want, what="plain vs oracle")
