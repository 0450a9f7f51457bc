"""GPU: a slice of the randomised differential campaign of tests/fuzz_parity.py (random multigraphs with
duplicate edges, self-loops, isolated nodes and hub rows; widths 64 / 128 / 256, 1-3 layers, dropout with
injected masks, batches with repeated ids, features at scales 1e-3 .. 1e3, the recompute gather forced on
or off) through GraphSAGE.forward, SkipGramLoss and autograd against the oracle: embeddings, loss and
every parameter gradient at the 1e-5 bar on the exact (float64) value, the fp32 oracle's own error as the
yardstick where fp32 cannot reach that, and ReLU kinks explained column by column.  The reference's
forward and loss are graphsage.py:116-144 and loss.py:97-136; 33 000 seeds of the full campaign ran clean
on the round-3 sources."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu



@pytest.mark.parametrize("first", [0, 40, 80])
def test_random_cases_against_the_oracle(first):
    import fuzz_parity
    from graphsage_amd import ops
    keep = ops.HIDDEN1_MIN_BYTES
    try:
        for seed in range(first, first + 40):
            fuzz_parity.one_case(seed)
    finally:
        ops.HIDDEN1_MIN_BYTES = keep


@pytest.mark.parametrize("first", [0, 60])
def test_random_inference_cases_against_the_oracle(first):
    """The same for the inference forward (encoder.py:181-185: eval, no_grad), with the column-blocked
    layout of the wide models and the recompute gather forced on or off; 3 000 seeds ran clean."""
    import fuzz_parity
    from graphsage_amd import ops
    keep = (ops.HIDDEN1_MIN_BYTES, ops.BLOCKED_MIN_BYTES)
    try:
        for seed in range(first, first + 60):
            fuzz_parity.one_eval_case(seed)
    finally:
        ops.HIDDEN1_MIN_BYTES, ops.BLOCKED_MIN_BYTES = keep


def test_random_loss_cases_against_float64():
    """SkipGramLoss alone (loss.py:97-136, 161-193): embedding widths 1 .. 300, 1-9 negatives, scale factors
    0.25 .. 20, normalised and raw embeddings, repeated ids; loss, the seven details and d loss / d emb
    against float64 autograd of the oracle's restatement.  3 000 seeds ran clean."""
    import fuzz_parity
    for seed in range(150):
        fuzz_parity.one_loss_case(seed)


def test_random_in_kernel_dropout_cases_against_the_host_mirror():
    """relu -> dropout (graphsage.py:134-135) with the mask drawn IN the projection kernel, the path the
    benchmark times: every kernel family (first layer 8 -> 64/128/256, 64 -> 64 fp32 and fp16, the wide split
    GEMMs, the generic kernel), ragged row counts, p = 0.1 .. 0.9, 64-bit seeds; output = relu(pre) x mask /
    (1 - p) with the host mirror's mask (graphsage_amd/rng.py) and pre in float64.  3 000 seeds ran clean."""
    import fuzz_parity
    for seed in range(150):
        fuzz_parity.one_dropout_case(seed)


def test_random_graphs_csr_bit_exact_and_gathers_against_float64():
    """K1 + K2/K7 alone on random multigraphs (duplicates, self-loops, isolated nodes, hub rows): rowptr / col
    in both orientations equal bincount + stable argsort bit for bit, inv_deg equal as floats,
    the mean aggregation and its transposed backward at any width against float64.  3 000 seeds ran clean."""
    import fuzz_parity
    for seed in range(150):
        fuzz_parity.one_graph_case(seed)


def test_random_training_steps_against_float64_from_the_same_point():
    """Three steps of the trainer's loop body (trainer.py:213-240) through TrainStep per case.  Two correct
    fp32 trajectories part ways at the first ReLU kink a 1e-7 parameter difference crosses, so every step is
    judged from where it starts: the float64 oracle is given our parameters, both step; loss details at 1e-5,
    clipped gradients tensor by tensor at 2e-5 (relative Frobenius), a remaining difference only when a
    hidden pre-activation sits on the kink -- then in its column, with the layers behind equal up to the one
    factor clip_grad_norm_ introduces.  Dropout 0 / 0.3 / 0.5 with the masks drawn in the kernels from TrainStep's
    (base seed, step, layer) stream and mirrored on the host for the oracle.  8 000 cases (24 000 steps) ran clean."""
    import fuzz_parity
    from graphsage_amd import ops
    keep = ops.HIDDEN1_MIN_BYTES
    try:
        for seed in range(60):
            fuzz_parity.one_step_case(seed)
    finally:
        ops.HIDDEN1_MIN_BYTES = keep


def test_random_feeder_cases_bit_exact():
    """The integer work either side of the path on random inputs: the pair list of ragged
    walks, order included, against the restatement of pair_sampler.py:56-86; alias tables against the
    restatement of negative_sampler.py:76-141 (itself pinned by the reference-run golden tables); the
    kernel's draws against the host mirror of its counter hash, counters up to 2^34.  600 seeds ran clean."""
    import fuzz_parity
    for seed in range(40):
        fuzz_parity.one_feeder_case(seed)


def test_random_captured_steps_against_eager_steps():
    """The step replayed as a HIP graph (TrainStep.capture_step, the trainer's path on small graphs) against
    the same steps launched eagerly on random graphs, widths and batch shapes: stats of every step and the
    parameters after each (bitwise in about a third of the cases, within 2e-6 in the rest).  600 cases ran clean."""
    import fuzz_parity
    from graphsage_amd import ops
    keep = ops.HIDDEN1_MIN_BYTES
    try:
        for seed in range(30):
            fuzz_parity.one_captured_case(seed)
    finally:
        ops.HIDDEN1_MIN_BYTES = keep
