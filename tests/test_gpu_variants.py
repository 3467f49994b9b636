"""Operator-surface variants on the GPU against vectors captured from the imported reference (tests/golden/ops2.npz, written by
tests/golden/make_golden_r2.py): integer activations and integer RMSNorm (integer outputs: bit exact), attention with a sliding
window and with the integer softmax (float reductions in a different order than torch's bmm: 2e-5 / 2e-3 absolute)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def z(golden_dir):
    return np.load(os.path.join(golden_dir, "ops2.npz"))


def test_integer_activations_bit_exact(dev, z):
    from vllm_i64_amd import layers as L
    x = torch.from_numpy(z["act_x"]).to(dev)
    assert np.array_equal(L.silu_integer(x).cpu().numpy(), z["act_silu"])
    assert np.array_equal(L.sigmoid_integer(x).cpu().numpy(), z["act_sigmoid"])
    assert np.array_equal(L.softplus_integer(x).cpu().numpy(), z["act_softplus"])
    assert L.silu_integer(x.long()).dtype == torch.int64 and np.array_equal(L.silu_integer(x.long()).cpu().numpy(), z["act_silu"])


def test_integer_rmsnorm(dev, z):
    from vllm_i64_amd import layers as L
    w = torch.from_numpy(z["rmsq_w"]).to(dev)
    # rows whose sum of squares is exact in fp32 in any order: every output bit equals the reference's
    ye = L.rmsnorm_integer(torch.from_numpy(z["rmsq_xe"]).to(dev), w, 1e-6).cpu().numpy()
    assert np.array_equal(ye, z["rmsq_ye"])
    # random rows: the reference's fp32 mean comes from torch's own summation order; a Q7 value can differ by one step where
    # x * rsqrt(...) * 128 sits on a rounding tie.  One step of Q7 x Q12 is |w| / 128.
    y = L.rmsnorm_integer(torch.from_numpy(z["rmsq_x"]).to(dev), w, 1e-6).cpu().numpy()
    diff = np.abs(y - z["rmsq_y"])
    step = np.abs(z["rmsq_w"].astype(np.float64)) / 4096.0 / 128.0
    assert (diff <= step[None, :] + 1e-9).all() and (diff > 0).mean() < 1e-3, ((diff > 0).mean(), diff.max())
    # the layer object with quantize() takes the integer path
    n = L.RMSNorm(torch.from_numpy(z["rmsq_w"].astype(np.float32) / 4096.0)).quantize()
    assert np.array_equal(n.weight_q12.numpy(), z["rmsq_w"])
    assert np.array_equal(n(torch.from_numpy(z["rmsq_xe"]).to(dev)).cpu().numpy(), z["rmsq_ye"])


@pytest.mark.parametrize("window", [None, 5])
def test_attention_sliding_window_and_integer_softmax(dev, z, window):
    from vllm_i64_amd import layers as L
    q, k, v = (torch.from_numpy(z[n]).to(dev) for n in ("at_q", "at_k", "at_v"))
    lens = [int(t) for t in z["at_lens"]]
    y = L.naive_varlen_attention(q, k, v, lens, 2, sliding_window=window).cpu().numpy()
    assert np.abs(y - z[f"at_varlen_f_{window}"]).max() <= 2e-5
    yi = L.naive_integer_varlen_attention(q, k, v, lens, 2, sliding_window=window).cpu().numpy()
    # a score within float rounding of a Q7 tie moves one table step: one weight changes by < 1 %
    assert np.abs(yi - z[f"at_varlen_i_{window}"]).max() <= 2e-3 and np.abs(yi - z[f"at_varlen_i_{window}"]).mean() <= 2e-5
    pos = torch.from_numpy(z["at_cached_pos"]).to(dev)
    qc, kf, vf = q[:5].contiguous(), k[:19].contiguous(), v[:19].contiguous()
    yc = L.naive_cached_attention(qc, kf, vf, 2, pos, sliding_window=window).cpu().numpy()
    assert np.abs(yc - z[f"at_cached_f_{window}"]).max() <= 2e-5
    yci = L.naive_integer_cached_attention(qc, kf, vf, 2, pos, sliding_window=window).cpu().numpy()
    assert np.abs(yci - z[f"at_cached_i_{window}"]).max() <= 2e-3 and np.abs(yci - z[f"at_cached_i_{window}"]).mean() <= 2e-5
    # the integer variant is NOT the float one: masked keys keep the table's smallest weight (the reference's behaviour)
    if window is None:
        assert np.abs(yi - y).max() > 1e-4


def test_integer_paged_decode_attention(dev, z):
    from vllm_i64_amd import layers as L
    d = lambda n: torch.from_numpy(z[n]).to(dev)
    out = L.naive_integer_paged_decode_attention(d("pdi_q"), d("pdi_k"), d("pdi_v"), d("pdi_bt"), d("pdi_sl"), num_kv_groups=2).cpu().numpy()
    assert not out[2].any()
    assert np.abs(out - z["pdi_out"]).max() <= 2e-3 and np.abs(out - z["pdi_out"]).mean() <= 2e-5


def test_moe_layer_with_learned_output_gates(dev, z):
    """use_shared_routed_gates (token_routed_mlp.py:84-90,174-178): shared_gate * shared + routed_gate * routed, against the
    reference module's captured int8 output; through the layer object and through the model runner's layer call."""
    from oracle import model_ref as M
    from vllm_i64_amd.config import ModelConfig
    from vllm_i64_amd.layers import TokenRoutedMLPInt8
    from vllm_i64_amd.model import I64MoEModel
    ocfg = M.named_config("tiny-moe")
    w = dict(M.make_weights(ocfg, seed=0))
    x, ids = torch.from_numpy(z["gate_x"]).to(dev), torch.from_numpy(z["gate_ids"]).to(dev)
    layer = TokenRoutedMLPInt8(ModelConfig.from_any(ocfg), {"exp_gate_up": w["l1.exp_gate_up"], "exp_down": w["l1.exp_down"],
                                                           "sh_gate_up": w["l1.sh_gate_up"], "sh_down": w["l1.sh_down"],
                                                           "table": w["l1.table"], "shared_gate": torch.tensor(0.8),
                                                           "routed_gate": torch.tensor(1.3)}, device=dev)
    y = layer(x, ids).cpu().numpy()
    assert np.abs(y - z["gate_y"]).max() <= 1e-2 and np.abs(y - z["gate_y"]).mean() <= 1e-5
    w["l1.shared_gate"], w["l1.routed_gate"] = torch.tensor(0.8), torch.tensor(1.3)
    model = I64MoEModel(ModelConfig.from_any(ocfg), w, device=dev, max_seqs=2, num_kv_blocks=8)
    y2 = model.moe_layer(1, x, ids).cpu().numpy()
    assert np.abs(y2 - z["gate_y"]).max() <= 1e-2 and np.abs(y2 - z["gate_y"]).mean() <= 1e-5
    # without the gates the configured scales (1.5 / 0.75) apply: a different result
    y3 = model.moe_layer(0, x, ids).cpu().numpy()
    assert np.abs(y3 - z["gate_y"]).max() > 1e-2
