"""Pin the CPU oracle (oracle/) against golden vectors produced by the reference classes
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_names, load_golden, rel_l2, state_from_meta
from oracle import dprnn_numpy as onp
from oracle import dprnn_torch_cpu as otc
from oracle.weights import full_cfg, make_mixture, make_state, state_spec

TOL = 2e-6      # rel-L2 of an fp32 restatement vs the reference's fp32 CPU output
SMALL = [n for n in golden_names() if "full" not in n and "k250" not in n]


def _run_numpy(g, dtype=np.float32, taps=None):
    m = g["meta"]
    sd = make_state(m["kind"], m["cfg"], m["seed"])
    if m["kind"] == "bss":
        return onp.tasnet_forward(sd, m["cfg"], g["mix"], dtype, taps), None
    fn = onp.spe_forward if m["kind"] == "spe" else onp.ira_forward
    return fn(sd, m["cfg"], g["mix"], g["aux"], g["aux_len"], m["bn_training"], dtype, taps)


def _run_torch(g):
    m = g["meta"]
    sd = otc.to_torch_state(make_state(m["kind"], m["cfg"], m["seed"]))
    if m["kind"] == "bss":
        return otc.tasnet_forward(sd, m["cfg"], torch.from_numpy(g["mix"])).numpy(), None
    fn = otc.spe_forward if m["kind"] == "spe" else otc.ira_forward
    est, logits = fn(sd, m["cfg"], torch.from_numpy(g["mix"]), torch.from_numpy(g["aux"]),
                     torch.tensor(float(g["aux_len"])), m["bn_training"])
    return est.numpy(), logits.numpy()


@pytest.mark.parametrize("name", SMALL)
def test_numpy_oracle_matches_reference(name):
    g = load_golden(name)
    out, logits = _run_numpy(g)
    assert out.shape == g["out"].shape
    assert rel_l2(out, g["out"]) < TOL
    if logits is not None:
        assert rel_l2(logits, g["logits"]) < TOL


@pytest.mark.parametrize("name", [n for n in SMALL if "tiny" in n])
def test_numpy_oracle_fp64_bounds_reference_noise(name):
    g = load_golden(name)
    out, _ = _run_numpy(g, np.float64)
    assert rel_l2(out, g["out"]) < 5e-6


@pytest.mark.parametrize("name", ["bss_tiny_ln_sigmoid", "bss_tiny_gln_relu", "bss_wide_small", "spe_tiny_cat",
                                  "spe_wide_small_cat"])
def test_numpy_oracle_intermediates(name):
    g = load_golden(name)
    taps = {}
    _run_numpy(g, np.float32, taps)
    checked = 0
    for k in g:
        if not k.startswith("tap_"):
            continue
        key = k[4:]
        if key not in taps:
            continue
        assert taps[key].shape == g[k].shape, key
        assert rel_l2(taps[key], g[k]) < TOL, key
        checked += 1
    assert checked >= 2


@pytest.mark.parametrize("name", [n for n in golden_names() if ("cat" in n or n.startswith("bss")) and
                                  "gru" not in n and "rnn" not in n or n in ("bss_tiny_gru_unidir", "bss_tiny_rnn", "spe_full_b1",
                                                                             "ira_full_b1", "ira_full_long_b1")])
def test_torch_cpu_oracle_matches_reference(name):
    g = load_golden(name)
    if g["meta"]["kind"] != "bss" and g["meta"]["cfg"]["fusion_type"] != "cat":
        pytest.skip("torch-cpu oracle restates cat fusion only")
    out, logits = _run_torch(g)
    assert out.shape == g["out"].shape
    assert rel_l2(out, g["out"]) < TOL
    if logits is not None:
        assert rel_l2(logits, g["logits"]) < TOL


def test_state_contract_matches_reference():
    with open(os.path.join(GOLDEN, "state_contract.json")) as f:
        contract = json.load(f)
    for tag, c in contract.items():
        kind, fus, nt = tag.split("/")
        spec = state_spec(kind, dict(fusion_type=fus, norm_type=nt))
        assert [(k, list(s)) for k, s in spec] == [(k, list(s)) for k, s in c["keys"]], tag
    assert contract["bss/cat/ln"]["n_params"] == 3651073
    assert contract["spe/cat/ln"]["n_params"] == 4038658
    assert contract["ira/cat/ln"]["n_params"] == 4071554


def test_recipe_is_deterministic():
    a = make_state("bss", dict(n_repeats=1), 7)
    b = make_state("bss", dict(n_repeats=1), 7)
    c = make_state("bss", dict(n_repeats=1), 8)
    k = "separation.dprnn_blocks.0.intra_rnn.rnn.weight_hh_l0"
    assert np.array_equal(a[k], b[k]) and not np.array_equal(a[k], c[k])
    assert np.array_equal(make_mixture(2, 50, 3), make_mixture(2, 50, 3))
    g = load_golden("bss_tiny_b1")
    assert np.array_equal(g["mix"], make_mixture(1, 64, 1000 + g["meta"]["seed"]))


def test_segmentation_overlap_add_identity():
    """OLA(seg(x)) = 2x when hop = K/2 (dprnn.py:189-217)."""
    x = np.random.default_rng(0).standard_normal((2, 3, 101)).astype(np.float32)
    for K, P in [(10, 5), (16, 8), (250, 125)]:
        seg = onp.segmentation(x, K, P)
        assert seg.shape[-1] == onp.n_chunks(101, K, P)
        assert np.array_equal(onp.overlap_add(seg, 101, K, P), 2 * x)


GRAD_SMALL = [n for n in golden_names("grad_bss") if "full" not in n]


@pytest.mark.parametrize("name", GRAD_SMALL)
def test_oracle_training_step_matches_reference_autograd(name):
    """Backward of the CPU restatement (stock ATen operators + the restated asteroid PIT SI-SDR loss) against the
    gradients the reference model classes produced: pins the oracle the training kernels are checked with."""
    import zlib
    from oracle.sisdr import pit_neg_sisdr_torch
    g = load_golden(name)
    m = g["meta"]
    sd = {k: torch.from_numpy(np.asarray(v)).requires_grad_(True) for k, v in make_state("bss", m["cfg"], m["seed"]).items()}
    out = otc.tasnet_forward_autograd(sd, m["cfg"], torch.from_numpy(g["mix"]))
    assert rel_l2(out.detach().numpy(), g["out"]) < TOL
    loss, idx = pit_neg_sisdr_torch(out, torch.from_numpy(g["tgt"]))
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    assert np.array_equal(idx.numpy(), g["perm_index"])
    loss.backward()
    for k, v in sd.items():
        got, nrm = v.grad.numpy(), float(g["norm_" + k])
        if "grad_" + k in g:
            err = np.linalg.norm(got.astype(np.float64) - g["grad_" + k]) / nrm
        else:
            rng = np.random.default_rng([zlib.crc32(k.encode()), got.size])
            pr = rng.standard_normal((8, got.size)).astype(np.float32).astype(np.float64) @ got.reshape(-1).astype(np.float64)
            err = np.abs(pr - g["proj_" + k]).max() / nrm
        assert err < 5e-6, (k, err)


@pytest.mark.parametrize("name", ["grad_spe_wide_cat", "grad_spe_wide_cat_bneval"])
def test_oracle_spe_training_step_matches_reference_autograd(name):
    """The same for DPRNN-Spe ('cat' fusion; BatchNorm in train and eval mode): trainer_spe.py:36-45."""
    import zlib
    from oracle.sisdr import pit_neg_sisdr_torch
    g = load_golden(name)
    m = g["meta"]
    sd = {}
    for k, v in make_state("spe", m["cfg"], m["seed"]).items():
        t = torch.from_numpy(np.asarray(v))
        sd[k] = t.requires_grad_(True) if t.is_floating_point() and "running_" not in k else t
    est, logits = otc.spe_forward_autograd(sd, m["cfg"], torch.from_numpy(g["mix"]), torch.from_numpy(g["aux"]),
                                           torch.tensor(float(g["aux_len"])), m["bn_training"])
    l, _ = pit_neg_sisdr_torch(est.unsqueeze(1), torch.from_numpy(g["tgt"]).unsqueeze(1))
    loss = l + 0.5 * torch.nn.CrossEntropyLoss()(logits, torch.from_numpy(g["spk_idx"]))
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    loss.backward()
    checked = 0
    for k, v in sd.items():
        if "norm_" + k not in g:
            continue
        got, nrm = v.grad.numpy(), float(g["norm_" + k])
        if "grad_" + k in g:
            err = np.linalg.norm(got.astype(np.float64) - g["grad_" + k]) / nrm
        else:
            rng = np.random.default_rng([zlib.crc32(k.encode()), got.size])
            pr = rng.standard_normal((8, got.size)).astype(np.float32).astype(np.float64) @ got.reshape(-1).astype(np.float64)
            err = np.abs(pr - g["proj_" + k]).max() / nrm
        assert err < 5e-6, (k, err)
        checked += 1
    assert checked > 60


# ---- trained-like fixtures (tests/golden/make_golden.py:TRAINED_CASES): heavy-tailed LSTM rows, forget-gate offsets, norm gains
# far from 1.  Such a DPRNN is close to a chaotic map, so the fixtures carry the reference's own fp32-vs-float64 distance and the
# float64 output; a restatement is pinned by (a) its fp32 form within max(TOL, 3 x that distance) of the reference's fp32 output
# and (b) its float64 form within 1e-9 of the reference's float64 output (same graph, no rounding to speak of).
@pytest.mark.parametrize("name", [n for n in golden_names("bss", trained=True) if "full" not in n])
def test_oracles_match_reference_on_trained_like_weights(name):
    g = load_golden(name)
    m = g["meta"]
    gap = m["ref_fp32_vs_fp64_rel_l2"]
    sd = state_from_meta(m)
    out32 = onp.tasnet_forward(sd, m["cfg"], g["mix"], np.float32)
    out64 = onp.tasnet_forward(sd, m["cfg"], g["mix"], np.float64)
    assert rel_l2(out64, g["out64"]) < 1e-9
    assert rel_l2(out32, g["out"]) < max(TOL, 3 * gap), (rel_l2(out32, g["out"]), gap)
    out_t = otc.tasnet_forward(otc.to_torch_state(sd), m["cfg"], torch.from_numpy(g["mix"])).numpy()
    assert rel_l2(out_t, g["out"]) < max(TOL, 3 * gap)


def test_torch_cpu_oracle_matches_reference_on_trained_like_weights_full_size():
    g = load_golden("bss_full_trained_rms1")
    m = g["meta"]
    out_t = otc.tasnet_forward(otc.to_torch_state(state_from_meta(m)), m["cfg"], torch.from_numpy(g["mix"])).numpy()
    assert rel_l2(out_t, g["out"]) < max(TOL, 3 * m["ref_fp32_vs_fp64_rel_l2"])
    assert rel_l2(out_t, g["out64"]) < 3 * m["ref_fp32_vs_fp64_rel_l2"]


@pytest.mark.parametrize("name", golden_names("spe", trained=True) + golden_names("ira", trained=True))
def test_oracles_match_reference_on_trained_like_weights_speaker_conditioned(name):
    """The Spe / IRA families under the same weight statistics (make_golden.TRAINED_CASES): both restatements against the reference's
    fp32 output and logits, the numpy one in float64 against the reference's float64 run (small case only: the shipped size takes the
    numpy oracle minutes)."""
    g = load_golden(name)
    m = g["meta"]
    gap = m["ref_fp32_vs_fp64_rel_l2"]
    sd = state_from_meta(m)
    if m["cfg"]["fusion_type"] == "cat":                 # the torch-CPU restatement covers the cat fusion (the others: numpy)
        fwd_t = otc.spe_forward if m["kind"] == "spe" else otc.ira_forward
        out_t, logits_t = fwd_t(otc.to_torch_state(sd), m["cfg"], torch.from_numpy(g["mix"]), torch.from_numpy(g["aux"]),
                                torch.tensor(float(g["aux_len"])))
        assert rel_l2(out_t.numpy(), g["out"]) < max(TOL, 3 * gap), (rel_l2(out_t.numpy(), g["out"]), gap)
        assert rel_l2(logits_t.numpy(), g["logits"]) < max(TOL, 3 * gap)
    if "full" not in name:
        fwd_n = onp.spe_forward if m["kind"] == "spe" else onp.ira_forward
        out64, logits64 = fwd_n(sd, m["cfg"], g["mix"], g["aux"], float(g["aux_len"]), dtype=np.float64)
        assert rel_l2(out64, g["out64"]) < 1e-9 and rel_l2(logits64, g["logits64"]) < 1e-9
        out32, _ = fwd_n(sd, m["cfg"], g["mix"], g["aux"], float(g["aux_len"]), dtype=np.float32)
        assert rel_l2(out32, g["out"]) < max(TOL, 3 * gap)
