import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: redundant-by-design repetitions of a covered case (every tiny / narrow golden in the opt-in "
                                       "gate-GEMM modes, ...): skipped unless the marker expression names it -- `-m \"gpu and slow\"`")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` (the driver's run, 900 s limit) leaves the `slow` repetitions out; `-m "gpu and slow"` or DPRNN_TEST_SLOW=1 runs them."""
    if "slow" in (config.getoption("-m") or "") or os.environ.get("DPRNN_TEST_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow repetition: run with -m \"gpu and slow\" (or DPRNN_TEST_SLOW=1)")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(bytes(d["meta"]).decode())
    return d


def golden_names(prefix="", trained=False):
    """Fixture names; the trained-like fixtures (make_golden.TRAINED_CASES: other weight recipe, two-sided tolerance) are listed
    only on request and have their own tests."""
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f.startswith(prefix) and (("trained" in f) == trained))


def state_from_meta(meta):
    """The numpy-recipe state_dict a fixture was generated with (meta.style / meta.recipe: synthetic.make_state)."""
    from oracle.weights import make_state
    return make_state(meta["kind"], meta["cfg"], meta["seed"], style=meta.get("style", "init"), **(meta.get("recipe") or {}))


def model_from_meta(meta, device=None):
    return build_model(meta["kind"], meta["cfg"], meta["seed"], device, style=meta.get("style", "init"), recipe=meta.get("recipe"))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.fixture(scope="session")
def root():
    return ROOT


def build_model(kind, cfg, seed, device=None, style="init", recipe=None):
    """Our module (tss_with_dprnn_amd) with the numpy-recipe weights loaded strict=True (style: synthetic.make_state)."""
    import torch
    import tss_with_dprnn_amd as pkg
    from oracle.weights import full_cfg, make_state
    cfg = full_cfg(**cfg)
    base = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
            "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]
    spe = base + ["O", "P", "embeddings_size", "num_spks", "fusion_type"]
    if kind == "bss":
        model = pkg.DPRNNTasNet(**{k: cfg[k] for k in base})
    else:
        from tss_with_dprnn_amd import dprnn_spe, dprnn_spe_ira
        cls = dprnn_spe.DPRNNSpeTasNet if kind == "spe" else dprnn_spe_ira.DPRNNSpeIRATasNet
        model = cls(**{k: cfg[k] for k in spe})
    sd = make_state(kind, cfg, seed, style=style, **(recipe or {}))
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.eval()
    if device is not None:
        model = model.to(device)
    return model, sd, cfg
