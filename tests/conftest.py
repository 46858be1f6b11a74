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


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(bytes(d["meta"]).decode())
    return d


def golden_names(prefix=""):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz") and f.startswith(prefix))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.fixture(scope="session")
def root():
    return ROOT


def build_model(kind, cfg, seed, device=None):
    """Our module (tss_with_dprnn_amd) with the numpy-recipe weights loaded strict=True."""
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
    sd = make_state(kind, cfg, seed)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.eval()
    if device is not None:
        model = model.to(device)
    return model, sd, cfg
