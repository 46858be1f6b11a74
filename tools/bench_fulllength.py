#!/usr/bin/env python3
"""Config 5 alone (DPRNN-Spe-IRA, full-length utterances from the Libri2Mix-test length distribution, B=1 semantics
through inference.separate_many): sweep of the sequence tile and the stream count on one MI355X."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("HWQ", "16"))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import engine, inference, sharding
from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_state

BASE = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
        "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]
SPE = BASE + ["O", "P", "embeddings_size", "num_spks", "fusion_type"]
dev = "cuda:0"
kind = sys.argv[1] if len(sys.argv) > 1 else "ira"
cfg = full_cfg(norm_type="ln")
cls = {"bss": pkg.DPRNNTasNet, "spe": pkg.DPRNNSpeTasNet, "ira": pkg.DPRNNSpeIRATasNet}[kind]
m = cls(**{k: cfg[k] for k in (BASE if kind == "bss" else SPE)}).eval()
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state(kind, cfg, 0).items()})
m = m.to(dev)
rng = np.random.default_rng(0)
lens = np.clip(np.exp(rng.normal(np.log(36720), 0.38, 3000)), 24000, 111920).astype(int)
mine = sharding.shard_by_length(lens.tolist(), 8)[0][:48]
wavs = [torch.from_numpy(make_mixture(1, int(lens[i]), seed=100 + i)[0]).to(dev) for i in mine]
aux = None if kind == "bss" else [(torch.from_numpy(make_mixture(1, 24000, seed=900 + i)[0]).to(dev), 24000.0) for i in mine]
secs = float(sum(lens[i] for i in mine)) / 8000.0
with torch.no_grad():
    for tile in (0, 16, 32):
        engine.SEQ_TILE["intra"] = engine.SEQ_TILE["inter"] = tile
        for ns in (8, 16, 48):
            fn = lambda: inference.separate_many(m, wavs, max_streams=ns, extra_inputs=aux)
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter(); fn(); fn(); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 2
            print(f"{kind} tile={tile:2d} streams={ns:2d}: {dt:.3f} s = {len(mine) / dt:6.1f} utt/s, {secs / dt:5.0f}x real time", flush=True)
