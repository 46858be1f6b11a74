#!/usr/bin/env python3
"""DPRNN-Spe (config 3: B=64, 3 s mixture + 3 s reference) forward: utt/s and the per-entry-point HIP-event breakdown."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import engine
from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_state

BASE = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
        "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]
SPE = BASE + ["O", "P", "embeddings_size", "num_spks", "fusion_type"]
dev = "cuda:0"
fusion = sys.argv[1] if len(sys.argv) > 1 else "cat"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = full_cfg(norm_type="ln", fusion_type=fusion)
m = pkg.DPRNNSpeTasNet(**{k: cfg[k] for k in SPE}).eval()
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state("spe", cfg, 0).items()})
m = m.to(dev)
x = torch.from_numpy(make_mixture(B, 24000, seed=2)).to(dev)
a = torch.from_numpy(make_mixture(B, 24000, seed=3)).to(dev)
ln = torch.tensor(24000.0)
with torch.no_grad():
    for _ in range(2):
        m(x, a, ln)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m(x, a, ln)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"DPRNN-Spe ({fusion}) B={B}: {dt * 1e3:.1f} ms/step = {B / dt:.1f} utt/s")
    names = set(pkg._lib.SIGNATURES)
    engine.PROFILE = {"names": names, "events": {}}
    m(x, a, ln)
    torch.cuda.synchronize()
    prof, engine.PROFILE = engine.PROFILE, None
rows = sorted(((sum(e0.elapsed_time(e1) for e0, e1 in ev), len(ev), n) for n, ev in prof["events"].items()), reverse=True)
tot = sum(r[0] for r in rows)
print(f"profiled forward: {tot:.1f} ms inside C-ABI calls (launches of the stream groups overlap)")
for ms, cnt, n in rows[:18]:
    print(f"  {n:34s} {cnt:4d} calls {ms:9.2f} ms {100 * ms / tot:5.1f} %")
