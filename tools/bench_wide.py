#!/usr/bin/env python3
"""A DPRNN-TasNet WIDER than the kernels' build on the any-width kernels (wide.py): utt/s at B x 3 s and the per-entry-point
HIP-event breakdown.  usage: tools/bench_wide.py [feature_size] [hidden_size] [B] [input_size]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import engine
from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_state

BASE = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
        "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]
dev = "cuda:0"
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
N = int(sys.argv[4]) if len(sys.argv) > 4 else 64
cfg = full_cfg(feature_size=F, hidden_size=H, input_size=N)
m = pkg.DPRNNTasNet(**{k: cfg[k] for k in BASE}).eval()
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state("bss", cfg, 0).items()})
m = m.to(dev)
x = torch.from_numpy(make_mixture(B, 24000, seed=2)).to(dev)
geo = engine.make_geometry(B, 24000, N, cfg["chunk_length"], cfg["hop_length"], cfg["kernel_size"], cfg["stride"])
R = cfg["n_repeats"]
# algorithmic FLOPs: per slab row and path 2*(F+H)*4H*2 (two directions) + 2*2H*F (Linear); head 2*F*2F per row; per frame and speaker 2*(2*F*F + F*N)
flop = geo.rows * R * 2 * (2 * (F + H) * 4 * H * 2 + 2 * 2 * H * F) + geo.rows * 2 * F * 2 * F + B * 2 * geo.L * 2 * (2 * F * F + F * N) + B * geo.L * 2 * N * F
with torch.no_grad():
    for _ in range(2):
        m(x)
    torch.cuda.synchronize()
    n = 3
    t0 = time.perf_counter()
    for _ in range(n):
        m(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"DPRNN-TasNet N={N} F={F} H={H} B={B} x 3 s on the any-width kernels: {dt * 1e3:.1f} ms/step = {B / dt:.1f} utt/s; "
          f"{flop / 1e9:.0f} GFLOP algorithmic -> {flop / dt / 1e12:.1f} TFLOP/s ({flop / dt / 157.3e12:.2f} of the fp32 MFMA peak); "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    names = set(pkg._lib.SIGNATURES)
    engine.PROFILE = {"names": names, "events": {}}
    m(x)
    torch.cuda.synchronize()
    prof, engine.PROFILE = engine.PROFILE, None
rows = sorted(((sum(e0.elapsed_time(e1) for e0, e1 in ev), len(ev), n) for n, ev in prof["events"].items()), reverse=True)
tot = sum(r[0] for r in rows)
print(f"profiled forward: {tot:.1f} ms inside C-ABI calls")
for ms, cnt, n in rows[:12]:
    print(f"  {n:34s} {cnt:4d} calls {ms:9.2f} ms {100 * ms / tot:5.1f} %")
