#!/usr/bin/env python3
"""Where does the memory of a captured forward sit?  Reserved / allocated bytes and the engine pool's size after each of the first
forwards of the bench's batch (eager, capture, replay ...)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import engine
from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_state

CTOR = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
        "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]
cfg = full_cfg()
model = pkg.DPRNNTasNet(**{k: cfg[k] for k in CTOR}).eval()
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state("bss", cfg, 0).items()})
model = model.to("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.from_numpy(make_mixture(B, 24000, seed=1)).to("cuda:0")
dev = torch.device("cuda:0")
def show(tag):
    torch.cuda.synchronize()
    ms = torch.cuda.memory_stats(dev)
    pool = engine._pool_reserved_bytes(engine._POOLS[str(dev)]) if str(dev) in engine._POOLS else 0
    print(f"{tag:34s} reserved {ms['reserved_bytes.all.current'] / 2**30:6.2f} GiB  allocated {ms['allocated_bytes.all.current'] / 2**30:6.2f}  "
          f"peak allocated {ms['allocated_bytes.all.peak'] / 2**30:6.2f}  engine pool {pool / 2**30:6.2f}  driver allocs / frees {ms['num_device_alloc']} / {ms['num_device_free']}", flush=True)
show("model + input")
with torch.no_grad():
    for i in range(4):
        o = model(x)
        show(f"after forward {i + 1} ({['eager', 'capture', 'replay', 'replay'][i]})")
    del o
    engine.empty_cache()
    show("after engine.empty_cache()")
    for mode in (0,):
        engine.GEMM_MODE = mode
        for i in range(3):
            o = model(x)
            show(f"mode {mode} forward {i + 1}")
print(torch.cuda.memory_summary(dev, abbreviated=True))
