#!/usr/bin/env python3
"""How long does the host take to ENQUEUE one forward of the headline batch, against the time the GPU needs to run it?  Also
prints the cgroup's CPU-throttling counters around the measurement (a throttled host starves the GPU between launches)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_state

def throttle():
    for p in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        if os.path.exists(p):
            d = dict(l.split() for l in open(p).read().strip().splitlines())
            return {k: int(d[k]) for k in ("nr_periods", "nr_throttled", "throttled_usec", "throttled_time") if k in d}
    return {}

CTOR = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
        "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]
cfg = full_cfg()
model = pkg.DPRNNTasNet(**{k: cfg[k] for k in CTOR}).eval()
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state("bss", cfg, 0).items()})
model = model.to("cuda:0")
x = torch.from_numpy(make_mixture(64, 24000, seed=1)).to("cuda:0")
with torch.no_grad():
    for _ in range(3):
        model(x)
    torch.cuda.synchronize()
    for rep in range(4):
        t_before = throttle()
        enq = []
        t0 = time.perf_counter()
        for _ in range(10):
            t1 = time.perf_counter()
            model(x)
            enq.append(time.perf_counter() - t1)
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        t_after = throttle()
        print(f"rep {rep}: 10 forwards enqueued in {1e3 * t_enq:.1f} ms (per forward median {1e3 * sorted(enq)[5]:.2f}, max {1e3 * max(enq):.2f}), "
              f"finished after {1e3 * t_all:.1f} ms = {1e3 * t_all / 10:.1f} ms per forward; throttling delta "
              f"{ {k: t_after[k] - t_before.get(k, 0) for k in t_after} }", flush=True)
