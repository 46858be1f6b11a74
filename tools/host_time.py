import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import inference
from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_state
BASE = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
        "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]
cfg = full_cfg()
m = pkg.DPRNNTasNet(**{k: cfg[k] for k in BASE}).eval()
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state("bss", cfg, 0).items()})
m = m.to("cuda:0")
wavs = [torch.from_numpy(make_mixture(1, 24000 + 8 * i, seed=i)[0]).to("cuda:0") for i in range(32)]
with torch.no_grad():
    inference.separate_many(m, wavs, max_streams=16); torch.cuda.synchronize()
    t0 = time.perf_counter()
    inference.separate_many(m, wavs, max_streams=16)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"32 utterances (BSS, ~3 s each), 16 streams: host enqueue {1e3 * (t1 - t0):.0f} ms, total {1e3 * (t2 - t0):.0f} ms "
          f"-> {32 / (t2 - t0):.1f} utt/s; host per forward {1e3 * (t1 - t0) / 32:.2f} ms")
    x = wavs[0][None]
    m(x); torch.cuda.synchronize()
    t0 = time.perf_counter(); m(x); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"single B=1 forward: host {1e3 * (t1 - t0):.2f} ms, total {1e3 * (t2 - t0):.2f} ms")
