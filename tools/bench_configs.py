#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations on one MI355X (not the bench.py headline):
  config 1: DPRNN-TasNet B=1 (example.ipynb path)         -> latency
  config 3: DPRNN-Spe (cat) B=64, 3 s mixtures + 3 s reference
  config 5: DPRNN-Spe-IRA, full-length utterances drawn from the Libri2Mix-test length distribution
            (min 24 000, median 36 720, mean 40 196, max 111 920 samples: SURVEY.md 8d), B=1 semantics, run
            through the multi-stream harness (inference.separate_many); this rank's share of an 8-GPU split."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("HWQ", "16"))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import inference, sharding
from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_state

BASE = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
        "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]
SPE = BASE + ["O", "P", "embeddings_size", "num_spks", "fusion_type"]
dev = "cuda:0"


def build(kind):
    cfg = full_cfg(norm_type="ln")
    cls = {"bss": pkg.DPRNNTasNet, "spe": pkg.DPRNNSpeTasNet, "ira": pkg.DPRNNSpeIRATasNet}[kind]
    m = cls(**{k: cfg[k] for k in (BASE if kind == "bss" else SPE)}).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state(kind, cfg, 0).items()})
    return m.to(dev)


def timed(fn, n=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


with torch.no_grad():
    m = build("bss")
    x1 = torch.from_numpy(make_mixture(1, 24000, seed=1)).to(dev)
    dt = timed(lambda: m(x1), n=10, warm=3)
    print(f"config 1  DPRNN-TasNet B=1 3 s: {dt * 1e3:.1f} ms/forward = {1 / dt:.1f} utt/s")
    del m
    m = build("spe")
    x = torch.from_numpy(make_mixture(64, 24000, seed=2)).to(dev)
    a = torch.from_numpy(make_mixture(64, 24000, seed=3)).to(dev)
    dt = timed(lambda: m(x, a, torch.tensor(24000.0)))
    print(f"config 3  DPRNN-Spe (cat) B=64 3 s: {dt * 1e3:.1f} ms/step = {64 / dt:.1f} utt/s")
    del m
    m = build("ira")
    rng = np.random.default_rng(0)
    # log-normal fit of the test-set lengths (samples @ 8 kHz), clipped to the observed range, 3000 utterances;
    # this process plays rank 0 of 8: its length-balanced share
    lens = np.clip(np.exp(rng.normal(np.log(36720), 0.38, 3000)), 24000, 111920).astype(int)
    mine = sharding.shard_by_length(lens.tolist(), 8)[0][:48]          # bounded sample of the share
    wavs = [torch.from_numpy(make_mixture(1, int(lens[i]), seed=100 + i)[0]).to(dev) for i in mine]
    aux = [(torch.from_numpy(make_mixture(1, 24000, seed=900 + i)[0]).to(dev), 24000.0) for i in mine]
    secs = float(sum(lens[i] for i in mine)) / 8000.0
    for ns in (16, 48):
        dt = timed(lambda: inference.separate_many(m, wavs, max_streams=ns, extra_inputs=aux), n=2, warm=1)
        print(f"config 5  DPRNN-Spe-IRA full-length, {len(mine)} utterances ({secs:.0f} s of audio, mean {secs / len(mine):.2f} s) "
              f"through {ns} streams: {dt:.2f} s = {len(mine) / dt:.1f} utt/s per GPU, {secs / dt:.0f}x real time")
    one = timed(lambda: [m(w[None], a_[0][None], torch.tensor(a_[1])) for w, a_ in zip(wavs[:8], aux[:8])], n=1, warm=1)
    print(f"          same model, first 8 utterances one at a time on one stream (the reference's loop shape): "
          f"{8 / one:.1f} utt/s")
