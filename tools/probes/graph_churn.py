#!/usr/bin/env python3
"""Captures and replays forwards of many (batch size, mode) keys in one process: does graph eviction / re-capture hold up?
    python tools/probes/graph_churn.py CACHE_MAX SIZES [ROUNDS]      e.g.  4 16,24,32,48,64,96,128
    CHURN_LEGACY=1 ...   round 3's FIRST scheme, rebuilt here for the record: a private pool per capture, the oldest graph destroyed
                         AFTER the new one is captured (between its instantiation and its first launch), nothing returned to the driver
                         by hand -- the arm that segfaulted inside hipGraphLaunch at the sixteenth capture.
    CHURN_LEGACY=2 ...   the same private pools and the same absence of empty_cache() calls, but the oldest graph is destroyed BEFORE
                         the new capture begins.  (torch.cuda.graph.__enter__ calls empty_cache() itself right before capture_begin in
                         BOTH arms, so in both the destroyed graphs' segments go back to the driver outside any capture.)
The caching allocator's driver-level counters (segments allocated / freed, retries) are printed around every capture and -- "inside" --
between the first and the last launch of the captured forward: a free counted "inside" is a hipFree issued while the capture was in
progress."""
import collections, faulthandler, os, sys
faulthandler.enable()
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import engine
from tss_with_dprnn_amd._lib import SAT_SLOTS, load_library
from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_state

CTOR = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
        "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]
engine.GRAPH_CACHE_MAX = int(sys.argv[1])
sizes = [int(v) for v in sys.argv[2].split(",")]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 1
samples = int(os.environ.get("SAMPLES", "24000"))
legacy = os.environ.get("CHURN_LEGACY") in ("1", "2")
evict_first = os.environ.get("CHURN_LEGACY") == "2"
cfg = full_cfg()
model = pkg.DPRNNTasNet(**{k: cfg[k] for k in CTOR}).eval()
model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state("bss", cfg, 0).items()})
model = model.to("cuda:0")
dev = torch.device("cuda:0")


def counters():
    ms = torch.cuda.memory_stats(dev)
    return {k: int(ms.get(k, 0)) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries")} | \
        {"reserved_GiB": round(ms.get("reserved_bytes.all.current", 0) / 2**30, 1)}


class LegacyGraphed:                       # engine._Graphed as it was before the shared pool
    def __init__(self, fn, inputs):
        lib = load_library()
        self.inputs = [t.clone() for t in inputs]
        self.flags = torch.zeros(SAT_SLOTS, dtype=torch.int32, device=dev)
        self.graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(dev)
        lib.dprnn_saturation_begin(self.flags.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        engine._TLS.guard, engine._TLS.capturing = 1, True
        try:
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.inside0 = counters()
                self.outputs = fn(*self.inputs)
                self.inside1 = counters()
        finally:
            engine._TLS.guard, engine._TLS.capturing = 0, False
            lib.dprnn_saturation_begin(0, 0)

    def replay(self, inputs):
        for s_, t in zip(self.inputs, inputs):
            s_.copy_(t)
        self.graph.replay()
        return self.outputs.clone()


cache = collections.OrderedDict()
ncap = 0
for rd in range(rounds):
    for B in sizes:
        x = torch.from_numpy(make_mixture(B, samples, seed=1)).to(dev)
        for mode in (0, 1, 3):
            engine.GEMM_MODE, engine.SPLIT_MIN_BATCH = mode, 0
            with torch.no_grad():
                if legacy:
                    fn = lambda w: engine._tasnet_forward_impl(model, w, None)
                    outs = [fn(x)]
                    while evict_first and len(cache) >= engine.GRAPH_CACHE_MAX:     # arm 2: the oldest graph dies BEFORE the capture
                        cache.popitem(last=False)
                    c0 = counters()
                    g = LegacyGraphed(fn, (x,))
                    c1 = counters()
                    ncap += 1
                    cache[(B, mode)] = g
                    while len(cache) > engine.GRAPH_CACHE_MAX:          # arm 1: AFTER the capture, before the new graph's first launch
                        cache.popitem(last=False)
                    print(f"  capture {ncap}: before {c0} inside {g.inside0} .. {g.inside1} after {c1}; first replay ...", flush=True)
                    outs += [g.replay((x,)) for _ in range(3)]
                else:
                    c0 = counters()
                    outs = [model(x) for _ in range(4)]
                    c1 = counters()
                    print(f"  eager + capture + 2 replays: before {c0} after {c1}", flush=True)
            torch.cuda.synchronize()
            load_library().dprnn_set_gemm_mode(0)
            assert all(torch.equal(outs[1], o) for o in outs[2:]), (B, mode)
            print(f"round {rd} B={B} mode {mode}: ok, graphs {engine.GRAPH_STATS if not legacy else ncap}, cached {len(cache) if legacy else len(engine._GRAPH_CACHE)}, "
                  f"reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB", flush=True)
print("done")
