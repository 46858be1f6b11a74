#!/usr/bin/env python3
"""One DPRNN path alone on the chip in both launch structures (MI355X, gate-GEMM mode 3, 64-sequence tiles):
  concurrent: both directions in one launch (dprnn_rnn_fwd_hp_ex, pending norm folded in) + the f16-pieces projection
  serial:     direction 0 with its half of the Linear (phase 1), then direction 1 (phase 2)   [csrc/recur_split8_f16_lin.hip]
usage: python tools/bench_serial.py [--batch 32,64] [--iters 5] [--tile 2064]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tss_with_dprnn_amd as pkg  # noqa: E402,F401
from tss_with_dprnn_amd import engine  # noqa: E402


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", default="32,64")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--tile", type=int, default=2064)
    ap.add_argument("--S", type=int, default=194)
    ap.add_argument("--K", type=int, default=250)
    a = ap.parse_args()
    st = engine.Stage(torch.device("cuda:0"))
    lib = st.lib
    S, K = a.S, a.K
    w = [torch.randn(512, 128, device="cuda") * 0.08, torch.randn(512, 128, device="cuda") * 0.08,
         torch.randn(512, device="cuda") * 0.05, torch.randn(512, device="cuda") * 0.05]
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device="cuda")
    bsum = torch.empty(2 * 512, device="cuda")
    st.call("dprnn_rnn_pack_ex", 3, 0, *[t.data_ptr() for t in w], *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr())
    W = torch.randn(128, 256, device="cuda") / 16
    b = torch.randn(128, device="cuda")
    linp = torch.empty(lib.dprnn_lin_pack_floats(), device="cuda")
    st.call("dprnn_lin_pack_f16", W.data_ptr(), 2, linp.data_ptr())
    gamma, beta = torch.ones(128, device="cuda"), torch.zeros(128, device="cuda")
    for B in [int(v) for v in a.batch.split(",")]:
        rows = B * S * K
        X = torch.randn(rows, 128, device="cuda") * 0.5
        X2 = torch.empty_like(X)
        Z = torch.randn(rows, 128, device="cuda")
        H = torch.empty(rows, 256, device="cuda")
        stats_in = torch.zeros(B, 2, device="cuda", dtype=torch.float64)
        stats_in[:, 1] = S * K * 128
        stats = torch.zeros(B, 2, device="cuda", dtype=torch.float64)
        for name, addr, T, per in (("intra", (B * S, K, B * S, 0, K, 1), K, S), ("inter", (B * K, S, K, S * K, 1, K), S, K)):
            norm = (Z.data_ptr(), stats_in.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-8)
            wgs = -(-addr[0] // (a.tile % 1000))

            def rec_conc():
                st.call("dprnn_rnn_fwd_hp_ex", 3, 0, X.data_ptr(), *norm, per, S * K, X2.data_ptr(), packed.data_ptr(), bsum.data_ptr(),
                        H.data_ptr(), *addr, 2, a.tile)

            def proj():
                lib.dprnn_set_gemm_mode(3)
                st.call("dprnn_proj_stats_hp_fwd", H.data_ptr(), W.data_ptr(), b.data_ptr(), Z.data_ptr(), stats.data_ptr(), rows, S * K, 0, 0, 256)
                lib.dprnn_set_gemm_mode(0)

            def ph1():
                st.call("dprnn_rnn_lin_fwd", 1, 0, X.data_ptr(), *norm, per, S * K, X2.data_ptr(), packed.data_ptr(), bsum.data_ptr(), 2,
                        linp.data_ptr(), 0, Z.data_ptr(), 0, 0, *addr, a.tile)

            def ph2():
                st.call("dprnn_rnn_lin_fwd", 2, 1, X2.data_ptr(), 0, 0, 0, 0, 0.0, per, S * K, 0, packed.data_ptr(), bsum.data_ptr(), 2,
                        linp.data_ptr(), b.data_ptr(), Z.data_ptr(), Z.data_ptr(), stats.data_ptr(), *addr, a.tile)

            state = torch.empty(lib.dprnn_rnn_lin_state_floats(addr[0], a.tile), device="cuda")
            Z2 = torch.empty_like(Z)

            def sp1():
                st.call("dprnn_rnn_lin_split_fwd", 1, X.data_ptr(), *norm, per, S * K, X2.data_ptr(), packed.data_ptr(), bsum.data_ptr(), linp.data_ptr(),
                        0, Z2.data_ptr(), 0, state.data_ptr(), *addr, a.tile)

            def sp2():
                st.call("dprnn_rnn_lin_split_fwd", 2, X2.data_ptr(), 0, 0, 0, 0, 0.0, per, S * K, 0, packed.data_ptr(), bsum.data_ptr(), linp.data_ptr(),
                        b.data_ptr(), Z2.data_ptr(), stats.data_ptr(), state.data_ptr(), *addr, a.tile)

            ts = {k: timed(f, a.iters) for k, f in (("split 1", sp1), ("split 2", sp2))}
            print(f"B={B} {name}: time-split fold (both directions per launch, {2 * wgs} workgroups): phase 1 {ts['split 1']:.3f} + phase 2 {ts['split 2']:.3f} = "
                  f"{ts['split 1'] + ts['split 2']:.3f} ms; us per step {ts['split 1'] / (T / 2) * 1e3:.2f} / {ts['split 2'] / (T / 2) * 1e3:.2f}", flush=True)
            t = {k: timed(f, a.iters) for k, f in (("concurrent recurrence", rec_conc), ("projection", proj), ("phase 1", ph1), ("phase 2", ph2))}
            conc, ser = t["concurrent recurrence"] + t["projection"], t["phase 1"] + t["phase 2"]
            print(f"B={B} {name}: {addr[0]} sequences x {T} steps, {wgs} workgroups per direction | concurrent {t['concurrent recurrence']:.3f} + "
                  f"projection {t['projection']:.3f} = {conc:.3f} ms | serial {t['phase 1']:.3f} + {t['phase 2']:.3f} = {ser:.3f} ms "
                  f"({ser / conc:.3f} x) | us per step: concurrent {t['concurrent recurrence'] / T * 1e3:.2f}, phase 1 {t['phase 1'] / T * 1e3:.2f}, "
                  f"phase 2 {t['phase 2'] / T * 1e3:.2f}", flush=True)
            Z.normal_()


if __name__ == "__main__":
    main()
