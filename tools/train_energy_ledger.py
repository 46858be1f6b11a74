#!/usr/bin/env python3
"""Joules, milliseconds and watts of the training step's launches, each queued alone for a few seconds in the shapes of one 32-utterance
micro-batch of the shipped model (194 chunks of 250): forward recurrence with the gate tape, BPTT, LSTM weight gradient, GroupNorm backward
sums.  Answers one question: which of them run AT the board's power limit (then time = joules, and only fewer bytes / FLOPs help) and which
below it (then a pipe or a dependency chain paces them, and occupancy / overlap can help).
usage: python tools/train_energy_ledger.py [--seconds 3]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from forward_energy_ledger import measure_calls


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    import tss_with_dprnn_amd as pkg  # noqa: F401
    from tss_with_dprnn_amd import engine
    dev = torch.device("cuda:0")
    st = engine.Stage(dev)
    lib = st.lib
    g = torch.Generator(device="cuda").manual_seed(1)
    B, S, K = a.batch, 194, 250
    rows = B * S * K
    p = lambda t: t.data_ptr()
    X = torch.randn(rows, 128, device="cuda", generator=g) * 0.5
    H = torch.empty(rows, 256, device="cuda")
    dH = torch.randn(rows, 256, device="cuda", generator=g) * 0.1
    dX = torch.empty(2, rows, 128, device="cuda")
    w = [torch.randn(512, 128, device="cuda", generator=g) * 0.08, torch.randn(512, 128, device="cuda", generator=g) * 0.08,
         torch.randn(512, device="cuda", generator=g) * 0.05, torch.randn(512, device="cuda", generator=g) * 0.05]
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device="cuda")
    bsum = torch.empty(2 * 512, device="cuda")
    packedT = torch.empty(lib.dprnn_rnn_pack_bwd_floats(2), device="cuda")
    st.call("dprnn_rnn_pack_ex", 3, 0, *[p(t) for t in w], *[p(t) for t in w], p(packed), p(bsum))
    st.call("dprnn_rnn_pack_bwd_ex", 3, 0, p(w[0]), p(w[1]), p(w[0]), p(w[1]), p(packedT))
    hmax = torch.full((1,), 0.5, device="cuda")
    res = {}
    for name, addr, T in (("intra", (B * S, K, B * S, 0, K, 1), K), ("inter", (B * K, S, K, S * K, 1, K), S)):
        nseq = addr[0]
        tape = torch.empty(lib.dprnn_rnn_tape_floats(0, nseq, T, 2), device="cuda")
        dG = torch.empty(lib.dprnn_rnn_bwd_dg_floats(0, 3, nseq, T, 2), device="cuda")
        fwd = lambda: st.call("dprnn_rnn_fwd_train_ex", 3 | 0x100, 0, p(X), 0, 0, 0, 0, 0.0, 1, 1, 0, p(packed), p(bsum), p(H), p(tape), *addr, 2)
        fwd()
        res["fwd_" + name] = measure_calls(f"forward recurrence with the packed gate tape, {name}-chunk", fwd, a.seconds)
        bwd = lambda: st.call("dprnn_rnn_bwd_ex", 3 | 0x100, p(hmax), 0, p(dH), p(tape), p(packedT), p(dX), p(dG), rows, *addr, 2)
        res["bptt_" + name] = measure_calls(f"BPTT (f16 pieces, 64-sequence workgroups), {name}-chunk", bwd, a.seconds)
        gcat = torch.empty(512, 256, device="cuda")
        gbias = torch.empty(512, device="cuda")
        ws = torch.empty(lib.dprnn_wgrad_workspace_floats(512, 256, T * (-(-nseq // 32) * 32)), device="cuda")
        nsq = -(-nseq // 4)
        wg = lambda: st.call("dprnn_wgrad_lstm_f16", packedT.data_ptr() + 4 * lib.dprnn_rnn_bwd_gradmax_offset(2), p(dG), p(X), p(H), 256, 0,
                             addr[0], addr[1], addr[2], addr[3], addr[4], addr[5], p(gcat), p(gbias), 0, p(ws))
        res["wgrad_" + name] = measure_calls(f"LSTM weight gradient of one direction (256 x 256 blocks), {name}-chunk", wg, a.seconds)
        del tape, dG, ws
    Z = torch.randn(rows, 128, device="cuda", generator=g)
    stats = torch.zeros(B, 2, device="cuda", dtype=torch.float64)
    stats[:, 1] = S * K * 128
    gamma = torch.ones(128, device="cuda")
    red = torch.zeros(2 * B + 256, device="cuda", dtype=torch.float64)
    dXs = torch.randn(rows, 128, device="cuda", generator=g)
    gn = lambda: st.call("dprnn_gn_bwd_stats", p(dXs), p(dX[0]), p(dX[1]), p(dXs), p(Z), p(stats), p(gamma), 1e-5, B, S * K, p(red), red.data_ptr() + 8 * 2 * B)
    res["gn_stats"] = measure_calls("GroupNorm backward sums + dX of three slabs (5 slab units)", gn, a.seconds)
    print(f"\n{'launch':70s} {'ms':>7s} {'W':>6s} {'J':>6s}")
    for k, r in res.items():
        print(f"{r['label']:70s} {r['ms_per_call']:7.3f} {r['mean_watts']:6.0f} {r['joules_per_call']:6.2f}   mid-run (W, sclk MHz) {r['mid_run_watts_sclk']}")


if __name__ == "__main__":
    main()
