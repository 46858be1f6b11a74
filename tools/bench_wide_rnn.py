#!/usr/bin/env python3
"""dprnn_wide_rnn_fwd alone at the headline geometry (B x 3 s, K 250 / P 125): ms and TFLOP/s per launch, intra- and inter-chunk.
usage: tools/bench_wide_rnn.py [F] [H] [B]      (DPRNN_WIDE_NT=1 forces 16-sequence workgroups)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tss_with_dprnn_amd import engine, wide
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
H = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0")
geo = engine.make_geometry(B, 24000, 64, 250, 125, 2, 1)
st = wide._Launcher(dev)
torch.manual_seed(0)
rnn = torch.nn.LSTM(F, H, batch_first=True, bidirectional=True).to(dev)
Wp, bv, ndir = wide._pack_rnn(st, rnn, 0, F, H, dev)
X = torch.randn(geo.rows, F, device=dev)
out = torch.empty(geo.rows, 2 * H, device=dev)
flop = geo.rows * 2 * (2 * (F + H) * 4 * H)
for name, args in (("intra", (B * geo.S, geo.K, 1, geo.K, 0, 1)), ("inter", (B * geo.K, geo.S, geo.K, geo.S * geo.K, 1, geo.K))):
    for _ in range(2):
        st.call("dprnn_wide_rnn_fwd", 0, X.data_ptr(), Wp.data_ptr(), bv.data_ptr(), out.data_ptr(), 2, F, H, *args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        st.call("dprnn_wide_rnn_fwd", 0, X.data_ptr(), Wp.data_ptr(), bv.data_ptr(), out.data_ptr(), 2, F, H, *args)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{name}: F={F} H={H} B={B} nseq={args[0]} T={args[1]}: {dt * 1e3:.2f} ms  {flop / dt / 1e12:.1f} TFLOP/s ({flop / dt / 157.3e12:.3f} of fp32 MFMA peak)  NT env={os.environ.get('DPRNN_WIDE_NT', '-')}")
if wide.f16_recurrence(0, F, H):
    Hbuf = out.view(-1)
    for name, args in (("intra", (B * geo.S, geo.K, 1, geo.K, 0, 1)), ("inter", (B * geo.K, geo.S, geo.K, geo.S * geo.K, 1, geo.K))):
        for _ in range(2):
            wide._recurrence(st, rnn, 0, X, geo.rows, Hbuf, F, H, args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            wide._recurrence(st, rnn, 0, X, geo.rows, Hbuf, F, H, args)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{name}, f16 pieces (absmax + pack + dprnn_wide_lstm_f16_fwd): {dt * 1e3:.2f} ms  {flop / dt / 1e12:.1f} TFLOP/s algorithmic "
              f"({flop / dt / 833.3e12:.3f} of the f16-pieces ceiling 2500 / 3)")
