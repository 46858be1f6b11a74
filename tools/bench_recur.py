#!/usr/bin/env python3
"""Micro-benchmark of dprnn_bilstm_fwd alone (MI355X): isolates per-step cost from grid quantisation.
usage: python tools/bench_recur.py [--fast 1]  -> table of (seq_tile, workgroups, T) -> us/step, TFLOP/s"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tss_with_dprnn_amd as pkg  # noqa: E402
from tss_with_dprnn_amd import engine  # noqa: E402


def run(st, seq_tile, nwg, T, ndir=2, iters=3):
    nseq = nwg * (abs(seq_tile) % 1000) // ndir       # 2000 + n: eight-wave split forms of n sequences
    X = torch.randn(nseq * T, 128, device="cuda") * 0.5
    H = torch.empty(nseq * T, ndir * 128, device="cuda")
    packed = torch.empty(ndir * engine.PACK_PER_DIR_MAX, device="cuda")
    bsum = torch.empty(ndir * 512, device="cuda")
    w = [torch.randn(512, 128, device="cuda") * 0.08, torch.randn(512, 128, device="cuda") * 0.08,
         torch.randn(512, device="cuda") * 0.05, torch.randn(512, device="cuda") * 0.05]
    st.call("dprnn_rnn_pack", 0, *[t.data_ptr() for t in w], *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr())
    args = (0, X.data_ptr(), 0, 0, 0, 0, 0.0, 1, 1, 0, packed.data_ptr(), bsum.data_ptr(), H.data_ptr(), nseq, T, nseq, 0, T, 1,
            ndir, seq_tile)
    st.call("dprnn_rnn_fwd", *args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        st.call("dprnn_rnn_fwd", *args)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = nseq * T * ndir * 2 * 256 * 512
    return ms, flops / ms / 1e9


def run_forward_shapes(st, B=32, S=194, K=250, iters=5, tile=2064):
    """The forward's own launches alone on the chip: one 32-utterance group's intra-chunk (B S sequences of K steps) and inter-chunk
    (B K sequences of S steps) recurrences, 64-sequence tiles, without and with the pending norm + residual folded into the x loads."""
    rows = B * S * K
    X = torch.randn(rows, 128, device="cuda") * 0.5
    X2 = torch.empty_like(X)
    Z = torch.randn(rows, 128, device="cuda")
    H = torch.empty(rows, 256, device="cuda")
    stats = torch.zeros(B, 2, device="cuda", dtype=torch.float64)
    stats[:, 1] = S * K * 128
    gamma, beta = torch.ones(128, device="cuda"), torch.zeros(128, device="cuda")
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device="cuda")
    bsum = torch.empty(2 * 512, device="cuda")
    w = [torch.randn(512, 128, device="cuda") * 0.08, torch.randn(512, 128, device="cuda") * 0.08,
         torch.randn(512, device="cuda") * 0.05, torch.randn(512, device="cuda") * 0.05]
    st.call("dprnn_rnn_pack", 0, *[t.data_ptr() for t in w], *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr())
    for name, addr, T, per in (("intra-chunk", (B * S, K, B * S, 0, K, 1), K, S), ("inter-chunk", (B * K, S, K, S * K, 1, K), S, K)):
        for fuse in (0, 1):
            if fuse:
                args = (0, X.data_ptr(), Z.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-8, per, S * K, X2.data_ptr(),
                        packed.data_ptr(), bsum.data_ptr(), H.data_ptr(), *addr, 2, tile)
            else:
                args = (0, X.data_ptr(), 0, 0, 0, 0, 0.0, 1, 1, 0, packed.data_ptr(), bsum.data_ptr(), H.data_ptr(), *addr, 2, tile)
            st.call("dprnn_rnn_fwd", *args)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                st.call("dprnn_rnn_fwd", *args)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            wgs = 2 * -(-addr[0] // 64)
            print(f"{name} of {B} utterances ({addr[0]} sequences x {T} steps, {wgs} workgroups), pending norm {'folded in' if fuse else 'none'}: "
                  f"{ms:.3f} ms = {ms / T * 1e3:.2f} us per step, {addr[0] * T * 2 * 2 * 256 * 512 / ms / 1e9:.1f} TFLOP/s", flush=True)


def power_probe(st, tile, B=32, S=194, K=250, launches=1500):
    """One launch shape (intra-chunk of 32 utterances) queued `launches` times; rocm-smi is read while the queue drains: time per
    launch, socket power, shader clock.  For the energy probes (DPRNN_LIB = a -DDPRNN_SPLIT8_DIAG build): the chip runs this kernel
    at its power cap, so time follows energy per step, not cycles per step (profiles/r03_power_bound.txt)."""
    import re
    import subprocess
    rows = B * S * K
    X = torch.randn(rows, 128, device="cuda") * 0.5
    H = torch.empty(rows, 256, device="cuda")
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device="cuda")
    bsum = torch.empty(2 * 512, device="cuda")
    w = [torch.randn(512, 128, device="cuda") * 0.08, torch.randn(512, 128, device="cuda") * 0.08,
         torch.randn(512, device="cuda") * 0.05, torch.randn(512, device="cuda") * 0.05]
    st.call("dprnn_rnn_pack", 0, *[t.data_ptr() for t in w], *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr())
    args = (0, X.data_ptr(), 0, 0, 0, 0, 0.0, 1, 1, 0, packed.data_ptr(), bsum.data_ptr(), H.data_ptr(), B * S, K, B * S, 0, K, 1, 2, tile)
    for _ in range(20):
        st.call("dprnn_rnn_fwd", *args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        st.call("dprnn_rnn_fwd", *args)
    e1.record()
    samples = []
    for _ in range(4):
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
        pw = re.search(r"Power \(W\): ([0-9.]+)", out)
        ck = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
        if pw and ck and not e1.query():
            samples.append((float(pw.group(1)), int(ck.group(1))))
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / launches
    print(f"power probe tile {tile} lib {os.environ.get('DPRNN_LIB', 'default')}: {ms:.3f} ms per launch = {ms / K * 1e3:.2f} us per step; "
          f"(socket W, sclk MHz) while the queue drained: {samples}", flush=True)


def run_train(st, nwg, T, ndir=2, iters=3):
    """dprnn_rnn_fwd_train (forward + gate tape) and dprnn_rnn_bwd (BPTT) on a grid of nwg 32-sequence workgroups."""
    lib = st.lib
    nseq = nwg * 32 // ndir
    rows = nseq * T
    X = torch.randn(rows, 128, device="cuda") * 0.5
    H = torch.empty(rows, ndir * 128, device="cuda")
    dH = torch.randn(rows, ndir * 128, device="cuda") * 0.1
    dX = torch.empty(ndir, rows, 128, device="cuda")
    dG = torch.empty(rows, ndir * 512, device="cuda")
    tape = torch.empty(lib.dprnn_rnn_tape_floats(0, nseq, T, ndir), device="cuda")
    packed = torch.empty(ndir * engine.PACK_PER_DIR_MAX, device="cuda")
    packedT = torch.empty(lib.dprnn_rnn_pack_bwd_floats(ndir), device="cuda")
    bsum = torch.empty(ndir * 512, device="cuda")
    w = [torch.randn(512, 128, device="cuda") * 0.08, torch.randn(512, 128, device="cuda") * 0.08,
         torch.randn(512, device="cuda") * 0.05, torch.randn(512, device="cuda") * 0.05]
    p = lambda t: t.data_ptr()
    st.call("dprnn_rnn_pack", 0, *[p(t) for t in w], *[p(t) for t in w], p(packed), p(bsum))
    st.call("dprnn_rnn_pack_bwd", 0, p(w[0]), p(w[1]), p(w[0]), p(w[1]), p(packedT))      # layout follows dprnn_set_bwd_gemm_mode
    addr = (nseq, T, nseq, 0, T, 1, ndir)
    mode, bmode = lib.dprnn_get_gemm_mode(), lib.dprnn_get_bwd_gemm_mode()
    flag = 0x100 if (mode == 3 and bmode == 3 and os.environ.get("DPRNN_TRAIN_TAPE_PACKED", "1") != "0") else 0      # the product's pairing
    calls = {"fwd_train": ("dprnn_rnn_fwd_train_ex", (mode | flag, 0, p(X), 0, 0, 0, 0, 0.0, 1, 1, 0, p(packed), p(bsum), p(H), p(tape), *addr)),
             "bwd": ("dprnn_rnn_bwd_ex", (bmode | flag, 0, 0, p(dH), p(tape), p(packedT), p(dX), p(dG), rows, *addr))}
    if mode == 3:       # the inference kernel on the same grid (64-sequence tiles, h as fp32): what the tape and the sequence-major cell cost
        calls["fwd (inference kernel)"] = ("dprnn_rnn_fwd", (0, p(X), 0, 0, 0, 0, 0.0, 1, 1, 0, p(packed), p(bsum), p(H), *addr, 2064))
    flops = nseq * T * ndir * 2 * 256 * 512
    out = {}
    for k, (fn, args) in calls.items():
        st.call(fn, *args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            st.call(fn, *args)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        out[k] = (ms, flops / ms / 1e9)
    return out


def bwd_stamps(st, nwg, T, ndir=2):
    """Phase breakdown of lstm_bwd_kernel from in-kernel cycle stamps of wave 0 (diagnostic twin, LSTM)."""
    lib = st.lib
    nseq = nwg * 32 // ndir
    rows = nseq * T
    X = torch.randn(rows, 128, device="cuda") * 0.5
    H = torch.empty(rows, ndir * 128, device="cuda")
    dH = torch.randn(rows, ndir * 128, device="cuda") * 0.1
    dX = torch.empty(ndir, rows, 128, device="cuda")
    dG = torch.empty(rows, ndir * 512, device="cuda")
    tape = torch.empty(lib.dprnn_rnn_tape_floats(0, nseq, T, ndir), device="cuda")
    packed = torch.empty(ndir * engine.PACK_PER_DIR_MAX, device="cuda")
    packedT = torch.empty(lib.dprnn_rnn_pack_bwd_floats(ndir), device="cuda")
    bsum = torch.empty(ndir * 512, device="cuda")
    w = [torch.randn(512, 128, device="cuda") * 0.08, torch.randn(512, 128, device="cuda") * 0.08,
         torch.randn(512, device="cuda") * 0.05, torch.randn(512, device="cuda") * 0.05]
    p = lambda t: t.data_ptr()
    st.call("dprnn_rnn_pack", 0, *[p(t) for t in w], *[p(t) for t in w], p(packed), p(bsum))
    st.call("dprnn_rnn_pack_bwd", 0, p(w[0]), p(w[1]), p(w[0]), p(w[1]), p(packedT))
    addr = (nseq, T, nseq, 0, T, 1, ndir)
    st.call("dprnn_rnn_fwd_train", 0, p(X), 0, 0, 0, 0, 0.0, 1, 1, 0, p(packed), p(bsum), p(H), p(tape), *addr)
    buf = torch.zeros(nwg, 8, device="cuda", dtype=torch.int64)
    fn = lib.dprnn_debug_bwd_stamps
    fn(buf.data_ptr())
    try:
        st.call("dprnn_rnn_bwd", 0, p(dH), p(tape), p(packedT), p(dX), p(dG), rows, *addr)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            st.call("dprnn_rnn_bwd", 0, p(dH), p(tape), p(packedT), p(dX), p(dG), rows, *addr)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"stamped twin: {ms:.3f} ms per launch = {nseq * T * ndir * 2 * 256 * 512 / ms / 1e9:.1f} TFLOP/s")
    finally:
        fn(None)
    b = buf.cpu().double()
    tot = b[:, 0].median()
    start = (b[:, 7] - b[:, 7].min()) / 100.0
    end = start + b[:, 6] / 100.0
    q = torch.tensor([0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0], dtype=torch.float64)
    print("   start-time percentiles (us):", [int(v) for v in torch.quantile(start, q).tolist()])
    print("   end-time percentiles (us)  :", [int(v) for v in torch.quantile(end, q).tolist()])
    names = ["gates+stores+prefetch issue", "barrier 1", "GEMM", "barrier 2", "dh hand-over + dX stores"]
    clk = float((b[:, 0] / b[:, 6]).median()) * 100.0
    print(f"lstm_bwd stamps (wave 0, median over {nwg} workgroups, T={T}): {tot / T:.0f} cycles per step, in-kernel clock {clk:.0f} MHz")
    for i, n in enumerate(names):
        v = b[:, 1 + i].median()
        print(f"   {n:30s} {v / T:8.0f} cycles/step  {100 * v / tot:5.1f} %")


def stamps(st, seq_tile, per_cu, T, ndir=2):
    ncu = st.lib.dprnn_num_cus()
    nwg = ncu * per_cu
    nseq = nwg * seq_tile // ndir
    X = torch.randn(nseq * T, 128, device="cuda") * 0.5
    H = torch.empty(nseq * T, ndir * 128, device="cuda")
    packed = torch.randn(ndir * engine.PACK_PER_DIR, device="cuda") * 0.08
    bsum = torch.randn(ndir * 512, device="cuda") * 0.1
    buf = torch.zeros(nwg, 8, device="cuda", dtype=torch.int64)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(12):      # long enough for the clock to settle; time the last 8
        if it == 4:
            e0.record()
        st.call("dprnn_bilstm_fwd_stamped", X.data_ptr(), packed.data_ptr(), bsum.data_ptr(), H.data_ptr(), nseq, T,
                nseq, 0, T, 1, ndir, seq_tile, buf.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 8
    tf = nseq * T * ndir * 2 * 256 * 512 / ms / 1e9
    b = buf.cpu().double()
    cyc, ticks, kloop, cell = b[:, 0].median(), b[:, 1].median(), b[:, 2].median(), b[:, 3].median()
    clk = cyc / ticks * 100.0
    start = (b[:, 4] - b[:, 4].min()) / 100.0          # us
    end = (b[:, 5] - b[:, 4].min()) / 100.0
    import collections
    late = int((start > 50).sum())
    q = torch.tensor([0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0], dtype=torch.float64)
    print("   end-time percentiles (us) [0,10,25,50,75,90,99,100]:", [int(v) for v in torch.quantile(end, q).tolist()])
    for x in range(8):
        sel = (b[:, 6].long() & 0xF) == x
        if sel.any():
            print(f"   xcc {x}: {int(sel.sum())} wgs, end median {end[sel].median():.0f} max {end[sel].max():.0f}, clock {float((b[sel, 0] / b[sel, 1]).median()) * 100:.0f} MHz")
    print(f"   dispatch: {nwg} wgs, start us min/median/max {start.min():.0f}/{start.median():.0f}/{start.max():.0f}, "
          f"end us min/median/max {end.min():.0f}/{end.median():.0f}/{end.max():.0f}, started later than 50 us: {late}")
    cu = collections.Counter((int(x) & 0xF, (int(h) >> 8) & 0xF, (int(h) >> 13) & 0x7, (int(h) >> 16) & 0xF) for x, h in zip(b[:, 6].tolist(), b[:, 7].tolist()))
    print(f"   distinct (xcc, cu, sh?, se) ids {len(cu)}, workgroups per id histogram {sorted(collections.Counter(cu.values()).items())}")
    print(f"stamps DIAG={os.environ.get('DPRNN_DIAG', '0')} seq_tile={seq_tile} wgs/CU={per_cu}: {tf:.1f} TFLOP/s, in-kernel clock {clk:.0f} MHz; per step: total {cyc / T:.0f} cyc, "
          f"k-loop {kloop / T:.0f}, cell+stores {cell / T:.0f}, barrier/other {(cyc - kloop - cell) / T:.0f}")


def stamps_split(st, seq_tile, T, ndir=2):
    """In-kernel stamps of the wide bf16-split kernel (gemm mode 1): one workgroup per CU, phase cycles of wave 0."""
    ncu = st.lib.dprnn_num_cus()
    nwg = ncu
    nseq = nwg * (seq_tile % 1000) // ndir
    X = torch.randn(nseq * T, 128, device="cuda") * 0.5
    H = torch.empty(nseq * T, ndir * 128, device="cuda")
    packed = torch.empty(ndir * engine.PACK_PER_DIR_MAX, device="cuda")
    bsum = torch.empty(ndir * 512, device="cuda")
    w = [torch.randn(512, 128, device="cuda") * 0.08, torch.randn(512, 128, device="cuda") * 0.08,
         torch.randn(512, device="cuda") * 0.05, torch.randn(512, device="cuda") * 0.05]
    st.call("dprnn_rnn_pack", 0, *[t.data_ptr() for t in w], *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr())
    buf = torch.zeros(nwg, 16, device="cuda", dtype=torch.int64)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(12):
        if it == 4:
            e0.record()
        st.call("dprnn_bilstm_fwd_stamped", X.data_ptr(), packed.data_ptr(), bsum.data_ptr(), H.data_ptr(), nseq, T,
                nseq, 0, T, 1, ndir, seq_tile, buf.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 8
    tf = nseq * T * ndir * 2 * 256 * 512 / ms / 1e9
    b = buf.cpu().double()
    for wv, name in ((0, "wave 0"), (8, "wave 4 (its SIMD partner)")):
        m = [b[:, wv + i].median().item() / T for i in range(7)]
        print(f"   {name}: per step total {m[0]:.0f} cyc: h half (or whole MFMA phase) {m[2]:.0f}, cell update {m[3]:.0f}, own x half {m[4]:.0f}, "
              f"wait + x staging {m[5]:.0f}, barrier wait {m[6]:.0f}")
    cyc, ticks, mf, cell = b[:, 0].median(), b[:, 1].median(), b[:, 2].median(), b[:, 3].median()
    tiles = (seq_tile % 1000 + 15) // 16
    products = 3 if st.lib.dprnn_get_gemm_mode() == 3 else 6
    floor = tiles * 8 * 4 * products * 16 * 2            # MFMAs of a wave x 16 cycles x two waves per SIMD
    print(f"split8 stamps seq_tile={seq_tile} (wave 0 of every workgroup, medians): {ms:.3f} ms = {tf:.1f} TFLOP/s (fp32-equivalent), in-kernel clock "
          f"{cyc / ticks * 100:.0f} MHz; per step: total {cyc / T:.0f} cyc = MFMA phase as seen by the wave {mf / T:.0f} + wait at barrier A, "
          f"cell update and x staging {cell / T:.0f} + barrier B / loop {(cyc - mf - cell) / T:.0f}  [MFMA floor per SIMD: {tiles} tiles x 8 slabs x "
          f"4 gates x {products} products x 16 cyc x 2 waves = {floor} cyc]")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fast", type=int, default=1)
    ap.add_argument("--T", type=int, default=200)
    ap.add_argument("--stamps", type=int, default=0)
    ap.add_argument("--gemm-mode", type=int, default=0, help="0 exact fp32, 1 bf16 split x6, 3 f16 pieces x3")
    ap.add_argument("--train", type=int, default=0, help="1: time the training recurrences (fwd with tape, BPTT) on a full grid")
    ap.add_argument("--forward-shapes", type=int, default=0, help="1: the forward's own launches of one 32-utterance group, alone on the chip")
    ap.add_argument("--power", type=int, default=0, help="1: queue one launch shape 1500 times and read rocm-smi meanwhile (time, W, sclk)")
    ap.add_argument("--tile", type=int, default=2064, help="seq_tile code of --forward-shapes / --stamps / --power")
    ap.add_argument("--only", type=str, default="", help="seq_tile,wgs_per_cu: run just this configuration (for rocprofv3 --pmc)")
    a = ap.parse_args()
    lib = pkg.load_library()
    lib.dprnn_set_fast_activations(a.fast)
    lib.dprnn_set_gemm_mode(a.gemm_mode)
    st = engine.Stage(torch.device("cuda:0"))
    ncu = lib.dprnn_num_cus()
    print(f"CUs {ncu} fast={a.fast} T={a.T}")
    if a.power:
        power_probe(st, a.tile)
        return
    if a.forward_shapes:
        run_forward_shapes(st, tile=a.tile)
        return
    if a.stamps and a.gemm_mode in (1, 3):
        for seq_tile in (a.tile,):
            stamps_split(st, seq_tile, a.T)
        return
    if a.stamps:
        for seq_tile, per_cu in ((32, 1), (32, 2), (64, 1)):
            stamps(st, seq_tile, per_cu, a.T)
        return
    if a.train == 2:
        bwd_stamps(st, ncu, a.T)            # one workgroup per CU: the kernel's own pace
        bwd_stamps(st, ncu * 2, a.T)        # two per CU: the production occupancy
        return
    if a.train == 3:
        # BPTT grid probe (round 5): the f16 BPTT owns 64 sequences and 128 KB of LDS per workgroup (one per CU); the same launch at 1, 1.5, 2,
        # 3 and 4 rounds of workgroups: what a round costs, and that a partial round costs a whole one.
        lib.dprnn_set_gemm_mode(3)
        lib.dprnn_set_bwd_gemm_mode(3)
        for per_cu in (1.0, 1.5, 2.0, 3.0, 4.0):
            nwg32 = int(ncu * 2 * per_cu)
            r = run_train(st, nwg32, a.T, iters=5)
            ms, tf = r["bwd"]
            msf, _ = r["fwd_train"]
            print(f"bwd f16, {per_cu:.1f} rounds of 64-sequence workgroups (one per CU): {ms:.3f} ms = {ms / per_cu:.3f} ms per full round, {tf:.1f} TFLOP/s | "
                  f"fwd_train (32-sequence tiles, {2 * per_cu:.0f} per CU): {msf:.3f} ms = {msf / per_cu:.3f} per round", flush=True)
        lib.dprnn_set_gemm_mode(0); lib.dprnn_set_bwd_gemm_mode(0)
        return
    if a.train:
        # the training recurrences on a full grid (two 32-sequence tiles per CU = one 64-sequence tile per CU), in both arithmetics:
        # forward with tape under --gemm-mode 0 | 3, BPTT under dprnn_set_bwd_gemm_mode 0 | 3 (bwd time includes the dH max pass)
        for bmode in (0, 3):
            lib.dprnn_set_gemm_mode(bmode)
            lib.dprnn_set_bwd_gemm_mode(bmode)
            for k, (ms, tf) in run_train(st, ncu * 2, a.T, iters=5).items():
                print(f"train {k} ({'f16 pieces' if bmode else 'fp32 MFMA'}): {ms:.3f} ms {tf:.1f} TFLOP/s = {tf / 157.3:.2f} of the fp32 MFMA peak")
        lib.dprnn_set_gemm_mode(0); lib.dprnn_set_bwd_gemm_mode(0)
        return
    if a.only:
        seq_tile, per_cu = (int(v) for v in a.only.split(","))
        ms, tf = run(st, seq_tile, ncu * per_cu, a.T, iters=5)
        print(f"only seq_tile={seq_tile} wgs/CU={per_cu}: {ms:.3f} ms {tf:.1f} TFLOP/s")
        return
    if a.gemm_mode != 0:       # bf16-split modes: the two-per-CU form, the round-1 64-sequence form and the wide one-per-CU forms
        np_ = 6 if a.gemm_mode == 1 else 3
        print(f"gemm mode {a.gemm_mode} ({np_} products): seq_tile wgs/CU ms us/step TFLOP/s(fp32-equivalent) bf16-MFMA-floor-us/step@2.4GHz")
        forms = ((32, 2), (2064, 1), (2080, 1), (2096, 1))
        if a.gemm_mode == 3:    # f16 pieces: the eight-wave kernels only
            forms = ((2016, 1), (2032, 1), (2032, 2), (2048, 1), (2048, 2), (2064, 1), (2080, 1), (2096, 1))
        for seq_tile, per_cu in forms:
            ms, tf = run(st, seq_tile, ncu * per_cu, a.T, iters=5)
            tiles = (abs(seq_tile) % 1000 + 15) // 16
            floor_us = per_cu * tiles * 32 * 8 * np_ * 16 / 4 / 2.4e3
            print(f"{seq_tile:8d} {per_cu:4d} {ms:8.3f} {ms * 1e3 / a.T:8.2f} {tf:8.1f} {floor_us:8.2f}")
        return
    print("seq_tile  wgs/CU   ms     us/step  TFLOP/s  mfma-cycles/step/SIMD(ideal)  implied-util@2.4GHz")
    for seq_tile in (16, 32, 48, 64):
        for per_cu in (1, 2, 3, 4):
            lds = 2 * seq_tile * 257 * 4
            if per_cu * lds > 160 * 1024 or (seq_tile > 32 and per_cu > 1):
                continue
            ms, tf = run(st, seq_tile, ncu * per_cu, a.T)
            us_step = ms * 1e3 / a.T
            ideal_cycles = per_cu * (seq_tile // 16) * 512 * 32
            util = ideal_cycles / 2.4e3 / us_step
            print(f"{seq_tile:8d} {per_cu:7d} {ms:7.3f} {us_step:8.2f} {tf:8.1f} {ideal_cycles:12d} {util:22.3f}")


if __name__ == "__main__":
    main()
