#!/usr/bin/env python3
"""Energy per recurrence step by component (MI355X): one launch shape of the f16-pieces recurrence kernel that fills every CU is
queued back to back for several seconds; the socket's ENERGY COUNTER (not the running-average power figure) is read before and
after, so joules per launch = delta / launches.  Run once per build of the library (DPRNN_LIB = a -DDPRNN_SPLIT8_DIAG=<mask>
build from tools/build_diag_libs.sh; results of those builds are wrong by construction, their time and energy are valid):

    python tools/energy_budget.py --label full            # the product kernel
    DPRNN_LIB=tools/probes/diag/libdprnn_diag4.so python tools/energy_budget.py --label no_mfma
    python tools/energy_budget.py --summarise gpurun_out/energy/*.json      # the budget table (component = full - probe)

Energy sources, first that works: hwmon energy1_input (uJ), `rocm-smi --showenergycounter` (accumulated uJ)."""
import argparse
import glob
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hwmon_energy_file():
    for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/energy1_input")):
        try:
            int(open(f).read())
            return f
        except (OSError, ValueError):
            continue
    return None


def read_energy_uj(src):
    """(microjoules, source name)"""
    if src and src.startswith("/sys"):
        return float(open(src).read()), "hwmon"
    out = subprocess.run(["rocm-smi", "--showenergycounter"], capture_output=True, text=True).stdout
    m = re.search(r"Accumulated Energy \(uJ\):\s*([0-9.]+)", out)
    if not m:
        raise RuntimeError("no energy counter in rocm-smi output:\n" + out)
    return float(m.group(1)), "rocm-smi"


def read_power_clock():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    pw = re.search(r"Power \(W\): ([0-9.]+)", out)
    ck = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
    return (float(pw.group(1)) if pw else None, int(ck.group(1)) if ck else None)


def measure(label, seconds, nwg, T, tile, fuse):
    import torch
    sys.path.insert(0, ROOT)
    import tss_with_dprnn_amd as pkg  # noqa: F401
    from tss_with_dprnn_amd import engine
    st = engine.Stage(torch.device("cuda:0"))
    ndir, nsr = 2, tile % 1000
    nseq = nwg * nsr // ndir
    rows = nseq * T
    g = torch.Generator(device="cuda").manual_seed(1)
    X = torch.randn(rows, 128, device="cuda", generator=g) * 0.5
    Z = torch.randn(rows, 128, device="cuda", generator=g) if fuse else None
    X2 = torch.empty_like(X) if fuse else None
    H = torch.empty(rows, ndir * 128, device="cuda")
    stats = torch.zeros(1, 2, device="cuda", dtype=torch.float64)
    stats[:, 1] = rows * 128
    gamma, beta = torch.ones(128, device="cuda"), torch.zeros(128, device="cuda")
    packed = torch.empty(ndir * engine.PACK_PER_DIR_MAX, device="cuda")
    bsum = torch.empty(ndir * 512, device="cuda")
    w = [torch.randn(512, 128, device="cuda", generator=g) * 0.08, torch.randn(512, 128, device="cuda", generator=g) * 0.08,
         torch.randn(512, device="cuda", generator=g) * 0.05, torch.randn(512, device="cuda", generator=g) * 0.05]
    p = lambda t: 0 if t is None else t.data_ptr()
    st.call("dprnn_rnn_pack_ex", 3, 0, *[p(t) for t in w], *[p(t) for t in w], p(packed), p(bsum))
    if fuse:      # pending norm of one "utterance" spanning the slab, x' to the other slab: the forward's own launch form
        args = (3, 0, p(X), p(Z), p(stats), p(gamma), p(beta), 1e-8, nseq, rows, p(X2), p(packed), p(bsum), p(H), nseq, T, nseq, 0, T, 1, ndir, tile)
    else:
        args = (3, 0, p(X), 0, 0, 0, 0, 0.0, 1, 1, 0, p(packed), p(bsum), p(H), nseq, T, nseq, 0, T, 1, ndir, tile)
    fn = "dprnn_rnn_fwd_hp_ex"       # h leaves as the projection's piece quads, as in the forward
    for _ in range(10):
        st.call(fn, *args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        st.call(fn, *args)
    e1.record()
    torch.cuda.synchronize()
    ms_est = e0.elapsed_time(e1) / 20
    n = max(50, int(seconds * 1e3 / ms_est))
    # warm the chip to its steady state (clock and power settle over ~1 s), then the measured window
    for _ in range(n // 3):
        st.call(fn, *args)
    torch.cuda.synchronize()
    src = _hwmon_energy_file()
    uj0, how = read_energy_uj(src)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        st.call(fn, *args)
    e1.record()
    mid = read_power_clock()           # while the queue drains
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    uj1, _ = read_energy_uj(src)
    gpu_ms = e0.elapsed_time(e1)
    # idle reference for the static share: the same read-out over one idle second
    time.sleep(0.5)
    i0, _ = read_energy_uj(src)
    ti0 = time.perf_counter()
    time.sleep(1.0)
    i1, _ = read_energy_uj(src)
    ti1 = time.perf_counter()
    res = {
        "label": label, "lib": os.environ.get("DPRNN_LIB", "default"), "energy_source": how,
        "shape": {"workgroups": nwg, "sequences_per_workgroup": nsr, "steps": T, "pending_norm_folded": bool(fuse), "tile": tile},
        "launches": n, "ms_per_launch": gpu_ms / n, "us_per_step": gpu_ms / n / T * 1e3,
        "wall_s": t1 - t0, "gpu_s": gpu_ms / 1e3,
        "joules": (uj1 - uj0) / 1e6, "joules_per_launch": (uj1 - uj0) / 1e6 / n,
        "millijoules_per_step": (uj1 - uj0) / 1e3 / n / T,
        "mean_watts": (uj1 - uj0) / 1e6 / (t1 - t0),
        "mid_run_watts_sclk": mid, "idle_watts": (i1 - i0) / 1e6 / (ti1 - ti0),
    }
    print(json.dumps(res), flush=True)
    return res


def summarise(paths):
    rs = {}
    for pth in paths:
        for line in open(pth):
            line = line.strip()
            if line.startswith("{"):
                r = json.loads(line)
                rs[r["label"]] = r
    full = rs.get("full")
    if not full:
        print("no 'full' record")
        return
    print(f"# energy source: {full['energy_source']}; shape {full['shape']}; idle {full['idle_watts']:.0f} W")
    print(f"{'build':28s} {'ms/launch':>9s} {'us/step':>8s} {'mean W':>7s} {'mJ/step':>8s} {'mid-run (W, MHz)':>18s}   component = full - this build: mJ/step (share of full), us/step (share)")
    comp_e = comp_t = 0.0
    for k, r in rs.items():
        de = full["millijoules_per_step"] - r["millijoules_per_step"]
        dt = full["us_per_step"] - r["us_per_step"]
        extra = ""
        if k not in ("full", "all_removed"):
            extra = f"   {de:7.3f} ({100 * de / full['millijoules_per_step']:5.1f} %)  {dt:6.2f} ({100 * dt / full['us_per_step']:5.1f} %)"
            if not k.startswith("x_"):
                comp_e += de
                comp_t += dt
        print(f"{k:28s} {r['ms_per_launch']:9.3f} {r['us_per_step']:8.2f} {r['mean_watts']:7.0f} {r['millijoules_per_step']:8.3f} {str(tuple(r['mid_run_watts_sclk'])):>18s}{extra}")
    if "all_removed" in rs:
        rest = rs["all_removed"]
        tot_e = comp_e + rest["millijoules_per_step"]
        tot_t = comp_t + rest["us_per_step"]
        print(f"# closure: sum of components {comp_e:.3f} mJ + everything-removed skeleton {rest['millijoules_per_step']:.3f} mJ = {tot_e:.3f} mJ "
              f"against the full kernel's {full['millijoules_per_step']:.3f} mJ ({100 * tot_e / full['millijoules_per_step']:.1f} %); "
              f"time {comp_t:.2f} + {rest['us_per_step']:.2f} = {tot_t:.2f} us against {full['us_per_step']:.2f} ({100 * tot_t / full['us_per_step']:.1f} %)")


def cumulative(paths):
    """Telescoping budget: the builds remove one more component each, in file order, so consecutive differences are the
    components and they add up to the full kernel by construction; the marginal view (each component removed alone from the full
    kernel: summarise) does not, because a shorter step runs at a higher clock and a different power."""
    rs = []
    for pth in paths:
        for line in open(pth):
            if line.strip().startswith("{"):
                rs.append(json.loads(line))
    full = rs[0]
    print(f"# energy source: {full['energy_source']}; shape {full['shape']}; idle {full['idle_watts']:.0f} W; full kernel "
          f"{full['millijoules_per_step']:.3f} mJ and {full['us_per_step']:.2f} us per step at {full['mean_watts']:.0f} W")
    print(f"{'build (cumulative removal)':32s} {'us/step':>8s} {'mean W':>7s} {'mJ/step':>8s}   component removed at this stage: mJ/step (share of full), us/step (share)")
    prev = None
    for r in rs:
        extra = ""
        if prev is not None:
            de, dt = prev["millijoules_per_step"] - r["millijoules_per_step"], prev["us_per_step"] - r["us_per_step"]
            extra = f"   {de:7.3f} ({100 * de / full['millijoules_per_step']:5.1f} %)  {dt:6.2f} ({100 * dt / full['us_per_step']:5.1f} %)"
        print(f"{r['label']:32s} {r['us_per_step']:8.2f} {r['mean_watts']:7.0f} {r['millijoules_per_step']:8.3f}{extra}")
        prev = r
    last = rs[-1]
    print(f"{'what is left (loop, barriers)':32s} {'':8s} {'':7s} {'':8s}   {last['millijoules_per_step']:7.3f} ({100 * last['millijoules_per_step'] / full['millijoules_per_step']:5.1f} %)  "
          f"{last['us_per_step']:6.2f} ({100 * last['us_per_step'] / full['us_per_step']:5.1f} %)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--label", default="full")
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--workgroups", type=int, default=256)
    ap.add_argument("--steps", type=int, default=250)
    ap.add_argument("--tile", type=int, default=2064)
    ap.add_argument("--fuse", type=int, default=1)
    ap.add_argument("--summarise", nargs="*")
    ap.add_argument("--cumulative", nargs="*")
    a = ap.parse_args()
    if a.cumulative is not None:
        cumulative(a.cumulative)
    elif a.summarise is not None:
        summarise(a.summarise)
    else:
        measure(a.label, a.seconds, a.workgroups, a.steps, a.tile, a.fuse)
