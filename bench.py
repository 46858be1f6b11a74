#!/usr/bin/env python3
"""Headline benchmark: DPRNN-TasNet forward throughput (utterances/sec, 3 s @ 8 kHz, batch 64 per GPU).

    python bench.py --gpus N --steps K --warmup W            (any N: with N > 1 and no WORLD_SIZE in the environment the
                                                              script starts its own N ranks, see _self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the driver's form)
    python bench.py --mode train [--gpus N]                   (BASELINE.json configs[3]: training step, 64 per GPU)

A "step" is one forward of the HIP path over one batch of 64 synthetic 3-second mixtures already
resident in HBM (BASELINE.json configs[1]).  Utterances are independent, so ranks shard the batch with
no data-path collective (weak scaling: 64 utterances per GPU).  Rank 0 prints ONE JSON line with
  value        : utterances/s in the engine's default arithmetic (`dtype` names it: fp32 operands as two f16 pieces,
                 three products on the f16 MFMA, fp32 accumulate; every forward checks the saturation flags of that
                 arithmetic before it returns -- the check is inside the timed region);
  exact_fp32   : the SAME batch, steps and warm-up under DPRNN_GEMM_MODE=0 (fp32 operands on the fp32 MFMA, the
                 reference's own precision), timed right after the headline region, with its own roofline (peak 157.3);
  roofline     : the dominant kernel (fused BiLSTM recurrence, dprnn_rnn_fwd_ex) timed with HIP events
                 on its launch streams inside the timed region; achieved = algorithmic FLOPs per launch
                 (rows * 2 dirs * 2*256*512) / MEAN LAUNCH DURATION (what rocprofv3 --stats reports per kernel);
                 peak = the ceiling of the arithmetic that runs (2500 / 3 = 833 algorithmic TFLOP/s for three f16
                 products per product; 157.3 for the fp32 MFMA).  chip_level_tflops = the same FLOPs / the union of
                 the (overlapping) launch intervals;
  power        : socket power and shader clock (rocm-smi) during the timed region: the forward runs at the board's power limit
                 (~1380 W), which holds the clock near 1.73 of 2.4 GHz (profiles/r03_power_bound.txt);
  cpu_baseline : the reference's CPU path (oracle/dprnn_torch_cpu.py: the same ATen CPU ops the
                 reference dispatches to) timed on this host's cores per BASELINE.md section 3 (N=1 only).
  distributed  : what torch.distributed actually saw (world size, backend, one device name per rank).
"""
import argparse
import contextlib
import gc
import json
import os
import re
import shutil
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL / HIP IPC between the ranks of one node needs dmabuf handles on this driver: set before anything imports torch, in
# every launch form (the self-launch parent, a rank under the driver's torch.distributed.run, a plain single process)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

SAMPLES = 24000          # 3 s @ 8 kHz
PER_GPU_BATCH = 64       # BASELINE.json configs[1]
FLOP_PER_UTT = 350.80e9
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
CTOR = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
        "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["fwd", "train"], default="fwd",
                    help="fwd = the headline forward (configs[1]); train = fwd + PIT SI-SDR loss + bwd + gradient all-reduce "
                         "+ clip + Adam (configs[3], 64 utterances per GPU)")
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="utterances per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power", action="store_true", help="do not sample rocm-smi (socket power, shader clock) during the timed region")
    ap.add_argument("--no-optional-modes", action="store_true", help="skip the extra timing of the opt-in gate-GEMM modes")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the lines of BASELINE.json configs[2] (DPRNN-Spe cat, B=64) and configs[4] (DPRNN-Spe-IRA, full-length "
                         "utterances) that follow the headline region")
    ap.add_argument("--fast-act", type=int, default=1)
    ap.add_argument("--groups", type=int, default=None, help="utterance groups on separate HIP streams")
    ap.add_argument("--gemm-mode", type=int, default=None, choices=[0, 1, 3],
                    help="arithmetic of the contractions: 3 = fp32 operands as two f16 pieces of 2^e-scaled values, three products on "
                         "the f16 MFMA, fp32 accumulate (the engine's default, fp32-equivalent, saturation-checked); 1 = three bf16 "
                         "pieces, six products (fp32-equivalent, no range assumption); 0 = exact fp32 MFMA")
    ap.add_argument("--rowgemm-grid", type=int, default=None, help="workgroups of the HBM-bound f16 row GEMMs (dprnn_set_rowgemm_grid)")
    ap.add_argument("--min-group", type=int, default=None, help="smallest utterance group (engine.MIN_GROUP_BATCH)")
    ap.add_argument("--no-standalone", action="store_true", help="skip roofline.standalone (the dominant launch shapes timed alone on the chip, ~4 s): "
                    "under rocprofv3 its launches would mix into the per-kernel averages of the forward's own")
    ap.add_argument("--group-sizes", type=str, default="", help="experiment: explicit utterance-group sizes, e.g. 20,18,14,12")
    ap.add_argument("--seq-tile-intra", type=int, default=0)
    ap.add_argument("--serial-tile-intra", type=int, default=0, help="seq_tile of the direction-serial launches (engine.SEQ_TILE['serial_intra'])")
    ap.add_argument("--serial-tile-inter", type=int, default=0)
    ap.add_argument("--seq-tile-inter", type=int, default=0)
    return ap.parse_args(argv)


def _self_launch(n: int, script=None, argv=None) -> int:
    """`python bench.py --gpus N` without an outer torchrun: start N ranks as CHILD processes (one per GPU) from this
    parent, which has not touched the GPU (nothing but the standard library is imported yet) and never will; relay
    rank 0's JSON line and the children's exit code.  No exec: the parent waits and exits with the child's code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + \
        list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in proc.stdout:                                 # rank 0's JSON line -> stdout; library chatter (gloo) -> stderr
        out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
        out.write(line)
        out.flush()
    return proc.wait()


def _host_threads():
    """Thread count for the CPU leg: the physical cores this process may use."""
    logical = os.cpu_count() or 1
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = logical
    cores, model = set(), ""
    try:
        phys = None
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("physical id"):
                    phys = ln.split(":")[1].strip()
                elif ln.startswith("core id"):
                    cores.add((phys, ln.split(":")[1].strip()))
                elif ln.startswith("model name") and not model:
                    model = ln.split(":", 1)[1].strip()
    except OSError:
        pass
    physical = len(cores) or logical
    quota = None
    try:                                                     # cgroup v2 CPU quota of the box's share, if any
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
            if q != "max":
                quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    threads = min(v for v in (physical, allowed, quota) if v)
    return threads, dict(logical_cpus=logical, physical_cores=physical, affinity_cpus=allowed, cgroup_quota_cpus=quota,
                         cpu_model=model)


def cpu_baseline(sd, cfg, model, dev):
    """The CPU path timed beside the GPU one (the checker's leg of the bench: the only place bench.py touches oracle/),
    per BASELINE.md section 3: stock torch CPU ops, torch.set_num_threads(physical cores available), B = 1 (the
    reference's Inferencer / example.ipynb path, one utterance per call) and B = 8, 3 warm-up + 5 timed forwards each,
    median -> utt/s.  And -- the second half of BASELINE.json's metric, "SI-SDR vs ref" -- the parity of the GPU output with
    the CPU output on the same 8 two-source mixtures (SURVEY.md 8d recipe)."""
    import numpy as np
    import torch
    from oracle import dprnn_torch_cpu as otc
    from oracle.sisdr import pit_si_sdr, si_sdr
    from tss_with_dprnn_amd.synthetic import make_sources
    threads, host = _host_threads()
    torch.set_num_threads(threads)
    tsd = otc.to_torch_state(sd)
    src, mix = make_sources(8, SAMPLES)
    x = torch.from_numpy(mix)
    res, t_all = {}, time.perf_counter()
    for b in (1, 8):
        xb = x[:b]
        for _ in range(3):
            cpu_out = otc.tasnet_forward(tsd, cfg, xb)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            cpu_out = otc.tasnet_forward(tsd, cfg, xb)
            ts.append(time.perf_counter() - t0)
        res[b] = dict(utt_per_s=b / float(np.median(ts)), median_s=float(np.median(ts)), min_s=min(ts), max_s=max(ts))
    spent = time.perf_counter() - t_all
    best = max(res, key=lambda b: res[b]["utt_per_s"])
    base = dict(value=res[best]["utt_per_s"], unit="utt/s", cores=threads, kind="port",
                sample=f"oracle/dprnn_torch_cpu.py (stock torch CPU ops, no_grad), B=1 and B=8 x 3 s @ 8 kHz, 3 warm-up + 5 timed "
                       f"forwards each, median; value = B={best}; {threads} threads; {spent:.0f} s of CPU work",
                b1_utt_per_s=res[1]["utt_per_s"], b8_utt_per_s=res[8]["utt_per_s"], b1=res[1], b8=res[8], host=host)
    with torch.no_grad():
        gpu_out = model(x.to(dev)).cpu().numpy().astype(np.float64)
    cpu_out = cpu_out.numpy().astype(np.float64)
    T = min(gpu_out.shape[-1], src.shape[-1])
    s_gpu, _ = pit_si_sdr(gpu_out[..., :T], src[..., :T].astype(np.float64))      # best-permutation SI-SDR per utterance, dB
    s_cpu, _ = pit_si_sdr(cpu_out[..., :T], src[..., :T].astype(np.float64))
    parity = dict(sample="8 synthetic two-source mixtures x 3 s, GPU output vs the CPU path's output",
                  rel_l2=float(np.linalg.norm(gpu_out - cpu_out) / np.linalg.norm(cpu_out)),
                  si_sdr_gpu_vs_cpu_db=float(np.min(si_sdr(gpu_out.reshape(-1, gpu_out.shape[-1]), cpu_out.reshape(-1, cpu_out.shape[-1])))),
                  max_abs_delta_si_sdr_db=float(np.max(np.abs(np.asarray(s_gpu) - np.asarray(s_cpu)))),
                  tolerance_db=0.01)
    return base, parity


class PowerSampler:
    """Socket power and shader clock (rocm-smi, a child process every 0.4 s) while a timed region runs: the forward runs at
    the board's power limit, and the clock the power management leaves it is what the kernel cycle counts have to be read
    against (profiles/r03_power_bound.txt).  Rank 0 only; absent or unreadable rocm-smi -> no samples, nothing else changes."""

    def __init__(self, on, device_index=0):
        self.device_index = device_index
        self.samples, self._stop, self._thread = [], threading.Event(), None
        self.on = on and shutil.which("rocm-smi") is not None

    def _run(self):
        while not self._stop.is_set():
            try:
                txt = subprocess.run(["rocm-smi", "-d", str(self.device_index), "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
                pw = re.search(r"Power \(W\): ([0-9.]+)", txt)
                ck = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", txt)
                if pw and ck:
                    self.samples.append((float(pw.group(1)), int(ck.group(1))))
            except Exception:
                return
            self._stop.wait(0.4)

    def __enter__(self):
        if self.on:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=15)

    def summary(self):
        if not self.samples:
            return None
        tail = self.samples[len(self.samples) // 2:]            # rocm-smi's power is a running average: the later half has settled
        return {"socket_W": round(sum(p for p, _ in tail) / len(tail), 1), "sclk_MHz": round(sum(c for _, c in tail) / len(tail)),
                "samples": len(self.samples), "sclk_MHz_max_of_part": 2400,
                "note": "rocm-smi --showpower --showclocks every 0.4 s during the timed region, mean of the later half of the samples"}


def _launch_stats(events, t_ref):
    """Per-launch HIP-event durations on the launch streams.  With utterance groups on several streams the launches
    overlap, so the chip-level rate is (sum of algorithmic FLOPs) / (union of the launch intervals); kernel_ms is the
    plain mean launch duration (what rocprofv3 --stats reports)."""
    import numpy as np
    spans = sorted((t_ref.elapsed_time(a), t_ref.elapsed_time(b)) for a, b in events)
    kern_ms = float(np.mean([e - s for s, e in spans])) if spans else float("nan")
    union_ms, cur_s, cur_e = 0.0, None, None
    for s_, e_ in spans:
        if cur_e is None or s_ > cur_e:
            union_ms += (cur_e - cur_s) if cur_e is not None else 0.0
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    union_ms += (cur_e - cur_s) if cur_e is not None else 0.0
    conc = sum(e - s_ for s_, e in spans) / union_ms if union_ms > 0 else None
    return kern_ms, union_ms, conc, len(spans)


def standalone_folded_launches(engine, dev, B, S, K, ncu, seconds=0.8):
    """The four launch shapes of the headline forward's dominant kernel (dprnn_rnn_lin_split_fwd: phase 1 / 2 of an intra-chunk and
    of an inter-chunk path, one pipeline of B utterances, pending norm folded in) ALONE on the chip, each queued back to back for
    `seconds` and HIP-event timed over the last two thirds (the first third brings the board to its power limit).  In the forward
    two pipelines' launches share the chip, which is what `roofline.kernel_ms` sees; this is the same kernel without a neighbour.
    Same argument lists as engine._path_folded; synthetic operands (the kernel's time does not depend on the values)."""
    import torch
    st = engine.Stage(dev)
    lib = st.lib
    g = torch.Generator(device=dev).manual_seed(1)
    rows = B * S * K
    p = lambda t: t.data_ptr()
    X = torch.randn(rows, 128, device=dev, generator=g) * 0.5
    Z = torch.randn(rows, 128, device=dev, generator=g)
    X2, Z2 = torch.empty_like(X), torch.empty_like(Z)
    stats = torch.zeros(B, 2, device=dev, dtype=torch.float64)
    stats[:, 1] = S * K * 128
    stats2 = torch.zeros_like(stats)
    gamma, beta = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device=dev)
    bsum = torch.empty(2 * 512, device=dev)
    w = [torch.randn(512, 128, device=dev, generator=g) * 0.08, torch.randn(512, 128, device=dev, generator=g) * 0.08,
         torch.randn(512, device=dev, generator=g) * 0.05, torch.randn(512, device=dev, generator=g) * 0.05]
    st.call("dprnn_rnn_pack_ex", 3, 0, *[p(t) for t in w], *[p(t) for t in w], p(packed), p(bsum))
    Wp = torch.randn(128, 256, device=dev, generator=g) / 16
    bp = torch.randn(128, device=dev, generator=g)
    linp = torch.empty(lib.dprnn_lin_pack_floats(), device=dev)
    st.call("dprnn_lin_pack_f16", p(Wp), 2, p(linp))
    out = {}
    for name, intra, addr, per in (("intra", True, (B * S, K, B * S, 0, K, 1), S), ("inter", False, (B * K, S, K, S * K, 1, K), K)):
        form, tile = engine.fold_plan(B * per, ncu, 2, intra)          # the tile the two-pipeline forward picks for this path
        if form != "split":
            return {"skipped": f"the forward's {name} path is not in the time-split folded form here ({form})"}
        state = torch.empty(lib.dprnn_rnn_lin_state_floats(B * per, tile), device=dev)
        tail = (p(packed), p(bsum), p(linp))
        a1 = (1, p(X), p(Z), p(stats), p(gamma), p(beta), 1e-8, per, S * K, p(X2), *tail, 0, p(Z2), 0, p(state), *addr, tile)
        a2 = (2, p(X2), 0, 0, 0, 0, 0.0, per, S * K, 0, *tail, p(bp), p(Z2), p(stats2), p(state), *addr, tile)
        st.call("dprnn_rnn_lin_split_fwd", *a1)           # phase 2 continues from the state phase 1 of the same path left
        for ph, a in ((1, a1), (2, a2)):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            for _ in range(8):
                st.call("dprnn_rnn_lin_split_fwd", *a)
            e1.record()
            torch.cuda.synchronize(dev)
            n = max(12, int(seconds * 1e3 / (e0.elapsed_time(e1) / 8)))
            for _ in range(n // 3):
                st.call("dprnn_rnn_lin_split_fwd", *a)
            e1.record()
            for _ in range(n - n // 3):
                st.call("dprnn_rnn_lin_split_fwd", *a)
            e2.record()
            torch.cuda.synchronize(dev)
            out[f"{name}_phase{ph}_ms"] = e1.elapsed_time(e2) / (n - n // 3)
        del state
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args.gpus))

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but torch.distributed.run started {world} ranks")
    # rehearsal knobs (a one-GPU box cannot host two RCCL ranks): DPRNN_BENCH_BACKEND=gloo DPRNN_BENCH_SHARE_GPU=1
    backend = os.environ.get("DPRNN_BENCH_BACKEND", "nccl")
    if os.environ.get("DPRNN_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    dist_info = dict(world_size=1, backend=None, devices=[torch.cuda.get_device_name(dev)])
    # under torch.distributed.run the process group is formed for ONE rank too: the driver's launch form then exercises the same
    # RCCL init / barrier / max-over-ranks calls at every N
    use_dist = world > 1 or ("WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ)
    if use_dist:
        import ctypes
        import torch.distributed as dist
        # RCCL prints a version banner on the C stdout when its first communicator comes up; stdout carries ONE JSON line, so the
        # file descriptor points at stderr while the group forms
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=dev)      # RCCL over xGMI
            else:
                dist.init_process_group(backend=backend)
            names = [None] * world
            dist.all_gather_object(names, f"rank {rank}: cuda:{local_rank} {torch.cuda.get_device_name(dev)}")
            dist.barrier()
            torch.cuda.synchronize(dev)
        finally:
            sys.stdout.flush()
            ctypes.CDLL(None).fflush(None)
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        dist_info = dict(world_size=dist.get_world_size(), backend=dist.get_backend(), devices=names)

    import tss_with_dprnn_amd as pkg
    from tss_with_dprnn_amd import engine, sharding
    from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_sources, make_state

    cfg = full_cfg()
    model = pkg.DPRNNTasNet(**{k: cfg[k] for k in CTOR}).eval()
    sd = make_state("bss", cfg, 0)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model = model.to(dev)
    lib = pkg.load_library()
    lib.dprnn_set_fast_activations(args.fast_act)
    lib.dprnn_set_fast_head_activations(args.fast_act)
    if args.rowgemm_grid is not None:
        lib.dprnn_set_rowgemm_grid(args.rowgemm_grid)
    engine.SEQ_TILE["intra"], engine.SEQ_TILE["inter"] = args.seq_tile_intra, args.seq_tile_inter
    engine.SEQ_TILE["serial_intra"], engine.SEQ_TILE["serial_inter"] = args.serial_tile_intra, args.serial_tile_inter
    if args.groups is not None:
        engine.STREAM_GROUPS = args.groups
    if args.min_group is not None:
        engine.MIN_GROUP_BATCH = args.min_group
    if args.gemm_mode is None:
        args.gemm_mode = engine.DEFAULT_GEMM_MODE
    engine.GEMM_MODE = args.gemm_mode
    if args.groups is not None:
        engine.SPLIT_ONE_GROUP = False          # an explicit group count also applies to the bf16-split modes
    if args.group_sizes:
        engine.GROUP_SIZES = [int(v) for v in args.group_sizes.split(",")]

    B = args.batch

    def sync():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    geo = engine.make_geometry(B, SAMPLES, cfg["input_size"], cfg["chunk_length"], cfg["hop_length"],
                               cfg["kernel_size"], cfg["stride"])
    lstm_flop_per_step = geo.rows * 2 * (2 * 256 * 512) * 2 * cfg["n_repeats"]      # rows * 2 dirs * (2*256*512) per path

    if args.mode == "train":
        line = bench_train(args, model, dev, world, rank, dist if world > 1 else None, sync, lstm_flop_per_step, make_sources)
        if rank == 0:
            line["distributed"] = dist_info
            print(json.dumps(line), flush=True)
        if use_dist:
            dist.destroy_process_group()
        return

    x = torch.from_numpy(make_mixture(B, SAMPLES, seed=1234 + rank)).to(dev)     # resident in HBM before timing
    REC = "dprnn_rnn_fwd_ex"            # the recurrence launches (the dominant kernel) ...
    FOLD = "dprnn_rnn_lin_split_fwd"    # ... with the following Linear folded in (round 5, the default in mode 3): two launches per path, each
    #                                     half the steps of both directions (csrc/recur_split8_f16_lin.hip)
    REC_NAMES = {REC, "dprnn_rnn_fwd_hp_ex", FOLD, "dprnn_rnn_lin_fwd"}     # ... under any entry point

    def host_state():
        """What can stall the enqueueing host: the caching allocator going back to the driver (segments allocated / freed, retries) and
        the cgroup throttling this process (periods in which the box's CPU quota ran out)."""
        ms = torch.cuda.memory_stats(dev)
        out = {k: int(ms.get(k, 0)) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_ooms")}
        out["reserved_GiB"] = ms.get("reserved_bytes.all.current", 0) / 2**30
        for p_ in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
            if os.path.exists(p_):
                d_ = dict(ln.split() for ln in open(p_).read().strip().splitlines())
                out.update({k: int(d_[k]) for k in ("nr_throttled", "throttled_usec") if k in d_})
                break
        return out

    def timed_region(mode, steps, warmup, profile=False, power=None):
        """`warmup` untimed forwards, then exactly `steps` timed ones between barriers, under gate-GEMM mode `mode`.  Every
        forward ends with the saturation check of the f16-pieces arithmetic (one stream synchronisation), so the host clock
        around each call is that step's latency.  profile: HIP events around every recurrence launch (engine.PROFILE) -- this
        forces the eager launch sequence, a replayed HIP graph has no place for them."""
        engine.GEMM_MODE = mode
        with torch.no_grad():
            for _ in range(warmup):
                o = model(x)
            sync()
            engine.PROFILE = {"names": set(REC_NAMES), "events": {}} if profile else None
            gc.collect()
            gc.disable()                # the host enqueues ~300 launches per step: no collector pauses inside the timed region
            t_ref = torch.cuda.Event(enable_timing=True)
            t_ref.record()
            per_step = []
            h0 = host_state()
            t0 = time.perf_counter()
            tp = t0
            with (power if power is not None else contextlib.nullcontext()):
                for _ in range(steps):
                    o = model(x)
                    if mode != 3:           # modes without the check return before the GPU is done: make the step boundary explicit
                        torch.cuda.current_stream(dev).synchronize()
                    tn = time.perf_counter()
                    per_step.append(tn - tp)
                    tp = tn
                sync()
                dt = time.perf_counter() - t0
            h1 = host_state()
            gc.enable()
            prof, engine.PROFILE = engine.PROFILE, None
        if not profile or not engine.GRAPHS:      # (the eager profiling pass after a graph-replay region is not the region the line reports)
            host_delta[mode] = {k: (h1[k] - h0[k] if k != "reserved_GiB" else h1[k]) for k in h1}
        return o, dt, per_step, prof, t_ref

    host_delta = {}

    sat0 = dict(engine.SATURATION)
    graphs0 = dict(engine.GRAPH_STATS)
    # With HIP graphs on (the engine's default) the timed region replays the captured forward; the per-launch kernel timing of the
    # roofline then comes from a second, eager pass of the same batch (PROF_STEPS steps).  DPRNN_GRAPHS=0: one region serves both.
    PROF_STEPS = min(args.steps, 10)
    power = PowerSampler(rank == 0 and not args.no_power, local_rank)
    out, elapsed, per_step, prof, t_ref = timed_region(args.gemm_mode, args.steps, args.warmup, profile=not engine.GRAPHS, power=power)
    graph_replays = engine.GRAPH_STATS["replays"] - graphs0["replays"]
    prof_steps = args.steps
    if engine.GRAPHS:
        _, _, _, prof, t_ref = timed_region(args.gemm_mode, PROF_STEPS, 1, profile=True)
        prof_steps = PROF_STEPS
    assert out.shape == (B, 2, SAMPLES) and bool(torch.isfinite(out).all())
    sat_checks = engine.SATURATION["checks"] - sat0["checks"]
    sat_fallbacks = engine.SATURATION["fallbacks"] - sat0["fallbacks"]
    if use_dist and world == 1:         # one rank under torch.distributed.run: the same collective the N > 1 runs use
        t_el = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else None)
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        elapsed = float(t_el.item())
    else:
        elapsed = sharding.max_over_ranks(elapsed, device=dev if backend == "nccl" else None)   # slowest rank defines the step

    BF16_PEAK = 2500.0                  # dense 16-bit MFMA peak (f16 = bf16), /opt/skills/guides/MI355X_MICROARCH.md
    PRODUCTS = {0: 1, 1: 6, 3: 3}

    def roofline(mode, prof, t_ref, steps, utt_per_s):
        """Roofline object of the recurrence kernel for a timed region run under `mode`."""
        kern_ms, union_ms, conc, n_ev = _launch_stats([ev for nm in sorted(REC_NAMES) for ev in prof["events"].get(nm, [])], t_ref)
        engine.GEMM_MODE = mode
        n_groups = len(engine.group_parts(B, engine.effective_gemm_mode(B)))
        folded = bool(prof["events"].get(FOLD))
        if folded:
            # a launch meets every row of its pipeline ONCE (half the steps of both directions): the gate GEMM [x | h] @ [W_ih | W_hh]^T of
            # one direction plus that direction's half of the Linear 256 -> 128
            flop_per_launch = geo.rows / n_groups * (2 * 256 * 512 + 2 * 128 * 128)
            flop_chip = geo.rows * 2 * (2 * 256 * 512 + 2 * 128 * 128) * 2 * cfg["n_repeats"]
        else:
            flop_per_launch = geo.rows / n_groups * 2 * (2 * 256 * 512)          # rows * 2 dirs * (2*256*512)
            flop_chip = lstm_flop_per_step
        per_launch = flop_per_launch / (kern_ms * 1e-3) / 1e12
        chip = flop_chip * steps / (union_ms * 1e-3) / 1e12 if union_ms > 0 else float("nan")
        products = PRODUCTS[mode]
        peak = PEAK_F32_MFMA_TFLOPS if mode == 0 else BF16_PEAK / products
        piece = "f16" if mode == 3 else "bf16"
        traffic, traffic_src = None, None
        try:
            names, key = {0: (["r01_pmc_traffic.json"], "bilstm_kernel"), 1: (["r02_pmc_traffic.json"], "bilstm_split8_kernel"),
                          3: ((["r05_pmc_traffic_f16.json"] if folded else ["r05_pmc_traffic_f16_unfolded_same_box.json", "r04_pmc_traffic_f16.json"]),
                              "bilstm_split8_kernel_f16")}[mode]
            name = next(n for n in names if os.path.exists(os.path.join(ROOT, "profiles", n)))
            with open(os.path.join(ROOT, "profiles", name)) as f:
                pm = json.load(f)[key]
            if pm["group_batch"] * n_groups == B:
                traffic, traffic_src = pm["traffic_GB_per_launch"], name
        except (OSError, KeyError, ValueError, StopIteration):
            pass
        return {"bound": "mfma",
                "kernel": "bilstm_kernel (dprnn_rnn_fwd_ex, LSTM cell, exact fp32 MFMA)" if mode == 0 else
                ("bilstm_split8_kernel<.., PJ> (dprnn_rnn_lin_split_fwd: LSTM cell + the following Linear folded in, both directions per launch, the "
                 f"path cut in time into two launches; fp32 operands as {piece} pieces, {products} products on the {piece} MFMA)" if folded else
                 f"bilstm_split8_kernel (dprnn_rnn_fwd_ex, LSTM cell, fp32 operands as {piece} pieces, {products} products on the {piece} MFMA)"),
                # achieved: algorithmic FLOPs of one launch / the MEAN duration of a launch (HIP events on the launch streams,
                # = rocprofv3 --stats' per-kernel average).  The utterance groups' launches overlap on the chip, so this includes
                # the time a launch waits for CUs the other group's launch still holds.
                "achieved": per_launch, "peak": peak, "unit": "TFLOP/s (algorithmic = fp32-equivalent)",
                "frac": per_launch / peak if per_launch is not None else None,
                "traffic": traffic, "traffic_unit": f"GB per launch (PMC, profiles/{traffic_src})",
                "peak_note": "157.3 = fp32 MFMA peak" if mode == 0 else
                f"{BF16_PEAK:.0f} TFLOP/s dense {piece} MFMA / {products} products per algorithmic FLOP; the fp32 MFMA peak is "
                f"{PEAK_F32_MFMA_TFLOPS} (achieved / that = {per_launch / PEAK_F32_MFMA_TFLOPS:.2f})",
                "executed_mfma_tflops": None if mode == 0 else per_launch * products,
                # the data-sheet peak assumes 2.4 GHz; this forward runs at the board's power limit, which holds the shader clock
                # lower (the line's `power` object), and a register-only f16 MFMA loop reaches 2024 TFLOP/s on this chip for the
                # same reason (profiles/r03_power_bound.txt): frac against `peak` cannot exceed ~0.8
                "peak_at_power_limit_note": None if mode == 0 else "1852 TFLOP/s dense f16 MFMA sustained for 4 s with operands in registers, "
                "0.70 pJ/FLOP, 1.93 GHz at ~1300 W (tools/probes/mfma_shape_energy.hip, profiles/r04_mfma_shape_energy.txt; short bursts 2024: "
                "mfma_f16_coexec.hip) = 617 algorithmic; the forward holds the socket at ~1380 W (see `power`)",
                # the kernel runs at the board's power limit, so time = energy: where the joules of a step go (MFMAs 46-56 %, HBM traffic
                # 19 %, weight stream from L2 8-12 %, cell arithmetic 9-11 %), measured with the socket's energy counter
                "energy_budget": None if mode == 0 else "profiles/r04_energy_budget.txt",
                # ... and of a whole forward: folded recurrence launches ~100 % (round 4: recurrences 83 %, projections 16 %); the forward takes the SUM of its kernels' stand-alone
                # times (each is at the cap alone), so overlapping pipelines neither helps nor hurts
                "energy_ledger": None if mode == 0 else "profiles/r05_forward_energy_ledger.txt",
                "algorithmic_gflop_per_launch": flop_per_launch / 1e9,
                # slab units per launch: read X per direction (2) + write H (2) [+ read Z per direction (2) + write x' (1) when
                # the previous norm is folded in: all launches of a forward but the first]
                # folded: phase 1 reads x, z and writes x', partial products (4; the first path has no pending norm: 2), phase 2 reads x',
                # partial products and writes z (3): 7 units per path where rounds 2-4 moved 10 (profiles/r05_pmc_traffic_f16.json: 7.49 measured)
                "algorithmic_gb_per_launch": geo.rows / n_groups * 128 * 4 *
                (((4 + 3) * (2 * cfg["n_repeats"] - 1) + (2 + 3)) / (4 * cfg["n_repeats"]) if folded else
                 (4 + 7 * (2 * cfg["n_repeats"] - 1)) / (2 * cfg["n_repeats"]) if engine.FUSE_NORM else 4) / 1e9,
                "kernel_ms": kern_ms, "launches_timed": n_ev, "launches_per_step": (4 if folded else 2) * cfg["n_repeats"] * n_groups,
                "linear_folded_in": folded,
                # reading `frac` across rounds: rounds 2-4 timed the recurrence launch alone (813.7 GFLOP in ~4.3 ms between two
                # pipelines: 0.23) and left the projection kernel that followed it (101.7 GFLOP, ~1.6 ms in the forward) out; the folded
                # launches carry that work and that time -- like for like, round 4 was (813.7 + 101.7) GFLOP / (4.3 + 1.6) ms = 0.186
                "frac_note": ("per launch of one phase (half the steps of both directions + their halves of the Linear); a launch shares the chip "
                              "with the other pipeline's, so its duration is about twice its stand-alone time: chip_level_frac is the rate of "
                              "the chip while any of them is in flight.  Round 4's 0.23 excluded the projection kernel; with it: 0.186") if folded else None,
                "concurrent_streams": n_groups, "busy_ms_per_step": union_ms / steps,
                # the same FLOPs over the UNION of the launch intervals: the chip-level rate while any recurrence launch is in flight
                "chip_level_tflops": chip, "chip_level_frac": chip / peak, "mean_concurrency": conc,
                "whole_forward_frac": utt_per_s * FLOP_PER_UTT / 1e12 / peak,
                # the HBM side of the same forward -- 1.433 GB per utterance with ideal fusion
                # against 8 TB/s (the path is matrix-bound: this stays a small fraction)
                "whole_forward_hbm_tb_s": utt_per_s * 1.433e9 / 1e12,
                "whole_forward_hbm_frac": utt_per_s * 1.433e9 / 8e12,
                "whole_forward_tflops": utt_per_s * FLOP_PER_UTT / 1e12}

    def latency(per_step):
        v = np.sort(np.asarray(per_step)) * 1e3
        slow = [int(i) for i in np.argsort(per_step)[::-1][:3]]
        return {"ms_per_step_p50": float(np.percentile(v, 50)), "ms_per_step_p99": float(np.percentile(v, 99)),
                "ms_per_step_min": float(v[0]), "ms_per_step_max": float(v[-1]), "steps": int(v.size),
                "p99_over_p50": float(np.percentile(v, 99) / np.percentile(v, 50)),
                "slowest_steps": {str(i): round(1e3 * per_step[i], 2) for i in slow}}

    value = world * B * args.steps / elapsed
    head_roof = roofline(args.gemm_mode, prof, t_ref, prof_steps, value / world)
    head_roof["timed_in"] = ("the timed region itself (eager launches)" if not engine.GRAPHS else
                             f"a second pass of {prof_steps} eager forwards of the same batch right after the timed region, which replays a "
                             "captured HIP graph (no place for events inside a graph)")
    n_groups = head_roof["concurrent_streams"]
    if rank == 0 and world == 1 and args.gemm_mode == 3 and head_roof["linear_folded_in"] and n_groups == 2 and B % 2 == 0 and not args.group_sizes and not args.no_standalone:
        # the same kernel without the other pipeline's launches beside it, measured live (~4 s); never inside the timed region
        try:
            sa = standalone_folded_launches(engine, dev, B // n_groups, geo.S, geo.K, lib.dprnn_num_cus())
            ms = [v for k, v in sa.items() if k.endswith("_ms")]
            if ms:
                sa_ms = float(np.mean(ms))
                sa_tflops = head_roof["algorithmic_gflop_per_launch"] / sa_ms
                sa.update({"kernel_ms": sa_ms, "achieved": sa_tflops, "frac": sa_tflops / head_roof["peak"],
                           "note": "one pipeline's launch alone on the chip (mean of the four launch shapes of a DPRNN block, each queued back to "
                                   "back at the board's power limit); `frac` above is the same launch while the other pipeline's shares the chip"})
            head_roof["standalone"] = sa
        except Exception as exc:          # a diagnostic: never costs the line
            head_roof["standalone"] = {"error": f"{type(exc).__name__}: {exc}"}
        torch.cuda.empty_cache()

    # The reference's own precision, first class: the same batch, the same number of steps and warm-ups under
    # DPRNN_GEMM_MODE=0 (fp32 operands on the fp32 MFMA), with its own roofline against the fp32 MFMA peak.
    exact = None
    ref_out = out.double()
    if args.gemm_mode != 0 and world == 1 and not args.no_optional_modes:
        o0, dt0, ps0, prof0, tref0 = timed_region(0, args.steps, args.warmup, profile=not engine.GRAPHS)
        v0 = B * args.steps / dt0
        ex_steps = args.steps
        if engine.GRAPHS:
            _, _, _, prof0, tref0 = timed_region(0, PROF_STEPS, 1, profile=True)
            ex_steps = PROF_STEPS
        exact = dict({"metric": "utterances/sec (3 s @ 8 kHz) DPRNN-TasNet fwd, exact fp32 MFMA arithmetic (DPRNN_GEMM_MODE=0)",
                      "value": v0, "unit": "utt/s", "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt0 / args.steps,
                      "dtype": "f32", "rel_l2_vs_headline_output": float((o0.double() - ref_out).norm() / ref_out.norm()),
                      "roofline": roofline(0, prof0, tref0, ex_steps, v0)}, **latency(ps0))
        del o0
    # mode 1 (three bf16 pieces, six products: the fall-back arithmetic of a saturated forward), 3 steps after the timed regions
    mode_names = {0: "exact_fp32_mfma", 1: "bf16_split_6_products", 3: "f16_pieces_3_products"}
    optional = None
    if world == 1 and not args.no_optional_modes:
        optional = {}
        with torch.no_grad():
            for mode in (1, 3):
                if mode == args.gemm_mode:
                    continue
                engine.GEMM_MODE = mode
                o = model(x)
                sync()
                t1 = time.perf_counter()
                for _ in range(3):
                    o = model(x)
                sync()
                dt = (time.perf_counter() - t1) / 3
                optional[mode_names[mode]] = {"value": B / dt, "unit": "utt/s", "ms_per_step": 1e3 * dt,
                                              "rel_l2_vs_headline_mode": float((o.double() - ref_out).norm() / ref_out.norm())}
    others = None
    if world == 1 and not args.no_other_configs and not args.no_optional_modes:
        t_oc = time.perf_counter()
        others = bench_other_configs(args, dev, args.gemm_mode)
        # BASELINE.json configs[3]'s per-GPU share: the training step (forward with tape, PIT SI-SDR loss, backward, clip, Adam) on a copy of
        # the model, 5 warm-ups + 10 steps, with its own roofline object (bench_train: what `--mode train` prints as a line of its own)
        import argparse
        import copy
        engine.drop_graphs()
        del out, ref_out
        torch.cuda.empty_cache()
        targs = argparse.Namespace(**dict(vars(args), steps=10, warmup=5, no_power=True))
        tmodel = copy.deepcopy(model)
        try:
            others["train_b64"] = bench_train(targs, tmodel, dev, world, rank, None, sync, lstm_flop_per_step, make_sources)
        finally:
            del tmodel
            torch.cuda.empty_cache()
        others["seconds_spent"] = time.perf_counter() - t_oc
        others["note"] = ("timed after the headline region on the same GPU, same arithmetic as `value` (dtype), inputs resident in HBM; "
                          "NOT part of `value`")
    engine.GEMM_MODE = args.gemm_mode
    dtype = {0: "f32", 1: "f32 via 3 x bf16 split, 6 products, f32 accumulate",
             3: "f32 via 2 x f16 pieces of 2^e-scaled operands (22+ bits), 3 products, f32 accumulate; saturation-checked per forward"}[args.gemm_mode]
    if rank == 0:
        line = {
            "metric": "utterances/sec (3 s @ 8 kHz) DPRNN-TasNet fwd", "value": value, "unit": "utt/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": "DPRNN-TasNet BSS inference, synthetic 3 s @ 8 kHz, batch=64 per GPU "
                                   "(BASELINE.json configs[1])",
                       "per_gpu_batch": B, "samples": SAMPLES, "parallelism": f"utterance-sharded x{world}",
                       "fast_activations": bool(args.fast_act), "stream_groups": engine.STREAM_GROUPS,
                       "stream_groups_used": n_groups, "rowgemm_grid": lib.dprnn_get_rowgemm_grid(),
                       "hip_graphs": bool(engine.GRAPHS), "graph_replays_in_timed_region": graph_replays,
                       "gemm_mode": {0: "exact fp32 MFMA (DPRNN_GEMM_MODE=0)", 1: "fp32-equivalent bf16 split, 6 products (DPRNN_GEMM_MODE=1)",
                                     3: "fp32-equivalent f16 pieces, 3 products (DPRNN_GEMM_MODE=3)"}[args.gemm_mode]
                       + (" (default)" if args.gemm_mode == engine.DEFAULT_GEMM_MODE else "")},
            "latency": latency(per_step),
            "power": power.summary(),
            "host": dict(host_delta.get(args.gemm_mode, {}), gc="disabled in the timed region",
                         note="deltas over the timed region: driver-level allocations / frees of the caching allocator, allocation "
                              "retries, cgroup CPU-throttling periods of this process; reserved_GiB at its end"),
            "saturation": {"checks": sat_checks, "fallbacks": sat_fallbacks,
                           "note": "every forward in the f16-pieces arithmetic reads its saturation flags before it returns (inside the "
                                   "timed region); a fall-back re-runs the forward in mode 1"},
            "roofline": head_roof,
            "distributed": dist_info,
        }
        if exact is not None:
            line["exact_fp32"] = exact
        if optional:
            line["other_gemm_modes"] = dict(optional, note="the same batch under the other split setting of engine.GEMM_MODE / "
                                            "DPRNN_GEMM_MODE, 3 steps after the timed regions; NOT part of `value`")
        if others:
            line["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], line["parity"] = cpu_baseline(sd, cfg, model, dev)
            line["speedup_vs_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


SPE_CTOR = CTOR + ["O", "P", "embeddings_size", "num_spks", "fusion_type"]
FLOP_PER_UTT_SPE = 356.4e9       # DPRNN-Spe (cat), 3-s mixture + 3-s reference
FLOP_PER_UTT_IRA = 712.8e9       # two speaker-encoder + two DPRNN passes, per 3 s of mixture


def bench_other_configs(args, dev, mode):
    """BASELINE.json configs[2] and configs[4] on this GPU, after the headline region, under the same discipline (untimed
    warm-ups, then timed steps between synchronisations, inputs resident in HBM), in the headline's arithmetic:
      spe_cat_b64      DPRNN-Spe (cat fusion), 64 x (3-s mixture + 3-s reference)            dprnn_spe.py:314-327
      ira_full_length  DPRNN-Spe-IRA over full-length utterances (Libri2Mix-test length distribution, one rank's
                       length-balanced eighth of the 3000), batch-of-one semantics per utterance through
                       inference.separate_many's ragged batches                              dprnn_spe_ira.py:179-190, inferencer_spe.py:20-54
    Each carries value (utt/s), ms_per_step, its algorithmic GFLOP and a roofline object: the recurrence launches timed with HIP
    events (one eager pass), algorithmic FLOPs per launch / mean launch duration against the arithmetic's ceiling."""
    import numpy as np
    import torch
    import tss_with_dprnn_amd as pkg
    from tss_with_dprnn_amd import engine, inference, sharding
    from tss_with_dprnn_amd.synthetic import full_cfg, make_mixture, make_state
    products = {0: 1, 1: 6, 3: 3}[mode]
    peak = PEAK_F32_MFMA_TFLOPS if mode == 0 else 2500.0 / products
    engine.GEMM_MODE = mode
    folded_names = {"dprnn_rnn_lin_split_fwd", "dprnn_rnn_lin_split_ragged_fwd"}
    rec_names = {"dprnn_rnn_fwd_ex", "dprnn_rnn_fwd_hp_ex", "dprnn_rnn_fwd_ragged_ex", "dprnn_rnn_fwd_ragged_hp_ex"} | folded_names

    def build(kind, cls):
        cfg = full_cfg(norm_type="ln", fusion_type="cat")
        m = cls(**{k: cfg[k] for k in SPE_CTOR}).eval()
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state(kind, cfg, 0).items()}, strict=True)
        return m.to(dev)

    def timed(fn, warmup, steps):
        with torch.no_grad():
            for _ in range(warmup):
                fn()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize(dev)
            return (time.perf_counter() - t0) / steps

    def rec_roofline(fn, flop_per_launch_of, whole_tflops):
        """One eager pass with HIP events around every recurrence launch."""
        engine.PROFILE = {"names": set(rec_names), "events": {}}
        t_ref = torch.cuda.Event(enable_timing=True)
        t_ref.record()
        try:
            with torch.no_grad():
                fn()
            torch.cuda.synchronize(dev)
        finally:
            prof, engine.PROFILE = engine.PROFILE, None
        ev = [e for n in sorted(rec_names) for e in prof["events"].get(n, [])]
        kern_ms, union_ms, conc, n_ev = _launch_stats(ev, t_ref)
        # a folded launch meets every row once: one direction's gate GEMM + its half of the Linear (see the headline's roofline)
        folded = any(prof["events"].get(n) for n in folded_names)
        per_launch = flop_per_launch_of(n_ev, folded) / (kern_ms * 1e-3) / 1e12 if n_ev else None
        if not n_ev:
            kern_ms = None
        return {"bound": "mfma", "kernel": "the fused BiLSTM recurrence (bilstm_split8_kernel / bilstm_kernel), as in the headline's roofline",
                "linear_folded_in": folded,
                "achieved": per_launch, "peak": peak, "unit": "TFLOP/s (algorithmic = fp32-equivalent)",
                "frac": per_launch / peak if per_launch is not None else None,
                "traffic": None, "kernel_ms": kern_ms, "launches_timed": n_ev, "mean_concurrency": conc,
                "whole_forward_tflops": whole_tflops, "whole_forward_frac": whole_tflops / peak,
                "timed_in": "one eager pass after the timed steps (HIP events on the launch streams)"}

    out = {}
    # ---- configs[2]: DPRNN-Spe (cat), B = 64
    B = PER_GPU_BATCH
    m = build("spe", pkg.DPRNNSpeTasNet)
    x = torch.from_numpy(make_mixture(B, SAMPLES, seed=2)).to(dev)
    a = torch.from_numpy(make_mixture(B, SAMPLES, seed=3)).to(dev)
    ln = torch.tensor(float(SAMPLES))
    steps, warm = min(args.steps, 20), max(3, min(args.warmup, 5))         # eager, capture, replay: three forwards before the clock
    dt = timed(lambda: m(x, a, ln), warm, steps)
    geo = engine.make_geometry(B, SAMPLES, 64, 250, 125, 2, 1)
    n_groups = len(engine.group_parts(B, engine.effective_gemm_mode(B)))
    out["spe_cat_b64"] = {
        "workload": "DPRNN-Spe (cat fusion) TSS inference with the speaker-embedding branch, synthetic 3-s mixture + 3-s reference, "
                    "batch=64 (BASELINE.json configs[2])",
        "value": B / dt, "unit": "utt/s", "ms_per_step": 1e3 * dt, "steps": steps, "warmup": warm,
        "algorithmic_gflop_per_utterance": FLOP_PER_UTT_SPE / 1e9,
        "roofline": rec_roofline(lambda: m(x, a, ln), lambda n_ev, folded: geo.rows / n_groups * ((2 * 256 * 512 + 2 * 128 * 128) if folded else 2 * (2 * 256 * 512)),
                                 B / dt * FLOP_PER_UTT_SPE / 1e12)}
    del m, x, a
    engine.drop_graphs()
    # ---- configs[4]: DPRNN-Spe-IRA, full-length utterances (this rank's eighth of the 3000-utterance test-set length distribution)
    m = build("ira", pkg.DPRNNSpeIRATasNet)
    rng = np.random.default_rng(0)
    lens = np.clip(np.exp(rng.normal(np.log(36720), 0.38, 3000)), 24000, 111920).astype(int)
    share = sharding.shard_by_length(lens.tolist(), 8)[0]
    order = sorted(share, key=lambda i: -lens[i])            # length-sorted: a ragged batch holds utterances of similar length
    wavs = [torch.from_numpy(make_mixture(1, int(lens[i]), seed=100 + i)[0]).to(dev) for i in order]
    aux = [(torch.from_numpy(make_mixture(1, SAMPLES, seed=900 + i)[0]).to(dev), float(SAMPLES)) for i in order]
    secs = float(sum(lens[i] for i in order)) / 8000.0
    gflop = float(sum(FLOP_PER_UTT_IRA * lens[i] / SAMPLES for i in order)) / 1e9
    run = lambda: inference.separate_many(m, wavs, extra_inputs=aux, ragged=True)
    dt = timed(run, 1, 1)
    rows_total = float(sum(engine.make_geometry(1, int(lens[i]), 64, 250, 125, 2, 1).rows for i in order))
    n_batches = len(inference.ragged_chunks(list(range(len(order))), [int(lens[i]) for i in order], dev))
    out["ira_full_length"] = {
        "workload": f"DPRNN-Spe-IRA 2-iteration refined inference, {len(order)} full-length utterances (one rank's length-balanced "
                    "eighth of 3000 drawn from the Libri2Mix-test length distribution, 3 .. 14 s), batch-of-one semantics per utterance, "
                    f"{n_batches} ragged batch(es) of <= {inference.RAGGED_MAX_SAMPLES / 8000:.0f} s of audio (BASELINE.json configs[4], one GPU's share)",
        "value": len(order) / dt, "unit": "utt/s", "ms_per_step": 1e3 * dt, "steps": 1, "warmup": 1,
        "step": "one pass over the rank's whole share", "audio_seconds": secs, "times_real_time": secs / dt,
        "algorithmic_gflop_per_step": gflop, "algorithmic_gflop_per_utterance_of_3s": FLOP_PER_UTT_IRA / 1e9,
        # launches of a pass: 2 DPRNN passes x 12 paths (x 2 time phases when the Linear is folded in) per ragged batch; mean FLOPs per
        # launch = all slab rows x (2 dirs x 2*256*512 [+ the Linear's 2*256*128]) x 24 / launches
        "roofline": rec_roofline(run, lambda n_ev, folded: rows_total * (2 * (2 * 256 * 512) + (2 * 256 * 128 if folded else 0)) * 24 / max(n_ev, 1),
                                 gflop / dt / 1e3),
        "ragged_batches_per_step": n_batches}
    del m, wavs, aux
    engine.drop_graphs()
    # ---- beyond BASELINE.json's widths: a DPRNN-TasNet at feature_size = hidden_size = 256 (the reference constructor takes any width,
    # dprnn.py:117-120) on the any-width kernels (wide.py, csrc/wide.hip: exact fp32 MFMA arithmetic whatever the mode), 64 x 3 s
    F2 = H2 = 256
    cfg = full_cfg(feature_size=F2, hidden_size=H2)
    m = pkg.DPRNNTasNet(**{k: cfg[k] for k in CTOR}).eval()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in make_state("bss", cfg, 0).items()}, strict=True)
    m = m.to(dev)
    x = torch.from_numpy(make_mixture(B, SAMPLES, seed=2)).to(dev)
    dt = timed(lambda: m(x), 1, 3)
    R = cfg["n_repeats"]
    rec_flop = geo.rows * 2 * (2 * (F2 + H2) * 4 * H2)                 # one launch: both directions of one path over the whole batch
    flop = (geo.rows * R * 2 * (2 * (2 * (F2 + H2) * 4 * H2) + 2 * 2 * H2 * F2) + geo.rows * 2 * F2 * 2 * F2
            + B * 2 * geo.L * 2 * (2 * F2 * F2 + F2 * 64) + B * geo.L * 2 * 64 * F2)
    rec_names = ("dprnn_wide_lstm_f16_fwd", "dprnn_wide_rnn_fwd")
    engine.PROFILE = {"names": set(rec_names), "events": {}}
    t_ref = torch.cuda.Event(enable_timing=True)
    t_ref.record()
    try:
        with torch.no_grad():
            m(x)
        torch.cuda.synchronize(dev)
    finally:
        prof, engine.PROFILE = engine.PROFILE, None
    f16 = bool(prof["events"].get("dprnn_wide_lstm_f16_fwd"))
    kern_ms, _, _, n_ev = _launch_stats([e for n in rec_names for e in prof["events"].get(n, [])], t_ref)
    wpeak = 2500.0 / 3 if f16 else PEAK_F32_MFMA_TFLOPS
    out["tasnet_f256_h256_b64"] = {
        "workload": "DPRNN-TasNet BSS inference at feature_size = hidden_size = 256 (wider than every shipped YAML), synthetic 3 s @ 8 kHz, "
                    "batch=64, on the any-width kernels",
        "value": B / dt, "unit": "utt/s", "ms_per_step": 1e3 * dt, "steps": 3, "warmup": 1,
        "dtype": ("f32; LSTM recurrences via 2 x f16 pieces of 2^e-scaled operands (e from the measured maxima: range by construction), 3 products, "
                  "f32 accumulate; Linears / convolutions on the fp32 MFMA") if f16 else "f32 (fp32 MFMA)",
        "algorithmic_gflop_per_utterance": flop / B / 1e9,
        "roofline": {"bound": "mfma",
                     "kernel": ("wide_lstm_f16_kernel (dprnn_wide_lstm_f16_fwd: LSTM cell, 64 sequences per workgroup, x_t | h_t-1 as f16 piece pairs in LDS, "
                                "weight fragments streamed from L2)") if f16 else "wide_rnn_kernel (dprnn_wide_rnn_fwd: fp32 MFMA)",
                     "achieved": rec_flop / (kern_ms * 1e-3) / 1e12 if n_ev else None, "peak": wpeak,
                     "unit": "TFLOP/s (algorithmic = fp32-equivalent)", "frac": rec_flop / (kern_ms * 1e-3) / 1e12 / wpeak if n_ev else None,
                     "peak_note": "2500 TFLOP/s dense f16 MFMA / 3 products per algorithmic FLOP" if f16 else "fp32 MFMA peak",
                     "traffic": None, "kernel_ms": kern_ms if n_ev else None, "launches_timed": n_ev,
                     "whole_forward_tflops": flop / dt / 1e12, "whole_forward_frac_of_fp32_mfma_peak": flop / dt / 1e12 / PEAK_F32_MFMA_TFLOPS}}
    del m, x
    return out


def bench_train(args, model, dev, world, rank, dist, sync, lstm_flop_per_step, make_sources):
    """BASELINE.json configs[3]: one training step per GPU = forward with tape + PIT SI-SDR loss + backward over 64
    utterances (balanced micro-batches, trainer.train_step), ONE all-reduce of the flat 14.6 MB gradient bucket (RCCL),
    clip_grad_norm_(5) + Adam(lr 1e-3, wd 1e-5) (config_bss.yaml:30-33,53).  Weak scaling: global batch = 64 x N."""
    import torch
    from tss_with_dprnn_amd import engine, sharding, trainer, training
    B = args.batch
    model.train()
    opt = training.FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    lossm = training.PITSiSdrLoss()
    trainer.broadcast_parameters(model)
    reducer = trainer.FlatGradAllReducer(model.parameters())
    reducer.timing = True
    src, mix = make_sources(B, SAMPLES, seed=1236 + 2 * rank)
    mix_d, tgt_d = torch.from_numpy(mix).to(dev), torch.from_numpy(src).to(dev)
    for _ in range(args.warmup):
        loss, _ = trainer.train_step(model, lossm, opt, mix_d, tgt_d, clip_norm=5, reducer=reducer)
    sync()
    reducer.events.clear()
    names = {"dprnn_rnn_bwd_ex", "dprnn_rnn_bwd", "dprnn_lstm_bwd"}
    engine.PROFILE = {"names": names, "events": {}}
    t_ref = torch.cuda.Event(enable_timing=True)
    t_ref.record()
    power = PowerSampler(rank == 0 and not args.no_power, dev.index or 0)
    t0 = time.perf_counter()
    with power:
        for _ in range(args.steps):
            loss, _ = trainer.train_step(model, lossm, opt, mix_d, tgt_d, clip_norm=5, reducer=reducer)
        sync()
        elapsed = time.perf_counter() - t0
    prof, engine.PROFILE = engine.PROFILE, None
    on_device = dist is None or dist.get_backend() == "nccl"
    elapsed = sharding.max_over_ranks(elapsed, device=dev if on_device else None)
    ar_ms = [a.elapsed_time(b) for a, b in reducer.events]
    ev = [e for n in names for e in prof["events"].get(n, [])]
    kern_ms, union_ms, conc, n_ev = _launch_stats(ev, t_ref)
    achieved = lstm_flop_per_step * args.steps / (union_ms * 1e-3) / 1e12 if union_ms > 0 else float("nan")
    value = world * B * args.steps / elapsed
    assert bool(torch.isfinite(loss).all())
    from tss_with_dprnn_amd import training as _tr
    f16_bwd, f16_fwd = _tr.train_modes()[1] == 3, _tr.train_modes()[0] == 3
    # BPTT in f16 pieces is bound by its HBM streams, not by the matrix pipe: per slab row and both directions it reads the
    # tape (2 x 128 units x 5 floats; x 4 words with the gates packed to 24-bit fixed point, training.TAPE_PACKED) and dH (256) and
    # writes dG (1024) and dX (2 x 128): 11 264 bytes (10 240 packed)
    geo = engine.make_geometry(B, SAMPLES, 64, 250, 125, 2, 1)
    tape_packed = bool(_tr.TAPE_PACKED and f16_bwd and f16_fwd)
    bwd_gb_per_step = 12 * geo.rows * (10240 if tape_packed else 11264) / 1e9
    hbm = bwd_gb_per_step * args.steps / (union_ms * 1e-3) / 1e3 if union_ms > 0 else float("nan")
    traffic, traffic_src = None, None
    # the newest PMC passes of this kernel in the tape format of this run
    for cand in ("profiles/r04_pmc_traffic_train_f16_packed_tape.json", "profiles/r04_pmc_traffic_train_f16.json"):
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), cand)) as fh:
                rec = json.load(fh)["lstm_bwd8_f16_kernel"]
            if (rec.get("tape_bytes_per_row", 5120) == 4096) != tape_packed:
                continue
            traffic, traffic_src = rec["traffic_GB_per_launch"], cand
            break
        except (OSError, KeyError, ValueError):
            continue
    if f16_bwd:
        roof = {"bound": "hbm", "kernel": "lstm_bwd8_f16_kernel (dprnn_rnn_bwd_ex, BPTT, step GEMM in f16 pieces)", "achieved": hbm,
                "peak": 8.0, "unit": "TB/s", "frac": hbm / 8.0, "traffic": traffic,
                "traffic_unit": f"GB per launch of one 32-utterance micro-batch (PMC, {traffic_src}); algorithmic {bwd_gb_per_step / 24:.2f}",
                "algorithmic_gb_per_step": bwd_gb_per_step, "gate_tape": "24-bit fixed point, 4096 B per row" if tape_packed else "fp32, 5120 B per row", "matrix_tflops_algorithmic": achieved,
                "matrix_frac_of_f16_pieces_ceiling": achieved / (2500.0 / 3)}
    else:
        roof = {"bound": "mfma", "kernel": "lstm_bwd_kernel (dprnn_rnn_bwd_ex, BPTT, fp32 MFMA)", "achieved": achieved,
                "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": None}
    return {
        "metric": "utterances/sec (3 s @ 8 kHz) DPRNN-TasNet training step", "value": value, "unit": "utt/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" + ("; recurrence GEMMs via 2 x f16 pieces of 2^e-scaled operands, 3 products, f32 accumulate ("
                          + " and ".join(n for n, on in (("forward", f16_fwd), ("BPTT", f16_bwd)) if on) + ")" if (f16_fwd or f16_bwd) else ""),
        "data": "synthetic",
        "config": {"workload": "DPRNN-TasNet training step (PIT SI-SDR loss, clip 5, Adam), synthetic 3 s @ 8 kHz, "
                               f"batch={B} per GPU, global batch {B * world} (BASELINE.json configs[3])",
                   "per_gpu_batch": B, "samples": SAMPLES, "parallelism": f"data-parallel x{world}, one flat gradient all-reduce"},
        "allreduce": {"bytes": reducer.numel * 4, "ms_mean": sum(ar_ms) / len(ar_ms) if ar_ms else None,
                      "ms_max": max(ar_ms) if ar_ms else None, "calls": len(ar_ms),
                      "note": "gather copy + all-reduce + scale + scatter copy, HIP events on the step's stream"},
        "roofline": dict(roof, kernel_ms=kern_ms, launches_timed=n_ev, busy_ms_per_step=union_ms / args.steps, mean_concurrency=conc,
                         whole_step_tflops=value / world * 3 * FLOP_PER_UTT / 1e12,
                         whole_step_frac_of_fp32_mfma_peak=value / world * 3 * FLOP_PER_UTT / 1e12 / PEAK_F32_MFMA_TFLOPS),
        "power": power.summary(),
        "loss": float(loss.item()), "peak_memory_GiB": torch.cuda.max_memory_allocated(dev) / 2**30,
    }


if __name__ == "__main__":
    main()
