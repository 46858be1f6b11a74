#!/usr/bin/env python3
"""Headline benchmark: DPRNN-TasNet forward throughput (utterances/sec, 3 s @ 8 kHz, batch 64 per GPU).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one forward of the HIP path over one batch of 64 synthetic 3-second mixtures already
resident in HBM (BASELINE.json configs[1]).  Utterances are independent, so ranks shard the batch with
no data-path collective (weak scaling: 64 utterances per GPU).  Rank 0 prints ONE JSON line with
  roofline     : the dominant kernel (fused BiLSTM recurrence, dprnn_bilstm_fwd) timed with HIP events
                 on its launch stream inside the timed region; achieved = algorithmic FLOPs per launch
                 (rows * 2 dirs * 2*256*512) / mean launch time; peak = 157.3 TFLOP/s fp32 MFMA.
  cpu_baseline : the reference's CPU path (oracle/dprnn_torch_cpu.py: the same ATen CPU ops the
                 reference dispatches to) timed on this host's cores on a bounded sample (N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SAMPLES = 24000          # 3 s @ 8 kHz
PER_GPU_BATCH = 64       # BASELINE.json configs[1]
FLOP_PER_UTT = 350.80e9
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
CTOR = ["input_size", "feature_size", "hidden_size", "chunk_length", "kernel_size", "hop_length", "n_repeats",
        "bidirectional", "rnn_type", "norm_type", "activation_type", "dropout", "stride"]


def cpu_baseline(sd, cfg, threads, model, dev, batch=4, iters=3):
    """The CPU path timed beside the GPU one (the checker's leg of the bench: the only place bench.py touches oracle/),
    and -- the second half of BASELINE.json's metric, "SI-SDR vs ref" -- the parity of the GPU output with the CPU output
    on the same `batch` two-source mixtures (SURVEY.md 8d recipe)."""
    from oracle import dprnn_torch_cpu as otc
    from oracle.sisdr import pit_si_sdr, si_sdr
    from oracle.weights import make_sources
    torch.set_num_threads(threads)
    tsd = otc.to_torch_state(sd)
    src, mix = make_sources(batch, SAMPLES)
    x = torch.from_numpy(mix)
    otc.tasnet_forward(tsd, cfg, x[:1])                      # warm-up
    t0 = time.perf_counter()
    for _ in range(iters):
        cpu_out = otc.tasnet_forward(tsd, cfg, x)
    dt = time.perf_counter() - t0
    base = dict(value=batch * iters / dt, unit="utt/s", cores=threads, kind="port",
                sample=f"{iters} forwards of batch {batch} x 3 s @ 8 kHz, oracle/dprnn_torch_cpu.py "
                       f"(stock torch CPU ops), {threads} threads of {os.cpu_count()} logical CPUs, {dt:.1f} s")
    with torch.no_grad():
        gpu_out = model(x.to(dev)).cpu().numpy().astype(np.float64)
    cpu_out = cpu_out.numpy().astype(np.float64)
    T = min(gpu_out.shape[-1], src.shape[-1])
    s_gpu, _ = pit_si_sdr(gpu_out[..., :T], src[..., :T].astype(np.float64))      # best-permutation SI-SDR per utterance, dB
    s_cpu, _ = pit_si_sdr(cpu_out[..., :T], src[..., :T].astype(np.float64))
    parity = dict(sample=f"{batch} synthetic two-source mixtures x 3 s, GPU output vs the CPU path's output",
                  rel_l2=float(np.linalg.norm(gpu_out - cpu_out) / np.linalg.norm(cpu_out)),
                  si_sdr_gpu_vs_cpu_db=float(np.min(si_sdr(gpu_out.reshape(-1, gpu_out.shape[-1]), cpu_out.reshape(-1, cpu_out.shape[-1])))),
                  max_abs_delta_si_sdr_db=float(np.max(np.abs(np.asarray(s_gpu) - np.asarray(s_cpu)))),
                  tolerance_db=0.01)
    return base, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="utterances per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-optional-modes", action="store_true", help="skip the extra timing of the opt-in gate-GEMM modes")
    ap.add_argument("--fast-act", type=int, default=1)
    ap.add_argument("--groups", type=int, default=None, help="utterance groups on separate HIP streams")
    ap.add_argument("--gemm-mode", type=int, default=0,
                    help="0 exact fp32 MFMA (headline); 1/2 = optional bf16-split gate GEMM with 6/3 products")
    ap.add_argument("--min-group", type=int, default=None, help="smallest utterance group (engine.MIN_GROUP_BATCH)")
    ap.add_argument("--group-sizes", type=str, default="", help="experiment: explicit utterance-group sizes, e.g. 20,18,14,12")
    ap.add_argument("--seq-tile-intra", type=int, default=0)
    ap.add_argument("--seq-tile-inter", type=int, default=0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # rehearsal knobs (a one-GPU box cannot host two RCCL ranks): DPRNN_BENCH_BACKEND=gloo DPRNN_BENCH_SHARE_GPU=1
    backend = os.environ.get("DPRNN_BENCH_BACKEND", "nccl")
    if os.environ.get("DPRNN_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)      # RCCL over xGMI
        else:
            dist.init_process_group(backend=backend)

    import tss_with_dprnn_amd as pkg
    from tss_with_dprnn_amd import engine
    from oracle.weights import full_cfg, make_mixture, make_state

    cfg = full_cfg()
    model = pkg.DPRNNTasNet(**{k: cfg[k] for k in CTOR}).eval()
    sd = make_state("bss", cfg, 0)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model = model.to(dev)
    lib = pkg.load_library()
    lib.dprnn_set_fast_activations(args.fast_act)
    lib.dprnn_set_fast_head_activations(args.fast_act)
    engine.SEQ_TILE["intra"], engine.SEQ_TILE["inter"] = args.seq_tile_intra, args.seq_tile_inter
    if args.groups is not None:
        engine.STREAM_GROUPS = args.groups
    if args.min_group is not None:
        engine.MIN_GROUP_BATCH = args.min_group
    engine.GEMM_MODE = args.gemm_mode
    if args.group_sizes:
        engine.GROUP_SIZES = [int(v) for v in args.group_sizes.split(",")]

    B = args.batch
    x = torch.from_numpy(make_mixture(B, SAMPLES, seed=1234 + rank)).to(dev)     # resident in HBM before timing

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    with torch.no_grad():
        for _ in range(args.warmup):
            out = model(x)
        sync()
        engine.PROFILE = {"names": {"dprnn_rnn_fwd"}, "events": {}}
        t_ref = torch.cuda.Event(enable_timing=True)
        t_ref.record()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = model(x)
        sync()
        elapsed = time.perf_counter() - t0
        prof, engine.PROFILE = engine.PROFILE, None
    assert out.shape == (B, 2, SAMPLES) and bool(torch.isfinite(out).all())

    from tss_with_dprnn_amd import sharding
    elapsed = sharding.max_over_ranks(elapsed, device=dev if backend == "nccl" else None)   # slowest rank defines the step

    # dominant kernel: per-launch HIP-event durations on the launch streams.  With utterance groups on several
    # streams the launches overlap, so the chip-level rate is (sum of algorithmic FLOPs) / (union of the launch
    # intervals); kernel_ms is the plain mean launch duration (what rocprofv3 --stats reports).
    ev = prof["events"].get("dprnn_rnn_fwd", [])
    spans = sorted((t_ref.elapsed_time(a), t_ref.elapsed_time(b)) for a, b in ev)
    kern_ms = float(np.mean([e - s for s, e in spans])) if spans else float("nan")
    union_ms, cur_s, cur_e = 0.0, None, None
    for s_, e_ in spans:
        if cur_e is None or s_ > cur_e:
            union_ms += (cur_e - cur_s) if cur_e is not None else 0.0
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    union_ms += (cur_e - cur_s) if cur_e is not None else 0.0
    geo = engine.make_geometry(B, SAMPLES, cfg["input_size"], cfg["chunk_length"], cfg["hop_length"],
                               cfg["kernel_size"], cfg["stride"])
    n_groups = len(engine._split(B))
    launches_per_step = 2 * cfg["n_repeats"] * n_groups
    flop_per_launch = geo.rows / n_groups * 2 * (2 * 256 * 512)          # rows * 2 dirs * (2*256*512)
    total_flop = geo.rows * 2 * (2 * 256 * 512) * 2 * cfg["n_repeats"] * args.steps
    achieved = total_flop / (union_ms * 1e-3) / 1e12 if union_ms > 0 else float("nan")
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            pm = json.load(f)["bilstm_kernel"]
        if pm["group_batch"] * n_groups == B:
            traffic = pm["traffic_GB_per_launch"]
    except (OSError, KeyError, ValueError):
        pass

    # Optional arithmetic modes of the gate GEMM, timed AFTER (and outside) the headline region: same batch, same
    # protocol, 3 steps each; deviation is measured against the exact-fp32 output of this very run.
    optional = None
    if world == 1 and args.gemm_mode == 0 and not args.no_optional_modes:
        optional = {}
        ref_out = out.double()
        with torch.no_grad():
            for mode, name in ((1, "bf16_split_6_products"), (2, "bf16_split_3_products")):
                engine.GEMM_MODE = mode
                o = model(x)
                sync()
                t1 = time.perf_counter()
                for _ in range(3):
                    o = model(x)
                sync()
                dt = (time.perf_counter() - t1) / 3
                optional[name] = {"value": B / dt, "unit": "utt/s", "ms_per_step": 1e3 * dt,
                                  "rel_l2_vs_exact_fp32_path": float((o.double() - ref_out).norm() / ref_out.norm())}
            engine.GEMM_MODE = 0
    if rank == 0:
        value = world * B * args.steps / elapsed
        line = {
            "metric": "utterances/sec (3 s @ 8 kHz) DPRNN-TasNet fwd", "value": value, "unit": "utt/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.gemm_mode == 0 else "f32 via bf16-split operands, f32 accumulate",
            "data": "synthetic",
            "config": {"workload": "DPRNN-TasNet BSS inference, synthetic 3 s @ 8 kHz, batch=64 per GPU "
                                   "(BASELINE.json configs[1])",
                       "per_gpu_batch": B, "samples": SAMPLES, "parallelism": f"utterance-sharded x{world}",
                       "fast_activations": bool(args.fast_act), "stream_groups": engine.STREAM_GROUPS,
                       "gemm_mode": ["exact fp32 MFMA", "bf16-split x6 on bf16 MFMA (opt-in)", "bf16-split x3 on bf16 MFMA (opt-in)"][args.gemm_mode]},
            "roofline": {"bound": "mfma", "kernel": "bilstm_kernel (dprnn_rnn_fwd, LSTM cell)",
                         "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic, "traffic_unit": "GB per launch (PMC, profiles/r01_pmc_traffic.json)",
                         "algorithmic_gflop_per_launch": flop_per_launch / 1e9,
                         # slab units per launch: read X per direction (2) + write H (2) [+ read Z per direction (2) + write x' (1) when
                         # the previous norm is folded in: all launches of a forward but the first]
                         "algorithmic_gb_per_launch": geo.rows / n_groups * 128 * 4 *
                         ((4 + 7 * (2 * cfg["n_repeats"] - 1)) / (2 * cfg["n_repeats"]) if engine.FUSE_NORM else 4) / 1e9,
                         "kernel_ms": kern_ms, "launches_timed": len(ev), "launches_per_step": launches_per_step,
                         "concurrent_streams": n_groups, "busy_ms_per_step": union_ms / args.steps,
                         # achieved = per_launch_tflops x mean_concurrency: the launches of the four pipelines overlap
                         "per_launch_tflops": flop_per_launch / (kern_ms * 1e-3) / 1e12,
                         "mean_concurrency": sum(e - s_ for s_, e in spans) / union_ms if union_ms > 0 else None,
                         "whole_forward_frac": value / world * FLOP_PER_UTT / 1e12 / PEAK_F32_MFMA_TFLOPS},
        }
        if optional is not None:
            line["optional_gemm_modes"] = dict(optional, note="opt-in (engine.GEMM_MODE / dprnn_set_gemm_mode): LSTM gate GEMM with "
                                               "bf16-split operands on the bf16 MFMA, fp32 accumulate; NOT part of `value`")
        if world == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            line["cpu_baseline"], line["parity"] = cpu_baseline(sd, cfg, threads, model, dev)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
