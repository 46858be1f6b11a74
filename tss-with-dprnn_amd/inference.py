"""Batched / multi-stream inference harness around the MI355X forward path.

The reference evaluates one full-length utterance per call (src/inferencers/inferencer.py:54-58,
inferencer_spe.py:27-32): the serial depth of 2 664 LSTM steps then bounds throughput.  Utterances are
independent, so here

* utterances of equal length are stacked into one batch (zero-padding to a common length is NOT done: it would
  change the per-utterance GroupNorm statistics, dprnn.py:76-77),
* every length bucket runs as its own pipeline on a HIP stream from a small pool, most work first, so the short
  kernels of different utterances fill the chip together,
* across GPUs the utterance list is dealt by length with sharding.shard_by_length, no collective on the path.

ROCm maps HIP streams onto 4 hardware queues by default, which caps the concurrency of single-utterance pipelines:
export GPU_MAX_HW_QUEUES=16 before the process touches the GPU when using separate_many on many short utterances
(measured on MI355X, 32 utterances of 3 s: 75 utt/s with 4 queues, 136 with 8, 193 with 16).  Do NOT set it for the
large-batch path: the 4 utterance groups of a 64-utterance batch run 8 % slower with 16 queues.

pairwise_si_sdr / pit_si_sdr evaluate the reference's metric (asteroid pairwise_neg_sisdr + PITLossWrapper
'pw_mtx', trainer.py:39, inferencer.py:30,60) on the device in fp64 (dprnn_pairwise_sisdr).
"""
from __future__ import annotations

import itertools
from collections import defaultdict
from typing import List, Sequence

import torch

import os

from ._lib import DprnnError, check, load_library


RAGGED_MIN_UTTS = 8          # fewer utterances than this do not fill the chip through one ragged launch per path
# A ragged batch is as large as memory allows: its launches then span many rounds of the 256 CUs and the half-empty last round stops
# mattering (round 5, one MI355X, 375 full-length utterances through DPRNN-Spe-IRA: batches of 64: 189 utt/s one at a time, 201 two in
# flight; 96: 212; 160-256: 215; all 375 in one batch: 223).  Bounds: utterances, and audio samples -- the slab arena of a ragged forward
# is ~7.2 KB per sample (X, X2, Z, Z2, H x 2 and the mask-head scratch at 2 slab rows of 512 B per sample), so 16 M samples (2000 s at 8 kHz)
# are ~115 GB of the 288 -- and never more than 60 % of what is free on the device.
RAGGED_MAX_UTTS = int(os.environ.get("DPRNN_RAGGED_MAX_UTTS", "512"))
RAGGED_MAX_SAMPLES = int(os.environ.get("DPRNN_RAGGED_MAX_SAMPLES", str(16_000_000)))
RAGGED_BYTES_PER_SAMPLE = 7200


def ragged_chunks(order: Sequence[int], lengths: Sequence[int], device=None) -> List[List[int]]:
    """Consecutive runs of `order` (indices into lengths, in samples), each one ragged batch."""
    cap = RAGGED_MAX_SAMPLES
    if device is not None and torch.cuda.is_available():
        # what the next batch's arena can come from: memory the driver still has, and what torch's caching allocator holds unused -- the
        # previous batch's arena above all (without it a second call on a warm process saw 100+ GB "missing" and cut its batches in three)
        free, _ = torch.cuda.mem_get_info(device)
        free += max(0, torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device))
        cap = max(1, min(cap, int(0.6 * free / RAGGED_BYTES_PER_SAMPLE)))
    chunks, cur, tot = [], [], 0
    for i in order:
        n = int(lengths[i])
        if cur and (len(cur) >= RAGGED_MAX_UTTS or tot + n > cap):
            chunks.append(cur)
            cur, tot = [], 0
        cur.append(i)
        tot += n
    if cur:
        chunks.append(cur)
    return chunks
RAGGED_PIPELINES = int(os.environ.get("DPRNN_RAGGED_PIPELINES", "2"))       # ragged batches in flight on separate streams (separate_ragged)


def separate_ragged(model, wavs: Sequence[torch.Tensor], extra_inputs=None) -> List:
    """Utterances of DIFFERENT lengths through one launch per DPRNN path (engine.dprnn_process_ragged): every output equals
    that utterance's batch-of-one forward (the reference's semantics, inferencer.py:54-58 / inferencer_spe.py:27-32)."""
    from . import engine
    outs: List = [None] * len(wavs)
    # original order: the speaker branch's BatchNorm layers (when in train mode, as in the reference's Inferencer) update their
    # running statistics utterance by utterance as the one-at-a-time loop does (for IRA the two speaker passes of an
    # utterance are not adjacent here: outputs are unaffected, the running buffers end slightly differently); the
    # recurrence launches order their workgroups by length themselves
    order = list(range(len(wavs)))
    ira = bool(getattr(model, "_ira", False))
    chunks = ragged_chunks(order, [int(w.numel()) for w in wavs], wavs[0].device)

    def forward(idx):
        if extra_inputs is None:
            return engine.tasnet_forward_ragged(model, [wavs[i] for i in idx])
        return engine.spe_forward_ragged(model, [wavs[i] for i in idx], [extra_inputs[i][0] for i in idx],
                                         [extra_inputs[i][1] for i in idx], ira=ira)
    module = getattr(model, "separation", None)
    # Several ragged batches: RAGGED_PIPELINES of them in flight on separate streams.  One ragged batch is ONE pipeline -- a launch of
    # ~640 workgroups is 2.5 rounds of the 256 CUs and the half-empty third is lost; a second pipeline's launches fill it (the same
    # effect as the two utterance groups of engine.dprnn_process: 64 x 3 s as one pipeline 800 utt/s, as two 865).  Not with BatchNorm
    # in train mode (its running statistics are updated utterance by utterance, in order), and not inside a graph capture.
    pipelined = (len(chunks) > 1 and RAGGED_PIPELINES > 1 and not getattr(engine._TLS, "capturing", False)
                 and not (module is not None and engine._bn_training(module)))
    with torch.no_grad():
        if not pipelined:
            for idx in chunks:
                for i, r in zip(idx, forward(idx)):
                    outs[i] = r
            return outs
        dev = wavs[0].device
        main = torch.cuda.current_stream(dev)
        pool = engine.bucket_streams(dev, RAGGED_PIPELINES)
        fork = torch.cuda.Event()
        fork.record(main)
        for n, idx in enumerate(chunks):
            s = pool[n % len(pool)]
            if n < len(pool):
                s.wait_event(fork)
            with torch.cuda.stream(s):
                res = forward(idx)
                for i, r in zip(idx, res):
                    outs[i] = r
                    for t in (r if isinstance(r, tuple) else (r,)):
                        t.record_stream(main)
        for s in pool:
            done = torch.cuda.Event()
            done.record(s)
            main.wait_event(done)
    return outs


def separate_many(model, wavs: Sequence[torch.Tensor], max_streams: int = 8, extra_inputs=None, ragged=None) -> List:
    """Run `model` over utterances of arbitrary (different) lengths; returns one output per utterance, in order.

    wavs: 1-D float32 device tensors.  For DPRNN-Spe / IRA models pass extra_inputs = [(aux 1-D tensor, aux_len)]
    and each output is (estimate, logits).  ragged: None = one ragged launch per path when the model and the gate-GEMM mode
    support it and there are at least RAGGED_MIN_UTTS utterances, else equal-length batches on streams; True / False force.
    The whole call is ONE saturation-guarded unit (engine._guarded): the forwards inside enqueue without synchronising, the
    flags of all of them are read once at the end, and a saturated call is repeated in gate-GEMM mode 1."""
    if len(wavs) == 0:
        return []
    dev = wavs[0].device
    if not all(w.is_cuda and w.dim() == 1 and w.dtype == torch.float32 for w in wavs):
        raise DprnnError("separate_many expects 1-D float32 tensors on a HIP device")
    from . import engine
    run = lambda: _separate_many(model, wavs, max_streams, extra_inputs, ragged)
    return engine._guarded(dev, run, engine._in_mode(1, run), "separate_many", getattr(model, "separation", None))


def separate_many_async(model, wavs: Sequence[torch.Tensor], max_streams: int = 8, extra_inputs=None, ragged=None):
    """separate_many without its final host synchronisation: everything is enqueued, and an engine.ForwardTicket comes back whose
    result() -- after the saturation flags of the whole call have arrived -- returns the list of outputs (or the range-free repeat's).
    A caller keeps two such calls in flight: enqueue bucket k + 1, then collect bucket k (inferencer.BatchedInferencer)."""
    from . import engine
    if len(wavs) == 0:
        return engine.ForwardTicket(None, [], None, None, None, "separate_many", [], [], None)
    dev = wavs[0].device
    if not all(w.is_cuda and w.dim() == 1 and w.dtype == torch.float32 for w in wavs):
        raise DprnnError("separate_many expects 1-D float32 tensors on a HIP device")
    run = lambda: _separate_many(model, wavs, max_streams, extra_inputs, ragged)
    module = getattr(model, "separation", None)
    if engine._bn_training(module) if module is not None else False:
        # train-mode BatchNorm updates its running buffers per utterance, in order, and a discarded forward must put them back before
        # the next one reads them: such models are not pipelined
        outs = engine._guarded(dev, run, engine._in_mode(1, run), "separate_many", module)
        return engine.ForwardTicket(dev, outs, None, None, None, "separate_many", [], [], None)
    _, ticket = engine._guarded_async(dev, run, engine._in_mode(1, run), "separate_many", module)
    return ticket


def _separate_many(model, wavs, max_streams, extra_inputs, ragged) -> List:
    from . import engine
    dev = wavs[0].device
    if ragged is None:
        ragged = len(wavs) >= RAGGED_MIN_UTTS and engine.ragged_supported(model.separation)
    if ragged:
        return separate_ragged(model, wavs, extra_inputs)
    buckets = defaultdict(list)
    for i, w in enumerate(wavs):
        key = (w.numel(),) if extra_inputs is None else (w.numel(), extra_inputs[i][0].numel(), float(extra_inputs[i][1]))
        buckets[key].append(i)
    order = sorted(buckets.items(), key=lambda kv: -kv[0][0] * len(kv[1]))     # most work first
    main = torch.cuda.current_stream(dev)
    pool = engine.bucket_streams(dev, max(1, min(max_streams, len(order))))       # the same streams on every call
    fork = torch.cuda.Event()
    fork.record(main)
    outs: List = [None] * len(wavs)
    with torch.no_grad():
        for n, (key, idxs) in enumerate(order):
            s = pool[n % len(pool)]
            s.wait_event(fork)
            with torch.cuda.stream(s):
                x = torch.stack([wavs[i] for i in idxs])
                if extra_inputs is None:
                    y = model(x)
                    parts = [y[j] for j in range(len(idxs))]
                else:
                    aux = torch.stack([extra_inputs[i][0] for i in idxs])
                    est, logits = model(x, aux, torch.tensor(key[2]))
                    parts = [(est[j], logits[j]) for j in range(len(idxs))]
                for j, i in enumerate(idxs):
                    outs[i] = parts[j]
                    for t in (parts[j] if isinstance(parts[j], tuple) else (parts[j],)):
                        t.record_stream(main)
    for s in pool:
        done = torch.cuda.Event()
        done.record(s)
        main.wait_event(done)
    return outs


def pairwise_si_sdr(est: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
    """est [B, n_est, T], tgt [B, n_tgt, T] (float32, device) -> [B, n_est, n_tgt] float64 SI-SDR in dB."""
    if not (est.is_cuda and tgt.is_cuda):
        raise DprnnError("pairwise_si_sdr runs on the HIP device only")
    est, tgt = est.contiguous().float(), tgt.contiguous().float()
    B, ne, T = est.shape
    nt = tgt.shape[1]
    if tgt.shape[0] != B or tgt.shape[2] != T:
        raise DprnnError("estimate / target shapes differ")
    out = torch.empty(B, ne, nt, device=est.device, dtype=torch.float64)
    lib = load_library()
    rc = lib.dprnn_pairwise_sisdr(est.data_ptr(), tgt.data_ptr(), out.data_ptr(), B, ne, nt, T,
                                  torch.cuda.current_stream(est.device).cuda_stream)
    check(lib, rc, "dprnn_pairwise_sisdr")
    return out


def pit_si_sdr(est: torch.Tensor, tgt: torch.Tensor):
    """Permutation-invariant mean SI-SDR (PITLossWrapper pit_from='pw_mtx' semantics, maximising SI-SDR):
    returns (best mean SI-SDR per item [B] float64, best permutation [B, n_src] = target of estimate i,
    estimates reordered to the targets' order)."""
    pw = pairwise_si_sdr(est, tgt)                       # [B, est, tgt]
    n = pw.shape[1]
    perms = torch.tensor(list(itertools.permutations(range(n))), device=pw.device)      # [n!, n]
    ar = torch.arange(n, device=pw.device)
    scores = torch.stack([pw[:, ar, p].mean(dim=1) for p in perms], dim=1)              # [B, n!]
    best = scores.argmax(dim=1)
    perm = perms[best]                                   # [B, n]
    inv = torch.argsort(perm, dim=1)                     # estimate index for each target
    reordered = torch.gather(est, 1, inv.unsqueeze(-1).expand(-1, -1, est.shape[-1]))
    return scores.gather(1, best[:, None])[:, 0], perm, reordered
