"""Host-side orchestration of the MI355X DPRNN forward path.

Python here only sizes buffers, reads live parameter pointers and enqueues the HIP kernels of
libdprnn_mi355x.so (include/dprnn_mi355x.h) on torch's current stream; all arithmetic happens in
those kernels.  Stage order follows the reference forward:
  DPRNNTasNet.forward   /root/reference/src/models/dprnn.py:271-283
  DPRNN.forward         dprnn.py:166-187   (bottleneck, segmentation, blocks, mask head)
  DPRNNBlock.forward    dprnn.py:79-99
"""
from __future__ import annotations

import collections
import contextlib
import ctypes
import functools
import os
import threading
import warnings
from dataclasses import dataclass
from typing import Optional

import torch

from ._lib import SAT_NAMES, SAT_SLOTS, SCALE_SLOTS, DprnnError, check, load_library

F_BUILT = 128      # feature_size / hidden_size the kernels are built for (config_bss.yaml:19-20)
N_BUILT = 64       # input_size (config_bss.yaml:18)
KS_MAX = 8
PACK_PER_DIR = 4 * 16 * 8 * 256            # fp32 fragments of one direction (floats)
PACK_PER_DIR_MAX = 4 * 8 * 8 * 3 * 64 * 4   # the split modes' piece planes (u32 words; three planes of bf16 pieces, two of f16)
H_PIECES = os.environ.get("DPRNN_H_PIECES", "1") != "0"     # mode 3: the recurrence writes h as the projection's f16 piece operand
CELL_ID = {"LSTM": 0, "GRU": 1, "RNN": 2}       # dprnn_rnn_pack / dprnn_rnn_fwd cell codes

# process-wide tuning knob: sequences per workgroup in the recurrence kernel (0 = library default)
SEQ_TILE = {"intra": 0, "inter": 0, "serial_intra": 0, "serial_inter": 0}

# utterance groups run on separate HIP streams (0/1 = everything on the current stream).  Utterances are
# independent, so each group is a complete pipeline of its own: while one group's recurrence kernel drains
# its last workgroups (grid quantisation) or runs a bandwidth-bound stage, the other group's kernels fill
# the idle CUs.
STREAM_GROUPS = 4
MIN_GROUP_BATCH = 16         # utterances per group, about (much smaller groups lose more per kernel than overlap wins)

# Fold every GroupNorm + residual into the loader of the stage that consumes it (the next recurrence kernel, or the
# mask head for the last one) instead of a separate pass over the slab.  Off => explicit dprnn_norm_residual_fwd.
FUSE_NORM = True

# Arithmetic mode of the contractions (include/dprnn_mi355x.h: dprnn_set_gemm_mode / the *_ex entry points).
#   3 (default) = every fp32 operand as a sum of TWO f16 pieces of a power-of-two multiple of it (activations x 2^4, weights x
#                 a per-matrix 2^e that puts max|W| in [2^13, 2^14)): 22+ operand bits, three cross products on the f16 MFMA
#                 with fp32 accumulation, the dropped product v2 w2 ~ 2^-24 |v||w| -- one fp32 product rounding
#                 (recur_split.h; tests/studies/split_precision_study.py) -- in the recurrence, the Linear after it, the bottleneck and the
#                 mask head.  f16 has 5 exponent bits: |x| > 4062 does not fit 2^4 x.  Every conversion site raises a device flag
#                 when that happens (dprnn_saturation_begin / _flags); every forward checks the flags before it returns and, if
#                 one is set, runs again in mode 1 (_guarded below) -- a clamped value never reaches the caller;
#   1           = three bf16 pieces, the six cross products of order <= 2^-16 on the bf16 MFMA (product error ~2^-24, no
#                 range assumption at all: bf16 has fp32's exponent); from SPLIT_MIN_BATCH utterances up (below: mode 0);
#   0           = exact fp32 operands on the fp32 MFMA (recur.hip): bit-for-bit an fma chain; DPRNN_GEMM_MODE=0 selects it.
# GRU / Elman cells always run the exact fp32 kernels (training: training.TRAIN_GEMM_MODE / TRAIN_BWD_GEMM_MODE).
GEMM_MODES = (0, 1, 3)
DEFAULT_GEMM_MODE = int(os.environ.get("DPRNN_GEMM_MODE", "3"))
if DEFAULT_GEMM_MODE not in GEMM_MODES:
    raise DprnnError(f"DPRNN_GEMM_MODE={DEFAULT_GEMM_MODE}: the gate-GEMM modes are 0 (exact fp32 MFMA), 1 (three bf16 pieces) "
                     "and 3 (two f16 pieces, the default)")
GEMM_MODE = DEFAULT_GEMM_MODE
# DPRNNTasNet inference of a model narrower than the build: feature / hidden sizes up to this run natively on the any-width kernels (exact
# fp32 MFMA arithmetic, work proportional to the real widths) instead of as a zero-padded twin on the 128-wide fused kernels.
NARROW_NATIVE_MAX = int(os.environ.get("DPRNN_NARROW_NATIVE_MAX", "0"))
SPLIT_ONE_GROUP = True      # split modes: SPLIT_GROUPS pipelines instead of STREAM_GROUPS (the wide kernels take one workgroup per CU)

# What a forward does when the f16-pieces arithmetic met a value beyond its range: "fallback" = run the call again in mode 1
# (default), "raise" = DprnnError.  SATURATION counts the checks, the fall-backs and keeps the flags of the last one.
ON_SATURATION = os.environ.get("DPRNN_ON_SATURATION", "fallback")
# Diagnostics only (A/B timing of the check itself): DPRNN_SATURATION_CHECK=0 turns the detection off -- values beyond the range
# would then be clamped silently, so no product path sets it.
SATURATION_CHECK = os.environ.get("DPRNN_SATURATION_CHECK", "1") != "0"
SATURATION = {"checks": 0, "fallbacks": 0, "last_flags": None, "last_what": None}
_STATS_LOCK = threading.Lock()       # SATURATION / GRAPH_STATS are process-wide counters updated from any thread
_TLS = threading.local()


# ---- adaptive activation scale of the f16-pieces arithmetic (include/dprnn_mi355x.h: dprnn_scale_*; csrc/common.h).  One device
# record per (model, device): the largest |x| every conversion site has met; the kernels of a forward read their scale 2^s from it
# and leave their maxima in it, all on the device (no host round trip, and the same addresses on every replay of a captured
# forward).  A model whose residual stream leaves the default range (|x| > 4062) is therefore repeated in mode 1 ONCE; its next
# forwards run in mode 3 under a scale that fits.  DPRNN_ADAPTIVE_SCALE=0: the constant 2^4 of rounds 2 and 3.
ADAPTIVE_SCALE = os.environ.get("DPRNN_ADAPTIVE_SCALE", "1") != "0"
SLOT_BOTTLENECK, SLOT_AUX_BOTTLENECK, SLOT_HEAD_CONV, SLOT_HEAD_GATE, SLOT_SPK, SLOT_PATH0 = 0, 1, 2, 3, 4, 5
SAT_NONFINITE = 5              # DPRNN_SAT_NONFINITE (csrc/common.h)
PIN_AFTER_FALLBACKS = 2          # consecutive saturated forwards after which a model stays in mode 1


# optional per-kernel timing (bench.py): {"names": set of C-ABI entry points, "events": {name: [(start, end)]}}
# HIP events are recorded on the stream the kernels are launched on (torch's current stream).
PROFILE = None


@dataclass
class Geometry:
    B: int
    T: int          # input samples
    L: int          # encoder frames
    K: int
    P: int
    S: int
    ks: int
    stride: int
    N: int

    @property
    def rows(self) -> int:          # rows of the [B][S][K] slab
        return self.B * self.S * self.K

    @property
    def T_out(self) -> int:         # ConvTranspose1d output length (encoder_decoder.py:40-42)
        return (self.L - 1) * self.stride + self.ks


def make_geometry(B: int, T: int, N: int, K: int, P: int, ks: int, stride: int) -> Geometry:
    if T < ks:
        raise DprnnError(f"input of {T} samples is shorter than the encoder kernel ({ks})")
    L = (T - ks) // stride + 1                       # nn.Conv1d output length
    S = (L + K) // P + 1                             # unfold with padding K (dprnn.py:192-198)
    return Geometry(B, T, L, K, P, S, ks, stride, N)


WS_PARTS = ("Y", "X", "X2", "Z", "H", "stats", "packed", "bsum", "affine", "taps", "out")     # DPRNN_WS_* of the C header


def workspace_parts(geo: Geometry, n_blocks: int, split_gemm: bool = False) -> dict:
    """Bytes of every device buffer of one forward, from the library itself (dprnn_workspace_bytes): the arenas below are
    sized with it, so a C host calling the same function allocates exactly what this host does."""
    import ctypes
    lib = load_library()
    parts = (ctypes.c_int64 * len(WS_PARTS))()
    total = lib.dprnn_workspace_bytes(geo.B, geo.T, geo.N, geo.K, geo.P, geo.ks, geo.stride, n_blocks, int(split_gemm), parts)
    if total < 0:
        check(lib, int(total), "dprnn_workspace_bytes")
    out = dict(zip(WS_PARTS, (int(v) for v in parts)))
    out["total"] = int(total)
    return out


def _p(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


def _param(t: torch.Tensor, name: str, device) -> torch.Tensor:
    if t.device != device:
        raise DprnnError(f"parameter {name} lives on {t.device}, input on {device}: call model.to(device)")
    if t.dtype != torch.float32:
        raise DprnnError(f"parameter {name} is {t.dtype}; the MI355X path computes in float32")
    if not t.is_contiguous():
        raise DprnnError(f"parameter {name} is not contiguous")
    return t


def check_supported(cfg: dict) -> None:
    """Fail loudly for constructor settings the HIP kernels are not built for."""
    bad = []
    if cfg["feature_size"] != F_BUILT or cfg["hidden_size"] != F_BUILT:
        bad.append(f"feature_size/hidden_size must be {F_BUILT} here (got {cfg['feature_size']}/{cfg['hidden_size']}; narrower models run "
                   "through a zero-padded twin, padded.py)")
    if cfg["input_size"] != N_BUILT:
        bad.append(f"input_size must be {N_BUILT} here (got {cfg['input_size']}; narrower: zero-padded twin, padded.py)")
    if cfg["rnn_type"] not in CELL_ID:
        bad.append(f"rnn_type {cfg['rnn_type']!r}: torch.nn has LSTM, GRU and RNN recurrences")
    if cfg["kernel_size"] > KS_MAX:
        bad.append(f"kernel_size must be <= {KS_MAX}")
    if cfg["norm_type"] not in ("ln", "gLN"):
        bad.append(f"norm_type {cfg['norm_type']!r} (reference leaves the norms undefined, dprnn.py:72-77)")
    if cfg["activation_type"] not in ("sigmoid", "relu"):
        bad.append(f"activation_type {cfg['activation_type']!r} (dprnn.py:161-164)")
    # dropout: any value in [0, 1] is accepted and has no effect, exactly as in the reference, whose only use of it is the
    # num_layers=1 torch RNN constructor (dprnn.py:23-28; the mirror's SingleRNN raises / warns as torch does)
    if cfg["hop_length"] < 1 or cfg["hop_length"] > cfg["chunk_length"]:
        bad.append("hop_length must be in [1, chunk_length]")
    if bad:
        raise NotImplementedError("DPRNN MI355X path: unsupported configuration: " + "; ".join(bad))


def _count_scale(stats_slot: torch.Tensor, norm, st: "Stage", n: int) -> None:
    """Zero-padded twins of narrow models (padded.py): the kernels divide a global norm's sums by rows x BUILT channels; scaling
    (sum, sum of squares) by built / real channels makes that rows x REAL channels for the mean and the second moment alike.
    The factor is applied ON THE STREAM THAT PRODUCED THE SUMS (an utterance group's side stream: torch's current stream is
    neither ordered after that group's projection nor before its next reader) and to that launch's n utterances only (the slot
    handed to a group runs on to the end of the batch)."""
    f = norm.__dict__.get("_dprnn_count_scale", 1.0)
    if f != 1.0:
        with torch.cuda.stream(st.tstream):
            stats_slot[:n].mul_(f)


def _norm_params(norm: torch.nn.Module):
    if hasattr(norm, "gamma"):                      # norms.GlobLN (norms.py:20-21), eps 1e-8
        return norm.gamma, norm.beta, 1e-8
    return norm.weight, norm.bias, 1e-5             # nn.GroupNorm(1, C), eps 1e-5


class Stage:
    """Thin launcher: stream + error checking around the C ABI."""

    def __init__(self, device, stream: Optional[torch.cuda.Stream] = None):
        self.lib = load_library()
        self.device = device
        self.tstream = stream if stream is not None else torch.cuda.current_stream(device)
        self.stream = self.tstream.cuda_stream

    def call(self, name: str, *args):
        prof = PROFILE
        if prof is not None and name in prof["names"]:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(self.tstream)
            rc = getattr(self.lib, name)(*args, self.stream)
            e1.record(self.tstream)
            prof["events"].setdefault(name, []).append((e0, e1))
        else:
            rc = getattr(self.lib, name)(*args, self.stream)
        check(self.lib, rc, name)


def bottleneck_affine(st: Stage, wav, enc_w, bott_norm, geo: Geometry, stats_slot, mul=None, add=None):
    """Per-utterance affine (sc, sh) of the bottleneck norm, with the add/mul/film fusions folded in.
    reference: encoder_decoder.py:25-33 (stats of the encoder output), dprnn.py:131-136, dprnn_spe.py:194-229."""
    dev = wav.device
    gamma, beta, eps = _norm_params(bott_norm)
    st.call("dprnn_encoder_stats", _p(wav), _p(enc_w), _p(stats_slot), geo.B, geo.T, geo.L, geo.N, geo.ks, geo.stride)
    _count_scale(stats_slot, bott_norm, st, geo.B)
    sc = torch.empty(geo.B, geo.N, device=dev, dtype=torch.float32)
    sh = torch.empty_like(sc)
    st.call("dprnn_bottleneck_prep", _p(stats_slot), _p(_param(gamma, "norm.weight", dev)),
            _p(_param(beta, "norm.bias", dev)), eps, _p(mul), _p(add), _p(sc), _p(sh), geo.B, geo.L, geo.N)
    return sc, sh


def bottleneck_frames(st: Stage, wav, enc_w, sc, sh, conv, geo: Geometry, row_bias=None, row_scale=None, gemm_mode=None,
                      scale_slot: int = SLOT_BOTTLENECK, min_floats: int = 0):
    """1x1 conv of the normalised encoder output -> Y [B][L][F] (dprnn.py:135-136,169; 'cat' fusion as row_bias,
    'att' fusion as row_scale).  gemm_mode: arithmetic of the product (3: f16 pieces like the Linears and the mask head, else the
    fp32 MFMA); default GEMM_MODE's.  This call comes before dprnn_process selects the forward's mode, so it selects its own and
    puts the library's back."""
    dev = wav.device
    w = _param(conv.weight, "conv.weight", dev)
    if w.shape[0] != F_BUILT or w.shape[1] < geo.N:
        raise DprnnError(f"1x1 conv weight {tuple(w.shape)} does not fit the {geo.N}->{F_BUILT} kernel")
    # min_floats: the block is sized so that, once the frames are segmented and released, a group's slab fits into it
    nY = geo.B * geo.L * F_BUILT
    Y = torch.empty(max(nY, min_floats), device=dev, dtype=torch.float32)[:nY].view(geo.B, geo.L, F_BUILT)
    mode = current_mode() if gemm_mode is None else gemm_mode
    prev = st.lib.dprnn_get_gemm_mode()
    st.lib.dprnn_set_gemm_mode(3 if mode == 3 else 0)
    _set_slot(st.lib, scale_slot)
    try:
        if row_scale is not None:
            st.call("dprnn_bottleneck_att_fwd", _p(wav), _p(enc_w), _p(sc), _p(sh), _p(w), w.shape[1],
                    _p(_param(conv.bias, "conv.bias", dev)), _p(row_scale), _p(Y),
                    geo.B, geo.T, geo.L, geo.N, geo.ks, geo.stride)
        else:
            st.call("dprnn_bottleneck_fwd", _p(wav), _p(enc_w), _p(sc), _p(sh), _p(w), w.shape[1],
                    _p(_param(conv.bias, "conv.bias", dev)), _p(row_bias), _p(Y),
                    geo.B, geo.T, geo.L, geo.N, geo.ks, geo.stride)
    finally:
        st.lib.dprnn_set_gemm_mode(prev)
    return Y


def frames_from_waveform(st: Stage, wav, enc_w, bott_norm, bott_conv, geo: Geometry, stats_slot,
                         mul=None, add=None, row_bias=None, min_floats: int = 0):
    """Encoder -> bottleneck norm -> (fusion) -> bottleneck 1x1 conv, output Y [B][L][F]."""
    sc, sh = bottleneck_affine(st, wav, enc_w, bott_norm, geo, stats_slot, mul, add)
    return bottleneck_frames(st, wav, enc_w, sc, sh, bott_conv, geo, row_bias, min_floats=min_floats)


# Side streams of the utterance groups: the SAME streams for every forward of a (thread, device).  torch.cuda.Stream() hands out
# the 32 streams of its per-device pool round-robin, and HIP maps streams onto a few hardware queues: taking fresh streams per forward made
# every 16th forward of the bench (the position of the cycle whose two streams collide on a queue) 5 ms slower -- the stall that
# round 2 could not place (profiles/r03_host_stall.txt).  Pools are created under a lock (forwards may come from several threads).
_SIDE_POOL: dict = {}
_STREAM_LOCK = threading.Lock()
# tools/probes/stream_pair_timing.py (round 3, one box): of the 16 pairs of torch's pool exactly one -- the first two streams it hands
# out in a process -- made every EAGER forward 5 ms slower.  DPRNN_SKIP_FIRST_STREAM_PAIR=1 takes those two and leaves them unused
# (only meaningful when nobody took streams before the engine; box- and torch-version-specific, hence opt-in).  The default path
# replays a captured graph, whose launch topology does not depend on the stream pool.
SKIP_FIRST_STREAM_PAIR = os.environ.get("DPRNN_SKIP_FIRST_STREAM_PAIR", "0") == "1"


def _persistent_streams(pools: dict, key, dev, n: int):
    with _STREAM_LOCK:
        pool = pools.get(key)
        if pool is None:
            if SKIP_FIRST_STREAM_PAIR and not _SIDE_POOL and "_skipped" not in _SIDE_POOL:
                _SIDE_POOL["_skipped"] = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
            pool = pools.setdefault(key, [])
        while len(pool) < n:
            pool.append(torch.cuda.Stream(device=dev))
        return pool[:n]


def _side_streams(dev, n: int):
    return _persistent_streams(_SIDE_POOL, (threading.get_ident(), str(dev)), dev, n)


_BUCKET_POOL: dict = {}


def bucket_streams(dev, n: int):
    """Persistent streams for the per-bucket pipelines of inference.separate_many (fresh streams per call walk torch's pool and
    change the queue mapping from call to call)."""
    return _persistent_streams(_BUCKET_POOL, (threading.get_ident(), str(dev)), dev, n)


def _rnn_tensors(rnn, dev):
    r = rnn.rnn
    ndir = 2 if r.bidirectional else 1
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    fw = [_param(getattr(r, n), n, dev) for n in names]
    bw = [_param(getattr(r, n + "_reverse"), n + "_reverse", dev) for n in names] if ndir == 2 else [None] * 4
    return ndir, fw, bw


class _Group:
    """One utterance group [b0, b1): its own stream, slab arena and geometry."""

    def __init__(self, st: Stage, geo: Geometry, b0: int, b1: int, dev):
        self.st, self.b0, self.b1 = st, b0, b1
        self.geo = Geometry(b1 - b0, geo.T, geo.L, geo.K, geo.P, geo.S, geo.ks, geo.stride, geo.N)
        # X [rows][F], X2 [rows][F], Z [rows][F] | H [rows][2F]; U and V (mask head) reuse the H region.
        # X / X2 ping-pong when the norm is folded into the next recurrence (both directions read the old slab).
        # Allocated in two steps: X before the segmentation (its target), the rest after the bottleneck frames have been released --
        # the first group's X2 then takes the frames' block (bottleneck_frames sizes it for that), which keeps the peak of a forward,
        # and with it the pool of a captured one, one slab smaller.
        ws = workspace_parts(self.geo, 0)
        self._n, self._h, self._dev = ws["X"] // 4, ws["H"] // 4, dev
        assert self._n == self.geo.rows * F_BUILT and ws["X2"] == ws["Z"] == ws["X"]
        self.X = torch.empty(self._n, device=dev, dtype=torch.float32)
        self.X2 = self.Z = self.H = None
        self.buffers = [self.X]
        self.pending = None           # (stats_slot, gamma, beta, eps) of a norm not yet applied to X
        self.n_concurrent = 1         # utterance groups in flight together (set by dprnn_process)
        self.mode = 0                 # gate-GEMM mode of this forward (set by dprnn_process; the packed weights are in it)

    def alloc_rest(self):
        self.X2 = torch.empty(self._n, device=self._dev, dtype=torch.float32)
        zh = torch.empty(self._n + self._h, device=self._dev, dtype=torch.float32)
        self.Z, self.H = zh[: self._n], zh[self._n:]
        self.buffers += [self.X2, zh]


# LIN_FOLD (round 5): the Linear after a bidirectional LSTM folded into the recurrence (mode 3; csrc/recur_split8_f16_lin.hip): every h_t
# is multiplied with its direction's half of the Linear while it is the next step's operand in LDS, so h never reaches HBM and the
# projection kernel disappears -- 7 slab rows of HBM traffic per row and path instead of 10 (profiles/r05_pmc_traffic_f16.json: 7.49
# measured against 10.43).  The two halves of a row's product have to meet; two launch structures do that without any
# synchronisation inside a kernel:
#   "split"  both directions in every launch, the path cut in TIME (dprnn_rnn_lin_split_fwd): phase 1 walks the first half of every
#            sequence in each direction (x', partial products), phase 2 the second halves (z, GroupNorm sums).  Launches of 2 x tiles
#            workgroups like the unfolded form, so two pipelines still pack the chip: +3 % at 64 x 3 s.
#   "serial" one direction per launch (dprnn_rnn_lin_fwd): a launch has ONE workgroup per tile -- 194 / 250 for 64 x 3 s on 256 CUs, 85 %
#            of the chip on average: 3 % SLOWER than no fold there, 4 % faster at 128 x 3 s, where a launch fills the chip 1.5 x over
#            (profiles/r05_linear_fold.txt).
#   "off"    recurrence + projection kernel (rounds 2-4).
#   "auto"   (default) per path, by how far the launches of the pipelines in flight cover the chip: serial from FOLD_SERIAL_MIN_FILL,
#            split from FOLD_SPLIT_MIN_FILL, else off -- a folded step costs 13-21 % more than an unfolded one (+12.5 % MFMAs, the
#            weight pieces of the Linear, a second set of stores), which pays only when the projection kernel it replaces would have
#            had to queue for CUs (profiles/r05_batch_sweep_fold.txt: below ~29 utterances the unfolded form is 1-2 % faster).
LIN_FOLD = os.environ.get("DPRNN_LIN_FOLD", "auto")
FOLD_SPLIT_MIN_FILL = float(os.environ.get("DPRNN_FOLD_SPLIT_MIN_FILL", "0.98"))
FOLD_SERIAL_MIN_FILL = float(os.environ.get("DPRNN_FOLD_SERIAL_MIN_FILL", "1.5"))
FOLD_TILES = (2032, 2048, 2064)


def fold_plan(nplan: int, ncu: int, concurrent: int, intra: bool):
    """(form, seq_tile) of one DPRNN path whose launches cover nplan sequences per pipeline, `concurrent` pipelines in flight:
    ("off", 0), ("split", tile) or ("serial", tile)."""
    if LIN_FOLD == "off":
        return "off", 0
    key = "intra" if intra else "inter"
    tiles = lambda t: -(-nplan // (t % 1000))
    if LIN_FOLD in ("serial", "auto"):
        tile = SEQ_TILE["serial_" + key] or split_tile(nplan, 1, ncu, concurrent, 3, FOLD_TILES)
        if tile in FOLD_TILES and (LIN_FOLD == "serial" or concurrent * tiles(tile) >= FOLD_SERIAL_MIN_FILL * ncu):
            return "serial", tile
    if LIN_FOLD in ("split", "auto"):
        tile = SEQ_TILE[key] or split_tile(nplan, 2, ncu, concurrent, 3, path=key)
        if tile in FOLD_TILES and (LIN_FOLD == "split" or 2 * concurrent * tiles(tile) >= FOLD_SPLIT_MIN_FILL * ncu):
            return "split", tile
    return "off", 0


def _path_folded(g: "_Group", form: str, tile: int, lin, stats_slot, packed, bsum, linp, intra: bool, addr, dev):
    """One bidirectional LSTM path with its Linear folded in (mode 3): two launches; leaves the new z in g.Z and its sums in stats_slot."""
    st, geo = g.st, g.geo
    spu, rpu = (geo.S if intra else geo.K), geo.S * geo.K
    lb = _p(_param(lin.bias, "linear.bias", dev))
    norm = (0, 0, 0, 0, 0.0)
    xout = 0
    if g.pending is not None:
        pstats, pg, pb, peps = g.pending
        norm, xout = (_p(g.Z), _p(pstats), _p(pg), _p(pb), peps), _p(g.X2)
    if form == "split":
        if getattr(g, "Z2", None) is None:      # the partial products / the new z: the pending z is still read while they are written
            g.Z2 = torch.empty_like(g.Z)
            g.lin_state = torch.empty(max(st.lib.dprnn_rnn_lin_state_floats(n_, t_) for n_ in (geo.B * geo.S, geo.B * geo.K) for t_ in FOLD_TILES),
                                      device=dev, dtype=torch.float32)
            g.buffers += [g.Z2, g.lin_state]
        tail = (_p(packed), _p(bsum), _p(linp))
        st.call("dprnn_rnn_lin_split_fwd", 1, _p(g.X), *norm, spu, rpu, xout, *tail, 0, _p(g.Z2), 0, _p(g.lin_state), *addr, tile)
        if g.pending is not None:
            g.X, g.X2 = g.X2, g.X
        st.call("dprnn_rnn_lin_split_fwd", 2, _p(g.X), 0, 0, 0, 0, 0.0, spu, rpu, 0, *tail, lb, _p(g.Z2), _p(stats_slot), _p(g.lin_state), *addr, tile)
        g.Z, g.Z2 = g.Z2, g.Z
    else:       # serial: the pending z is consumed row by row ahead of the partial products that replace it, in the same buffer
        tail = (_p(packed), _p(bsum), 2, _p(linp))
        st.call("dprnn_rnn_lin_fwd", 1, 0, _p(g.X), *norm, spu, rpu, xout, *tail, 0, _p(g.Z), 0, 0, *addr, tile)
        if g.pending is not None:
            g.X, g.X2 = g.X2, g.X
        st.call("dprnn_rnn_lin_fwd", 2, 1, _p(g.X), 0, 0, 0, 0, 0.0, spu, rpu, 0, *tail, lb, _p(g.Z), _p(g.Z), _p(stats_slot), *addr, tile)


def _run_path(g: _Group, rnn, lin, norm, stats_slot, packed, bsum, intra: bool, dbg, fuse: bool, scale_slot: int = SLOT_PATH0, linp=None):
    st, geo, Z, Hbuf = g.st, g.geo, g.Z, g.H
    _set_slot(st.lib, scale_slot)            # the record word this path's pack call ran under
    dev = Z.device
    ndir = 2 if rnn.rnn.bidirectional else 1
    B, S, K = geo.B, geo.S, geo.K
    # intra: sequences (b,s) walk k (dprnn.py:84-86); inter: sequences (b,k) walk s (dprnn.py:94-95)
    addr = (B * S, K, B * S, 0, K, 1) if intra else (B * K, S, K, S * K, 1, K)
    tile = SEQ_TILE["intra" if intra else "inter"]
    if tile == 0:
        # Sequences per workgroup, from measurements on MI355X (profiles/): 32 (two workgroups per CU) when that still
        # gives every CU its two workgroups, counting the utterance groups that run concurrently on other streams;
        # 16 below that (B=1: 28 ms instead of 48 ms per forward); the inter-chunk path, whose 64 sequences are one
        # contiguous 32 KB run of rows per step, takes 64 (one workgroup per CU, half the L2 weight bytes per FLOP)
        # when there are enough of those: +1.8 % at B=64.
        ncu = st.lib.dprnn_num_cus()
        wgs = lambda t: -(-addr[0] // t) * ndir * g.n_concurrent
        if not intra and wgs(64) >= 1.5 * ncu:
            tile = 64
        else:
            tile = 32 if wgs(32) >= 2 * ncu else 16
    cell = CELL_ID[rnn.rnn.mode]
    if cell != 0 and tile > 32:
        tile = 32
    nplan = addr[0] // g.geo.B * max(g.geo.B, getattr(g, "plan_B", 0))
    if cell == 0 and g.mode != 0:
        # every pipeline of a forward takes the tile of the LARGEST one: pipelines on different tiles finish apart, and the forward
        # waits for the slower (B = 25 as 12 + 13 utterances: 32.4 ms mixed against 30.9 / 31.9 ms on either tile for both)
        tile = SEQ_TILE["intra" if intra else "inter"] or split_tile(nplan, ndir, st.lib.dprnn_num_cus(), g.n_concurrent, g.mode,
                                                                     path="intra" if intra else "inter")
    w = _param(lin.weight, "linear.weight", dev)
    if w.shape != (F_BUILT, ndir * F_BUILT):
        raise DprnnError(f"linear weight shape {tuple(w.shape)} does not match ndir={ndir}")
    gamma, beta, eps = _norm_params(norm)
    gamma, beta = _param(gamma, "norm.weight", dev), _param(beta, "norm.bias", dev)
    if linp is not None and g.mode == 3 and cell == 0 and ndir == 2 and dbg is None and fuse:
        form, ftile = fold_plan(nplan, st.lib.dprnn_num_cus(), g.n_concurrent, intra)
        if form != "off":
            _path_folded(g, form, ftile, lin, stats_slot, packed, bsum, linp, intra, addr, dev)
            _count_scale(stats_slot, norm, st, B)
            g.pending = (stats_slot, gamma, beta, eps)
            return
    # f16 pieces: the recurrence hands the projection its operand (H as piece quads, dprnn_rnn_fwd_hp_ex) -- unless someone wants to see h
    hp = H_PIECES and g.mode == 3 and cell == 0 and dbg is None and tile >= 2000
    rnn_fwd = "dprnn_rnn_fwd_hp_ex" if hp else "dprnn_rnn_fwd_ex"
    if g.pending is None:
        st.call(rnn_fwd, g.mode, cell, _p(g.X), 0, 0, 0, 0, 0.0, 1, 1, 0, _p(packed), _p(bsum), _p(Hbuf), *addr, ndir, tile)
    else:       # previous path's norm + residual applied on the fly; x' lands in the other slab
        pstats, pg, pb, peps = g.pending
        st.call(rnn_fwd, g.mode, cell, _p(g.X), _p(Z), _p(pstats), _p(pg), _p(pb), peps, S if intra else K, S * K,
                _p(g.X2), _p(packed), _p(bsum), _p(Hbuf), *addr, ndir, tile)
        g.X, g.X2 = g.X2, g.X
        g.pending = None
    if dbg is not None:
        dbg.append(Hbuf[: geo.rows * ndir * F_BUILT].view(B, S, K, ndir * F_BUILT).clone())
    if hp:
        st.call("dprnn_proj_stats_hp_fwd", _p(Hbuf), _p(w), _p(_param(lin.bias, "linear.bias", dev)), _p(Z), _p(stats_slot),
                geo.rows, S * K, 0, 0, ndir * F_BUILT)
    else:
        st.call("dprnn_proj_stats_fwd", _p(Hbuf), _p(w), _p(_param(lin.bias, "linear.bias", dev)), _p(Z), _p(stats_slot),
                geo.rows, S * K, ndir * F_BUILT)
    _count_scale(stats_slot, norm, st, B)
    if fuse:
        g.pending = (stats_slot, gamma, beta, eps)
    else:
        st.call("dprnn_norm_residual_fwd", _p(g.X), _p(Z), _p(stats_slot), _p(gamma), _p(beta), eps, _p(g.X), B, S * K)


# bf16-split gate GEMM: measured cost of one time step of one workgroup, in microseconds (profiles/r02_split_wide.txt): the
# two-per-CU 32-sequence form and the one-per-CU eight-wave forms (seq_tile code 2000 + sequences); a launch takes
# ceil(workgroups / slots) rounds of T steps each
SPLIT_STEP_US = {32: 25.0, 2064: 18.6, 2080: 23.3}
SPLIT_GROUPS = 2             # utterance groups (streams) in the bf16-split modes: one group's launches rarely fill 256 CUs exactly


# gate-GEMM mode 3 (f16 pieces), eight-wave kernels only (profiles/r02_f16_pieces.txt): 16 and 64 / 80 sequences one-per-CU, 32 two-per-CU
SPLIT_STEP_US_F16 = {2016: 3.8, 2032: 13.2, 2064: 11.6, 2080: 15.3}   # 2016: first weight pieces resident in registers (few utterances)


SPLIT_WIDE_TILE_MARGIN = 1.13


# A launch of 64-sequence tiles whose workgroups -- all pipelines' together -- exceed the CUs by a few per cent leaves those few waiting a
# whole launch for a free CU; the pipeline they belong to falls behind the other one and finishes the forward alone on half the chip
# (round 5, profiles/r05_tile_zone.txt: -3 .. -5 % at 33-37 and 43-46 utterances, the "steps" of rounds 3 and 4).  The 16-sequence tile has
# four times the workgroups and packs such batches; the wide tile is taken when everything fits one round (fill <= 1) or from the
# fill where its lower cost per sequence wins again -- measured: 1.14 on the intra-chunk path, 1.33 on the inter-chunk path.
WIDE_TILE_MIN_FILL = {"intra": 1.14, "inter": 1.33}


def split_tile(nseq: int, ndir: int, ncu: int, concurrent: int = 1, mode: int = 1, only=None, path=None) -> int:
    """seq_tile code for the split recurrence over nseq sequences x ndir directions: the form whose grid takes the
    least time, rounds x cost of a step (64 x 3 s in two groups: 194 / 250 workgroups of 64 sequences per launch)."""
    best, best_t, t16 = None, None, None
    for tile, us in (SPLIT_STEP_US_F16 if mode == 3 else SPLIT_STEP_US).items():
        if only is not None and tile not in only:
            continue
        wgs = ndir * -(-nseq // (tile % 1000))
        two_per_cu = tile % 1000 == 32
        slots = ncu * (2 if two_per_cu else 1)
        # one pipeline pays for every partly filled round of workgroups; with several pipelines interleaving on the chip the tail of
        # one launch is filled by the others' kernels, so the cost is the launch's share of the chip's work (at least one workgroup's time)
        rounds = -(-wgs // slots) if concurrent == 1 else max(1.0, wgs * concurrent / slots) / concurrent
        t = rounds * us
        if two_per_cu and wgs * concurrent <= ncu:     # one 32-sequence workgroup per CU runs at the MFMA's pace, not the L2's
            t *= 0.6
        if tile == 2016:
            t16 = t
        if best_t is None or t < best_t - 1e-9:
            best, best_t = tile, t
    # The wider tiles' step costs were measured on a full grid; inside a forward with two pipelines their advantage over the
    # 16-sequence resident-weights tile is smaller than this model says (profiles/r04_inter_tile_sweep.txt: predicted 4 % / 11 % /
    # 17 % better at 13 / 14 / 15 utterances per pipeline, measured 3 % worse / 1 % worse / 1.5 % better): switch from 13 % up.
    if best != 2016 and t16 is not None and concurrent > 1 and t16 <= SPLIT_WIDE_TILE_MARGIN * best_t:
        best = 2016
    if path is not None and mode == 3 and best is not None and best != 2016 and t16 is not None and concurrent > 1 and best % 1000 >= 48:
        fill = concurrent * ndir * -(-nseq // (best % 1000)) / ncu          # one workgroup per CU at these tiles
        if 1.0 < fill < WIDE_TILE_MIN_FILL[path]:
            best = 2016
    return best


GROUP_SIZES = None        # experiment knob: explicit utterance-group sizes (must sum to the batch), else an even split


SPLIT_MIN_BATCH = int(os.environ.get("DPRNN_SPLIT_MIN_BATCH", "20"))    # mode 1: below this the exact fp32 kernels' small tiles are faster (profiles/r02_split_wide.txt: B = 16 62 vs 54 ms)


def effective_gemm_mode(B: int, blocks=None) -> int:
    """GEMM_MODE as it applies to a batch of B: the bf16-split kernels need launches that fill the chip, and only the
    LSTM cell has them; everything else runs the exact fp32 kernels (which are never less accurate)."""
    mode = current_mode()
    if mode == 0 or (mode != 3 and B < SPLIT_MIN_BATCH):      # the f16 form wins at every size, down to one utterance
        return 0
    if blocks is not None and not all(b.intra_rnn.rnn.mode == "LSTM" and b.inter_rnn.rnn.mode == "LSTM" for b in blocks):
        return 0
    return mode


def _leaves_exact_mode(fn):
    """The library's default gate-GEMM mode is per-thread state (dprnn_set_gemm_mode); the recurrences get their mode as an
    argument (dprnn_rnn_pack_ex / dprnn_rnn_fwd_ex) and the packed weights carry their format, the row GEMMs (projection,
    bottleneck, mask head) read the thread's default.  A forward sets it and, whatever happens, hands the thread back in the
    exact fp32 mode, so that no later caller of the C ABI on this thread runs under a mode it did not ask for."""
    @functools.wraps(fn)
    def run(st, *args, **kwargs):
        try:
            return fn(st, *args, **kwargs)
        finally:
            st.lib.dprnn_set_gemm_mode(0)
    return run


def _even_parts(B: int, n: int):
    cuts = [round(i * B / n) for i in range(n + 1)]
    return [(cuts[i], cuts[i + 1]) for i in range(n) if cuts[i + 1] > cuts[i]]


# Utterance groups by batch size (round 4; profiles/r04_batch_sweep.txt, r04_group_sweep.txt).  One group is best while its launches
# fit one round of workgroups (16-sequence tiles: up to 8 utterances); beyond that a single pipeline pays for every partly filled
# round of every launch (B = 9: the inter-chunk launch needs 282 workgroups on 256 CUs, a second round for 26 of them), and throughput
# DROPPED when a caller added utterances: B = 9 was 11 % below B = 8, B = 11 / 17 / 22 likewise.  Two independent pipelines on two
# streams fill each other's idle CUs -- measured +12 .. +25 % at B = 9 .. 24 in mode 3, +10 .. +13 % at B = 11 .. 44 in mode 0 -- so the
# split modes run two groups from 9 utterances up (was: from 32), the exact mode two from 11 (was: from 24), three from 40, four from 56.
def plan_groups(B: int, gemm_mode: int) -> int:
    if gemm_mode != 0:
        # 21 .. 24 utterances (round 5, profiles/r05_tile_zone.txt): as two pipelines the larger one's intra-chunk launch of 16-sequence
        # tiles (2 x 134 workgroups from 11 utterances) no longer fits the CUs and that pipeline falls behind: three pipelines of 7-8,
        # +1.9 % at 21; from 29 on the inter-chunk path takes 64-sequence tiles, which two pipelines of <= 16 fill exactly
        return 1 if B <= 8 else 3 if 21 <= B <= 24 and SPLIT_GROUPS == 2 else SPLIT_GROUPS
    return 1 if B <= 10 else 2 if B < 40 else 3 if B < 56 else min(4, STREAM_GROUPS)


def _split_mode_parts(B: int):
    """Utterance groups of the split modes: the wide kernels take one workgroup per CU, so few, large pipelines."""
    return _even_parts(B, max(1, min(plan_groups(B, 3), B)))


def group_parts(B: int, gemm_mode: int):
    """The utterance groups dprnn_process will use for a batch of B (bench.py sizes its roofline arithmetic with it)."""
    if gemm_mode != 0 and GROUP_SIZES is None and SPLIT_ONE_GROUP:
        return _split_mode_parts(B)
    return _split(B)


def _split(B: int):
    if GROUP_SIZES and sum(GROUP_SIZES) == B:
        cuts = [0]
        for n_ in GROUP_SIZES:
            cuts.append(cuts[-1] + n_)
        return [(cuts[i], cuts[i + 1]) for i in range(len(GROUP_SIZES))]
    if MIN_GROUP_BATCH != 16 or STREAM_GROUPS != 4:          # explicit knobs (bench.py --groups / --min-group, sweeps): the old rule
        n = max(1, min(STREAM_GROUPS, round(B / MIN_GROUP_BATCH)))
    else:
        n = plan_groups(B, 0)
    return _even_parts(B, n)


def pack_weights(st: Stage, sep, geo: Geometry, debug=None) -> dict:
    """Everything of a dprnn_process call that depends on the weights and the batch geometry only: the arithmetic mode, the utterance
    groups, and every LSTM (and, where a path will take the folded form, every Linear) in its kernel's fragment order -- packed on
    st's stream from the LIVE parameters (an optimizer step between two forwards shows in the next one)."""
    dev = st.device
    B, K, S = geo.B, geo.K, geo.S
    blocks = list(sep.dprnn_blocks)
    mode = effective_gemm_mode(B, blocks)
    packed = torch.empty(2 * len(blocks), 2 * PACK_PER_DIR_MAX, device=dev, dtype=torch.float32)
    bsum = torch.empty(2 * len(blocks), 2 * 4 * F_BUILT, device=dev, dtype=torch.float32)
    fuse = FUSE_NORM and debug is None          # debug taps want the residual stream after every path
    parts = _split(B) if debug is None else [(0, B)]
    if mode != 0 and GROUP_SIZES is None and SPLIT_ONE_GROUP and debug is None:
        parts = _split_mode_parts(B)   # few, large pipelines (two: their launches interleave)
    plan_B = max(b1 - b0 for b0, b1 in parts)
    # the Linears are packed for the folded launches only when some path of this forward will take them (fold_plan: batch-size dependent)
    folds = mode == 3 and fuse and any(fold_plan(plan_B * (S if intra else K), st.lib.dprnn_num_cus(), len(parts), intra)[0] != "off"
                                       for intra in (True, False))
    linp = torch.empty(2 * len(blocks), st.lib.dprnn_lin_pack_floats(), device=dev, dtype=torch.float32) if folds else None
    for i, blk in enumerate(blocks):
        for j, (rnn, lin) in enumerate(((blk.intra_rnn, blk.intra_linear), (blk.inter_rnn, blk.inter_linear))):
            ndir, fw, bw = _rnn_tensors(rnn, dev)
            _set_slot(st.lib, path_slot(i, j == 1))     # the pack folds the path's x scale into W_ih
            st.call("dprnn_rnn_pack_ex", mode, CELL_ID[rnn.rnn.mode], *[_p(t) for t in fw], *[_p(t) for t in bw],
                    _p(packed[2 * i + j]), _p(bsum[2 * i + j]))
            if folds and ndir == 2 and rnn.rnn.mode == "LSTM":
                st.call("dprnn_lin_pack_f16", _p(_param(lin.weight, "linear.weight", dev)), 2, _p(linp[2 * i + j]))
    return dict(mode=mode, packed=packed, bsum=bsum, linp=linp, parts=parts, plan_B=plan_B, fuse=fuse, key=(B, S, K, debug is None))


def prepack(st: Stage, sep, geo: Geometry, debug=None) -> dict:
    """pack_weights for a forward that calls dprnn_process more than once with the same weights (DPRNN-Spe-IRA's two passes share one
    pack).  (Round 5 also tried packing on a side stream beside the bottleneck -- ~60 small launches, 0.35 ms of the forward's 1.3 ms
    serial prologue: 0.5 % SLOWER, 886.9 / 886.5 / 881.1 against 891.7 / 891.1 / 890.0 utt/s interleaved on one box, gpurun_out r05u;
    the fork / join costs more than the launches it hides.)"""
    return pack_weights(st, sep, geo, debug)


@_leaves_exact_mode
def dprnn_process(st: Stage, sep, Y, wav, enc_w, dec_w, geo: Geometry, stats, relu_act: bool,
                  want_masked: bool = False, debug: Optional[dict] = None, masks_out: Optional[torch.Tensor] = None, packs=None):
    """Segmentation -> DPRNN blocks -> mask head -> mask*enc -> decoder taps.

    reference: dprnn.py:171-186 (== dprnn_spe.py:231-248), dprnn.py:277, encoder_decoder.py:40-42.
    Returns taps [B][2][L][ks] (and mask*enc of speaker 0 as [B][L][N] if want_masked).
    stats: [1 + 2*n_blocks][B][2] float64, zeroed; slot 0 belongs to the bottleneck."""
    # Y in a one-element list: the caller hands the frames over -- they are dead once segmented (0.8 GB at 64 x 3 s), and with no
    # reference left their block serves the forward's later allocations (a captured forward's pool is that much smaller)
    owned = isinstance(Y, list)
    if owned:
        Y = Y.pop()
    dev = Y.device
    B, L, K, P, S = geo.B, geo.L, geo.K, geo.P, geo.S
    blocks = list(sep.dprnn_blocks)

    if packs is None:
        packs = pack_weights(st, sep, geo, debug)
    elif packs["key"] != (B, S, K, debug is None):
        raise DprnnError("dprnn_process: the packed weights were prepared for another batch geometry")
    mode, packed, bsum, linp, parts, plan_B, fuse = (packs[k] for k in ("mode", "packed", "bsum", "linp", "parts", "plan_B", "fuse"))
    st.lib.dprnn_set_gemm_mode(mode)            # the row GEMMs of this forward (this thread's default)

    taps = torch.empty(B, 2, L, geo.ks, device=dev, dtype=torch.float32)
    masked = torch.empty(B, L, geo.N, device=dev, dtype=torch.float32) if want_masked else None

    if len(parts) == 1:
        groups = [_Group(st, geo, 0, B, dev)]
    else:
        fork = torch.cuda.Event()
        fork.record(st.tstream)
        groups = []
        for (b0, b1), side in zip(parts, _side_streams(dev, len(parts))):
            side.wait_event(fork)
            groups.append(_Group(Stage(dev, side), geo, b0, b1, dev))

    for g in groups:
        g.n_concurrent = len(groups)
        g.plan_B = plan_B
        g.mode = mode
        g.st.call("dprnn_segment_fwd", _p(Y[g.b0:]), _p(g.X), g.geo.B, L, K, P, S)
    if debug is not None:
        debug["segmented"] = groups[0].X.view(B, S, K, F_BUILT).clone()
    if owned:
        # The frames' block is released here and (bottleneck_frames sizes it for that) becomes the FIRST pipeline's X2, which that pipeline
        # writes from its second recurrence on -- while another pipeline's segmentation may still be reading the frames.  Eagerly
        # record_stream() makes the allocator wait for every reader.  Inside a capture there is no such bookkeeping (and cross-waits
        # between the forked streams crashed hipStreamEndCapture with three pipelines: gpurun_out/r05n_call.log), so a captured forward
        # with several pipelines keeps the frames until its join instead -- one block more in the graph pool (0.8 GB at 64 x 3 s).
        if len(groups) > 1 and getattr(_TLS, "capturing", False):
            groups[0].buffers.append(Y)
        elif len(groups) > 1:
            for g in groups:
                Y.record_stream(g.st.tstream)       # read by the groups' streams: the allocator waits for them before it reuses the block
        del Y
    for g in groups:
        g.alloc_rest()
    for i, blk in enumerate(blocks):
        for intra in (True, False):
            rnn, lin, norm = (blk.intra_rnn, blk.intra_linear, blk.intra_norm) if intra else \
                (blk.inter_rnn, blk.inter_linear, blk.inter_norm)
            slot = 1 + 2 * i + (0 if intra else 1)
            dbg = [] if debug is not None else None
            for g in groups:            # interleaved submission: every group advances one path at a time
                _run_path(g, rnn, lin, norm, stats[slot][g.b0:], packed[2 * i + (0 if intra else 1)],
                          bsum[2 * i + (0 if intra else 1)], intra, dbg, fuse, path_slot(i, not intra),
                          None if linp is None else linp[2 * i + (0 if intra else 1)])
            if debug is not None:
                debug[f"block{i}." + ("intra_rnn" if intra else "inter_rnn")] = dbg[0]
                debug[f"block{i}" + (".mid" if intra else "")] = groups[0].X.view(B, S, K, F_BUILT).clone()

    for g in groups:
        gB = g.geo.B
        U = g.H[: gB * L * 2 * F_BUILT]
        V = g.H[gB * L * 2 * F_BUILT: 2 * gB * L * 2 * F_BUILT]
        head_w = (_p(_param(sep.prelu.weight, "prelu.weight", dev)), _p(_param(sep.conv2d.weight, "conv2d.weight", dev)),
                  _p(_param(sep.conv2d.bias, "conv2d.bias", dev)))
        _set_slot(st.lib, SLOT_HEAD_CONV)
        if g.pending is None:
            g.st.call("dprnn_head_conv_ola_fwd", _p(g.X), *head_w, _p(U), gB, L, K, P, S)
        else:
            pstats, pg, pb, peps = g.pending
            g.st.call("dprnn_head_conv_ola_norm_fwd", _p(g.X), _p(g.Z), _p(pstats), _p(pg), _p(pb), peps, *head_w,
                      _p(U), gB, L, K, P, S)
            g.pending = None
        _set_slot(st.lib, SLOT_HEAD_GATE)
        g.st.call("dprnn_head_gate_fwd", _p(U), _p(_param(sep.out[0].weight, "out.0.weight", dev)),
                  _p(_param(sep.out[0].bias, "out.0.bias", dev)), _p(_param(sep.gate[0].weight, "gate.0.weight", dev)),
                  _p(_param(sep.gate[0].bias, "gate.0.bias", dev)), _p(V), gB * L * 2)
        if masks_out is not None:       # separation-level callers (DPRNN._dprnn_process) want the masks themselves
            g.st.call("dprnn_head_mask_decode_fwd_train", _p(V), _p(_param(sep.end_conv1x1.weight, "end_conv1x1.weight", dev)),
                      _p(wav[g.b0:]), _p(enc_w), _p(dec_w), _p(taps[g.b0:]), _p(masks_out[g.b0:]),
                      _p(masked[g.b0:] if want_masked else None), 1 if relu_act else 0, gB, geo.T, L, geo.N, geo.ks, geo.stride)
        else:
            g.st.call("dprnn_head_mask_decode_fwd", _p(V), _p(_param(sep.end_conv1x1.weight, "end_conv1x1.weight", dev)),
                      _p(wav[g.b0:]), _p(enc_w), _p(dec_w), _p(taps[g.b0:]), _p(masked[g.b0:] if want_masked else None),
                      1 if relu_act else 0, gB, geo.T, L, geo.N, geo.ks, geo.stride)
        if debug is not None:
            debug["ola_conv"] = U.view(B, L, 2 * F_BUILT).clone()
            debug["gated"] = V.view(B, L, 2, F_BUILT).clone()
    if len(groups) > 1:             # join: the caller's stream continues only after every group is done
        for g in groups:
            done = torch.cuda.Event()
            done.record(g.st.tstream)
            st.tstream.wait_event(done)
            if not getattr(_TLS, "capturing", False):       # a captured graph owns its memory pool: nothing to hand over
                for buf in g.buffers:
                    buf.record_stream(g.st.tstream)
    return (taps, masked) if want_masked else taps


def decode(st: Stage, taps, geo: Geometry, nspk: int):
    """ConvTranspose1d overlap (encoder_decoder.py:40-42) -> [B][nspk][T_out]."""
    out = torch.empty(geo.B, nspk, geo.T_out, device=taps.device, dtype=torch.float32)
    st.call("dprnn_decoder_fwd", _p(taps), _p(out), geo.B, nspk, geo.T_out, geo.L, geo.ks, geo.stride)
    return out


def check_input(x: torch.Tensor, what: str = "input") -> torch.Tensor:
    if not isinstance(x, torch.Tensor) or x.dim() != 2:
        raise DprnnError(f"{what} must be a [batch, samples] tensor")
    if not x.is_cuda:
        raise DprnnError(
            f"{what} is on {x.device}: the DPRNN MI355X path runs only on a HIP device (no CPU fallback); "
            "move the model and the input to 'cuda'")
    if x.dtype != torch.float32:
        raise DprnnError(f"{what} must be float32 (got {x.dtype})")
    return x.contiguous()


def _tasnet_forward_impl(model, wav: torch.Tensor, debug: Optional[dict] = None) -> torch.Tensor:
    """DPRNNTasNet.forward (dprnn.py:271-283): [B, T] -> [B, 2, T_out]."""
    cfg = model.cfg
    check_supported(cfg)
    wav = check_input(wav)
    dev = wav.device
    with torch.cuda.device(dev), _scale_record(model, dev) as rec:
        st = Stage(dev)
        geo = make_geometry(wav.shape[0], wav.shape[1], cfg["input_size"], cfg["chunk_length"], cfg["hop_length"],
                            cfg["kernel_size"], cfg["stride"])
        if geo.T_out < 2:
            raise DprnnError("utterance too short")
        sep = model.separation
        enc_w = _param(model.encoder.conv1d.weight, "encoder.conv1d.weight", dev)
        dec_w = _param(model.decoder.weight, "decoder.weight", dev)
        stats = torch.zeros(1 + 2 * len(sep.dprnn_blocks), geo.B, 2, device=dev, dtype=torch.float64)
        first = group_parts(geo.B, effective_gemm_mode(geo.B, list(sep.dprnn_blocks)))[0] if debug is None else (0, geo.B)
        packs = prepack(st, sep, geo, debug)
        frames = [frames_from_waveform(st, wav, enc_w, sep.bottleneck[0], sep.bottleneck[1], geo, stats[0],
                                       min_floats=(first[1] - first[0]) * geo.S * geo.K * F_BUILT)]
        if debug is not None:
            debug["bottleneck"] = frames[0].clone()
        taps = dprnn_process(st, sep, frames, wav, enc_w, dec_w, geo, stats, cfg["activation_type"] == "relu", debug=debug, packs=packs)
        out = decode(st, taps, geo, 2)
        _scale_end(st, rec)
        return out


# ------------------------------------------------------------------------------------------------ DPRNN-Spe / IRA
def check_supported_spe(cfg: dict) -> None:
    check_supported(cfg)
    bad = []
    if cfg["O"] != 128 or cfg["P"] != 256 or cfg["embeddings_size"] != 128:
        bad.append(f"speaker encoder must be O=128, P=256, embeddings_size=128 (got {cfg['O']}, {cfg['P']}, {cfg['embeddings_size']})")
    if cfg["fusion_type"] not in ("cat", "add", "mul", "film", "att"):
        bad.append(f"fusion_type {cfg['fusion_type']!r} (dprnn_spe.py:88-105 knows cat, add, mul, film, att)")
    if cfg["kernel_size"] < 2:
        bad.append("kernel_size must be >= 2 (aux_T divides by kernel_size // 2, dprnn_spe.py:159)")
    if bad:
        raise NotImplementedError("DPRNN-Spe MI355X path: unsupported configuration: " + "; ".join(bad))


def _linear_small(st: Stage, lin_w, lin_b, x, k0: int = 0, K: Optional[int] = None):
    dev = x.device
    w = _param(lin_w, "linear.weight", dev)
    w2 = w.reshape(w.shape[0], -1)
    K = w2.shape[1] - k0 if K is None else K
    out = torch.empty(x.shape[0], w2.shape[0], device=dev, dtype=torch.float32)
    st.call("dprnn_linear_small_fwd", _p(x), x.shape[1], _p(w2), w2.shape[1], k0, K,
            _p(_param(lin_b, "linear.bias", dev)) if lin_b is not None else 0, _p(out), out.shape[1], x.shape[0], w2.shape[0])
    return out


def _pointwise(st: Stage, A, rows, kdim, conv_w, conv_b, cout, in_scale=None, in_shift=None, aff_stride=0,
               rows_per_utt=0, in_prelu=None, col_stats=None, in_mean=None, out=None):
    dev = A.device
    w = _param(conv_w, "conv.weight", dev)
    if out is None:
        out = torch.empty(rows, cout, device=dev, dtype=torch.float32)
    # mode 3: the 1x1 convs of the speaker branch run in f16 pieces as well (one activation-scale slot for all of them); a launch that
    # also has to form BatchNorm batch statistics (col_stats) stays on the fp32 kernel inside the library
    prev = st.lib.dprnn_get_gemm_mode()
    st.lib.dprnn_set_gemm_mode(3 if current_mode() == 3 else 0)
    _set_slot(st.lib, SLOT_SPK)
    try:
        for col0 in range(0, cout, 128):
            st.call("dprnn_pointwise_fwd", _p(A), kdim, kdim, _p(in_scale), _p(in_shift), _p(in_mean), aff_stride, rows_per_utt,
                    _p(in_prelu), _p(w), kdim, _p(_param(conv_b, "conv.bias", dev)) if conv_b is not None else 0,
                    _p(out), cout, col0, _p(col_stats), rows)
    finally:
        st.lib.dprnn_set_gemm_mode(prev)
    return out


@contextlib.contextmanager
def eval_bn_folds(st: Stage, module):
    """Inside the block, the (scale, shift, mean) of every EVAL-mode BatchNorm of `module` is the one computed here, once, on st's stream --
    they depend on the layer's parameters and running statistics only.  A ragged batch runs the speaker branch once (or twice) per
    utterance on side streams forked from st afterwards: six dprnn_bn_prep launches per utterance and pass were a third of its launches."""
    dev = st.device
    folds = {id(m): _bn_fold(st, m, None, 1, dev) for m in module.modules()
             if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and not m.training} if RAGGED_BN_FOLDS_ONCE else {}
    prev = getattr(_TLS, "bn_folds", None)
    _TLS.bn_folds = folds
    try:
        yield
    finally:
        _TLS.bn_folds = prev


def _bn_fold(st: Stage, bn, col_stats, count, dev):
    if bn.momentum is None:
        raise NotImplementedError("BatchNorm1d(momentum=None) (cumulative average) is not supported")
    if col_stats is None:
        folds = getattr(_TLS, "bn_folds", None)
        if folds is not None and id(bn) in folds:
            return folds[id(bn)]
    C = bn.num_features
    scale = torch.empty(C, device=dev, dtype=torch.float32)
    shift = torch.empty_like(scale)
    mean = torch.empty_like(scale)
    st.call("dprnn_bn_prep", _p(_param(bn.weight, "bn.weight", dev)), _p(_param(bn.bias, "bn.bias", dev)),
            _p(bn.running_mean), _p(bn.running_var), _p(bn.num_batches_tracked), _p(col_stats), count,
            float(bn.momentum), float(bn.eps), _p(scale), _p(shift), _p(mean), C)
    return scale, shift, mean          # y -> (y - mean) * scale + shift


def _resblock(st: Stage, blk, Xin, B: int, T: int, cin: int, cout: int):
    """ResBlock.forward (dprnn_spe.py:31-42) on channels-last rows [B*T][cin] -> [B][T//3][cout]."""
    dev = Xin.device
    rows = B * T
    tr1, tr2 = blk.batch_norm1.training, blk.batch_norm2.training
    cs1 = torch.zeros(cout, 2, device=dev, dtype=torch.float64) if tr1 else None
    y1 = _pointwise(st, Xin, rows, cin, blk.conv1.weight, None, cout, col_stats=cs1)
    s1, t1, m1 = _bn_fold(st, blk.batch_norm1, cs1, rows, dev)
    cs2 = torch.zeros(cout, 2, device=dev, dtype=torch.float64) if tr2 else None
    y2 = _pointwise(st, y1, rows, cout, blk.conv2.weight, None, cout, in_scale=s1, in_shift=t1, in_mean=m1,
                    in_prelu=_param(blk.prelu1.weight, "prelu1.weight", dev), col_stats=cs2)
    s2, t2, m2 = _bn_fold(st, blk.batch_norm2, cs2, rows, dev)
    res = _pointwise(st, Xin, rows, cin, blk.conv_downsample.weight, None, cout) if blk.downsample else Xin
    out = torch.empty(B, T // 3, cout, device=dev, dtype=torch.float32)
    st.call("dprnn_resblock_pool_fwd", _p(y2), _p(s2), _p(t2), _p(m2), _p(res), _p(_param(blk.prelu2.weight, "prelu2.weight", dev)),
            _p(out), B, T, cout)
    return out, T // 3


def speaker_embedding(st: Stage, sep, A0, B: int, T: int, aux_len_dev, ks: int):
    """spk_encoder[2:] + time mean (dprnn_spe.py:117-122,156-163) from the first conv's output A0 [B*T][128]."""
    se = sep.spk_encoder
    h, t = _resblock(st, se[2], A0, B, T, 128, 128)
    h, t = _resblock(st, se[3], h, B, t, 128, 256)
    h, t = _resblock(st, se[4], h, B, t, 256, 256)
    if t < 1:
        raise DprnnError("reference utterance too short for three MaxPool1d(3) stages")
    E = _pointwise(st, h, B * t, 256, se[5].weight, se[5].bias, 128)
    emb = torch.empty(B, 128, device=A0.device, dtype=torch.float32)
    st.call("dprnn_time_mean_fwd", _p(E), _p(aux_len_dev), aux_len_dev.numel(), _p(emb), B, t, 128, ks)
    return emb


def _upload(host_tensor: torch.Tensor, dev) -> torch.Tensor:
    """Host -> device without stalling the host: a copy from pageable memory is staged by the driver in stream order, so the host
    waits for everything queued on that stream before it; pinned memory + non_blocking does not."""
    if getattr(_TLS, "capturing", False):
        raise DprnnError("host-to-device upload inside a graph capture")
    return host_tensor.contiguous().pin_memory().to(dev, non_blocking=True)


def _aux_len_tensor(aux_len, B: int, dev):
    if isinstance(aux_len, torch.Tensor) and aux_len.device == dev:
        t = aux_len.to(torch.float32).reshape(-1).contiguous()
    elif isinstance(aux_len, (int, float)) or (isinstance(aux_len, torch.Tensor) and aux_len.numel() == 1):
        t = torch.full((1,), float(aux_len), dtype=torch.float32, device=dev)          # a fill kernel: no host-to-device copy at all
    else:
        t = _upload(torch.as_tensor(aux_len, dtype=torch.float32).reshape(-1), dev)
    if t.numel() not in (1, B):
        raise DprnnError(f"aux_len must be a scalar or hold one value per utterance (got {t.numel()} for batch {B})")
    return t


def _fusion_operands(st: Stage, sep, cfg, emb):
    """(mul, add, row_bias) of the bottleneck for every fusion (dprnn_spe.py:165-229); 'att' additionally needs the
    per-frame row scale from _att_row_scale."""
    ft = cfg["fusion_type"]
    if ft == "att":          # output * (att*a + a) = output * a * (att + 1): a is the per-channel mul  (dprnn_spe.py:214-222)
        return _linear_small(st, sep.fusion_linear.weight, sep.fusion_linear.bias, emb), None, None
    if ft == "cat":          # conv over cat([x, emb]) = conv_x(x) + W[:, N:] @ emb  (dprnn_spe.py:186-192)
        return None, None, _linear_small(st, sep.bottleneck[1].weight, None, emb, k0=cfg["input_size"])
    if ft == "add":
        return None, _linear_small(st, sep.fusion_linear.weight, sep.fusion_linear.bias, emb), None
    if ft == "mul":
        return _linear_small(st, sep.fusion_linear.weight, sep.fusion_linear.bias, emb), None, None
    if ft == "film":
        return (_linear_small(st, sep.fusion_linear_1.weight, sep.fusion_linear_1.bias, emb),
                _linear_small(st, sep.fusion_linear_2.weight, sep.fusion_linear_2.bias, emb), None)
    raise NotImplementedError(ft)


def _att_row_scale(st: Stage, sep, cfg, wav, enc_w, geo: Geometry, stats_slot, a_vec):
    """Per-frame factor of the 'att' fusion: softmax attention over the averaged, normalised encoder output,
    nearest-upsampled to L, plus one (dprnn_spe.py:98-105,178-183,214-222)."""
    dev = wav.device
    g, b_, eps = _norm_params(sep.bottleneck[0])
    sc0 = torch.empty(geo.B, geo.N, device=dev, dtype=torch.float32)
    sh0 = torch.empty_like(sc0)
    st.call("dprnn_bottleneck_prep", _p(stats_slot), _p(g), _p(b_), eps, 0, 0, _p(sc0), _p(sh0), geo.B, geo.L, geo.N)
    ks_a = sep.average.kernel_size[0]
    if sep.average.stride[0] != ks_a or sep.average.groups != geo.N:
        raise DprnnError("unexpected averaging conv geometry in the 'att' fusion")
    La = (geo.L - ks_a) // ks_a + 1
    scores = torch.empty(geo.B, La, device=dev, dtype=torch.float32)
    row_scale = torch.empty(geo.B, geo.L, device=dev, dtype=torch.float32)
    st.call("dprnn_att_row_scale", _p(wav), _p(enc_w), _p(sc0), _p(sh0), _p(_param(sep.average.weight, "average.weight", dev)),
            _p(_param(sep.average.bias, "average.bias", dev)), _p(a_vec), _p(scores), _p(row_scale),
            geo.B, geo.T, geo.L, geo.N, geo.ks, geo.stride, ks_a)
    return row_scale


def _spe_forward_impl(model, wav, aux, aux_len, ira: bool = False, debug: Optional[dict] = None):
    """DPRNNSpeTasNet.forward (dprnn_spe.py:314-327) / DPRNNSpeIRATasNet.forward (dprnn_spe_ira.py:179-190):
    (mix [B,T], aux [B,Ta], aux_len) -> (estimate [B,T_out], speaker logits [B,num_spks])."""
    cfg = model.cfg
    check_supported_spe(cfg)
    wav, aux = check_input(wav), check_input(aux, "aux")
    if aux.shape[0] != wav.shape[0]:
        raise DprnnError("mixture and reference batch sizes differ")
    dev = wav.device
    with torch.cuda.device(dev), _scale_record(model, dev) as rec:
        st = Stage(dev)
        N, ks, stride = cfg["input_size"], cfg["kernel_size"], cfg["stride"]
        geo = make_geometry(wav.shape[0], wav.shape[1], N, cfg["chunk_length"], cfg["hop_length"], ks, stride)
        geo_a = make_geometry(aux.shape[0], aux.shape[1], N, cfg["chunk_length"], cfg["hop_length"], ks, stride)
        B = geo.B
        sep = model.separation
        packs = prepack(st, sep, geo, debug)        # beside the speaker branch and the bottleneck; both IRA passes use the one pack
        enc_w = _param(model.encoder.conv1d.weight, "encoder.conv1d.weight", dev)
        dec_w = _param(model.decoder.weight, "decoder.weight", dev)
        n_blocks = len(sep.dprnn_blocks)
        n_pass = 2 if ira else 1
        stats = torch.zeros(n_pass * (1 + 2 * n_blocks) + 2, B, 2, device=dev, dtype=torch.float64)
        aux_len_dev = _aux_len_tensor(aux_len, B, dev)
        se = sep.spk_encoder

        # speaker embedding of the reference utterance (dprnn_spe.py:156-163)
        sc_a, sh_a = bottleneck_affine(st, aux, enc_w, se[0], geo_a, stats[-1])
        A0 = bottleneck_frames(st, aux, enc_w, sc_a, sh_a, se[1], geo_a, scale_slot=SLOT_AUX_BOTTLENECK).view(B * geo_a.L, F_BUILT)
        v0 = speaker_embedding(st, sep, A0, B, geo_a.L, aux_len_dev, ks)
        if debug is not None:
            debug["emb"] = v0.clone()

        # first (only) separation pass
        mul, add, row_bias = _fusion_operands(st, sep, cfg, v0)
        sc, sh = bottleneck_affine(st, wav, enc_w, sep.bottleneck[0], geo, stats[0], mul, add)
        row_scale = _att_row_scale(st, sep, cfg, wav, enc_w, geo, stats[0], mul) if cfg["fusion_type"] == "att" else None
        Y = bottleneck_frames(st, wav, enc_w, sc, sh, sep.bottleneck[1], geo, row_bias, row_scale)
        relu_act = cfg["activation_type"] == "relu"
        v_last = v0
        if not ira:
            taps = dprnn_process(st, sep, Y, wav, enc_w, dec_w, geo, stats[: 1 + 2 * n_blocks], relu_act, debug=debug, packs=packs)
        else:
            taps, d0 = dprnn_process(st, sep, Y, wav, enc_w, dec_w, geo, stats[: 1 + 2 * n_blocks], relu_act,
                                     want_masked=True, packs=packs)
            # second embedding from the first estimate d0 = mask0 * enc  (dprnn_spe_ira.py:79-92)
            st.call("dprnn_rows_stats", _p(d0), _p(stats[-2]), B, geo.L, N)
            _count_scale(stats[-2], se[0], st, B)
            g0, b0, _ = _norm_params(se[0])
            sc_d = torch.empty(B, N, device=dev, dtype=torch.float32)
            sh_d = torch.empty_like(sc_d)
            st.call("dprnn_bottleneck_prep", _p(stats[-2]), _p(_param(g0, "spk_encoder.0.weight", dev)),
                    _p(_param(b0, "spk_encoder.0.bias", dev)), 1e-5, 0, 0, _p(sc_d), _p(sh_d), B, geo.L, N)
            A0d = _pointwise(st, d0.view(B * geo.L, N), B * geo.L, N, se[1].weight, se[1].bias, 128,
                             in_scale=sc_d, in_shift=sh_d, aff_stride=N, rows_per_utt=geo.L)
            v1 = speaker_embedding(st, sep, A0d, B, geo.L, aux_len_dev, ks)
            v1 = _linear_small(st, sep.aux_linear.weight, sep.aux_linear.bias, torch.cat((v0, v1), dim=1).contiguous())
            if debug is not None:
                debug["v1"] = v1.clone()
            # second pass re-uses the cached norm output (dprnn_spe_ira.py:96): same sc/sh, new fusion operands
            mul2, add2, row_bias2 = _fusion_operands(st, sep, cfg, v1)
            if mul2 is not None or add2 is not None:
                g, b_, eps = _norm_params(sep.bottleneck[0])
                st.call("dprnn_bottleneck_prep", _p(stats[0]), _p(g), _p(b_), eps, _p(mul2), _p(add2), _p(sc), _p(sh),
                        B, geo.L, N)
            row_scale2 = _att_row_scale(st, sep, cfg, wav, enc_w, geo, stats[0], mul2) if cfg["fusion_type"] == "att" else None
            Y = bottleneck_frames(st, wav, enc_w, sc, sh, sep.bottleneck[1], geo, row_bias2, row_scale2)
            taps = dprnn_process(st, sep, Y, wav, enc_w, dec_w, geo, stats[1 + 2 * n_blocks: 2 * (1 + 2 * n_blocks)],
                                 relu_act, debug=debug, packs=packs)
            v_last = v1
        logits = _linear_small(st, sep.pred_linear.weight, sep.pred_linear.bias, v_last)
        est = decode(st, taps, geo, 1)                    # speaker 0 only (dprnn_spe.py:325)
        _scale_end(st, rec)
        return est[:, 0, :], logits


# ------------------------------------------------------------------------------------------------ ragged batches
# Full-length test utterances differ in length (Libri2Mix test: 3 .. 14 s), and the reference evaluates them one at a time
# (src/inferencers/inferencer.py:54-58, inferencer_spe.py:27-32): zero-padding to a common length would change the
# per-utterance GroupNorm statistics.  Here utterances of DIFFERENT lengths go through ONE launch per DPRNN path: their
# [S_u][K][F] slabs are concatenated, the intra-chunk recurrence sees sum(S_u) sequences of K steps, the inter-chunk
# recurrence gets one 64-sequence workgroup descriptor per (utterance, 64 chunk positions) with its own step count S_u, and
# the per-utterance statistics are indexed through small tables.  Everything before the slab (encoder statistics,
# bottleneck, speaker branch) and after it (mask head, decoder) is launched per utterance: < 3 % of the work.
# Needs the bf16-split gate GEMM (the eight-wave 64-sequence kernel) and LSTM cells; callers fall back to per-length
# batches on streams otherwise (inference.separate_many).
RAGGED_STREAMS = int(os.environ.get("DPRNN_RAGGED_STREAMS", "8"))           # side streams for the per-utterance launches around the ragged slab (small kernels: they overlap)
# Reference utterances of EQUAL length go through the speaker branch as one batch (eval-mode BatchNorm: rows of different utterances never
# meet), this many at a time (3 KB of intermediates per reference sample: 64 x 3 s = 4.6 GB).  1: one utterance per launch set, as before.
RAGGED_AUX_BATCH = int(os.environ.get("DPRNN_RAGGED_AUX_BATCH", "64"))
# The mask head's gate (row-wise) as ONE launch over every utterance of a ragged batch instead of one per utterance.  0: per utterance.
RAGGED_GATE_BATCHED = os.environ.get("DPRNN_RAGGED_GATE_BATCHED", "1") != "0"
# ... and its conv2d + overlap-add as one launch too (dprnn_head_conv_ola_ragged_fwd, gate-GEMM mode 3).  0: one launch per utterance.
RAGGED_HEAD_CONV_BATCHED = os.environ.get("DPRNN_RAGGED_HEAD_CONV_BATCHED", "1") != "0"
# Eval-mode BatchNorm folds of the speaker encoder once per ragged forward instead of once per utterance and pass (eval_bn_folds).  0: per utterance.
RAGGED_BN_FOLDS_ONCE = os.environ.get("DPRNN_RAGGED_BN_FOLDS_ONCE", "1") != "0"
_FAN_POOL: dict = {}


# ------------------------------------------------------------------------------------------------ HIP graphs
# A forward is ~300 kernel launches enqueued from Python (~5 ms of host time at B = 64, 2-3 x the GPU time at B = 1).  Shapes are
# static and parameters are read through their (stable) device addresses, so the launch sequence of a forward -- every stream of
# it -- is captured once per (model, input shapes, mode, knobs) into a HIP graph and replayed with ONE launch: inputs are copied
# into the graph's static buffers, outputs are cloned out of them, the saturation flags are the graph's own buffer (its kernels
# carry that address) and are read after every replay exactly as after an eager forward.  The first forward of a key runs
# eagerly (it also performs the one-time hipFuncSetAttribute calls), the second captures, later ones replay.
#
# Memory (round 4): every capture of a device allocates from ONE engine-owned pool (torch.cuda.MemPool, held for the life of
# the process or until release_graph_pool()).  A forward's workspace dies inside its own capture, so the next capture reuses the
# same addresses: the pool is as large as the largest forward that was captured (6.4 GB at 64 x 3 s), not the sum over cached
# graphs (round 3: a private pool per capture, 25 GB for four graphs), and evicting a graph frees no device memory -- the event
# round 3's crash was traced to (segments of evicted graphs' private pools going back to the driver while a later capture was in
# progress: DESIGN.md 6b).  Graphs that share memory must not run concurrently: replays are serialised on the pool (a lock around
# the enqueue, an event between consecutive replays), whatever thread or stream they come from.  Before a capture the engine
# synchronises and returns the caching allocator's unused segments to the driver (empty_cache() below, under the capture lock), and
# it does not capture at all when the device has less free memory than the forward needs (the caching allocator's out-of-memory
# retry frees cached segments, and would do so inside the capture): such a key stays eager, with one warning.  A capture that
# raises leaves its key eager for good as well.  DPRNN_GRAPHS=0 turns graphs off.
GRAPHS = os.environ.get("DPRNN_GRAPHS", "1") != "0"
GRAPH_CACHE_MAX = 16
GRAPH_STATS = {"captures": 0, "replays": 0, "capture_failures": 0, "kept_eager_for_memory": 0}


# Every module-level setting of this file is listed in exactly one of these two tuples (tests/test_abi_and_host.py checks that no
# upper-case name is left out): LAUNCH_KNOBS change the launch sequence of a forward -- which kernels, tiles, streams, arithmetic --
# and are therefore part of every captured graph's key; the others are constants, bookkeeping, or policy applied outside a capture.
# A knob added to the module and forgotten here fails that test instead of replaying a stale graph.
LAUNCH_KNOBS = ("NARROW_NATIVE_MAX", "STREAM_GROUPS", "MIN_GROUP_BATCH", "FUSE_NORM", "SPLIT_ONE_GROUP", "SPLIT_GROUPS", "GROUP_SIZES", "SEQ_TILE", "SPLIT_WIDE_TILE_MARGIN",
                "SPLIT_MIN_BATCH", "H_PIECES", "ADAPTIVE_SCALE", "GEMM_MODE", "SPLIT_STEP_US", "SPLIT_STEP_US_F16", "RAGGED_STREAMS", "RAGGED_AUX_BATCH", "RAGGED_GATE_BATCHED", "RAGGED_HEAD_CONV_BATCHED", "RAGGED_BN_FOLDS_ONCE",
                "SATURATION_CHECK", "LIN_FOLD", "FOLD_SPLIT_MIN_FILL", "FOLD_SERIAL_MIN_FILL", "FOLD_TILES", "WIDE_TILE_MIN_FILL")
OTHER_SETTINGS = ("F_BUILT", "N_BUILT", "KS_MAX", "PACK_PER_DIR", "PACK_PER_DIR_MAX", "CELL_ID", "GEMM_MODES", "DEFAULT_GEMM_MODE",
                  "ON_SATURATION", "SATURATION", "SAT_NONFINITE", "PIN_AFTER_FALLBACKS", "PROFILE", "WS_PARTS", "SKIP_FIRST_STREAM_PAIR",
                  "GRAPHS", "GRAPH_CACHE_MAX", "GRAPH_STATS", "SLOT_BOTTLENECK", "SLOT_AUX_BOTTLENECK", "SLOT_HEAD_CONV", "SLOT_HEAD_GATE",
                  "SLOT_SPK", "SLOT_PATH0", "LAUNCH_KNOBS", "OTHER_SETTINGS")


def _freeze(v):
    if isinstance(v, dict):
        return tuple(sorted((k, _freeze(x)) for k, x in v.items()))
    if isinstance(v, (list, tuple)):
        return tuple(_freeze(x) for x in v)
    return v


def launch_knobs() -> tuple:
    """The current value of everything that shapes a forward's launch sequence (this thread's arithmetic mode included)."""
    lib = load_library()
    g = globals()
    return tuple((k, _freeze(g[k])) for k in LAUNCH_KNOBS) + (
        ("mode", current_mode()), ("fast_act", lib.dprnn_get_fast_activations()), ("fast_head_act", lib.dprnn_get_fast_head_activations()),
        ("rowgemm_grid", lib.dprnn_get_rowgemm_grid()))


# names `settings` may change: every launch knob, and the policy / bookkeeping settings that are not constants of the build
_SETTABLE = set(LAUNCH_KNOBS) | {"ON_SATURATION", "PIN_AFTER_FALLBACKS", "PROFILE", "GRAPHS", "GRAPH_CACHE_MAX"}


def current_settings() -> dict:
    """{name: value} of everything `settings` accepts (a snapshot: mutable values are copied one level deep)."""
    g = globals()
    return {k: (dict(g[k]) if isinstance(g[k], dict) else list(g[k]) if isinstance(g[k], list) else g[k]) for k in sorted(_SETTABLE)}


@contextlib.contextmanager
def settings(**values):
    """`with engine.settings(GEMM_MODE=0, STREAM_GROUPS=2): ...` -- engine settings for the forwards inside the block, put back on the
    way out whatever happens.  The one place that checks the NAME: an unknown name raises instead of creating a module attribute
    nobody reads (assigning `engine.NAME = v` directly stays possible -- bench.py and the sweeps do -- and has no such check).  The
    launch knobs are part of every captured graph's key (launch_knobs), so a block with other knobs never replays a graph captured
    outside it.  Process-wide like the assignments it replaces: not for two threads with different settings at once."""
    unknown = sorted(set(values) - _SETTABLE)
    if unknown:
        raise DprnnError(f"engine.settings: no such setting {unknown}; known: {sorted(_SETTABLE)}")
    g = globals()
    before = {k: g[k] for k in values}
    g.update(values)
    try:
        yield
    finally:
        g.update(before)


# ------------------------------------------------------------------------------------------------ guarded entry points
# What the model classes call.  Each runs its forward with saturation detection (see _guarded) and, should the f16-pieces
# arithmetic have met a value beyond its range, again in mode 1 (three bf16 pieces: no range assumption).
def _in_mode(mode: int, fn):
    def run():
        with use_gemm_mode(mode):
            return fn()
    return run


def _forward_bytes(model, wav, factor: float, aux=None) -> int:
    """Device memory one forward allocates, about: the library's own workspace figure for the geometry (dprnn_workspace_bytes) times
    a margin for what surrounds the slab (outputs, the speaker branch's activations over the reference utterance)."""
    cfg = model.cfg
    geo = make_geometry(wav.shape[0], wav.shape[1], cfg["input_size"], cfg["chunk_length"], cfg["hop_length"], cfg["kernel_size"], cfg["stride"])
    need = factor * workspace_parts(geo, len(model.separation.dprnn_blocks), True)["total"]
    if aux is not None:
        need += 8 * aux.numel() * F_BUILT * 4          # frames x 128 channels x a handful of ResBlock intermediates
    return int(need)


def _pinned(model, fn):
    """fn() under mode 1 when repeated saturation has pinned `model` there (see _guarded), else as is."""
    if model.__dict__.get("_dprnn_pin_mode1", False) and current_mode() == 3:
        with use_gemm_mode(1):
            return fn()
    return fn()


def _twin(model):
    """(the model the kernels run, its padded.Twin or None): narrow configurations run as zero-padded twins at the built widths."""
    from . import padded
    return padded.twin_of(model)


def _wider_than_built(model) -> bool:
    """Beyond the fused kernels' build -- wider (padded.fits) or an encoder filter longer than KS_MAX taps (Conv-TasNet-style kernel_size 16):
    such a model runs its inference forward on the any-width kernels (wide.py).  So does a DPRNNTasNet NARROWER than the build whose
    feature and hidden size are at most NARROW_NATIVE_MAX: its zero-padded twin would pay for 128-wide kernels."""
    from . import padded
    cfg = model.cfg
    if not padded.fits(cfg) or cfg["kernel_size"] > KS_MAX:
        return True
    return ("fusion_type" not in cfg and padded.needs_twin(cfg) and max(cfg["feature_size"], cfg["hidden_size"]) <= NARROW_NATIVE_MAX)


def tasnet_forward(model, wav: torch.Tensor, debug: Optional[dict] = None):
    wav = check_input(wav)
    if _wider_than_built(model):
        from . import wide
        out = wide.tasnet_forward(model, wav, debug)             # exact fp32: nothing to check afterwards
        if getattr(_TLS, "want_ticket", False):
            return out, ForwardTicket(wav.device, out, None, None, None, "DPRNNTasNet.forward (any-width kernels)", [], [], None)
        return out
    model, _ = _twin(model)
    if debug is not None:
        run = lambda: _tasnet_forward_impl(model, wav, debug)
        return _pinned(model, lambda: _guarded(wav.device, run, _in_mode(1, run), "DPRNNTasNet.forward", None, owner=model))
    return _pinned(model, lambda: _graphed("tasnet", model, (wav,), lambda w: _tasnet_forward_impl(model, w, None), "DPRNNTasNet.forward",
                                           need_bytes=lambda: _forward_bytes(model, wav, 1.25), owner=model))


def spe_forward(model, wav, aux, aux_len, ira: bool = False, debug: Optional[dict] = None):
    wav, aux = check_input(wav), check_input(aux, "aux")
    if _wider_than_built(model):
        from . import wide
        out = wide.spe_forward(model, wav, aux, aux_len, ira, debug)           # exact fp32: nothing to check afterwards
        if getattr(_TLS, "want_ticket", False):
            return out, ForwardTicket(wav.device, out, None, None, None, "DPRNNSpeTasNet.forward (any-width kernels)", [], [], None)
        return out
    narrow = model
    model, tw = _twin(model)
    if tw is not None and _bn_training(model.separation):
        # a train-mode BatchNorm updates its running statistics: they belong to the caller's (narrow) model
        if getattr(_TLS, "want_ticket", False):
            raise NotImplementedError("forward_async on a narrow model with BatchNorm in train mode")
        try:
            return _spe_forward_twin(model, wav, aux, aux_len, ira, debug)
        finally:
            torch.cuda.current_stream(wav.device).synchronize()
            tw.write_back_buffers(narrow)
    return _spe_forward_twin(model, wav, aux, aux_len, ira, debug)


def _spe_forward_twin(model, wav, aux, aux_len, ira: bool = False, debug: Optional[dict] = None):
    if debug is not None:
        run = lambda: _spe_forward_impl(model, wav, aux, aux_len, ira, debug)
        return _pinned(model, lambda: _guarded(wav.device, run, _in_mode(1, run), "DPRNNSpeTasNet.forward", model.separation, owner=model))
    alen = _aux_len_tensor(aux_len, wav.shape[0], wav.device)          # a device tensor: one of the graph's inputs
    return _pinned(model, lambda: _graphed("ira" if ira else "spe", model, (wav, aux, alen), lambda w, a, n: _spe_forward_impl(model, w, a, n, ira, None),
                                           "DPRNNSpeTasNet.forward", model.separation, need_bytes=lambda: _forward_bytes(model, wav, 2.0, aux), owner=model))


def forward_async(model, *inputs):
    """model(*inputs) without the host synchronisation that ends every synchronous forward in the default arithmetic (the read of
    the saturation flags): returns (output, ticket).  The output is enqueued on the current stream like any torch result; it is
    CHECKED only when ticket.result() is called, which returns it -- or, had a value left the f16 range, the range-free repeat.  A
    caller can keep several forwards in flight (upload the next batch, compute metrics of the previous one) and collect the tickets
    in order.  DPRNNTasNet, DPRNNSpeTasNet and DPRNNSpeIRATasNet, inference only."""
    if torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters()):
        raise NotImplementedError("forward_async is an inference entry point (call it under torch.no_grad())")
    if getattr(_TLS, "want_ticket", False) or getattr(_TLS, "guard", 0) > 0:
        raise DprnnError("forward_async inside another guarded forward")
    if _bn_training(model):
        # BatchNorm in train mode updates its running statistics in every forward, and a late fall-back restores the snapshot its own
        # forward took: with several forwards in flight that would roll back what later ones wrote.  Checked before it returns instead.
        out = model(*inputs)
        return out, ForwardTicket(inputs[0].device, out, None, None, None, "forward_async (synchronous: BatchNorm in train mode)", [], [], None)
    _TLS.want_ticket = True
    try:
        res = model(*inputs)
    finally:
        _TLS.want_ticket = False
    if not (isinstance(res, tuple) and len(res) == 2 and isinstance(res[1], ForwardTicket)):
        raise DprnnError(f"forward_async: {type(model).__name__}.forward did not go through the guarded inference path")
    return res


def _sep_off_build(sep) -> bool:
    """A separator whose widths are not exactly the fused kernels' build: its separation-level forwards run on the any-width kernels."""
    return (sep.input_size, sep.feature_size, sep.dprnn_blocks[0].intra_rnn.hidden_size) != (N_BUILT, F_BUILT, F_BUILT)


def separator_masks(sep, frames: torch.Tensor):
    if _sep_off_build(sep):
        from . import wide
        return wide.separator_masks(sep, frames)
    frames = _check_frames(frames, F_BUILT, "bottleneck output")
    run = lambda: _separator_masks_impl(sep, frames)
    return _guarded(frames.device, run, _in_mode(1, run), "DPRNN._dprnn_process", None, finite_inputs=[frames])


def separator_forward(sep, enc: torch.Tensor):
    if _sep_off_build(sep):
        from . import wide
        return wide.separator_forward(sep, enc)
    enc = _check_frames(enc, sep.input_size, "separator input")
    run = lambda: _separator_forward_impl(sep, enc)
    return _guarded(enc.device, run, _in_mode(1, run), "DPRNN.forward", None)


def auxiliary_embedding(sep, aux: torch.Tensor, aux_len):
    aux = _check_frames(aux, sep.input_size, "aux")
    run = lambda: _auxiliary_embedding_impl(sep, aux, aux_len)
    return _guarded(aux.device, run, _in_mode(1, run), "DPRNNSpe._auxiliary", sep.spk_encoder, finite_inputs=[aux])


def tasnet_forward_ragged(model, wavs):
    for w in wavs:
        check_input(w.reshape(1, -1))
    if _wider_than_built(model):
        from . import wide
        return wide.tasnet_forward_ragged(model, wavs)
    model, _ = _twin(model)
    run = lambda: _tasnet_forward_ragged_impl(model, wavs)
    return _guarded(wavs[0].device, run, _in_mode(1, run), "DPRNNTasNet ragged forward", None)


def spe_forward_ragged(model, wavs, auxs, aux_lens, ira: bool = False):
    for w in wavs:
        check_input(w.reshape(1, -1))
    if _wider_than_built(model):
        from . import wide
        return wide.spe_forward_ragged(model, wavs, auxs, aux_lens, ira)
    narrow = model
    model, tw = _twin(model)
    if tw is not None and _bn_training(model.separation):
        try:
            run = lambda: _spe_forward_ragged_impl(model, wavs, auxs, aux_lens, ira)
            return _guarded(wavs[0].device, run, _in_mode(1, run), "DPRNNSpeTasNet ragged forward", model.separation)
        finally:
            torch.cuda.current_stream(wavs[0].device).synchronize()
            tw.write_back_buffers(narrow)
    run = lambda: _spe_forward_ragged_impl(model, wavs, auxs, aux_lens, ira)
    return _guarded(wavs[0].device, run, _in_mode(1, run), "DPRNNSpeTasNet ragged forward", model.separation)


# ---- the rest of the engine (round 5: cut along its seams; every setting above stays the one namespace callers read and assign) ----
from .scale import (  # noqa: E402,F401
    ForwardTicket, _bn_buffers, _count, _guarded, _guarded_async, _saturated, _scale_end, _scale_record, _scale_tensor,
    _set_slot, current_mode, path_slot, reset_scale, scale_exponents, unpin, use_gemm_mode,
)
from .submodules import (  # noqa: E402,F401
    _auxiliary_embedding_impl, _check_frames, _device_f32, _no_grad_only, _rows_affine, _separator_forward_impl,
    _separator_masks_impl, block_forward, conv1x1_forward, decoder_forward, encoder_forward, fusion_forward, linear_forward,
    norm_forward, resblock_forward, single_rnn_forward,
)
from .ragged import (  # noqa: E402,F401
    _Fan, _bn_training, _spe_forward_ragged_impl, _tasnet_forward_ragged_impl, dprnn_process_ragged, ragged_supported,
)
from .graphs import (  # noqa: E402,F401
    _CAPTURE_LOCK, _EnginePool, _GRAPH_CACHE, _GRAPH_EAGER_KEYS, _GRAPH_LOCK, _GRAPH_RETIRED, _GRAPH_SEEN, _Graphed, _POOLS,
    _clone_out, _engine_pool, _graph_key, _graphed, _guard_call, _keep_eager, _pool_reserved_bytes, _replay_guarded, _retire,
    capture_lock, drop_graphs, empty_cache,
)
