"""Saturation guard and adaptive activation scale of the f16-pieces arithmetic (host side): the per-model scale record and its per-forward
working copy, _guarded / _guarded_async / ForwardTicket (a forward whose kernels met a value beyond the f16 range is repeated in a
range-free arithmetic), the thread's gate-GEMM mode.  Part of the engine: settings live in engine.py and are read through it (`E.NAME`)
at call time; engine.py re-exports everything defined here."""
from __future__ import annotations

import contextlib
import ctypes
import warnings
from typing import Optional

import torch

from . import engine as E
from ._lib import SAT_NAMES, SAT_SLOTS, SCALE_SLOTS, DprnnError, check, load_library


def _count(stats: dict, key: str, n: int = 1) -> int:
    with E._STATS_LOCK:
        stats[key] = stats[key] + n
        return stats[key]


def current_mode() -> int:
    """The mode this thread's forwards run in: engine.GEMM_MODE unless a `use_gemm_mode` block is active on the thread."""
    m = getattr(E._TLS, "mode", None)
    return E.GEMM_MODE if m is None else m


@contextlib.contextmanager
def use_gemm_mode(mode: int):
    """Run the forwards of this block, on this thread, in gate-GEMM mode `mode` (other threads keep theirs: the C library's mode
    is per thread too, and the packed weights carry their format)."""
    if mode not in E.GEMM_MODES:
        raise DprnnError(f"gate-GEMM mode {mode}: choose one of {E.GEMM_MODES}")
    prev = getattr(E._TLS, "mode", None)
    E._TLS.mode = mode
    try:
        yield
    finally:
        E._TLS.mode = prev


def path_slot(i_block: int, inter: bool) -> int:
    """Record slot of DPRNN path (block, intra / inter): its pack call, its recurrence launch."""
    return E.SLOT_PATH0 + (2 * i_block + int(inter)) % (SCALE_SLOTS - E.SLOT_PATH0)


def _set_slot(lib, slot: int) -> None:
    lib.dprnn_scale_slot(slot)


def _scale_tensor(owner, dev):
    recs = owner.__dict__.setdefault("_dprnn_scale_records", {})
    rec = recs.get(str(dev))
    if rec is None:
        if getattr(E._TLS, "capturing", False):
            raise DprnnError("activation-scale record created inside a graph capture (the first forward of a key runs eagerly)")
        lib = load_library()
        rec = torch.empty(4 * SCALE_SLOTS, device=dev, dtype=torch.float32)       # sticky maxima | unused | low-streak length | low-streak maximum
        check(lib, lib.dprnn_scale_init(rec.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "dprnn_scale_init")
        recs[str(dev)] = rec
    return rec


@contextlib.contextmanager
def _scale_record(owner, dev):
    """Forks `owner`'s record for the launches of this block (this thread, mode 3 only): the forward reads its scales from its OWN
    snapshot of the sticky maxima and collects its maxima there (csrc/api.hip: dprnn_scale_fork), so another thread's forward of the
    same model cannot change a scale between this one's pack call and the launches that read the pack; a captured forward's
    snapshot lives in the graph's pool and is taken again on every replay.  Yields (record, working copy) or None; the caller ends
    its forward with _scale_end, which joins the working copy into the record."""
    if owner is None or not E.ADAPTIVE_SCALE or current_mode() != 3 or getattr(E._TLS, "scale_depth", 0) > 0:
        yield None
        return
    lib = load_library()
    rec = _scale_tensor(owner, dev)
    work = torch.empty(2 * SCALE_SLOTS, device=dev, dtype=torch.float32)
    check(lib, lib.dprnn_scale_fork(rec.data_ptr(), work.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "dprnn_scale_fork")
    E._TLS.scale_depth = 1
    try:
        yield (rec, work)
    finally:
        E._TLS.scale_depth = 0
        lib.dprnn_scale_begin(0)


def scale_exponents(owner, dev=None):
    """{slot: s} of a model's record (2^s multiplies the activations of that site before the f16 split; 4 = the default): a
    synchronising read, for tests and diagnostics."""
    import math
    recs = getattr(owner, "_dprnn_scale_records", {})
    if not recs:
        return {}
    rec = recs[str(dev)] if dev is not None else next(iter(recs.values()))
    out = {}
    for i, v in enumerate(rec[:SCALE_SLOTS].cpu().tolist()):
        ex = math.frexp(v)[1] if 0.0 < v < 3.0e38 else 9
        out[i] = max(-8, min(4, 13 - ex))
    return out


def _scale_end(st, rec) -> None:
    """Last launch of a forward that ran under a record: the forward's maxima become part of the sticky ones (and a site that has
    stayed far below its sticky maximum for eight forwards comes down again)."""
    if rec is not None:
        st.call("dprnn_scale_join", rec[0].data_ptr(), rec[1].data_ptr())


def reset_scale(owner) -> None:
    """Forget what a model's activations have looked like so far: every conversion site starts from the default range again."""
    for dev, rec in getattr(owner, "_dprnn_scale_records", {}).items():
        lib = load_library()
        with torch.cuda.device(rec.device):
            check(lib, lib.dprnn_scale_init(rec.data_ptr(), torch.cuda.current_stream(rec.device).cuda_stream), "dprnn_scale_init")


def unpin(owner) -> None:
    """Let a model that was pinned to mode 1 after repeated saturation try the default arithmetic again."""
    owner.__dict__["_dprnn_pin_mode1"] = False
    owner.__dict__["_dprnn_sat_streak"] = 0


def _bn_buffers(module):
    if module is None:
        return []
    return [b for m in module.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training
            for b in (m.running_mean, m.running_var, m.num_batches_tracked) if b is not None]


def _guarded(dev, primary, fallback, what: str, module=None, active: Optional[bool] = None, flags: Optional[torch.Tensor] = None,
             owner=None, begin: bool = True, finite_inputs=None):
    """primary() with saturation detection on this thread.  If an f16-pieces kernel it launched converted a value beyond the
    f16 range (or met Inf / NaN), its result is discarded: BatchNorm running buffers primary() updated are restored and
    fallback() -- the same call in a range-free arithmetic -- supplies the result.  One stream synchronisation per call (the
    flags come back with a 32-byte copy).  Nested calls are covered by the outermost guard.
    active: whether detection is needed at all (default: the thread's mode is 3).  flags: the device buffer to register (a captured
    HIP graph's kernels carry the address of the buffer that was registered when they were captured).  begin=False: primary()
    clears and registers the flags itself (a graph replay does it inside its serialised section).  finite_inputs: tensors whose
    NaN / Inf content must raise the non-finite flag here -- the entry points that do not pass through dprnn_bottleneck_prep (where
    every full forward's input is checked): a float maximum skips NaN and the f16 conversion clamps it, so without this a NaN sample
    would come back as finite numbers from SingleRNN / DPRNNBlock / _dprnn_process / _auxiliary where the reference returns NaN."""
    if active is None:
        active = current_mode() == 3
    if not active or not E.SATURATION_CHECK or getattr(E._TLS, "guard", 0) > 0:
        return primary()
    lib = load_library()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev)
        if flags is None:
            flags = torch.empty(SAT_SLOTS, dtype=torch.int32, device=dev)
        bufs = _bn_buffers(module)
        keep = [b.clone() for b in bufs]
        if begin:
            check(lib, lib.dprnn_saturation_begin(flags.data_ptr(), stream.cuda_stream), "dprnn_saturation_begin")
        E._TLS.guard = 1
        host = (ctypes.c_int32 * SAT_SLOTS)()
        try:
            for t in (finite_inputs or ()):              # one reduction per input, OR-ed into the flag record on the device: no extra sync
                flags[E.SAT_NONFINITE: E.SAT_NONFINITE + 1].bitwise_or_(torch.isfinite(t).all().logical_not().to(torch.int32).view(1))
            out = primary()
            rc = lib.dprnn_saturation_flags(ctypes.addressof(host), stream.cuda_stream)
        finally:
            E._TLS.guard = 0
            lib.dprnn_saturation_begin(0, 0)
        if rc < 0:
            check(lib, rc, "dprnn_saturation_flags")
        _count(E.SATURATION, "checks")
        if rc == 0:
            if owner is not None:
                owner.__dict__["_dprnn_sat_streak"] = 0
            return out
        del out
        return _saturated(dev, list(host), what, bufs, keep, fallback, owner)


class ForwardTicket:
    """What forward_async / separate_many_async hand back beside the (enqueued, not yet checked) output: result() waits for the
    forward's saturation flags -- 32 bytes copied to pinned host memory behind the forward's last kernel -- and returns the output, or,
    when the f16-pieces arithmetic met a value beyond its range, the result of the same call repeated in a range-free arithmetic
    (exactly what the synchronous call does; a caller that never asks keeps a possibly clamped output: always ask)."""

    def __init__(self, dev, out, host, event, fallback, what, bufs, keep, owner):
        self.dev, self._out, self._host, self._event = dev, out, host, event
        self._fallback, self._what, self._bufs, self._keep, self._owner = fallback, what, bufs, keep, owner
        self._done = host is None

    def ready(self) -> bool:
        """True once the forward and its flag copy have finished on the device (never blocks)."""
        return self._done or self._event.query()

    def result(self):
        if self._done:
            return self._out
        self._event.synchronize()
        self._done = True
        host = [int(v) for v in self._host.tolist()]
        _count(E.SATURATION, "checks")
        if not any(host):
            if self._owner is not None:
                self._owner.__dict__["_dprnn_sat_streak"] = 0
            return self._out
        self._out = _saturated(self.dev, host, self._what, self._bufs, self._keep, self._fallback, self._owner)
        return self._out


def _saturated(dev, host, what, bufs, keep, fallback, owner):
    """A forward's flags came back set: count it, say so (once), restore what the discarded forward changed, run the fall-back."""
    hit = [SAT_NAMES[i] for i in range(SAT_SLOTS) if host[i]]
    n_fallbacks = _count(E.SATURATION, "fallbacks")
    with E._STATS_LOCK:
        E.SATURATION["last_flags"], E.SATURATION["last_what"] = list(host), what
    msg = (f"{what}: the f16-pieces arithmetic met values beyond its range ({', '.join(hit)}: |x| > 4062 after scaling, or "
           "Inf / NaN)")
    if E.ON_SATURATION == "raise":
        raise DprnnError(msg + "; DPRNN_ON_SATURATION=raise")
    if owner is not None:
        # The activation-scale record has adapted by now (the saturated forward left its maxima): the next forward normally fits.
        # A model that saturates again and again (values that grow from call to call, Inf / NaN inputs) stops paying twice.
        streak = owner.__dict__.get("_dprnn_sat_streak", 0) + 1
        owner.__dict__["_dprnn_sat_streak"] = streak
        if streak >= E.PIN_AFTER_FALLBACKS and not owner.__dict__.get("_dprnn_pin_mode1", False):
            owner.__dict__["_dprnn_pin_mode1"] = True
            warnings.warn(f"{what}: {streak} consecutive forwards left the range of the f16-pieces arithmetic; this model now runs in "
                          "gate-GEMM mode 1 (three bf16 pieces, no range assumption) -- engine.unpin(model) returns it to the default",
                          RuntimeWarning, stacklevel=4)
    if n_fallbacks == 1:
        warnings.warn(msg + " -- running the call again in a range-free arithmetic (further fall-backs are counted in "
                      "engine.SATURATION without a warning)", RuntimeWarning, stacklevel=4)
    with torch.cuda.device(dev):
        for b, k in zip(bufs, keep):
            b.copy_(k)
        prev = getattr(E._TLS, "guard", 0)
        E._TLS.guard = 1                 # the fall-back arithmetic has no range assumption: nothing to detect
        try:
            return fallback()
        finally:
            E._TLS.guard = prev


def _guarded_async(dev, primary, fallback, what: str, module=None, flags: Optional[torch.Tensor] = None, owner=None,
                   begin: bool = True):
    """_guarded without its stream synchronisation: primary() is enqueued, the flags travel to pinned host memory behind it, and
    the check (with the repeat, if needed) happens in ForwardTicket.result().  Returns (output, ticket)."""
    if current_mode() != 3 or not E.SATURATION_CHECK or getattr(E._TLS, "guard", 0) > 0:
        out = primary()
        return out, ForwardTicket(dev, out, None, None, None, what, [], [], owner)
    lib = load_library()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev)
        if flags is None:
            flags = torch.empty(SAT_SLOTS, dtype=torch.int32, device=dev)
        bufs = _bn_buffers(module)
        keep = [b.clone() for b in bufs]
        host = torch.empty(SAT_SLOTS, dtype=torch.int32).pin_memory()
        if begin:
            check(lib, lib.dprnn_saturation_begin(flags.data_ptr(), stream.cuda_stream), "dprnn_saturation_begin")
        E._TLS.guard = 1
        try:
            out = primary(host) if not begin else primary()      # a graph replay copies its flags out inside its serialised section
        finally:
            E._TLS.guard = 0
            lib.dprnn_saturation_begin(0, 0)
        if begin:
            host.copy_(flags, non_blocking=True)
        event = torch.cuda.Event()
        event.record(stream)
        return out, ForwardTicket(dev, out, host, event, fallback, what, bufs, keep, owner)
