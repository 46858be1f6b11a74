"""HIP-graph capture and replay of the inference forwards: one engine-owned memory pool per device, graphs created, first-launched and
destroyed under one lock (engine.capture_lock), the graph key, eviction.  Part of the engine (see engine.py: DESIGN.md section 6b)."""
from __future__ import annotations

import collections
import contextlib
import threading
import warnings

import torch

from . import engine as E
from ._lib import SAT_SLOTS, check, load_library


_GRAPH_CACHE: "collections.OrderedDict" = collections.OrderedDict()


_GRAPH_SEEN: "collections.OrderedDict" = collections.OrderedDict()


_GRAPH_EAGER_KEYS: set = set()            # keys whose capture failed or did not fit: eager from then on


_GRAPH_RETIRED: list = []                 # evicted graphs: destroyed under _CAPTURE_LOCK once no thread is replaying them (_retire)


_GRAPH_LOCK = threading.Lock()


_CAPTURE_LOCK = threading.RLock()        # torch registers generator state per capture, process-wide: one capture at a time; also


                                         # held by empty_cache() / drop_graphs(): no segment goes back to the driver during a capture


class _EnginePool:
    """The memory pool every captured forward of one device allocates from, and the order of their replays."""

    def __init__(self):
        self.pool = torch.cuda.MemPool()
        self.lock = threading.RLock()
        self.last = None                 # event recorded behind the latest replay


_POOLS: dict = {}


def _engine_pool(dev) -> _EnginePool:
    key = str(dev)
    with _GRAPH_LOCK:
        ep = _POOLS.get(key)
        if ep is None:
            ep = _POOLS[key] = _EnginePool()
        return ep


@contextlib.contextmanager
def capture_lock():
    """The lock under which this engine instantiates, first-launches and destroys its HIP graphs.  A host process that owns OTHER
    graphs (torch.cuda.CUDAGraph of its own, torch.compile's reduce-overhead mode, another library's) must create, first-replay and
    destroy them inside `with engine.capture_lock():` -- or run the engine with DPRNN_GRAPHS=0.  Why: on this ROCm stack a graph
    DESTROYED between another graph's instantiation and that graph's first launch faults hipGraphLaunch (round 3's segfault, located
    in round 4: profiles/r04_graph_churn.txt; no hipFree inside any capture is involved).  The engine keeps its own graphs out of
    that window with this lock; it cannot see anybody else's.  Re-entrant; also taken by engine.empty_cache() and drop_graphs()."""
    with _CAPTURE_LOCK:
        yield


def empty_cache() -> None:
    """torch.cuda.empty_cache() that waits for a capture in progress (use this one beside a serving engine: segments must not go
    back to the driver while another thread captures a forward)."""
    with _CAPTURE_LOCK:
        if torch.cuda.is_available():
            torch.cuda.empty_cache()


def _clone_out(o):
    if isinstance(o, torch.Tensor):
        return o.clone()
    if isinstance(o, (tuple, list)):
        return type(o)(_clone_out(v) for v in o)
    return o


def _retire(graphs=()) -> None:
    """Caller holds _CAPTURE_LOCK.  Evicted graphs are DESTROYED here and nowhere else: dropping the cache's reference is not enough
    when another thread fetched the graph a moment earlier and is still replaying it -- its reference would die outside the lock,
    possibly between a capture and that graph's first launch.  A graph in use stays on the list until a later call."""
    with _GRAPH_LOCK:
        _GRAPH_RETIRED.extend(graphs)
        idle = [g for g in _GRAPH_RETIRED if g.users == 0]
        _GRAPH_RETIRED[:] = [g for g in _GRAPH_RETIRED if g.users != 0]
    for g in idle:
        g.destroy()


class _Graphed:
    def __init__(self, dev, fn, inputs, ep: _EnginePool):
        lib = load_library()
        self.ep, self.dev = ep, dev
        self.users = 0                   # threads between fetching this graph from the cache and the end of their replay (_GRAPH_LOCK)
        prev = getattr(E._TLS, "guard", 0)
        E._TLS.guard, E._TLS.capturing = 1, True
        # Captured OUTSIDE inference mode whatever the caller's state: tensors created under torch.inference_mode() are inference
        # tensors -- the static inputs could then not be copied into from a plain no_grad caller, and torch's CUDA generator creates
        # its graph-safe state tensors at the first capture of a process: made under inference mode they break every later capture
        # outside it ("Inplace update to inference tensor outside InferenceMode", seen on the MI355X).  One graph serves both states.
        try:
            with torch.inference_mode(False), torch.no_grad():
                self.graph = torch.cuda.CUDAGraph()
                self.inputs = [t.clone() for t in inputs]           # static inputs and flags: ordinary allocations (a few MB)
                self.flags = torch.zeros(SAT_SLOTS, dtype=torch.int32, device=dev)
                torch.cuda.synchronize(dev)
                check(lib, lib.dprnn_saturation_begin(self.flags.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "dprnn_saturation_begin")
                with torch.cuda.graph(self.graph, pool=ep.pool.id, capture_error_mode="thread_local"):
                    self.outputs = fn(*self.inputs)
        finally:
            E._TLS.guard, E._TLS.capturing = prev, False
            lib.dprnn_saturation_begin(0, 0)

    def destroy(self):
        """Release the executable graph now (caller holds _CAPTURE_LOCK and has seen users == 0)."""
        graph, self.graph = self.graph, None
        if graph is not None:
            graph.reset()
        self.outputs = self.inputs = None

    def replay(self, inputs, host_flags=None):
        ep = self.ep
        lib = load_library()
        with ep.lock:                    # graphs of one pool share their workspace addresses: one at a time, in enqueue order
            cur = torch.cuda.current_stream(self.dev)
            if ep.last is not None:
                cur.wait_event(ep.last)
            # the graph's own flag buffer is cleared (and registered for the read that follows) behind the previous replay
            check(lib, lib.dprnn_saturation_begin(self.flags.data_ptr(), cur.cuda_stream), "dprnn_saturation_begin")
            for s_, t in zip(self.inputs, inputs):
                s_.copy_(t)
            self.graph.replay()
            out = _clone_out(self.outputs)
            if host_flags is not None:
                host_flags.copy_(self.flags, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(cur)
            ep.last = ev
        return out


def _graph_key(kind: str, model, inputs, extra=()):
    params = tuple(p.data_ptr() for p in model.parameters()) + tuple(b.data_ptr() for b in model.buffers())
    bn = tuple(m.training for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm))
    knobs = E.launch_knobs() + (("pinned", bool(model.__dict__.get("_dprnn_pin_mode1", False))),)
    return (kind, threading.get_ident(), id(model), tuple((tuple(t.shape), str(t.device)) for t in inputs), params, bn, knobs, extra)


def _pool_reserved_bytes(ep: _EnginePool) -> int:
    try:
        return sum(int(seg.get("total_size", 0)) for seg in torch.cuda.memory_snapshot(ep.pool.id))
    except Exception:
        return 0


def _keep_eager(key, why: str) -> None:
    with _GRAPH_LOCK:
        first = not _GRAPH_EAGER_KEYS
        _GRAPH_EAGER_KEYS.add(key)
        while len(_GRAPH_EAGER_KEYS) > 256:
            _GRAPH_EAGER_KEYS.pop()
    if first:
        warnings.warn(f"HIP-graph capture skipped, this forward stays on eager launches: {why} (further ones are counted in "
                      "engine.GRAPH_STATS without a warning)", RuntimeWarning, stacklevel=4)


def _guard_call(dev, primary, fallback, what, module, owner):
    """_guarded, or -- inside forward_async -- _guarded_async: (output, ticket)."""
    if getattr(E._TLS, "want_ticket", False):
        return E._guarded_async(dev, primary, fallback, what, module, owner=owner)
    return E._guarded(dev, primary, fallback, what, module, owner=owner)


def _graphed(kind: str, model, inputs, fn, what: str, module=None, extra=(), need_bytes=None, owner=None):
    """fn(*inputs) -- an inference forward over device tensors -- eagerly, or through its captured HIP graph (see above); either
    way under the saturation guard, falling back to an eager forward in mode 1.  need_bytes: device memory the forward allocates
    (a number or a callable), checked against what is free before capturing."""
    dev = inputs[0].device
    eager = lambda: fn(*inputs)
    usable = (E.GRAPHS and E.PROFILE is None and getattr(E._TLS, "guard", 0) == 0 and not getattr(E._TLS, "capturing", False)
              and not torch.is_grad_enabled())
    if not usable:
        return _guard_call(dev, eager, E._in_mode(1, eager), what, module, owner)
    key = _graph_key(kind, model, inputs, extra)
    with _GRAPH_LOCK:
        g = _GRAPH_CACHE.get(key)
        seen = 0
        if g is not None:
            _GRAPH_CACHE.move_to_end(key)
            g.users += 1
        elif key not in _GRAPH_EAGER_KEYS:
            seen = _GRAPH_SEEN.get(key, 0) + 1
            _GRAPH_SEEN[key] = seen
            _GRAPH_SEEN.move_to_end(key)
            while len(_GRAPH_SEEN) > 64:
                _GRAPH_SEEN.popitem(last=False)
    if g is None and seen < 2:
        return _guard_call(dev, eager, E._in_mode(1, eager), what, module, owner)
    with torch.cuda.device(dev):
        if g is None:
            ep = _engine_pool(dev)
            with _CAPTURE_LOCK:
                with _GRAPH_LOCK:
                    evicted = []
                    while len(_GRAPH_CACHE) >= max(1, E.GRAPH_CACHE_MAX):
                        evicted.append(_GRAPH_CACHE.popitem(last=False)[1])
                # nothing of the eager forwards' cached segments is left for the allocator to release while the capture runs
                torch.cuda.synchronize(dev)
                _retire(evicted)
                del evicted
                torch.cuda.empty_cache()
                if owner is not None and E.ADAPTIVE_SCALE and E.current_mode() == 3:
                    E._scale_tensor(owner, dev)            # the activation-scale record exists before the capture (never allocated inside one)
                need = need_bytes() if callable(need_bytes) else need_bytes
                if need is not None:
                    free, _ = torch.cuda.mem_get_info(dev)
                    have = free + _pool_reserved_bytes(ep)
                    if 1.1 * need > have:
                        E._count(E.GRAPH_STATS, "kept_eager_for_memory")
                        _keep_eager(key, f"{what} needs about {need / 2**30:.1f} GiB, {have / 2**30:.1f} GiB are free on {dev}")
                        return _guard_call(dev, eager, E._in_mode(1, eager), what, module, owner)
                try:
                    g = E._Graphed(dev, fn, inputs, ep)
                except Exception as exc:          # out of memory in the pool, a synchronising call inside the forward, ...
                    E._count(E.GRAPH_STATS, "capture_failures")
                    try:
                        torch.cuda.synchronize(dev)
                    except Exception:
                        pass
                    _keep_eager(key, f"capture of {what} raised {type(exc).__name__}: {exc}")
                    g = None
                if g is not None:
                    E._count(E.GRAPH_STATS, "captures")
                    with _GRAPH_LOCK:
                        _GRAPH_CACHE[key] = g
                        g.users += 1
                    # The new graph's FIRST launch happens here, still under the capture lock: graphs are destroyed only under this
                    # lock (eviction above, drop_graphs), and a graph destroyed between another graph's instantiation and its first
                    # launch is what crashed hipGraphLaunch in round 3 (profiles/r04_graph_churn.txt).
                    return _replay_guarded(g, dev, inputs, eager, what, module, owner)
            return _guard_call(dev, eager, E._in_mode(1, eager), what, module, owner)
        return _replay_guarded(g, dev, inputs, eager, what, module, owner)


def _replay_guarded(g, dev, inputs, eager, what, module, owner):
    """Replays g (whose `users` count the caller raised when it took g from the cache) and lets go of it."""
    def run(host_flags=None):
        E._count(E.GRAPH_STATS, "replays")
        return g.replay(inputs, host_flags)
    try:
        if getattr(E._TLS, "want_ticket", False):
            return E._guarded_async(dev, run, E._in_mode(1, eager), what, module, flags=g.flags, owner=owner, begin=False)
        return E._guarded(dev, run, E._in_mode(1, eager), what, module, flags=g.flags, owner=owner, begin=False)
    finally:
        with _GRAPH_LOCK:
            g.users -= 1


def drop_graphs(release_pool: bool = True) -> None:
    """Forget every captured forward; release_pool also hands the engine's graph memory pool back to the driver."""
    with _CAPTURE_LOCK:                  # never while another thread captures: segments go back to the driver here
        with _GRAPH_LOCK:
            dropped = list(_GRAPH_CACHE.values())
            _GRAPH_CACHE.clear()
            _GRAPH_SEEN.clear()
            _GRAPH_EAGER_KEYS.clear()
            pools = list(_POOLS.values()) if release_pool else []
            if release_pool:
                _POOLS.clear()
        if torch.cuda.is_available() and (dropped or pools or _GRAPH_RETIRED):
            torch.cuda.synchronize()
            _retire(dropped)
            del dropped, pools
            torch.cuda.empty_cache()
