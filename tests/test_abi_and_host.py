"""CPU-side checks: the C-ABI library builds/loads and exports every symbol of include/dprnn_mi355x.h,
the nn.Module mirrors keep the reference's state_dict contract, and the product path refuses to run
without a HIP device (no fallback)."""
import ctypes
import json
import os
import re

import pytest
import torch

from conftest import GOLDEN, ROOT, build_model

import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import _lib, engine


@pytest.fixture(scope="module")
def lib_path():
    path = pkg.library_path()
    if not os.path.exists(path):
        pkg.build_library()
    return path


def _header_functions():
    src = "".join(open(os.path.join(ROOT, "include", h)).read() for h in sorted(os.listdir(os.path.join(ROOT, "include")))
                  if h.endswith(".h"))
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dprnn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    names = _header_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/*.h but not exported"
    assert sorted(_lib.ALL_SYMBOLS) == names, "python binding table and header disagree"


def test_binding_table_has_one_argtype_per_header_parameter():
    """A ctypes call with fewer argtypes than arguments passes the extra ones as C ints: a 64-bit stream handle is then cut to
    32 bits and only the default stream (0) survives.  Every entry of the binding table must list as many argument types as
    the header prototype has parameters (the trailing `void* stream` included)."""
    src = "".join(open(os.path.join(ROOT, "include", h)).read() for h in sorted(os.listdir(os.path.join(ROOT, "include")))
                  if h.endswith(".h"))
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {m.group(1): m.group(2) for m in re.finditer(r"\b(dprnn_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S)}
    tables = dict(_lib.SIGNATURES)
    tables.update(_lib._VOID)
    tables.update({k: v[0] for k, v in _lib._SPECIAL.items()})
    assert set(tables) == set(protos)
    for name, args in tables.items():
        params = protos[name].strip()
        n = 0 if params in ("", "void") else len(params.split(","))
        assert len(args) == n, f"{name}: {len(args)} argtypes in _lib.py, {n} parameters in the header"


def test_library_loads_and_reports_abi(lib_path):
    lib = pkg.load_library()
    assert lib.dprnn_abi_version() == 3
    assert lib.dprnn_last_error() is not None


def test_null_pointer_is_rejected_without_gpu(lib_path):
    lib = pkg.load_library()
    rc = lib.dprnn_segment_fwd(None, None, 1, 10, 4, 2, 8, None)
    assert rc == -1 and b"null" in lib.dprnn_last_error()


@pytest.mark.parametrize("nt", ["ln", "gLN"])
def test_state_dict_contract_bss(nt):
    contract = json.load(open(os.path.join(GOLDEN, "state_contract.json")))[f"bss/cat/{nt}"]
    m = pkg.DPRNNTasNet(64, 128, 128, 250, 2, 125, 6, True, "LSTM", nt, "sigmoid", 0)
    assert [(k, list(v.shape)) for k, v in m.state_dict().items()] == [(k, list(s)) for k, s in contract["keys"]]
    assert sum(p.numel() for p in m.parameters()) == contract["n_params"]


@pytest.mark.parametrize("tag", ["spe/cat/ln", "spe/add/gLN", "spe/mul/ln", "spe/film/ln", "spe/att/ln", "ira/cat/ln", "ira/cat/gLN"])
def test_state_dict_contract_spe_ira(tag):
    from tss_with_dprnn_amd.dprnn_spe import DPRNNSpeTasNet
    from tss_with_dprnn_amd.dprnn_spe_ira import DPRNNSpeIRATasNet
    kind, fus, nt = tag.split("/")
    contract = json.load(open(os.path.join(GOLDEN, "state_contract.json")))[tag]
    cls = DPRNNSpeTasNet if kind == "spe" else DPRNNSpeIRATasNet
    m = cls(64, 128, 128, 250, 2, 125, 6, True, "LSTM", nt, "sigmoid", 0, None, 128, 256, 128, 251, fus)
    assert [(k, list(v.shape)) for k, v in m.state_dict().items()] == [(k, list(s)) for k, s in contract["keys"]]
    assert sum(p.numel() for p in m.parameters()) == contract["n_params"]


def test_dprnn_spe_module_reexports_tasnet():
    """scripts/test/config_bss.yaml:12 targets src.models.dprnn_spe.DPRNNTasNet."""
    from tss_with_dprnn_amd import dprnn_spe
    assert dprnn_spe.DPRNNTasNet is pkg.DPRNNTasNet


def test_constructor_defaults_match_reference():
    m = pkg.DPRNNTasNet(64)                      # dprnn.py:237-241 defaults
    assert m.cfg["chunk_length"] == 200 and m.cfg["hop_length"] == 100 and m.cfg["stride"] == 1
    assert m.cfg["norm_type"] == "ln" and m.cfg["activation_type"] == "sigmoid"
    assert isinstance(m.separation.bottleneck[0], torch.nn.GroupNorm)
    d = pkg.DPRNN(64)                            # dprnn.py:117-120: default norm is gLN here
    assert hasattr(d.bottleneck[0], "gamma")


def test_recipe_weights_load_strict():
    model, sd, cfg = build_model("bss", dict(n_repeats=1), 3)
    for k, v in model.state_dict().items():
        assert torch.equal(v, torch.from_numpy(sd[k])), k


def test_cpu_input_fails_loudly():
    model, _, _ = build_model("bss", dict(n_repeats=1), 3)
    with pytest.raises(pkg._lib.DprnnError, match="no CPU fallback"):
        model(torch.zeros(2, 400))


def test_unsupported_config_fails_loudly():
    m = pkg.DPRNNTasNet(8, 16, 16, 10, 2, 5, 1).eval()
    with pytest.raises(NotImplementedError, match="feature_size"):
        engine.check_supported(m.cfg)
    engine.check_supported(pkg.DPRNNTasNet(64, rnn_type="GRU").eval().cfg)      # GRU and RNN have kernels too
    with pytest.raises(AttributeError):
        pkg.DPRNNTasNet(64, rnn_type="QRNN")                                       # getattr(nn, rnn_type) in the reference


def test_training_path_has_no_cpu_fallback():
    m = pkg.DPRNNTasNet(64, n_repeats=1).train()
    with pytest.raises(pkg._lib.DprnnError, match="no CPU fallback"):     # differentiable forward: HIP device only
        m(torch.zeros(1, 300))
    g = pkg.DPRNNTasNet(64, n_repeats=1, rnn_type="GRU").train()        # GRU / RNN cells train too: same refusal of CPU tensors
    with pytest.raises(pkg._lib.DprnnError, match="no CPU fallback"):
        g(torch.zeros(1, 300))


def test_saved_activation_estimate():
    from tss_with_dprnn_amd import training
    geo = engine.make_geometry(1, 24000, 64, 250, 125, 2, 1)
    per_utt = training.saved_bytes_per_utt(geo, 12, [2] * 12)
    assert 3.5e9 < per_utt < 5e9                       # ~4.2 GB per 3 s utterance: 64 do not fit in 288 GB
    assert training.SAVE_BUDGET_BYTES // per_utt >= 32


def test_geometry_matches_reference_shapes():
    g = engine.make_geometry(64, 24000, 64, 250, 125, 2, 1)
    assert (g.L, g.S, g.T_out, g.rows) == (23999, 194, 24000, 64 * 194 * 250)
    g = engine.make_geometry(3, 203, 64, 10, 3, 4, 2)
    assert g.L == 100 and g.S == (100 + 10) // 3 + 1 and g.T_out == 202


def test_build_from_clean_in_a_scratch_directory(tmp_path):
    """build_library(force=True) compiles every HIP source from nothing (the in-tree objects are git-ignored and travel to
    the GPU box prebuilt, so no other test exercises a cold build); built beside the tree, the shipped library stays."""
    out = pkg.build_library(force=True, build_dir=str(tmp_path / "cold"))
    assert out.startswith(str(tmp_path)) and os.path.exists(out)
    objs = sorted(f for f in os.listdir(tmp_path / "cold") if f.endswith(".o"))
    srcs = sorted(f[:-4] + ".o" for f in os.listdir(os.path.join(ROOT, "tss-with-dprnn_amd", "csrc")) if f.endswith(".hip"))
    assert objs == srcs, "the Makefile does not build every csrc/*.hip"
    lib = ctypes.CDLL(out)
    for n in _header_functions():
        assert hasattr(lib, n), n
    assert os.path.exists(pkg.library_path())


def test_workspace_bytes_matches_the_host_geometry(lib_path):
    """dprnn_workspace_bytes (pure host arithmetic) against engine.make_geometry for the shipped config."""
    geo = engine.make_geometry(64, 24000, 64, 250, 125, 2, 1)
    ws = engine.workspace_parts(geo, 6)
    assert (geo.L, geo.S) == (23999, 194)
    assert ws["X"] == ws["X2"] == ws["Z"] == geo.rows * 128 * 4 == 64 * 194 * 250 * 512
    assert ws["H"] == 2 * ws["X"] and ws["Y"] == 64 * 23999 * 512
    assert ws["stats"] == 13 * 64 * 2 * 8 and ws["taps"] == 64 * 2 * 23999 * 2 * 4 and ws["out"] == 64 * 2 * 24000 * 4
    assert ws["total"] == sum(v for k, v in ws.items() if k != "total")
    assert 8.5e9 < ws["total"] < 9.0e9                    # 64 utterances x 3 s in one group: 8.8 GB of the 288 GB
    short = engine.workspace_parts(engine.make_geometry(2, 300, 64, 250, 125, 2, 1), 6)
    assert short["H"] >= max(2 * short["X"], 2 * 2 * 299 * 2 * 128 * 4)                     # the mask head's U + V fit too
    lib = pkg.load_library()
    assert lib.dprnn_workspace_bytes(1, 1, 64, 250, 125, 2, 1, 6, 0, None) == -1           # T < kernel size
    assert b"geometry" in lib.dprnn_last_error()


def test_dropout_is_the_no_op_it_is_in_the_reference():
    """dprnn.py:23-28 hands dropout to a num_layers=1 torch RNN: torch validates it, warns and never applies it."""
    with pytest.warns(UserWarning, match="num_layers greater than 1"):
        m = pkg.DPRNNTasNet(64, n_repeats=1, dropout=0.3)
    engine.check_supported(m.cfg)
    with pytest.raises(ValueError, match=r"range \[0, 1\]"):
        pkg.DPRNNTasNet(64, n_repeats=1, dropout=1.5)
    ref = torch.nn.LSTM(4, 4, num_layers=1, dropout=0)            # the same checks torch performs
    assert ref.dropout == 0


def test_split_tile_returns_codes_the_library_accepts():
    """engine.split_tile picks the recurrence tile of the split gate-GEMM modes from the launch geometry: whatever the batch,
    the code must be one the dispatcher builds (recur_split.hip: 32 / 2064 / 2080 / 2096 for the bf16 modes; 2016 .. 2096 for
    f16 pieces), a handful of utterances must get the low-latency 16-sequence tile and the headline launches the 64-sequence one."""
    from tss_with_dprnn_amd import engine
    for nseq in (1, 7, 194, 250, 1552, 3104, 6208, 8000, 12416, 50000):
        for ndir in (1, 2):
            for conc in (1, 2):
                assert engine.split_tile(nseq, ndir, 256, conc, 3) in (2016, 2032, 2048, 2064, 2080, 2096)
                assert engine.split_tile(nseq, ndir, 256, conc, 1) in (32, 2064, 2080, 2096)
    assert engine.split_tile(194, 2, 256, 1, 3) == 2016            # B = 1, intra-chunk path
    assert engine.split_tile(6208, 2, 256, 2, 3) == 2064           # B = 64 as two groups of 32, intra-chunk path
    assert engine.split_tile(8000, 2, 256, 2, 3) == 2064           # ... inter-chunk path


def test_forwards_hand_the_library_back_in_the_exact_mode():
    """engine._leaves_exact_mode: whatever a forward does -- return or raise -- the process-wide gate-GEMM mode is 0 afterwards."""
    from tss_with_dprnn_amd import engine

    class FakeLib:
        mode = 3
        def dprnn_set_gemm_mode(self, m):
            self.mode = m

    class FakeStage:
        lib = FakeLib()

    @engine._leaves_exact_mode
    def ok(st):
        st.lib.dprnn_set_gemm_mode(3)
        return 7

    @engine._leaves_exact_mode
    def boom(st):
        st.lib.dprnn_set_gemm_mode(1)
        raise ValueError("x")

    st = FakeStage()
    assert ok(st) == 7 and st.lib.mode == 0
    st.lib.mode = 3
    with pytest.raises(ValueError):
        boom(st)
    assert st.lib.mode == 0


def test_bottleneck_selects_its_arithmetic_and_puts_the_library_mode_back():
    """engine.bottleneck_frames runs before dprnn_process selects the forward's gate-GEMM mode: it launches under its own
    (3 = f16 pieces, anything else = fp32 MFMA) and restores whatever the library was in, also when the launch fails."""
    from tss_with_dprnn_amd import engine

    seen = []

    class FakeLib:
        mode = 1
        def dprnn_set_gemm_mode(self, m):
            self.mode = m
        def dprnn_get_gemm_mode(self):
            return self.mode
        def dprnn_scale_slot(self, slot):            # the bottleneck names its activation-scale slot before it launches
            seen.append(("slot", slot))
            return 0

    class FakeStage:
        lib = FakeLib()
        fail = False
        def call(self, name, *args):
            seen.append((name, self.lib.mode))
            if self.fail:
                raise engine.DprnnError("launch failed")

    st = FakeStage()
    conv = torch.nn.Conv1d(64, 128, 1)
    geo = engine.make_geometry(2, 400, 64, 50, 25, 2, 1)
    wav, enc_w, sc = torch.zeros(2, 400), torch.zeros(64, 1, 2), torch.zeros(2, 64)
    args = (st, wav, enc_w, sc, sc, conv, geo)
    engine.bottleneck_frames(*args, gemm_mode=3)
    assert seen[-1] == ("dprnn_bottleneck_fwd", 3) and st.lib.mode == 1
    engine.bottleneck_frames(*args, gemm_mode=1)                       # the bf16 modes have no row-GEMM form: fp32 MFMA
    assert seen[-1] == ("dprnn_bottleneck_fwd", 0) and st.lib.mode == 1
    engine.bottleneck_frames(*args, row_scale=torch.zeros(2 * geo.L), gemm_mode=3)
    assert seen[-1] == ("dprnn_bottleneck_att_fwd", 3) and st.lib.mode == 1
    old = engine.GEMM_MODE
    try:
        engine.GEMM_MODE = 0
        engine.bottleneck_frames(*args)                                # default: the engine's mode
        assert seen[-1] == ("dprnn_bottleneck_fwd", 0)
    finally:
        engine.GEMM_MODE = old
    st.fail = True
    with pytest.raises(engine.DprnnError):
        engine.bottleneck_frames(*args, gemm_mode=3)
    assert st.lib.mode == 1


def test_utterance_groups_partition_the_batch_and_leave_one_pipeline_only_while_it_fits_one_round():
    """engine.group_parts / plan_groups (round 4): the groups are a balanced partition of the batch; a single pipeline is chosen only
    while every recurrence launch of the batch fits ONE round of workgroups on the chip (16-sequence tiles, 3-s utterances: 194 chunks of
    250 positions) -- beyond that a lone pipeline pays for a partly filled second round of every launch and throughput per utterance
    drops when a caller adds utterances (profiles/r04_batch_sweep_before.txt: B = 9 was 11 % below B = 8)."""
    from tss_with_dprnn_amd import engine
    S, K, ncu = 194, 250, 256
    for mode in (0, 3):
        prev_n = 1
        for B in range(1, 129):
            parts = engine.group_parts(B, mode)
            assert parts[0][0] == 0 and parts[-1][1] == B and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert min(sizes) >= 1 and max(sizes) - min(sizes) <= 1
            n = len(parts)
            assert n == min(B, engine.plan_groups(B, mode))
            assert n >= prev_n or (mode == 3 and B == 25)        # never fewer pipelines for a larger batch -- but the measured window 21 .. 24
            prev_n = n
            if n == 1 and mode == 3:
                for nseq in (B * S, B * K):                                        # intra- and inter-chunk launches, both directions
                    assert 2 * -(-nseq // 16) <= ncu, (B, nseq)
    assert engine.plan_groups(9, 3) == 2 and engine.plan_groups(64, 3) == 2 and engine.plan_groups(64, 0) == 4


def test_wide_recurrence_tiles_need_a_clear_predicted_advantage():
    """engine.split_tile (round 4, profiles/r04_inter_tile_sweep.txt): with two pipelines the 64-sequence tile is taken only when the
    cost model predicts it at least SPLIT_WIDE_TILE_MARGIN better than the 16-sequence resident-weights tile (measured: predicted
    4 % / 11 % / 17 % better at 13 / 14 / 15 utterances per pipeline, measured 3 % worse / 1 % worse / 1.5 % better), and once taken
    it stays taken as the pipeline grows; a lone pipeline is not affected."""
    from tss_with_dprnn_amd import engine
    K, ncu = 250, 256
    tiles = [engine.split_tile(n * K, 2, ncu, 2, 3) for n in range(1, 33)]
    assert tiles[12 - 1] == 2016 and tiles[13 - 1] == 2016 and tiles[14 - 1] == 2016 and tiles[15 - 1] == 2064
    first_wide = tiles.index(2064)
    assert all(t == 2064 for t in tiles[first_wide:]) and all(t == 2016 for t in tiles[:first_wide])
    old = engine.SPLIT_WIDE_TILE_MARGIN
    try:
        engine.SPLIT_WIDE_TILE_MARGIN = 1.0
        assert engine.split_tile(13 * K, 2, ncu, 2, 3) == 2064                    # the bare cost model's choice
    finally:
        engine.SPLIT_WIDE_TILE_MARGIN = old


def test_the_exported_capture_lock_is_the_one_the_engine_captures_under():
    """INTEGRATION.md, "Other graphs in the same process": engine.capture_lock() hands a host the very lock the engine holds while it
    evicts, captures and first-launches a graph; it is re-entrant (a capture may run inside the host's own critical section) and
    excludes other threads."""
    import threading
    from tss_with_dprnn_amd import engine
    got = []
    with engine.capture_lock():
        assert engine._CAPTURE_LOCK._is_owned()
        with engine.capture_lock():                        # re-entrant
            pass
        t = threading.Thread(target=lambda: got.append(engine._CAPTURE_LOCK.acquire(blocking=False)))
        t.start(); t.join()
        assert got == [False]                              # another thread cannot capture / destroy meanwhile
    assert not engine._CAPTURE_LOCK._is_owned()
    t = threading.Thread(target=lambda: (got.append(engine._CAPTURE_LOCK.acquire(blocking=False)), engine._CAPTURE_LOCK.release()))
    t.start(); t.join()
    assert got == [False, True]


def test_evicted_graphs_are_destroyed_under_the_capture_lock_and_only_when_idle():
    """engine._retire: dropping the cache's reference does not destroy a graph another thread is still replaying; such a
    graph is parked and destroyed by a later call -- always by the holder of the capture lock."""
    from tss_with_dprnn_amd import engine

    class Fake:
        def __init__(self):
            self.users, self.destroyed_under_lock = 0, None

        def destroy(self):
            self.destroyed_under_lock = engine._CAPTURE_LOCK._is_owned()
    busy, idle = Fake(), Fake()
    busy.users = 1
    with engine.capture_lock():
        engine._retire([busy, idle])
    assert idle.destroyed_under_lock is True and busy.destroyed_under_lock is None and busy in engine._GRAPH_RETIRED
    busy.users = 0
    with engine.capture_lock():
        engine._retire()
    assert busy.destroyed_under_lock is True and not engine._GRAPH_RETIRED


def test_wide_tiles_are_not_taken_where_they_exceed_the_chip_by_a_few_workgroups():
    """engine.split_tile with a path (round 5, profiles/r05_tile_zone.txt): two pipelines whose 64-sequence launches together exceed the
    CUs by up to 14 % (intra-chunk) / 33 % (inter-chunk) run the 16-sequence tile instead; one round or less, or well beyond, keeps the
    wide tile."""
    from tss_with_dprnn_amd import engine
    S, K, ncu = 194, 250, 256
    tile = lambda n_utt, path: engine.split_tile(n_utt * (S if path == "intra" else K), 2, ncu, 2, 3, path=path)
    assert [tile(n, "inter") for n in (15, 16, 17, 18, 21, 22, 24, 32)] == [2064, 2064, 2016, 2016, 2016, 2064, 2064, 2064]
    assert [tile(n, "intra") for n in (19, 21, 22, 23, 24, 32)] == [2064, 2064, 2016, 2016, 2064, 2064]
    assert engine.split_tile(17 * K, 2, ncu, 2, 3) == 2064             # without a path: the bare cost model (tools, tests)


def test_the_linear_fold_is_taken_where_the_launches_cover_the_chip(monkeypatch):
    """engine.fold_plan (profiles/r05_linear_fold.txt): per DPRNN path, the time-split fold from where the two pipelines' launches
    (2 directions x tiles each) cover the CUs, the direction-serial fold where one direction's launches cover them 1.5 times, neither
    below -- and the explicit settings."""
    from tss_with_dprnn_amd import engine
    S, K, ncu = 194, 250, 256

    def plan(B):
        parts = engine.group_parts(B, 3)
        pB = max(b - a for a, b in parts)
        return [engine.fold_plan(pB * n, ncu, len(parts), intra) for n, intra in ((S, True), (K, False))]
    monkeypatch.setattr(engine, "LIN_FOLD", "auto")
    assert plan(1) == plan(8) == plan(28) == [("off", 0), ("off", 0)]
    assert plan(31) == [("off", 0), ("split", 2064)] and plan(40)[0] == ("off", 0)
    assert plan(35) == [("off", 0), ("off", 0)]            # 64-sequence inter-chunk tiles would exceed the CUs by a few workgroups: 16-sequence tiles
    assert plan(41) == [("split", 2064), ("off", 0)] and plan(45) == [("off", 0), ("split", 2064)]
    assert plan(47) == plan(64) == plan(96) == [("split", 2064), ("split", 2064)]
    assert plan(128) == [("serial", 2064), ("serial", 2064)]
    monkeypatch.setattr(engine, "LIN_FOLD", "off")
    assert plan(64) == [("off", 0), ("off", 0)]
    monkeypatch.setattr(engine, "LIN_FOLD", "serial")
    assert plan(64) == [("serial", 2064), ("serial", 2064)]
    monkeypatch.setattr(engine, "LIN_FOLD", "split")
    assert plan(64) == [("split", 2064), ("split", 2064)] and plan(2) == [("off", 0), ("off", 0)]      # 16-sequence tiles have no folded form


def test_no_undefined_names_in_the_package(root):
    """engine.py is cut into scale / submodules / ragged / graphs (round 5), whose code reads engine-level names through the module object
    at call time: a name that exists nowhere would only fail on the GPU box, in the branch that uses it.  tools/check_names.py (this image
    has no pyflakes) resolves every global a function loads and every `E.attribute` against the imported engine."""
    import subprocess
    import sys
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "check_names.py")], capture_output=True, text=True, cwd=root)
    assert res.returncode == 0, res.stdout + res.stderr


def test_every_engine_setting_is_classified_for_the_graph_key():
    """engine.LAUNCH_KNOBS (part of every captured graph's key) and engine.OTHER_SETTINGS together name every upper-case module-level
    setting of engine.py: a knob added later and forgotten fails here instead of replaying a graph captured under another value."""
    import types
    from tss_with_dprnn_amd import engine
    names = {k for k, v in vars(engine).items()
             if k.isupper() and not k.startswith("_") and not isinstance(v, (types.ModuleType, types.FunctionType, type))
             and k not in ("SAT_NAMES", "SAT_SLOTS", "SCALE_SLOTS")}                # imported from _lib
    listed = set(engine.LAUNCH_KNOBS) | set(engine.OTHER_SETTINGS)
    assert names - listed == set(), f"unclassified settings: {sorted(names - listed)}"
    assert listed - names == set(), f"listed but gone: {sorted(listed - names)}"
    assert not set(engine.LAUNCH_KNOBS) & set(engine.OTHER_SETTINGS)


def test_engine_settings_block_checks_names_and_restores():
    """engine.settings(**knobs): the named settings hold inside the block and are back afterwards (also when the block raises); an
    unknown name raises instead of creating an attribute nobody reads; a launch knob changed by the block changes launch_knobs()
    -- the part of every captured graph's key -- so the block cannot replay a graph captured outside it."""
    import pytest
    from tss_with_dprnn_amd import engine
    before = engine.current_settings()
    assert set(engine.LAUNCH_KNOBS) <= set(before)
    key0 = engine.launch_knobs()
    with engine.settings(GEMM_MODE=0, STREAM_GROUPS=2, GRAPHS=False):
        assert (engine.GEMM_MODE, engine.STREAM_GROUPS, engine.GRAPHS) == (0, 2, False)
        assert engine.launch_knobs() != key0
    assert engine.current_settings() == before and engine.launch_knobs() == key0
    with pytest.raises(RuntimeError):
        with engine.settings(LIN_FOLD="off"):
            assert engine.LIN_FOLD == "off"
            raise RuntimeError("inside")
    assert engine.current_settings() == before
    with pytest.raises(engine.DprnnError, match="no such setting"):
        with engine.settings(STREAM_GROUP=2):
            pass
    assert not hasattr(engine, "STREAM_GROUP")
    with pytest.raises(engine.DprnnError):                      # constants of the build are not settings
        with engine.settings(F_BUILT=256):
            pass


def test_padded_twin_places_every_block_and_scales_every_global_norm():
    """padded.Twin (CPU: construction and parameter placement only; the arithmetic is checked on the GPU against the narrow goldens): every
    value of the narrow model appears in the twin exactly once and everything else is zero; gate blocks of the RNN weights, direction
    blocks of the Linears, speaker blocks of conv2d and the (x | embedding) column blocks land at the built offsets; every global norm
    carries built / real channels as its element-count correction; the twin follows an in-place parameter update."""
    import numpy as np
    import torch
    from conftest import build_model
    from tss_with_dprnn_amd import padded
    tiny = dict(input_size=8, feature_size=16, hidden_size=16, chunk_length=10, kernel_size=2, hop_length=5, n_repeats=2)
    tspe = dict(tiny, O=16, P=24, embeddings_size=12, num_spks=7)
    for kind, over in (("bss", tiny), ("bss", dict(tiny, rnn_type="GRU", bidirectional=False)), ("spe", dict(tspe, fusion_type="cat", norm_type="ln")),
                       ("spe", dict(tspe, fusion_type="film", norm_type="gLN")), ("ira", dict(tspe, fusion_type="att", norm_type="ln"))):
        m, _, _ = build_model(kind, over, 3, None)
        tm, tw = padded.twin_of(m)
        assert tw is not None and tm.cfg["input_size"] == 64 and tm.cfg["feature_size"] == 128 and tm.cfg["hidden_size"] == 128
        tsd = tm.state_dict()
        for k, v in m.state_dict().items():
            a, b = v.double(), tsd[k].double()
            assert abs(a.sum().item() - b.sum().item()) <= 1e-9 * max(1.0, a.abs().sum().item()), k
            assert abs(a.abs().sum().item() - b.abs().sum().item()) <= 1e-9 * max(1.0, a.abs().sum().item()), k
        scales = {mod.__dict__["_dprnn_count_scale"] for mod in tm.modules() if "_dprnn_count_scale" in mod.__dict__}
        assert scales == {8.0}, scales                                  # 64 / 8 for the norms over N, 128 / 16 for those over F
        assert padded.twin_of(m)[1] is tw                               # built once
    m, _, _ = build_model("bss", tiny, 3, None)
    tm, tw = padded.twin_of(m)
    blk, tblk = m.separation.dprnn_blocks[0], tm.separation.dprnn_blocks[0]
    w, W = blk.intra_rnn.rnn.weight_ih_l0, tblk.intra_rnn.rnn.weight_ih_l0
    for g in range(4):                                                  # gate blocks i, f, g, o
        assert torch.equal(W[128 * g: 128 * g + 16, :16], w[16 * g: 16 * (g + 1)]) and float(W[128 * g + 16: 128 * (g + 1)].abs().sum()) == 0.0
    lw, LW = blk.intra_linear.weight, tblk.intra_linear.weight        # [F, 2H]: forward | backward columns
    assert torch.equal(LW[:16, :16], lw[:, :16]) and torch.equal(LW[:16, 128:144], lw[:, 16:]) and float(LW[16:].abs().sum()) == 0.0
    c, C = m.separation.conv2d.weight, tm.separation.conv2d.weight    # [2F, F, 1, 1]: speaker blocks
    assert torch.equal(C[:16, :16], c[:16]) and torch.equal(C[128:144, :16], c[16:])
    with torch.no_grad():
        blk.intra_norm.weight.mul_(3.0)
    tm2, _ = padded.twin_of(m)
    assert tm2 is tm and torch.equal(tblk.intra_norm.weight[:16], blk.intra_norm.weight) and float(tblk.intra_norm.weight[16:].abs().sum()) == 0.0
    big = dict(tiny, feature_size=256, hidden_size=256, input_size=64)
    mb, _, _ = build_model("bss", big, 3, None)
    with pytest.raises(NotImplementedError, match="above the kernels' build"):
        padded.twin_of(mb)


def test_placed_parameters_hand_a_narrow_model_the_real_blocks_of_its_twins_gradient():
    """Training a narrow model (padded.placed_parameters): the twin's parameters as differentiable functions of the narrow ones.  Values equal
    the twin's own; a gradient w.r.t. a placed parameter reaches the narrow parameter as exactly the blocks that are real (gate blocks,
    direction blocks, speaker blocks), and nothing of the padding."""
    import torch
    from conftest import build_model
    from tss_with_dprnn_amd import padded
    tiny = dict(input_size=8, feature_size=16, hidden_size=16, chunk_length=10, kernel_size=2, hop_length=5, n_repeats=1)
    m, _, _ = build_model("bss", tiny, 5, None)
    tm, tw = padded.twin_of(m)
    placed = padded.placed_parameters(m, tw)
    twin = dict(tm.named_parameters())
    gen = torch.Generator().manual_seed(0)
    up = {k: torch.randn(v.shape, generator=gen) for k, v in twin.items()}
    total = 0.0
    for k, v in placed.items():
        assert torch.equal(v.detach(), twin[k].detach()), k
        total = total + (v * up[k]).sum()
    total.backward()
    for k, prm in m.named_parameters():
        assert prm.grad is not None and prm.grad.shape == prm.shape, k
        want = torch.zeros_like(prm)
        scratch = torch.zeros_like(twin[k])
        # the inverse of the placement, by placing an index image: element i of the narrow tensor lands where its index lands
        idx = torch.arange(prm.numel(), dtype=torch.float32).reshape(prm.shape) + 1.0
        padded._place(scratch, idx, padded._blocks(k, tuple(prm.shape), tw.narrow_cfg))
        where = scratch.reshape(-1).nonzero().squeeze(1)
        assert where.numel() == prm.numel(), k
        want.reshape(-1)[(scratch.reshape(-1)[where] - 1.0).long()] = up[k].reshape(-1)[where]
        assert torch.equal(prm.grad, want), k


def test_which_models_take_the_any_width_kernels_and_which_limits_raise(monkeypatch):
    """Host-side routing of configurations off the fused kernels' build (engine._wider_than_built, engine._sep_off_build, wide.check_supported*):
    wider models and long encoder filters go to wide.py, narrower ones stay zero-padded twins (unless NARROW_NATIVE_MAX says otherwise), the
    speaker-conditioned classes go there only with feature / hidden size wide; every limit is a NotImplementedError that names itself."""
    import tss_with_dprnn_amd as pkg
    from tss_with_dprnn_amd import dprnn_spe, engine, wide
    mk = lambda **kw: pkg.DPRNNTasNet(**dict(dict(input_size=64, chunk_length=16, n_repeats=1), **kw))
    assert not engine._wider_than_built(mk())                                            # the build itself
    assert engine._wider_than_built(mk(feature_size=256, hidden_size=256))
    assert engine._wider_than_built(mk(hidden_size=129)) and engine._wider_than_built(mk(input_size=65))
    assert engine._wider_than_built(mk(kernel_size=16))                                  # Conv-TasNet's encoder: beyond KS_MAX taps
    narrow = mk(feature_size=64, hidden_size=48)
    assert not engine._wider_than_built(narrow)                                          # a zero-padded twin ...
    monkeypatch.setattr(engine, "NARROW_NATIVE_MAX", 64)
    assert engine._wider_than_built(narrow) and not engine._wider_than_built(mk(feature_size=96, hidden_size=96))
    monkeypatch.setattr(engine, "NARROW_NATIVE_MAX", 0)
    spe = dprnn_spe.DPRNNSpeTasNet(64, feature_size=64, hidden_size=64, chunk_length=16, n_repeats=1)
    monkeypatch.setattr(engine, "NARROW_NATIVE_MAX", 128)
    assert not engine._wider_than_built(spe)                                             # ... the speaker-conditioned classes always
    monkeypatch.setattr(engine, "NARROW_NATIVE_MAX", 0)
    assert engine._sep_off_build(narrow.separation) and not engine._sep_off_build(mk().separation)
    assert engine.ragged_supported(mk(feature_size=256, hidden_size=256, rnn_type="GRU").separation)       # wide.tasnet_forward_ragged: any cell
    assert engine.ragged_supported(dprnn_spe.DPRNNSpeTasNet(64, feature_size=256, hidden_size=256, chunk_length=16, n_repeats=1).separation)
    ok = dict(feature_size=512, hidden_size=512, rnn_type="GRU", norm_type="gLN", activation_type="relu", hop_length=8, chunk_length=16)
    wide.check_supported(ok)
    for bad, what in ((dict(ok, hidden_size=528), "hidden_size 528"), (dict(ok, feature_size=1024), "160 KB"),
                      (dict(ok, rnn_type="QRNN"), "rnn_type"), (dict(ok, hop_length=17), "hop_length")):
        with pytest.raises(NotImplementedError, match=what):
            wide.check_supported(bad)
    wspe = dprnn_spe.DPRNNSpeTasNet(64, feature_size=256, hidden_size=256, chunk_length=16, n_repeats=1)
    wide.check_supported_spe(wspe.cfg)
    with pytest.raises(NotImplementedError, match="input_size 64"):
        wide.check_supported_spe(dict(wspe.cfg, input_size=48))
    with pytest.raises(NotImplementedError, match="speaker encoder"):
        wide.check_supported_spe(dict(wspe.cfg, P=512))
    with pytest.raises(NotImplementedError, match="kernel_size"):
        wide.check_supported_spe(dict(wspe.cfg, kernel_size=16))


def test_only_test_infrastructure_imports_the_oracle():
    """oracle/ is the checker: tests/ (its studies included), __graft_entry__.smoke() / build()'s import check and
    bench.py's cpu_baseline leg may import it; the package, the tools and everything else may not."""
    import ast
    allowed = {"bench.py": {"cpu_baseline"}, "__graft_entry__.py": {"build", "smoke"}}
    skip_dirs = {".git", "gpurun_out", "tests", "oracle", "__pycache__", "build"}
    offenders = []
    for top, dirs, files in os.walk(ROOT):
        dirs[:] = [d for d in dirs if d not in skip_dirs]
        for f in files:
            if not f.endswith(".py"):
                continue
            path = os.path.join(top, f)
            rel = os.path.relpath(path, ROOT)
            tree = ast.parse(open(path, encoding="utf-8").read(), filename=rel)

            def names(node):
                if isinstance(node, ast.Import):
                    return [a.name for a in node.names]
                if isinstance(node, ast.ImportFrom) and node.level == 0:
                    return [node.module or ""]
                return []

            def walk(node, fn):
                for child in ast.iter_child_nodes(node):
                    inner = child.name if isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef)) and fn is None else fn
                    for n in names(child):
                        if n == "oracle" or n.startswith("oracle."):
                            if inner not in allowed.get(rel, ()):
                                offenders.append(f"{rel}:{child.lineno} ({inner or 'module level'})")
                    walk(child, inner)

            walk(tree, None)
    assert not offenders, offenders
