"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the nn.Module mirror
and the C ABI, against (a) golden vectors produced by the reference classes and (b) the CPU oracle.

Tolerances (fp32 path): rel-L2 <= 1e-5 and max-abs <= 1e-5 * max|y| against the reference's fp32 CPU
output (its own fp32-vs-fp64 noise is ~2e-7 rel-L2), SI-SDR within 0.01 dB (BASELINE.json)."""
import numpy as np
import pytest
import torch

from conftest import build_model, golden_names, load_golden, model_from_meta, rel_l2
from oracle import dprnn_numpy as onp
from oracle import dprnn_torch_cpu as otc
from oracle.sisdr import si_sdr
from oracle.weights import make_mixture, make_sources, make_state

from tss_with_dprnn_amd import engine
from tss_with_dprnn_amd._lib import DprnnError

pytestmark = pytest.mark.gpu

REL_TOL = 1e-5
ABS_TOL = 1e-5
BSS_GPU_CASES = [n for n in golden_names("bss_wide") if not n.startswith("bss_wider")] + ["bss_full_b2"]     # bss_wider_*: test_gpu_wide.py


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _close(got, want, what=""):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all(), what
    r = rel_l2(got, want)
    m = np.abs(got - want).max() / (np.abs(want).max() + 1e-30)
    assert r <= REL_TOL and m <= 10 * ABS_TOL, f"{what}: rel-L2 {r:.3e}, max-abs/max {m:.3e}"
    return r


@pytest.mark.parametrize("name", BSS_GPU_CASES)
def test_tasnet_matches_reference_golden(name):
    g = load_golden(name)
    m = g["meta"]
    model, _, _ = build_model("bss", m["cfg"], m["seed"], _dev())
    with torch.no_grad():
        out = model(torch.from_numpy(g["mix"]).to(_dev()))
    torch.cuda.synchronize()
    assert out.shape == g["out"].shape
    _close(out.cpu().numpy(), g["out"], name)


def test_stage_taps_match_reference():
    g = load_golden("bss_wide_small")
    m = g["meta"]
    model, _, cfg = build_model("bss", m["cfg"], m["seed"], _dev())
    dbg = {}
    with torch.no_grad():
        model(torch.from_numpy(g["mix"]).to(_dev()), debug=dbg)
    # reference layouts: bottleneck [B,F,L]; blocks [B,F,K,S]; ours: [B,L,F] / [B,S,K,F]
    _close(dbg["bottleneck"].permute(0, 2, 1).cpu().numpy(), g["tap_bottleneck"], "bottleneck")
    for i in range(cfg["n_repeats"]):
        _close(dbg[f"block{i}"].permute(0, 3, 2, 1).cpu().numpy(), g[f"tap_block{i}"], f"block{i}")


def test_si_sdr_parity_full_size():
    """BASELINE.json: SI-SDR of the GPU output within +-0.01 dB of the CPU path on identical inputs."""
    g = load_golden("bss_full_b2")
    m = g["meta"]
    model, _, _ = build_model("bss", m["cfg"], m["seed"], _dev())
    with torch.no_grad():
        out = model(torch.from_numpy(g["mix"]).to(_dev())).cpu().numpy()
    tgt, _ = make_sources(2, 24000, seed=77)
    a = si_sdr(out, tgt)
    b = si_sdr(g["out"], tgt)
    assert np.abs(a - b).max() <= 0.01, (a, b)
    assert si_sdr(out, g["out"]).min() > 80.0        # GPU vs reference CPU output, dB


def test_batch_invariance_and_determinism():
    cfg = dict(chunk_length=16, hop_length=8, n_repeats=1)
    model, _, _ = build_model("bss", cfg, 5, _dev())
    x = torch.from_numpy(make_mixture(5, 333, seed=9)).to(_dev())
    with torch.no_grad():
        full = model(x)
        again = model(x)
        one = model(x[3:4])
    assert torch.equal(full, again), "forward is not run-to-run deterministic"
    assert one.shape == (1, 2, 333)
    _close(one[0].cpu().numpy(), full[3].cpu().numpy(), "batch invariance")


@pytest.mark.parametrize("B,T", [(1, 24000), pytest.param(3, 9000, marks=pytest.mark.slow)])      # (3, 9000): 11 s of CPU oracle between B = 1 and the B = 64 case below
def test_against_cpu_oracle_shipped_config(B, T):
    """Full-width, full-depth model vs the torch-CPU oracle on fresh seeded input."""
    model, sd, cfg = build_model("bss", dict(), 123, _dev())
    x = make_mixture(B, T, seed=4321)
    want = otc.tasnet_forward(otc.to_torch_state(sd), cfg, torch.from_numpy(x)).numpy()
    with torch.no_grad():
        got = model(torch.from_numpy(x).to(_dev())).cpu().numpy()
    _close(got, want, f"shipped B={B} T={T}")


@pytest.mark.parametrize("fast", [0, 1])
def test_both_activation_forms_stay_within_tolerance(fast):
    """Default = hardware exp/rcp forms (1); the libm forms (0) must agree with the reference as well."""
    import tss_with_dprnn_amd as pkg
    g = load_golden("bss_full_b2")
    m = g["meta"]
    model, _, _ = build_model("bss", m["cfg"], m["seed"], _dev())
    lib = pkg.load_library()
    lib.dprnn_set_fast_activations(fast)
    lib.dprnn_set_fast_head_activations(fast)
    try:
        with torch.no_grad():
            out = model(torch.from_numpy(g["mix"]).to(_dev())).cpu().numpy()
    finally:
        lib.dprnn_set_fast_activations(1)
        lib.dprnn_set_fast_head_activations(1)
    r = rel_l2(out, g["out"])
    print(f"activations fast={fast}: rel-L2 vs reference {r:.3e}")
    assert r <= REL_TOL


def test_fused_and_explicit_norm_agree():
    """engine.FUSE_NORM folds every norm+residual into the consumer's loader; the explicit pass must agree."""
    from tss_with_dprnn_amd import engine
    g = load_golden("bss_wide_k250")
    m = g["meta"]
    model, _, _ = build_model("bss", m["cfg"], m["seed"], _dev())
    x = torch.from_numpy(g["mix"]).to(_dev())
    old = engine.FUSE_NORM
    try:
        outs = []
        for f in (True, False):
            engine.FUSE_NORM = f
            with torch.no_grad():
                outs.append(model(x).cpu().numpy())
    finally:
        engine.FUSE_NORM = old
    _close(outs[0], g["out"], "fused")
    _close(outs[1], g["out"], "explicit")
    assert rel_l2(outs[0], outs[1]) < 1e-6


@pytest.mark.parametrize("mode,tol", [(1, 1e-6), (3, 1e-6)])
def test_optional_split_gemm_modes_full_size(mode, tol):
    """engine.GEMM_MODE 1 / 3 (fp32-class split gate GEMMs: six bf16 products, three products of f16 pieces) must sit at
    the fp32 path's own noise level against the reference output (the 3-product bf16 form, mode 2, left the tree in round 3) inside the 1e-5
    parity tolerance."""
    from tss_with_dprnn_amd import engine
    g = load_golden("bss_full_b2")
    m = g["meta"]
    model, _, _ = build_model("bss", m["cfg"], m["seed"], _dev())
    x = torch.from_numpy(g["mix"]).to(_dev())
    old = engine.GEMM_MODE, engine.SPLIT_MIN_BATCH
    engine.GEMM_MODE, engine.SPLIT_MIN_BATCH = mode, 0       # two utterances: force the split kernels at this size
    try:
        assert engine.effective_gemm_mode(2) == mode
        with torch.no_grad():
            out = model(x).cpu().numpy()
        engine.GEMM_MODE = 0
        with torch.no_grad():
            exact = model(x).cpu().numpy()
    finally:
        engine.GEMM_MODE, engine.SPLIT_MIN_BATCH = old
    r = rel_l2(out, g["out"])
    tgt, _ = make_sources(2, 24000, seed=77)
    d = np.abs(si_sdr(out, tgt) - si_sdr(g["out"], tgt)).max()
    print(f"gemm mode {mode}: rel-L2 vs reference {r:.2e} (exact fp32 path {rel_l2(exact, g['out']):.2e}), |dSI-SDR| {d:.1e} dB")
    assert r <= tol and d <= 0.01


def test_stream_groups_do_not_change_results():
    """Utterance groups on separate HIP streams (engine.STREAM_GROUPS) must be bit-identical to one stream."""
    from tss_with_dprnn_amd import engine
    model, _, _ = build_model("bss", dict(chunk_length=50, hop_length=25, n_repeats=2), 11, _dev())
    x = torch.from_numpy(make_mixture(19, 3000, seed=3)).to(_dev())
    old, old_min = engine.STREAM_GROUPS, engine.MIN_GROUP_BATCH
    engine.MIN_GROUP_BATCH = 4
    try:
        outs = []
        for n in (1, 2, 3):
            engine.STREAM_GROUPS = n
            with torch.no_grad():
                outs.append(model(x))
            torch.cuda.synchronize()
        assert len(engine._split(19)) == 3
    finally:
        engine.STREAM_GROUPS = old
    # not bit-identical by construction: the fp64 GroupNorm sums are accumulated in a different grouping
    # when the rows are split differently, which can flip the last bit of the fp32 mean / rstd
    for o in outs[1:]:
        d = (o - outs[0]).abs().max().item() / outs[0].abs().max().item()
        print("stream groups max-abs/max", d)
        assert d < 2e-6
    engine.STREAM_GROUPS = 3
    try:
        with torch.no_grad():
            again = model(x)
    finally:
        engine.STREAM_GROUPS, engine.MIN_GROUP_BATCH = old, old_min
    assert torch.equal(again, outs[2]), "multi-stream forward is not run-to-run deterministic"


@pytest.mark.parametrize("B,T", [(1, 2), (2, 3), (1, 100), (3, 251), (2, 1000), (65, 600), (1, 111920)])
def test_edge_lengths_and_batches_vs_cpu_oracle(B, T):
    """Utterances shorter than a chunk (L < K, a single frame), odd lengths, and a batch that splits unevenly
    into stream groups (65), and the longest Libri2Mix-test utterance (111 920 samples, 898 chunks: SURVEY.md 8d)
    -- shipped chunk geometry K=250/P=125, two blocks."""
    model, sd, cfg = build_model("bss", dict(n_repeats=2), 31, _dev())
    x = make_mixture(B, T, seed=T)
    want = otc.tasnet_forward(otc.to_torch_state(sd), cfg, torch.from_numpy(x)).numpy()
    with torch.no_grad():
        got = model(torch.from_numpy(x).to(_dev())).cpu().numpy()
    assert got.shape == want.shape
    _close(got, want, f"edge B={B} T={T}")


def test_input_validation_messages():
    import tss_with_dprnn_amd as pkg
    model, _, _ = build_model("bss", dict(n_repeats=1), 32, _dev())
    with pytest.raises(pkg._lib.DprnnError, match="float32"):
        model(torch.zeros(1, 500, device=_dev(), dtype=torch.float64))
    with pytest.raises(pkg._lib.DprnnError, match="batch, samples"):
        model(torch.zeros(500, device=_dev()))
    with pytest.raises(pkg._lib.DprnnError, match="shorter than the encoder kernel"):
        model(torch.zeros(1, 1, device=_dev()))
    x = torch.randn(4, 1200, device=_dev())[:, ::2]            # non-contiguous view is accepted (made contiguous)
    with torch.no_grad():
        a = model(x)
        b = model(x.contiguous())
    assert torch.equal(a, b)


def test_headline_batch_b64_full_size():
    """BASELINE.json configs[1] at its own size through the DEFAULT engine settings (the stream-group split and the
    sequence-tile heuristics bench.py runs with): 64 x 24 000 samples, shipped config.  Utterances 0, 15, 16 and 63
    (first / last of a 16-utterance group, first of the next) against the CPU oracle; every utterance against its own
    batch-of-one forward, which takes the small-batch tiles on a single stream (tolerance 2e-6: the two differ only in
    the grouping of the fp64 GroupNorm sums and the MFMA tile a sequence lands in)."""
    from tss_with_dprnn_amd import engine
    assert not any(engine.SEQ_TILE.values()) and engine.GEMM_MODE == engine.DEFAULT_GEMM_MODE and engine.GROUP_SIZES is None
    model, sd, cfg = build_model("bss", dict(), 0, _dev())
    B, T = 64, 24000
    x = make_mixture(B, T, seed=1234)
    xd = torch.from_numpy(x).to(_dev())
    with torch.no_grad():
        got = model(xd)
        torch.cuda.synchronize()
        assert got.shape == (B, 2, T) and bool(torch.isfinite(got).all())
        worst = 0.0
        for b in range(B):
            one = model(xd[b:b + 1])
            r = float(((one[0].double() - got[b].double()).norm() / one[0].double().norm()).item())
            worst = max(worst, r)
            assert r <= 2e-6, f"utterance {b}: batched vs batch-of-one rel-L2 {r:.3e}"
    got = got.cpu().numpy()
    tsd = otc.to_torch_state(sd)
    pick = [0, 15, 16, 63]
    want = otc.tasnet_forward(tsd, cfg, torch.from_numpy(x[pick])).numpy()
    r = _close(got[pick], want, "B=64 headline batch, utterances 0/15/16/63 vs CPU oracle")
    src = make_sources(len(pick), T, seed=78)[0]
    assert np.abs(si_sdr(got[pick], src) - si_sdr(want, src)).max() <= 0.01
    print(f"B=64 x 24000: rel-L2 vs CPU oracle {r:.2e}; worst batched-vs-single {worst:.2e}")


# every forward golden (grad_* are training steps).  Since round 4 that includes the widths BELOW the kernels' build -- the 14 "tiny"
# fixtures (input_size 8, feature / hidden 16, speaker encoder 16 / 24 / 12) and the *_narrow_* ones: they run as zero-padded twins
# (tss-with-dprnn_amd/padded.py), so they pin the HIP path too, not only the oracle.
# (widths ABOVE the build, *_wider*, run on the any-width kernels whatever the mode: tests/test_gpu_wide.py)
FORWARD_GOLDENS = [n for n in golden_names() if not n.startswith("grad_") and "_wider" not in n]


# In the opt-in modes 0 and 1 a representative third runs by default -- every full-size golden, the wide ones that differ in kernel path
# (cells, fusions, norms, trained-like weights, BatchNorm in train mode) -- and the tiny / narrow twins and the remaining wide variants
# are marked `slow` (the same kernels at other shapes; they all run in the default mode 3): the driver's `pytest -m gpu` has a time limit.
OPT_IN_MODE_GOLDENS = {"bss_full_b2", "bss_wide_small", "bss_wide_small_gln_relu", "bss_wide_hop5", "bss_wide_k250", "bss_narrow64_small",
                       "spe_full_b1", "spe_full_att_b1", "spe_wide_small_film", "spe_wide_small_cat_bntrain", "spe_narrow_att_bntrain",
                       "ira_full_b1", "ira_wide_small_cat", "bss_tiny_b1"}        # (the trained-like goldens have their own tests)
assert OPT_IN_MODE_GOLDENS <= set(FORWARD_GOLDENS)


def _mode_cases():
    for name in FORWARD_GOLDENS:
        for mode in (0, 1, 3):
            marks = () if mode == 3 or name in OPT_IN_MODE_GOLDENS else (pytest.mark.slow,)
            yield pytest.param(name, mode, marks=marks, id=f"{name}-{mode}")


@pytest.mark.parametrize("name,mode", list(_mode_cases()))
def test_every_golden_in_every_fp32_class_gate_gemm_mode(name, mode):
    """EVERY forward golden generated from the reference classes (BSS, Spe, IRA; LSTM cells) under each fp32-class arithmetic
    of the gate GEMM, forced at every size: 0 = exact fp32 MFMA, 1 = six bf16 products, 3 = three products of f16 pieces
    (the default).  Same tolerances in all three (rel-L2 1e-5, |dSI-SDR| 0.01 dB).  GRU / Elman goldens always run the
    exact kernels (effective_gemm_mode), whatever the mode."""
    from tss_with_dprnn_amd import engine
    g = load_golden(name)
    m = g["meta"]
    model, _, _ = build_model(m["kind"], m["cfg"], m["seed"], _dev())
    if m.get("bn_training"):
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.train()
    old = engine.GEMM_MODE, engine.SPLIT_MIN_BATCH
    engine.GEMM_MODE, engine.SPLIT_MIN_BATCH = mode, 0
    try:
        assert engine.effective_gemm_mode(1) == mode
        with torch.no_grad():
            if m["kind"] == "bss":
                out = model(torch.from_numpy(g["mix"]).to(_dev()))
                logits = None
            else:
                out, logits = model(torch.from_numpy(g["mix"]).to(_dev()), torch.from_numpy(g["aux"]).to(_dev()),
                                    torch.tensor(float(g["aux_len"])))
        torch.cuda.synchronize()
        # the mode is process-wide library state: a forward hands the library back in the exact mode
        assert engine.Stage(_dev()).lib.dprnn_get_gemm_mode() == 0
    finally:
        engine.GEMM_MODE, engine.SPLIT_MIN_BATCH = old
    r = _close(out.cpu().numpy(), g["out"], f"{name} (gate GEMM mode {mode})")
    if logits is not None:
        assert rel_l2(logits.cpu().numpy(), g["logits"]) <= REL_TOL
    print(f"{name}: rel-L2 vs reference {r:.2e} in gate-GEMM mode {mode}")


@pytest.mark.parametrize("form", ["split", "serial"])
@pytest.mark.parametrize("name", ["bss_full_b2", "bss_wide_small", "bss_wide_hop5", "bss_wide_k250", "bss_narrow64_small", "spe_full_b1",
                                  "spe_wide_small_film", "ira_full_b1"])
def test_goldens_with_the_linear_folded_in(name, form, monkeypatch):
    """The folded launch structures (engine.LIN_FOLD, section 3a of DESIGN.md) are chosen from ~31 utterances up, the reference fixtures
    hold 1-3: here both forms are FORCED on them (64-sequence tiles, whatever the batch) -- the time-split form (odd and even T, the
    shipped K = 250 / S = 194 among them) and the direction-serial form against the reference's own outputs, at the golden tolerances."""
    monkeypatch.setattr(engine, "LIN_FOLD", form)
    monkeypatch.setattr(engine, "GEMM_MODE", 3)
    for k in ("intra", "inter", "serial_intra", "serial_inter"):
        monkeypatch.setitem(engine.SEQ_TILE, k, 2064)
    g = load_golden(name)
    m = g["meta"]
    model, _, _ = build_model(m["kind"], m["cfg"], m["seed"], _dev())
    calls = []
    real = engine.Stage.call

    def spy(self, fn, *a):
        calls.append(fn)
        return real(self, fn, *a)
    monkeypatch.setattr(engine.Stage, "call", spy)
    with torch.no_grad():
        if m["kind"] == "bss":
            out, logits = model(torch.from_numpy(g["mix"]).to(_dev())), None
        else:
            out, logits = model(torch.from_numpy(g["mix"]).to(_dev()), torch.from_numpy(g["aux"]).to(_dev()), torch.tensor(float(g["aux_len"])))
    torch.cuda.synchronize()
    want_fn = "dprnn_rnn_lin_split_fwd" if form == "split" else "dprnn_rnn_lin_fwd"
    assert want_fn in calls and "dprnn_proj_stats_hp_fwd" not in calls and "dprnn_proj_stats_fwd" not in calls, sorted(set(calls))
    r = _close(out.cpu().numpy(), g["out"], f"{name} ({form} fold)")
    if logits is not None:
        assert rel_l2(logits.cpu().numpy(), g["logits"]) <= REL_TOL
    print(f"{name}: rel-L2 vs reference {r:.2e} with the Linear folded in ({form})")
    engine.drop_graphs()


@pytest.mark.parametrize("kind", ["bss", "spe"])
@pytest.mark.parametrize("where", ["residual_stream", "bottleneck", "input_nan"])
def test_values_beyond_the_f16_range_fall_back_to_bf16_pieces(kind, where):
    """The default arithmetic (two f16 pieces of 2^4 x) cannot hold |x| > 4062.  A norm weight of 3000 drives the residual stream
    (or the bottleneck's input rows) past that: the conversion sites raise their device flags, the forward notices before it
    returns, runs again in mode 1 (three bf16 pieces: fp32's exponent range) and the caller gets the fp32-class result -- checked
    against the CPU oracle at the usual tolerance.  With DPRNN_ON_SATURATION=raise the same input is an error.  A NaN sample is
    flagged too and comes back as NaN (as the reference would return it), never as a clamped finite number."""
    import warnings
    from tss_with_dprnn_amd.synthetic import make_mixture
    cfg = dict(chunk_length=50, hop_length=25, n_repeats=2)
    if kind == "spe":
        cfg.update(fusion_type="cat")
    model, sd, cfg = build_model(kind, cfg, 21, _dev())
    key = "separation.dprnn_blocks.0.intra_norm.weight" if where != "bottleneck" else "separation.bottleneck.0.weight"
    if where != "input_nan":
        sd = dict(sd)
        sd[key] = (np.asarray(sd[key]) * 3000.0).astype(np.float32)
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    x = make_mixture(3, 3100, seed=8)
    if where == "input_nan":
        x[1, 1234] = np.nan
    args = [torch.from_numpy(x).to(_dev())]
    if kind == "spe":
        aux = make_mixture(3, 2900, seed=9)
        args += [torch.from_numpy(aux).to(_dev()), torch.tensor(2900.0)]
    assert engine.current_mode() == 3
    before = dict(engine.SATURATION)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        got = model(*args)
    assert engine.SATURATION["fallbacks"] == before["fallbacks"] + 1 and engine.SATURATION["checks"] == before["checks"] + 1
    flags = engine.SATURATION["last_flags"]
    assert flags[2] == 1 if where == "bottleneck" else (flags[0] == 1 or where == "input_nan")
    assert engine.Stage(_dev()).lib.dprnn_get_gemm_mode() == 0 and engine.current_mode() == 3
    got = (got[0] if kind == "spe" else got).cpu().numpy()
    if kind == "bss":
        want = otc.tasnet_forward(otc.to_torch_state(sd), cfg, torch.from_numpy(x)).numpy()
    else:
        want = otc.spe_forward(otc.to_torch_state(sd), cfg, torch.from_numpy(x), torch.from_numpy(aux), torch.tensor(2900.0))[0].numpy()
    if where == "input_nan":
        assert np.isnan(got[1]).all() and np.isnan(want[1]).all()
        got, want = got[[0, 2]], want[[0, 2]]
    r = _close(got, want, f"{kind} / {where} after the fall-back")
    # the clamped arithmetic alone would have been wrong: that is what the flag is for (with the constant 2^4 scale of rounds 2 / 3:
    # the model's activation-scale record has adapted by now, the same input would fit)
    old, old_ad = engine.ON_SATURATION, engine.ADAPTIVE_SCALE
    engine.ON_SATURATION, engine.ADAPTIVE_SCALE = "raise", False
    try:
        with torch.no_grad(), pytest.raises(DprnnError, match="beyond its range"):
            model(*args)
    finally:
        engine.ON_SATURATION, engine.ADAPTIVE_SCALE = old, old_ad
    # the same model in mode 1 from the start: no detection needed, same answer
    with torch.no_grad(), engine.use_gemm_mode(1):
        again = model(*args)
    again = (again[0] if kind == "spe" else again).cpu().numpy()
    if where == "input_nan":
        again = again[[0, 2]]
    assert rel_l2(again, got) < 1e-6
    print(f"{kind} / {where}: flags {flags}, rel-L2 vs CPU oracle after the fall-back {r:.2e}")


@pytest.mark.parametrize("where", ["residual_stream", pytest.param("bottleneck", marks=pytest.mark.slow)])      # 12-14 s each: the second site by -m "gpu and slow"
def test_a_model_beyond_the_default_range_pays_the_repeat_once(where):
    """Round 4 (unbounded affine gains, dprnn.py:76-77): a norm gain of 3000 puts the residual stream / the bottleneck rows beyond
    |x| = 4062.  The FIRST forward saturates and is repeated in mode 1; it leaves its maxima in the model's activation-scale
    record, so forwards 2 .. 10 run in mode 3 under a scale that fits: exactly one fall-back over ten forwards, every one of the
    ten within 1e-5 of the CPU oracle, and forwards 3 .. 10 (graph replays) within 1.1 x of the same model's time with ordinary
    gains."""
    import time
    import warnings
    from tss_with_dprnn_amd.synthetic import make_mixture
    over = dict(chunk_length=50, hop_length=25, n_repeats=2)
    key = "separation.dprnn_blocks.0.intra_norm.weight" if where == "residual_stream" else "separation.bottleneck.0.weight"
    xs = [make_mixture(4, 6000, seed=40 + i) for i in range(10)]

    def timed_forwards(model):
        outs, ts = [], []
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            for x in xs:
                xd = torch.from_numpy(x).to(_dev())
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                o = model(xd)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
                outs.append(o.cpu().numpy())
        return outs, ts
    engine.drop_graphs()
    plain, _, _ = build_model("bss", over, 21, _dev())
    _, t_plain = timed_forwards(plain)
    model, sd, cfg = build_model("bss", over, 21, _dev())
    sd = dict(sd)
    sd[key] = (np.asarray(sd[key]) * 3000.0).astype(np.float32)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    before = dict(engine.SATURATION)
    outs, t_big = timed_forwards(model)
    assert engine.SATURATION["fallbacks"] == before["fallbacks"] + 1, "exactly one fall-back over ten forwards"
    assert engine.SATURATION["checks"] == before["checks"] + 10
    tsd = otc.to_torch_state(sd)
    for i in (0, 1, 2, 9):             # the saturated forward (repeated in mode 1), the first ones under the adapted scale, the last
        want = otc.tasnet_forward(tsd, cfg, torch.from_numpy(xs[i])).numpy()
        _close(outs[i], want, f"forward {i + 1} of 10 ({where})")
    for i in range(2, 10):             # same input sizes, same arithmetic from forward 3 on: all finite, none degenerate
        assert np.isfinite(outs[i]).all() and np.abs(outs[i]).max() > 0
    s = engine.scale_exponents(model)
    assert min(s.values()) < 4 and not model.__dict__.get("_dprnn_pin_mode1", False), s
    # (a repeat would double or triple the time; forwards of 2 ms carry host jitter, so the fastest of each series is compared)
    ratio = float(min(t_big[2:]) / min(t_plain[2:]))
    assert ratio < 1.25, (ratio, t_big, t_plain)
    print(f"{where}: scale exponents {sorted(set(s.values()))}, forwards 3..10 at {ratio:.3f} x the unsaturated time")
    engine.drop_graphs()


def test_the_activation_scale_comes_down_again_after_an_outlier():
    """the sticky maxima only ever grew -- one forward with large values cost every later forward of that model up to 12 bits
    of its second f16 piece.  Round 5 (csrc/api.hip: dprnn_scale_join): a site whose forward maximum stays 16 x below its sticky
    maximum for eight consecutive forwards falls back to the largest value of that streak (never below the default range); every
    forward reads its scales from its own snapshot of the record.  A norm gain x 3000 for two forwards, then the normal weights:
    the exponents are back at 4 after eight more forwards, every output is fp32-class throughout, and engine.reset_scale does the same at once."""
    import warnings
    from tss_with_dprnn_amd.synthetic import make_mixture
    over = dict(chunk_length=50, hop_length=25, n_repeats=2)
    key = "separation.dprnn_blocks.0.intra_norm.weight"
    model, sd, cfg = build_model("bss", over, 21, _dev())
    x = make_mixture(3, 5000, seed=77)
    xd = torch.from_numpy(x).to(_dev())
    want = otc.tasnet_forward(otc.to_torch_state(sd), cfg, torch.from_numpy(x)).numpy()
    prm = dict(model.named_parameters())[key]
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        prm.mul_(3000.0)
        for _ in range(2):
            model(xd)
        assert min(engine.scale_exponents(model).values()) < 4
        prm.div_(3000.0)
        outs = [model(xd).cpu().numpy() for _ in range(10)]
    for i, o in enumerate(outs):
        _close(o, want, f"forward {i + 1} after the outlier")
    assert set(engine.scale_exponents(model).values()) == {4}, engine.scale_exponents(model)
    assert not model.__dict__.get("_dprnn_pin_mode1", False)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        prm.mul_(3000.0)
        model(xd); model(xd)
        prm.div_(3000.0)
    assert min(engine.scale_exponents(model).values()) < 4
    engine.reset_scale(model)
    assert set(engine.scale_exponents(model).values()) == {4}
    engine.drop_graphs()


def test_repeated_saturation_pins_the_model_to_mode_1():
    """Values no scale can hold (here a NaN sample: flagged where every forward begins, the record ignores it) saturate every
    forward: after two in a row the model runs in mode 1 directly -- one warning, no further repeats -- until engine.unpin."""
    import warnings
    from tss_with_dprnn_amd.synthetic import make_mixture
    model, _, _ = build_model("bss", dict(chunk_length=50, hop_length=25, n_repeats=2), 22, _dev())
    x = make_mixture(2, 3000, seed=3)
    x[1, 100] = np.nan
    xd = torch.from_numpy(x).to(_dev())
    before = dict(engine.SATURATION)
    with torch.no_grad(), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        outs = [model(xd).cpu().numpy() for _ in range(5)]
    assert engine.SATURATION["fallbacks"] == before["fallbacks"] + 2 and model.__dict__["_dprnn_pin_mode1"]
    assert sum("now runs in gate-GEMM mode 1" in str(m.message) for m in w) == 1
    for o in outs:
        assert np.isnan(o[1]).all() and np.isfinite(o[0]).all() and rel_l2(o[0], outs[0][0]) < 1e-6
    engine.unpin(model)
    good = torch.from_numpy(make_mixture(2, 3000, seed=3)).to(_dev())
    with torch.no_grad():
        model(good)
    assert engine.SATURATION["fallbacks"] == before["fallbacks"] + 2 and not model.__dict__["_dprnn_pin_mode1"]


def test_narrow_models_run_as_zero_padded_twins_and_follow_their_parameters():
    """padded.py: a model narrower than the kernels' build is computed on a twin at the built widths whose extra rows / columns are
    zero.  The twin follows the model (an in-place parameter update shows in the next forward, graph replay included), train-mode
    BatchNorm statistics land in the NARROW model's buffers as torch's own modules would leave them, widths above the build raise.
    (Training a narrow model runs on the twin as well: tests/test_gpu_train.py, grad_*_narrow_* against the reference's gradients.)"""
    import warnings
    from tss_with_dprnn_amd import padded
    over = dict(input_size=48, feature_size=64, hidden_size=64, O=64, P=96, embeddings_size=40, chunk_length=16, hop_length=8, n_repeats=1,
                fusion_type="cat", norm_type="ln")
    model, sd, cfg = build_model("spe", over, 64, _dev())
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.train()
    mix, aux = make_mixture(3, 400, seed=1), make_mixture(3, 350, seed=2)
    args = (torch.from_numpy(mix).to(_dev()), torch.from_numpy(aux).to(_dev()), torch.tensor(350.0))
    ref_bufs = {k: v.clone() for k, v in model.named_buffers()}
    with torch.no_grad():
        est, logits = model(*args)
    want, wlog = otc.spe_forward(otc.to_torch_state(sd), cfg, torch.from_numpy(mix), torch.from_numpy(aux), torch.tensor(350.0), training_bn=True)
    _close(est.cpu().numpy(), want.numpy(), "narrow Spe, BatchNorm in train mode")
    assert rel_l2(logits.cpu().numpy(), wlog.numpy()) <= REL_TOL
    # the running statistics moved in the caller's model (real channels only), one step of momentum 0.1 from (0, 1)
    bn = model.separation.spk_encoder[2].batch_norm1
    assert int(bn.num_batches_tracked) == 1 and bn.running_mean.shape == (64,)
    assert not torch.equal(bn.running_mean, ref_bufs["separation.spk_encoder.2.batch_norm1.running_mean"])
    assert float((bn.running_var - 0.9).abs().max()) < 0.2 + 1.0                     # 0.9 * 1 + 0.1 * batch variance, finite
    tw = model.__dict__["_dprnn_twin"]
    assert isinstance(tw, padded.Twin) and tw.model.cfg["feature_size"] == 128 and tw.model.cfg["input_size"] == 64
    # eval mode, replays, and an optimizer-style in-place update of the narrow parameters
    model.eval()
    with torch.no_grad():
        a = [model(*args)[0].cpu().numpy() for _ in range(3)]
        for prm in model.parameters():
            prm.mul_(1.01)
        b = model(*args)[0].cpu().numpy()
    assert np.array_equal(a[1], a[2]) and rel_l2(a[0], a[1]) < 1e-6
    sd2 = {k: (np.asarray(v) * np.float32(1.01) if k in dict(model.named_parameters()) else np.asarray(v)) for k, v in sd.items()}
    sd2.update({k: v.detach().cpu().numpy() for k, v in model.named_buffers()})
    want2 = otc.spe_forward(otc.to_torch_state(sd2), cfg, torch.from_numpy(mix), torch.from_numpy(aux), torch.tensor(350.0))[0].numpy()
    _close(b, want2, "narrow Spe after an in-place parameter update")
    assert rel_l2(b, a[0]) > 1e-4
    # beyond the build: inference runs on the any-width kernels (tests/test_gpu_wide.py), training at such widths names the limit
    import tss_with_dprnn_amd as pkg
    big = pkg.DPRNNTasNet(64, 256, 256, 16, 2, 8, 1).to(_dev()).eval()
    with torch.no_grad():
        assert big(torch.from_numpy(make_mixture(1, 300, seed=3)).to(_dev())).shape == (1, 2, 300)
    with pytest.raises(NotImplementedError, match="above the kernels' build"):
        big(torch.from_numpy(make_mixture(1, 300, seed=3)).to(_dev()))
    model.train()
    est, logits = model(*args)                       # under autograd: the training kernels on the twin
    assert est.requires_grad and logits.requires_grad and est.shape == (3, 400)
    engine.drop_graphs()


@pytest.mark.parametrize("mode", [0, 3])
def test_narrow_twin_with_several_utterance_groups(mode):
    """A zero-padded twin's GroupNorm sums are rescaled by built / real channels (engine._count_scale).  With more than one utterance
    group (13 utterances: two pipelines on side streams in every mode) that factor has to be applied on the group's own stream, after
    its sums exist and before its next reader, and to that group's utterances only -- round 4 applied it on the main stream to the
    rest of the batch (silently wrong norms from B = 9 / 11 up).  Eager and graph replay, against the CPU oracle."""
    over = dict(input_size=48, feature_size=64, hidden_size=64, chunk_length=16, hop_length=8, n_repeats=2, norm_type="ln")
    model, sd, cfg = build_model("bss", over, 65, _dev())
    B = 13
    mix = make_mixture(B, 700, seed=11)
    want = otc.tasnet_forward(otc.to_torch_state(sd), cfg, torch.from_numpy(mix)).numpy()
    x = torch.from_numpy(mix).to(_dev())
    assert len(engine.group_parts(B, mode)) > 1
    old = engine.GRAPHS
    try:
        for graphs in (False, True):
            engine.GRAPHS = graphs
            with torch.no_grad(), engine.use_gemm_mode(mode):
                outs = [model(x).cpu().numpy() for _ in range(3 if graphs else 1)]
            for o in outs:
                _close(o, want, f"narrow BSS twin, B = {B}, mode {mode}, graphs {graphs}")
    finally:
        engine.GRAPHS = old
        engine.drop_graphs()


def test_in_range_forwards_do_not_fall_back():
    model, _, _ = build_model("bss", dict(chunk_length=50, hop_length=25, n_repeats=2), 22, _dev())
    from tss_with_dprnn_amd.synthetic import make_mixture
    before = dict(engine.SATURATION)
    with torch.no_grad():
        for scale in (1.0, 1e-3, 30.0):
            model(torch.from_numpy(make_mixture(2, 3000, seed=1) * np.float32(scale)).to(_dev()))
    assert engine.SATURATION["fallbacks"] == before["fallbacks"] and engine.SATURATION["checks"] == before["checks"] + 3


def test_two_threads_two_modes():
    """The gate-GEMM mode is per thread (C library and engine.use_gemm_mode) and packed weights carry their format: two host
    threads drive two models in different modes at the same time, each on its own stream, and get what they get alone."""
    import threading
    from tss_with_dprnn_amd.synthetic import make_mixture
    cfg = dict(chunk_length=50, hop_length=25, n_repeats=2)
    models = [build_model("bss", cfg, 40 + i, _dev())[0] for i in range(2)]
    xs = [torch.from_numpy(make_mixture(3, 4000, seed=60 + i)).to(_dev()) for i in range(2)]
    modes = [0, 3]
    alone = []
    for m, x, mode in zip(models, xs, modes):
        with torch.no_grad(), engine.use_gemm_mode(mode):
            alone.append(m(x).cpu().numpy())
    results, errors = [None, None], []
    start = threading.Barrier(2)

    def work(i):
        try:
            stream = torch.cuda.Stream(device=_dev())
            start.wait()
            outs = []
            with torch.no_grad(), torch.cuda.stream(stream), engine.use_gemm_mode(modes[i]):
                for _ in range(6):
                    outs.append(models[i](xs[i]))
                    assert engine.current_mode() == modes[i]
            stream.synchronize()
            results[i] = [o.cpu().numpy() for o in outs]
        except Exception as e:       # noqa: BLE001 -- reported by the main thread
            errors.append(e)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for i in range(2):
        for o in results[i]:
            assert np.array_equal(o, alone[i]), f"thread {i} (mode {modes[i]}) saw another thread's mode"
    assert engine.current_mode() == engine.DEFAULT_GEMM_MODE


# ---- trained-like statistics (tests/golden/make_golden.py:TRAINED_CASES, generated from the reference classes): heavy-tailed
# LSTM rows, forget-gate bias +1..+3, norm gains far from 1, inputs at RMS 1 and 1e-3 -- where an arithmetic that leans on benign
# magnitudes would show it.  A DPRNN with such weights is close to a chaotic map: every fixture carries the distance of the
# reference's own fp32 output from the same reference module in float64 (gap) and that float64 output.  Two bars, in EVERY
# fp32-class mode (0 exact fp32 MFMA, 1 bf16 x 6, 3 f16 pieces -- the default), forced at this batch size:
#   * against the reference's fp32 output: the unchanged REL_TOL (1e-5), or 3 x gap where the reference's own noise is larger
#     (only the literal x1..x30 / [0.2, 8] recipe on the small model: gap 5e-5);
#   * against the reference's float64 output: no further than 2.5 x the reference's own fp32 distance from it.
# And none of them may have needed the saturation fall-back: these magnitudes are inside the f16-pieces range.
@pytest.mark.parametrize("mode", [0, 1, 3])
@pytest.mark.parametrize("name", golden_names("bss", trained=True) + golden_names("spe", trained=True) + golden_names("ira", trained=True))
def test_trained_like_weights_in_every_fp32_class_mode(name, mode):
    g = load_golden(name)
    m = g["meta"]
    gap = m["ref_fp32_vs_fp64_rel_l2"]
    model, _, _ = model_from_meta(m, _dev())
    old = engine.GEMM_MODE, engine.SPLIT_MIN_BATCH
    engine.GEMM_MODE, engine.SPLIT_MIN_BATCH = mode, 0
    before = dict(engine.SATURATION)
    try:
        assert engine.effective_gemm_mode(1) == mode
        with torch.no_grad():
            if m["kind"] == "bss":
                out = model(torch.from_numpy(g["mix"]).to(_dev())).cpu().numpy()
            else:           # the speaker-conditioned families: estimate and speaker logits (dprnn_spe.py:314-327, dprnn_spe_ira.py:179-190)
                est, logits = model(torch.from_numpy(g["mix"]).to(_dev()), torch.from_numpy(g["aux"]).to(_dev()), torch.tensor(float(g["aux_len"])))
                out = est.cpu().numpy()
                assert rel_l2(logits.cpu().numpy(), g["logits"]) <= max(REL_TOL, 3 * gap)
    finally:
        engine.GEMM_MODE, engine.SPLIT_MIN_BATCH = old
    assert engine.SATURATION["fallbacks"] == before["fallbacks"], engine.SATURATION
    assert np.isfinite(out).all()
    r32, r64 = rel_l2(out, g["out"]), rel_l2(out, g["out64"])
    print(f"{name} mode {mode}: rel-L2 vs reference fp32 {r32:.2e}, vs reference float64 {r64:.2e} (reference fp32 vs float64: {gap:.2e}); "
          f"max |residual stream| {max(m['residual_stream_absmax_after_block']):.0f}")
    assert r32 <= max(REL_TOL, 3 * gap), (r32, gap)
    assert r64 <= max(3e-6, 2.5 * gap), (r64, gap)


# ---- HIP graphs (engine._graphed): the second forward of a (model, shapes, mode, knobs) key is captured, later ones replay
@pytest.mark.parametrize("kind", ["bss", "spe_bntrain", "ira"])
def test_graph_replay_equals_eager_forward(kind):
    """Eager, captured and replayed forwards of the same input agree bit for bit; a replay with NEW input values and with weights
    changed in place (same parameter storage) equals the eager forward of that input with those weights; a BatchNorm in train mode
    leaves the same running statistics behind; a NaN sample is still noticed (the graph's own saturation flags) and comes back NaN."""
    from tss_with_dprnn_amd.synthetic import make_mixture
    over = dict(chunk_length=50, hop_length=25, n_repeats=2)
    if kind != "bss":
        over.update(fusion_type="cat", norm_type="ln")
    model, sd, cfg = build_model("bss" if kind == "bss" else ("ira" if kind == "ira" else "spe"), over, 91, _dev())
    if "bntrain" in kind:
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.train()
    B, T = 3, 3300

    def inputs(seed):
        a = [torch.from_numpy(make_mixture(B, T, seed=seed)).to(_dev())]
        if kind != "bss":
            a += [torch.from_numpy(make_mixture(B, 3000, seed=seed + 50)).to(_dev()), torch.tensor(3000.0)]
        return a

    def first(o):
        return (o[0] if isinstance(o, tuple) else o).cpu().numpy()

    def bn_state():
        return {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
    assert engine.GRAPHS
    engine.drop_graphs()
    stats0 = dict(engine.GRAPH_STATS)
    bn0 = bn_state()
    with torch.no_grad():
        outs = [first(model(*inputs(1))) for _ in range(4)]                 # eager, capture + replay, replay, replay
        bn_graph = bn_state()
        y_new = first(model(*inputs(2)))                                     # replay with other input values
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1.01)                                                 # in place: the graph reads the same storage
        y_w = first(model(*inputs(2)))
    assert engine.GRAPH_STATS["captures"] == stats0["captures"] + 1 and engine.GRAPH_STATS["replays"] >= stats0["replays"] + 5
    if "bntrain" not in kind:
        for o in outs[1:]:
            assert np.array_equal(o, outs[0])
    old = engine.GRAPHS
    engine.GRAPHS = False
    try:
        with torch.no_grad():
            want_w = first(model(*inputs(2)))
            for p in model.parameters():
                p.div_(1.01)
            model.load_state_dict({**model.state_dict(), **bn0})
            eager = [first(model(*inputs(1))) for _ in range(4)]
            bn_eager = bn_state()
            want_new = first(model(*inputs(2)))
    finally:
        engine.GRAPHS = old
    for a, b in zip(outs, eager):
        assert rel_l2(a, b) < 1e-6
    for k in bn_eager:
        assert torch.allclose(bn_graph[k].float(), bn_eager[k].float(), rtol=1e-5, atol=1e-7), k
    assert rel_l2(y_w, want_w) < 1e-6
    if "bntrain" not in kind:          # (train-mode BatchNorm: y_new saw different running statistics than want_new)
        assert rel_l2(y_new, want_new) < 1e-6
    # a NaN sample through the captured graph (bss: the key of inputs(1) is cached again after the eager block above)
    if kind == "bss":
        with torch.no_grad():
            model(*inputs(1)); model(*inputs(1))
            bad = inputs(3)
            bad[0][1, 77] = float("nan")
            before = dict(engine.SATURATION)
            got = first(model(*bad))
        assert engine.SATURATION["fallbacks"] == before["fallbacks"] + 1
        assert np.isnan(got[1]).all() and np.isfinite(got[[0, 2]]).all()
    engine.drop_graphs()


def test_graph_cache_evicts_before_it_captures(monkeypatch):
    """More keys than GRAPH_CACHE_MAX in one process: the engine evicts the oldest graph BEFORE capturing the next (and returns its pool
    to the driver outside the capture), the cache never exceeds its bound, a key that comes back is captured again, and every replay
    equals the eager forward of the same input.  (The full-size walk that used to segfault in hipGraphLaunch: tools/probes/graph_churn.py.)"""
    from tss_with_dprnn_amd import engine
    model, _, _ = build_model("bss", dict(n_repeats=2, chunk_length=50, hop_length=25), 91, _dev())
    monkeypatch.setattr(engine, "GRAPH_CACHE_MAX", 2)
    assert engine.GRAPHS
    engine.drop_graphs()
    c0 = engine.GRAPH_STATS["captures"]
    xs = {B: torch.from_numpy(make_mixture(B, 2400, seed=B)).to(_dev()) for B in (1, 2, 3, 4)}
    want = {}
    old = engine.GRAPHS
    engine.GRAPHS = False
    try:
        with torch.no_grad():
            for B, x in xs.items():
                want[B] = model(x).cpu().numpy()
    finally:
        engine.GRAPHS = old
    with torch.no_grad():
        for B in (1, 2, 3, 4, 1, 3):                       # 1 and 3 come back after their graphs were evicted
            outs = [model(xs[B]).cpu().numpy() for _ in range(3)]            # eager or capture, then replays
            assert len(engine._GRAPH_CACHE) <= 2
            for o in outs:
                assert rel_l2(o, want[B]) < 1e-6
    assert engine.GRAPH_STATS["captures"] == c0 + 6
    engine.drop_graphs()
    assert len(engine._GRAPH_CACHE) == 0



def test_captured_forwards_share_one_pool_and_replay_in_any_order():
    """Round 4: every capture allocates from the engine's one memory pool, so four cached graphs hold the workspace of the LARGEST
    forward, not the sum; graphs that share addresses are serialised on the pool, so replays interleaved over two streams still equal
    the eager forwards."""
    from tss_with_dprnn_amd import engine
    model, _, _ = build_model("bss", dict(n_repeats=2), 91, _dev())
    assert engine.GRAPHS
    engine.drop_graphs()
    sizes = (4, 3, 2, 1)                                   # largest first: later captures fit into what the first one reserved
    xs = {B: torch.from_numpy(make_mixture(B, 8000, seed=B)).to(_dev()) for B in sizes}
    old = engine.GRAPHS
    engine.GRAPHS = False
    try:
        with torch.no_grad():
            want = {B: model(xs[B]).cpu().numpy() for B in sizes}
    finally:
        engine.GRAPHS = old
    torch.cuda.synchronize()
    engine.empty_cache()
    with torch.no_grad():
        model(xs[4]); model(xs[4])                         # eager, capture
        torch.cuda.synchronize()
        pool_after_first = engine._pool_reserved_bytes(engine._engine_pool(_dev()))
        for B in sizes[1:]:
            model(xs[B]); model(xs[B])
        torch.cuda.synchronize()
        pool_after_all = engine._pool_reserved_bytes(engine._engine_pool(_dev()))
        assert len(engine._GRAPH_CACHE) == 4
        assert pool_after_first > 0 and pool_after_all <= 1.15 * pool_after_first, (pool_after_first, pool_after_all)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        got = []
        for i, B in enumerate((1, 4, 2, 4, 3, 1, 4)):
            with torch.cuda.stream(s1 if i % 2 == 0 else s2):
                got.append((B, model(xs[B])))
        torch.cuda.synchronize()
    for B, o in got:
        assert rel_l2(o.cpu().numpy(), want[B]) < 1e-6, B
    engine.drop_graphs()
    assert not engine._POOLS


def test_failed_capture_and_short_memory_leave_the_forward_eager(monkeypatch):
    """A capture that raises (or that the device has no room for) must not reach the caller, and must not be attempted again on every
    call of that key: the forward stays on eager launches, with one warning."""
    import warnings
    from tss_with_dprnn_amd import engine
    model, _, _ = build_model("bss", dict(n_repeats=2, chunk_length=50, hop_length=25), 91, _dev())
    x = torch.from_numpy(make_mixture(2, 2400, seed=4)).to(_dev())
    engine.drop_graphs()
    old = engine.GRAPHS
    engine.GRAPHS = False
    try:
        with torch.no_grad():
            want = model(x).cpu().numpy()
    finally:
        engine.GRAPHS = old
    calls = {"n": 0}
    real = engine._Graphed

    class Boom(real):
        def __init__(self, *a, **k):
            calls["n"] += 1
            raise RuntimeError("capture refused (test)")
    monkeypatch.setattr(engine, "_Graphed", Boom)
    f0 = engine.GRAPH_STATS["capture_failures"]
    with torch.no_grad(), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        outs = [model(x).cpu().numpy() for _ in range(5)]
    assert calls["n"] == 1 and engine.GRAPH_STATS["capture_failures"] == f0 + 1
    assert sum("stays on eager launches" in str(m.message) for m in w) == 1
    for o in outs:
        assert rel_l2(o, want) < 1e-6
    monkeypatch.setattr(engine, "_Graphed", real)
    engine.drop_graphs()
    # not enough free memory for the forward's workspace: no capture either
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda *a, **k: (1 << 20, 1 << 40))
    k0, c0 = engine.GRAPH_STATS["kept_eager_for_memory"], engine.GRAPH_STATS["captures"]
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        outs = [model(x).cpu().numpy() for _ in range(4)]
    assert engine.GRAPH_STATS["kept_eager_for_memory"] == k0 + 1 and engine.GRAPH_STATS["captures"] == c0
    for o in outs:
        assert rel_l2(o, want) < 1e-6
    engine.drop_graphs()


def test_one_graph_serves_inference_mode_and_no_grad():
    """Static inputs cloned under torch.inference_mode() would be inference tensors (a replay under plain no_grad could not copy into
    them) and torch's generator state created at a first capture under inference mode breaks later captures outside it: the engine
    always captures outside inference mode, and the one graph replays under both."""
    from tss_with_dprnn_amd import engine
    model, _, _ = build_model("bss", dict(n_repeats=2, chunk_length=50, hop_length=25), 91, _dev())
    x = torch.from_numpy(make_mixture(2, 2400, seed=4)).to(_dev())
    engine.drop_graphs()
    c0, f0 = engine.GRAPH_STATS["captures"], engine.GRAPH_STATS["capture_failures"]
    with torch.inference_mode():
        a = [model(x).cpu().numpy() for _ in range(3)]
    with torch.no_grad():
        b = [model(x).cpu().numpy() for _ in range(3)]
    with torch.inference_mode():
        a.append(model(x).cpu().numpy())
    assert engine.GRAPH_STATS["captures"] == c0 + 1 and engine.GRAPH_STATS["capture_failures"] == f0
    for o in a + b:
        assert rel_l2(o, a[0]) < 1e-6
    engine.drop_graphs()
