"""GPU tests of the any-width path (tss-with-dprnn_amd/wide.py, csrc/wide.hip, include/dprnn_mi355x_wide.h): DPRNNTasNet models WIDER than
the fused kernels' build.  Kernel level: every entry point through ctypes against plain torch fp32 on the same inputs.  Model level: the
reference's own outputs (tests/golden/bss_wider_*.npz, written by make_golden.py from the reference classes) and the CPU oracle at the
shipped chunk geometry.  Tolerances as everywhere on this path: rel-L2 <= 1e-5, |dSI-SDR| <= 0.01 dB."""
import numpy as np
import pytest
import torch

from conftest import build_model, golden_names, load_golden, rel_l2
from oracle import dprnn_torch_cpu as otc
from oracle.sisdr import si_sdr
from oracle.weights import make_mixture, make_sources

pytestmark = pytest.mark.gpu
REL_TOL = 1e-5
WIDER = golden_names("bss_wider")


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _stage():
    from tss_with_dprnn_amd import wide
    return wide._Launcher(_dev())


def _p(t):
    return 0 if t is None else t.data_ptr()


def _rel(a, b):
    return rel_l2(a.detach().cpu().numpy(), b.detach().cpu().numpy())


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("M,K,O", [(1, 1, 1), (63, 17, 5), (64, 16, 64), (129, 70, 133), (1000, 256, 512), (257, 400, 96), (5000, 131, 2)])
def test_gemm_plain(M, K, O):
    g = torch.Generator().manual_seed(M * 7 + K)
    A = torch.randn(M, K, generator=g).to(_dev())
    W = torch.randn(O, K, generator=g).to(_dev())
    b = torch.randn(O, generator=g).to(_dev())
    st = _stage()
    got = st.gemm(A, W, b, M, K, O)
    want = (A.double() @ W.double().t() + b.double()).float()
    assert _rel(got, want) <= 2e-6
    got = st.gemm(A, W, None, M, K, O, act=1)
    assert _rel(got, torch.relu(A.double() @ W.double().t()).float()) <= 2e-6


@pytest.mark.parametrize("act", [0, 1, 2, 3])
def test_gemm_with_every_option(act):
    """Per-utterance input affine + PReLU in the loader, bias, activation, the row-mapped multiplier and the fp64 sums of the outputs."""
    B, L, rep, K, O = 3, 37, 2, 70, 133
    rows_per_utt = rep * L
    M = B * rows_per_utt
    g = torch.Generator().manual_seed(11 + act)
    A = torch.randn(M, K, generator=g).to(_dev())
    W = (torch.randn(O, K, generator=g) / K ** 0.5).to(_dev())
    bias = torch.randn(O, generator=g).to(_dev())
    sc = (torch.rand(B, K, generator=g) + 0.5).to(_dev())
    sh = torch.randn(B, K, generator=g).to(_dev())
    pw = torch.tensor([0.25]).to(_dev())
    mul = torch.randn(B * L, O, generator=g).to(_dev())
    stats = torch.zeros(B, 2, dtype=torch.float64, device=_dev())
    st = _stage()
    got = st.gemm(A, W, bias, M, K, O, sc=sc, sh=sh, rows_per_utt=rows_per_utt, prelu=pw, act=act, mul=mul, mul_L=L, mul_rep=rep, stats=stats)
    a = A.view(B, rows_per_utt, K).double() * sc.double()[:, None] + sh.double()[:, None]
    a = torch.where(a >= 0, a, 0.25 * a)
    y = a @ W.double().t() + bias.double()
    y = [y, torch.relu(y), torch.sigmoid(y), torch.tanh(y)][act]
    y = y.view(B, rep, L, O) * mul.double().view(B, 1, L, O)
    want = y.reshape(M, O)
    assert _rel(got, want.float()) <= 2e-6
    yg = got.double().view(B, -1)                       # the sums are those of the fp32 outputs (float partial sums of a lane's 16 values)
    assert torch.allclose(stats, torch.stack([yg.sum(1), (yg * yg).sum(1)], 1), rtol=1e-5, atol=2e-3)


def test_gemm_sums_when_a_strip_spans_utterances():
    B, rows_per_utt, K, O = 9, 5, 24, 20                     # five rows per utterance: every 16-row strip spans several
    M = B * rows_per_utt
    g = torch.Generator().manual_seed(3)
    A = torch.randn(M, K, generator=g).to(_dev())
    W = torch.randn(O, K, generator=g).to(_dev())
    stats = torch.zeros(B, 2, dtype=torch.float64, device=_dev())
    st = _stage()
    got = st.gemm(A, W, None, M, K, O, rows_per_utt=rows_per_utt, stats=stats)
    y = got.double().view(B, -1)
    assert torch.allclose(stats, torch.stack([y.sum(1), (y * y).sum(1)], 1), rtol=1e-5, atol=2e-3)


def _torch_rnn(kind, F, H, bidir, seed):
    torch.manual_seed(seed)
    rnn = getattr(torch.nn, kind)(F, H, num_layers=1, batch_first=True, bidirectional=bidir)
    with torch.no_grad():
        for p in rnn.parameters():
            p.mul_(1.5)
    return rnn


@pytest.mark.parametrize("kind,F,H,bidir", [("LSTM", 256, 256, True), ("LSTM", 133, 131, True), ("GRU", 144, 136, False), ("GRU", 160, 200, True),
                                            ("RNN", 70, 67, True), ("LSTM", 128, 320, False), ("LSTM", 16, 512, True)])
@pytest.mark.parametrize("intra", [True, False])
def test_recurrence_against_torch(kind, F, H, bidir, intra):
    """dprnn_wide_rnn_pack + dprnn_wide_rnn_fwd against torch's own RNN modules on the CPU, both row addressings of the slab."""
    from tss_with_dprnn_amd import wide
    B, S, K = 2, 7, 11
    rnn = _torch_rnn(kind, F, H, bidir, 5)
    X = torch.randn(B, S, K, F, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        if intra:
            want = rnn(X.view(B * S, K, F))[0].reshape(B, S, K, -1)
        else:
            want = rnn(X.permute(0, 2, 1, 3).reshape(B * K, S, F))[0].reshape(B, K, S, -1).permute(0, 2, 1, 3)
    st = _stage()
    dev = _dev()
    cell = wide.CELL_ID[kind]
    rnn = rnn.to(dev)
    Wp, bv, ndir = wide._pack_rnn(st, rnn, cell, F, H, dev)
    Xd = X.to(dev).contiguous()
    out = torch.full((B * S * K, ndir * H), float("nan"), device=dev)
    if intra:
        args = (B * S, K, 1, K, 0, 1)
    else:
        args = (B * K, S, K, S * K, 1, K)
    st.call("dprnn_wide_rnn_fwd", cell, _p(Xd), _p(Wp), _p(bv), _p(out), ndir, F, H, *args)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    assert _rel(out.view(B, S, K, -1), want.contiguous()) <= 3e-6


@pytest.mark.parametrize("F,H,bidir", [(256, 256, True), (160, 200, True), (133, 131, False), (40, 16, True), (128, 320, True), (512, 512, False)])
@pytest.mark.parametrize("intra", [True, False])
def test_f16_pieces_lstm_recurrence_against_torch(F, H, bidir, intra):
    """dprnn_wide_absmax + dprnn_wide_lstm_pack_f16 + dprnn_wide_lstm_f16_fwd (csrc/wide_f16.hip: two f16 pieces per operand, three products,
    64 sequences per workgroup, scales from the measured maxima) against torch's LSTM on the CPU; inputs 50 x larger than unit scale in one
    half of the sequences, tiny in the other (the launch's one activation scale has to serve both)."""
    from tss_with_dprnn_amd import wide
    assert wide.f16_recurrence(0, F, H)
    B, S, K = 3, 7, 23
    rnn = _torch_rnn("LSTM", F, H, bidir, 5)
    X = torch.randn(B, S, K, F, generator=torch.Generator().manual_seed(9))
    X[0] *= 50.0
    X[2] *= 1e-3
    with torch.no_grad():
        if intra:
            want = rnn(X.view(B * S, K, F))[0].reshape(B, S, K, -1)
        else:
            want = rnn(X.permute(0, 2, 1, 3).reshape(B * K, S, F))[0].reshape(B, K, S, -1).permute(0, 2, 1, 3)
    st = _stage()
    dev = _dev()
    Xd = X.to(dev).contiguous().view(B * S * K, F)
    ndir = 2 if bidir else 1
    Hbuf = torch.full((B * S * K * ndir * H,), float("nan"), device=dev)
    addr = (B * S, K, 1, K, 0, 1) if intra else (B * K, S, K, S * K, 1, K)
    out, nd = wide._recurrence(st, rnn.to(dev), 0, Xd, B * S * K, Hbuf, F, H, addr)
    torch.cuda.synchronize()
    assert nd == ndir and torch.isfinite(out).all()
    assert _rel(out.view(B, S, K, -1), want.contiguous()) <= 3e-6


@pytest.mark.parametrize("M,K,O,rows_per_utt", [(1000, 512, 256, 250), (777, 400, 133, 5), (64, 32, 16, 64), (5000, 262, 300, 1250)])
def test_f16_pieces_linear_against_torch(M, K, O, rows_per_utt):
    """dprnn_wide_linear_pack_f16 + dprnn_wide_linear_f16: the Linear after an LSTM recurrence (rows in (-1, 1)) in two f16 pieces per operand,
    three products, with the fp64 norm sums of its outputs."""
    from tss_with_dprnn_amd import wide
    g = torch.Generator().manual_seed(M + K)
    A = torch.tanh(torch.randn(M, K, generator=g) * 2.0).to(_dev())
    lin = torch.nn.Linear(K, O)
    with torch.no_grad():
        lin.weight.mul_(3.0)
    lin = lin.to(_dev())
    st = _stage()
    Wp, scal, b = wide._pack_linear_f16(st, lin, K, O, _dev())
    B = (M + rows_per_utt - 1) // rows_per_utt
    stats = torch.zeros(B, 2, dtype=torch.float64, device=_dev())
    out = torch.full((M, O), float("nan"), device=_dev())
    st.call("dprnn_wide_linear_f16", _p(A), K, _p(Wp), _p(scal), _p(b), _p(out), O, _p(stats), rows_per_utt, M, K, O)
    want = (A.double() @ lin.weight.double().t() + lin.bias.double()).float()
    assert torch.isfinite(out).all() and _rel(out, want) <= 1e-6
    y = out.double()
    s = torch.zeros(B, 2, dtype=torch.float64, device=_dev())
    idx = (torch.arange(M, device=_dev()) // rows_per_utt)
    s[:, 0].index_add_(0, idx, y.sum(1)); s[:, 1].index_add_(0, idx, (y * y).sum(1))
    assert torch.allclose(stats, s, rtol=1e-5, atol=2e-3)


def test_recurrence_with_32_sequence_tiles(monkeypatch):
    """The 32-sequence workgroups (DPRNN_WIDE_NT=2: half the weight stream, one workgroup per CU; the default is 16), and a count that
    is no multiple of 32."""
    from tss_with_dprnn_amd import wide
    monkeypatch.setenv("DPRNN_WIDE_NT", "2")
    st = _stage()
    n = 32 * 5 + 37
    F = H = 144
    T = 5
    rnn = _torch_rnn("LSTM", F, H, True, 6)
    X = torch.randn(n, T, F, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        want = rnn(X)[0]
    dev = _dev()
    Wp, bv, ndir = wide._pack_rnn(st, rnn.to(dev), 0, F, H, dev)
    out = torch.empty(n * T, 2 * H, device=dev)
    X_d = X.to(dev)
    st.call("dprnn_wide_rnn_fwd", 0, _p(X_d), _p(Wp), _p(bv), _p(out), 2, F, H, n, T, 1, T, 0, 1)
    assert _rel(out.view(n, T, -1), want) <= 3e-6


@pytest.mark.parametrize("L,K,P,C", [(150, 16, 8, 256), (100, 10, 3, 133), (7, 16, 8, 20), (299, 250, 125, 144)])
def test_segmentation_and_overlap_add_against_unfold_fold(L, K, P, C):
    """dprnn.py:189-217: torch's unfold / fold with padding K."""
    from torch.nn.functional import fold, unfold
    B, nspk = 2, 2
    dev = _dev()
    st = _stage()
    g = torch.Generator().manual_seed(L)
    Y = torch.randn(B, L, C, generator=g)
    S = (L + K) // P + 1
    X = torch.empty(B, S, K, C, device=dev)
    Y_d = Y.to(dev)
    st.call("dprnn_wide_segment_fwd", _p(Y_d), _p(X), B, L, K, P, S, C)
    ref = unfold(Y.permute(0, 2, 1).unsqueeze(-1), kernel_size=(K, 1), padding=(K, 0), stride=(P, 1)).reshape(B, C, K, -1)
    assert ref.shape[-1] == S
    assert torch.equal(X.cpu(), ref.permute(0, 3, 2, 1).contiguous())
    U = torch.randn(B, S, K, nspk * C, generator=g)
    fr = torch.empty(B * nspk, L, C, device=dev)
    U_d = U.to(dev)
    st.call("dprnn_wide_ola_fwd", _p(U_d), _p(fr), B, L, K, P, S, C, nspk)
    u = U.permute(0, 3, 2, 1).reshape(B * nspk, C, K, S)                       # [B, 2C, K, S] -> [2B, C, K, S] (dprnn.py:176)
    want = fold(u.reshape(B * nspk, C * K, S), (L, 1), kernel_size=(K, 1), padding=(K, 0), stride=(P, 1)).reshape(B * nspk, C, L)
    assert _rel(fr, want.permute(0, 2, 1).contiguous()) <= 1e-6


def test_encoder_stats_and_norm_prep():
    B, T, N, ks, stride = 3, 500, 70, 4, 2
    dev = _dev()
    st = _stage()
    g = torch.Generator().manual_seed(1)
    wav = torch.randn(B, T, generator=g)
    w = torch.randn(N, 1, ks, generator=g)
    L = (T - ks) // stride + 1
    enc = torch.empty(B * L, N, device=dev)
    wav_d, w_d = wav.to(dev), w.to(dev)
    st.call("dprnn_wide_encoder_fwd", _p(wav_d), _p(w_d), _p(enc), B, T, L, N, ks, stride)
    want = torch.relu(torch.nn.functional.conv1d(wav.unsqueeze(1), w, stride=stride))          # [B, N, L]
    assert _rel(enc.view(B, L, N), want.permute(0, 2, 1).contiguous()) <= 1e-6
    stats = torch.zeros(B, 2, dtype=torch.float64, device=dev)
    st.call("dprnn_wide_flat_stats", _p(enc), _p(stats), B, L * N)
    gn = torch.nn.GroupNorm(1, N)
    with torch.no_grad():
        gn.weight.uniform_(0.5, 2.0)
        gn.bias.normal_()
    sc, sh = st.norm_affine(stats, gn.to(dev), B, L, N)
    got = enc.view(B, L, N) * sc[:, None] + sh[:, None]
    assert _rel(got, gn.cpu()(want).permute(0, 2, 1).contiguous()) <= 2e-6


# ------------------------------------------------------------------------------------------------ models
@pytest.mark.parametrize("name", WIDER)
def test_wider_models_match_the_reference(name):
    g = load_golden(name)
    m = g["meta"]
    model, _, _ = build_model("bss", m["cfg"], m["seed"], _dev())
    with torch.no_grad():
        out = model(torch.from_numpy(g["mix"]).to(_dev()))
    assert out.shape == g["out"].shape
    r = rel_l2(out.cpu().numpy(), g["out"])
    assert r <= REL_TOL, f"{name}: rel-L2 {r:.3e}"
    print(f"{name}: rel-L2 vs reference {r:.2e}")


def test_stage_taps_of_the_256_wide_model():
    g = load_golden("bss_wider256_small")
    m = g["meta"]
    model, _, cfg = build_model("bss", m["cfg"], m["seed"], _dev())
    dbg = {}
    with torch.no_grad():
        model(torch.from_numpy(g["mix"]).to(_dev()), debug=dbg)
    assert rel_l2(dbg["bottleneck"].permute(0, 2, 1).cpu().numpy(), g["tap_bottleneck"]) <= REL_TOL
    for i in range(cfg["n_repeats"]):
        assert rel_l2(dbg[f"block{i}"].permute(0, 3, 2, 1).cpu().numpy(), g[f"tap_block{i}"]) <= REL_TOL, i
    masks = g["tap_masks"] * g["tap_encoder"][:, None]                          # [B, 2, N, L]
    assert rel_l2(dbg["masked"].permute(0, 1, 3, 2).cpu().numpy(), masks) <= REL_TOL


def test_256_wide_model_at_the_shipped_geometry_against_the_cpu_oracle():
    """F = H = 256 at chunk 250 / hop 125 (the shipped chunk geometry), 1.5-s utterances, against the CPU path."""
    model, sd, cfg = build_model("bss", dict(feature_size=256, hidden_size=256, n_repeats=2), 31, _dev())
    B, T = 2, 12000
    x = make_mixture(B, T, seed=77)
    want = otc.tasnet_forward(otc.to_torch_state(sd), cfg, torch.from_numpy(x)).numpy()
    with torch.no_grad():
        got = model(torch.from_numpy(x).to(_dev())).cpu().numpy()
    assert rel_l2(got, want) <= REL_TOL
    src, _ = make_sources(B, T, seed=78)
    assert np.abs(si_sdr(got, src) - si_sdr(want, src)).max() <= 0.01


def test_batch_invariance_determinism_and_ragged_entry_points():
    from tss_with_dprnn_amd import engine, inference
    cfg = dict(feature_size=160, hidden_size=144, chunk_length=16, hop_length=8, n_repeats=1)
    model, sd, cfg = build_model("bss", cfg, 5, _dev())
    x = torch.from_numpy(make_mixture(5, 333, seed=9)).to(_dev())
    with torch.no_grad():
        full = model(x)
        again = model(x)
        one = model(x[3:4])
    assert torch.equal(full, again)
    assert one.shape == (1, 2, 333)
    assert rel_l2(one[0].cpu().numpy(), full[3].cpu().numpy()) <= 2e-6
    lens = [333, 200, 333, 150, 200]
    wavs = [x[i, :n].contiguous() for i, n in enumerate(lens)]
    outs = inference.separate_many(model, wavs, ragged=False)          # equal lengths batched, one batch per stream
    rag = engine.tasnet_forward_ragged(model, wavs)                    # one ragged batch (below)
    for i, n in enumerate(lens):
        with torch.no_grad():
            ref = model(wavs[i][None])[0]
        assert outs[i].shape == (2, n) and rel_l2(outs[i].cpu().numpy(), ref.cpu().numpy()) <= 2e-6, i
        assert rag[i].shape == (2, n) and rel_l2(rag[i].cpu().numpy(), ref.cpu().numpy()) <= 2e-6, i
    with torch.no_grad():
        out, ticket = engine.forward_async(model, x)
    assert torch.equal(ticket.result(), full)


def test_nan_input_comes_back_as_nan_like_the_reference():
    model, _, _ = build_model("bss", dict(feature_size=144, hidden_size=144, chunk_length=16, hop_length=8, n_repeats=1), 5, _dev())
    x = torch.from_numpy(make_mixture(2, 200, seed=1)).to(_dev())
    x[1, 50] = float("nan")
    with torch.no_grad():
        out = model(x)
    assert torch.isfinite(out[0]).all() and torch.isnan(out[1]).any()


def test_limits_raise_and_name_themselves():
    import tss_with_dprnn_amd as pkg
    from tss_with_dprnn_amd import dprnn_spe
    dev = _dev()
    x = torch.zeros(1, 400, device=dev)
    big = pkg.DPRNNTasNet(64, feature_size=128, hidden_size=528, chunk_length=16, n_repeats=1).to(dev)
    with torch.no_grad(), pytest.raises(NotImplementedError, match="hidden_size 528"):
        big(x)
    lds = pkg.DPRNNTasNet(64, feature_size=1024, hidden_size=512, chunk_length=16, n_repeats=1).to(dev)
    with torch.no_grad(), pytest.raises(NotImplementedError, match="160 KB"):
        lds(x)
    wide = pkg.DPRNNTasNet(64, feature_size=256, hidden_size=256, chunk_length=16, n_repeats=1).to(dev)
    with pytest.raises(NotImplementedError):                 # gradients at these widths: not built
        wide(x)
    spe = dprnn_spe.DPRNNSpeTasNet(64, feature_size=256, hidden_size=256, chunk_length=16, n_repeats=1).to(dev)
    with pytest.raises(NotImplementedError):                 # training of the speaker-conditioned classes at these widths: not built
        spe(x, x, torch.tensor(400.0))


@pytest.mark.parametrize("name", golden_names("spe_wider") + golden_names("ira_wider"))
def test_wider_speaker_conditioned_models_match_the_reference(name):
    """DPRNNSpeTasNet / DPRNNSpeIRATasNet with feature / hidden size above the build (input_size and speaker encoder at the built widths):
    speaker branch and fusion operands on the built kernels, DPRNN stack on the any-width ones; estimate and speaker logits against the
    reference classes' outputs, BatchNorm in eval or (as the reference's Inferencer leaves it) train mode."""
    from conftest import state_from_meta
    g = load_golden(name)
    m = g["meta"]
    model, sd, _ = build_model(m["kind"], m["cfg"], m["seed"], _dev())
    if m.get("bn_training"):
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.train()
    with torch.no_grad():
        out, logits = model(torch.from_numpy(g["mix"]).to(_dev()), torch.from_numpy(g["aux"]).to(_dev()), torch.tensor(float(g["aux_len"])))
    assert out.shape == g["out"].shape
    r, rl = rel_l2(out.cpu().numpy(), g["out"]), rel_l2(logits.cpu().numpy(), g["logits"])
    assert r <= REL_TOL and rl <= REL_TOL, f"{name}: estimate rel-L2 {r:.3e}, logits {rl:.3e}"
    print(f"{name}: rel-L2 vs reference {r:.2e} (logits {rl:.2e})")


def test_wider_speaker_conditioned_entry_points():
    """separate_many (equal-length buckets), forward_async, batch invariance for a wide DPRNN-Spe-IRA; what stays out raises by name."""
    from tss_with_dprnn_amd import dprnn_spe, engine, inference
    cfg = dict(feature_size=160, hidden_size=144, chunk_length=16, hop_length=8, n_repeats=1, fusion_type="film", norm_type="ln")
    model, sd, cfg = build_model("ira", cfg, 7, _dev())
    x = torch.from_numpy(make_mixture(3, 400, seed=9)).to(_dev())
    a = torch.from_numpy(make_mixture(3, 350, seed=10)).to(_dev())
    with torch.no_grad():
        full, lg = model(x, a, torch.tensor(350.0))
        one, lg1 = model(x[1:2], a[1:2], torch.tensor(350.0))
        (out2, lg2), ticket = engine.forward_async(model, x, a, torch.tensor(350.0))
    assert full.shape == (3, 400) and rel_l2(one[0].cpu().numpy(), full[1].cpu().numpy()) <= 2e-6
    assert torch.equal(ticket.result()[0], full)
    outs = inference.separate_many(model, [x[i] for i in range(3)], extra_inputs=[(a[i], 350.0) for i in range(3)])
    for i in range(3):
        assert rel_l2(outs[i][0].cpu().numpy(), full[i].cpu().numpy()) <= 2e-6
    narrow_in = dprnn_spe.DPRNNSpeTasNet(48, feature_size=256, hidden_size=256, chunk_length=16, n_repeats=1).to(_dev())
    with torch.no_grad(), pytest.raises(NotImplementedError, match="input_size 64"):
        narrow_in(x, a, torch.tensor(350.0))
    wide_spk = dprnn_spe.DPRNNSpeTasNet(64, feature_size=256, hidden_size=256, chunk_length=16, n_repeats=1, O=256).to(_dev())
    with torch.no_grad(), pytest.raises(NotImplementedError, match="speaker encoder"):
        wide_spk(x, a, torch.tensor(350.0))


@pytest.mark.parametrize("over", [dict(feature_size=256, hidden_size=256, n_repeats=2), dict(input_size=96, feature_size=160, hidden_size=200, rnn_type="GRU",
                                                                                               bidirectional=False, norm_type="gLN", n_repeats=1),
                                  dict(feature_size=144, hidden_size=136, chunk_length=20, hop_length=7, rnn_type="RNN", n_repeats=1)])
def test_ragged_batches_of_a_wider_model(over, monkeypatch):
    """Utterances of DIFFERENT lengths through one recurrence launch per path (wide._ragged_batch: concatenated slabs, tile descriptors with
    per-utterance step counts on the inter-chunk path): every output equals that utterance's batch-of-one forward (the reference's
    semantics, inferencer.py:54-58) and the CPU oracle; separate_many takes this route by itself; small slab budgets split the batch."""
    from tss_with_dprnn_amd import inference, wide
    cfg = dict(dict(chunk_length=16, hop_length=8), **over)
    model, sd, cfg = build_model("bss", cfg, 41, _dev())
    lens = [900, 333, 1501, 64, 700, 333, 1200, 250, 999]
    wavs = [torch.from_numpy(make_mixture(1, n, seed=60 + i)[0]).to(_dev()) for i, n in enumerate(lens)]
    assert len(wavs) >= inference.RAGGED_MIN_UTTS
    outs = inference.separate_many(model, wavs)
    tsd = otc.to_torch_state(sd)
    for i, n in enumerate(lens):
        with torch.no_grad():
            one = model(wavs[i][None])[0]
        assert outs[i].shape == (2, n) and rel_l2(outs[i].cpu().numpy(), one.cpu().numpy()) <= 2e-6, i
        if i in (0, 3):
            want = otc.tasnet_forward(tsd, cfg, wavs[i][None].cpu())[0].numpy()
            assert rel_l2(outs[i].cpu().numpy(), want) <= REL_TOL, i
    monkeypatch.setattr(wide, "RAGGED_MAX_SLAB_BYTES", 40 << 20)       # a few utterances per ragged batch
    again = wide.tasnet_forward_ragged(model, wavs)
    for a, b in zip(again, outs):
        assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) <= 2e-6


@pytest.mark.parametrize("kind,fusion,bn_train", [("spe", "att", True), ("ira", "cat", False), ("ira", "film", True)])
def test_ragged_batches_of_wider_speaker_conditioned_models(kind, fusion, bn_train):
    """Mixtures AND reference utterances of different lengths (inferencer_spe.py:20-54): speaker branch, fusion and bottleneck per utterance,
    the DPRNN stack of every pass as one ragged slab; estimate and logits equal each utterance's batch-of-one forward."""
    from tss_with_dprnn_amd import inference
    cfg = dict(feature_size=160, hidden_size=144, chunk_length=16, hop_length=8, n_repeats=1, fusion_type=fusion, norm_type="ln")
    model, sd, cfg = build_model(kind, cfg, 43, _dev())
    if bn_train:
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.train()
    lens = [900, 333, 1501, 420, 700, 333, 1200, 250]
    alens = [400, 377, 350, 500, 377, 420, 333, 390]
    wavs = [torch.from_numpy(make_mixture(1, n, seed=70 + i)[0]).to(_dev()) for i, n in enumerate(lens)]
    auxs = [torch.from_numpy(make_mixture(1, n, seed=90 + i)[0]).to(_dev()) for i, n in enumerate(alens)]
    outs = inference.separate_many(model, wavs, extra_inputs=[(a, float(n)) for a, n in zip(auxs, alens)])
    for i, n in enumerate(lens):
        with torch.no_grad():
            est, logits = model(wavs[i][None], auxs[i][None], torch.tensor(float(alens[i])))
        assert outs[i][0].shape == (n,) and rel_l2(outs[i][0].cpu().numpy(), est[0].cpu().numpy()) <= 2e-6, i
        assert rel_l2(outs[i][1].cpu().numpy(), logits[0].cpu().numpy()) <= 2e-6, i


# ------------------------------------------------------------------------------------------------ separation-level entry points
@pytest.mark.parametrize("over", [dict(feature_size=256, hidden_size=256), dict(input_size=96, feature_size=160, hidden_size=200, norm_type="gLN",
                                                                                activation_type="relu"),
                                  dict(feature_size=64, hidden_size=64), dict(input_size=40, feature_size=96, hidden_size=80, rnn_type="GRU")])
def test_separator_level_forwards_at_any_width(over):
    """DPRNN.forward (dprnn.py:166-187) and DPRNN._dprnn_process (:171-186) on materialised [B, C, L] tensors, as the reference's
    DPRNNRawNet drives them (dprnn_rawnet.py:79-105): every width that is not exactly the fused kernels' build -- wider or narrower --
    runs on the any-width kernels."""
    from oracle import dprnn_numpy as onp
    model, sd, cfg = build_model("bss", dict(chunk_length=16, hop_length=8, n_repeats=2, **over), 73, _dev())
    enc = onp.encoder(sd, cfg, make_mixture(3, 333, seed=8))                                   # [B, N, L]
    want = onp.separator(sd, cfg, enc)
    with torch.no_grad():
        got = model.separation(torch.from_numpy(enc).to(_dev()))
    assert got.shape == want.shape == (3, 2, cfg["input_size"], 332)
    assert rel_l2(got.cpu().numpy(), want) <= REL_TOL
    x = onp.channel_norm(sd, "separation.bottleneck.0", enc, cfg["norm_type"])
    x = onp.conv1x1(sd["separation.bottleneck.1.weight"], sd["separation.bottleneck.1.bias"], x)   # [B, F, L]
    with torch.no_grad():
        got2 = model.separation._dprnn_process(torch.from_numpy(np.ascontiguousarray(x)).to(_dev()), 3, 332)
    assert rel_l2(got2.cpu().numpy(), want) <= REL_TOL
    with pytest.raises(NotImplementedError, match="inference-only"):
        model.separation(torch.from_numpy(enc).to(_dev()))


@pytest.mark.parametrize("over,B,T", [(dict(kernel_size=16, stride=8, chunk_length=12, hop_length=6), 2, 1000),       # Conv-TasNet-style long filters
                                      (dict(chunk_length=10, hop_length=10), 3, 157),                               # no overlap between chunks
                                      (dict(chunk_length=7, hop_length=2), 2, 90),                                  # hop that does not divide the chunk
                                      (dict(chunk_length=250, hop_length=125), 1, 3),                               # two encoder frames, one real chunk row
                                      (dict(kernel_size=3, stride=1, chunk_length=8, hop_length=4, bidirectional=False), 2, 64)])
def test_edge_geometries_against_the_cpu_oracle(over, B, T):
    """Encoder filter lengths beyond the built path's 8 taps, hops equal to / not dividing the chunk, utterances of a few frames."""
    model, sd, cfg = build_model("bss", dict(feature_size=144, hidden_size=136, n_repeats=1, **over), 17, _dev())
    x = make_mixture(B, T, seed=T)
    want = otc.tasnet_forward(otc.to_torch_state(sd), cfg, torch.from_numpy(x)).numpy()
    with torch.no_grad():
        got = model(torch.from_numpy(x).to(_dev())).cpu().numpy()
    assert got.shape == want.shape
    assert rel_l2(got, want) <= REL_TOL, rel_l2(got, want)


def test_long_encoder_filters_at_the_built_widths_take_the_any_width_path():
    """kernel_size 16 / stride 8 (Conv-TasNet's encoder) at 64 / 128 / 128: beyond the fused kernels' 8 taps, so the any-width kernels run it."""
    model, sd, cfg = build_model("bss", dict(kernel_size=16, stride=8, chunk_length=20, hop_length=10, n_repeats=1), 19, _dev())
    x = make_mixture(2, 4000, seed=5)
    want = otc.tasnet_forward(otc.to_torch_state(sd), cfg, torch.from_numpy(x)).numpy()
    with torch.no_grad():
        got = model(torch.from_numpy(x).to(_dev())).cpu().numpy()
    assert got.shape == want.shape and rel_l2(got, want) <= REL_TOL


@pytest.mark.parametrize("rnn_type,bidir,N,F,H", [("LSTM", True, 64, 256, 256), ("GRU", False, 96, 96, 80), ("RNN", True, 40, 133, 70)])
def test_sub_module_forwards_at_any_width(rnn_type, bidir, N, F, H):
    """SingleRNN.forward, DPRNNBlock.forward (dprnn.py:33-37, 79-99) and Encoder.forward (encoder_decoder.py:25-33) of models off the fused
    kernels' build, against the reference's permutations written with stock torch modules on the CPU."""
    model, _, cfg = build_model("bss", dict(input_size=N, feature_size=F, hidden_size=H, chunk_length=16, hop_length=8, n_repeats=1, rnn_type=rnn_type,
                                            bidirectional=bidir, kernel_size=16 if N == 96 else 2), 77, _dev())
    blk = model.separation.dprnn_blocks[0]

    def cpu_rnn(single):
        r = getattr(torch.nn, rnn_type)(F, H, num_layers=1, batch_first=True, bidirectional=single.bidirectional)
        r.load_state_dict({k: v.cpu() for k, v in single.rnn.state_dict().items()})
        return r
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal((37, 21, F)).astype(np.float32))
    with torch.no_grad():
        got = blk.inter_rnn(x.to(_dev())).cpu().numpy()
        want = cpu_rnn(blk.inter_rnn)(x)[0].numpy()
    assert got.shape == want.shape and rel_l2(got, want) < 3e-6
    B, K, S = 2, 16, 9
    xb = torch.from_numpy(rng.standard_normal((B, F, K, S)).astype(np.float32))
    nd = 2 if bidir else 1
    with torch.no_grad():
        got = blk(xb.to(_dev())).cpu().numpy()
        intra, inter = cpu_rnn(blk.intra_rnn), cpu_rnn(blk.inter_rnn)
        lin1 = torch.nn.Linear(2 * H, F); lin1.load_state_dict({k: v.cpu() for k, v in blk.intra_linear.state_dict().items()})
        lin2 = torch.nn.Linear(nd * H, F); lin2.load_state_dict({k: v.cpu() for k, v in blk.inter_linear.state_dict().items()})
        n1 = torch.nn.GroupNorm(1, F); n1.load_state_dict({k: v.cpu() for k, v in blk.intra_norm.state_dict().items()})
        n2 = torch.nn.GroupNorm(1, F); n2.load_state_dict({k: v.cpu() for k, v in blk.inter_norm.state_dict().items()})
        o = xb.transpose(1, -1).reshape(B * S, K, F)
        o = lin1(intra(o)[0]).reshape(B, S, K, F).transpose(1, -1)
        o = n1(o) + xb
        y = o
        o = o.transpose(1, 2).transpose(2, -1).reshape(B * K, S, F)
        o = lin2(inter(o)[0]).reshape(B, K, S, F).transpose(1, -1).transpose(2, -1).contiguous()
        want = (n2(o) + y).numpy()
    assert got.shape == want.shape and rel_l2(got, want) < 5e-6, rel_l2(got, want)
    if N != 64 or cfg["kernel_size"] > 8:
        wav = torch.from_numpy(rng.standard_normal((3, 501)).astype(np.float32))
        with torch.no_grad():
            e = model.encoder(wav.to(_dev()))
            want = torch.relu(torch.nn.functional.conv1d(wav[:, None, :], model.encoder.conv1d.weight.detach().cpu(), stride=model.encoder.conv1d.stride[0]))
        assert e.shape == want.shape and rel_l2(e.cpu().numpy(), want.numpy()) < 1e-6
