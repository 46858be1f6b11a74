"""GPU parity of the DPRNN-Spe / DPRNN-Spe-IRA path (speaker encoder, fusion, two-pass IRA) against the
reference's golden vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import build_model, load_golden, rel_l2
from tss_with_dprnn_amd._lib import DprnnError
from oracle import dprnn_numpy as onp
from oracle.weights import make_mixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REL_TOL = 1e-5


def _run(model, g, bn_train=False):
    if bn_train:        # the reference Inferencer leaves BatchNorm in train mode
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.train()
    with torch.no_grad():
        est, logits = model(torch.from_numpy(g["mix"]).to(DEV), torch.from_numpy(g["aux"]).to(DEV),
                            torch.tensor(float(g["aux_len"])))
    torch.cuda.synchronize()
    return est.cpu().numpy(), logits.cpu().numpy()


@pytest.mark.parametrize("name", ["spe_wide_small_cat", "spe_wide_small_cat_bntrain", "spe_full_b1", "ira_wide_small_cat",
                                  "spe_wide_small_att", "spe_full_att_b1", "spe_wide_small_film",
                                  # BASELINE.json configs[4]: IRA at the shipped depth / chunk geometry; 3 s and 36 720 samples
                                  "ira_full_b1", "ira_full_long_b1"])
def test_spe_matches_reference_golden(name):
    g = load_golden(name)
    m = g["meta"]
    model, _, _ = build_model(m["kind"], m["cfg"], m["seed"], DEV)
    bn0 = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
    est, logits = _run(model, g, m["bn_training"])
    assert est.shape == g["out"].shape and logits.shape == g["logits"].shape
    assert np.isfinite(est).all()
    assert rel_l2(logits, g["logits"]) <= REL_TOL, rel_l2(logits, g["logits"])
    assert rel_l2(est, g["out"]) <= REL_TOL, rel_l2(est, g["out"])
    sd = model.state_dict()
    for k, v in bn0.items():
        if m["bn_training"]:     # torch semantics: running buffers move, the counter increments (twice per IRA pass)
            assert not torch.equal(sd[k], v), k
        else:
            assert torch.equal(sd[k], v), k


def test_bn_train_running_stats_follow_torch():
    """One ResBlock-sized check of the in-place running-statistics update against nn.BatchNorm1d itself."""
    import tss_with_dprnn_amd as pkg
    from tss_with_dprnn_amd import engine
    st = engine.Stage(torch.device(DEV))
    C, n = 128, 777
    y = torch.randn(n, C, device=DEV) * 1.7 + 0.3
    bn = torch.nn.BatchNorm1d(C).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
        bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.5, 1.5)
    ref = torch.nn.BatchNorm1d(C).to(DEV)
    ref.load_state_dict(bn.state_dict())
    ref.train()
    want = ref(y.t().unsqueeze(0)).squeeze(0).t()
    cs = torch.stack([y.double().sum(0), (y.double() ** 2).sum(0)], dim=1).contiguous()
    scale = torch.empty(C, device=DEV); shift = torch.empty(C, device=DEV); mean = torch.empty(C, device=DEV)
    st.call("dprnn_bn_prep", bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
            bn.num_batches_tracked.data_ptr(), cs.data_ptr(), n, 0.1, 1e-5, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), C)
    torch.cuda.synchronize()
    got = (y - mean) * scale + shift
    assert rel_l2(got.detach().cpu().numpy(), want.detach().cpu().numpy()) < 1e-6
    assert torch.allclose(bn.running_mean, ref.running_mean, rtol=1e-6, atol=1e-7)
    assert torch.allclose(bn.running_var, ref.running_var, rtol=1e-6, atol=1e-7)
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("fusion,nt", [("add", "ln"), ("mul", "gLN"), ("film", "ln"), ("cat", "gLN"), ("att", "ln"), ("att", "gLN")])
def test_fusions_match_cpu_oracle(fusion, nt):
    cfg = dict(chunk_length=16, hop_length=8, n_repeats=1, fusion_type=fusion, norm_type=nt)
    model, sd, cfg = build_model("spe", cfg, 70, DEV)
    B, T = 3, 400
    mix, aux = make_mixture(B, T, seed=1), make_mixture(B, 350, seed=2)
    want, wlog = onp.spe_forward(sd, cfg, mix, aux, np.float32(350.0))
    with torch.no_grad():
        est, logits = model(torch.from_numpy(mix).to(DEV), torch.from_numpy(aux).to(DEV), torch.tensor(350.0))
    assert rel_l2(logits.cpu().numpy(), wlog) <= REL_TOL
    assert rel_l2(est.cpu().numpy(), want) <= REL_TOL


def test_ira_fusion_mul_matches_cpu_oracle():
    cfg = dict(chunk_length=16, hop_length=8, n_repeats=1, fusion_type="mul", norm_type="ln")
    model, sd, cfg = build_model("ira", cfg, 71, DEV)
    B, T = 2, 400
    mix, aux = make_mixture(B, T, seed=3), make_mixture(B, T, seed=4)
    want, wlog = onp.ira_forward(sd, cfg, mix, aux, np.float32(T))
    with torch.no_grad():
        est, logits = model(torch.from_numpy(mix).to(DEV), torch.from_numpy(aux).to(DEV), torch.tensor(float(T)))
    assert rel_l2(logits.cpu().numpy(), wlog) <= REL_TOL
    assert rel_l2(est.cpu().numpy(), want) <= REL_TOL


def test_ira_att_matches_cpu_oracle():
    cfg = dict(chunk_length=16, hop_length=8, n_repeats=1, fusion_type="att", norm_type="ln")
    model, sd, cfg = build_model("ira", cfg, 72, DEV)
    B, T = 2, 401
    mix, aux = make_mixture(B, T, seed=5), make_mixture(B, 377, seed=6)
    want, wlog = onp.ira_forward(sd, cfg, mix, aux, np.float32(377))
    with torch.no_grad():
        est, logits = model(torch.from_numpy(mix).to(DEV), torch.from_numpy(aux).to(DEV), torch.tensor(377.0))
    assert rel_l2(logits.cpu().numpy(), wlog) <= REL_TOL
    assert rel_l2(est.cpu().numpy(), want) <= REL_TOL


# ------------------------------------------------------------------ separation-level entry points
def _enc(sd, cfg, wav):
    return onp.encoder(sd, cfg, wav)                  # [B, N, L] as the reference's Encoder returns it


@pytest.mark.parametrize("nt,act", [("ln", "sigmoid"), ("gLN", "relu")])
def test_separator_forward_on_materialised_encoder_output(nt, act):
    """DPRNN.forward (dprnn.py:166-187) called on an encoder output tensor, as code holding [B, N, L] tensors does."""
    model, sd, cfg = build_model("bss", dict(chunk_length=16, hop_length=8, n_repeats=2, norm_type=nt, activation_type=act), 73, DEV)
    enc = _enc(sd, cfg, make_mixture(3, 333, seed=8))
    want = onp.separator(sd, cfg, enc)
    with torch.no_grad():
        got = model.separation(torch.from_numpy(enc).to(DEV))
    assert got.shape == want.shape == (3, 2, 64, 332)
    assert rel_l2(got.cpu().numpy(), want) <= REL_TOL
    with pytest.raises(NotImplementedError, match="inference-only"):
        model.separation(torch.from_numpy(enc).to(DEV))          # grad mode: the backward lives in the TasNet wrappers


@pytest.mark.parametrize("fusion,nt", [("cat", "ln"), ("att", "gLN"), ("film", "ln")])
def test_dprnn_spe_forward_on_materialised_encoder_outputs(fusion, nt):
    """DPRNNSpe.forward (dprnn_spe.py:125-154) = _auxiliary + bottleneck[0] + _fusion + bottleneck[1] + _dprnn_process."""
    cfg = dict(chunk_length=16, hop_length=8, n_repeats=1, fusion_type=fusion, norm_type=nt)
    model, sd, cfg = build_model("spe", cfg, 74, DEV)
    B, T, Ta = 2, 402, 377
    enc, aenc = _enc(sd, cfg, make_mixture(B, T, seed=11)), _enc(sd, cfg, make_mixture(B, Ta, seed=12))
    emb = onp.auxiliary(sd, cfg, aenc, np.float32(Ta), False)
    x = onp.channel_norm(sd, "separation.bottleneck.0", enc, nt)
    x = onp.conv1x1(sd["separation.bottleneck.1.weight"], sd["separation.bottleneck.1.bias"], onp.fusion(sd, cfg, emb, x))
    want = onp.dprnn_process(sd, cfg, x, enc.shape[-1])
    wlog = emb @ sd["separation.pred_linear.weight"].T + sd["separation.pred_linear.bias"]
    with torch.no_grad():
        masks, logits = model.separation(torch.from_numpy(enc).to(DEV), torch.from_numpy(aenc).to(DEV), torch.tensor(float(Ta)))
    assert rel_l2(logits.cpu().numpy(), wlog) <= REL_TOL
    assert rel_l2(masks.cpu().numpy(), want) <= REL_TOL


def test_rawnet_style_subclass_uses_the_device_pieces():
    """src/models/dprnn_rawnet.py:9-105 subclasses DPRNNSpe, swaps the speaker encoder and drives bottleneck[i], _fusion,
    _dprnn_process and pred_linear itself; the same call sequence on our DPRNNSpe (with a stand-in speaker encoder)."""
    from tss_with_dprnn_amd.dprnn_spe import DPRNNSpe

    class Sub(DPRNNSpe):
        def forward(self, input, aux):                      # dprnn_rawnet.py:73-102
            B, _, L = input.size()
            aux = self._auxiliary(aux)
            output = self.bottleneck[0](input)
            output = self._fusion(aux, output, L)
            output = self.bottleneck[1](output)
            output = self._dprnn_process(output, B, L)
            return output, self.pred_linear(aux)

        def _auxiliary(self, aux):                          # stand-in for RawNet3: any [B, E] embedding
            return aux.mean(dim=-1).repeat(1, 2) * 3.0

    cfg = dict(chunk_length=16, hop_length=8, n_repeats=1, fusion_type="mul", norm_type="ln")
    _, sd, cfg = build_model("spe", cfg, 75, None)
    sub = Sub(64, 128, 128, 16, 8, 1, True, "LSTM", "ln", "sigmoid", 0, 128, 256, 128, cfg["num_spks"], 2, "mul")
    sub.load_state_dict({k[len("separation."):]: torch.from_numpy(np.asarray(v)) for k, v in sd.items()
                         if k.startswith("separation.")}, strict=True)
    sub = sub.eval().to(DEV)
    enc, aenc = _enc(sd, cfg, make_mixture(2, 300, seed=13)), _enc(sd, cfg, make_mixture(2, 280, seed=14))
    emb = (np.tile(aenc.mean(-1), (1, 2)) * 3.0).astype(np.float32)
    x = onp.channel_norm(sd, "separation.bottleneck.0", enc, "ln")
    x = onp.conv1x1(sd["separation.bottleneck.1.weight"], sd["separation.bottleneck.1.bias"], onp.fusion(sd, cfg, emb, x))
    want = onp.dprnn_process(sd, cfg, x, enc.shape[-1])
    with torch.no_grad():
        masks, logits = sub(torch.from_numpy(enc).to(DEV), torch.from_numpy(aenc).to(DEV))
    assert rel_l2(masks.cpu().numpy(), want) <= REL_TOL
    assert rel_l2(logits.cpu().numpy(), emb @ sd["separation.pred_linear.weight"].T + sd["separation.pred_linear.bias"]) <= REL_TOL


@pytest.mark.parametrize("fusion", ["cat", "add", "mul", "film", "att"])
@pytest.mark.parametrize("norm_type", ["ln", "gLN"])
def test_separator_glue_modules_run_on_the_hip_kernels(fusion, norm_type):
    """Round 4: bottleneck[0] (GroupNorm / gLN), _fusion (all five), bottleneck[1] (1x1 conv, 192 input channels for 'cat'),
    pred_linear / fusion_linear* (per-utterance Linears) -- the pieces a DPRNNRawNet-style subclass calls one by one
    (dprnn_rawnet.py:79-105) -- against the stock torch CPU operators the reference's modules are (float64), on odd frame counts."""
    from conftest import build_model, rel_l2
    import torch.nn.functional as F
    model, _, cfg = build_model("spe", dict(chunk_length=16, hop_length=8, n_repeats=1, fusion_type=fusion, norm_type=norm_type), 78, DEV)
    sep = model.separation
    rng = np.random.default_rng(5)
    B, N, L = 3, 64, 301
    x = torch.from_numpy((rng.standard_normal((B, N, L)) * 2.0 + 0.3).astype(np.float32))
    emb = torch.from_numpy(rng.standard_normal((B, 128)).astype(np.float32))
    cpu = {k: v.detach().cpu().double() for k, v in sep.state_dict().items()}
    with torch.no_grad():
        got_n = sep.bottleneck[0](x.to(DEV))
        got_f = sep._fusion(emb.to(DEV), got_n, L)
        got_c = sep.bottleneck[1](got_f)
        got_p = sep.pred_linear(emb.to(DEV))
    xd = x.double()
    if norm_type == "ln":
        want_n = F.group_norm(xd, 1, cpu["bottleneck.0.weight"], cpu["bottleneck.0.bias"], 1e-5)
    else:
        mean = xd.mean(dim=(1, 2), keepdim=True)
        var = (xd - mean).pow(2).mean(dim=(1, 2), keepdim=True)
        want_n = (xd - mean) / (var + 1e-8).sqrt() * cpu["bottleneck.0.gamma"].view(1, -1, 1) + cpu["bottleneck.0.beta"].view(1, -1, 1)
    assert rel_l2(got_n.cpu().numpy(), want_n.numpy()) < 2e-6
    e = emb.double()
    lin = lambda n: F.linear(e, cpu[n + ".weight"], cpu[n + ".bias"]).unsqueeze(-1)
    if fusion == "cat":
        want_f = torch.cat((want_n, e.unsqueeze(-1).expand(-1, -1, L)), dim=1)
    elif fusion == "add":
        want_f = want_n + lin("fusion_linear")
    elif fusion == "mul":
        want_f = want_n * lin("fusion_linear")
    elif fusion == "film":
        want_f = want_n * lin("fusion_linear_1") + lin("fusion_linear_2")
    else:
        a = lin("fusion_linear")
        avg = F.conv1d(want_n, cpu["average.weight"], cpu["average.bias"], stride=2, groups=N)
        att = torch.softmax((avg * a).sum(dim=1, keepdim=True), dim=-1)
        want_f = want_n * F.interpolate(att * a + a, size=L, mode="nearest")
    assert got_f.shape == want_f.shape and rel_l2(got_f.cpu().numpy(), want_f.numpy()) < 3e-6
    want_c = F.conv1d(want_f, cpu["bottleneck.1.weight"], cpu["bottleneck.1.bias"])
    assert rel_l2(got_c.cpu().numpy(), want_c.numpy()) < 3e-6
    assert rel_l2(got_p.cpu().numpy(), F.linear(e, cpu["pred_linear.weight"], cpu["pred_linear.bias"]).numpy()) < 2e-6
    with pytest.raises(NotImplementedError, match="inference-only"):
        sep.pred_linear(emb.to(DEV))                       # gradients enabled: no silent torch fallback
    with pytest.raises(DprnnError, match="HIP device"):
        with torch.no_grad():
            sep.bottleneck[0](x)                           # CPU tensor: no CPU fallback


# ---- sub-module forwards (inference): the reference's classes have them (dprnn.py:33-37,79-99, encoder_decoder.py:25-33,40-49,
# dprnn_spe.py:31-42, dprnn_spe_ira.py:53-115); the mirrors run them through the same HIP kernels.  Checked against stock torch
# CPU modules carrying the same weights (the operators the reference's classes are made of).
def _cpu_lstm_like(rnn):
    mod = getattr(torch.nn, rnn.rnn.mode)(rnn.rnn.input_size, rnn.rnn.hidden_size, 1, batch_first=True, bidirectional=rnn.rnn.bidirectional)
    with torch.no_grad():
        for n, p in mod.named_parameters():
            p.copy_(getattr(rnn.rnn, n).detach().cpu())
    return mod


@pytest.mark.parametrize("rnn_type,bidir", [("LSTM", True), ("GRU", False), ("RNN", True)])
def test_single_rnn_and_block_forward(rnn_type, bidir):
    from conftest import build_model, rel_l2
    model, _, cfg = build_model("bss", dict(chunk_length=16, hop_length=8, n_repeats=1, rnn_type=rnn_type, bidirectional=bidir), 77, DEV)
    blk = model.separation.dprnn_blocks[0]
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal((37, 21, 128)).astype(np.float32))
    with torch.no_grad():
        got = blk.inter_rnn(x.to(DEV)).cpu().numpy()
        want = _cpu_lstm_like(blk.inter_rnn)(x)[0].numpy()
    assert got.shape == want.shape and rel_l2(got, want) < 2e-6
    # the whole block on [B, F, K, S] against the reference's permutations (dprnn.py:79-99) written with stock modules
    B, K, S = 2, 16, 9
    xb = torch.from_numpy(rng.standard_normal((B, 128, K, S)).astype(np.float32))
    with torch.no_grad():
        got = blk(xb.to(DEV)).cpu().numpy()
        def cpu(m):
            return m.cpu() if hasattr(m, "cpu") else m
        intra, inter = _cpu_lstm_like(blk.intra_rnn), _cpu_lstm_like(blk.inter_rnn)
        lin1 = torch.nn.Linear(256, 128); lin1.load_state_dict({k: v.cpu() for k, v in blk.intra_linear.state_dict().items()})
        lin2 = torch.nn.Linear(128 * (2 if bidir else 1), 128); lin2.load_state_dict({k: v.cpu() for k, v in blk.inter_linear.state_dict().items()})
        n1 = torch.nn.GroupNorm(1, 128); n1.load_state_dict({k: v.cpu() for k, v in blk.intra_norm.state_dict().items()})
        n2 = torch.nn.GroupNorm(1, 128); n2.load_state_dict({k: v.cpu() for k, v in blk.inter_norm.state_dict().items()})
        o = xb.transpose(1, -1).reshape(B * S, K, 128)
        o = lin1(intra(o)[0]).reshape(B, S, K, 128).transpose(1, -1)
        o = n1(o) + xb
        y = o
        o = o.transpose(1, 2).transpose(2, -1).reshape(B * K, S, 128)
        o = lin2(inter(o)[0]).reshape(B, K, S, 128).transpose(1, -1).transpose(2, -1).contiguous()
        want = (n2(o) + y).numpy()
    assert got.shape == want.shape and rel_l2(got, want) < 3e-6, rel_l2(got, want)


def test_encoder_decoder_resblock_forward():
    from conftest import build_model, rel_l2
    model, _, cfg = build_model("spe", dict(chunk_length=16, hop_length=8, n_repeats=1, kernel_size=4), 78, DEV)
    rng = np.random.default_rng(4)
    wav = torch.from_numpy(rng.standard_normal((3, 501)).astype(np.float32))
    enc_w = model.encoder.conv1d.weight.detach().cpu()
    with torch.no_grad():
        e = model.encoder(wav.to(DEV))
        want = torch.relu(torch.nn.functional.conv1d(wav[:, None, :], enc_w, stride=model.encoder.conv1d.stride[0]))
        assert e.shape == want.shape and rel_l2(e.cpu().numpy(), want.numpy()) < 1e-6
        z = torch.from_numpy(rng.standard_normal(tuple(want.shape)).astype(np.float32))
        d = model.decoder(z.to(DEV))
        wantd = torch.nn.functional.conv_transpose1d(z, model.decoder.weight.detach().cpu(), stride=model.decoder.stride[0]).squeeze()
        assert d.shape == wantd.shape and rel_l2(d.cpu().numpy(), wantd.numpy()) < 1e-6
        d1 = model.decoder(z[:1].to(DEV))            # B = 1: the reference squeezes only dim 1 (encoder_decoder.py:44-48)
        assert tuple(d1.shape) == (1, wantd.shape[-1])
        rb = model.separation.spk_encoder[3]         # ResBlock(128 -> 256) in eval mode
        xr = torch.from_numpy(rng.standard_normal((2, 128, 95)).astype(np.float32))
        got = rb(xr.to(DEV)).cpu().numpy()
        import copy
        from tss_with_dprnn_amd import dprnn_spe
        ref = copy.deepcopy(rb).cpu()
        y = ref.conv1(xr); y = ref.prelu1(ref.batch_norm1(y)); y = ref.batch_norm2(ref.conv2(y))
        y = ref.maxpool(ref.prelu2(y + ref.conv_downsample(xr)))
        assert got.shape == tuple(y.shape) and rel_l2(got, y.numpy()) < 2e-6


def test_ira_separator_forward_on_encoder_outputs():
    """DPRNNSpeIRA.forward(input [B,N,L], aux [B,N,La], aux_len) (dprnn_spe_ira.py:53-115) against the TasNet wrapper's own
    pieces: d1 decoded must equal DPRNNSpeIRATasNet.forward's estimate, the logits must agree."""
    from conftest import build_model, rel_l2
    from oracle.weights import make_mixture
    model, _, cfg = build_model("ira", dict(chunk_length=16, hop_length=8, n_repeats=1, fusion_type="cat", norm_type="ln"), 79, DEV)
    mix = torch.from_numpy(make_mixture(2, 700, seed=5)).to(DEV)
    aux = torch.from_numpy(make_mixture(2, 640, seed=6)).to(DEV)
    alen = torch.tensor(640.0)
    with torch.no_grad():
        est, logits = model(mix, aux, alen)
        d1, lg = model.separation(model.encoder(mix), model.encoder(aux), alen)
        est2 = model.decoder(d1.contiguous())
    assert rel_l2(lg.cpu().numpy(), logits.cpu().numpy()) < 2e-6
    assert rel_l2(est2.cpu().numpy(), est.cpu().numpy()) < 2e-6


def test_spe_cat_batch_b64_full_size():
    """BASELINE.json configs[2] at its own size: DPRNN-Spe (cat fusion, shipped config) on 64 x 24 000 samples with a 3-s reference
    utterance each, default engine settings (stream groups, tile heuristics, speaker branch at B = 64).  Utterances 0 / 31 / 32 / 63
    against the torch-CPU oracle, every utterance (estimate and speaker logits) against its own batch-of-one forward."""
    from conftest import build_model, rel_l2
    from oracle import dprnn_torch_cpu as otc
    from tss_with_dprnn_amd import engine
    assert not any(engine.SEQ_TILE.values()) and engine.GEMM_MODE == engine.DEFAULT_GEMM_MODE and engine.GROUP_SIZES is None
    model, sd, cfg = build_model("spe", dict(norm_type="ln", fusion_type="cat"), 3, DEV)
    B, T = 64, 24000
    x, a = make_mixture(B, T, seed=1234), make_mixture(B, T, seed=1235)
    xd, ad = torch.from_numpy(x).to(DEV), torch.from_numpy(a).to(DEV)
    alen = torch.tensor(float(T))
    before = dict(engine.SATURATION)
    with torch.no_grad():
        est, logits = model(xd, ad, alen)
        torch.cuda.synchronize()
        assert est.shape == (B, T) and logits.shape == (B, cfg["num_spks"]) and bool(torch.isfinite(est).all())
        worst = 0.0
        for b in range(B):
            e1, l1 = model(xd[b:b + 1], ad[b:b + 1], alen)
            r = float(((e1[0].double() - est[b].double()).norm() / e1[0].double().norm()).item())
            rl = float(((l1[0].double() - logits[b].double()).norm() / l1[0].double().norm()).item())
            worst = max(worst, r, rl)
            assert r <= 2e-6 and rl <= 2e-6, f"utterance {b}: batched vs batch-of-one rel-L2 {r:.3e} (logits {rl:.3e})"
    assert engine.SATURATION["fallbacks"] == before["fallbacks"]
    pick = [0, 31, 32, 63]
    we, wl = otc.spe_forward(otc.to_torch_state(sd), cfg, torch.from_numpy(x[pick]), torch.from_numpy(a[pick]), alen)
    r = rel_l2(est.cpu().numpy()[pick], we.numpy())
    rl = rel_l2(logits.cpu().numpy()[pick], wl.numpy())
    assert r <= 1e-5 and rl <= 1e-5, (r, rl)
    print(f"DPRNN-Spe (cat) B=64 x 24000: rel-L2 vs CPU oracle {r:.2e} (logits {rl:.2e}); worst batched-vs-single {worst:.2e}")


def test_a_nan_sample_reaches_the_caller_of_a_sub_module_as_nan():
    """The f16-pieces conversion clamps and its maximum tracking skips NaN: the sub-module entry points (which do not pass through the
    bottleneck's statistics, where full forwards are checked) test their input for non-finite values and repeat in the range-free
    arithmetic, so SingleRNN / DPRNNBlock return NaN where the reference would -- never finite numbers made from a clamped NaN."""
    import warnings
    from conftest import build_model
    from tss_with_dprnn_amd import engine
    model, _, _ = build_model("bss", dict(chunk_length=16, hop_length=8, n_repeats=1), 77, DEV)
    blk = model.separation.dprnn_blocks[0]
    x = torch.randn(3, 20, 128, device=DEV)
    x[1, 7, 5] = float("nan")
    before = dict(engine.SATURATION)
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        h = blk.intra_rnn(x)
    assert engine.SATURATION["fallbacks"] == before["fallbacks"] + 1 and engine.SATURATION["last_flags"][5] == 1
    assert torch.isnan(h[1]).any() and torch.isfinite(h[0]).all() and torch.isfinite(h[2]).all()
    xb = torch.randn(2, 128, 16, 9, device=DEV)
    xb[0, 3, 2, 1] = float("inf")
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        y = blk(xb)
    assert engine.SATURATION["fallbacks"] == before["fallbacks"] + 2
    assert not torch.isfinite(y[0]).all() and torch.isfinite(y[1]).all()
