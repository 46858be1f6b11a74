"""Multi-stream inference harness and the on-device SI-SDR / PIT metric."""
import numpy as np
import pytest
import torch

from conftest import build_model, rel_l2
from oracle.sisdr import pit_si_sdr as pit_ref, si_sdr as si_sdr_ref
from oracle.weights import make_mixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_pairwise_sisdr_matches_fp64_oracle():
    from tss_with_dprnn_amd.inference import pairwise_si_sdr, pit_si_sdr
    rng = np.random.default_rng(0)
    tgt = rng.standard_normal((3, 2, 8000)).astype(np.float32)
    est = (tgt[:, ::-1] + 0.05 * rng.standard_normal((3, 2, 8000))).astype(np.float32) + 0.3
    est[1] = est[1, ::-1]                                   # item 1 is not swapped
    pw = pairwise_si_sdr(torch.from_numpy(est).to(DEV), torch.from_numpy(tgt).to(DEV)).cpu().numpy()
    want = si_sdr_ref(est[:, :, None, :], tgt[:, None, :, :])
    assert np.abs(pw - want).max() < 1e-6
    score, perm, reordered = pit_si_sdr(torch.from_numpy(est).to(DEV), torch.from_numpy(tgt).to(DEV))
    wscore, wperm = pit_ref(est, tgt)
    assert np.abs(score.cpu().numpy() - wscore).max() < 1e-6
    assert (perm.cpu().numpy() == wperm).all()
    assert si_sdr_ref(reordered.cpu().numpy(), tgt).min() > 20


def test_separate_many_equals_one_by_one():
    from tss_with_dprnn_amd.inference import separate_many
    model, _, _ = build_model("bss", dict(chunk_length=50, hop_length=25, n_repeats=2), 21, DEV)
    lengths = [3000, 4100, 3000, 2500, 4100, 3000, 3333]
    wavs = [torch.from_numpy(make_mixture(1, n, seed=i)[0]).to(DEV) for i, n in enumerate(lengths)]
    outs = separate_many(model, wavs, max_streams=3)
    torch.cuda.synchronize()
    assert [tuple(o.shape) for o in outs] == [(2, n) for n in lengths]
    with torch.no_grad():
        for w, o in zip(wavs, outs):
            one = model(w[None])[0]
            assert rel_l2(o.cpu().numpy(), one.cpu().numpy()) < 1e-6


def test_separate_many_spe():
    from tss_with_dprnn_amd.inference import separate_many
    model, _, _ = build_model("spe", dict(chunk_length=50, hop_length=25, n_repeats=1, norm_type="ln"), 22, DEV)
    lengths = [3000, 3500, 3000]
    wavs = [torch.from_numpy(make_mixture(1, n, seed=i)[0]).to(DEV) for i, n in enumerate(lengths)]
    aux = [(torch.from_numpy(make_mixture(1, 2800, seed=50 + i)[0]).to(DEV), 2800.0) for i in range(3)]
    outs = separate_many(model, wavs, extra_inputs=aux)
    torch.cuda.synchronize()
    with torch.no_grad():
        for w, a, (est, logits) in zip(wavs, aux, outs):
            e1, l1 = model(w[None], a[0][None], torch.tensor(a[1]))
            assert rel_l2(est.cpu().numpy(), e1[0].cpu().numpy()) < 1e-6
            assert rel_l2(logits.cpu().numpy(), l1[0].cpu().numpy()) < 1e-6


def test_batched_inferencer_matches_per_utterance_loop(tmp_path):
    """BatchedInferencer.run (length buckets, streams) against the reference's B = 1 loop semantics
    (inferencer.py:54-70) evaluated with the CPU restatements of the PIT reordering and the si_sdr metric."""
    import json
    from oracle.sisdr import pit_si_sdr as pit_np, si_sdr_eval_metric
    from oracle.weights import make_sources
    from tss_with_dprnn_amd.inferencer import BatchedInferencer
    model, _, _ = build_model("bss", dict(chunk_length=16, hop_length=8, n_repeats=1), 21, DEV)
    lengths = [900, 1300, 900, 2047, 1300, 900]
    test_set = []
    for i, T in enumerate(lengths):
        src, mix = make_sources(1, T, seed=500 + i)
        test_set.append((torch.from_numpy(mix[0]), torch.from_numpy(src[0])))
    inf = BatchedInferencer(model, DEV, test_savedir=str(tmp_path))
    final = inf.run(test_set)
    want = []
    with torch.no_grad():
        for mix, src in test_set:
            out = model(mix.to(DEV).unsqueeze(0)).cpu().numpy()          # the reference loop: one utterance per call
            s = src.numpy()[None, :, : out.shape[-1]]
            _, perm = pit_np(out, s)
            reordered = out[:, np.argsort(perm[0])]
            sd = si_sdr_eval_metric(reordered, s).mean()
            inp = si_sdr_eval_metric(np.broadcast_to(mix.numpy()[None, None, : s.shape[-1]], s.shape), s).mean()
            want.append((sd, inp))
    want = np.array(want)
    assert abs(final["si_sdr"] - want[:, 0].mean()) < 1e-3
    assert abs(final["si_sdr_imp"] - (want[:, 0] - want[:, 1]).mean()) < 1e-3
    assert json.load(open(tmp_path / "final_metrics.json")) == final
    rows = open(tmp_path / "all_metrics.csv").read().strip().split("\n")
    assert len(rows) == len(lengths) + 1
    got = np.array([[float(v) for v in r.split(",")[1:]] for r in rows[1:]])
    assert np.allclose(got, want, atol=2e-3)



@pytest.fixture
def split_gate_gemm():
    """The ragged kernels exist in the split gate-GEMM modes only: with DPRNN_GEMM_MODE=0 in the environment the ragged tests
    run in mode 3."""
    from tss_with_dprnn_amd import engine
    old = engine.GEMM_MODE
    if old == 0:
        engine.GEMM_MODE = 3
    yield
    engine.GEMM_MODE = old

@pytest.mark.parametrize("kind", ["bss", "spe", "ira", "spe_att_bntrain", "bss+fold", "ira+fold"])
def test_ragged_batch_equals_one_by_one(kind, split_gate_gemm, monkeypatch):
    """SURVEY.md 8f-3 / inferencer_spe.py:27-32: utterances of different lengths through ONE launch per DPRNN path must
    give what the reference's one-at-a-time loop gives.  Lengths leave ragged tiles everywhere (chunk counts 11 .. 47,
    K = 40 chunk positions against 64-sequence tiles); the speaker branch keeps batch-of-one BatchNorm statistics."""
    from tss_with_dprnn_amd import engine, inference
    from conftest import build_model
    from oracle.weights import make_mixture
    if kind.endswith("+fold"):          # the Linear folded into the ragged recurrence launches (dprnn_rnn_lin_split_ragged_fwd: every workgroup cut at
        kind = kind[:-5]                # the middle of its own T); the one-by-one loop runs 16-sequence tiles, which have no folded form
        monkeypatch.setattr(engine, "LIN_FOLD", "split")
        monkeypatch.setattr(engine, "GEMM_MODE", 3)
    over = dict(chunk_length=40, hop_length=20, n_repeats=2)
    if kind != "bss":
        over.update(fusion_type="att" if "att" in kind else "cat", norm_type="ln")
    model, _, cfg = build_model("bss" if kind == "bss" else ("ira" if kind == "ira" else "spe"), over, 81, DEV)
    if "bntrain" in kind:
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.train()
    lens = [901, 377, 640, 233, 901, 512, 777, 300, 845, 199]
    wavs = [torch.from_numpy(make_mixture(1, n, seed=300 + i)[0]).to(DEV) for i, n in enumerate(lens)]
    extra = None
    if kind != "bss":
        alen = [400, 333, 512, 400, 280, 333, 450, 512, 300, 400]
        extra = [(torch.from_numpy(make_mixture(1, n, seed=400 + i)[0]).to(DEV), float(n)) for i, n in enumerate(alen)]
    assert engine.ragged_supported(model.separation)
    bn0 = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k or "num_batches" in k}
    got = inference.separate_many(model, wavs, extra_inputs=extra, ragged=True)
    bn_ragged = {k: v.clone() for k, v in model.state_dict().items() if k in bn0}
    model.load_state_dict({**model.state_dict(), **bn0})          # the one-by-one loop starts from the same running statistics
    with torch.no_grad():
        for i, w in enumerate(wavs):
            if extra is None:
                want = model(w[None])[0]
                assert got[i].shape == want.shape == (2, lens[i])
                r = float(((got[i].double() - want.double()).norm() / want.double().norm()).item())
                assert r <= 2e-6, (i, r)
            else:
                est, logits = model(w[None], extra[i][0][None], torch.tensor(extra[i][1]))
                r = float(((got[i][0].double() - est[0].double()).norm() / est[0].double().norm()).item())
                rl = float(((got[i][1].double() - logits[0].double()).norm() / logits[0].double().norm()).item())
                assert r <= 2e-6 and rl <= 2e-6, (i, r, rl)
    if "bntrain" in kind:                                         # ... and leaves the same running statistics behind
        for k, v in model.state_dict().items():
            if k in bn0 and v.dtype.is_floating_point:
                assert torch.allclose(v, bn_ragged[k], rtol=1e-5, atol=1e-7), k


@pytest.mark.parametrize("kind", ["bss", "ira"])
def test_several_ragged_batches_run_as_two_pipelines(kind, split_gate_gemm, monkeypatch):
    """inference.separate_ragged (round 5): more utterances than one ragged batch holds are processed as RAGGED_PIPELINES batches in flight
    on separate streams (here: batches of 3, so four of them on two streams); every output still equals its batch-of-one forward, and
    the sequential form (RAGGED_PIPELINES = 1) gives the same."""
    from tss_with_dprnn_amd import engine, inference
    from conftest import build_model
    from oracle.weights import make_mixture
    over = dict(chunk_length=40, hop_length=20, n_repeats=2)
    if kind != "bss":
        over.update(fusion_type="cat", norm_type="ln")
    model, _, cfg = build_model(kind if kind == "bss" else "ira", over, 84, DEV)
    lens = [901, 377, 640, 233, 901, 512, 777, 300, 845, 199]
    wavs = [torch.from_numpy(make_mixture(1, n, seed=800 + i)[0]).to(DEV) for i, n in enumerate(lens)]
    extra = None
    if kind != "bss":
        extra = [(torch.from_numpy(make_mixture(1, 400 + 7 * i, seed=850 + i)[0]).to(DEV), float(400 + 7 * i)) for i in range(len(lens))]
    monkeypatch.setattr(inference, "RAGGED_MAX_UTTS", 3)
    assert inference.RAGGED_PIPELINES == 2 and len(inference.ragged_chunks(list(range(10)), lens)) == 4
    got = inference.separate_many(model, wavs, extra_inputs=extra, ragged=True)
    monkeypatch.setattr(inference, "RAGGED_PIPELINES", 1)
    seq = inference.separate_many(model, wavs, extra_inputs=extra, ragged=True)
    first = lambda o: o[0] if isinstance(o, tuple) else o
    with torch.no_grad():
        for i, w in enumerate(wavs):
            want = model(w[None])[0] if extra is None else model(w[None], extra[i][0][None], torch.tensor(extra[i][1]))[0][0]
            for what, o in (("two pipelines", got[i]), ("one pipeline", seq[i])):
                r = float(((first(o).double() - want.double()).norm() / want.double().norm()).item())
                assert r <= 2e-6, (what, i, r)
            assert torch.equal(first(got[i]), first(seq[i]))


@pytest.mark.parametrize("hop", [40, 30, 27])
def test_ragged_batch_with_hops_above_half_a_chunk(hop, split_gate_gemm):
    """hop_length > chunk_length / 2 (the reference's hop_length is a free constructor argument, dprnn.py:128): an utterance's
    slab then has fewer than 2 L rows, so its mask-head scratch (U [L][2F] + V [2L][F]) no longer fits the share of H its S K rows
    own -- every utterance gets max(S K 2F, 4 L F) floats at its own offset.  Ten utterances whose heads run concurrently on the
    side streams, against the one-by-one loop."""
    from tss_with_dprnn_amd import inference
    from conftest import build_model
    from oracle.weights import make_mixture
    model, _, cfg = build_model("bss", dict(chunk_length=40, hop_length=hop, n_repeats=2), 83, DEV)
    lens = [901, 377, 640, 233, 901, 512, 777, 300, 845, 1203]
    for n in lens:      # the point of the test: the slab is smaller than the head's scratch
        L = n - 1
        S = (L + 40) // hop + 1
        assert S * 40 < 2 * L or n < 400
    wavs = [torch.from_numpy(make_mixture(1, n, seed=700 + i)[0]).to(DEV) for i, n in enumerate(lens)]
    got = inference.separate_many(model, wavs, ragged=True)
    torch.cuda.synchronize()
    with torch.no_grad():
        for i, w in enumerate(wavs):
            want = model(w[None])[0]
            assert got[i].shape == want.shape == (2, lens[i])
            r = float(((got[i].double() - want.double()).norm() / want.double().norm()).item())
            assert r <= 2e-6, (hop, i, r)


def test_ragged_full_config_against_cpu_oracle(split_gate_gemm):
    """Shipped depth / chunk geometry (6 blocks, K 250 / P 125), three lengths between 1.2 and 3.6 s in one ragged batch,
    against the torch-CPU oracle of each utterance."""
    from tss_with_dprnn_amd import inference
    from conftest import build_model
    from oracle import dprnn_torch_cpu as otc
    from oracle.weights import make_mixture
    model, sd, cfg = build_model("bss", dict(), 82, DEV)
    lens = [28800, 9600, 17001]
    xs = [make_mixture(1, n, seed=500 + i) for i, n in enumerate(lens)]
    got = inference.separate_many(model, [torch.from_numpy(x[0]).to(DEV) for x in xs], ragged=True)
    tsd = otc.to_torch_state(sd)
    for i, x in enumerate(xs):
        want = otc.tasnet_forward(tsd, cfg, torch.from_numpy(x)).numpy()[0]
        assert rel_l2(got[i].cpu().numpy(), want) <= 1e-5, (i, rel_l2(got[i].cpu().numpy(), want))


@pytest.mark.parametrize("kind", ["bss", "spe"])
def test_forward_async_keeps_forwards_in_flight_and_checks_them_at_the_ticket(kind):
    """Round 4: engine.forward_async enqueues a forward WITHOUT the stream synchronisation of the saturation check and returns
    (output, ticket); ticket.result() performs the 32-byte read.  Three forwards in flight (eager, capture + first replay, replay),
    collected afterwards: each equals the synchronous forward of the same input, and every one was checked exactly once."""
    from tss_with_dprnn_amd import engine
    over = dict(chunk_length=50, hop_length=25, n_repeats=2)
    if kind == "spe":
        over.update(fusion_type="cat", norm_type="ln")
    model, _, _ = build_model(kind, over, 31, DEV)
    engine.drop_graphs()

    def inputs(seed):
        a = [torch.from_numpy(make_mixture(3, 3300, seed=seed)).to(DEV)]
        if kind == "spe":
            a += [torch.from_numpy(make_mixture(3, 3000, seed=seed + 50)).to(DEV), torch.tensor(3000.0)]
        return a
    first = lambda o: (o[0] if isinstance(o, tuple) else o)
    before = dict(engine.SATURATION)
    with torch.no_grad():
        pending = [engine.forward_async(model, *inputs(s)) for s in (1, 2, 3, 1)]
        assert engine.SATURATION["checks"] == before["checks"], "nothing is checked before a ticket is asked"
        got = [first(t.result()).cpu().numpy() for _, t in pending]
        assert engine.SATURATION["checks"] == before["checks"] + 4 and engine.SATURATION["fallbacks"] == before["fallbacks"]
        for (out, t), g in zip(pending, got):
            assert t.ready() and first(t.result()) is first(out)              # no fall-back: the ticket hands back the enqueued output
        want = [first(model(*inputs(s))).cpu().numpy() for s in (1, 2, 3, 1)]
    for g, w in zip(got, want):
        assert rel_l2(g, w) < 1e-6
    assert np.array_equal(got[0], got[3]) or rel_l2(got[0], got[3]) < 1e-6
    with pytest.raises(NotImplementedError):
        engine.forward_async(model, *inputs(1))                                # gradients enabled: not an inference call
    engine.drop_graphs()


def test_a_flagged_async_forward_still_returns_the_range_free_result():
    """The asynchronous path keeps the safety of the synchronous one: a forward whose values leave the f16 range (norm gain x 3000,
    adaptive scale off so that it saturates every time) comes back from ticket.result() as the mode-1 result, eager and replayed."""
    import warnings
    from tss_with_dprnn_amd import engine
    model, sd, cfg = build_model("bss", dict(chunk_length=50, hop_length=25, n_repeats=2), 21, DEV)
    sd = dict(sd)
    key = "separation.dprnn_blocks.0.intra_norm.weight"
    sd[key] = (np.asarray(sd[key]) * 3000.0).astype(np.float32)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    x = torch.from_numpy(make_mixture(3, 3100, seed=8)).to(DEV)
    engine.drop_graphs()
    old = engine.ADAPTIVE_SCALE
    engine.ADAPTIVE_SCALE = False
    try:
        with torch.no_grad(), engine.use_gemm_mode(1):
            want = model(x).cpu().numpy()
        before = dict(engine.SATURATION)
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            tickets = []
            for _ in range(2):                                              # (a third would pin the model to mode 1)
                out, t = engine.forward_async(model, x)
                tickets.append((out, t))
                engine.unpin(model)
            got = [t.result().cpu().numpy() for _, t in tickets]
            engine.unpin(model)
        assert engine.SATURATION["fallbacks"] == before["fallbacks"] + 2
        for g in got:
            assert rel_l2(g, want) < 1e-6
    finally:
        engine.ADAPTIVE_SCALE = old
        engine.unpin(model)
        engine.drop_graphs()


def test_separate_many_async_equals_separate_many():
    from tss_with_dprnn_amd import engine
    from tss_with_dprnn_amd.inference import separate_many, separate_many_async
    model, _, _ = build_model("bss", dict(chunk_length=50, hop_length=25, n_repeats=2), 21, DEV)
    lengths = [3000, 4100, 3000, 2500, 4100, 3000, 3333, 2800, 3900, 2500]
    wavs = [torch.from_numpy(make_mixture(1, n, seed=i)[0]).to(DEV) for i, n in enumerate(lengths)]
    with torch.no_grad():
        t1 = separate_many_async(model, wavs[:8])                              # ragged (>= 8 utterances)
        t2 = separate_many_async(model, wavs[3:], max_streams=3, ragged=False)  # per-length pipelines
        a, b = t1.result(), t2.result()
        want = separate_many(model, wavs)
    torch.cuda.synchronize()
    for o, w in zip(a, want[:8]):
        assert rel_l2(o.cpu().numpy(), w.cpu().numpy()) < 1e-6
    for o, w in zip(b, want[3:]):
        assert rel_l2(o.cpu().numpy(), w.cpu().numpy()) < 1e-6
