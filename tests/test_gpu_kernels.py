"""Stage-level GPU tests: each C-ABI entry point against the numpy oracle on seeded inputs, including
ragged sizes (sequence counts that do not fill a tile, utterance boundaries inside a row tile)."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import dprnn_numpy as onp
from oracle.weights import make_state, full_cfg

import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import engine

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _st():
    return engine.Stage(torch.device(DEV))


@pytest.fixture(autouse=True)
def _exact_gemm_mode():
    """The gate-GEMM mode is process-wide library state: every kernel test starts (and leaves the library) in mode 0."""
    lib = _st().lib
    lib.dprnn_set_gemm_mode(0)
    yield
    lib.dprnn_set_gemm_mode(0)


_KEEP = []


def _t(a):
    """Host array -> device tensor that stays alive (its data_ptr() is handed to the C ABI)."""
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    _KEEP.append(t)
    if len(_KEEP) > 256:
        torch.cuda.synchronize()
        del _KEEP[:128]
    return t


def _lstm_sd(seed, bidir=True):
    sd = make_state("bss", dict(n_repeats=1, bidirectional=bidir), seed)
    p = "separation.dprnn_blocks.0." + ("intra_rnn" if bidir else "inter_rnn")
    return sd, p


@pytest.mark.parametrize("seq_tile", [16, -16, 32, 48, 64])
@pytest.mark.parametrize("mode", ["intra", "inter"])
def test_bilstm_matches_oracle(seq_tile, mode):
    B, S, K = 2, 7, 11                     # 14 / 22 sequences: never a multiple of a tile
    rng = np.random.default_rng(abs(seq_tile) + (0 if mode == "intra" else 1))
    X = rng.standard_normal((B, S, K, 128)).astype(np.float32)
    sd, p = _lstm_sd(42)
    if mode == "intra":
        seq = X.reshape(B * S, K, 128)
    else:
        seq = X.transpose(0, 2, 1, 3).reshape(B * K, S, 128)
    want = onp.single_rnn(sd, p, seq, "LSTM", True)
    want = want.reshape(B, S, K, 256) if mode == "intra" else want.reshape(B, K, S, 256).transpose(0, 2, 1, 3)
    st = _st()
    packed = torch.empty(2 * engine.PACK_PER_DIR, device=DEV)
    bsum = torch.empty(2 * 512, device=DEV)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    w = [_t(sd[f"{p}.rnn.{n}"]) for n in names] + [_t(sd[f"{p}.rnn.{n}_reverse"]) for n in names]
    st.call("dprnn_lstm_pack", *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr())
    Xd = _t(X)
    H = torch.full((B, S, K, 256), float("nan"), device=DEV)
    if mode == "intra":
        st.call("dprnn_bilstm_fwd", Xd.data_ptr(), packed.data_ptr(), bsum.data_ptr(), H.data_ptr(),
                B * S, K, B * S, 0, K, 1, 2, seq_tile)
    else:
        st.call("dprnn_bilstm_fwd", Xd.data_ptr(), packed.data_ptr(), bsum.data_ptr(), H.data_ptr(),
                B * K, S, K, S * K, 1, K, 2, seq_tile)
    torch.cuda.synchronize()
    got = H.cpu().numpy()
    assert np.isfinite(got).all()
    assert rel_l2(got, want) < 2e-6


@pytest.mark.parametrize("mode,seq_tile", [("intra", 32), ("inter", 16), ("intra", -16), ("inter", 64)])
def test_bilstm_with_pending_norm(mode, seq_tile):
    """dprnn_bilstm_norm_fwd: x' = X + norm(Z) applied in the loader, x' written once, H computed from x'."""
    B, S, K = 3, 6, 9
    rng = np.random.default_rng(abs(seq_tile))
    X = rng.standard_normal((B, S, K, 128)).astype(np.float32)
    Zr = (rng.standard_normal((B, S, K, 128)) * 1.3 + 0.2).astype(np.float32)
    gamma = (1 + 0.1 * rng.standard_normal(128)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(128)).astype(np.float32)
    eps = 1e-5
    zb = Zr.reshape(B, -1).astype(np.float64)
    mean, var = zb.mean(1)[:, None, None, None], zb.var(1)[:, None, None, None]
    Xn = (X + ((Zr - mean) / np.sqrt(var + eps)) * gamma + beta).astype(np.float32)
    sd, p = _lstm_sd(44)
    if mode == "intra":
        want = onp.single_rnn(sd, p, Xn.reshape(B * S, K, 128), "LSTM", True).reshape(B, S, K, 256)
    else:
        want = onp.single_rnn(sd, p, Xn.transpose(0, 2, 1, 3).reshape(B * K, S, 128), "LSTM", True)
        want = want.reshape(B, K, S, 256).transpose(0, 2, 1, 3)
    st = _st()
    packed = torch.empty(2 * engine.PACK_PER_DIR, device=DEV)
    bsum = torch.empty(2 * 512, device=DEV)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    w = [_t(sd[f"{p}.rnn.{n}"]) for n in names] + [_t(sd[f"{p}.rnn.{n}_reverse"]) for n in names]
    st.call("dprnn_lstm_pack", *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr())
    stats = _t(np.stack([zb.sum(1), (zb ** 2).sum(1)], axis=1))
    Xd, Zd = _t(X), _t(Zr)
    Xo = torch.full((B, S, K, 128), float("nan"), device=DEV)
    H = torch.full((B, S, K, 256), float("nan"), device=DEV)
    addr = (B * S, K, B * S, 0, K, 1) if mode == "intra" else (B * K, S, K, S * K, 1, K)
    st.call("dprnn_bilstm_norm_fwd", Xd.data_ptr(), Zd.data_ptr(), stats.data_ptr(), _t(gamma).data_ptr(), _t(beta).data_ptr(),
            eps, S if mode == "intra" else K, S * K, Xo.data_ptr(), packed.data_ptr(), bsum.data_ptr(), H.data_ptr(),
            *addr, 2, seq_tile)
    torch.cuda.synchronize()
    assert rel_l2(Xo.cpu().numpy(), Xn) < 1e-6
    assert rel_l2(H.cpu().numpy(), want) < 2e-6


@pytest.mark.parametrize("mode,tol", [(1, 2e-6)])
@pytest.mark.parametrize("path", ["intra", "inter"])
def test_bilstm_split_gemm_modes(mode, tol, path):
    """bf16-split gate GEMM (recur_split.hip, mode 1): three pieces, six products ~ fp32 accuracy."""
    B, S, K = 3, 9, 23              # 27 / 69 sequences: ragged against 32- and 64-sequence tiles
    rng = np.random.default_rng(mode)
    X = rng.standard_normal((B, S, K, 128)).astype(np.float32)
    sd, p = _lstm_sd(45)
    if path == "intra":
        want = onp.single_rnn(sd, p, X.reshape(B * S, K, 128), "LSTM", True).reshape(B, S, K, 256)
    else:
        want = onp.single_rnn(sd, p, X.transpose(0, 2, 1, 3).reshape(B * K, S, 128), "LSTM", True)
        want = want.reshape(B, K, S, 256).transpose(0, 2, 1, 3)
    st = _st()
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device=DEV)
    bsum = torch.empty(2 * 512, device=DEV)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    w = [_t(sd[f"{p}.rnn.{n}"]) for n in names] + [_t(sd[f"{p}.rnn.{n}_reverse"]) for n in names]
    Xd = _t(X)
    H = torch.full((B, S, K, 256), float("nan"), device=DEV)
    addr = (B * S, K, B * S, 0, K, 1) if path == "intra" else (B * K, S, K, S * K, 1, K)
    st.lib.dprnn_set_gemm_mode(mode)
    try:
        assert st.lib.dprnn_get_gemm_mode() == mode
        st.call("dprnn_rnn_pack", 0, *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr())
        st.call("dprnn_rnn_fwd", 0, Xd.data_ptr(), 0, 0, 0, 0, 0.0, 1, 1, 0, packed.data_ptr(), bsum.data_ptr(), H.data_ptr(),
                *addr, 2, 0)
        torch.cuda.synchronize()
    finally:
        st.lib.dprnn_set_gemm_mode(0)
    got = H.cpu().numpy()
    assert np.isfinite(got).all()
    err = rel_l2(got, want)
    print(f"split mode {mode} {path}: rel-L2 {err:.2e}")
    assert err < tol


@pytest.mark.parametrize("seq_tile", [32, 2064, 2080, 2096])
@pytest.mark.parametrize("mode,tol", [(1, 2e-6)])
@pytest.mark.parametrize("path,fused", [("intra", False), ("inter", False), ("intra", True), ("inter", True)])
def test_bilstm_split_wide_tiles(seq_tile, mode, tol, path, fused):
    """The bf16-split kernels (two 32-sequence workgroups per CU; the eight-wave form with 64 / 80 / 96 sequences, one workgroup
    per CU, swizzled image): every tile size against the numpy oracle, with sequence counts that leave ragged last tiles, tiles
    straddling utterances on the inter-chunk path, and the previous path's norm folded into the loader."""
    B, S, K = 3, 37, 41             # 111 intra sequences of 41 steps, 123 inter sequences of 37 steps
    rng = np.random.default_rng(10000 * mode + abs(seq_tile))
    X = rng.standard_normal((B, S, K, 128)).astype(np.float32)
    sd, p = _lstm_sd(46)
    Xn = X
    if fused:
        Zr = (rng.standard_normal((B, S, K, 128)) * 1.3 + 0.2).astype(np.float32)
        gamma = (1 + 0.1 * rng.standard_normal(128)).astype(np.float32)
        beta = (0.1 * rng.standard_normal(128)).astype(np.float32)
        zb = Zr.reshape(B, -1).astype(np.float64)
        mean, var = zb.mean(1)[:, None, None, None], zb.var(1)[:, None, None, None]
        Xn = (X + ((Zr - mean) / np.sqrt(var + 1e-5)) * gamma + beta).astype(np.float32)
    if path == "intra":
        want = onp.single_rnn(sd, p, Xn.reshape(B * S, K, 128), "LSTM", True).reshape(B, S, K, 256)
    else:
        want = onp.single_rnn(sd, p, Xn.transpose(0, 2, 1, 3).reshape(B * K, S, 128), "LSTM", True)
        want = want.reshape(B, K, S, 256).transpose(0, 2, 1, 3)
    st = _st()
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device=DEV)
    bsum = torch.empty(2 * 512, device=DEV)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    w = [_t(sd[f"{p}.rnn.{n}"]) for n in names] + [_t(sd[f"{p}.rnn.{n}_reverse"]) for n in names]
    Xd = _t(X)
    H = torch.full((B, S, K, 256), float("nan"), device=DEV)
    Xo = torch.full((B, S, K, 128), float("nan"), device=DEV)
    addr = (B * S, K, B * S, 0, K, 1) if path == "intra" else (B * K, S, K, S * K, 1, K)
    norm = (0, 0, 0, 0, 0.0, 1, 1, 0)
    if fused:
        stats = _t(np.stack([zb.sum(1), (zb ** 2).sum(1)], axis=1))
        Zd, gd, bd = _t(Zr), _t(gamma), _t(beta)
        norm = (Zd.data_ptr(), stats.data_ptr(), gd.data_ptr(), bd.data_ptr(), 1e-5, S if path == "intra" else K, S * K, Xo.data_ptr())
    st.lib.dprnn_set_gemm_mode(mode)
    try:
        st.call("dprnn_rnn_pack", 0, *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr())
        st.call("dprnn_rnn_fwd", 0, Xd.data_ptr(), *norm, packed.data_ptr(), bsum.data_ptr(), H.data_ptr(), *addr, 2, seq_tile)
        torch.cuda.synchronize()
    finally:
        st.lib.dprnn_set_gemm_mode(0)
    got = H.cpu().numpy()
    assert np.isfinite(got).all()
    err = rel_l2(got, want)
    assert err < tol, f"tile {seq_tile} mode {mode} {path} fused={fused}: rel-L2 {err:.2e}"
    if fused:
        assert rel_l2(Xo.cpu().numpy(), Xn) < 1e-6


@pytest.mark.parametrize("seq_tile", [2016, 2032, 2048, 2064, 2080, 2096])
@pytest.mark.parametrize("path,fused", [("intra", False), ("inter", False), ("intra", True), ("inter", True)])
@pytest.mark.parametrize("wscale,xscale", [(1.0, 1.0), (37.0, 0.02), (1e-3, 40.0)])
def test_bilstm_f16_pieces(seq_tile, path, fused, wscale, xscale):
    """Gate-GEMM mode 3 (two f16 pieces of 2^e-scaled operands, three products, eight-wave kernels): fp32-class accuracy
    against the numpy oracle at every tile size, with the pending norm folded in, and with weights / activations far from
    unit scale (the per-LSTM weight exponent and the f16 subnormal second pieces are what keeps those exact)."""
    B, S, K = 3, 37, 41
    rng = np.random.default_rng(seq_tile + int(10 * wscale))
    X = (xscale * rng.standard_normal((B, S, K, 128))).astype(np.float32)
    sd, p = _lstm_sd(46)
    sd = {k: (v * np.float32(wscale) if ".rnn.weight_" in k else v) for k, v in sd.items()}
    Xn = X
    if fused:
        Zr = (rng.standard_normal((B, S, K, 128)) * 1.3 + 0.2).astype(np.float32)
        gamma = (xscale * (1 + 0.1 * rng.standard_normal(128))).astype(np.float32)
        beta = (xscale * 0.1 * rng.standard_normal(128)).astype(np.float32)
        zb = Zr.reshape(B, -1).astype(np.float64)
        mean, var = zb.mean(1)[:, None, None, None], zb.var(1)[:, None, None, None]
        Xn = (X + ((Zr - mean) / np.sqrt(var + 1e-5)) * gamma + beta).astype(np.float32)
    if path == "intra":
        want = onp.single_rnn(sd, p, Xn.reshape(B * S, K, 128), "LSTM", True).reshape(B, S, K, 256)
    else:
        want = onp.single_rnn(sd, p, Xn.transpose(0, 2, 1, 3).reshape(B * K, S, 128), "LSTM", True)
        want = want.reshape(B, K, S, 256).transpose(0, 2, 1, 3)
    st = _st()
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device=DEV)
    bsum = torch.empty(2 * 512, device=DEV)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    w = [_t(sd[f"{p}.rnn.{n}"]) for n in names] + [_t(sd[f"{p}.rnn.{n}_reverse"]) for n in names]
    Xd = _t(X)
    H = torch.full((B, S, K, 256), float("nan"), device=DEV)
    Xo = torch.full((B, S, K, 128), float("nan"), device=DEV)
    addr = (B * S, K, B * S, 0, K, 1) if path == "intra" else (B * K, S, K, S * K, 1, K)
    norm = (0, 0, 0, 0, 0.0, 1, 1, 0)
    if fused:
        stats = _t(np.stack([zb.sum(1), (zb ** 2).sum(1)], axis=1))
        Zd, gd, bd = _t(Zr), _t(gamma), _t(beta)
        norm = (Zd.data_ptr(), stats.data_ptr(), gd.data_ptr(), bd.data_ptr(), 1e-5, S if path == "intra" else K, S * K, Xo.data_ptr())
    results = {}
    for mode in (3, 0):                     # the exact fp32 MFMA kernels on the same input: the yardstick
        st.lib.dprnn_set_gemm_mode(mode)
        st.call("dprnn_rnn_pack", 0, *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr())
        st.call("dprnn_rnn_fwd", 0, Xd.data_ptr(), *norm, packed.data_ptr(), bsum.data_ptr(), H.data_ptr(), *addr, 2,
                seq_tile if mode else 32)
        torch.cuda.synchronize()
        got = H.cpu().numpy()
        assert np.isfinite(got).all()
        results[mode] = rel_l2(got, want)
    assert results[3] < max(1e-6, 3 * results[0]), f"tile {seq_tile} {path} fused={fused} w x{wscale} x x{xscale}: f16 pieces " \
        f"{results[3]:.2e}, exact fp32 kernels {results[0]:.2e}"
    if fused:
        assert rel_l2(Xo.cpu().numpy(), Xn) < 1e-6


def _sat_flags(st, fn):
    """fn() between dprnn_saturation_begin and dprnn_saturation_flags on this thread: (return code, the eight flags)."""
    import ctypes
    flags = torch.empty(8, dtype=torch.int32, device=DEV)
    host = (ctypes.c_int32 * 8)()
    assert st.lib.dprnn_saturation_begin(flags.data_ptr(), st.stream) == 0
    try:
        fn()
        rc = st.lib.dprnn_saturation_flags(ctypes.addressof(host), st.stream)
    finally:
        st.lib.dprnn_saturation_begin(0, 0)
    return rc, list(host)


@pytest.mark.parametrize("seq_tile", [2016, 2064, 2080])
@pytest.mark.parametrize("huge", [5000.0, 1e6, float("inf")])
def test_bilstm_f16_pieces_flag_values_beyond_the_f16_range(seq_tile, huge):
    """|x| > 4062 does not fit 2^4 x in f16: the piece conversion clamps AND raises saturation flag 0 (Inf likewise; a NaN sample is
    flagged where every forward starts, by the statistics dprnn_bottleneck_prep reads -- test_gpu_parity.py), so that the host can
    repeat the call in a range-free mode; the same launch on in-range input leaves every flag clear, and mode 1
    (bf16 pieces: fp32's exponent) on the out-of-range input matches the oracle -- what the fall-back returns."""
    B, S, K = 1, 3, 40
    rng = np.random.default_rng(5)
    X = rng.standard_normal((B, S, K, 128)).astype(np.float32)
    sd, p = _lstm_sd(46)
    sd = {k: (v * np.float32(1e-3) if ".rnn.weight_ih" in k else v) for k, v in sd.items()}    # small W_ih: 5000 |w| does NOT saturate the gates
    st = _st()
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device=DEV)
    bsum = torch.empty(2 * 512, device=DEV)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    w = [_t(sd[f"{p}.rnn.{n}"]) for n in names] + [_t(sd[f"{p}.rnn.{n}_reverse"]) for n in names]
    H = torch.full((B, S, K, 256), float("nan"), device=DEV)

    def run(Xn, mode, tile):
        Xd = _t(Xn)
        st.call("dprnn_rnn_pack_ex", mode, 0, *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr())
        st.call("dprnn_rnn_fwd_ex", mode, 0, Xd.data_ptr(), 0, 0, 0, 0, 0.0, 1, 1, 0, packed.data_ptr(), bsum.data_ptr(), H.data_ptr(),
                B * S, K, B * S, 0, K, 1, 2, tile)
        torch.cuda.synchronize()
    rc, flags = _sat_flags(st, lambda: run(X, 3, seq_tile))
    assert rc == 0 and not any(flags), flags
    Xh = X.copy()
    Xh[0, 1, 7, 5] = huge
    rc, flags = _sat_flags(st, lambda: run(Xh, 3, seq_tile))
    assert rc == 1 and flags[0] == 1 and not any(flags[1:]), flags
    got3 = H.cpu().numpy().copy()
    want = onp.single_rnn(sd, p, Xh.reshape(B * S, K, 128), "LSTM", True).reshape(B, S, K, 256)
    assert rel_l2(got3[0, [0, 2]], want[0, [0, 2]]) < 1e-6          # other chunks (sequences) are independent of the row
    if np.isfinite(huge):
        if huge == 5000.0:      # the clamped launch is WRONG on the row's own sequence (that is why the flag matters) ...
            assert rel_l2(got3[0, 1], want[0, 1]) > 1e-4
        rc, flags = _sat_flags(st, lambda: run(Xh, 1, 2064))            # ... and the fall-back arithmetic is right
        assert rc == 0 and not any(flags)
        assert rel_l2(H.cpu().numpy(), want) < 2e-6


@pytest.mark.parametrize("seq_tile", [2016, 2064])
@pytest.mark.parametrize("amp,want_s", [(3000.0, 1), (3.0e5, -6), (3.0, 4)])
def test_bilstm_f16_pieces_follow_the_activation_scale_record(seq_tile, amp, want_s):
    """Round 4: the x scale 2^s of the f16-pieces recurrence comes from a device record (dprnn_scale_begin / _slot / _update).  First
    launch under a fresh record: s = 4, an input of amplitude `amp` saturates (beyond 4062) and leaves its maximum in the record;
    after dprnn_scale_update the same launch (and the pack, which folds 2^(4 - s) into W_ih) runs at s = clamp(13 - exponent, -8, 4):
    no flag, fp32-class accuracy against the numpy oracle -- h keeps its 2^4 scale whatever x does."""
    B, S, K = 2, 5, 33
    rng = np.random.default_rng(11)
    X = rng.standard_normal((B, S, K, 128)).astype(np.float32)
    X *= np.float32(amp / np.abs(X).max())
    sd, p = _lstm_sd(46)
    sd = {k: (v * np.float32(1.0 / amp) if ".rnn.weight_ih" in k else v) for k, v in sd.items()}      # gates stay in their active range
    want = onp.single_rnn(sd, p, X.reshape(B * S, K, 128), "LSTM", True).reshape(B, S, K, 256)
    st = _st()
    lib = st.lib
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device=DEV)
    bsum = torch.empty(2 * 512, device=DEV)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    w = [_t(sd[f"{p}.rnn.{n}"]) for n in names] + [_t(sd[f"{p}.rnn.{n}_reverse"]) for n in names]
    H = torch.full((B, S, K, 256), float("nan"), device=DEV)
    Xd = _t(X)
    rec = torch.empty(64, device=DEV)
    st.call("dprnn_scale_init", rec.data_ptr())
    slot = 7

    def run():
        assert lib.dprnn_scale_begin(rec.data_ptr()) == 0 and lib.dprnn_scale_slot(slot) == 0
        try:
            st.call("dprnn_rnn_pack_ex", 3, 0, *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr())
            st.call("dprnn_rnn_fwd_ex", 3, 0, Xd.data_ptr(), 0, 0, 0, 0, 0.0, 1, 1, 0, packed.data_ptr(), bsum.data_ptr(), H.data_ptr(),
                    B * S, K, B * S, 0, K, 1, 2, seq_tile)
            st.call("dprnn_scale_update", rec.data_ptr())
        finally:
            lib.dprnn_scale_begin(0)
        torch.cuda.synchronize()
    rc, flags = _sat_flags(st, run)
    assert (rc == 1 and flags[0] == 1) == (amp > 4062.0 / 1.0001), (amp, flags)
    state = rec.cpu().numpy()
    assert abs(state[slot] - max(256.0, amp)) <= 1e-3 * amp and state[32 + slot] == 0.0          # sticky maximum, tracker reset
    assert all(state[i] == 256.0 for i in range(32) if i != slot)
    rc, flags = _sat_flags(st, run)
    assert rc == 0 and not any(flags), flags
    got = H.cpu().numpy()
    assert rel_l2(got, want) < 1e-6, (amp, rel_l2(got, want))
    import math
    assert max(-8, min(4, 13 - math.frexp(float(state[slot]))[1])) == want_s
    # without a record the library behaves as before: s = 4
    rc, flags = _sat_flags(st, lambda: (st.call("dprnn_rnn_pack_ex", 3, 0, *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr()),
                                         st.call("dprnn_rnn_fwd_ex", 3, 0, Xd.data_ptr(), 0, 0, 0, 0, 0.0, 1, 1, 0, packed.data_ptr(), bsum.data_ptr(),
                                                 H.data_ptr(), B * S, K, B * S, 0, K, 1, 2, seq_tile), torch.cuda.synchronize()))
    assert (rc == 1) == (amp > 4062.0)


def test_packed_weights_carry_their_format():
    """The layout dprnn_rnn_pack writes depends on the cell and on the gate-GEMM mode; the library records it with the buffer and
    every launch checks it: a launch under another mode, on a buffer nobody packed, or on the backward's pack is an error
    (DPRNN_EFORMAT = -3) with a message, never numbers."""
    B, S, K = 1, 2, 8
    sd, p = _lstm_sd(46)
    st = _st()
    lib = st.lib
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device=DEV)
    other = torch.empty(2 * engine.PACK_PER_DIR_MAX, device=DEV)
    bsum = torch.empty(2 * 512, device=DEV)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    w = [_t(sd[f"{p}.rnn.{n}"]) for n in names] + [_t(sd[f"{p}.rnn.{n}_reverse"]) for n in names]
    X = _t(np.random.default_rng(0).standard_normal((B, S, K, 128)).astype(np.float32))
    H = torch.zeros(B, S, K, 256, device=DEV)
    addr = (B * S, K, B * S, 0, K, 1)

    def launch(mode, buf, tile):
        return lib.dprnn_rnn_fwd_ex(mode, 0, X.data_ptr(), 0, 0, 0, 0, 0.0, 1, 1, 0, buf.data_ptr(), bsum.data_ptr(), H.data_ptr(),
                                    *addr, 2, tile, st.stream)
    for pack_mode, tile in ((0, 32), (1, 2064), (3, 2064)):
        st.call("dprnn_rnn_pack_ex", pack_mode, 0, *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr())
        for launch_mode, ltile in ((0, 32), (1, 2064), (3, 2064)):
            rc = launch(launch_mode, packed, ltile)
            if launch_mode == pack_mode:
                assert rc == 0
            else:
                assert rc == -3 and b"pack and launch under the same mode" in lib.dprnn_last_error()
        # the thread-default entry points check the same way
        lib.dprnn_set_gemm_mode(3 if pack_mode != 3 else 0)
        try:
            rc = lib.dprnn_rnn_fwd(0, X.data_ptr(), 0, 0, 0, 0, 0.0, 1, 1, 0, packed.data_ptr(), bsum.data_ptr(), H.data_ptr(),
                                   *addr, 2, 2064 if pack_mode != 3 else 32, st.stream)
        finally:
            lib.dprnn_set_gemm_mode(0)
        assert rc == -3
    lib.dprnn_pack_forget(other.data_ptr())
    assert launch(0, other, 32) == -3 and b"was not filled" in lib.dprnn_last_error()
    # a GRU pack is not an LSTM pack
    gsd, gp = _lstm_sd(3)
    st.call("dprnn_rnn_pack_ex", 0, 0, *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr())
    rc = lib.dprnn_rnn_fwd_ex(0, 1, X.data_ptr(), 0, 0, 0, 0, 0.0, 1, 1, 0, packed.data_ptr(), bsum.data_ptr(), H.data_ptr(),
                              *addr, 2, 32, st.stream)
    assert rc == -3
    assert lib.dprnn_rnn_pack_ex(2, 0, *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr(), st.stream) == -1    # mode 2 is gone
    torch.cuda.synchronize()


def test_bilstm_unidirectional():
    B, S, K = 1, 5, 9
    X = np.random.default_rng(3).standard_normal((B, S, K, 128)).astype(np.float32)
    sd, p = _lstm_sd(7, bidir=False)
    want = onp.single_rnn(sd, p, X.transpose(0, 2, 1, 3).reshape(B * K, S, 128), "LSTM", False)
    want = want.reshape(B, K, S, 128).transpose(0, 2, 1, 3)
    st = _st()
    packed = torch.empty(engine.PACK_PER_DIR, device=DEV)
    bsum = torch.empty(512, device=DEV)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    w = [_t(sd[f"{p}.rnn.{n}"]) for n in names]
    st.call("dprnn_lstm_pack", *[t.data_ptr() for t in w], 0, 0, 0, 0, packed.data_ptr(), bsum.data_ptr())
    H = torch.full((B, S, K, 128), float("nan"), device=DEV)
    st.call("dprnn_bilstm_fwd", _t(X).data_ptr(), packed.data_ptr(), bsum.data_ptr(), H.data_ptr(),
            B * K, S, K, S * K, 1, K, 1, 0)
    torch.cuda.synchronize()
    assert rel_l2(H.cpu().numpy(), want) < 2e-6


def test_bilstm_rejects_tiles_wider_than_the_32_bit_offsets():
    """The step loop addresses a tile with 32-bit per-lane byte offsets below 2^31: a geometry
    whose tile could span more is refused by the launcher before anything is launched.  The buffers are complete for
    the single sequence described, so even a launcher that failed to refuse would stay in bounds."""
    st = _st()
    X = torch.zeros(128, device=DEV)
    H = torch.zeros(256, device=DEV)
    packed = torch.zeros(2 * engine.PACK_PER_DIR_MAX, device=DEV)
    bsum = torch.zeros(2 * 512, device=DEV)
    with pytest.raises(RuntimeError, match="2 GiB"):
        st.call("dprnn_bilstm_fwd", X.data_ptr(), packed.data_ptr(), bsum.data_ptr(), H.data_ptr(),
                1, 1, 1, 1 << 40, 1, 1, 2, 0)           # one sequence, one step; outer stride of 2^40 rows
    torch.cuda.synchronize()


@pytest.mark.parametrize("kdim", [256, 128])
@pytest.mark.parametrize("mode,rows_per", [(0, 77), (3, 77), (0, 2111), (3, 2111)])
def test_proj_stats_and_norm_residual(kdim, mode, rows_per):
    """mode 3: the f16-pieces row GEMM (rowgemm_f16_kernel) takes the projection over; same tolerance as the fp32 MFMA form."""
    B = 3                                  # utterance boundaries fall inside 64-row tiles
    M = B * rows_per
    rng = np.random.default_rng(kdim)
    H = rng.standard_normal((M, kdim)).astype(np.float32)
    W = (rng.standard_normal((128, kdim)) / np.sqrt(kdim)).astype(np.float32)
    b = rng.standard_normal(128).astype(np.float32)
    X = rng.standard_normal((M, 128)).astype(np.float32)
    gamma = (1 + 0.1 * rng.standard_normal(128)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(128)).astype(np.float32)
    Zw = H.astype(np.float64) @ W.T.astype(np.float64) + b
    st = _st()
    Z = torch.empty(M, 128, device=DEV)
    stats = torch.zeros(B, 2, device=DEV, dtype=torch.float64)
    st.lib.dprnn_set_gemm_mode(mode)
    st.call("dprnn_proj_stats_fwd", _t(H).data_ptr(), _t(W).data_ptr(), _t(b).data_ptr(), Z.data_ptr(),
            stats.data_ptr(), M, rows_per, kdim)
    torch.cuda.synchronize()
    st.lib.dprnn_set_gemm_mode(0)
    assert rel_l2(Z.cpu().numpy(), Zw) < 1e-6
    zb = Zw.reshape(B, -1)
    assert np.allclose(stats.cpu().numpy()[:, 0], zb.sum(1), rtol=1e-6, atol=1e-3)
    assert np.allclose(stats.cpu().numpy()[:, 1], (zb ** 2).sum(1), rtol=1e-6)
    for eps in (1e-5, 1e-8):
        mean = zb.mean(1)[:, None, None]
        var = zb.var(1)[:, None, None]
        want = X.reshape(B, rows_per, 128) + ((Zw.reshape(B, rows_per, 128) - mean) / np.sqrt(var + eps)) * gamma + beta
        Xd = _t(X)
        st.call("dprnn_norm_residual_fwd", Xd.data_ptr(), Z.data_ptr(), stats.data_ptr(), _t(gamma).data_ptr(),
                _t(beta).data_ptr(), eps, Xd.data_ptr(), B, rows_per)
        torch.cuda.synchronize()
        assert rel_l2(Xd.cpu().numpy().reshape(B, rows_per, 128), want) < 1e-6


@pytest.mark.parametrize("seq_tile", [2016, 2032, 2064, 2096])
@pytest.mark.parametrize("ndir,ragged", [(2, False), (1, False), (2, True)])
def test_h_as_piece_quads_feeds_the_projection_bit_for_bit(seq_tile, ndir, ragged):
    """dprnn_rnn_fwd_hp_ex writes H as the f16 piece quads the mode-3 projection forms itself (recur_split8.h TRANS, rowgemm.hip PIECES):
    dprnn_proj_stats_hp_fwd on it returns EXACTLY the Z and the GroupNorm sums of dprnn_rnn_fwd_ex + dprnn_proj_stats_fwd, on both
    addressing forms, uni- and bidirectional, with a partial last tile; the piece quads decode to the fp32 h of the plain launch."""
    B, S, K = 3, 9, 21
    rng = np.random.default_rng(seq_tile + ndir)
    X = rng.standard_normal((B, S, K, 128)).astype(np.float32)
    sd, p = _lstm_sd(46)
    st = _st()
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device=DEV)
    bsum = torch.empty(2 * 512, device=DEV)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    w = [_t(sd[f"{p}.rnn.{n}"]) for n in names] + ([_t(sd[f"{p}.rnn.{n}_reverse"]) for n in names] if ndir == 2 else [None] * 4)
    ptr = lambda t: 0 if t is None else t.data_ptr()
    st.call("dprnn_rnn_pack_ex", 3, 0, *[ptr(t) for t in w], packed.data_ptr(), bsum.data_ptr())
    Xd = _t(X)
    kd = ndir * 128
    M = B * S * K
    W = _t((rng.standard_normal((128, kd)) / np.sqrt(kd)).astype(np.float32))
    b = _t(rng.standard_normal(128).astype(np.float32))
    addr = (B * K, S, K, S * K, 1, K)                      # inter-chunk addressing: sequences (b, k) walk s
    starts = _t(np.arange(B + 1, dtype=np.int64) * S * K)
    out = {}
    for hp in (False, True):
        H = torch.full((M, kd), float("nan"), device=DEV)
        Z = torch.full((M, 128), float("nan"), device=DEV)
        stats = torch.zeros(B, 2, device=DEV, dtype=torch.float64)
        st.call("dprnn_rnn_fwd_hp_ex" if hp else "dprnn_rnn_fwd_ex", 3, 0, Xd.data_ptr(), 0, 0, 0, 0, 0.0, 1, 1, 0, packed.data_ptr(),
                bsum.data_ptr(), H.data_ptr(), *addr, ndir, seq_tile)
        st.lib.dprnn_set_gemm_mode(3)
        try:
            if hp:
                st.call("dprnn_proj_stats_hp_fwd", H.data_ptr(), W.data_ptr(), b.data_ptr(), Z.data_ptr(), stats.data_ptr(), M, S * K,
                        starts.data_ptr() if ragged else 0, B if ragged else 0, kd)
            elif ragged:
                st.call("dprnn_proj_stats_ragged_fwd", H.data_ptr(), W.data_ptr(), b.data_ptr(), Z.data_ptr(), stats.data_ptr(), M,
                        starts.data_ptr(), B, kd)
            else:
                st.call("dprnn_proj_stats_fwd", H.data_ptr(), W.data_ptr(), b.data_ptr(), Z.data_ptr(), stats.data_ptr(), M, S * K, kd)
        finally:
            st.lib.dprnn_set_gemm_mode(0)
        torch.cuda.synchronize()
        out[hp] = (H.cpu().numpy(), Z.cpu().numpy(), stats.cpu().numpy())
    assert np.isfinite(out[False][1]).all()
    assert np.array_equal(out[True][1], out[False][1]), "Z differs"
    assert np.array_equal(out[True][2], out[False][2]), "GroupNorm sums differ"
    q = out[True][0].view(np.float16).reshape(M, kd // 4, 2, 4).astype(np.float64)      # [row][unit quad][piece][unit]
    h = ((q[:, :, 0] + q[:, :, 1]) / 16.0).reshape(M, kd)
    assert np.abs(h - out[False][0].astype(np.float64)).max() < 2.0 ** -22


@pytest.mark.parametrize("seq_tile", [2032, 2048, 2064])
@pytest.mark.parametrize("path,fused", [("intra", False), ("inter", False), ("intra", True), ("inter", True)])
@pytest.mark.parametrize("wscale,xscale", [(1.0, 1.0), (37.0, 0.02)])
def test_direction_serial_bilstm_with_the_linear_folded_in(seq_tile, path, fused, wscale, xscale):
    """dprnn_rnn_lin_fwd (recur_split8.h PJ): direction 0 leaves x' and the partial products W[:, :H] h_fwd IN THE BUFFER ITS PENDING Z
    CAME FROM, direction 1 turns them into z = Linear(cat(h_fwd, h_bwd)) in place and accumulates the GroupNorm sums -- against the
    numpy oracle's SingleRNN + Linear (dprnn.py:86-88 / 95-96), on both addressing forms, with partial last tiles and utterance
    boundaries inside tiles, with and without a pending norm, and with weights / activations far from unit scale."""
    B, S, K = 3, 37, 41
    rng = np.random.default_rng(seq_tile + int(10 * wscale) + (7 if fused else 0))
    X = (xscale * rng.standard_normal((B, S, K, 128))).astype(np.float32)
    sd, p = _lstm_sd(46)
    sd = {k: (v * np.float32(wscale) if ".rnn.weight_" in k else v) for k, v in sd.items()}
    W = (rng.standard_normal((128, 256)) / 16.0).astype(np.float32)
    b = rng.standard_normal(128).astype(np.float32)
    Xn = X
    if fused:
        Zr = (rng.standard_normal((B, S, K, 128)) * 1.3 + 0.2).astype(np.float32)
        gamma = (xscale * (1 + 0.1 * rng.standard_normal(128))).astype(np.float32)
        beta = (xscale * 0.1 * rng.standard_normal(128)).astype(np.float32)
        zb = Zr.reshape(B, -1).astype(np.float64)
        mean, var = zb.mean(1)[:, None, None, None], zb.var(1)[:, None, None, None]
        Xn = (X + ((Zr - mean) / np.sqrt(var + 1e-5)) * gamma + beta).astype(np.float32)
    if path == "intra":
        h = onp.single_rnn(sd, p, Xn.reshape(B * S, K, 128), "LSTM", True).reshape(B, S, K, 256)
    else:
        h = onp.single_rnn(sd, p, Xn.transpose(0, 2, 1, 3).reshape(B * K, S, 128), "LSTM", True)
        h = h.reshape(B, K, S, 256).transpose(0, 2, 1, 3)
    want = h.astype(np.float64) @ W.T.astype(np.float64) + b
    st = _st()
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device=DEV)
    bsum = torch.empty(2 * 512, device=DEV)
    linp = torch.empty(st.lib.dprnn_lin_pack_floats(), device=DEV)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    w = [_t(sd[f"{p}.rnn.{n}"]) for n in names] + [_t(sd[f"{p}.rnn.{n}_reverse"]) for n in names]
    st.call("dprnn_rnn_pack_ex", 3, 0, *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr())
    Wd, bd_ = _t(W), _t(b)
    st.call("dprnn_lin_pack_f16", Wd.data_ptr(), 2, linp.data_ptr())
    Xd = _t(X)
    Xo = torch.full((B, S, K, 128), float("nan"), device=DEV)
    addr = (B * S, K, B * S, 0, K, 1) if path == "intra" else (B * K, S, K, S * K, 1, K)
    spu = S if path == "intra" else K
    ZP = torch.full((B, S, K, 128), float("nan"), device=DEV)          # pending z -> partial products -> new z, one buffer
    norm = (0, 0, 0, 0, 0.0)
    if fused:
        ZP.copy_(torch.from_numpy(Zr))
        stats_in = _t(np.stack([zb.sum(1), (zb ** 2).sum(1)], axis=1))
        gd, bd = _t(gamma), _t(beta)
        norm = (ZP.data_ptr(), stats_in.data_ptr(), gd.data_ptr(), bd.data_ptr(), 1e-5)
    stats = torch.zeros(B, 2, device=DEV, dtype=torch.float64)
    st.call("dprnn_rnn_lin_fwd", 1, 0, Xd.data_ptr(), *norm, spu, S * K, Xo.data_ptr() if fused else 0, packed.data_ptr(), bsum.data_ptr(), 2,
            linp.data_ptr(), 0, ZP.data_ptr(), 0, 0, *addr, seq_tile)
    st.call("dprnn_rnn_lin_fwd", 2, 1, (Xo if fused else Xd).data_ptr(), 0, 0, 0, 0, 0.0, spu, S * K, 0, packed.data_ptr(), bsum.data_ptr(), 2,
            linp.data_ptr(), bd_.data_ptr(), ZP.data_ptr(), ZP.data_ptr(), stats.data_ptr(), *addr, seq_tile)
    torch.cuda.synchronize()
    got = ZP.cpu().numpy()
    assert np.isfinite(got).all()
    if fused:
        assert rel_l2(Xo.cpu().numpy(), Xn) < 1e-6
    # yardstick: the separate kernels of the same arithmetic (both directions in one launch, then the f16-pieces projection)
    H = torch.empty(B, S, K, 256, device=DEV)
    Z2 = torch.empty(B, S, K, 128, device=DEV)
    stats2 = torch.zeros(B, 2, device=DEV, dtype=torch.float64)
    st.call("dprnn_rnn_fwd_ex", 3, 0, _t(Xn).data_ptr(), 0, 0, 0, 0, 0.0, 1, 1, 0, packed.data_ptr(), bsum.data_ptr(), H.data_ptr(), *addr, 2, seq_tile)
    st.lib.dprnn_set_gemm_mode(3)
    st.call("dprnn_proj_stats_fwd", H.data_ptr(), Wd.data_ptr(), bd_.data_ptr(), Z2.data_ptr(), stats2.data_ptr(), B * S * K, S * K, 256)
    st.lib.dprnn_set_gemm_mode(0)
    torch.cuda.synchronize()
    e_new, e_old = rel_l2(got, want), rel_l2(Z2.cpu().numpy(), want)
    assert e_new < max(1e-6, 2 * e_old), f"tile {seq_tile} {path} fused={fused}: folded {e_new:.2e}, separate kernels {e_old:.2e}"
    # the sums are those of the rows actually written (fp32 partial sums of 128 elements, fp64 beyond) ...
    sg = stats.cpu().numpy()
    gb = got.astype(np.float64).reshape(B, -1)
    assert np.allclose(sg[:, 0], gb.sum(1), rtol=1e-9, atol=1e-6 * np.abs(gb).sum(1).max())
    assert np.allclose(sg[:, 1], (gb ** 2).sum(1), rtol=1e-6)
    # ... and the oracle's, as far as the rows are (x 37 weights saturate the cell: both kernel forms sit ~1e-4 from the oracle there)
    wb = want.reshape(B, -1)
    tol = max(1e-6, 4 * e_new)
    assert np.allclose(sg[:, 0], wb.sum(1), rtol=0, atol=tol * np.abs(wb).sum(1).max()), (e_new, e_old)
    assert np.allclose(sg[:, 1], (wb ** 2).sum(1), rtol=4 * tol), (e_new, e_old)
    print(f"tile {seq_tile} {path} fused={fused} w x{wscale}: folded {e_new:.2e}, separate {e_old:.2e}")


@pytest.mark.parametrize("seq_tile", [2032, 2048, 2064])
@pytest.mark.parametrize("path,fused,dims", [("intra", False, (3, 37, 41)), ("inter", False, (3, 37, 41)), ("intra", True, (3, 37, 41)),
                                             ("inter", True, (3, 37, 41)), ("intra", True, (2, 5, 1)), ("inter", True, (2, 2, 6)),
                                             ("inter", False, (1, 3, 4))])
def test_time_split_bilstm_with_the_linear_folded_in(seq_tile, path, fused, dims):
    """dprnn_rnn_lin_split_fwd (recur_split8.h: split_m): both directions in both launches, every sequence cut at M = T / 2 -- phase 1
    leaves x', the partial products of the rows each direction meets and the cell state; phase 2 finishes every row from the other
    side -- against the numpy oracle's SingleRNN + Linear (dprnn.py:86-88 / 95-96): odd and even T, T = 1, 2, 3 (a direction without
    a step in one phase), partial tiles, utterance boundaries inside tiles, with and without a pending norm."""
    B, S, K = dims
    rng = np.random.default_rng(seq_tile + B * 100 + S * 10 + K + (7 if fused else 0))
    X = rng.standard_normal((B, S, K, 128)).astype(np.float32)
    sd, p = _lstm_sd(46)
    W = (rng.standard_normal((128, 256)) / 16.0).astype(np.float32)
    b = rng.standard_normal(128).astype(np.float32)
    Xn = X
    if fused:
        Zr = (rng.standard_normal((B, S, K, 128)) * 1.3 + 0.2).astype(np.float32)
        gamma = (1 + 0.1 * rng.standard_normal(128)).astype(np.float32)
        beta = (0.1 * rng.standard_normal(128)).astype(np.float32)
        zb = Zr.reshape(B, -1).astype(np.float64)
        mean, var = zb.mean(1)[:, None, None, None], zb.var(1)[:, None, None, None]
        Xn = (X + ((Zr - mean) / np.sqrt(var + 1e-5)) * gamma + beta).astype(np.float32)
    if path == "intra":
        h = onp.single_rnn(sd, p, Xn.reshape(B * S, K, 128), "LSTM", True).reshape(B, S, K, 256)
    else:
        h = onp.single_rnn(sd, p, Xn.transpose(0, 2, 1, 3).reshape(B * K, S, 128), "LSTM", True)
        h = h.reshape(B, K, S, 256).transpose(0, 2, 1, 3)
    want = h.astype(np.float64) @ W.T.astype(np.float64) + b
    st = _st()
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device=DEV)
    bsum = torch.empty(2 * 512, device=DEV)
    linp = torch.empty(st.lib.dprnn_lin_pack_floats(), device=DEV)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    w = [_t(sd[f"{p}.rnn.{n}"]) for n in names] + [_t(sd[f"{p}.rnn.{n}_reverse"]) for n in names]
    st.call("dprnn_rnn_pack_ex", 3, 0, *[t.data_ptr() for t in w], packed.data_ptr(), bsum.data_ptr())
    Wd, bd_ = _t(W), _t(b)
    st.call("dprnn_lin_pack_f16", Wd.data_ptr(), 2, linp.data_ptr())
    Xd = _t(X)
    Xo = torch.full((B, S, K, 128), float("nan"), device=DEV)
    addr = (B * S, K, B * S, 0, K, 1) if path == "intra" else (B * K, S, K, S * K, 1, K)
    spu = S if path == "intra" else K
    state = torch.full((st.lib.dprnn_rnn_lin_state_floats(addr[0], seq_tile),), float("nan"), device=DEV)
    P = torch.full((B, S, K, 128), float("nan"), device=DEV)
    norm = (0, 0, 0, 0, 0.0)
    if fused:
        stats_in = _t(np.stack([zb.sum(1), (zb ** 2).sum(1)], axis=1))
        Zd, gd, bd = _t(Zr), _t(gamma), _t(beta)
        norm = (Zd.data_ptr(), stats_in.data_ptr(), gd.data_ptr(), bd.data_ptr(), 1e-5)
    stats = torch.zeros(B, 2, device=DEV, dtype=torch.float64)
    tail = (packed.data_ptr(), bsum.data_ptr(), linp.data_ptr())
    st.call("dprnn_rnn_lin_split_fwd", 1, Xd.data_ptr(), *norm, spu, S * K, Xo.data_ptr() if fused else 0, *tail, 0, P.data_ptr(), 0,
            state.data_ptr(), *addr, seq_tile)
    st.call("dprnn_rnn_lin_split_fwd", 2, (Xo if fused else Xd).data_ptr(), 0, 0, 0, 0, 0.0, spu, S * K, 0, *tail, bd_.data_ptr(), P.data_ptr(),
            stats.data_ptr(), state.data_ptr(), *addr, seq_tile)
    torch.cuda.synchronize()
    got = P.cpu().numpy()
    assert np.isfinite(got).all()
    if fused:
        assert rel_l2(Xo.cpu().numpy(), Xn) < 1e-6
    H = torch.empty(B, S, K, 256, device=DEV)
    Z2 = torch.empty(B, S, K, 128, device=DEV)
    stats2 = torch.zeros(B, 2, device=DEV, dtype=torch.float64)
    st.call("dprnn_rnn_fwd_ex", 3, 0, _t(Xn).data_ptr(), 0, 0, 0, 0, 0.0, 1, 1, 0, packed.data_ptr(), bsum.data_ptr(), H.data_ptr(), *addr, 2, seq_tile)
    st.lib.dprnn_set_gemm_mode(3)
    st.call("dprnn_proj_stats_fwd", H.data_ptr(), Wd.data_ptr(), bd_.data_ptr(), Z2.data_ptr(), stats2.data_ptr(), B * S * K, S * K, 256)
    st.lib.dprnn_set_gemm_mode(0)
    torch.cuda.synchronize()
    e_new, e_old = rel_l2(got, want), rel_l2(Z2.cpu().numpy(), want)
    assert e_new < max(1e-6, 2 * e_old), f"tile {seq_tile} {path} fused={fused} {dims}: folded {e_new:.2e}, separate kernels {e_old:.2e}"
    sg = stats.cpu().numpy()
    gb = got.astype(np.float64).reshape(B, -1)
    assert np.allclose(sg[:, 0], gb.sum(1), rtol=1e-9, atol=1e-6 * np.abs(gb).sum(1).max())
    assert np.allclose(sg[:, 1], (gb ** 2).sum(1), rtol=1e-6)
    wb = want.reshape(B, -1)
    assert np.allclose(sg[:, 0], wb.sum(1), rtol=0, atol=4e-6 * np.abs(wb).sum(1).max())
    assert np.allclose(sg[:, 1], (wb ** 2).sum(1), rtol=2e-5)


@pytest.mark.parametrize("path", ["intra", "inter"])
def test_folded_linear_of_a_unidirectional_lstm(path):
    """dprnn_rnn_lin_fwd phase 2 alone (P = NULL): a unidirectional LSTM + its Linear 128 -> 128 in one launch (the reference's inter-chunk RNN
    may be unidirectional, dprnn.py:68), against the numpy oracle."""
    B, S, K = 2, 19, 23
    rng = np.random.default_rng(91)
    X = rng.standard_normal((B, S, K, 128)).astype(np.float32)
    sd, p = _lstm_sd(47, bidir=False)
    W = (rng.standard_normal((128, 128)) / 11.0).astype(np.float32)
    b = rng.standard_normal(128).astype(np.float32)
    if path == "intra":
        h = onp.single_rnn(sd, p, X.reshape(B * S, K, 128), "LSTM", False).reshape(B, S, K, 128)
    else:
        h = onp.single_rnn(sd, p, X.transpose(0, 2, 1, 3).reshape(B * K, S, 128), "LSTM", False).reshape(B, K, S, 128).transpose(0, 2, 1, 3)
    want = h.astype(np.float64) @ W.T.astype(np.float64) + b
    st = _st()
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device=DEV)
    bsum = torch.empty(2 * 512, device=DEV)
    linp = torch.empty(st.lib.dprnn_lin_pack_floats(), device=DEV)
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    w = [_t(sd[f"{p}.rnn.{n}"]) for n in names]
    st.call("dprnn_rnn_pack_ex", 3, 0, *[t.data_ptr() for t in w], 0, 0, 0, 0, packed.data_ptr(), bsum.data_ptr())
    st.call("dprnn_lin_pack_f16", _t(W).data_ptr(), 1, linp.data_ptr())
    addr = (B * S, K, B * S, 0, K, 1) if path == "intra" else (B * K, S, K, S * K, 1, K)
    Z = torch.full((B, S, K, 128), float("nan"), device=DEV)
    stats = torch.zeros(B, 2, device=DEV, dtype=torch.float64)
    st.call("dprnn_rnn_lin_fwd", 2, 0, _t(X).data_ptr(), 0, 0, 0, 0, 0.0, S if path == "intra" else K, S * K, 0, packed.data_ptr(), bsum.data_ptr(), 1,
            linp.data_ptr(), _t(b).data_ptr(), 0, Z.data_ptr(), stats.data_ptr(), *addr, 2064)
    torch.cuda.synchronize()
    got = Z.cpu().numpy()
    assert np.isfinite(got).all() and rel_l2(got, want) < 1e-6, rel_l2(got, want)
    gb = got.astype(np.float64).reshape(B, -1)
    assert np.allclose(stats.cpu().numpy()[:, 1], (gb ** 2).sum(1), rtol=1e-6)


def test_direction_serial_form_refuses_what_it_does_not_do():
    st = _st()
    z = torch.zeros(64, device=DEV)
    args = lambda phase, dir_, Z=0, P=z.data_ptr(), Zout=z.data_ptr(), so=z.data_ptr(), lb=z.data_ptr(), tile=0: (
        phase, dir_, z.data_ptr(), Z, z.data_ptr(), z.data_ptr(), z.data_ptr(), 1e-5, 1, 1, z.data_ptr() + 4, z.data_ptr(), z.data_ptr(), 2,
        z.data_ptr(), lb, P, Zout, so, 1, 1, 1, 1, 1, 1, tile)
    for bad, msg in ((args(3, 0), "phase"), (args(1, 2), "direction"), (args(1, 0, P=0), "partial"), (args(2, 1, so=0), "stats_out"),
                     (args(2, 1, Z=z.data_ptr()), "pending norm"), (args(1, 0, tile=2016), "seq_tile")):
        with pytest.raises(RuntimeError, match=msg):
            st.call("dprnn_rnn_lin_fwd", *bad)


@pytest.mark.parametrize("mode", [0, 3])
@pytest.mark.parametrize("kdim", [256, 128])
def test_proj_stats_ragged(kdim, mode):
    """dprnn_proj_stats_ragged_fwd: utterance boundaries from a table of first rows (ragged batches), boundaries inside row
    tiles and a last partial tile."""
    rng = np.random.default_rng(kdim + 7)
    rows = [77, 130, 45]
    starts = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    M = int(starts[-1])
    H = rng.standard_normal((M, kdim)).astype(np.float32)
    W = (rng.standard_normal((128, kdim)) / np.sqrt(kdim)).astype(np.float32)
    b = rng.standard_normal(128).astype(np.float32)
    Zw = H.astype(np.float64) @ W.T.astype(np.float64) + b
    st = _st()
    Z = torch.full((M, 128), float("nan"), device=DEV)
    stats = torch.zeros(len(rows), 2, device=DEV, dtype=torch.float64)
    st.lib.dprnn_set_gemm_mode(mode)
    st.call("dprnn_proj_stats_ragged_fwd", _t(H).data_ptr(), _t(W).data_ptr(), _t(b).data_ptr(), Z.data_ptr(),
            stats.data_ptr(), M, _t(starts).data_ptr(), len(rows), kdim)
    torch.cuda.synchronize()
    assert rel_l2(Z.cpu().numpy(), Zw) < 1e-6
    got = stats.cpu().numpy()
    for u in range(len(rows)):
        zu = Zw[starts[u]: starts[u + 1]]
        assert abs(got[u, 0] - zu.sum()) <= 1e-5 * np.abs(zu).sum()
        assert abs(got[u, 1] - (zu ** 2).sum()) <= 1e-6 * (zu ** 2).sum()


@pytest.mark.parametrize("K,P,L,B", [(10, 5, 101, 2), (16, 5, 77, 2), (8, 8, 40, 2), (250, 125, 700, 2), (7, 3, 5, 3), (3, 2, 50, 1),
                                     (10, 5, 2000, 11)])       # the last one: more chunks than workgroups (grid-stride loop)
def test_segment_matches_unfold(K, P, L, B):
    Y = np.random.default_rng(K).standard_normal((B, L, 128)).astype(np.float32)
    S = onp.n_chunks(L, K, P)
    want = onp.segmentation(Y.transpose(0, 2, 1), K, P).transpose(0, 3, 2, 1)      # [B,S,K,F]
    X = torch.full((B, S, K, 128), float("nan"), device=DEV)
    _st().call("dprnn_segment_fwd", _t(Y).data_ptr(), X.data_ptr(), B, L, K, P, S)
    torch.cuda.synchronize()
    assert np.array_equal(X.cpu().numpy(), want)


@pytest.mark.parametrize("mode", [0, 3])        # 3: the f16-pieces row GEMM takes over the mask head (rowgemm_f16t_kernel)
@pytest.mark.parametrize("K,P,L", [(10, 5, 101), (16, 5, 77), (250, 125, 700)])
def test_head_conv_ola_matches_oracle(K, P, L, mode):
    B = 2
    _st().lib.dprnn_set_gemm_mode(mode)
    rng = np.random.default_rng(L)
    S = onp.n_chunks(L, K, P)
    X = rng.standard_normal((B, S, K, 128)).astype(np.float32)
    alpha = np.array([0.2], dtype=np.float32)
    W = (rng.standard_normal((256, 128)) / 11).astype(np.float32)
    b = rng.standard_normal(256).astype(np.float32)
    xr = X.transpose(0, 3, 2, 1)                                                    # [B,F,K,S]
    h = onp.conv1x1(W, b, onp.prelu(alpha, xr), np.float64).reshape(B * 2, 128, K, S)
    want = onp.overlap_add(h, L, K, P).reshape(B, 256, L).transpose(0, 2, 1)        # [B,L,256]
    U = torch.empty(B, L, 256, device=DEV)
    _st().call("dprnn_head_conv_ola_fwd", _t(X).data_ptr(), _t(alpha).data_ptr(), _t(W).data_ptr(), _t(b).data_ptr(),
               U.data_ptr(), B, L, K, P, S)
    torch.cuda.synchronize()
    assert rel_l2(U.cpu().numpy(), want) < 2e-6


@pytest.mark.parametrize("mode", [0, 3])
@pytest.mark.parametrize("ks,stride,T,relu", [(2, 1, 300, 0), (4, 2, 203, 1), (2, 1, 65, 1), (8, 4, 523, 0)])
def test_encoder_bottleneck_and_decode_chain(ks, stride, T, relu, mode):
    """encoder stats -> bottleneck and mask/decode chain vs oracle with random V (stage inputs)."""
    B, N = 3, 64
    _st().lib.dprnn_set_gemm_mode(mode)
    rng = np.random.default_rng(T)
    cfg = full_cfg(kernel_size=ks, stride=stride, n_repeats=1)
    wav = (0.1 * rng.standard_normal((B, T))).astype(np.float32)
    We = (rng.standard_normal((N, 1, ks)) / np.sqrt(ks)).astype(np.float32)
    Wd = (rng.standard_normal((N, 1, ks)) / 8).astype(np.float32)
    sd = {"encoder.conv1d.weight": We, "decoder.weight": Wd}
    enc = onp.encoder(sd, cfg, wav, np.float64)                                    # [B,N,L]
    L = enc.shape[-1]
    st = _st()
    stats = torch.zeros(B, 2, device=DEV, dtype=torch.float64)
    st.call("dprnn_encoder_stats", _t(wav).data_ptr(), _t(We).data_ptr(), stats.data_ptr(), B, T, L, N, ks, stride)
    torch.cuda.synchronize()
    s = stats.cpu().numpy()
    assert np.allclose(s[:, 0], enc.reshape(B, -1).sum(1), rtol=1e-5)
    assert np.allclose(s[:, 1], (enc.reshape(B, -1) ** 2).sum(1), rtol=1e-5)
    # bottleneck (ln) with a per-utterance row bias
    gamma = (1 + 0.1 * rng.standard_normal(N)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(N)).astype(np.float32)
    Wb = (rng.standard_normal((128, N)) / 8).astype(np.float32)
    bb = rng.standard_normal(128).astype(np.float32)
    rb = rng.standard_normal((B, 128)).astype(np.float32)
    sdn = {"n.weight": gamma, "n.bias": beta}
    want = onp.conv1x1(Wb, bb, onp.channel_norm(sdn, "n", enc, "ln", np.float64), np.float64) + rb[:, :, None]
    sc = torch.empty(B, N, device=DEV); sh = torch.empty(B, N, device=DEV)
    st.call("dprnn_bottleneck_prep", stats.data_ptr(), _t(gamma).data_ptr(), _t(beta).data_ptr(), 1e-5, 0, 0,
            sc.data_ptr(), sh.data_ptr(), B, L, N)
    Y = torch.empty(B, L, 128, device=DEV)
    st.call("dprnn_bottleneck_fwd", _t(wav).data_ptr(), _t(We).data_ptr(), sc.data_ptr(), sh.data_ptr(), _t(Wb).data_ptr(),
            N, _t(bb).data_ptr(), _t(rb).data_ptr(), Y.data_ptr(), B, T, L, N, ks, stride)
    torch.cuda.synchronize()
    assert rel_l2(Y.cpu().numpy(), want.transpose(0, 2, 1)) < 2e-6
    # gate -> end conv -> act -> mask*enc -> decoder
    U = rng.standard_normal((B, L, 2, 128)).astype(np.float32)
    Wo = (rng.standard_normal((128, 128)) / 11).astype(np.float32); bo = rng.standard_normal(128).astype(np.float32)
    Wg = (rng.standard_normal((128, 128)) / 11).astype(np.float32); bg = rng.standard_normal(128).astype(np.float32)
    Wend = (rng.standard_normal((N, 128)) / 11).astype(np.float32)
    u = U.astype(np.float64)
    v = np.tanh(u @ Wo.T + bo) * (1 / (1 + np.exp(-(u @ Wg.T + bg))))
    mk = v @ Wend.T
    mk = np.maximum(mk, 0) if relu else 1 / (1 + np.exp(-mk))                      # [B,L,2,N]
    me = mk * enc.transpose(0, 2, 1)[:, :, None, :]
    want_out = np.stack([onp.decoder(sd, cfg, me[:, :, i].transpose(0, 2, 1), np.float64) for i in range(2)], 1)
    V = torch.empty(B * L * 2, 128, device=DEV)
    st.call("dprnn_head_gate_fwd", _t(U).data_ptr(), _t(Wo).data_ptr(), _t(bo).data_ptr(), _t(Wg).data_ptr(),
            _t(bg).data_ptr(), V.data_ptr(), B * L * 2)
    taps = torch.empty(B, 2, L, ks, device=DEV)
    masked = torch.empty(B, L, N, device=DEV)
    st.call("dprnn_head_mask_decode_fwd", V.data_ptr(), _t(Wend).data_ptr(), _t(wav).data_ptr(), _t(We).data_ptr(),
            _t(Wd).data_ptr(), taps.data_ptr(), masked.data_ptr(), relu, B, T, L, N, ks, stride)
    T_out = (L - 1) * stride + ks
    out = torch.empty(B, 2, T_out, device=DEV)
    st.call("dprnn_decoder_fwd", taps.data_ptr(), out.data_ptr(), B, 2, T_out, L, ks, stride)
    torch.cuda.synchronize()
    assert rel_l2(V.cpu().numpy().reshape(B, L, 2, 128), v) < 2e-6
    assert rel_l2(masked.cpu().numpy(), me[:, :, 0]) < 2e-6
    assert rel_l2(out.cpu().numpy(), want_out) < 2e-6
