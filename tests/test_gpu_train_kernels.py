"""Stage-level GPU tests of the training (backward) kernels against torch CPU autograd of the same stock operators
the reference dispatches to (nn.LSTM, F.linear): test infrastructure only, the product never calls them."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle.weights import make_state

import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import engine

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
_KEEP = []


def _st():
    return engine.Stage(torch.device(DEV))


def _t(a):
    t = (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).contiguous().to(DEV)
    _KEEP.append(t)
    if len(_KEEP) > 256:
        torch.cuda.synchronize()
        del _KEEP[:128]
    return t


def _p(t):
    return 0 if t is None else t.data_ptr()


def _cpu_lstm(seed):
    sd = make_state("bss", dict(n_repeats=1), seed)
    p = "separation.dprnn_blocks.0.intra_rnn.rnn."
    lstm = torch.nn.LSTM(128, 128, 1, batch_first=True, bidirectional=True)
    with torch.no_grad():
        for n, prm in lstm.named_parameters():
            prm.copy_(torch.from_numpy(sd[p + n]))
    return lstm


NAMES = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]


@pytest.mark.parametrize("path,B,S,K", [("intra", 3, 9, 23), ("inter", 2, 11, 37), ("intra", 1, 40, 5)])
def test_lstm_backward_matches_autograd(path, B, S, K):
    """dprnn_lstm_fwd_train + dprnn_lstm_bwd + dprnn_wgrad vs autograd of nn.LSTM (dprnn.py:23-37 under
    trainer.py:106): dX, dW_ih, dW_hh, db of both directions; ragged against the 32-sequence tile."""
    rng = np.random.default_rng(B * 100 + S)
    lstm = _cpu_lstm(7)
    X = torch.from_numpy(rng.standard_normal((B, S, K, 128)).astype(np.float32))
    dH = torch.from_numpy(rng.standard_normal((B, S, K, 256)).astype(np.float32))
    Xr = X.clone().requires_grad_(True)
    if path == "intra":
        out = lstm(Xr.reshape(B * S, K, 128))[0].reshape(B, S, K, 256)
    else:
        out = lstm(Xr.permute(0, 2, 1, 3).reshape(B * K, S, 128))[0].reshape(B, K, S, 256).permute(0, 2, 1, 3)
    out.backward(dH)

    st = _st()
    lib = st.lib
    w = [_t(getattr(lstm, n).detach()) for n in NAMES] + [_t(getattr(lstm, n + "_reverse").detach()) for n in NAMES]
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device=DEV)
    bsum = torch.empty(2 * 512, device=DEV)
    packedT = torch.empty(2 * 512 * 256, device=DEV)
    rows = B * S * K
    addr = (B * S, K, B * S, 0, K, 1) if path == "intra" else (B * K, S, K, S * K, 1, K)
    nseq, T = addr[0], addr[1]
    tape = torch.empty(lib.dprnn_lstm_tape_floats(nseq, T, 2), device=DEV)
    Xd, dHd = _t(X), _t(dH)
    H = torch.full((rows, 256), float("nan"), device=DEV)
    dXd = torch.full((2, rows, 128), float("nan"), device=DEV)
    dG = torch.full((rows, 1024), float("nan"), device=DEV)
    st.call("dprnn_rnn_pack", 0, *[_p(t) for t in w], _p(packed), _p(bsum))
    st.call("dprnn_lstm_pack_bwd", _p(w[0]), _p(w[1]), _p(w[4]), _p(w[5]), _p(packedT))
    st.call("dprnn_lstm_fwd_train", _p(Xd), 0, 0, 0, 0, 0.0, 1, 1, 0, _p(packed), _p(bsum), _p(H), _p(tape), *addr, 2)
    st.call("dprnn_lstm_bwd", _p(dHd), _p(tape), _p(packedT), _p(dXd), _p(dG), rows, *addr, 2)
    torch.cuda.synchronize()
    assert rel_l2(H.cpu().numpy().reshape(B, S, K, 256), out.detach().numpy()) < 2e-6
    dX = (dXd[0] + dXd[1]).cpu().numpy().reshape(B, S, K, 128)
    assert np.isfinite(dX).all()
    err = rel_l2(dX, Xr.grad.numpy())
    print(f"lstm bwd {path}: dX rel-L2 {err:.2e}")
    assert err < 5e-6

    # weight gradients: dW = dG^T [x | h_prev]
    ws = torch.empty(lib.dprnn_wgrad_workspace_floats(512, 256, rows), device=DEV)
    t_stride = addr[5]
    for d, suf in enumerate(["", "_reverse"]):
        dW = torch.full((512, 256), float("nan"), device=DEV)
        db0 = torch.full((512,), float("nan"), device=DEV)
        db1 = torch.full((512,), float("nan"), device=DEV)
        st.call("dprnn_wgrad", dG.data_ptr() + 4 * 512 * d, 1024, 512, _p(Xd), 128, 128, 0,
                H.data_ptr() + 4 * 128 * d, 256, 128, 1 if d == 0 else -1, rows, t_stride, T,
                _p(dW), 256, 1, _p(db0), _p(db1), 0, _p(ws))
        torch.cuda.synchronize()
        g_ih = getattr(lstm, "weight_ih_l0" + suf).grad.numpy()
        g_hh = getattr(lstm, "weight_hh_l0" + suf).grad.numpy()
        g_b = getattr(lstm, "bias_ih_l0" + suf).grad.numpy()
        dWn = dW.cpu().numpy()
        e1, e2 = rel_l2(dWn[:, :128], g_ih), rel_l2(dWn[:, 128:], g_hh)
        e3, e4 = rel_l2(db0.cpu().numpy(), g_b), rel_l2(db1.cpu().numpy(), g_b)
        print(f"  dir {d}: dW_ih {e1:.2e} dW_hh {e2:.2e} db {e3:.2e}")
        assert max(e1, e2, e3, e4) < 1e-5


def test_lstm_backward_f16_pieces_flags_gradients_beyond_their_scale():
    """The f16-pieces BPTT multiplies da by 2^(6 - e), e = exponent of max|dH| (handed over by dH's producer); a gate gradient more
    than 2^10 above that does not fit f16 and is clamped.  The gate gradients go to dG as the very pieces the BPTT multiplied with
    (quad layout), and the weight-gradient GEMM that reads them back sees a clamped pair and raises flag 4.  With the true maximum
    nothing is flagged and dX / dW / db match autograd; with a maximum 2^13 too small (what an exploding recurrence amounts to) the
    flag is up -- training.backward then repeats the step in the exact kernels.  Also: the backward pack carries its format like
    the forward's (DPRNN_EFORMAT on a mismatch)."""
    import ctypes
    B, S, K = 2, 9, 23
    rng = np.random.default_rng(11)
    lstm = _cpu_lstm(7)
    X = torch.from_numpy(rng.standard_normal((B, S, K, 128)).astype(np.float32))
    dH = torch.from_numpy(rng.standard_normal((B, S, K, 256)).astype(np.float32))
    Xr = X.clone().requires_grad_(True)
    out = lstm(Xr.reshape(B * S, K, 128))[0].reshape(B, S, K, 256)
    out.backward(dH)
    st = _st()
    lib = st.lib
    w = [_t(getattr(lstm, n).detach()) for n in NAMES] + [_t(getattr(lstm, n + "_reverse").detach()) for n in NAMES]
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device=DEV)
    bsum = torch.empty(2 * 512, device=DEV)
    packedT = torch.empty(lib.dprnn_rnn_pack_bwd_floats(2), device=DEV)
    rows = B * S * K
    addr = (B * S, K, B * S, 0, K, 1)
    tape = torch.empty(lib.dprnn_lstm_tape_floats(addr[0], addr[1], 2), device=DEV)
    Xd, dHd = _t(X), _t(dH)
    H = torch.empty(rows, 256, device=DEV)
    dXd = torch.empty(2, rows, 128, device=DEV)
    dG = torch.empty(lib.dprnn_rnn_bwd_dg_floats(0, 3, addr[0], addr[1], 2), device=DEV)       # quad layout (27 sequences -> 7 quads)
    st.call("dprnn_rnn_pack_ex", 0, 0, *[_p(t) for t in w], _p(packed), _p(bsum))
    st.call("dprnn_rnn_fwd_train_ex", 0, 0, _p(Xd), 0, 0, 0, 0, 0.0, 1, 1, 0, _p(packed), _p(bsum), _p(H), _p(tape), *addr, 2)
    ws = torch.empty(lib.dprnn_wgrad_workspace_floats(512, 256, K * 32 * (-(-addr[0] // 32))), device=DEV)
    dW = torch.empty(512, 256, device=DEV)
    dWr = torch.empty(512, 256, device=DEV)
    db = torch.empty(512, device=DEV)
    nsq = -(-addr[0] // 4)

    def run(hmax_value):
        hmax = torch.full((1,), hmax_value, device=DEV)
        flags = torch.empty(8, dtype=torch.int32, device=DEV)
        host = (ctypes.c_int32 * 8)()
        assert lib.dprnn_saturation_begin(flags.data_ptr(), st.stream) == 0
        try:
            st.call("dprnn_rnn_pack_bwd_ex", 3, 0, _p(w[0]), _p(w[1]), _p(w[4]), _p(w[5]), _p(packedT))
            st.call("dprnn_rnn_bwd_ex", 3, _p(hmax), 0, _p(dHd), _p(tape), _p(packedT), _p(dXd), _p(dG), rows, *addr, 2)
            gm = packedT.data_ptr() + 4 * lib.dprnn_rnn_bwd_gradmax_offset(2)
            st.call("dprnn_wgrad_lstm_f16", gm, dG.data_ptr(), _p(Xd), H.data_ptr(), 256, 0, *addr, _p(dW), _p(db), 0, _p(ws))
            st.call("dprnn_wgrad_lstm_f16", gm, dG.data_ptr() + 4 * K * nsq * 2048, _p(Xd), H.data_ptr() + 4 * 128, 256, 1, *addr,
                    _p(dWr), 0, 0, _p(ws))
            rc = lib.dprnn_saturation_flags(ctypes.addressof(host), st.stream)
        finally:
            lib.dprnn_saturation_begin(0, 0)
        return rc, list(host)
    true_max = float(dH.abs().max())
    rc, flags = run(true_max)
    assert rc == 0 and not any(flags), flags
    dX = (dXd[0] + dXd[1]).cpu().numpy().reshape(B, S, K, 128)
    assert rel_l2(dX, Xr.grad.numpy()) < 5e-6
    g = np.concatenate([lstm.weight_ih_l0.grad.numpy(), lstm.weight_hh_l0.grad.numpy()], axis=1)
    gr = np.concatenate([lstm.weight_ih_l0_reverse.grad.numpy(), lstm.weight_hh_l0_reverse.grad.numpy()], axis=1)
    assert rel_l2(dW.cpu().numpy(), g) < 1e-5 and rel_l2(dWr.cpu().numpy(), gr) < 1e-5
    assert rel_l2(db.cpu().numpy(), lstm.bias_ih_l0.grad.numpy()) < 1e-5
    rc, flags = run(true_max / 8192.0)
    assert rc == 1 and flags[4] == 1 and not any(flags[:4]), flags
    # pack / launch mismatch in either direction, and a forward pack handed to the backward
    st.call("dprnn_rnn_pack_bwd_ex", 0, 0, _p(w[0]), _p(w[1]), _p(w[4]), _p(w[5]), _p(packedT))
    assert lib.dprnn_rnn_bwd_ex(3, 0, 0, _p(dHd), _p(tape), _p(packedT), _p(dXd), _p(dG), rows, *addr, 2, st.stream) == -3
    assert b"pack and launch under the same mode" in lib.dprnn_last_error()
    st.call("dprnn_rnn_pack_bwd_ex", 3, 0, _p(w[0]), _p(w[1]), _p(w[4]), _p(w[5]), _p(packedT))
    assert lib.dprnn_rnn_bwd_ex(0, 0, 0, _p(dHd), _p(tape), _p(packedT), _p(dXd), _p(dG), rows, *addr, 2, st.stream) == -3
    assert lib.dprnn_rnn_bwd_ex(0, 0, 0, _p(dHd), _p(tape), _p(packed), _p(dXd), _p(dG), rows, *addr, 2, st.stream) == -3
    torch.cuda.synchronize()


@pytest.mark.parametrize("path,B,S,K", [("intra", 3, 9, 23), ("inter", 2, 11, 70)])
def test_lstm_backward_packed_gate_tape(path, B, S, K):
    """The f16-pieces training forward can leave the gates as 24-bit fixed point (gemm_mode 3 | 0x100: three words per (row, unit)
    instead of four, csrc/common.h tape_pack) for the f16-pieces BPTT launched with the same flag: dX and the gate gradients dG
    are those of the fp32 tape to ~1e-7 (a gate is held to 2^-25 / 2^-24 absolute) and dX matches autograd of nn.LSTM like the
    fp32 tape does.  The flag is refused outside that pair of kernels."""
    rng = np.random.default_rng(B * 100 + S)
    lstm = _cpu_lstm(7)
    X = torch.from_numpy(rng.standard_normal((B, S, K, 128)).astype(np.float32))
    dH = torch.from_numpy(rng.standard_normal((B, S, K, 256)).astype(np.float32))
    Xr = X.clone().requires_grad_(True)
    if path == "intra":
        out = lstm(Xr.reshape(B * S, K, 128))[0].reshape(B, S, K, 256)
    else:
        out = lstm(Xr.permute(0, 2, 1, 3).reshape(B * K, S, 128))[0].reshape(B, K, S, 256).permute(0, 2, 1, 3)
    out.backward(dH)
    st = _st()
    lib = st.lib
    w = [_t(getattr(lstm, n).detach()) for n in NAMES] + [_t(getattr(lstm, n + "_reverse").detach()) for n in NAMES]
    packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device=DEV)
    bsum = torch.empty(2 * 512, device=DEV)
    packedT = torch.empty(lib.dprnn_rnn_pack_bwd_floats(2), device=DEV)
    rows = B * S * K
    addr = (B * S, K, B * S, 0, K, 1) if path == "intra" else (B * K, S, K, S * K, 1, K)
    Xd, dHd = _t(X), _t(dH)
    hmax = torch.full((1,), float(dH.abs().max()), device=DEV)
    st.call("dprnn_rnn_pack_ex", 3, 0, *[_p(t) for t in w], _p(packed), _p(bsum))
    st.call("dprnn_rnn_pack_bwd_ex", 3, 0, _p(w[0]), _p(w[1]), _p(w[4]), _p(w[5]), _p(packedT))
    res = {}
    for flag in (0, 0x100):
        tape = torch.full((lib.dprnn_lstm_tape_floats(addr[0], addr[1], 2),), float("nan"), device=DEV)
        H = torch.full((rows, 256), float("nan"), device=DEV)
        dXd = torch.full((2, rows, 128), float("nan"), device=DEV)
        dG = torch.zeros(lib.dprnn_rnn_bwd_dg_floats(0, 3, addr[0], addr[1], 2), device=DEV)
        st.call("dprnn_rnn_fwd_train_ex", 3 | flag, 0, _p(Xd), 0, 0, 0, 0, 0.0, 1, 1, 0, _p(packed), _p(bsum), _p(H), _p(tape), *addr, 2)
        st.call("dprnn_rnn_bwd_ex", 3 | flag, _p(hmax), 0, _p(dHd), _p(tape), _p(packedT), _p(dXd), _p(dG), rows, *addr, 2)
        torch.cuda.synchronize()
        res[flag] = (H.cpu().numpy(), (dXd[0] + dXd[1]).cpu().numpy(), dG.view(torch.int32).cpu().numpy(), tape)
    assert np.array_equal(res[0][0], res[0x100][0])                      # the forward itself is untouched
    ref = Xr.grad.numpy().reshape(rows, 128)
    e_plain, e_packed = rel_l2(res[0][1], ref), rel_l2(res[0x100][1], ref)
    e_pair = rel_l2(res[0x100][1], res[0][1])
    print(f"packed tape {path}: dX vs autograd: fp32 tape {e_plain:.2e}, packed {e_packed:.2e}; packed vs fp32 tape {e_pair:.2e}")
    assert np.isfinite(res[0x100][1]).all() and e_packed < 5e-6 and e_pair < 1e-6
    # the gate gradients as the f16 piece pairs the weight-gradient GEMM reads: first pieces differ in their last bits at most
    a, b = res[0][2], res[0x100][2]
    first = lambda v: np.ascontiguousarray(v.reshape(-1, 4)[:, :2]).view(np.float16).astype(np.float32)    # words {p1, p1, p2, p2} per quad
    assert float(np.abs(first(a) - first(b)).max()) <= 2.0 ** -9 * max(1.0, float(np.abs(first(a)).max()))
    # slot 3 of the packed tape is never written (the NaN fill survives)
    assert bool(torch.isnan(res[0x100][3].view(-1, 5, 256)[:, 3]).all())
    assert not bool(torch.isnan(res[0][3].view(-1, 5, 256)[:, 3]).all())
    # the flag outside its pair of kernels
    tape = res[0][3]
    assert lib.dprnn_rnn_fwd_train_ex(0x100, 0, _p(Xd), 0, 0, 0, 0, 0.0, 1, 1, 0, _p(packed), _p(bsum), _p(H), _p(tape), *addr, 2, st.stream) == -1
    assert b"packed gate tape" in lib.dprnn_last_error()
    assert lib.dprnn_rnn_bwd_ex(0x100, 0, 0, _p(dHd), _p(tape), _p(packedT), _p(dXd), _p(dG), rows, *addr, 2, st.stream) == -1
    assert b"packed gate tape" in lib.dprnn_last_error()
    torch.cuda.synchronize()


@pytest.mark.parametrize("rows", [1256, 198, 22, 1])
@pytest.mark.parametrize("kd,nd", [(128, 256), (128, 128), (64, 128), (256, 128), (128, 64)])
def test_dgrad_matches_matmul(kd, nd, rows):
    rng = np.random.default_rng(kd + nd)
    dOut = rng.standard_normal((rows, kd)).astype(np.float32)
    k0 = kd if kd != 256 else 128
    W0 = rng.standard_normal((k0, nd)).astype(np.float32)
    W1 = rng.standard_normal((kd - k0, nd)).astype(np.float32) if k0 < kd else None
    W = W0 if W1 is None else np.concatenate([W0, W1], 0)
    want = dOut.astype(np.float64) @ W.astype(np.float64)
    st = _st()
    dA = torch.full((rows, nd), float("nan"), device=DEV)
    st.call("dprnn_dgrad", _p(_t(dOut)), kd, _p(_t(W0)), nd, k0, _p(_t(W1)) if W1 is not None else 0, nd, _p(dA), nd, nd, rows)
    torch.cuda.synchronize()
    assert rel_l2(dA.cpu().numpy(), want) < 1e-6


@pytest.mark.parametrize("M,N,rows,acc", [(512, 256, 5000, 0), (256, 128, 777, 1), (64, 128, 4099, 0), (128, 64, 130, 1), (128, 256, 3000, 0)])
def test_wgrad_matches_matmul(M, N, rows, acc):
    """dW (+)= P^T Q with column sums of P, ragged row counts, 64-wide operands, transposed destination."""
    rng = np.random.default_rng(M + N + rows)
    ldp, ldq = M + 8, 128
    P = rng.standard_normal((rows, ldp)).astype(np.float32)
    Q0 = rng.standard_normal((rows, ldq)).astype(np.float32)
    Q1 = rng.standard_normal((rows, ldq)).astype(np.float32) if N > 128 else None
    n0 = min(N, 128)
    Qf = Q0[:, :n0] if Q1 is None else np.concatenate([Q0, Q1[:, :N - 128]], 1)
    want = P[:, :M].astype(np.float64).T @ Qf.astype(np.float64)
    wb = P[:, :M].astype(np.float64).sum(0)
    st = _st()
    init = rng.standard_normal((N, M)).astype(np.float32)          # destination stored transposed: dW[m][n] at n*M + m
    dW = _t(init.copy())
    db = _t(np.ones(M, np.float32))
    ws = torch.empty(st.lib.dprnn_wgrad_workspace_floats(M, N, rows), device=DEV)
    st.call("dprnn_wgrad", _p(_t(P)), ldp, M, _p(_t(Q0)), ldq, n0, 0, _p(_t(Q1)) if Q1 is not None else 0, ldq, N - 128 if Q1 is not None else 0, 0,
            rows, 1, 1, _p(dW), 1, M, _p(db), 0, acc, _p(ws))
    torch.cuda.synchronize()
    got = dW.cpu().numpy().T.astype(np.float64) - (init.T if acc else 0)
    gb = db.cpu().numpy().astype(np.float64) - (1 if acc else 0)
    assert rel_l2(got, want) < 2e-6
    assert rel_l2(gb, wb) < 2e-6


def test_rows64_groupnorm_backward_matches_autograd():
    """dprnn_rows64_affine / dprnn_rows64_norm_bwd (GroupNorm(1,64) of the materialised d0 in DPRNN-Spe-IRA) vs F.group_norm."""
    rng = np.random.default_rng(11)
    B, L, N = 3, 77, 64
    x = torch.from_numpy(rng.standard_normal((B, L, N)).astype(np.float32) * 1.5 + 0.2)
    x[:, ::3, ::2] = 0.0                                            # d0 = mask*relu(...) holds exact zeros
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, N).astype(np.float32)).requires_grad_(True)
    beta = torch.from_numpy(rng.uniform(-0.3, 0.3, N).astype(np.float32)).requires_grad_(True)
    dy = torch.from_numpy(rng.standard_normal((B, L, N)).astype(np.float32))
    xr = x.clone().requires_grad_(True)
    y = torch.nn.functional.group_norm(xr.transpose(1, 2), 1, gamma, beta, 1e-5).transpose(1, 2)
    y.backward(dy)
    st = _st()
    xd, dyd, g, b = _t(x), _t(dy), _t(gamma.detach()), _t(beta.detach())
    stats = torch.zeros(B, 2, device=DEV, dtype=torch.float64)
    st.call("dprnn_rows_stats", _p(xd), _p(stats), B, L, N)
    sc = torch.empty(B, N, device=DEV); sh = torch.empty(B, N, device=DEV)
    st.call("dprnn_bottleneck_prep", _p(stats), _p(g), _p(b), 1e-5, 0, 0, _p(sc), _p(sh), B, L, N)
    A = torch.empty(B * L, N, device=DEV)
    st.call("dprnn_rows64_affine", _p(xd), _p(sc), _p(sh), _p(A), B, L)
    red = torch.zeros(2 * B, device=DEV, dtype=torch.float64)
    dgb = torch.zeros(2 * N, device=DEV, dtype=torch.float64)
    dX = torch.full((B * L, N), float("nan"), device=DEV)
    st.call("dprnn_rows64_norm_bwd", _p(xd), _p(dyd), _p(stats), _p(g), 1e-5, _p(red), _p(dgb), _p(dX), B, L)
    torch.cuda.synchronize()
    assert rel_l2(A.cpu().numpy().reshape(B, L, N), y.detach().numpy()) < 1e-6
    assert rel_l2(dX.cpu().numpy().reshape(B, L, N), xr.grad.numpy()) < 2e-6
    assert rel_l2(dgb[:N].cpu().numpy(), gamma.grad.numpy()) < 2e-6
    assert rel_l2(dgb[N:].cpu().numpy(), beta.grad.numpy()) < 2e-6


def test_mask_decode_backward_with_encoder_domain_gradient():
    """dprnn_mask_decode_bwd incl. dM0 (gradient of d0 = mask_0*enc, DPRNN-Spe-IRA) vs autograd of the stock operators."""
    rng = np.random.default_rng(12)
    B, T, N, ks, stride = 2, 203, 64, 4, 2
    L = (T - ks) // stride + 1
    wav = torch.from_numpy(rng.standard_normal((B, T)).astype(np.float32))
    We = torch.from_numpy((rng.standard_normal((N, 1, ks)) * 0.5).astype(np.float32)).requires_grad_(True)
    Wd = torch.from_numpy((rng.standard_normal((N, 1, ks)) * 0.5).astype(np.float32)).requires_grad_(True)
    E = torch.from_numpy(rng.standard_normal((B, 2, N, L)).astype(np.float32)).requires_grad_(True)      # end_conv1x1 output
    d_out = torch.from_numpy(rng.standard_normal((B, 2, (L - 1) * stride + ks)).astype(np.float32))
    d_d0 = torch.from_numpy(rng.standard_normal((B, N, L)).astype(np.float32))
    enc = torch.relu(torch.nn.functional.conv1d(wav[:, None], We, stride=stride))                         # [B,N,L]
    M = torch.sigmoid(E)
    masked = M * enc[:, None]
    out = torch.stack([torch.nn.functional.conv_transpose1d(masked[:, i], Wd, stride=stride)[:, 0] for i in range(2)], 1)
    ((out * d_out).sum() + (masked[:, 0] * d_d0).sum()).backward()
    st = _st()
    mask_rows = M.detach().permute(0, 3, 1, 2).reshape(B * L * 2, N)                                      # rows (b,l,spk)
    dE = torch.full((B * L * 2, N), float("nan"), device=DEV)
    dEnc = torch.full((B * L, N), float("nan"), device=DEV)
    dWd = torch.zeros(N * ks, device=DEV, dtype=torch.float64)
    dM0 = d_d0.permute(0, 2, 1).reshape(B * L, N)
    st.call("dprnn_mask_decode_bwd", _p(_t(d_out)), _p(_t(wav)), _p(_t(We.detach())), _p(_t(Wd.detach())), _p(_t(mask_rows)), 0,
            _p(dE), _p(dEnc), _p(dWd), _p(_t(dM0)), B, T, L, N, ks, stride)
    torch.cuda.synchronize()
    want_dE = E.grad.permute(0, 3, 1, 2).reshape(B * L * 2, N).numpy()
    assert rel_l2(dE.cpu().numpy(), want_dE) < 2e-6
    assert rel_l2(dWd.cpu().numpy().reshape(N, ks), Wd.grad.numpy()[:, 0]) < 2e-6


@pytest.mark.parametrize("kd,nd,f16", [(128, 256, True), (128, 256, False), (128, 128, False)])
def test_dgrad_leaves_the_maximum_of_its_output(kd, nd, f16):
    """dprnn_dgrad_store_absmax: the next dprnn_dgrad also writes max|dA| (epilogue of the f16 form, a pass after the fp32 forms);
    the selector is consumed by that call only."""
    rng = np.random.default_rng(7)
    rows = 1256
    dOut = (rng.standard_normal((rows, kd)) * 3e-4).astype(np.float32)
    W0 = rng.standard_normal((kd, nd)).astype(np.float32)
    st = _st()
    dA = torch.full((rows, nd), float("nan"), device=DEV)
    slot = torch.full((1,), 123.0, device=DEV)
    d = _t(dOut)
    if f16:
        amax = torch.zeros(1, device=DEV)
        st.call("dprnn_absmax", _p(d), rows * kd, _p(amax))
        st.lib.dprnn_dgrad_use_f16_pieces(amax.data_ptr())
    st.lib.dprnn_dgrad_store_absmax(slot.data_ptr())
    st.call("dprnn_dgrad", _p(d), kd, _p(_t(W0)), nd, kd, 0, nd, _p(dA), nd, nd, rows)
    torch.cuda.synchronize()
    want = dOut.astype(np.float64) @ W0.astype(np.float64)
    assert rel_l2(dA.cpu().numpy(), want) < 1e-6
    assert float(slot) == float(dA.abs().max())
    slot.fill_(123.0)
    st.call("dprnn_dgrad", _p(d), kd, _p(_t(W0)), nd, kd, 0, nd, _p(dA), nd, nd, rows)     # selector already consumed
    torch.cuda.synchronize()
    assert float(slot) == 123.0


def test_groupnorm_backward_leaves_the_maximum_of_dz():
    """dprnn_gn_bwd_dz_store_absmax: same dZ as without it, and max|dZ| in the slot (what dprnn_absmax(dZ) gives)."""
    rng = np.random.default_rng(3)
    B, rpu = 3, 517
    Z = _t(rng.standard_normal((B * rpu, 128)).astype(np.float32))
    dY = _t((rng.standard_normal((B * rpu, 128)) * 1e-3).astype(np.float32))
    gamma = _t(rng.uniform(0.5, 1.5, 128).astype(np.float32))
    zz = Z.view(B, -1).double()
    stats = torch.stack([zz.sum(1), (zz * zz).sum(1)], 1).contiguous()
    red = _t(rng.standard_normal(2 * B)).double().contiguous()
    st = _st()
    outs = []
    slot = torch.full((1,), 123.0, device=DEV)
    for use in (False, True):
        dZ = torch.full((B * rpu, 128), float("nan"), device=DEV)
        if use:
            st.lib.dprnn_gn_bwd_dz_store_absmax(slot.data_ptr())
        st.call("dprnn_gn_bwd_dz", _p(dY), _p(Z), _p(stats), _p(red), _p(gamma), 1e-8, B, rpu, _p(dZ))
        torch.cuda.synchronize()
        outs.append(dZ)
        if not use:
            assert float(slot) == 123.0
    assert torch.equal(outs[0], outs[1])
    ref = torch.zeros(1, device=DEV)
    st.call("dprnn_absmax", _p(outs[1]), B * rpu * 128, _p(ref))
    torch.cuda.synchronize()
    assert float(slot) == float(ref) == float(outs[1].abs().max())


@pytest.mark.parametrize("mode", [0, 3])
@pytest.mark.parametrize("ks,stride,T,relu", [(2, 1, 300, 0), (4, 2, 203, 1)])
def test_training_forward_of_the_mask_head_keeps_what_the_backward_needs(ks, stride, T, relu, mode):
    """dprnn_head_gate_fwd_train / dprnn_head_mask_decode_fwd_train in both arithmetics: the same V, taps and mask * enc as the
    inference entry points, plus [tanh | sigmoid] and the mask themselves (float64 numpy as the reference)."""
    rng = np.random.default_rng(T + mode)
    B, N = 3, 64
    L = (T - ks) // stride + 1
    R = B * L * 2
    U = rng.standard_normal((R, 128)).astype(np.float32)
    Wo = (rng.standard_normal((128, 128)) / 11).astype(np.float32); bo = rng.standard_normal(128).astype(np.float32)
    Wg = (rng.standard_normal((128, 128)) / 11).astype(np.float32); bg = rng.standard_normal(128).astype(np.float32)
    Wend = (rng.standard_normal((N, 128)) / 11).astype(np.float32)
    wav = (0.1 * rng.standard_normal((B, T))).astype(np.float32)
    We = (rng.standard_normal((N, 1, ks)) / np.sqrt(ks)).astype(np.float32)
    Wd = (rng.standard_normal((N, 1, ks)) / 8).astype(np.float32)
    u = U.astype(np.float64)
    ta, sg = np.tanh(u @ Wo.T + bo), 1 / (1 + np.exp(-(u @ Wg.T + bg)))
    mk = (ta * sg) @ Wend.T.astype(np.float64)
    mk = np.maximum(mk, 0) if relu else 1 / (1 + np.exp(-mk))
    st = _st()
    st.lib.dprnn_set_gemm_mode(mode)
    try:
        d = {k: _t(v) for k, v in dict(U=U, Wo=Wo, bo=bo, Wg=Wg, bg=bg, Wend=Wend, wav=wav, We=We, Wd=Wd).items()}
        V0, V1 = torch.empty(R, 128, device=DEV), torch.empty(R, 128, device=DEV)
        TG = torch.full((R, 256), float("nan"), device=DEV)
        st.call("dprnn_head_gate_fwd", _p(d["U"]), _p(d["Wo"]), _p(d["bo"]), _p(d["Wg"]), _p(d["bg"]), _p(V0), R)
        st.call("dprnn_head_gate_fwd_train", _p(d["U"]), _p(d["Wo"]), _p(d["bo"]), _p(d["Wg"]), _p(d["bg"]), _p(V1), _p(TG), R)
        taps0, taps1 = torch.empty(B, 2, L, ks, device=DEV), torch.empty(B, 2, L, ks, device=DEV)
        m0, m1 = torch.empty(B, L, N, device=DEV), torch.empty(B, L, N, device=DEV)
        mask = torch.full((R, N), float("nan"), device=DEV)
        st.call("dprnn_head_mask_decode_fwd", _p(V0), _p(d["Wend"]), _p(d["wav"]), _p(d["We"]), _p(d["Wd"]), _p(taps0), _p(m0), relu,
                B, T, L, N, ks, stride)
        st.call("dprnn_head_mask_decode_fwd_train", _p(V1), _p(d["Wend"]), _p(d["wav"]), _p(d["We"]), _p(d["Wd"]), _p(taps1), _p(mask),
                _p(m1), relu, B, T, L, N, ks, stride)
        torch.cuda.synchronize()
    finally:
        st.lib.dprnn_set_gemm_mode(0)
    assert torch.equal(V0, V1) and torch.equal(taps0, taps1) and torch.equal(m0, m1)
    assert rel_l2(TG.cpu().numpy(), np.concatenate([ta, sg], 1)) < 2e-6
    assert rel_l2(mask.cpu().numpy(), mk) < 2e-6


@pytest.mark.parametrize("B,rpu,dyscale,seed", [(3, 1237, 1.0, 0), (2, 48500 // 25, 3e-4, 1), (5, 40, 50.0, 2), (1, 33, 1.0, 3)])
def test_fused_norm_and_linear_backward_in_one_pass(B, rpu, dyscale, seed):
    """dprnn_gn_bwd_stats_ex + dprnn_lin_bwd_f16 (round 5): the GroupNorm backward's second pass, the Linear's data gradient and its weight
    gradient in one pass over the rows, dZ never stored, its f16 scale from a bound the first pass leaves.  Against float64 autograd of
    y = gamma * (z - mean) / sqrt(var + eps) + beta, z = H W^T + b (dprnn.py:88-90) for an upstream dY: dH, dW, db to f16-pieces
    accuracy; the bound is above max|dZ| and within a small factor of it; max|dH| lands in the slot; rows not a multiple of the 32-row
    chunk, utterance boundaries inside a chunk, several utterances per workgroup, gradients of very different magnitude."""
    rng = np.random.default_rng(seed)
    rows = B * rpu
    H = (rng.standard_normal((rows, 256)) * 0.4).astype(np.float32)
    W = (rng.standard_normal((128, 256)) / 16).astype(np.float32)
    b = (rng.standard_normal(128) * 0.1).astype(np.float32)
    gamma = (1.0 + 0.3 * rng.standard_normal(128)).astype(np.float32)
    dY = (rng.standard_normal((rows, 128)) * dyscale * (1.0 + 4.0 * (rng.random((rows, 1)) < 0.01))).astype(np.float32)
    eps = 1e-5
    Ht = torch.from_numpy(H).double().requires_grad_(True)
    Wt = torch.from_numpy(W).double().requires_grad_(True)
    bt = torch.from_numpy(b).double().requires_grad_(True)
    z = Ht @ Wt.T + bt
    zf = z.detach().float()                                     # what the forward left in HBM
    zb = z.reshape(B, rpu * 128)
    mean, var = zb.mean(1, keepdim=True), zb.var(1, unbiased=False, keepdim=True)
    y = ((zb - mean) / torch.sqrt(var + eps)).reshape(rows, 128) * torch.from_numpy(gamma).double()
    z.retain_grad()
    (y * torch.from_numpy(dY).double()).sum().backward()
    st = _st()
    lib = st.lib
    Zd, dYd, Hd, Wd, gd = _t(zf), _t(dY), _t(H), _t(W), _t(gamma)
    z64 = zf.double().reshape(B, -1)
    stats = _t(torch.stack([z64.sum(1), (z64 * z64).sum(1)], 1))
    red = torch.zeros(2 * B + 256, device=DEV, dtype=torch.float64)
    absmax = torch.empty(B, 2, device=DEV, dtype=torch.int32)
    st.call("dprnn_gn_bwd_stats_ex", _p(dYd), 0, 0, 0, _p(Zd), _p(stats), _p(gd), eps, B, rpu, _p(red), red.data_ptr() + 8 * 2 * B, _p(absmax))
    dH = torch.empty(rows, 256, device=DEV)
    dW = torch.empty(128, 256, device=DEV)
    db = torch.empty(128, device=DEV)
    hmax = torch.full((1,), -1.0, device=DEV)
    ws = torch.empty(lib.dprnn_lin_bwd_workspace_floats(rows), device=DEV)
    st.call("dprnn_lin_bwd_f16", _p(dYd), _p(Zd), _p(stats), _p(red), _p(absmax), _p(gd), eps, B, rpu, _p(Hd), _p(Wd), _p(dH), _p(dW), _p(db),
            _p(hmax), _p(ws))
    torch.cuda.synchronize()
    dz = z.grad.numpy()
    e_h, e_w, e_b = rel_l2(dH.cpu().numpy(), Ht.grad.numpy()), rel_l2(dW.cpu().numpy(), Wt.grad.numpy()), rel_l2(db.cpu().numpy(), bt.grad.numpy())
    # the bound the first pass leaves: rstd (max|gamma dy| + |m1| + max|x_hat| |m2|) per utterance
    am = absmax.view(torch.float32).cpu().double()
    cnt = rpu * 128.0
    rs = 1.0 / torch.sqrt(var.detach().reshape(B) + eps)
    r = red[:2 * B].cpu().reshape(B, 2)
    bound = float((rs * (am[:, 0] + (r[:, 0] / cnt).abs() + am[:, 1] * (r[:, 1] / cnt).abs())).max())
    true = float(np.abs(dz).max())
    print(f"B {B} rows/utt {rpu}: dH {e_h:.2e} dW {e_w:.2e} db {e_b:.2e}; max|dZ| {true:.3e}, bound {bound:.3e} ({bound / true:.2f} x); "
          f"max|dH| {float(hmax):.4e} vs {float(Ht.grad.abs().max()):.4e}")
    assert true <= bound * (1 + 1e-5) and bound < 8 * true
    assert abs(float(hmax) - float(Ht.grad.abs().max())) <= 1e-5 * float(Ht.grad.abs().max())
    assert max(e_h, e_w, e_b) < 3e-6, (e_h, e_w, e_b)
