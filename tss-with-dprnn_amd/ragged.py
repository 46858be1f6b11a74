"""Ragged batches: utterances of different lengths through one launch per DPRNN path (the reference evaluates one
utterance at a time, inferencer.py:48-78, inferencer_spe.py:20-54).  Part of the engine (see engine.py)."""
from __future__ import annotations

import threading

import torch

from . import engine as E
from ._lib import DprnnError


_T0 = [0.0]


def _mark(what: str) -> None:
    """DPRNN_RAGGED_TRACE=1: host-side time of a phase boundary of a ragged forward (ms since the forward began), for reading beside a kernel
    timeline: where the host was when the GPU went idle."""
    import os
    import time
    if os.environ.get("DPRNN_RAGGED_TRACE") == "1":
        if what == "begin":
            _T0[0] = time.perf_counter()
        print(f"[ragged host] {what:28s} {(time.perf_counter() - _T0[0]) * 1e3:8.1f} ms", flush=True)


class _Fan:
    """Independent per-utterance work dealt round-robin over persistent side streams; join() makes the caller's stream wait
    for all of it (and hands the tensors produced there over to it)."""

    def __init__(self, st: Stage, n_items: int, enable: bool = True):
        self.main = st
        n = min(E.RAGGED_STREAMS, n_items) if enable else 0
        if n > 1:
            pool = E._persistent_streams(E._FAN_POOL, (threading.get_ident(), str(st.device)), st.device, n)
            fork = torch.cuda.Event()
            fork.record(st.tstream)
            self.stages = [E.Stage(st.device, s) for s in pool]
            for sg in self.stages:
                sg.tstream.wait_event(fork)
        else:
            self.stages = []
        self.made = []

    def run(self, u: int, fn):
        """fn(stage) on the side stream of item u; tensors it returns are kept alive / handed over at join()."""
        if not self.stages:
            return fn(self.main)
        sg = self.stages[u % len(self.stages)]
        with torch.cuda.stream(sg.tstream):
            out = fn(sg)
        self.made.append(out)
        return out

    def join(self):
        def walk(o):
            if isinstance(o, torch.Tensor):
                o.record_stream(self.main.tstream)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    walk(v)
        for sg in self.stages:
            ev = torch.cuda.Event()
            ev.record(sg.tstream)
            self.main.tstream.wait_event(ev)
        walk(self.made)
        self.made = []


def _bn_training(module) -> bool:
    """BatchNorm in train mode updates its running buffers in place per utterance: such work stays on one stream, in order."""
    return any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training for m in module.modules())


def ragged_supported(sep) -> bool:
    from . import padded
    wide = sep.input_size > padded.N_BUILT or sep.feature_size > padded.F_BUILT or any(
        b.intra_rnn.rnn.hidden_size > padded.F_BUILT for b in sep.dprnn_blocks)
    if wide:                     # the any-width kernels batch ragged forwards themselves (wide.tasnet_forward_ragged / spe_forward_ragged), any cell;
        # a speaker encoder with BatchNorm in train mode included (one utterance after the other through the speaker branch, in order)
        return True
    return E.current_mode() in (1, 3) and all(b.intra_rnn.rnn.mode == "LSTM" and b.inter_rnn.rnn.mode == "LSTM" for b in sep.dprnn_blocks)


@E._leaves_exact_mode
def dprnn_process_ragged(st: Stage, sep, Ys, wavs, enc_w, dec_w, geos, relu_act: bool, want_masked: bool = False):
    """dprnn_process for utterances of different lengths.  Ys[u]: frames [1][L_u][F]; wavs[u]: [1][T_u]; geos[u]: Geometry
    with B = 1.  Returns (taps list [1][2][L_u][ks], masked list [1][L_u][N] or None)."""
    import numpy as np
    dev = Ys[0].device
    n = len(Ys)
    K, P, ks, N = geos[0].K, geos[0].P, geos[0].ks, geos[0].N
    blocks = list(sep.dprnn_blocks)
    if not ragged_supported(sep):
        raise DprnnError("ragged batches need gate-GEMM mode 1 or 3 and LSTM cells")
    mode = E.current_mode()
    st.lib.dprnn_set_gemm_mode(mode)
    rows = np.array([g.S * K for g in geos], dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    total = int(starts[-1])
    chunks = np.array([g.S for g in geos], dtype=np.int64)
    seq_utt = np.repeat(np.arange(n, dtype=np.int32), chunks)
    desc = []
    for u in np.argsort(-chunks, kind="stable"):                 # longest utterances first: their workgroups run longest
        for k0 in range(0, K, 64):
            r0 = int(starts[u]) + k0
            desc.append((r0 & 0xFFFFFFFF, r0 >> 32, int(chunks[u]), min(64, K - k0), int(u), 0, 0, 0))
    desc = np.array(desc, dtype=np.int64).astype(np.uint32).view(np.int32)
    l_off = np.concatenate([[0], np.cumsum([g.L for g in geos])]).astype(np.int64)          # frames of the utterances, back to back (mask head)
    tables = E._upload(torch.from_numpy(np.concatenate([rows.view(np.int32), starts.view(np.int32), l_off.view(np.int32), seq_utt, desc.reshape(-1)])), dev)
    o = 0
    utt_rows = tables[o: o + 2 * n]; o += 2 * n
    utt_starts = tables[o: o + 2 * (n + 1)]; o += 2 * (n + 1)
    l_off_d = tables[o: o + 2 * (n + 1)]; o += 2 * (n + 1)
    seq_utt_d = tables[o: o + seq_utt.size]; o += seq_utt.size
    desc_d = tables[o:]
    n_tiles = desc.shape[0]

    packed = torch.empty(2 * len(blocks), 2 * E.PACK_PER_DIR_MAX, device=dev, dtype=torch.float32)
    bsum = torch.empty(2 * len(blocks), 2 * 4 * E.F_BUILT, device=dev, dtype=torch.float32)
    for i, blk in enumerate(blocks):
        for j, rnn in enumerate((blk.intra_rnn, blk.inter_rnn)):
            _, fw, bw = E._rnn_tensors(rnn, dev)
            E._set_slot(st.lib, E.path_slot(i, j == 1))
            st.call("dprnn_rnn_pack_ex", mode, 0, *[E._p(t) for t in fw], *[E._p(t) for t in bw], E._p(packed[2 * i + j]), E._p(bsum[2 * i + j]))

    m = total * E.F_BUILT
    # H [total][2F] is dead once the last projection has read it; the mask head of utterance u then takes its U [L_u][2F] and
    # V [2 L_u][F] from the same region.  S_u K rows of 2F floats cover them when hop <= K / 2 (S K >= 2 L); a larger hop needs
    # more, so every utterance gets max(S_u K 2F, 4 L_u F) floats at its own offset (the non-ragged arena does the same through
    # dprnn_workspace_bytes).
    head_need = np.array([max(g.S * K * 2 * E.F_BUILT, 4 * g.L * E.F_BUILT) for g in geos], dtype=np.int64)
    head_off = np.concatenate([[0], np.cumsum(head_need)]).astype(np.int64)
    arena = torch.empty(3 * m + int(head_off[-1]), device=dev, dtype=torch.float32)
    X, X2, Z, H = arena[:m], arena[m: 2 * m], arena[2 * m: 3 * m], arena[3 * m:]
    stats = torch.zeros(2 * len(blocks), n, 2, device=dev, dtype=torch.float64)
    # the Linear folded into the recurrence (time-split form, LIN_FOLD above) where the one pipeline's launches cover the chip
    ncu = st.lib.dprnn_num_cus()
    tiles_intra, tiles_inter = -(-int(seq_utt.size) // 64), n_tiles
    covers = lambda tiles: E.LIN_FOLD == "split" or (E.LIN_FOLD == "auto" and 2 * tiles >= E.FOLD_SPLIT_MIN_FILL * ncu)
    fold = {True: mode == 3 and covers(tiles_intra), False: mode == 3 and covers(tiles_inter)}
    if any(fold.values()):
        Z2 = torch.empty(m, device=dev, dtype=torch.float32)
        lin_state = torch.empty(st.lib.dprnn_rnn_lin_state_floats(64 * max(tiles_intra, tiles_inter), 2064), device=dev, dtype=torch.float32)
        linp = torch.empty(2 * len(blocks), st.lib.dprnn_lin_pack_floats(), device=dev, dtype=torch.float32)
    fan = _Fan(st, n)
    for u in range(n):
        fan.run(u, lambda sg, u=u: sg.call("dprnn_segment_fwd", E._p(Ys[u]), E._p(X[int(starts[u]) * E.F_BUILT:]), 1, geos[u].L, K, P,
                                           geos[u].S))
    fan.join()
    pending = None
    for i, blk in enumerate(blocks):
        for intra in (True, False):
            rnn, lin, norm = (blk.intra_rnn, blk.intra_linear, blk.intra_norm) if intra else \
                (blk.inter_rnn, blk.inter_linear, blk.inter_norm)
            slot = 2 * i + (0 if intra else 1)
            ndir = 2 if rnn.rnn.bidirectional else 1
            pn = (0, 0, 0, 0, 0.0) if pending is None else (E._p(Z), E._p(pending[0]), E._p(pending[1]), E._p(pending[2]), pending[3])
            hp = E.H_PIECES and mode == 3
            E._set_slot(st.lib, E.path_slot(i, not intra))
            w = E._param(lin.weight, "linear.weight", dev)
            if w.shape != (E.F_BUILT, ndir * E.F_BUILT):
                raise DprnnError(f"linear weight shape {tuple(w.shape)} does not match ndir={ndir}")
            if fold[intra] and ndir == 2:
                st.call("dprnn_lin_pack_f16", E._p(w), 2, E._p(linp[slot]))
                rag = (E._p(utt_rows), E._p(seq_utt_d) if intra else 0, 0 if intra else E._p(desc_d), int(seq_utt.size) if intra else n_tiles)
                tail = (E._p(packed[slot]), E._p(bsum[slot]), E._p(linp[slot]))
                st.call("dprnn_rnn_lin_split_ragged_fwd", 1, E._p(X), *pn, *rag, 0 if pending is None else E._p(X2), *tail, 0, E._p(Z2), 0,
                        E._p(lin_state), K if intra else 0, K)
                if pending is not None:
                    X, X2 = X2, X
                st.call("dprnn_rnn_lin_split_ragged_fwd", 2, E._p(X), 0, 0, 0, 0, 0.0, *rag, 0, *tail, E._p(E._param(lin.bias, "linear.bias", dev)),
                        E._p(Z2), E._p(stats[slot]), E._p(lin_state), K if intra else 0, K)
                Z, Z2 = Z2, Z
                E._count_scale(stats[slot], norm, st, n)
                gamma, beta, eps = E._norm_params(norm)
                pending = (stats[slot], E._param(gamma, "norm.weight", dev), E._param(beta, "norm.bias", dev), eps)
                continue
            st.call("dprnn_rnn_fwd_ragged_hp_ex" if hp else "dprnn_rnn_fwd_ragged_ex", mode, E._p(X), *pn, E._p(utt_rows), E._p(seq_utt_d) if intra else 0, 0 if intra else E._p(desc_d),
                    int(seq_utt.size) if intra else n_tiles, 0 if pending is None else E._p(X2), E._p(packed[slot]), E._p(bsum[slot]), E._p(H),
                    K if intra else 0, K, ndir)
            if pending is not None:
                X, X2 = X2, X
            if hp:
                st.call("dprnn_proj_stats_hp_fwd", E._p(H), E._p(w), E._p(E._param(lin.bias, "linear.bias", dev)), E._p(Z), E._p(stats[slot]),
                        total, 1, E._p(utt_starts), n, ndir * E.F_BUILT)
            else:
                st.call("dprnn_proj_stats_ragged_fwd", E._p(H), E._p(w), E._p(E._param(lin.bias, "linear.bias", dev)), E._p(Z), E._p(stats[slot]),
                        total, E._p(utt_starts), n, ndir * E.F_BUILT)
            E._count_scale(stats[slot], norm, st, n)
            gamma, beta, eps = E._norm_params(norm)
            pending = (stats[slot], E._param(gamma, "norm.weight", dev), E._param(beta, "norm.bias", dev), eps)

    taps, masked = [], []
    head_w = (E._p(E._param(sep.prelu.weight, "prelu.weight", dev)), E._p(E._param(sep.conv2d.weight, "conv2d.weight", dev)),
              E._p(E._param(sep.conv2d.bias, "conv2d.bias", dev)))
    # The mask head.  conv2d + overlap-add and the decode address an utterance's own geometry and run per utterance on the side streams;
    # the gate between them (tanh(out(U)) * sigmoid(gate(U)), dprnn.py:157-158,180) is row-wise, so the utterances' U rows sit back to back
    # and it is ONE launch over all of them -- a per-utterance launch stages its 128 KB of weights for a few hundred rows per workgroup and
    # ran at a quarter of the batched rate (profiles/r05_ragged_ira_kernel_stats.csv).  U_all [sum L][2F] | V_all [2 sum L][F] in H's region.
    sum_l = int(l_off[-1])
    assert 4 * sum_l * E.F_BUILT <= int(head_off[-1])
    U_all, V_all = H[: 2 * sum_l * E.F_BUILT], H[2 * sum_l * E.F_BUILT: 4 * sum_l * E.F_BUILT]

    def head_conv(sg, u):
        g = geos[u]
        r0 = int(starts[u])
        U = U_all[int(l_off[u]) * 2 * E.F_BUILT: int(l_off[u + 1]) * 2 * E.F_BUILT]
        E._set_slot(sg.lib, E.SLOT_HEAD_CONV)
        if pending is None:
            sg.call("dprnn_head_conv_ola_fwd", E._p(X[r0 * E.F_BUILT:]), *head_w, E._p(U), 1, g.L, K, P, g.S)
        else:
            sg.call("dprnn_head_conv_ola_norm_fwd", E._p(X[r0 * E.F_BUILT:]), E._p(Z[r0 * E.F_BUILT:]), E._p(pending[0][u]), E._p(pending[1]),
                    E._p(pending[2]), pending[3], *head_w, E._p(U), 1, g.L, K, P, g.S)

    gate_w = (E._p(E._param(sep.out[0].weight, "out.0.weight", dev)), E._p(E._param(sep.out[0].bias, "out.0.bias", dev)),
              E._p(E._param(sep.gate[0].weight, "gate.0.weight", dev)), E._p(E._param(sep.gate[0].bias, "gate.0.bias", dev)))

    def head_decode(sg, u):
        g = geos[u]
        V = V_all[int(l_off[u]) * 2 * E.F_BUILT: int(l_off[u + 1]) * 2 * E.F_BUILT]
        if not E.RAGGED_GATE_BATCHED:
            E._set_slot(sg.lib, E.SLOT_HEAD_GATE)
            sg.call("dprnn_head_gate_fwd", E._p(U_all[int(l_off[u]) * 2 * E.F_BUILT:]), *gate_w, E._p(V), g.L * 2)
        t = torch.empty(1, 2, g.L, ks, device=dev, dtype=torch.float32)
        mk = torch.empty(1, g.L, N, device=dev, dtype=torch.float32) if want_masked else None
        sg.call("dprnn_head_mask_decode_fwd", E._p(V), E._p(E._param(sep.end_conv1x1.weight, "end_conv1x1.weight", dev)),
                E._p(wavs[u]), E._p(enc_w), E._p(dec_w), E._p(t), E._p(mk), 1 if relu_act else 0, 1, g.T, g.L, N, ks, g.stride)
        return t, mk

    if E.RAGGED_HEAD_CONV_BATCHED and mode == 3:       # one launch over every utterance's frames (a row finds its utterance in l_off)
        E._set_slot(st.lib, E.SLOT_HEAD_CONV)
        pn_ = (0, 0, 0, 0, 0.0) if pending is None else (E._p(Z), E._p(pending[0]), E._p(pending[1]), E._p(pending[2]), pending[3])
        st.call("dprnn_head_conv_ola_ragged_fwd", E._p(X), *pn_, *head_w, E._p(U_all), E._p(utt_starts), E._p(l_off_d), n, sum_l, K, P)
    else:
        fan = _Fan(st, n)
        for u in range(n):
            fan.run(u, lambda sg, u=u: head_conv(sg, u))
        fan.join()
    if E.RAGGED_GATE_BATCHED:
        E._set_slot(st.lib, E.SLOT_HEAD_GATE)
        st.call("dprnn_head_gate_fwd", E._p(U_all), *gate_w, E._p(V_all), 2 * sum_l)
    fan = _Fan(st, n)
    res = [fan.run(u, lambda sg, u=u: head_decode(sg, u)) for u in range(n)]
    fan.join()
    taps = [r[0] for r in res]
    masked = [r[1] for r in res]
    return taps, (masked if want_masked else None)


def _tasnet_forward_ragged_impl(model, wavs):
    """DPRNNTasNet.forward for a list of 1-D waveforms of different lengths -> list of [2, T_out_u] (each equal to the
    batch-of-one forward of that utterance)."""
    cfg = model.cfg
    E.check_supported(cfg)
    dev = wavs[0].device
    with torch.cuda.device(dev), E._scale_record(model, dev) as rec:
        st = E.Stage(dev)
        sep = model.separation
        enc_w = E._param(model.encoder.conv1d.weight, "encoder.conv1d.weight", dev)
        dec_w = E._param(model.decoder.weight, "decoder.weight", dev)
        ws = [E.check_input(w.reshape(1, -1)) for w in wavs]
        geos = [E.make_geometry(1, w.shape[1], cfg["input_size"], cfg["chunk_length"], cfg["hop_length"], cfg["kernel_size"],
                              cfg["stride"]) for w in ws]
        stats0 = torch.zeros(len(ws), 1, 2, device=dev, dtype=torch.float64)
        fan = _Fan(st, len(ws))
        Ys = [fan.run(u, lambda sg, u=u: E.frames_from_waveform(sg, ws[u], enc_w, sep.bottleneck[0], sep.bottleneck[1], geos[u],
                                                                stats0[u])) for u in range(len(ws))]
        fan.join()
        taps, _ = dprnn_process_ragged(st, sep, Ys, ws, enc_w, dec_w, geos, cfg["activation_type"] == "relu")
        fan = _Fan(st, len(ws))
        outs = [fan.run(u, lambda sg, u=u: E.decode(sg, taps[u], geos[u], 2)) for u in range(len(ws))]
        fan.join()
        E._scale_end(st, rec)
        return [o[0] for o in outs]


def _spe_forward_ragged_impl(model, wavs, auxs, aux_lens, ira: bool = False):
    """DPRNNSpeTasNet / DPRNNSpeIRATasNet forward for lists of mixtures and reference utterances of different lengths ->
    list of (estimate [T_out_u], logits [num_spks]); every utterance keeps batch-of-one semantics (BatchNorm statistics of
    the speaker branch included, as the reference's one-at-a-time loop has them)."""
    import numpy as np
    _mark("begin")
    cfg = model.cfg
    E.check_supported_spe(cfg)
    dev = wavs[0].device
    with torch.cuda.device(dev), E._scale_record(model, dev) as rec, E.eval_bn_folds(E.Stage(dev), model.separation):
        st = E.Stage(dev)
        N, ks, stride = cfg["input_size"], cfg["kernel_size"], cfg["stride"]
        sep = model.separation
        se = sep.spk_encoder
        enc_w = E._param(model.encoder.conv1d.weight, "encoder.conv1d.weight", dev)
        dec_w = E._param(model.decoder.weight, "decoder.weight", dev)
        relu_act = cfg["activation_type"] == "relu"
        n = len(wavs)
        ws = [E.check_input(w.reshape(1, -1)) for w in wavs]
        axs = [E.check_input(a.reshape(1, -1), "aux") for a in auxs]
        geos = [E.make_geometry(1, w.shape[1], N, cfg["chunk_length"], cfg["hop_length"], ks, stride) for w in ws]
        stats = torch.zeros(n, 4, 1, 2, device=dev, dtype=torch.float64)     # per utterance: bottleneck, aux, d0 statistics
        parallel = not _bn_training(sep)               # train-mode BatchNorm updates its buffers per utterance, in order

        # Speaker embeddings of the reference utterances.  Rows of different utterances meet nowhere in the speaker branch but in a
        # train-mode BatchNorm, so in eval mode the references of EQUAL length run through it as one batch on the caller's stream (19
        # launches for up to RAGGED_AUX_BATCH utterances instead of 19 small ones each: the per-utterance launches were 20 % of a 48-utterance
        # ragged forward, profiles/r05_ragged_ira_kernel_stats.csv); the rest -- and everything in train mode -- stays one at a time.
        emb_of, alen_of = [None] * n, [None] * n
        if parallel and E.RAGGED_AUX_BATCH > 1:
            by_len = {}
            for u in range(n):
                by_len.setdefault(int(axs[u].shape[1]), []).append(u)
            for Ta, us in by_len.items():
                for i in range(0, len(us), E.RAGGED_AUX_BATCH):
                    grp = us[i: i + E.RAGGED_AUX_BATCH]
                    if len(grp) < 2:
                        continue
                    geo_a = E.make_geometry(len(grp), Ta, N, cfg["chunk_length"], cfg["hop_length"], ks, stride)
                    alen = torch.cat([E._aux_len_tensor(aux_lens[u], 1, dev) for u in grp])
                    A = torch.cat([axs[u] for u in grp])
                    st_a = torch.zeros(len(grp), 2, device=dev, dtype=torch.float64)
                    sc_a, sh_a = E.bottleneck_affine(st, A, enc_w, se[0], geo_a, st_a)
                    A0 = E.bottleneck_frames(st, A, enc_w, sc_a, sh_a, se[1], geo_a, scale_slot=E.SLOT_AUX_BOTTLENECK).view(len(grp) * geo_a.L, E.F_BUILT)
                    v = E.speaker_embedding(st, sep, A0, len(grp), geo_a.L, alen, ks)
                    for j, u in enumerate(grp):
                        emb_of[u], alen_of[u] = v[j: j + 1], alen[j: j + 1]

        def first(sg, u):
            if emb_of[u] is not None:
                v0, alen = emb_of[u], alen_of[u]
            else:
                geo_a = E.make_geometry(1, axs[u].shape[1], N, cfg["chunk_length"], cfg["hop_length"], ks, stride)
                alen = E._aux_len_tensor(aux_lens[u], 1, dev)
                sc_a, sh_a = E.bottleneck_affine(sg, axs[u], enc_w, se[0], geo_a, stats[u, 1])
                A0 = E.bottleneck_frames(sg, axs[u], enc_w, sc_a, sh_a, se[1], geo_a, scale_slot=E.SLOT_AUX_BOTTLENECK).view(geo_a.L, E.F_BUILT)
                v0 = E.speaker_embedding(sg, sep, A0, 1, geo_a.L, alen, ks)
            mul, add, row_bias = E._fusion_operands(sg, sep, cfg, v0)
            sc, sh = E.bottleneck_affine(sg, ws[u], enc_w, sep.bottleneck[0], geos[u], stats[u, 0], mul, add)
            row_scale = E._att_row_scale(sg, sep, cfg, ws[u], enc_w, geos[u], stats[u, 0], mul) if cfg["fusion_type"] == "att" else None
            Y = E.bottleneck_frames(sg, ws[u], enc_w, sc, sh, sep.bottleneck[1], geos[u], row_bias, row_scale)
            return v0, Y, sc, sh, alen

        _mark("reference embeddings queued")
        fan = _Fan(st, n, parallel)
        r1 = [fan.run(u, lambda sg, u=u: first(sg, u)) for u in range(n)]
        fan.join()
        _mark("prologue queued")
        v0s, Ys = [r[0] for r in r1], [r[1] for r in r1]
        scs, alens = [(r[2], r[3]) for r in r1], [r[4] for r in r1]
        v_last = v0s
        if not ira:
            taps, _ = dprnn_process_ragged(st, sep, Ys, ws, enc_w, dec_w, geos, relu_act)
        else:
            taps, d0s = dprnn_process_ragged(st, sep, Ys, ws, enc_w, dec_w, geos, relu_act, want_masked=True)
            _mark("pass 1 + head queued")
            g0, b0, _ = E._norm_params(se[0])

            # The second speaker pass (dprnn_spe_ira.py:79-92) reads the first estimates d0_u, whose lengths differ.  Everything in it but the
            # GroupNorm statistics, the max-pools and the time mean is row-wise, so in eval mode the utterances' frames go through the three
            # ResBlocks and the last conv back to back, RAGGED_AUX_BATCH utterances at a time: every utterance's segment is padded (zero rows)
            # to a multiple of 27 frames, so that none of the three MaxPool1d(3) stages pools across a boundary and the rows MaxPool1d would drop
            # (a partial triple at the end, at every stage) land behind the utterance's real outputs, where the time mean does not look.
            # Per utterance: the statistics and the first conv (its GroupNorm affine is the utterance's own) in front, the time mean behind.
            v1_of = [None] * n
            if parallel and E.RAGGED_AUX_BATCH > 1 and n > 1:
                # groups of up to RAGGED_AUX_BATCH utterances, each on its own side stream from its first launch to its last (as many groups
                # as there are side streams when the batch allows: the per-utterance launches in front and behind are latency, not work)
                gsz = min(E.RAGGED_AUX_BATCH, max(8, -(-n // max(1, E.RAGGED_STREAMS))))

                def group(sg, grp):
                    lp = [-(-geos[u].L // 27) * 27 for u in grp]
                    off = np.concatenate([[0], np.cumsum(lp)]).astype(np.int64)
                    A0d_all = torch.zeros(int(off[-1]), E.F_BUILT, device=dev, dtype=torch.float32)
                    for j, u in enumerate(grp):
                        geo = geos[u]
                        sg.call("dprnn_rows_stats", E._p(d0s[u]), E._p(stats[u, 2]), 1, geo.L, N)
                        E._count_scale(stats[u, 2].view(1, 2), se[0], sg, 1)
                        sc_d = torch.empty(1, N, device=dev, dtype=torch.float32)
                        sh_d = torch.empty_like(sc_d)
                        sg.call("dprnn_bottleneck_prep", E._p(stats[u, 2]), E._p(E._param(g0, "spk_encoder.0.weight", dev)),
                                E._p(E._param(b0, "spk_encoder.0.bias", dev)), 1e-5, 0, 0, E._p(sc_d), E._p(sh_d), 1, geo.L, N)
                        E._pointwise(sg, d0s[u].view(geo.L, N), geo.L, N, se[1].weight, se[1].bias, 128, in_scale=sc_d, in_shift=sh_d,
                                     aff_stride=N, rows_per_utt=geo.L, out=A0d_all[int(off[j]): int(off[j]) + geo.L])
                    t = int(off[-1])
                    h, t = E._resblock(sg, se[2], A0d_all, 1, t, 128, 128)
                    h, t = E._resblock(sg, se[3], h, 1, t, 128, 256)
                    h, t = E._resblock(sg, se[4], h, 1, t, 256, 256)
                    Eall = E._pointwise(sg, h.view(t, 256), t, 256, se[5].weight, se[5].bias, 128)
                    embs = torch.empty(len(grp), 128, device=dev, dtype=torch.float32)
                    for j, u in enumerate(grp):
                        t3 = geos[u].L // 27                      # = ((L // 3) // 3) // 3: the rows MaxPool1d(3) x 3 keeps
                        if t3 < 1:
                            raise DprnnError("utterance too short for three MaxPool1d(3) stages")
                        sg.call("dprnn_time_mean_fwd", E._p(Eall[int(off[j]) // 27:]), E._p(alens[u]), 1, E._p(embs[j:]), 1, t3, 128, ks)
                    return embs

                groups = [list(range(i0, min(n, i0 + gsz))) for i0 in range(0, n, gsz)]
                fan = _Fan(st, len(groups))
                outs = [fan.run(gi, lambda sg, grp=grp: group(sg, grp)) for gi, grp in enumerate(groups)]
                fan.join()
                for grp, embs in zip(groups, outs):
                    for j, u in enumerate(grp):
                        v1_of[u] = embs[j: j + 1]

            def second(sg, u):
                geo = geos[u]
                if v1_of[u] is not None:
                    v1 = v1_of[u]
                else:
                    sg.call("dprnn_rows_stats", E._p(d0s[u]), E._p(stats[u, 2]), 1, geo.L, N)
                    E._count_scale(stats[u, 2].view(1, 2), se[0], sg, 1)
                    sc_d = torch.empty(1, N, device=dev, dtype=torch.float32)
                    sh_d = torch.empty_like(sc_d)
                    sg.call("dprnn_bottleneck_prep", E._p(stats[u, 2]), E._p(E._param(g0, "spk_encoder.0.weight", dev)),
                            E._p(E._param(b0, "spk_encoder.0.bias", dev)), 1e-5, 0, 0, E._p(sc_d), E._p(sh_d), 1, geo.L, N)
                    A0d = E._pointwise(sg, d0s[u].view(geo.L, N), geo.L, N, se[1].weight, se[1].bias, 128, in_scale=sc_d, in_shift=sh_d,
                                       aff_stride=N, rows_per_utt=geo.L)
                    v1 = E.speaker_embedding(sg, sep, A0d, 1, geo.L, alens[u], ks)
                v1 = E._linear_small(sg, sep.aux_linear.weight, sep.aux_linear.bias, torch.cat((v0s[u], v1), dim=1).contiguous())
                mul2, add2, row_bias2 = E._fusion_operands(sg, sep, cfg, v1)
                sc, sh = scs[u]
                if mul2 is not None or add2 is not None:
                    g, b_, eps = E._norm_params(sep.bottleneck[0])
                    sg.call("dprnn_bottleneck_prep", E._p(stats[u, 0]), E._p(g), E._p(b_), eps, E._p(mul2), E._p(add2), E._p(sc), E._p(sh), 1, geo.L, N)
                row_scale2 = E._att_row_scale(sg, sep, cfg, ws[u], enc_w, geo, stats[u, 0], mul2) if cfg["fusion_type"] == "att" else None
                return v1, E.bottleneck_frames(sg, ws[u], enc_w, sc, sh, sep.bottleneck[1], geo, row_bias2, row_scale2)

            _mark("second speaker pass queued")
            fan = _Fan(st, n, parallel)
            r2 = [fan.run(u, lambda sg, u=u: second(sg, u)) for u in range(n)]
            fan.join()
            _mark("second bottleneck queued")
            del Ys, d0s
            taps, _ = dprnn_process_ragged(st, sep, [r[1] for r in r2], ws, enc_w, dec_w, geos, relu_act)
            v_last = [r[0] for r in r2]

        def last(sg, u):
            logits = E._linear_small(sg, sep.pred_linear.weight, sep.pred_linear.bias, v_last[u])
            return E.decode(sg, taps[u], geos[u], 1), logits

        _mark("pass 2 + head queued")
        fan = _Fan(st, n)
        r3 = [fan.run(u, lambda sg, u=u: last(sg, u)) for u in range(n)]
        fan.join()
        E._scale_end(st, rec)
        _mark("everything queued")
        return [(est[0, 0, :], logits[0]) for est, logits in r3]
