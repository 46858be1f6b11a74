"""fp64 SI-SDR / PIT restatement (TEST INFRASTRUCTURE, see oracle/__init__.py).

The reference takes these from asteroid==0.7.0 (requirements.txt:1), which is not
under /root/reference: ``asteroid.losses.pairwise_neg_sisdr`` and
``PITLossWrapper(pit_from='pw_mtx')`` (call sites src/trainers/trainer.py:39,104;
src/inferencers/inferencer.py:30,60).  Published algorithm restated here:
zero-mean both signals; s_target = <e,t> t / (||t||^2 + eps);
SI-SDR = 10 log10(||s_target||^2 / (||e - s_target||^2 + eps) + eps), eps = 1e-8;
PIT = best permutation of the mean pairwise SI-SDR.
PARITY UNPINNED for this file: the reference holds no test vectors for the loss
and asteroid is not installed here; it is pinned only by its own known-answer
cases in tests/test_sisdr.py.
"""
from __future__ import annotations

import itertools

import numpy as np

EPS = 1e-8


def si_sdr(est: np.ndarray, tgt: np.ndarray) -> np.ndarray:
    """[..., T] x [..., T] -> [...] dB, computed in float64."""
    e = est.astype(np.float64)
    t = tgt.astype(np.float64)
    e = e - e.mean(-1, keepdims=True)
    t = t - t.mean(-1, keepdims=True)
    dot = (e * t).sum(-1, keepdims=True)
    s = dot * t / ((t * t).sum(-1, keepdims=True) + EPS)
    n = e - s
    return 10 * np.log10((s * s).sum(-1) / ((n * n).sum(-1) + EPS) + EPS)


def pit_si_sdr(est: np.ndarray, tgt: np.ndarray):
    """est, tgt [B, n_src, T] -> (mean best SI-SDR per item [B], best permutation [B, n_src])."""
    B, n, _ = est.shape
    pw = si_sdr(est[:, :, None, :], tgt[:, None, :, :])          # [B, est, tgt]
    perms = list(itertools.permutations(range(n)))
    scores = np.stack([np.mean([pw[:, i, p[i]] for i in range(n)], axis=0) for p in perms], axis=1)
    best = scores.argmax(1)
    return scores[np.arange(B), best], np.array([perms[b] for b in best])


def pit_neg_sisdr_torch(est, tgt):
    """The training loss of the reference (trainer.py:39,104) as differentiable torch ops on CPU:
    asteroid PairwiseNegSDR('sisdr') + PITLossWrapper(pit_from='pw_mtx'), published algorithm restated --
    mean over the batch of min over permutations of mean_i -SI-SDR(est_i, tgt_perm(i)).  fp32 like the reference."""
    import torch
    e = est - est.mean(-1, keepdim=True)
    t = tgt - tgt.mean(-1, keepdim=True)
    s_t = t.unsqueeze(1)                     # [B, 1, n, T]
    s_e = e.unsqueeze(2)                     # [B, n, 1, T]
    dot = (s_e * s_t).sum(-1, keepdim=True)
    energy = (s_t ** 2).sum(-1, keepdim=True) + EPS
    proj = dot * s_t / energy
    noise = s_e - proj
    sdr = (proj ** 2).sum(-1) / ((noise ** 2).sum(-1) + EPS)
    pw = -10 * torch.log10(sdr + EPS)        # [B, est, tgt]
    n = est.shape[1]
    perms = list(itertools.permutations(range(n)))
    losses = torch.stack([torch.stack([pw[:, i, p[i]] for i in range(n)], 1).mean(1) for p in perms], 1)   # [B, n!]
    best, idx = losses.min(1)
    return best.mean(), idx


def si_sdr_eval_metric(est: np.ndarray, tgt: np.ndarray) -> np.ndarray:
    """The 'si_sdr' that asteroid.metrics.get_metrics reports (inferencer.py:64-70): it delegates to
    pb_bss_eval.evaluation.si_sdr(reference, estimation) -- neither package is installed here, published formula
    restated (PARITY UNPINNED): no mean removal, no eps; 10 log10(|a t|^2 / |e - a t|^2), a = <e,t> / |t|^2."""
    e = est.astype(np.float64)
    t = tgt.astype(np.float64)
    a = (e * t).sum(-1, keepdims=True) / (t * t).sum(-1, keepdims=True)
    proj = a * t
    noise = e - proj
    return 10 * np.log10((proj * proj).sum(-1) / (noise * noise).sum(-1))
