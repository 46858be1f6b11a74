"""Evaluation loop of the reference on the MI355X path (BASELINE.json configs 2, 3, 5).

Mirror of ``Inferencer.run`` / ``InferencerSpe.run`` (src/inferencers/inferencer.py:48-78, inferencer_spe.py:20-54): for
every test item separate, reorder the estimates to the sources with the PIT loss (inferencer.py:30,60) and report SI-SDR
and SI-SDR improvement over the mixture (``final_metrics.json`` keys ``si_sdr`` / ``si_sdr_imp``, inferencer.py:88-104).
Differences, all on purpose:

* the B = 1 loop becomes length-bucketed batches on a pool of HIP streams (inference.separate_many); utterances are
  never zero-padded (that would change the per-utterance GroupNorm statistics),
* with more than one rank (one process per GPU) the items are dealt by length (sharding.shard_by_length) and only the
  per-utterance scalars are gathered -- no collective on the data path,
* SI-SDR is computed on the device in fp64 (dprnn_pairwise_sisdr_metric: the pb_bss_eval form asteroid's get_metrics
  reports).  STOI and PESQ are third-party CPU metrics outside the hot path and are not computed here.
"""
from __future__ import annotations

import csv
import json
import os
import time
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from . import sharding
from ._lib import DprnnError, check, load_library
from .inference import pit_si_sdr, ragged_chunks, separate_many, separate_many_async


def si_sdr_metric(est: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
    """est, tgt [B, n, T] float32 on the device -> [B, n] float64: SI-SDR of estimate i against target i in the
    evaluation-metric form (no mean removal, no eps)."""
    est, tgt = est.contiguous().float(), tgt.contiguous().float()
    if not (est.is_cuda and tgt.is_cuda) or est.shape != tgt.shape:
        raise DprnnError("si_sdr_metric expects equal-shape device tensors")
    B, n, T = est.shape
    out = torch.empty(B, n, n, device=est.device, dtype=torch.float64)
    lib = load_library()
    with torch.cuda.device(est.device):
        rc = lib.dprnn_pairwise_sisdr_metric(est.data_ptr(), tgt.data_ptr(), out.data_ptr(), B, n, n, T,
                                             torch.cuda.current_stream(est.device).cuda_stream)
    check(lib, rc, "dprnn_pairwise_sisdr_metric")
    return torch.diagonal(out, dim1=1, dim2=2)


def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


class BatchedInferencer:
    """``run(test_set)`` for blind separation items ``(mix [T], sources [n_src, T])``; ``run_spe(test_set)`` for target
    speaker extraction items ``(mix [T], target [T], reference [T_ref], spk_idx)``.  Returns the final-results dict and,
    on rank 0, writes ``all_metrics.csv`` / ``final_metrics.json`` into ``test_savedir`` like the reference."""

    def __init__(self, model, device, test_savedir: Optional[str] = None, logger=None, checkpoint_path: Optional[str] = None,
                 max_streams: int = 8):
        self.device = torch.device(device)
        self.model = model.to(self.device)
        if checkpoint_path is not None:                          # inferencer.py:32-43
            self.model.load_state_dict(torch.load(checkpoint_path, map_location="cpu")["model"])
            self.model = self.model.to(self.device)
        self.test_savedir, self.logger, self.max_streams = test_savedir, logger, max_streams
        self.rank, self.world = _world()

    def _my_items(self, lengths: Sequence[int]) -> List[int]:
        return sharding.shard_by_length(list(lengths), self.world)[self.rank]

    def _gather(self, idx: List[int], rows: torch.Tensor, n_total: int) -> torch.Tensor:
        """rows [len(idx), C] float64 for the items this rank owns -> [n_total, C] on every rank."""
        full = torch.zeros(n_total, rows.shape[1], dtype=torch.float64, device=rows.device)
        if idx:
            full[torch.tensor(idx, device=rows.device)] = rows
        if self.world > 1:
            dist.all_reduce(full)                                 # disjoint supports: the sum is the gather
        return full

    def _pipelined(self, mixes, extra=None):
        """separate_many over buckets as large as memory allows (inference.ragged_chunks; longest first: a ragged batch then holds utterances of similar
        length), two buckets in flight: while the device works on bucket k + 1 the host collects bucket k (its saturation check)
        and the caller's metric kernels queue up behind it -- no host stall between buckets."""
        order = sorted(range(len(mixes)), key=lambda j: -int(mixes[j].numel()))
        outs = [None] * len(mixes)
        pending = None
        for idx in ragged_chunks(order, [int(m.numel()) for m in mixes], self.device):
            ticket = separate_many_async(self.model, [mixes[j] for j in idx], self.max_streams,
                                         extra_inputs=None if extra is None else [extra[j] for j in idx])
            if pending is not None:
                for j, o in zip(pending[0], pending[1].result()):
                    outs[j] = o
            pending = (idx, ticket)
        if pending is not None:
            for j, o in zip(pending[0], pending[1].result()):
                outs[j] = o
        return outs

    def run(self, test_set) -> Dict[str, float]:
        start = time.time()
        n = len(test_set)
        items = [test_set[i] for i in range(n)]
        mine = self._my_items([int(m.shape[-1]) for m, _ in items])
        mixes = [items[i][0].to(self.device).float().reshape(-1) for i in mine]
        with torch.no_grad():
            outs = self._pipelined(mixes)
            rows = []
            for i, mix, out in zip(mine, mixes, outs):
                src = items[i][1].to(self.device).float()
                src = src[:, : out.shape[-1]].unsqueeze(0)
                est = out[:, : src.shape[-1]].unsqueeze(0)
                _, _, reordered = pit_si_sdr(est, src)             # source reordering, inferencer.py:60
                sd = si_sdr_metric(reordered, src).mean()
                inp = si_sdr_metric(mix[: src.shape[-1]].expand_as(src[0]).unsqueeze(0), src).mean()
                rows.append(torch.stack([sd, inp]))
        rows_t = torch.stack(rows) if rows else torch.zeros(0, 2, dtype=torch.float64, device=self.device)
        return self._finish(self._gather(mine, rows_t, n), start)

    def run_spe(self, test_set) -> Dict[str, float]:
        start = time.time()
        n = len(test_set)
        items = [test_set[i] for i in range(n)]
        mine = self._my_items([int(it[0].shape[-1]) for it in items])
        mixes = [items[i][0].to(self.device).float().reshape(-1) for i in mine]
        extra = [(items[i][2].to(self.device).float().reshape(-1), float(items[i][2].shape[-1])) for i in mine]
        with torch.no_grad():
            outs = self._pipelined(mixes, extra)
            rows = []
            for i, mix, (est, _) in zip(mine, mixes, outs):
                tgt = items[i][1].to(self.device).float().reshape(1, 1, -1)
                T = min(tgt.shape[-1], est.shape[-1])
                sd = si_sdr_metric(est.reshape(1, 1, -1)[..., :T], tgt[..., :T]).mean()
                inp = si_sdr_metric(mix.reshape(1, 1, -1)[..., :T], tgt[..., :T]).mean()
                rows.append(torch.stack([sd, inp]))
        rows_t = torch.stack(rows) if rows else torch.zeros(0, 2, dtype=torch.float64, device=self.device)
        return self._finish(self._gather(mine, rows_t, n), start)

    def _finish(self, table: torch.Tensor, start: float) -> Dict[str, float]:
        t = table.cpu()
        final = {"si_sdr": float(t[:, 0].mean()), "si_sdr_imp": float((t[:, 0] - t[:, 1]).mean())}    # inferencer.py:88-95
        if self.logger is not None and self.rank == 0:
            self.logger.info("Finished *** <Total time:{:.3f} min>.".format((time.time() - start) / 60))
            self.logger.info("Overall metrics :")
            self.logger.info(final)
        if self.test_savedir is not None and self.rank == 0:
            with open(os.path.join(self.test_savedir, "all_metrics.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["", "si_sdr", "input_si_sdr"])
                for i, (a, b) in enumerate(t.tolist()):
                    w.writerow([i, a, b])
            with open(os.path.join(self.test_savedir, "final_metrics.json"), "w") as f:
                json.dump(final, f, indent=0)
        return final
