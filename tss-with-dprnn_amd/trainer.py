"""Data-parallel training step loop for the MI355X path (SURVEY.md section 8e "Training (config 4)", 8f rank 2).

Mirror of the reference's ``Trainer`` step loop (src/trainers/trainer.py:86-133 train, :135-178 eval, :180-226 run,
:294-306 checkpoints) with the one addition BASELINE.json's north_star asks for: gradients are averaged over the ranks
with ONE all-reduce of a flat fp32 buffer (3 651 073 elements = 14.6 MB for the shipped model) between ``backward()``
and ``clip_grad_norm_``.  One process per GPU, ``torch.distributed`` backend "nccl" (= RCCL over xGMI) in production,
"gloo" in the CPU tests; no other collective is on the path (the logged loss is all-reduced once per epoch).

The model forward/backward and the loss are the HIP kernels of training.py; optimizer, LR scheduler and gradient
clipping are the stock torch objects the reference configures (scripts/train/config_bss.yaml:30-38), so checkpoints
keep the reference format ``{'epoch', 'optimizer', 'model'}`` and load in either code base.
"""
from __future__ import annotations

import os
import time
from collections import deque
from typing import Iterable, Optional

# dmabuf IPC for RCCL on this driver: must be in the environment before the HIP runtime comes up (see tss_with_dprnn_amd.py)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from .training import FusedAdam, PITSiSdrLoss


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def init_process_group(backend: str = "nccl", device: Optional[torch.device] = None, **kwargs) -> None:
    """``torch.distributed.init_process_group`` for one process per GPU (RANK / WORLD_SIZE / MASTER_* from the launcher, use
    127.0.0.1 as the rendezvous address on one node).  backend "nccl" is RCCL over xGMI; it needs HSA_ENABLE_IPC_MODE_LEGACY=0
    in the environment before the HIP runtime starts, which importing this package arranges -- a process that initialised the
    GPU earlier with another value is told so here instead of failing inside RCCL with `hipIpcGetMemHandle: invalid argument`."""
    if backend == "nccl":
        if os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
            raise RuntimeError("RCCL on this platform needs HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC); the environment has "
                               f"{os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')!r}")
        if device is None:
            device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(device)
        dist.init_process_group(backend="nccl", device_id=device, **kwargs)
    else:
        dist.init_process_group(backend=backend, **kwargs)


class FlatGradAllReducer:
    """Average ``param.grad`` over the ranks with a single all-reduce.

    The reference's loop calls ``optimizer.zero_grad()`` every step, which drops the grad tensors (set_to_none), so the
    flat bucket is refilled from whatever grads exist after ``backward()``: one gather copy, one collective, one
    scatter copy of 14.6 MB -- against ~0.5 s of backward kernels."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        self.flat: Optional[torch.Tensor] = None
        self.timing = False              # bench.py --mode train: HIP events around every average() on the current stream
        self.events = []

    def average(self) -> None:
        world = _world()
        if world == 1:
            return
        if self.timing and self.params[0].is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._average(world)
            e1.record()
            self.events.append((e0, e1))
        else:
            self._average(world)

    def _average(self, world: int) -> None:
        p0 = self.params[0]
        if self.flat is None or self.flat.device != p0.device:
            self.flat = torch.empty(self.numel, device=p0.device, dtype=torch.float32)
        views, off = [], 0
        for p in self.params:
            views.append(self.flat[off: off + p.numel()].view_as(p))
            off += p.numel()
        for p, v in zip(self.params, views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.mul_(1.0 / world)
        for p, v in zip(self.params, views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)


def broadcast_parameters(model: torch.nn.Module, src: int = 0) -> None:
    """Every rank starts from rank `src`'s weights and buffers (the reference builds the model from the hydra config on
    one process; replicas must agree before the first averaged step)."""
    if _world() == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src)


def _micro_batches(model, mix) -> int:
    """Utterances one forward / backward may process without re-running anything (training.SAVE_BUDGET_BYTES)."""
    from . import engine, training
    if not mix.is_cuda or not hasattr(model, "cfg") or not hasattr(model, "separation"):
        return mix.shape[0]
    cfg = model.cfg
    geo = engine.make_geometry(1, mix.shape[1], cfg["input_size"], cfg["chunk_length"], cfg["hop_length"],
                               cfg["kernel_size"], cfg["stride"])
    ndirs = [2 if r.rnn.bidirectional else 1 for r, _, _, _ in training._paths(model.separation)]
    return max(1, training.save_budget_bytes(mix.device) // training.saved_bytes_per_utt(geo, len(ndirs), ndirs))


def train_step(model, loss_module, optimizer, mix, target, clip_norm: float = 0.0,
               reducer: Optional[FlatGradAllReducer] = None):
    """One iteration of Trainer.train (trainer.py:98-118).  Returns (loss tensor, reordered estimates).

    The loss of the reference (PIT SI-SDR, mean over the batch) is a sum over utterances, so a batch whose activations
    exceed the budget is processed as balanced micro-batches -- forward, loss and backward each, gradients accumulating
    in ``param.grad`` with weights n_i / B -- instead of one forward whose backward has to re-run most of it (config 4,
    64 utterances per GPU: 2 x 32)."""
    optimizer.zero_grad()
    B = mix.shape[0]
    mb_max = _micro_batches(model, mix)
    if B > mb_max:
        n_mb = -(-B // mb_max)
        cuts = [round(i * B / n_mb) for i in range(n_mb + 1)]
        loss, ests = None, []
        for b0, b1 in zip(cuts[:-1], cuts[1:]):
            l_i, est_i = loss_module(model(mix[b0:b1]), target[b0:b1], return_est=True)
            w = (b1 - b0) / B
            (l_i * w).backward()
            loss = l_i.detach() * w if loss is None else loss + l_i.detach() * w
            ests.append(est_i.detach())
        est = torch.cat(ests)
    else:
        out = model(mix)
        loss, est = loss_module(out, target, return_est=True)
        loss.backward()
    if reducer is not None:
        reducer.average()
    if isinstance(optimizer, FusedAdam):
        optimizer.step(clip_norm=clip_norm)              # clip + Adam in two launches
    else:
        if clip_norm:
            torch.nn.utils.clip_grad_norm_(model.parameters(), clip_norm)
        optimizer.step()
    return loss.detach(), est.detach()


def train_step_spe(model, loss_module, optimizer, mix, target, reference, spk_idx, ce_gamma: float, clip_norm: float = 0.0,
                   reducer: Optional[FlatGradAllReducer] = None):
    """One iteration of TrainerSpe.train (trainer_spe.py:25-56): SI-SDR of the extracted speaker + ce_gamma * speaker CE."""
    optimizer.zero_grad()
    ref_len = torch.tensor(reference.shape[1], dtype=torch.float32, device=reference.device)
    est, logits = model(mix, reference, ref_len)
    l = loss_module(est.unsqueeze(1), target.unsqueeze(1))
    ce = torch.nn.functional.cross_entropy(logits, spk_idx)        # the caller-side loss of the reference, on [B, num_spks] logits
    loss = l + ce_gamma * ce
    loss.backward()
    if reducer is not None:
        reducer.average()
    if isinstance(optimizer, FusedAdam):
        optimizer.step(clip_norm=clip_norm)
    else:
        if clip_norm:
            torch.nn.utils.clip_grad_norm_(model.parameters(), clip_norm)
        optimizer.step()
    return loss.detach(), l.detach(), ce.detach()


class DistributedTrainer:
    """Step loop with the reference Trainer's semantics; ``logger`` / ``reporter`` are optional duck-typed objects
    (``logger.info(str)``, ``reporter.add_and_report(logs=dict, mode=str)``) so neither hydra nor wandb is needed."""

    def __init__(self, model, optimizer, lr_scheduler=None, clip_norm: float = 0.0, device=None, logger=None,
                 reporter=None, new_checkpoints_path: Optional[str] = None, n_checkpoints: int = 5, print_freq: int = 100,
                 cur_epoch: int = 0, checkpoint_path: Optional[str] = None, scheduler_on_eval_loss: bool = True):
        self.device = device if device is not None else next(model.parameters()).device
        self.model = model
        self.optimizer = optimizer
        self.lr_scheduler = lr_scheduler
        self.scheduler_on_eval_loss = scheduler_on_eval_loss          # ReduceLROnPlateau (trainer.py:197-198) vs ExponentialLR
        self.clip_norm = clip_norm or 0
        self.logger, self.reporter = logger, reporter
        self.loss_module = PITSiSdrLoss()
        self.cur_epoch = cur_epoch
        self.print_freq = print_freq
        self.rank = dist.get_rank() if _world() > 1 else 0
        self.new_checkpoints_path = new_checkpoints_path
        self.checkpoint_queue = deque(maxlen=n_checkpoints)
        if checkpoint_path is not None:                               # trainer.py:66-77
            cpt = torch.load(checkpoint_path, map_location="cpu")
            self.model.load_state_dict(cpt["model"])
            if "optimizer" in cpt:
                self.optimizer.load_state_dict(cpt["optimizer"])
            self.cur_epoch = cpt.get("epoch", cur_epoch)
        broadcast_parameters(self.model)
        self.reducer = FlatGradAllReducer(self.model.parameters())
        self.ce_gamma: Optional[float] = None       # set (TrainerSpe: config['ce_gamma']) to train DPRNN-Spe / IRA models

    def _info(self, msg):
        if self.logger is not None and self.rank == 0:
            self.logger.info(msg)

    def _mean_over_ranks(self, total: float, count: int) -> float:
        t = torch.tensor([total, float(count)], dtype=torch.float64, device=self.device)
        if _world() > 1:
            dist.all_reduce(t)
        return float(t[0] / max(t[1], 1.0))

    def _sync_buffers(self) -> None:
        """BatchNorm running statistics (DPRNN-Spe / IRA speaker encoder, dprnn_spe.py:20-21) are updated from each rank's
        own shard; before they are READ (eval, checkpoint) every rank takes rank 0's, as DistributedDataParallel's default
        broadcast_buffers does, so all ranks evaluate the same model and the checkpoint matches what was evaluated."""
        if _world() > 1:
            for b in self.model.buffers():
                dist.broadcast(b.data, 0)

    def _check_equal_steps(self, dataloader) -> None:
        """Every step ends in a collective: ranks whose loaders yield different step counts would deadlock in it."""
        if _world() == 1 or not hasattr(dataloader, "__len__"):
            return
        n = len(dataloader)
        t = torch.tensor([n, -n], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if int(t[0]) != -int(t[1]):
            raise RuntimeError(f"rank {self.rank}: dataloader yields {n} steps but the ranks disagree (min {-int(t[1])}, max "
                               f"{int(t[0])}); use a DistributedSampler with drop_last=True so every rank runs the same "
                               "number of gradient all-reduces")

    def train(self, dataloader) -> float:
        self.model.train()
        self._check_equal_steps(dataloader)
        total, steps = 0.0, 0
        start = time.time()
        for step, batch in enumerate(dataloader):
            if self.ce_gamma is None:                                 # Trainer.train: (mix, target)
                mix, target = batch
                loss, _ = train_step(self.model, self.loss_module, self.optimizer, mix.to(self.device), target.to(self.device),
                                     self.clip_norm, self.reducer)
            else:                                                     # TrainerSpe.train: (mix, target, reference, spk_idx)
                mix, target, reference, spk_idx = batch
                loss, _, _ = train_step_spe(self.model, self.loss_module, self.optimizer, mix.to(self.device),
                                            target.to(self.device), reference.to(self.device), spk_idx.to(self.device),
                                            self.ce_gamma, self.clip_norm, self.reducer)
            total += loss.item()
            steps += 1
            if step % self.print_freq == 0:
                self._info("<epoch:{:d}, iter:{:d}, lr:{:.3e}, loss:{:.3f}>.".format(
                    self.cur_epoch, step, self.optimizer.param_groups[0]["lr"], -total / (step + 1)))
        return self._log_epoch(total, steps, start, "train")

    def eval(self, dataloader) -> float:
        self.model.eval()
        self._sync_buffers()
        total, steps = 0.0, 0
        start = time.time()
        with torch.no_grad():
            for batch in dataloader:
                if self.ce_gamma is None:
                    mix, target = batch
                    out = self.model(mix.to(self.device))
                    loss = self.loss_module(out, target.to(self.device))
                else:                                                 # trainer_spe.py:88-97: SI-SDR only
                    mix, target, reference = batch[0].to(self.device), batch[1].to(self.device), batch[2].to(self.device)
                    ref_len = torch.tensor(reference.shape[1], dtype=torch.float32, device=self.device)
                    est, _ = self.model(mix, reference, ref_len)
                    loss = self.loss_module(est.unsqueeze(1), target.unsqueeze(1))
                total += loss.item()
                steps += 1
        return self._log_epoch(total, steps, start, "eval")

    def _log_epoch(self, total, steps, start, mode) -> float:
        mean = self._mean_over_ranks(total, steps)                    # the all-reduced mean the scheduler must see
        if self.reporter is not None and self.rank == 0:
            self.reporter.add_and_report(logs={"step": self.cur_epoch, "loss": -mean, "metrics": None}, mode=mode)
        self._info("Finished *** <epoch:{:d}, iter:{:d}, lr:{:.3e}, loss:{:.3f}, Total time:{:.3f} min>.".format(
            self.cur_epoch, steps, self.optimizer.param_groups[0]["lr"], -mean, (time.time() - start) / 60))
        return mean

    def run(self, train_loader, eval_loader, n_epochs: int, early_stop: int):
        best_loss, no_improve = 100500, 0                             # trainer.py:181
        history = []
        while self.cur_epoch < n_epochs:
            self.cur_epoch += 1
            train_loss = self.train(train_loader)
            eval_loss = self.eval(eval_loader)
            history.append((train_loss, eval_loss))
            if self.lr_scheduler is not None:
                if self.scheduler_on_eval_loss:
                    self.lr_scheduler.step(eval_loss)
                else:
                    self.lr_scheduler.step()
            if eval_loss >= best_loss:
                no_improve += 1
            else:
                best_loss, no_improve = eval_loss, 0
                self._save_checkpoint(best=True)
            if no_improve == early_stop:
                break
        self._save_checkpoint(best=False)
        return history

    def _save_checkpoint(self, best: bool = False) -> Optional[str]:
        self._sync_buffers()            # called by every rank (run() decides on the all-reduced eval loss)
        if self.new_checkpoints_path is None or self.rank != 0:
            return None
        cpt = {"epoch": self.cur_epoch, "optimizer": self.optimizer.state_dict(), "model": self.model.state_dict()}
        path = os.path.join(self.new_checkpoints_path, "{0}_{1}.pt".format(str(self.cur_epoch), "best" if best else "last"))
        torch.save(cpt, path)
        if len(self.checkpoint_queue) == self.checkpoint_queue.maxlen:
            old = self.checkpoint_queue[0]
            if os.path.exists(old) and old != path:
                os.remove(old)
        self.checkpoint_queue.append(path)
        return path
