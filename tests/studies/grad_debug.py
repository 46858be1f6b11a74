#!/usr/bin/env python3
"""Per-parameter gradient errors of one tests/golden/grad_*.npz case, in network order (debugging aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import build_model, load_golden
from test_gpu_train import grad_projection
from tss_with_dprnn_amd.training import PITSiSdrLoss

name = sys.argv[1]
g = load_golden(name); m = g["meta"]
model, _, _ = build_model(m["kind"], m["cfg"], m["seed"], "cuda:0")
model.train()
if m["kind"] == "bss":
    out = model(torch.from_numpy(g["mix"]).cuda())
    loss = PITSiSdrLoss()(out, torch.from_numpy(g["tgt"]).cuda())
else:
    if not m.get("bn_training", True):
        for mod in model.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.eval()
    est, logits = model(torch.from_numpy(g["mix"]).cuda(), torch.from_numpy(g["aux"]).cuda(), torch.tensor(float(g["aux_len"])))
    tgt = torch.from_numpy(g["tgt"]).cuda()
    loss = PITSiSdrLoss()(est.unsqueeze(1), tgt.unsqueeze(1)) + 0.5 * torch.nn.functional.cross_entropy(logits, torch.from_numpy(g["spk_idx"]).cuda())
loss.backward()
print("loss", loss.item(), float(g["loss"]))
for k, p in model.named_parameters():
    if "norm_" + k not in g:
        continue
    got = p.grad.cpu().numpy(); nrm = float(g["norm_" + k])
    if "grad_" + k in g:
        err = np.linalg.norm(got.astype(np.float64) - g["grad_" + k]) / nrm
    else:
        pr = grad_projection(k, got.shape).astype(np.float64) @ got.reshape(-1).astype(np.float64)
        err = np.abs(pr - g["proj_" + k]).max() / nrm
    print(f"{k:60s} {err:.2e}  norm {nrm:.3e}")

if len(sys.argv) > 2:      # kink check: sign disagreements of the PReLU input between the GPU forward and the CPU oracle
    from oracle import dprnn_torch_cpu as otc
    from tss_with_dprnn_amd.synthetic import make_state
    from tss_with_dprnn_amd import engine
    dbg = {}
    with torch.no_grad():
        engine.tasnet_forward(model, torch.from_numpy(g["mix"]).cuda(), dbg)
    nb = m["cfg"]["n_repeats"]
    xg = dbg[f"block{nb - 1}"].cpu().numpy()                     # [B,S,K,F]
    taps = {}
    otc.tasnet_forward(otc.to_torch_state(make_state("bss", m["cfg"], m["seed"])), m["cfg"], torch.from_numpy(g["mix"]), taps)
    xc = taps[f"block{nb - 1}"].numpy().transpose(0, 3, 2, 1)    # [B,N,K,S] -> [B,S,K,N]
    flips = np.argwhere((xg > 0) != (xc > 0))
    print("PReLU input: sign flips", len(flips), "min |x| cpu", np.abs(xc).min(), "rel-L2", np.linalg.norm(xg - xc) / np.linalg.norm(xc))
    for f in flips[:5]:
        print("  at", tuple(f), "gpu", xg[tuple(f)], "cpu", xc[tuple(f)])
