"""Diagnostic: one gradient golden on the GPU, per-parameter error and where in the tensor it sits.
   python tools/diag_grad_case.py grad_ira_wide_cat"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from conftest import build_model, load_golden, rel_l2   # noqa: E402
from tss_with_dprnn_amd.training import PITSiSdrLoss     # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "grad_ira_wide_cat"
DEV = "cuda:0"
g = load_golden(name)
m = g["meta"]
model, _, _ = build_model(m["kind"], m["cfg"], m["seed"], DEV)
model.train()
if not m["bn_training"]:
    for mod in model.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.eval()
est, logits = model(torch.from_numpy(g["mix"]).to(DEV), torch.from_numpy(g["aux"]).to(DEV), torch.tensor(float(g["aux_len"])))
print("est rel-L2", rel_l2(est.detach().cpu().numpy(), g["out"]), "logits", rel_l2(logits.detach().cpu().numpy(), g["logits"]))
tgt = torch.from_numpy(g["tgt"]).to(DEV)
loss = PITSiSdrLoss()(est.unsqueeze(1), tgt.unsqueeze(1)) + 0.5 * torch.nn.CrossEntropyLoss()(logits, torch.from_numpy(g["spk_idx"]).to(DEV))
print("loss", loss.item(), float(g["loss"]))
loss.backward()
torch.cuda.synchronize()
rows = []
for k, prm in model.named_parameters():
    if "grad_" + k not in g:
        continue
    got = prm.grad.detach().cpu().numpy().astype(np.float64)
    d = got - g["grad_" + k]
    nrm = float(g["norm_" + k])
    e = np.linalg.norm(d) / (nrm + 1e-30)
    flat = np.abs(d).reshape(-1)
    top = np.sort(flat)[::-1][:4]
    rows.append((e, k, (top ** 2).sum() / max((flat ** 2).sum(), 1e-300), flat.size))
for e, k, frac, n in sorted(rows, reverse=True)[:14]:
    print(f"{e:9.2e}  {k:55s} n={n:7d}  share of err^2 in top-4 elements {frac:.2f}")
