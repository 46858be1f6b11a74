"""Debug aid: run fwd_train + bwd of one LSTM path on the library named by DPRNN_LIB and dump tape / H / dX / dG to an
npz, so two builds can be compared piece by piece (python tools/ab_recur_train_dump.py out.npz)."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tss_with_dprnn_amd import engine

DEV = "cuda:0"
B, S, K = 3, 9, 23
rng = np.random.default_rng(5)
st = engine.Stage(torch.device(DEV))
lib = st.lib
w = [torch.from_numpy(rng.standard_normal(s).astype(np.float32) * 0.08).to(DEV)
     for s in [(512, 128), (512, 128), (512,), (512,)] * 2]
X = torch.from_numpy(rng.standard_normal((B, S, K, 128)).astype(np.float32)).to(DEV)
dH = torch.from_numpy(rng.standard_normal((B, S, K, 256)).astype(np.float32)).to(DEV)
p = lambda t: t.data_ptr()
packed = torch.empty(2 * engine.PACK_PER_DIR_MAX, device=DEV)
bsum = torch.empty(2 * 512, device=DEV)
packedT = torch.empty(2 * 512 * 256, device=DEV)
rows = B * S * K
addr = (B * S, K, B * S, 0, K, 1)
tape = torch.zeros(lib.dprnn_lstm_tape_floats(addr[0], addr[1], 2), device=DEV)
H = torch.zeros((rows, 256), device=DEV)
dX = torch.zeros((2, rows, 128), device=DEV)
dG = torch.zeros((rows, 1024), device=DEV)
st.call("dprnn_rnn_pack", 0, *[p(t) for t in w], p(packed), p(bsum))
st.call("dprnn_lstm_pack_bwd", p(w[0]), p(w[1]), p(w[4]), p(w[5]), p(packedT))
st.call("dprnn_lstm_fwd_train", p(X), 0, 0, 0, 0, 0.0, 1, 1, 0, p(packed), p(bsum), p(H), p(tape), *addr, 2)
st.call("dprnn_lstm_bwd", p(dH), p(tape), p(packedT), p(dX), p(dG), rows, *addr, 2)
torch.cuda.synchronize()
np.savez(sys.argv[1], tape=tape.cpu().numpy(), H=H.cpu().numpy(), dX=dX.cpu().numpy(), dG=dG.cpu().numpy())
