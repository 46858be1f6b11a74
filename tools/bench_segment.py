#!/usr/bin/env python3
"""Stand-alone time of the segmentation copy at the headline size (32 utterances x 23999 frames -> 194 chunks of 250)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tss_with_dprnn_amd as pkg
from tss_with_dprnn_amd import engine

lib = pkg.load_library()
st = engine.Stage(torch.device("cuda:0"))
B, L, K, P = 32, 23999, 250, 125
S = (L + K) // P + 1
Y = torch.randn(B, L, 128, device="cuda")
X = torch.empty(B, S, K, 128, device="cuda")
for _ in range(3):
    st.call("dprnn_segment_fwd", Y.data_ptr(), X.data_ptr(), B, L, K, P, S)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    st.call("dprnn_segment_fwd", Y.data_ptr(), X.data_ptr(), B, L, K, P, S)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
gb = (Y.numel() + X.numel()) * 4 / 1e9
print(f"segment: {ms:.3f} ms per launch, {gb / ms:.2f} TB/s ({gb:.2f} GB)")
