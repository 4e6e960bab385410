#!/usr/bin/env python3
"""tools/f32_paths_error.py -- max-normalised error of each fp32 3x3 path (SRGEN_F32_CONV3X3 = wino | direct | split)
against the reference's fp32 output (tests/golden/generator_fp32.npz) and each tap; run once per path (the switch is
read when the handle is created)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from superresolution_amd import Generator, weights as W
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "generator_fp32.npz"))
m = Generator(); m.load_state_dict(W.synthetic_state_dict(0)); m = m.cuda().eval()
x = torch.from_numpy(np.concatenate([W.synthetic_lr_batch(1, 0, "dense"), W.synthetic_lr_batch(1, 0, "sparse")], 0)).cuda()
with torch.no_grad():
    y = m(x)
ref = g["y_dense_sparse"]
err = float(np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max())
print(json.dumps({"path": os.environ.get("SRGEN_F32_CONV3X3", "split (default)"), "max_rel_err_vs_reference_fp32": err}))
