#!/usr/bin/env python3
"""Sum of the default-choice times of tools/sweep_wino.py's shapes (forward launch each) -- for tuning wino_ps_cost through its
experiment knobs (DC_WINO_V2_PENALTY / DC_WINO_V0_PENALTY)."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
from sweep_wino import SHAPES, timed  # noqa: E402
from depthcore import _lib  # noqa: E402
from depthcore.ops import ptr  # noqa: E402

L = _lib.lib()
tot = 0.0
for B, Ci, Co, H, W in SHAPES:
    x = torch.randn(B, Ci, H, W, device="cuda")
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
    y = torch.empty(B, Co, H, W, device="cuda")
    ws = torch.empty(L.dc_wino3x3_workspace(B, Ci, Co, H, W), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    t = timed(lambda: L.dc_wino3x3_fwd(ptr(x), ptr(w), ptr(y), ws.data_ptr(), B, Ci, Co, H, W, st))
    tot += t
print("sum of default choices: %.1f us" % tot)
