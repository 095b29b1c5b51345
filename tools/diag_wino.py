#!/usr/bin/env python3
"""Phase cycle breakdown of wino_ps_kernel from a -DWINO_DIAG build (tools/diag_wino.sh builds it)."""
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "self-supervised-depth-estimation_amd"))
from depthcore import _lib  # noqa: E402
from depthcore.ops import ptr  # noqa: E402


def main():
    L = _lib.lib()
    raw = ctypes.CDLL(os.environ["DEPTHCORE_LIB"])
    raw.dc_wino_set_diag.argtypes = [ctypes.c_void_p]
    for spec in sys.argv[1:]:
        B, Ci, Co, H, W = (int(v) for v in spec.split(","))
        x = torch.randn(B, Ci, H, W, device="cuda")
        w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
        y = torch.empty(B, Co, H, W, device="cuda")
        ws = torch.empty(L.dc_wino3x3_workspace(B, Ci, Co, H, W), dtype=torch.uint8, device="cuda")
        diag = torch.zeros(8 * 200000, dtype=torch.int64, device="cuda")
        raw.dc_wino_set_diag(diag.data_ptr())
        st = torch.cuda.current_stream().cuda_stream
        if os.environ.get("WINO_DIAG_KERNEL") == "wgrad":          # the weight-gradient kernel instead (mode 2 fields only)
            gy = torch.randn(B, Co, H, W, device="cuda")
            dw = torch.empty_like(w)
            ws = torch.empty(L.dc_wino3x3_wgrad_workspace(B, Ci, Co, H, W), dtype=torch.uint8, device="cuda")
            for _ in range(50):
                L.dc_wino3x3_wgrad(ptr(x), ptr(gy), ptr(dw), ws.data_ptr(), B, Ci, Co, H, W, st)
        else:
            for _ in range(50):
                L.dc_wino3x3_fwd(ptr(x), ptr(w), ptr(y), ws.data_ptr(), B, Ci, Co, H, W, st)
        torch.cuda.synchronize()
        d = diag.cpu().view(-1, 8)
        d = d[d[:, 4] > 0]
        if os.environ.get("WINO_DIAG_DUMP"):
            import numpy as np
            np.save(os.path.join(os.environ["WINO_DIAG_DUMP"], "%s_blocks_%s.npy" % (os.environ.get("WINO_DIAG_KERNEL", "wino"), spec.replace(",", "_"))), d.numpy())
        span = int(d[:, 7].max() - d[:, 7].min())               # first to last block start (s_memtime ticks)
        d = d.double()
        m = d.mean(0)
        print("%s: blocks %d | cycles/block: compute %.0f commit+wait %.0f issue %.0f barrier %.0f total(loop) %.0f | frac %s | "
              "prologue %.0f epilogue %.0f (%.2f / %.2f of the block) | block starts span %d"
              % (spec, d.shape[0], m[0], m[1], m[2], m[3], m[4], ["%.2f" % (v / m[4]) for v in m[:4]], m[5], m[6],
                 m[5] / (m[4] + m[5] + m[6]), m[6] / (m[4] + m[5] + m[6]), span))


if __name__ == "__main__":
    main()
