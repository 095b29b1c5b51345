import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "self-supervised-depth-estimation_amd")
for p in (REPO, PKG, os.path.join(REPO, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle works on small tensors: a GPU box's 100+ hardware threads only add scheduling overhead
    import torch
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    # the tests judge the product library, never a tuning / ablation build (tools/build_variant.sh)
    from depthcore import _lib
    if _lib.IS_VARIANT:
        raise pytest.UsageError("DEPTHCORE_LIB=%s: the tests only run on the product libdepthcore.so" % _lib.LIB_PATH)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    d = os.path.join(REPO, "tests", "golden")
    return {name: np.load(os.path.join(d, name + ".npz"), allow_pickle=False)
            for name in ("layers_ops", "trainer_losses", "decoders", "pose_even", "fusion_v3", "trainer_ablations", "convgru",
                         "data_pillow", "fusion_v3_noattn")}
