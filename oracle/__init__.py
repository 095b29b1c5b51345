"""oracle/ -- TEST INFRASTRUCTURE ONLY.

A CPU (fp32, plain torch / numpy) restatement of the reference's per-batch hot
path.  It exists to *check* the HIP path; nothing in the
product package (`self-supervised-depth-estimation_amd/`) imports it.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import from here.

Parity status:
  * geometry / photometric loss / decoder / pose decoder: PINNED -- checked
    against golden vectors generated in the build container by importing the
    reference's own `layers.py`, `networks/depth_decoder.py`,
    `networks/pose_decoder.py` and the unbound `Trainer.generate_images_pred`
    / `compute_losses` / `compute_reprojection_loss` methods
    (`tests/golden/make_golden.py`, fixtures in `tests/golden/*.npz`).
  * ResNet encoder: PARITY UNPINNED -- the reference delegates to torchvision
    (unpinned, not installed here; `networks/resnet_encoder.py:13`), so the
    restatement in `resnet_ref.py` follows torchvision's published ResNet v1.5
    layout and is pinned only by state_dict key/shape equality and the
    analytic parameter count (11,689,512 for resnet18).
"""
