"""depthcore -- host side of libdepthcore.so (hand-written HIP for MI355X / gfx950).

`ops` wraps the C ABI (include/depthcore.h) as torch.autograd.Functions; the
reference-shaped facade lives one level up (`layers.py`, `networks/`, `trainer.py`).
"""
from . import _lib  # noqa: F401
