"""Test infrastructure: the tape of discrete decisions a depthcore forward takes (depthcore.ops decision observers).

While active, every fused ReLU records its output (`("relu", y)`: the decision taken is y > 0; ReLUs folded into a convolution's
loader are materialised for the tape only) and the max-pool its argmax codes (`("maxpool", code)`), in call order.  The parity
tests replay these decisions inside the fp64 oracle (oracle/kinks.py) so that both sides evaluate the same smooth function, and
report how many of the imposed decisions differ from the oracle's own.  Records references, copies nothing."""
from depthcore import ops


class KinkTape:
    def __init__(self):
        self.entries = []

    def _observe(self, kind, t):
        self.entries.append((kind, t))

    def __enter__(self):
        ops.add_decision_observer(self._observe)
        return self

    def __exit__(self, *exc):
        ops.remove_decision_observer(self._observe)
        return False
