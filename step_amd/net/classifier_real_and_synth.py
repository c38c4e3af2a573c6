"""Classifier of classifier_stgcn_real_and_synth (net/classifier.py:9-113 there): block 0 has
no residual (:41), block 2 has stride 2 (:48), the output is raw logits (:87)."""
from .classifier import Classifier as _Base
from .blocks import st_gcn  # noqa: F401


class Classifier(_Base):
    BLOCKS = ((32, 1, {"residual": False}), (64, 2, {}), (64, 2, {}))
    RETURNS_SOFTMAX = False
