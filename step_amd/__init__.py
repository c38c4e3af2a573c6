"""step_amd - MI355X-native (gfx950) ST-GCN block stack for UttaranB127/STEP.

Drop-in module contract of the reference's ``net`` packages:
    step_amd.net.classifier.Classifier                (classifier_stgcn_real_only)
    step_amd.net.classifier_real_and_synth.Classifier (classifier_stgcn_real_and_synth)
    step_amd.net.classifier_hybrid.Classifier         (classifier_hybrid)
    step_amd.net.CVAE_stgcn.CVAE / Encoder / Decoder  (generator_cvae)
All arithmetic of the block stack runs in hand-written HIP kernels (libstep_hip.so, C ABI
in include/step_hip.h).  There is no CPU fallback: CPU tensors raise.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
