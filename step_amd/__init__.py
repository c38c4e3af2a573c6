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


def set_mfma_dtype(model, dtype):
    """Choose the MFMA operand type of the temporal convolutions of every st_gcn block of `model`:
    "f32" (default; 1e-4 parity with the reference) or "bf16" (BASELINE's bf16 training config:
    operands rounded to bf16 inside the kernels, fp32 accumulation, fp32 tensors and parameters;
    logits within 1e-2), or "f32_split": fp32 operands split into bf16 hi + lo (16 mantissa bits) and every product
    taken as three bf16 MFMA terms with fp32 accumulation - held to the fp32 bars of the parity tests at about three
    times the fp32-MFMA kernels' speed ("f32_split_wgrad": only the weight gradient that way; "f32_split_gcn": the
    graph conv of the 32 / 64-channel blocks too, step_gcn_fwd_split / step_gcn_bwd_split).  Blocks without a
    bf16 kernel (Kt != 75, 3 channels) stay fp32.
    The default "f32" is plain fp32 arithmetic with the temporal conv and both its gradients taken in the frequency
    domain (transform -> per-bin channel mixing -> inverse transform, all fp32 MFMA GEMMs: csrc/tcn_dft.hip) wherever that
    costs under 60 % of the direct kernels' multiply-adds (the 75-tap kernel at T <= ~600: 4-10x fewer) and on the direct
    implicit-GEMM kernels elsewhere (9 taps, T = 1024, 3 channels); "f32_direct" never leaves the direct kernels,
    "f32_dft" takes the frequency-domain path wherever its kernels exist.  All three are held to the same fp32 bars."""
    if dtype not in ("f32", "f32_direct", "bf16", "f32_split", "f32_split_wgrad", "f32_split_gcn", "f32_dft"):
        raise ValueError("mfma dtype must be 'f32', 'f32_direct', 'bf16', 'f32_split', 'f32_split_wgrad', 'f32_split_gcn' or "
                         "'f32_dft', got %r" % (dtype,))
    from .net.blocks import _StBlock
    for m in model.modules():
        if isinstance(m, _StBlock):
            m.mfma_dtype = dtype
    return model


__all__ = ["_lib", "set_mfma_dtype"]
