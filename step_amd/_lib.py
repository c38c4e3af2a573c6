"""ctypes binding of libstep_hip.so (the C ABI declared in include/step_hip.h).

There is NO fallback: if the shared library is missing or a call fails, this raises.
Tensors are passed as raw device pointers; the launch stream is torch's current stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STEP_HIP_LIB") or os.path.join(_HERE, "lib", "libstep_hip.so")   # override: A/B builds

P = ctypes.c_void_p
I = ctypes.c_int
F = ctypes.c_float
Z = ctypes.c_size_t
DBL = ctypes.c_double

# name -> (restype, argtypes); must match include/step_hip.h
SIGNATURES = {
    "step_abi_version": (I, []),
    "step_last_error": (ctypes.c_char_p, []),
    "step_bn_eval_prepare": (I, [P, P, P, P, F, P, P, I, P]),
    "step_bn_ws_bytes": (Z, [I]),
    "step_bn_train_prepare": (I, [P, I, I, I, I, I, P, P, P, P, F, F, P, P, P, P, P, Z, P]),
    "step_bn_train_stats": (I, [P, I, I, I, I, I, P, P, Z, P]),
    "step_bn_train_finalize": (I, [P, DBL, P, P, P, P, F, F, P, P, P, P, I, P]),
    "step_bn_train_prepare_static": (I, [P, I, I, I, I, I, P, P, P, P, F, F, P, P, P, P, P]),
    "step_affine_act": (I, [P, P, P, P, P, P, I, P, I, I, I, I, I, P]),
    "step_bn_bwd_reduce": (I, [P, P, P, P, P, I, I, I, I, I, I, P, P, Z, P]),
    "step_bn_bwd_apply": (I, [P, P, P, P, P, I, P, P, P, P, I, I, P, P, P, I, I, I, I, I, P, Z, P]),
    "step_bn_bwd_tail_pair": (I, [P, P, P, P, P, P, P, P, P, P, I, P, P, P, P, P, P, I, I, I, I, P, Z, P]),
    "step_masked_add": (I, [P, P, P, I, Z, P]),
    "step_gcn_fwd": (I, [P, P, P, P, P, I, P, I, P, P, P, I, P, I, I, I, I, I, I, P]),
    "step_gcn_fwd_split": (I, [P, P, P, P, P, I, P, I, P, P, P, I, P, I, I, I, I, I, I, P]),
    "step_gcn_fwd_stats_ws_bytes": (Z, [I]),
    "step_gcn_fwd_stats": (I, [P, P, P, P, P, I, P, I, P, P, P, I, P, I, I, I, I, I, I, P, P, P, Z, P]),
    "step_gcn_bwd_ws_bytes": (Z, [I, I, I, I, I]),
    "step_gcn_bwd": (I, [P, P, P, P, P, P, P, I, P, I, P, I, P, P, P, P, P, P,
                         I, I, I, I, I, I, P, Z, P]),
    "step_gcn_bwd_idres": (I, [P, P, P, P, P, P, P, I, P, I, P, I, P, P, P, P, P, P, P, P,
                               I, I, I, I, I, I, P, Z, P]),
    "step_gcn_bwd_split": (I, [P, P, P, P, P, P, P, I, P, I, P, I, P, P, P, P, P, P, P, P,
                               I, I, I, I, I, I, P, Z, P]),
    "step_tcn_packed_elems": (Z, [I, I, I, I, I]),
    "step_tcn_pack_weight": (I, [P, P, I, I, I, I, I, P]),
    "step_tcn_conv": (I, [P, P, P, I, P, P, P, P, P, P, P, I, P, I, I, I, I, I, I, I, I, P]),
    "step_tcn_conv_stats_ws_bytes": (Z, [I, I, I]),
    "step_tcn_conv_stats": (I, [P, P, P, I, P, P, P, I, I, I, I, I, I, I, I, P, P, Z, P]),
    "step_dft_rows": (I, [I, I]),
    "step_dft_table_geom": (I, [I, I, P, P]),
    "step_dft_groups": (I, [I]),
    "step_dft_fwd_seg": (I, [P, P, P, I, P, P, I, I, I, I, I, I, I, I, P]),
    "step_dft_inv_seg": (I, [P, P, P, P, P, P, P, P, I, P, I, I, I, I, I, I, I, P, P, Z, P]),
    "step_dft_fwd": (I, [P, P, P, I, P, P, I, I, I, I, P]),
    "step_dft_fwd_ph": (I, [P, P, P, I, P, P, I, I, I, I, I, P]),
    "step_dft_inv_ph": (I, [P, P, P, I, I, I, I, I, P]),
    "step_dft_mix": (I, [P, P, P, I, I, I, I, P]),
    "step_dft_inv_ws_bytes": (Z, [I, I, I]),
    "step_dft_inv": (I, [P, P, P, P, P, P, P, P, I, P, I, I, I, I, P, P, Z, P]),
    "step_dft_wspec": (I, [P, P, P, P, P, I, I, I, I, I, P]),
    "step_dft_wgrad_ws_bytes": (Z, [I, I, I, I]),
    "step_dft_wgrad": (I, [P, P, P, P, P, P, I, I, I, I, I, I, P, Z, P]),
    "step_tcn_conv_stats_bf16_ws_bytes": (Z, [I]),
    "step_tcn_conv_stats_bf16": (I, [P, P, P, I, P, P, P, I, I, I, I, I, I, I, I, I, P, P, Z, P]),
    "step_tcn_const_weights": (I, [P, P, I, I, I, I, I, P]),
    "step_tcn_const_fwd": (I, [P, P, P, P, P, P, P, P, I, P, I, I, I, I, P]),
    "step_tcn_packed_bytes_bf16": (Z, [I, I, I, I, I]),
    "step_tcn_pack_weight_bf16": (I, [P, P, I, I, I, I, I, P]),
    "step_tcn_conv_bf16": (I, [P, P, P, I, P, P, P, P, P, P, P, I, P, I, I, I, I, I, I, I, I, P]),
    "step_tcn_packed_bytes_split": (Z, [I, I, I, I, I]),
    "step_tcn_pack_weight_split": (I, [P, P, I, I, I, I, I, P]),
    "step_tcn_conv_split": (I, [P, P, P, I, P, P, P, P, P, P, P, I, P, I, I, I, I, I, I, I, I, P]),
    "step_tcn_wgrad_ws_bytes": (Z, [I, I, I, I, I, I]),
    "step_tcn_wgrad": (I, [P, P, P, P, I, P, P, I, I, I, I, I, I, I, I, P, Z, P]),
    "step_tcn_wgrad_bf16_ws_bytes": (Z, [I, I, I, I, I, I, I]),
    "step_tcn_wgrad_bf16": (I, [P, P, P, P, I, P, P, I, I, I, I, I, I, I, P, Z, P]),
    "step_tcn_wgrad_bf16_t": (I, [P, P, P, P, I, P, P, I, I, I, I, I, I, I, I, P, Z, P]),
    "step_tcn_wgrad_split": (I, [P, P, P, P, I, P, P, I, I, I, I, I, I, I, I, P, Z, P]),
    "step_vae_loss_ws_bytes": (Z, [I]),
    "step_vae_loss": (I, [P, P, P, P, F, P, P, P, P, I, I, I, I, I, P, Z, P]),
    "step_pose_align": (I, [P, P, P, I, I, I, P]),
    "step_pose_align_ragged": (I, [P, P, P, P, I, I, I, I, P]),
    "step_aff_features": (I, [P, I, P, I, I, DBL, I, P]),
    "step_aff_normalize": (I, [P, P, I, I, I, P]),
    "step_pool_fwd": (I, [P, P, I, I, I, P]),
    "step_pool_bwd": (I, [P, P, I, I, I, P]),
    "step_head_fwd": (I, [P, P, P, I, I, P, P, I, I, I, I, P]),
    "step_dense_fwd": (I, [P, P, P, I, P, I, I, I, P]),
    "step_dense_bwd": (I, [P, P, P, I, P, P, P, I, I, I, P]),
    "step_bn_rows_fwd": (I, [P, P, P, P, P, I, F, F, P, P, P, I, I, P]),
    "step_bn_rows_bwd": (I, [P, P, P, P, P, I, P, P, P, I, I, P]),
    "step_bn_rows_stats": (I, [P, P, I, I, P]),
    "step_bn_rows_apply": (I, [P, P, P, P, P, P, F, F, P, P, P, I, I, P]),
    "step_bn_rows_bwd_sums": (I, [P, P, P, P, P, P, P, I, I, P]),
    "step_bn_rows_bwd_apply": (I, [P, P, P, P, P, P, P, I, I, P]),
    "step_softmax_fwd": (I, [P, P, I, I, P]),
    "step_softmax_bwd": (I, [P, P, P, I, I, P]),
}

_lib = None


class StepHipError(RuntimeError):
    pass


def load():
    """Load the library once; raise loudly if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StepHipError(
                "libstep_hip.so not found at %s - build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if lib.step_abi_version() != 1:
            raise StepHipError("libstep_hip.so ABI version %d != 1" % lib.step_abi_version())
        _lib = lib
    return _lib


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise StepHipError("step_amd kernels need tensors on the GPU (got device %s); there is no "
                           "CPU path" % t.device)
    if not t.is_contiguous():
        raise StepHipError("non-contiguous tensor passed to the C ABI")
    return t.data_ptr()


def stream(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def call(name, *args):
    """Call an int-returning entry point on torch's current stream OF THE TENSORS' DEVICE; raise on error.
    Every tensor argument must live on one GPU; the launch (and any per-device kernel attribute the library
    sets) happens with that GPU current, whatever torch's current device is."""
    lib = load()
    dev = None
    conv = []
    for a in args:
        if a is None or isinstance(a, torch.Tensor):
            conv.append(_ptr(a))
            if a is not None:
                if dev is None:
                    dev = a.device
                elif a.device != dev:
                    raise StepHipError("%s: tensor arguments on different devices (%s and %s)"
                                       % (name, dev, a.device))
        else:
            conv.append(a)
    fn = getattr(lib, name)
    if dev is None or dev.index == torch.cuda.current_device():
        rc = fn(*conv, stream(dev))
    else:
        with torch.cuda.device(dev):
            rc = fn(*conv, stream(dev))
    if rc != 0:
        raise StepHipError("%s failed (%d): %s" % (name, rc, lib.step_last_error().decode()))


def query(name, *args):
    """Call a size_t-returning helper."""
    return getattr(load(), name)(*args)
