"""ctypes binding of libt2s_hip.so (C ABI: include/t2s.h).

The HIP library IS the product: there is no CPU or PyTorch fallback.  If the
shared object is missing, or a tensor is not on a GPU, the call fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("T2S_LIB", os.path.join(_HERE, "libt2s_hip.so"))   # T2S_LIB: A/B another build (tools/)

N_BLOCKS = 4
LAT_C, LAT_W, LAT = 64, 30, 1920
D_MODEL, N_TOK = 128, 480
TRAIN_F32, TRAIN_BF16 = 0, 1   # t2s.h: T2S_TRAIN_F32 / T2S_TRAIN_BF16
MATH_F32, MATH_BF16X3 = 0, 1    # t2s.h: T2S_MATH_F32 / T2S_MATH_BF16X3
DIT_N_TENSORS = 10 + 10 * N_BLOCKS   # t2s.h: T2S_DIT_N_TENSORS (pointers of t2s_dit_weights, declaration order)
MSE_SCRATCH_FLOATS = 1024       # t2s.h: T2S_MSE_SCRATCH_FLOATS

c_float_p = C.c_void_p  # device pointers travel as opaque addresses


class DitBlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("qkv_w", "qkv_b", "proj_w", "proj_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b", "ada_w", "ada_b")]


class DitWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("conv_w", "conv_b", "patch_w", "patch_b", "pos_embed", "ln_w", "ln_b", "out_w", "out_b",
                 "time_freqs")] + [("blk", DitBlockWeights * N_BLOCKS)]


class DitBlockGrads(C.Structure):
    _fields_ = DitBlockWeights._fields_


class DitGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("conv_w", "conv_b", "patch_w", "patch_b", "ln_w", "ln_b", "out_w", "out_b")] + \
               [("blk", DitBlockGrads * N_BLOCKS)]


class VaeStackWeights(C.Structure):
    _fields_ = [("conv3_w", C.c_void_p * 4), ("conv1_w", C.c_void_p * 4)]


class VaeWeights(C.Structure):
    _fields_ = [("hidden", C.c_int), ("res_hidden", C.c_int), ("n_res_layers", C.c_int), ("emb", C.c_int),
                ("dec_conv1_w", C.c_void_p), ("dec_conv1_b", C.c_void_p), ("dec_stack", VaeStackWeights),
                ("dec_ct1_w", C.c_void_p), ("dec_ct1_b", C.c_void_p), ("dec_ct2_w", C.c_void_p),
                ("dec_ct2_b", C.c_void_p),
                ("enc_conv1_w", C.c_void_p), ("enc_conv1_b", C.c_void_p), ("enc_conv2_w", C.c_void_p),
                ("enc_conv2_b", C.c_void_p), ("enc_conv3_w", C.c_void_p), ("enc_conv3_b", C.c_void_p),
                ("enc_stack", VaeStackWeights), ("enc_prevq_w", C.c_void_p), ("enc_prevq_b", C.c_void_p)]


class VaeEncGrads(C.Structure):
    _fields_ = [("conv1_w", C.c_void_p), ("conv1_b", C.c_void_p), ("conv2_w", C.c_void_p), ("conv2_b", C.c_void_p),
                ("conv3_w", C.c_void_p), ("conv3_b", C.c_void_p), ("stack_conv3_w", C.c_void_p * 4), ("stack_conv1_w", C.c_void_p * 4),
                ("prevq_w", C.c_void_p), ("prevq_b", C.c_void_p)]


TS2VEC_MAX_BLOCKS = 16


class Ts2vecWeights(C.Structure):
    _fields_ = [("input_dims", C.c_int), ("hidden", C.c_int), ("output_dims", C.c_int), ("depth", C.c_int),
                ("fc_w", C.c_void_p), ("fc_b", C.c_void_p),
                ("conv1_w", C.c_void_p * TS2VEC_MAX_BLOCKS), ("conv1_b", C.c_void_p * TS2VEC_MAX_BLOCKS),
                ("conv2_w", C.c_void_p * TS2VEC_MAX_BLOCKS), ("conv2_b", C.c_void_p * TS2VEC_MAX_BLOCKS),
                ("proj_w", C.c_void_p), ("proj_b", C.c_void_p)]


MLP_LAYERS, MLP_PACKED_FLOATS = 8, 397888   # t2s.h: T2S_MLP_LAYERS, T2S_MLP_PACKED_FLOATS
MLP_LAYER_FIELDS = (("value_w", "cross_attn.value.weight"), ("value_b", "cross_attn.value.bias"),
                    ("proj_w", "cross_attn.proj.weight"), ("proj_b", "cross_attn.proj.bias"),
                    ("norm2_w", "norm2.weight"), ("norm2_b", "norm2.bias"),
                    ("mlp0_w", "mlp.0.weight"), ("mlp0_b", "mlp.0.bias"), ("mlp2_w", "mlp.2.weight"), ("mlp2_b", "mlp.2.bias"),
                    ("pos0_w", "mlp2.0.weight"), ("pos0_b", "mlp2.0.bias"), ("pos2_w", "mlp2.2.weight"), ("pos2_b", "mlp2.2.bias"))


class MlpLayerWeights(C.Structure):
    _fields_ = [(f, C.c_void_p) for f, _ in MLP_LAYER_FIELDS]


class MlpWeights(C.Structure):
    _fields_ = [("layer", MlpLayerWeights * MLP_LAYERS)]


MLP_GRAD_PART_FLOATS = 391744           # t2s.h: T2S_MLP_GRAD_PART_FLOATS


class MlpGrads(C.Structure):            # t2s_mlp_grads: the same fields, writable
    _fields_ = [("layer", MlpLayerWeights * MLP_LAYERS)]


class SampleConfig(C.Structure):
    _fields_ = [("mode", C.c_int), ("steps", C.c_int), ("cfg_scale", C.c_float), ("batch", C.c_int),
                ("length", C.c_int), ("use_graph", C.c_int), ("seed", C.c_uint64), ("row0", C.c_uint32),
                ("ddpm_coef", C.c_void_p), ("t_values", C.c_void_p)]


MODE_DDPM, MODE_RF = 0, 1

# every symbol include/t2s.h declares: (name, restype, argtypes)
_VP, _I, _F, _U64, _U32 = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_uint32
SYMBOLS = {
    "t2s_last_error": (C.c_char_p, []),
    "t2s_version": (C.c_char_p, []),
    "t2s_dit_create": (_I, [C.POINTER(DitWeights), _I, C.POINTER(_VP)]),
    "t2s_dit_update_weights": (_I, [_VP, C.POINTER(DitWeights), _VP]),
    "t2s_dit_destroy": (None, [_VP]),
    "t2s_dit_max_seqs": (_I, [_VP]),
    "t2s_time_embedding": (_I, [_VP, _VP, _VP, _I, _VP]),
    "t2s_time_embedding_freqs": (_I, [_VP, _VP, _VP, _I, _VP]),
    "t2s_dit_weights_check": (_I, [C.POINTER(DitWeights), C.POINTER(C.c_uint64), _I]),
    "t2s_dit_forward": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _I, _VP]),
    "t2s_dit_forward_cfg": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _VP]),
    "t2s_dit_forward_cfg_rows": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _VP, _I, _VP]),
    "t2s_dit_read_stream": (_I, [_VP, _VP, _I, _VP]),
    "t2s_dit_timing_begin": (_I, [_VP]),
    "t2s_dit_timing_end": (_I, [_VP, C.POINTER(C.c_double)]),
    "t2s_dit_timing_end_ex": (_I, [_VP, C.POINTER(C.c_double), _I]),
    "t2s_dit_set_train_dtype": (_I, [_VP, _I]),
    "t2s_dit_set_math": (_I, [_VP, _I]),
    "t2s_eval_mse_wape": (_I, [_VP, _VP, _VP, _VP, _I, _I, _VP]),
    "t2s_eval_mrr": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _F, _VP]),
    "t2s_eval_ed": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "t2s_eval_crps": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP]),
    "t2s_eval_dtw": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "t2s_ts2vec_encode": (_I, [C.POINTER(Ts2vecWeights), _VP, _VP, _VP, _I, _I, _VP]),
    "t2s_mlp_pack": (_I, [C.POINTER(MlpWeights), _VP, _VP]),
    "t2s_mlp_forward": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _VP]),
    "t2s_mlp_backward": (_I, [C.POINTER(MlpWeights), _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.POINTER(MlpGrads), _VP, _U64, _I, _VP]),
    "t2s_attn_fwd_x3": (_I, [_VP, _VP, _VP, _VP, _I, _VP]),
    "t2s_attn_fwd_bf16": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _VP]),
    "t2s_dit_train_forward": (_I, [_VP, C.POINTER(DitWeights), _VP, _VP, _I, _VP, _VP, _I, _VP]),
    "t2s_dit_train_backward": (_I, [_VP, _VP, C.POINTER(DitGrads), _I, _VP]),
    "t2s_dit_train_input_grad": (_I, [_VP, _VP, _I, _VP]),
    "t2s_adamw_step": (_I, [_VP, _VP, _VP, _VP, _U64, _F, _F, _F, _F, _F, _I, _VP]),
    "t2s_adamw_step_multi": (_I, [_VP, _I, _U64, _F, _F, _F, _F, _F, _I, _VP]),
    "t2s_mse_backward": (_I, [_VP, _VP, _VP, _VP, _VP, _U64, _VP]),
    "t2s_attn_fwd": (_I, [_VP, _VP, _VP, _VP, _I, _VP]),
    "t2s_attn_fwd_packed": (_I, [_VP, _VP, _VP, _VP, _I, _VP]),
    "t2s_ddpm_step": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _F, _U64, _U32, _U32, _I, _VP]),
    "t2s_ddpm_p_sample": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _VP]),
    "t2s_ddpm_p_sample_n": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "t2s_ddpm_q_sample_n": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "t2s_vae_decode_w": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _VP]),
    "t2s_mse": (_I, [_VP, _VP, _VP, _U64, _VP]),
    "t2s_mse_ws": (_I, [_VP, _VP, _VP, _U64, _VP, _VP]),
    "t2s_rf_step": (_I, [_VP, _VP, _VP, _F, _F, _I, _VP]),
    "t2s_ddpm_q_sample": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _VP]),
    "t2s_rf_create_flow": (_I, [_VP, _VP, _VP, _VP, _I, _VP]),
    "t2s_philox_normal": (_I, [_VP, _U64, _U32, _U32, _I, _I, _VP]),
    "t2s_philox_uniform": (_I, [_VP, _U64, _U32, _U32, _I, _I, _VP]),
    "t2s_vae_create": (_I, [C.POINTER(VaeWeights), C.POINTER(_VP)]),
    "t2s_vae_destroy": (None, [_VP]),
    "t2s_vae_update_weights": (_I, [_VP, C.POINTER(VaeWeights), _VP]),
    "t2s_vae_encode_backward": (_I, [_VP, _VP, _VP, _VP, C.POINTER(VaeEncGrads), _I, _I, _VP]),
    "t2s_vae_decode": (_I, [_VP, _VP, _VP, _VP, _I, _I, _VP]),
    "t2s_vae_encode": (_I, [_VP, _VP, _VP, _VP, _I, _I, _VP]),
    "t2s_sampler_create": (_I, [_VP, _VP, C.POINTER(SampleConfig), C.POINTER(_VP)]),
    "t2s_sampler_destroy": (None, [_VP]),
    "t2s_sampler_run": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "t2s_sampler_set_lanes": (_I, [_VP, _I]),
    "t2s_sampler_set_loop_graph": (_I, [_VP, _I]),
    "t2s_sampler_set_row0": (_I, [_VP, _U32]),
    "t2s_sampler_graph_lanes": (_I, [_VP]),
    "t2s_sampler_lane_pool": (_I, []),
}

_lib: Optional[C.CDLL] = None


class T2SError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libt2s_hip.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise T2SError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C t2ms_amd/csrc` (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the .so does not export it
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().t2s_last_error().decode("utf-8", "replace")
        raise T2SError(f"{what or 't2s call'} failed (rc={rc}): {msg}")


def dev_ptr(t: Optional[torch.Tensor], name: str = "tensor", dtype=torch.float32) -> Optional[int]:
    """Raw device address of a contiguous CUDA(HIP) tensor; None passes NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise T2SError(f"{name} must live on a GPU (got device {t.device}); the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise T2SError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise T2SError(f"{name} must be contiguous")
    return t.data_ptr()


def as_f32(t: torch.Tensor) -> torch.Tensor:
    """fp32 contiguous view/copy on the tensor's own (GPU) device."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_DEVICE_LOCKS = {}


def device_lock(device) -> "threading.RLock":
    """Per device, re-entrant: held by everything in this package that allocates, frees, copies synchronously or synchronises
    the DEVICE -- building / growing / destroying a handle, a Sampler's create / stage / run / destroy -- so that none of it
    runs while another thread's sampler run has a stream capture open.  HIP answers a device-wide synchronous call from ANY
    thread (hipDeviceSynchronize = torch.cuda.synchronize(), a synchronous hipMemcpy) with an error while a capture is open
    on the device, thread-local capture mode notwithstanding, and that error invalidates the capture (reproduced in round 5:
    tools/stress_threads.py --unserialised).  The caller's OWN device-wide calls in other threads are not
    covered: see include/t2s.h "Threads"."""
    import threading
    return _DEVICE_LOCKS.setdefault(str(torch.device(device)), threading.RLock())


def destroy_locked(fn_name: str, device_key: str, ptr) -> None:
    """Finalizer body of the handle owners: t2s_*_destroy frees device memory -- under the device's lock (device_lock)."""
    with device_lock(device_key):
        getattr(lib(), fn_name)(ptr)


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream
