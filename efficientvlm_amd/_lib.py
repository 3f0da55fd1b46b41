"""ctypes binding of libevlm_hip.so (the C ABI declared in include/evlm_hip.h).

The product path has NO fallback: if the shared library is missing or a kernel call fails, this module
raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C efficientvlm_amd/csrc``.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EVLM_LIB", os.path.join(_HERE, "libevlm_hip.so"))

ABI_VERSION = 9      # evlm_abi_version() of the library this binding was written against (struct layouts, entry points)
F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_QUICK_GELU = 0, 1, 2
GATE_PRE, GATE_POST = 0, 1

_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64


class GemmArgs(C.Structure):
    _fields_ = [("dtype", _i), ("c_f32", _i), ("p_trans", _i), ("q_trans", _i),
                ("I", _i), ("J", _i), ("K", _i),
                ("ldp", _i), ("ldq", _i), ("ldc", _i), ("ldx", _i),
                ("P", _vp), ("Q", _vp), ("C", _vp), ("bias", _vp), ("gate", _vp),
                ("preact", _vp), ("aux", _vp), ("residual", _vp),
                ("alpha", _f), ("act", _i), ("gate_pos", _i), ("dact", _i), ("accumulate", _i), ("psum", _vp),
                ("sk_workspace", _vp), ("dgate", _vp),
                ("dropout_p", _f), ("rng_state", _vp), ("call_id", C.c_uint32)]


class WgradProblem(C.Structure):
    _fields_ = [("P", _vp), ("Q", _vp), ("C", _vp), ("psum", _vp),
                ("I", _i), ("J", _i), ("ldp", _i), ("ldq", _i), ("ldc", _i), ("assign", _i)]


class AttnFwdArgs(C.Structure):
    _fields_ = [("dtype", _i), ("p_dtype", _i),
                ("B", _i), ("H", _i), ("Lq", _i), ("Lk", _i), ("dh", _i),
                ("ldq", _i), ("ldk", _i), ("ldv", _i), ("ldo", _i), ("ldpr", _i),
                ("Q", _vp), ("K", _vp), ("V", _vp), ("kv_index", _vp), ("mask", _vp), ("head_gate", _vp),
                ("scale", _f), ("O", _vp), ("P", _vp), ("causal", _i),
                ("dropout_p", _f), ("rng_state", _vp), ("call_id", C.c_uint32),
                ("kd_teacher", _vp), ("kd_loss", _vp), ("kd_weight", _f), ("lse", _vp), ("Bkv", _i),
                ("kd_rowdot", _vp),
                ("kd_tq", _vp), ("kd_tk", _vp), ("kd_tld", _i), ("kd_tlse", _vp)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("dtype", _i), ("p_dtype", _i),
                ("B", _i), ("H", _i), ("Lq", _i), ("Lk", _i), ("dh", _i), ("Bkv", _i),
                ("ldq", _i), ("ldk", _i), ("ldv", _i), ("ldo", _i),
                ("lddq", _i), ("lddk", _i), ("lddv", _i), ("ldpr", _i),
                ("Q", _vp), ("K", _vp), ("V", _vp), ("P", _vp), ("dO", _vp), ("dP_ext", _vp),
                ("kv_index", _vp), ("head_gate", _vp), ("scale", _f),
                ("dS", _vp), ("dQ", _vp), ("dK", _vp), ("dV", _vp), ("dgate", _vp),
                ("dropout_p", _f), ("rng_state", _vp), ("call_id", C.c_uint32),
                ("kd_teacher", _vp), ("kd_gout", _vp), ("kd_weight", _f),
                ("lse", _vp), ("mask", _vp), ("causal", _i), ("P_ws", _vp), ("O", _vp), ("kd_rowdot", _vp),
                ("kd_tq", _vp), ("kd_tk", _vp), ("kd_tld", _i), ("kd_tlse", _vp)]


class XAttnFusedArgs(C.Structure):
    _fields_ = [("dtype", _i), ("Bimg", _i), ("Bq", _i), ("N", _i), ("Lq", _i), ("d", _i), ("H", _i), ("dh", _i),
                ("ldx", _i), ("ldq", _i), ("ldo", _i), ("ldpr", _i),
                ("X", _vp), ("Wkv", _vp), ("bias_kv", _vp), ("Q", _vp), ("kv_index", _vp), ("mask", _vp),
                ("head_gate", _vp), ("scale", _f), ("O", _vp), ("P", _vp)]


# name -> argtypes (every symbol include/evlm_hip.h declares; tests check the library exports all of them)
SIGNATURES = {
    "evlm_gemm": [C.POINTER(GemmArgs), _vp],
    "evlm_colsum": [_i, _vp, _i, _i, _i, _vp, _vp],
    "evlm_layernorm_fwd": [_i, _vp, _vp, _vp, _f, _i, _i, _vp, _vp, _vp, _vp],
    "evlm_wgrad_grouped": [C.POINTER(WgradProblem), _i, _i, _vp],
    "evlm_layernorm_bwd_blocks": [_i],
    "evlm_layernorm_bwd": [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "evlm_layernorm_bwd_add": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "evlm_layernorm_bwd_drop": [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _f, _vp, C.c_uint32, _vp, _vp, _vp, _vp],
    "evlm_layernorm_fwd_kd_slots": [],
    "evlm_layernorm_fwd_kd": [_i, _vp, _vp, _vp, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _f, _vp],
    "evlm_layernorm_bwd_kd": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp],
    "evlm_attention_fwd": [C.POINTER(AttnFwdArgs), _vp],
    "evlm_attention_bwd": [C.POINTER(AttnBwdArgs), _vp],
    "evlm_attention_lse_supported": [_i, _i, _i, _f],
    "evlm_xattn_fused_fwd": [C.POINTER(XAttnFusedArgs), _vp],
    "evlm_mse_fwd": [_i, _vp, _i, _vp, _i64, _f, _vp, _vp],
    "evlm_mse_bwd": [_i, _vp, _i, _vp, _i64, _f, _vp, _vp, _vp],
    "evlm_ce_fwd": [_i, _vp, _i, _i, _i, _vp, _i, _f, _vp, _vp, _vp, _vp],
    "evlm_ce_bwd": [_i, _vp, _i, _i, _i, _vp, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "evlm_transpose_grouped": [_vp, _i, _i, _vp],
    "evlm_copy_grouped": [_vp, _i, _i, _vp],
    "evlm_copy_few": [_vp, _vp, _vp, _i, _vp],
    "evlm_ce_weighted_fwd": [_i, _vp, _i, _i, _i, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp],
    "evlm_ce_weighted_bwd": [_i, _vp, _i, _i, _i, _vp, _i, _f, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "evlm_kl_fwd": [_i, _vp, _i, _i, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp],
    "evlm_kl_bwd": [_i, _vp, _i, _i, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "evlm_kl_fwd_rows": [_i, _vp, _i, _i, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "evlm_kl_bwd_rows": [_i, _vp, _i, _i, _vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp],
    "evlm_log_softmax_fwd": [_i, _vp, _i, _i, _i, _vp, _i, _vp],
    "evlm_log_softmax_bwd": [_i, _vp, _vp, _i, _i, _i, _vp, _vp],
    "evlm_bert_embed_fwd": [_i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "evlm_bert_embed_bwd": [_i, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp],
    "evlm_im2row": [_i, _vp, _i, _i, _i, _i, _vp, _vp],
    "evlm_vit_embed_fwd": [_i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "evlm_vit_embed_bwd": [_i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "evlm_gather_rows_fwd": [_i, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "evlm_gather_rows_bwd": [_i, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "evlm_l2norm_fwd": [_i, _vp, _i, _i, _i, _f, _vp, _vp, _vp],
    "evlm_l2norm_bwd": [_i, _vp, _vp, _vp, _i, _i, _vp, _vp],
    "evlm_cast": [_i, _vp, _i, _vp, _i64, _vp],
    "evlm_act_fwd": [_i, _vp, _i64, _i, _vp, _vp],
    "evlm_gated_act_bwd": [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "evlm_l0_sample_fwd": [_vp, _vp, _i64, _f, _vp, _vp],
    "evlm_l0_sample_bwd": [_vp, _vp, _vp, _i64, _f, _vp, _vp],
    "evlm_l0_deterministic": [_vp, _i, _i, _f, _f, _vp, _vp],
    "evlm_l0_lagrangian_fwd": [_vp, _i, _f, _f, _f, _f, _f, _f, _vp, _f, _vp, _vp, _vp, _vp],
    "evlm_l0_lagrangian_bwd": [_vp, _vp, _i, _i64, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "evlm_dropout": [_i, _vp, _vp, _i64, _f, _vp, C.c_uint32, _vp, _vp],
    "evlm_dropout_mask": [_i64, _f, _vp, C.c_uint32, _vp, _vp],
    "evlm_layernorm_bwd_reduce_grouped": [_vp, _i, _i, _vp],
    "evlm_mse_grouped": [_i, _i, _vp, _i, _i, _vp],
    "evlm_sample_negatives": [_vp, _i, _i, _vp, _vp, _vp, C.c_uint32, _vp, _vp, _vp, _vp],
    "evlm_select_batches_fwd": [_vp, _vp, _i, C.c_int64, _vp, _vp],
    "evlm_select_batches_bwd": [_i, _vp, _vp, _i, _i, C.c_int64, _vp, _vp],
    "evlm_itc_loss_fwd": [_i, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    "evlm_itc_loss_bwd": [_i, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp],
    "evlm_sumsq": [_vp, _i64, _vp, _vp, _vp],
    "evlm_adamw_step": [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _f, _f, _vp, _f, _vp, _vp, _vp],
}

_lib = None


def load():
    """dlopen libevlm_hip.so (once).  Raises with build instructions when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP kernels are the only implementation of this package "
            "(there is no CPU/PyTorch fallback). Build them with `make -C efficientvlm_amd/csrc` "
            "or `python -c 'import __graft_entry__ as g; g.build()'`.")
    lib = C.CDLL(LIB_PATH)
    lib.evlm_last_error.restype = C.c_char_p
    lib.evlm_abi_version.restype = _i
    lib.evlm_gemm_last_kernel.restype = C.c_char_p
    if lib.evlm_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} has ABI version {lib.evlm_abi_version()}, this package expects {ABI_VERSION}: "
                           "rebuild it (`make -C efficientvlm_amd/csrc`)")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library drift
        fn.argtypes = argtypes
        fn.restype = _i
    _lib = lib
    return lib


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError(f"libevlm_hip {what}: {_lib.evlm_last_error().decode()}")


def dt(t):
    """torch dtype (or tensor) -> EVLM dtype code"""
    d = t.dtype if torch.is_tensor(t) else t
    if d == torch.float32:
        return F32
    if d == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {d} (float32 / bfloat16 only)")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("efficientvlm_amd ops run on the GPU only (HIP kernels, no CPU fallback); "
                               "got a tensor on " + str(t.device))
