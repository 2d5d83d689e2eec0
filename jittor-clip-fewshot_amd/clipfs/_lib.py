"""ctypes binding of libclipfs_hip.so (C ABI in include/clipfs.h).

The library is the only compute path of this package: if it is missing the
import fails loudly -- there is NO PyTorch / CPU fallback.  Build it with
``python jittor-clip-fewshot_amd/build.py`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

ABI_VERSION = 2  # CLIPFS_ABI_VERSION of include/clipfs.h this table was written against

_HERE = os.path.dirname(os.path.abspath(__file__))
# CLIPFS_LIB_TAG=<tag>: an experiment build made with `build.py --tag <tag>` (A/B timing of kernel variants in one
# gpurun call; bench.py lists it among the active overrides).  Unset = the product library.
_TAG = os.environ.get("CLIPFS_LIB_TAG", "")
LIB_PATH = os.path.join(_HERE, f"libclipfs_hip_{_TAG}.so" if _TAG else "libclipfs_hip.so")

c_f32p = C.c_void_p  # device pointers travel as integers (tensor.data_ptr())
c_stream = C.c_void_p


class ClipfsError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_size_t),
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int),
        ("alpha", C.c_float),
        ("bias", C.c_void_p), ("residual", C.c_void_p), ("ldres", C.c_int),
        ("act", C.c_int), ("aux_out", C.c_void_p), ("aux_in", C.c_void_p),
        ("lora_t", C.c_void_p), ("lora_b", C.c_void_p),
        ("lora_r", C.c_int), ("lora_nseg", C.c_int), ("lora_seg_width", C.c_int),
        ("lora_scale", C.c_float),
        ("a_mode", C.c_int), ("img_res", C.c_int), ("patch", C.c_int), ("out_tokens", C.c_int),
        ("B_planes", C.c_void_p), ("b_format", C.c_int), ("workspace", C.c_void_p), ("workspace_floats", C.c_size_t),
        ("A_f16", C.c_void_p), ("C_f16", C.c_void_p), ("aux_f16", C.c_int),
        ("counters", C.c_void_p), ("counters_ints", C.c_size_t),
    ]


class Block(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "ln1_g", "ln1_b", "ln2_g", "ln2_b",
        "w_qkv", "b_qkv", "w_qkv_t",
        "w_o", "b_o", "w_o_t",
        "w_fc", "b_fc", "w_fc_t",
        "w_pr", "b_pr", "w_pr_t",
        "lora_a_qkv", "lora_b_qkv", "lora_a_o", "lora_b_o",
        "g_lora_a_qkv", "g_lora_b_qkv", "g_lora_a_o", "g_lora_b_o")] + [("lora_mask", C.c_uint)] + [
        (n, C.c_void_p) for n in ("w_qkv_p", "w_o_p", "w_fc_p", "w_pr_p", "w_qkv_t_p", "w_o_t_p", "w_fc_t_p", "w_pr_t_p")]


class Tower(C.Structure):
    _fields_ = [
        ("struct_size", C.c_size_t), ("block_size", C.c_size_t),
        ("width", C.c_int), ("heads", C.c_int), ("layers", C.c_int), ("seq", C.c_int), ("causal", C.c_int),
        ("lora_r", C.c_int), ("lora_scale", C.c_float), ("lora_dropout", C.c_float),
        ("dropout_seed", C.c_uint64), ("dropout_stream0", C.c_uint32), ("dropout_row0", C.c_uint32),
        ("blocks", C.POINTER(Block)), ("weight_format", C.c_int),
        ("gemm_counters", C.c_void_p), ("gemm_counters_ints", C.c_size_t),
    ]


_i, _f, _p, _sz = C.c_int, C.c_float, C.c_void_p, C.c_size_t
_u, _u64, _u32 = C.c_uint, C.c_uint64, C.c_uint32

# name -> (restype, argtypes).  Every symbol declared in include/clipfs.h is listed here and
# tests/test_abi.py checks the list against the header.
SIGNATURES = {
    "clipfs_abi_version": (_i, []),
    "clipfs_last_error": (C.c_char_p, []),
    "clipfs_source_stamp": (C.c_char_p, []),
    "clipfs_gemm_source_stamp": (C.c_char_p, []),
    "clipfs_gemm_nt": (_i, [C.POINTER(GemmArgs), _p]),
    "clipfs_split_bf16": (_i, [_p, _p, _sz, _p]),
    "clipfs_convert_f16": (_i, [_p, _p, _sz, _p]),
    "clipfs_gemm_splits": (_i, [_i, _i, _i]),
    "clipfs_gemm_workspace_floats": (_sz, [_i, _i, _i]),
    "clipfs_gemm_counter_ints": (_sz, [_i, _i, _i]),
    "clipfs_gemm_timing": (_i, [_i]),
    "clipfs_gemm_timing_collect": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "clipfs_gemm_timing_last_bytes": (C.c_double, []),
    "clipfs_layernorm_fwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "clipfs_layernorm_bwd": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "clipfs_layernorm_fwd_f16": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "clipfs_layernorm_fwd_lora_ok": (_i, [_i, _i, _i]),
    "clipfs_layernorm_fwd_lora": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p, _p, _i, _i, _u, _f, _u64, _u32, _u32, _p, _p]),
    "clipfs_layernorm_bwd_f16": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "clipfs_attention_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "clipfs_attention_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "clipfs_attention_lse_floats": (_sz, [_i, _i, _i]),
    "clipfs_attention_f16_fwd": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _p]),
    "clipfs_attention_f16_bwd": (_i, [_p, _i, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "clipfs_lora_keep_bits_ok": (_i, [_i, _i, _i, _i]),
    "clipfs_lora_down": (_i, [_p, _p, _p, _i, _i, _i, _i, _u, _f, _u64, _u32, _u32, _p, _p]),
    "clipfs_lora_bwd_work_floats": (_sz, [_i, _i, _i, _i]),
    "clipfs_lora_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _u, _f, _f, _u64, _u32, _u32, _p, _p, _p]),
    "clipfs_lora_bwd_f16dy_ok": (_i, [_i, _i, _i, _i]),
    "clipfs_lora_bwd_f16dy": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _u, _f, _f, _u64, _u32, _u32, _p, _p, _p]),
    "clipfs_vit_fill_special": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "clipfs_text_embed": (_i, [_p, _p, _p, _p, _i, _p, _i, _i, _i, _p]),
    "clipfs_token_rows_grad": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "clipfs_matmul_small": (_i, [_p, _p, _p, _i, _i, _i, C.c_long, C.c_long, C.c_long, C.c_long, _f, _p]),
    "clipfs_gather_eot": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "clipfs_scatter_rows": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "clipfs_gather_seq_rows": (_i, [_p, _sz, _p, _p, _i, _i, _i, _p]),
    "clipfs_add_seq_rows": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "clipfs_put_seq_rows": (_i, [_p, _p, _p, _sz, _i, _i, _i, _p]),
    "clipfs_eot_index": (_i, [_p, _p, _i, _i, _p]),
    "clipfs_gather_seq_rows_f16": (_i, [_p, _sz, _p, _p, _i, _i, _i, _p]),
    "clipfs_put_seq_rows_f16": (_i, [_p, _p, _p, _sz, _i, _i, _i, _p]),
    "clipfs_l2norm_fwd": (_i, [_p, _p, _p, _i, _i, _p]),
    "clipfs_l2norm_bwd": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "clipfs_class_mean_fwd": (_i, [_p, _p, _i, _i, _i, _p]),
    "clipfs_class_mean_bwd": (_i, [_p, _p, _p, _i, _i, _i, _p]),
    "clipfs_cross_entropy": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    "clipfs_topk": (_i, [_p, _p, _i, _i, _i, _p]),
    "clipfs_channel_affine": (_i, [_p, _p, _p, _p, _i, _i, _p]),
    "clipfs_logit_normalize": (_i, [_p, _p, _p, _i, _i, _p]),
    "clipfs_logit_normalize_bwd": (_i, [_p, _p, _p, _i, _i, _p]),
    "clipfs_colsum": (_i, [_p, _p, _p, _i, _i, _p]),
    "clipfs_l1_loss": (_i, [_p, _p, _sz, _p, _p, _f, _p]),
    "clipfs_kl_logits": (_i, [_p, _p, _p, _p, _i, _i, _f, _p]),
    "clipfs_adamw": (_i, [_p, _p, _p, _p, _sz, _i, _f, _f, _f, _f, _f, _f, _p]),
    "clipfs_mta_work_floats": (_sz, [_i, _i, _i, _i]),
    "clipfs_mta": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "clipfs_tta_views": (_i, [_p, _i, _i, _p, _i, _i, _p, _p, _p, _p]),
    "clipfs_bpe_create": (C.c_void_p, [C.c_char_p, _sz, _i]),
    "clipfs_bpe_destroy": (None, [C.c_void_p]),
    "clipfs_bpe_encode": (C.c_long, [C.c_void_p, _p, _p, _i, _p, _p, C.c_long]),
    "clipfs_nchw_to_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "clipfs_im2col_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "clipfs_maxpool3x3s2_nhwc": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "clipfs_global_avgpool_nhwc": (_i, [_p, _p, _i, _i, _i, _p]),
    "clipfs_tower_saved_floats": (_sz, [C.POINTER(Tower), _i]),
    "clipfs_tower_scratch_floats": (_sz, [C.POINTER(Tower), _i]),
    "clipfs_tower_counter_ints": (_sz, [C.POINTER(Tower), _i]),
    "clipfs_tower_fwd": (_i, [C.POINTER(Tower), _p, _i, _p, _p, _p]),
    "clipfs_tower_fwd_rows": (_i, [C.POINTER(Tower), _p, _p, _i, _p, _p, _p]),
    "clipfs_tower_rows_mode": (_i, [C.POINTER(Tower)]),
    "clipfs_tower_bwd": (_i, [C.POINTER(Tower), _p, _i, _p, _p, _i, _p]),
    "clipfs_tower_bwd_sparse": (_i, [C.POINTER(Tower), _p, _p, _p, _i, _p, _p, _i, _p]),
}

def new_gemm_args() -> GemmArgs:
    g = GemmArgs()
    g.struct_size = C.sizeof(GemmArgs)
    return g


def new_tower() -> Tower:
    t = Tower()
    t.struct_size = C.sizeof(Tower)
    t.block_size = C.sizeof(Block)
    return t


_lib = None


def load() -> C.CDLL:
    """dlopen the in-tree library (after torch, so the HIP runtime PyTorch already loaded is the one
    the library binds to: both carry SONAME libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ClipfsError(
            f"{LIB_PATH} not found: the HIP engine is not built.  Run "
            "`python jittor-clip-fewshot_amd/build.py` (needs hipcc).  There is no CPU fallback.")
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64 first)
    except Exception:  # pragma: no cover - symbol-export checks may run without torch
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL if hasattr(C, "RTLD_GLOBAL") else 0)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.clipfs_abi_version() != ABI_VERSION:
        raise ClipfsError(f"libclipfs_hip.so ABI version {lib.clipfs_abi_version()} != {ABI_VERSION} of this binding: "
                          "rebuild with `python jittor-clip-fewshot_amd/build.py --force`")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().clipfs_last_error().decode("utf-8", "replace")
        raise ClipfsError(f"{what or 'clipfs call'} failed (rc={rc}): {msg}")
