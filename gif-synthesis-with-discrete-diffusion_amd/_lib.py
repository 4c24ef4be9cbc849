"""ctypes binding of libgsdd.so (include/gsdd.h).  No fallback: a missing library is a hard error
the moment any operator is used (importing the package on a box without the .so still works so that
CPU-only host logic — configs, weight repacking — can be tested)."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSDD_LIB_PATH") or os.path.join(_HERE, "libgsdd.so")     # (the override: A/B runs of two builds)
_lib = None

EXPORTS = [
    "gsdd_last_error", "gsdd_version", "gsdd_abi_sizeof", "gsdd_gemm", "gsdd_row_stats", "gsdd_ncdhw_to_rows", "gsdd_preprocess_clip",
    "gsdd_axial_attention", "gsdd_pool3d", "gsdd_nearest_code", "gsdd_nearest_code_workspace_bytes", "gsdd_bn_train_workspace_bytes", "gsdd_bn_train",
    "gsdd_codebook_ema", "gsdd_code_perplexity", "gsdd_mse", "gsdd_conv_wgrad", "gsdd_bn_relu_bwd_workspace_bytes", "gsdd_bn_relu_bwd",
    "gsdd_relu_mask", "gsdd_lincomb", "gsdd_axial_attention_bwd", "gsdd_d3pm_embed", "gsdd_adaln_table", "gsdd_small_linear",
    "gsdd_d3pm_attention", "gsdd_d3pm_attention_workspace_bytes", "gsdd_d3pm_layer", "gsdd_d3pm_layer_pack", "gsdd_d3pm_layer_pack_h2", "gsdd_rows_linear_pack_many", "gsdd_rows_linear", "gsdd_d3pm_logits", "gsdd_d3pm_cross_attention", "gsdd_d3pm_step", "gsdd_d3pm_q_sample", "gsdd_d3pm_train_loss", "gsdd_d3pm_train_loss_bwd", "gsdd_d3pm_train_loss_grad", "gsdd_gelu2", "gsdd_ln_fwd", "gsdd_ln_bwd", "gsdd_wgrad",
    "gsdd_batch_rowsum", "gsdd_colsum", "gsdd_d3pm_attention_train", "gsdd_d3pm_attention_bwd", "gsdd_d3pm_attention_bwd_workspace_bytes", "gsdd_d3pm_embed_bwd", "gsdd_small_linear_bwd",
    "gsdd_adaln_bwd", "gsdd_adam", "gsdd_adam_multi", "gsdd_adam_multi_dev", "gsdd_advance",
    "gsdd_philox_uniform", "gsdd_graph_begin", "gsdd_graph_end", "gsdd_graph_launch", "gsdd_graph_destroy",
    "gsdd_event_create", "gsdd_event_record", "gsdd_event_elapsed_ms", "gsdd_event_destroy",
]

_p = C.c_void_p
_i = C.c_int
_i64 = C.c_int64


class GemmDesc(C.Structure):
    _fields_ = [
        ("in_", _p), ("N", _i), ("Di", _i), ("Hi", _i), ("Wi", _i), ("Cin", _i), ("in_pitch", _i),
        ("Do", _i), ("Ho", _i), ("Wo", _i), ("sd", _i), ("sh", _i), ("sw", _i),
        ("ntaps", _i), ("taps", _p), ("gather", _p),
        ("w", _p), ("Cout", _i),
        ("pro_scale", _p), ("pro_shift", _p), ("ln_stats", _p), ("ln_gamma", _p), ("ln_beta", _p), ("ln_sel", _p),
        ("ln_stride", _i), ("rows_per_batch", _i),
        ("epi_scale", _p), ("epi_shift", _p), ("bvec", _p), ("act", _i), ("residual", _p),
        ("out", _p), ("oD", _i), ("oH", _i), ("oW", _i), ("osd", _i), ("osh", _i), ("osw", _i),
        ("ood", _i), ("ooh", _i), ("oow", _i), ("out_pitch", _i), ("out_mode", _i), ("flags", _i),
    ]


class StepDesc(C.Structure):
    _fields_ = [
        ("logits_c", _p), ("logits_u", _p), ("tok_in", _p), ("tok_out", _p),
        ("B", _i), ("L", _i), ("K", _i), ("T", _i), ("guidance", C.c_float),
        ("sched", _p * 8), ("t_dev", _p), ("seed", C.c_uint64), ("stream_dev", _p), ("row0", _i64),
        ("post_dbg", _p), ("x0_dbg", _p), ("occupancy", _i),
    ]


class TrainDesc(C.Structure):
    _fields_ = [
        ("logits", _p), ("x0", _p), ("xt", _p), ("t_dev", _p), ("pt", _p),
        ("B", _i), ("L", _i), ("K", _i), ("T", _i), ("sched", _p * 8), ("mask_weight", C.c_float * 2),
        ("aux_weight", C.c_float), ("adaptive_aux", _i),
        ("kl", _p), ("nll", _p), ("aux", _p), ("x0_recon", _p), ("xt1_recon", _p),
        ("Lt_history", _p), ("Lt_count", _p), ("loss", _p), ("per_sample", _p), ("probs", _p),
    ]


class LayerDesc(C.Structure):
    _fields_ = [
        ("y", _p), ("x", _p), ("M", _i64), ("L", _i), ("n_embd", _i), ("hidden", _i), ("cvec", _p),
        ("wproj", _p), ("bproj", _p), ("ln2_g", _p), ("ln2_b", _p), ("w1", _p), ("b1", _p), ("w2", _p), ("b2", _p),
        ("ada", _p), ("t2", _p), ("wqkv", _p), ("bqkv", _p), ("qkv", _p), ("w2_x3", _p), ("wqkv_x3", _p), ("kv_img_bytes", _i64), ("kv_img", _p),
        ("layer_h2", _p), ("wqkv_h2", _p), ("variant", _i), ("range_flag", _p),
    ]


# variant / mode arguments of include/gsdd.h (GSDD_*): the library reads no environment variable; ops.py maps the debugging
# environment switches onto these
GEMM_EXACT_F32 = 1
AXIAL_AUTO, AXIAL_VALU = 0, 1
ATTN_AUTO, ATTN_P22, ATTN_P11, ATTN_A8, ATTN_A12, ATTN_F32PV, ATTN_KC256 = range(7)
LAYER_AUTO, LAYER_X3P, LAYER_H2 = 0, 2, 3
ATTN_BWD_AUTO, ATTN_BWD_VALU, ATTN_BWD_SPLIT, ATTN_BWD_FQC64, ATTN_BWD_FQC128, ATTN_BWD_NW8, ATTN_BWD_DBG1, ATTN_BWD_DBG2 = range(8)


class GsddError(RuntimeError):
    pass


def lib():
    """The loaded library; raises loudly if it has not been built (./build.sh or __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GsddError(f"{LIB_PATH} is missing: the HIP extension must be built (./build.sh); "
                            "there is no CPU fallback for this path")
        L = C.CDLL(LIB_PATH)
        L.gsdd_last_error.restype = C.c_char_p
        if not hasattr(L, "gsdd_abi_sizeof"):
            raise GsddError(f"{LIB_PATH} predates this binding (no gsdd_abi_sizeof): rebuild it with ./build.sh")
        L.gsdd_abi_sizeof.argtypes, L.gsdd_abi_sizeof.restype = [_i], _i64
        for which, (name, cls) in enumerate((("gsdd_gemm_desc", GemmDesc), ("gsdd_layer_desc", LayerDesc), ("gsdd_step_desc", StepDesc),
                                             ("gsdd_train_desc", TrainDesc))):
            if L.gsdd_abi_sizeof(which) != C.sizeof(cls):
                raise GsddError(f"{LIB_PATH} was built from another revision of include/gsdd.h: sizeof({name}) is {L.gsdd_abi_sizeof(which)} "
                                f"there and {C.sizeof(cls)} in this binding -- rebuild it with ./build.sh")
        L.gsdd_gemm.argtypes = [C.POINTER(GemmDesc), _p]
        L.gsdd_row_stats.argtypes = [_p, _i64, _i, C.c_float, _p, _p]
        L.gsdd_ncdhw_to_rows.argtypes = [_p, _i, _i, _i, _i, _i, _i, _i, _p, _p]
        L.gsdd_preprocess_clip.argtypes = [_p] + [_i] * 10 + [_p, _p]
        L.gsdd_axial_attention.argtypes = [_p, _i, _i, _i, _i, _i, _i, _p, _i, _p]
        L.gsdd_pool3d.argtypes = [_p, _i, _i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), _i, _i, _i, _i, _p, _i, _p]
        L.gsdd_nearest_code.argtypes = [_p, _i64, _i, _p, _i, _p, _p, _p, _i64, _p]
        L.gsdd_nearest_code_workspace_bytes.argtypes = [_i]
        L.gsdd_nearest_code_workspace_bytes.restype = _i64
        L.gsdd_bn_train_workspace_bytes.argtypes = [_i64, _i]
        L.gsdd_bn_train_workspace_bytes.restype = _i64
        L.gsdd_bn_train.argtypes = [_p, _i64, _i, _p, _p, C.c_float, C.c_float, _p, _p, _p, _p, _p, _p, _i64, _p]
        L.gsdd_conv_wgrad.argtypes = [C.POINTER(GemmDesc), _p, _i, _p, _p]
        L.gsdd_bn_relu_bwd_workspace_bytes.argtypes = [_i64, _i]
        L.gsdd_bn_relu_bwd_workspace_bytes.restype = _i64
        L.gsdd_bn_relu_bwd.argtypes = [_p, _p, _i64, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _p]
        L.gsdd_relu_mask.argtypes = [_p, _p, _p, _i64, _p]
        L.gsdd_lincomb.argtypes = [_p, _p, _p, C.c_float, _p, _i64, _p]
        L.gsdd_axial_attention_bwd.argtypes = [_p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p]
        L.gsdd_codebook_ema.argtypes = [_p, _p, _i64, _i, _i, C.c_float, _p, _p, _p, _p, _p, _p, _p, _i, _p]
        L.gsdd_code_perplexity.argtypes = [_p, _i, _i64, _p, _p]
        L.gsdd_mse.argtypes = [_p, _p, _i64, C.c_float, _p, _p, _i64, _p]
        L.gsdd_d3pm_embed.argtypes = [_p, _i, _i, _i, _p, _i, _p, _i, _p, _p]
        L.gsdd_adaln_table.argtypes = [_p, _i, _i, _p, _p, _p, _p]
        L.gsdd_small_linear.argtypes = [_p, _i, _i, _p, _p, _i, _p, _p]
        L.gsdd_d3pm_attention.argtypes = [_p, _p, _p, _i, _i, _i, _p, _p, _i64, _p, _i, _p]
        L.gsdd_d3pm_attention_workspace_bytes.argtypes = [_i, _i, _i]
        L.gsdd_d3pm_attention_workspace_bytes.restype = _i64
        L.gsdd_d3pm_layer.argtypes = [C.POINTER(LayerDesc), _p]
        L.gsdd_d3pm_layer_pack.argtypes = [_p, _p, _p, _p, _p, _p]
        L.gsdd_d3pm_layer_pack_h2.argtypes = [_p, _p, _p, _p, _p, _p, _p]
        L.gsdd_rows_linear_pack_many.argtypes = [_p, _i, _i, _i, _p]
        L.gsdd_rows_linear.argtypes = [_p, _i64, _i, _p, _i, _p, _p, _i, _p, _p, _i, _p]
        L.gsdd_d3pm_logits.argtypes = [_p, _i64, _i, _p, _p, _p, _p, _i, _p, _p]
        L.gsdd_d3pm_cross_attention.argtypes = [_p, _p, _p, _i, _i, _i, _i, _p, _p]
        L.gsdd_d3pm_step.argtypes = [C.POINTER(StepDesc), _p]
        L.gsdd_d3pm_q_sample.argtypes = [_p, _p, _i, _i, _i, _i, C.POINTER(_p), _p, C.c_uint64, _p, _i64, _p]
        L.gsdd_d3pm_train_loss.argtypes = [C.POINTER(TrainDesc), _p]
        L.gsdd_d3pm_train_loss_bwd.argtypes = [C.POINTER(TrainDesc), _p, _p]
        L.gsdd_d3pm_train_loss_grad.argtypes = [C.POINTER(TrainDesc), _p, _p]
        L.gsdd_gelu2.argtypes = [_p, _p, _p, _i64, _i, _p]
        L.gsdd_ln_fwd.argtypes = [_p, _i64, _i, C.c_float, _p, _p, _p, _i, _i, _p, _p, _p]
        L.gsdd_ln_bwd.argtypes = [_p, _p, _p, _p, _p, _i, _i, _i64, _i, _p, _p, _p, _p, _i, _i, _p]
        L.gsdd_wgrad.argtypes = [_p, _i, _p, _i, _i64, _i, _i, _p, _p, _p]
        L.gsdd_colsum.argtypes = [_p, _i, _i64, _i, _p, _p]
        L.gsdd_batch_rowsum.argtypes = [_p, _i, _i, _i, _p, _p]
        L.gsdd_d3pm_attention_train.argtypes = [_p, _p, _p, _i, _i, _i, _p, _p, _p, _i64, _i, _p]
        L.gsdd_d3pm_attention_bwd.argtypes = [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _i64, _i, _p]
        L.gsdd_d3pm_attention_bwd_workspace_bytes.argtypes = [_i, _i, _i]
        L.gsdd_d3pm_attention_bwd_workspace_bytes.restype = _i64
        L.gsdd_d3pm_embed_bwd.argtypes = [_p, _p, _i, _i, _i, _i, _p, _p, _p]
        L.gsdd_small_linear_bwd.argtypes = [_p, _p, _p, _i, _i, _i, _p, _p, _p, _p]
        L.gsdd_adaln_bwd.argtypes = [_p, _p, _i, _i, _p, _p, _p, _p, _p, _p]
        L.gsdd_adam.argtypes = [_p, _p, _p, _p, _i64, C.c_float, C.c_float, C.c_float, C.c_float, _i, _p]
        L.gsdd_adam_multi.argtypes = [_p, _i, C.c_float, C.c_float, C.c_float, C.c_float, _i, _p]
        L.gsdd_adam_multi_dev.argtypes = [_p, _i, C.c_float, C.c_float, C.c_float, C.c_float, _p, _p]
        L.gsdd_advance.argtypes = [_p, _i, _i64, _p, _i64, _p]
        L.gsdd_philox_uniform.argtypes = [C.c_uint64, _i64, _i64, _i64, _i, _p, _p]
        L.gsdd_graph_begin.argtypes = [_p]
        L.gsdd_graph_end.argtypes = [_p, C.POINTER(_p)]
        L.gsdd_graph_launch.argtypes = [_p, _p]
        L.gsdd_graph_destroy.argtypes = [_p]
        L.gsdd_event_create.argtypes = [C.POINTER(_p)]
        L.gsdd_event_record.argtypes = [_p, _p]
        L.gsdd_event_elapsed_ms.argtypes = [_p, _p, C.POINTER(C.c_float)]
        L.gsdd_event_destroy.argtypes = [_p]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise GsddError(f"gsdd error {rc}: {lib().gsdd_last_error().decode()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must live on the GPU and be contiguous."""
    if t is None:
        return None
    if not t.is_cuda:
        raise GsddError("the HIP path needs tensors on a ROCm device (no CPU fallback)")
    if not t.is_contiguous():
        raise GsddError("non-contiguous tensor handed to the C ABI")
    return C.c_void_p(t.data_ptr())


def stream_ptr(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)
