"""ctypes binding of libmural_hip.so (the C ABI declared in include/mural_hip.h).

The product has NO CPU fallback: if the shared library is missing or a tensor is not on a HIP device the
call fails loudly.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` (or ``make -C
mural_amd/csrc``).
"""
import ctypes as C
import os

import torch  # noqa: F401  -- must be loaded BEFORE the extension so both share torch's HIP runtime (libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmural_hip.so")

MURAL_OK, MURAL_E_INVALID, MURAL_E_RUNTIME, MURAL_E_WORKSPACE, MURAL_E_ENCODING = 0, 1, 2, 3, 4

_f32p = C.POINTER(C.c_float)


class MuralGenome(C.Structure):
    _fields_ = [("packed2", C.c_void_p), ("nmask", C.c_void_p), ("length", C.c_int64), ("amb_pos", C.c_void_p),
                ("amb_sym", C.c_void_p), ("n_amb", C.c_int64)]


class MuralBN(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p)]


class MuralAffine(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p)]


class MuralResBlock(C.Structure):
    _fields_ = [("bn1", MuralBN), ("conv1", MuralAffine), ("bn2", MuralBN), ("conv2", MuralAffine)]


class MuralTower(C.Structure):
    _fields_ = [("bn_in", MuralBN), ("conv_in", MuralAffine), ("rbs1", MuralResBlock * 2),
                ("bn_mid", MuralBN), ("conv_mid", MuralAffine), ("rbs2", MuralResBlock * 2),
                ("bn_out", MuralBN), ("conv_out", MuralAffine), ("fc_bn", MuralBN), ("fc", MuralAffine)]


class MuralLocal(C.Structure):
    _fields_ = [("emb", C.c_void_p), ("lin", MuralAffine * 2), ("bn", MuralBN * 2), ("out", MuralAffine)]


class MuralConvBN(C.Structure):
    _fields_ = [("conv", MuralAffine), ("bn", MuralBN)]


class MuralConvBlock(C.Structure):
    _fields_ = [("conv5_w", C.c_void_p), ("bn1", MuralBN), ("conv1_w", C.c_void_p), ("bn2", MuralBN)]


class MuralIndelShape(C.Structure):
    _fields_ = [("n_class", C.c_int32), ("channels", C.c_int32), ("ksize", C.c_int32), ("down", C.c_int32 * 6),
                ("use_reverse", C.c_int32), ("length", C.c_int32), ("bn_eps", C.c_float)]


class MuralIndelParams(C.Structure):
    _fields_ = [("sym", MuralConvBN), ("up_l", MuralConvBN * 6), ("up_b", MuralConvBlock * 6),
                ("down_l", MuralConvBN * 5), ("down_b", MuralConvBlock * 5), ("out1", MuralAffine), ("out_bn", MuralBN),
                ("out2", MuralAffine), ("fc_bn", MuralBN), ("fc", MuralAffine)]


class MuralRelayoutJob(C.Structure):
    _fields_ = [("W", C.c_void_p), ("wt_fwd", C.c_void_p), ("wt_dgrad", C.c_void_p), ("Cout", C.c_int32), ("Cin", C.c_int32),
                ("K", C.c_int32), ("reserved", C.c_int32), ("start", C.c_int64)]


class MuralTsvRows(C.Structure):
    _fields_ = [("chrom_names", C.c_char_p), ("n_chroms", C.c_int32), ("name_stride", C.c_int32), ("chrom_id", C.c_void_p),
                ("start", C.c_void_p), ("end", C.c_void_p), ("strand", C.c_void_p), ("label", C.c_void_p), ("prob", C.c_void_p),
                ("prob_f64", C.c_int32), ("n_class", C.c_int32), ("prob_stride", C.c_int64), ("perm", C.c_void_p), ("n", C.c_int64),
                ("layout", C.c_int32), ("reserved", C.c_int32)]


class MuralSnvShape(C.Structure):
    _fields_ = [("model_no", C.c_int32), ("n_class", C.c_int32), ("local_cols", C.c_int32), ("emb_rows", C.c_int32),
                ("hidden1", C.c_int32), ("hidden2", C.c_int32), ("channels", C.c_int32), ("ksize", C.c_int32),
                ("distal_len", C.c_int32), ("bn_eps", C.c_float)]


class MuralSnvParams(C.Structure):
    _fields_ = [("local", MuralLocal), ("mid", MuralTower), ("large", MuralTower)]


VP, I32, I64 = C.c_void_p, C.c_int32, C.c_int64

# every symbol include/mural_hip.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    "mural_op_relayout": (C.c_int, [VP, VP, I32, I32, I32, I32, VP]),
    "mural_op_conv1d": (C.c_int, [VP, VP, VP, VP, I64, I32, I32, I32, I32, VP, VP, I32, I32, VP, VP, VP]),
    "mural_op_bn_stats": (C.c_int, [VP, I64, I32, I32, I32, VP, VP]),
    "mural_op_bn_finalize": (C.c_int, [VP, C.c_double, I32, VP, VP, C.c_float, C.c_float, VP, VP, VP, VP, VP, VP, VP]),
    "mural_op_bn_apply": (C.c_int, [VP, I64, I32, I32, I32, VP, VP, VP, VP]),
    "mural_op_bn_backward": (C.c_int, [VP, VP, I64, I32, I32, I32, VP, VP, VP, VP, I32, VP, VP, VP, VP, VP, VP]),
    "mural_op_conv_wgrad": (C.c_int, [VP, VP, I64, I32, I32, I32, VP, VP, I32, VP, VP, VP, C.c_size_t, VP]),
    "mural_op_conv32_supported": (C.c_int, [I32]),
    "mural_op_conv32": (C.c_int, [VP, VP, VP, VP, I64, I32, I32, VP, VP, I32, I32, VP, VP, I32, I32, VP, VP, VP, VP, VP]),
    "mural_op_conv32_wgrad_scratch": (C.c_size_t, []),
    "mural_op_conv32_wgrad": (C.c_int, [VP, VP, I64, I32, VP, VP, I32, VP, VP, VP, C.c_size_t, VP]),
    "mural_op_maxpool_fwd": (C.c_int, [VP, I64, I32, I32, I32, I32, VP, VP, VP]),
    "mural_op_maxpool_bwd_needs_zero": (C.c_int, [I32, I32]),
    "mural_op_maxpool_bwd": (C.c_int, [VP, VP, I64, I32, I32, I32, I32, I32, VP, VP]),
    "mural_fasta_scan": (C.c_int, [C.c_char_p, I64, I32, VP, VP, VP, VP]),
    "mural_fasta_pack": (C.c_int, [C.c_char_p, I64, I64, VP, VP, VP, VP, I64, VP]),
    "mural_bed_read": (C.c_int, [C.c_char_p, I64, VP, VP, VP, VP, VP, I32, I32, VP, VP, VP]),
    "mural_bed_segment_order": (C.c_int, [VP, VP, VP, I64, I64, VP, VP, VP]),
    "mural_host_dense_to_symbols": (C.c_int, [VP, VP, I64, I32, VP, VP]),
    "mural_host_concat": (C.c_int, [VP, VP, I64, VP]),
    "mural_snv_forward_symbols": (C.c_int, [VP, VP, VP, I64, VP, VP, C.c_size_t, VP]),
    "mural_bed_index_scan": (C.c_int, [C.c_char_p, I64, I64, I64, I32, I64, VP, VP, VP, VP, VP, VP, VP, VP]),
    "mural_bed_parse_range": (C.c_int, [C.c_char_p, I64, I64, I64, I64, C.c_char_p, VP, VP, VP, VP]),
    "mural_tsv_row_bound": (C.c_int64, [C.POINTER(MuralTsvRows)]),
    "mural_tsv_format_workspace_bytes": (C.c_size_t, [I64]),
    "mural_tsv_format_device": (C.c_int, [C.POINTER(MuralTsvRows), VP, I64, VP, VP, C.c_size_t, VP]),
    "mural_tsv_format_host": (C.c_int, [C.POINTER(MuralTsvRows), VP, I64, VP, I32]),
    "mural_tsv_format_g4": (C.c_int, [C.c_double, VP]),
    "mural_focal_group_check": (C.c_int, [VP, I32, I64, I64, VP, I64, VP, VP]),
    "mural_eval_kmer_keys": (C.c_int, [VP, I64, I32, I32, I32, I32, I64, I64, VP, VP, VP]),
    "mural_eval_window_keys": (C.c_int, [VP, VP, I64, I64, VP, I32, VP, VP, VP]),
    "mural_eval_group_obs_pred": (C.c_int, [VP, VP, VP, I32, I64, I32, I32, VP, VP, VP]),
    "mural_eval_calib_metrics": (C.c_int, [VP, I32, VP, I64, I32, I32, VP, VP, VP, VP]),
    "mural_eval_dirichlet_fit_terms": (C.c_int, [VP, I32, VP, I64, I32, VP, I32, VP, VP, VP]),
    "mural_op_convg_out_length": (C.c_int, [I32, I32, I32, I32, I32]),
    "mural_op_convg_fwd": (C.c_int, [VP, VP, VP, VP, VP, I64, I32, I32, I32, I32, I32, I32, I32, VP]),
    "mural_op_convg_bwd_scratch": (C.c_size_t, [I32, I32, I32]),
    "mural_op_convg_bwd": (C.c_int, [VP, VP, VP, I64, I32, I32, I32, I32, I32, I32, I32, VP, VP, VP, VP, C.c_size_t, VP]),
    "mural_op_convg_bn_fwd": (C.c_int, [VP, VP, VP, VP, VP, I64, I32, I32, I32, I32, I32, I32, I32, VP, VP, C.c_float, C.c_float, VP, VP, VP,
                                        VP, I32, VP, VP, VP, VP]),
    "mural_op_convg_bn_bwd": (C.c_int, [VP, VP, VP, VP, VP, VP, I64, I32, I32, I32, I32, I32, I32, I32, I32, VP, VP, VP, VP, VP, VP, VP, VP,
                                        C.c_size_t, VP, VP]),
    "mural_op_relayout_multi": (C.c_int, [VP, I32, I64, VP]),
    "mural_op_act_fwd": (C.c_int, [VP, I64, I32, VP, VP]),
    "mural_op_act_bwd": (C.c_int, [VP, VP, I64, I32, VP, VP]),
    "mural_op_conv32_bwd": (C.c_int, [VP, VP, VP, I64, I32, VP, VP, I32, VP, VP, VP, VP, VP, VP, VP, C.c_size_t, VP]),
    "mural_op_bnconv32_fwd": (C.c_int, [VP, I64, I32, I32, VP, I32, VP, VP, C.c_float, C.c_float, VP, VP, VP, VP, VP, I32, VP, VP, VP, I32, VP, VP]),
    "mural_op_bnconv32_bwd": (C.c_int, [VP, VP, I64, I32, I32, VP, VP, VP, VP, VP, C.c_size_t, VP, VP, VP, VP, VP, VP, VP, VP, VP]),
    "mural_op_first_plan": (C.c_int, [I32, I32, VP, VP, VP]),
    "mural_op_first_fwd": (C.c_int, [VP, I64, I32, I32, I32, I32, I32, I32, I32, VP, VP, VP, VP, C.c_float, C.c_float, VP, VP, VP, VP, VP, VP, VP]),
    "mural_op_first_bwd": (C.c_int, [VP, VP, VP, I64, I32, I32, I32, I32, I32, I32, I32, VP, VP, VP, VP, VP, VP, VP, VP]),
    "mural_op_linear_fwd": (C.c_int, [VP, VP, VP, I64, I32, I32, VP, VP]),
    "mural_op_linear_bwd": (C.c_int, [VP, VP, VP, I64, I32, I32, VP, VP, VP, VP]),
    "mural_op_embedding_fwd": (C.c_int, [VP, VP, I64, I32, I32, VP, VP]),
    "mural_op_embedding_bwd": (C.c_int, [VP, VP, I64, I32, I32, VP, VP]),
    "mural_op_dropout": (C.c_int, [VP, I64, C.c_float, C.c_uint64, VP, VP, VP]),
    "mural_op_relu_mask": (C.c_int, [VP, VP, I64, VP, VP]),
    "mural_op_head_fwd": (C.c_int, [VP, VP, VP, I64, I32, VP, VP]),
    "mural_op_head_bwd": (C.c_int, [VP, VP, VP, VP, I64, I32, VP, VP, VP, VP]),
    "mural_op_dense_to_symbols": (C.c_int, [VP, I64, I32, VP, VP, VP]),
    "mural_calibrate_rows": (C.c_int, [VP, I64, I32, I32, VP, I32, C.c_double, VP, I32, VP]),
    "mural_snv_train_workspace_bytes": (C.c_size_t, [C.POINTER(MuralSnvShape), I64]),
    "mural_snv_train_forward": (C.c_int, [C.POINTER(MuralSnvShape), C.POINTER(MuralSnvParams), VP, VP, VP, I64, VP, VP, VP, C.c_float,
                                          VP, VP, C.c_size_t, VP, VP]),
    "mural_snv_train_backward": (C.c_int, [C.POINTER(MuralSnvShape), C.POINTER(MuralSnvParams), C.POINTER(MuralSnvParams), VP, VP, I64,
                                           VP, VP, VP, VP, C.c_size_t, VP]),
    "mural_op_ce_sum_fwd": (C.c_int, [VP, VP, I64, I32, VP, VP, VP]),
    "mural_op_ce_sum_bwd": (C.c_int, [VP, VP, VP, I64, I32, VP, VP]),
    "mural_op_clip_grad_norm": (C.c_int, [VP, I64, C.c_float, VP, VP, VP]),
    "mural_op_adam_flat": (C.c_int, [VP, VP, VP, VP, I64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, I64, VP]),
    "mural_last_error": (C.c_char_p, []),
    "mural_abi_version": (C.c_int, []),
    "mural_encode_kmer": (C.c_int, [C.POINTER(MuralGenome), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_void_p, C.c_void_p]),
    "mural_encode_onehot": (C.c_int, [C.POINTER(MuralGenome), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_void_p]),
    "mural_encode_symbols": (C.c_int, [C.POINTER(MuralGenome), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_void_p]),
    "mural_snv_model_create": (C.c_int, [C.POINTER(MuralSnvShape), C.POINTER(MuralSnvParams), C.POINTER(C.c_void_p)]),
    "mural_snv_model_destroy": (None, [C.c_void_p]),
    "mural_snv_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int64, C.c_int32]),
    "mural_snv_workspace_bytes_min": (C.c_size_t, [C.c_void_p, C.c_int64, C.c_int32]),
    "mural_snv_forward_dense": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                          C.c_size_t, C.c_void_p, C.c_void_p]),
    "mural_snv_forward_packed": (C.c_int, [C.c_void_p, C.POINTER(MuralGenome), C.c_void_p, C.c_void_p, C.c_int64,
                                           C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mural_snv_forward_front": (C.c_int, [C.c_void_p, C.POINTER(MuralGenome), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mural_snv_forward_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mural_snv_reuse_supported": (C.c_int, [C.c_void_p]),
    "mural_snv_reuse_chunk_span": (C.c_int64, []),
    "mural_snv_reuse_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32]),
    "mural_snv_forward_packed_reuse": (C.c_int, [C.c_void_p, C.POINTER(MuralGenome), C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                                 C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                                                 C.c_void_p]),
    "mural_snv_debug_taps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "mural_snv_tap_layout": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "mural_indel_model_create": (C.c_int, [C.POINTER(MuralIndelShape), C.POINTER(MuralIndelParams), C.POINTER(C.c_void_p)]),
    "mural_indel_model_destroy": (None, [C.c_void_p]),
    "mural_indel_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int64]),
    "mural_indel_forward_dense": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t,
                                            C.c_void_p]),
    "mural_indel_forward_packed": (C.c_int, [C.c_void_p, C.POINTER(MuralGenome), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                             C.c_void_p, C.c_size_t, C.c_void_p]),
    "mural_indel_train_workspace_bytes": (C.c_size_t, [C.POINTER(MuralIndelShape), I64]),
    "mural_indel_train_forward": (C.c_int, [C.POINTER(MuralIndelShape), C.POINTER(MuralIndelParams), VP, I64, C.c_float, C.c_uint64, VP,
                                            C.c_float, VP, VP, C.c_size_t, VP]),
    "mural_indel_train_backward": (C.c_int, [C.POINTER(MuralIndelShape), C.POINTER(MuralIndelParams), C.POINTER(MuralIndelParams), VP, VP,
                                             I64, C.c_float, C.c_uint64, VP, VP, C.c_size_t, VP]),
    "mural_snv_kernel_name": (C.c_char_p, []),
    "mural_profile_begin": (C.c_int, []),
    "mural_profile_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

# the validation hooks / diagnostics of include/mural_hip_debug.h: exported by the debug flavour only (libmural_hip_debug.so)
DEBUG_PROTOTYPES = {
    "mural_debug_last_ws_layout": (C.c_int, [VP, I32]),
    "mural_debug_cl_conv32_fwd": (C.c_int, [VP, I64, I32, I32, VP, VP, VP, VP, VP, VP, VP, VP, I32, VP, VP, VP, I32, VP, VP]),
    "mural_debug_cl_conv32_bwd": (C.c_int, [VP, VP, VP, I64, I32, VP, I32, VP, VP, VP, VP, VP]),
    "mural_debug_cw_wfrag": (C.c_int, [VP, VP, VP]),
    "mural_debug_cw_set_stamps": (C.c_int, [VP]),
    "mural_debug_first_set_stamps": (C.c_int, [VP]),
    "mural_debug_lt_set_stamps": (C.c_int, [VP]),
    "mural_debug_cw_conv32_fwd": (C.c_int, [VP, I64, I32, I32, VP, VP, VP, VP, VP, VP, VP, VP, I32, VP, VP, VP, I32, VP, VP, VP]),
    "mural_debug_cw_conv32_bwd": (C.c_int, [VP, VP, VP, I64, I32, VP, VP, I32, VP, VP, VP, VP, VP, VP]),
    "mural_debug_cl_bn_stats": (C.c_int, [VP, I64, I32, VP, VP]),
    "mural_debug_conv1d": (C.c_int, [VP, VP, VP, VP, I64, I32, I32, I32, I32, I32, I32, I32, I32, VP, VP, I32, VP]),
    "mural_debug_conv1d_set_stamps": (C.c_int, [VP]),
    "mural_debug_poison_lds": (C.c_int, [VP]),
    "mural_debug_convblock": (C.c_int, [VP, VP, VP, VP, VP, VP, VP, I64, I32, I32, VP, VP, VP, I32, I32, VP, VP, VP, VP, VP, VP, I32, VP]),
    "mural_debug_cb8_set_stamps": (C.c_int, [VP]),
    "mural_debug_set_stamps": (C.c_int, [C.c_void_p]),
    "mural_debug_list_switches": (C.c_int, [C.c_char_p, C.c_size_t]),
}

_lib = None
DEBUG_LIB_PATH = os.path.join(_HERE, "libmural_hip_debug.so")


def flavor():
    """'debug' when MURAL_HIP_FLAVOR=debug is in the environment at the first call of lib(): the library with the validation hooks of
    include/mural_hip_debug.h, the only one that honours the development switches (csrc/common.h: dev_env).  tests/conftest.py and the
    tools set it; bench.py, __graft_entry__.smoke() and every user of the package run the product library."""
    return "debug" if os.environ.get("MURAL_HIP_FLAVOR", "") == "debug" else "product"


def lib():
    """Load (once) and return the shared library with typed prototypes; raises if it is not built."""
    global _lib
    if _lib is None:
        debug = flavor() == "debug"
        path = DEBUG_LIB_PATH if debug else LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: the HIP extension has not been built. "
                "Run `make -C mural_amd/csrc` (needs hipcc, --offload-arch=gfx950). There is no CPU fallback.")
        handle = C.CDLL(path)
        protos = dict(PROTOTYPES, **DEBUG_PROTOTYPES) if debug else PROTOTYPES
        for name, (res, args) in protos.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc == MURAL_OK:
        return
    msg = lib().mural_last_error().decode("utf-8", "replace")
    if rc == MURAL_E_INVALID:
        raise ValueError(msg)
    raise RuntimeError(f"libmural_hip error {rc}: {msg}")


def require_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a HIP device (got {t.device}); mural_amd has no CPU path")
    return t


def current_stream_ptr(device):
    """Raw hipStream_t of torch's current stream on `device` (one C call: this sits on every kernel launch path)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(idx))
