"""ctypes binding of libctn_hip.so (C ABI declared in include/ctn.h).

Loading fails loudly: there is no CPU or PyTorch fallback behind these calls.
"""
import ctypes as C
import os

# PyTorch-ROCm bundles its own libamdhip64.so.7.  Importing torch first makes the dynamic loader resolve
# this library's dependency to that same runtime instance, so stream handles taken from torch are valid
# here; loading libctn_hip.so first would bring in a second, separate HIP runtime.
import torch  # noqa: F401  (must precede CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CTN_LIB", os.path.join(_HERE, "libctn_hip.so"))   # CTN_LIB: diagnostic builds only

XF_PRELU, XF_NORM, XF_NORMBWD = 1, 2, 4
ARITH_DEFAULT, ARITH_F32, ARITH_BF16X2, ARITH_BF16X3 = 0, 1, 2, 3     # CTN_ARITH_* of include/ctn.h
STAT_SLOTS = 32          # CTN_STAT_SLOTS of include/ctn.h
PIT_NOALLOC_MAX_WORK = 65536   # CTN_PIT_NOALLOC_MAX_WORK of include/ctn.h
EPI_STORE, EPI_STATS, EPI_HEADS, EPI_SIGMOID, EPI_PRELUBWD, EPI_NORMBWD, EPI_ADD = range(7)

_p = C.c_void_p
_i32 = C.c_int32
_i64 = C.c_int64
_f32 = C.c_float
_f64 = C.c_double


class Finish(C.Structure):
    _fields_ = [("count", _p), ("out0", _p), ("out1", _p), ("rstd", _p), ("n", _f64), ("eps", _f32), ("kind", _i32)]


FIN_STATS, FIN_COEFS = 0, 1


class GemmNNArgs(C.Structure):
    _fields_ = [
        ("M", _i32), ("T", _i32), ("K", _i32), ("batch", _i32),
        ("A", _p), ("lda", _i64), ("A2", _p), ("lda2", _i64),
        ("K1", _i32), ("xflags", _i32),
        ("bias", _p), ("bias2", _p),
        ("X", _p), ("ldx", _i64), ("xbs", _i64),
        ("Xb", _p), ("ldxb", _i64), ("xbbs", _i64),
        ("X2", _p), ("ldx2", _i64), ("x2bs", _i64),
        ("alpha", _p),
        ("mean", _p), ("rstd", _p), ("c1", _p), ("c2", _p),
        ("ssb", _i64), ("sst", _i64),
        ("gamma", _p), ("beta", _p),
        ("Y", _p), ("ldy", _i64), ("ybs", _i64),
        ("emode", _i32), ("Mo", _i32),
        ("R", _p), ("ldr", _i64), ("rbs", _i64),
        ("Y2", _p), ("ldy2", _i64), ("y2bs", _i64),
        ("accumulate", _i32), ("tile_width", _i32), ("arith", _i32), ("prep_kernel", _i32),
        ("ealpha", _p), ("esums", _p),
        ("ecol", _p), ("ecol_bs", _i64), ("ecol_ld", _i64),
        ("E", _p), ("lde", _i64), ("ebs", _i64),
        ("emean", _p), ("erstd", _p), ("essb", _i64), ("esst", _i64),
        ("egamma", _p),
        ("A_prep", _p),
        ("fin", Finish),
    ]


class WPrepItem(C.Structure):
    _fields_ = [
        ("src", _p), ("ld", _i64), ("dst", _p),
        ("rows", _i32), ("cols", _i32), ("transpose", _i32), ("k_off", _i32), ("m_off", _i32),
        ("K", _i32), ("M", _i32), ("pad0", _i32),
    ]


class TransposeItem(C.Structure):
    _fields_ = [("src", _p), ("dst", _p), ("ldd", _i64), ("coff", _i64), ("M", _i32), ("K", _i32)]


class GemmNTArgs(C.Structure):
    _fields_ = [
        ("M", _i32), ("N", _i32), ("T", _i32), ("batch", _i32),
        ("A", _p), ("lda", _i64), ("abs_", _i64),
        ("Ab", _p), ("ldab", _i64), ("abbs", _i64),
        ("M1", _i32), ("aflags", _i32),
        ("A2", _p), ("lda2", _i64), ("a2bs", _i64),
        ("a_alpha", _p),
        ("a_mean", _p), ("a_rstd", _p), ("a_c1", _p), ("a_c2", _p),
        ("a_ssb", _i64), ("a_sst", _i64),
        ("a_gamma", _p),
        ("a_rowsum", _p), ("a_rowsum2", _p), ("a_dalpha", _p),
        ("B", _p), ("ldb", _i64), ("bbs", _i64),
        ("bflags", _i32), ("kchunk", _i32), ("arith", _i32), ("pad0", _i32),
        ("b_alpha", _p),
        ("b_mean", _p), ("b_rstd", _p), ("b_ssb", _i64), ("b_sst", _i64),
        ("b_gamma", _p), ("b_beta", _p),
        ("C", _p), ("ldc", _i64), ("C2", _p), ("ldc2", _i64),
        ("a_rowsum_b", _p), ("wdot", _p), ("ldwd", _i64), ("wdot2", _p), ("ldwd2", _i64), ("wdot_sums", _p),
        ("ws", _p), ("ws_floats", _i64),
    ]


class DwFwdArgs(C.Structure):
    _fields_ = [
        ("batch", _i32), ("C", _i32), ("T", _i32), ("P", _i32),
        ("dilation", _i32), ("pad_left", _i32), ("xflags", _i32), ("cln", _i32),
        ("x", _p), ("ldx", _i64), ("xbs", _i64),
        ("alpha_in", _p),
        ("mean", _p), ("rstd", _p), ("ssb", _i64), ("sst", _i64),
        ("gamma", _p), ("beta", _p),
        ("w", _p), ("bias", _p),
        ("y", _p), ("ldy", _i64), ("ybs", _i64),
        ("alpha_out", _p), ("esums", _p),
        ("ecol", _p), ("ecol_ld", _i64),
        ("fin", Finish),
    ]


class DwBwdArgs(C.Structure):
    _fields_ = [
        ("batch", _i32), ("C", _i32), ("T", _i32), ("P", _i32),
        ("dilation", _i32), ("pad_left", _i32), ("xflags", _i32), ("oflags", _i32),
        ("cln", _i32), ("pad0", _i32),
        ("dn2", _p), ("lddn2", _i64), ("dn2bs", _i64),
        ("y2", _p), ("ldy2", _i64), ("y2bs", _i64),
        ("alpha2", _p),
        ("mean2", _p), ("rstd2", _p), ("c1_2", _p), ("c2_2", _p), ("ssb2", _i64), ("sst2", _i64),
        ("gamma2", _p),
        ("dgamma2", _p), ("dbeta2", _p), ("dalpha2", _p),
        ("x", _p), ("ldx", _i64), ("xbs", _i64),
        ("alpha1", _p),
        ("mean1", _p), ("rstd1", _p), ("ssb1", _i64), ("sst1", _i64),
        ("gamma1", _p), ("beta1", _p),
        ("w", _p), ("dw", _p), ("dbias", _p),
        ("dx", _p), ("lddx", _i64), ("dxbs", _i64),
        ("isums", _p), ("icol", _p), ("icol_ld", _i64),
        ("dgamma1", _p), ("dbeta1", _p),
        ("bias", _p), ("recompute_y2", _i32), ("pad1", _i32),
        ("fin", Finish),
    ]


class HeadsBwdArgs(C.Structure):
    _fields_ = [
        ("batch", _i32), ("T", _i32), ("H", _i32), ("M1", _i32), ("M2", _i32),
        ("dY1", _p), ("lddy1", _i64), ("dy1bs", _i64),
        ("dY2", _p), ("lddy2", _i64), ("dy2bs", _i64),
        ("y2", _p), ("ldy2", _i64), ("y2bs", _i64),
        ("alpha", _p),
        ("mean", _p), ("rstd", _p), ("ssb", _i64),
        ("gamma", _p), ("beta", _p),
        ("W_prep", _p),
        ("dn2", _p), ("lddn2", _i64), ("dn2bs", _i64),
        ("esums", _p),
        ("dW1", _p), ("lddw1", _i64),
        ("dW2", _p), ("lddw2", _i64),
        ("rowsum1", _p), ("rowsum2", _p),
        ("ldmin", _i32), ("pad0", _i32),
    ]


# name -> (restype, argtypes); every symbol include/ctn.h declares
PROTOTYPES = {
    "ctn_version": (C.c_char_p, []),
    "ctn_last_error": (C.c_char_p, []),
    "ctn_sizeof_gemm_nn_args": (C.c_int, []),
    "ctn_sizeof_gemm_nt_args": (C.c_int, []),
    "ctn_sizeof_dw_fwd_args": (C.c_int, []),
    "ctn_sizeof_dw_bwd_args": (C.c_int, []),
    "ctn_sizeof_heads_bwd_args": (C.c_int, []),
    "ctn_heads_bwd": (C.c_int, [C.POINTER(HeadsBwdArgs), _p]),
    "ctn_probe_copy": (C.c_int, [_p, _p, _p, _i64, _p]),
    "ctn_probe_stream": (C.c_int, [_p, _p, _i64, C.c_int, _p]),
    "ctn_probe_mfma_f32": (C.c_int, [_p, _p, C.c_int, C.c_int, _p]),
    "ctn_probe_mfma_bf16": (C.c_int, [_p, _p, C.c_int, C.c_int, _p]),
    "ctn_transpose": (C.c_int, [_p, C.c_int, C.c_int, _p, _i64, _i64, _p]),
    "ctn_permute": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, _i64, _i64, C.c_int, _i64, _i64, C.c_int, _i64, _i64, _p]),
    "ctn_sizeof_wprep_item": (C.c_int, []),
    "ctn_sizeof_transpose_item": (C.c_int, []),
    "ctn_transpose_batch": (C.c_int, [C.POINTER(TransposeItem), C.c_int, _p]),
    "ctn_weight_prep_bytes": (_i64, [C.c_int, C.c_int]),
    "ctn_weight_prep": (C.c_int, [C.POINTER(WPrepItem), C.c_int, _p]),
    "ctn_segment": (C.c_int, [_p, _i64, _i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p]),
    "ctn_overlap_add": (C.c_int, [_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p, _i64, _i64, _p]),
    "ctn_rows_gemm": (C.c_int, [_p, _i64, _p, _i64, C.c_int, _p, _i64, C.c_int, C.c_int, _p, _i64, C.c_int, _p]),
    "ctn_rows_wgrad": (C.c_int, [_p, _i64, _p, _i64, _i64, C.c_int, C.c_int, _p, _i64, _p, C.c_float, C.c_int, _p]),
    "ctn_lstm_fwd": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "ctn_lstm_bwd": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p, _p, _p]),
    "ctn_gemm_nn": (C.c_int, [C.POINTER(GemmNNArgs), _p]),
    "ctn_gemm_nt": (C.c_int, [C.POINTER(GemmNTArgs), _p]),
    "ctn_gemm_nt_whole_launches": (C.c_int64, []),
    "ctn_gemm_nt_ws_floats": (C.c_int64, [C.POINTER(GemmNTArgs)]),
    "ctn_norm_stats": (C.c_int, [_p, C.c_int, C.c_int, C.c_int, _i64, _p, _p, _p, _i64, _p]),
    "ctn_gln_finalize": (C.c_int, [_p, C.c_int, _f64, _f32, _p, _p, _p]),
    "ctn_cln_finalize": (C.c_int, [_p, C.c_int, C.c_int, C.c_int, _i64, _f32, _p, _p, _i64, _p]),
    "ctn_norm_apply": (C.c_int, [_p, C.c_int, C.c_int, C.c_int, _i64, _p, _p, _p, _i64, _i64, _p, _p, _p, _i64, _p]),
    "ctn_norm_bwd_reduce": (C.c_int, [_p, _i64, _p, _i64, C.c_int, C.c_int, C.c_int, _p, _p, _p, _i64, _i64, _p,
                                      _p, _p, _i64, _p, _p, _p]),
    "ctn_gln_bwd_coefs": (C.c_int, [_p, _p, C.c_int, _f64, _p, _p, _p]),
    "ctn_heads_bwd_coefs": (C.c_int, [_p, _i64, _p, _i64, C.c_int, C.c_int, C.c_int, _p, _p, _p, _p, _p, C.c_int, _f64,
                                      _p, _p, _p, _i64, _p, _i64, _p, _p, _p]),
    "ctn_cln_bwd_coefs": (C.c_int, [_p, _i64, _p, _p, _i64, C.c_int, C.c_int, C.c_int, _f32, _p, _p, _p]),
    "ctn_norm_bwd_apply": (C.c_int, [_p, _i64, _p, _i64, C.c_int, C.c_int, C.c_int, _p, _p, _p, _p, _p, _i64, _i64,
                                     _p, _p, _i64, _p, _p]),
    "ctn_dw_fwd": (C.c_int, [C.POINTER(DwFwdArgs), _p]),
    "ctn_dw_bwd": (C.c_int, [C.POINTER(DwBwdArgs), _p]),
    "ctn_encoder_fwd": (C.c_int, [_p, _i64, C.c_int, C.c_int, C.c_int, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p,
                                  C.c_int, _i64, _p, _p, _i64, _p]),
    "ctn_maskdecode_fwd": (C.c_int, [_p, _i64, _p, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _p, C.c_int, C.c_int,
                                     _p, _i64, _i64, C.c_int, C.c_int, _p]),
    "ctn_maskdecode_bwd": (C.c_int, [_p, _i64, _i64, C.c_int, C.c_int, _p, _i64, _p, _i64, C.c_int, C.c_int,
                                     C.c_int, C.c_int, _p, C.c_int, C.c_int, C.c_int, _p, _i64, _p, _i64, _p, _p]),
    "ctn_encoder_bwd": (C.c_int, [_p, _i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _p,
                                  C.c_int, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _p, _p,
                                  _i64, _p]),
    "ctn_pit_sisdr_fwd": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, _i64, _i64, _f32, C.c_int, _p, _p, _p, _p, _p,
                                    _p]),
    "ctn_pit_sisdr_fwd_noalloc": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, _i64, _i64, _f32, C.c_int, _p, _p, _p, _p,
                                            _p]),
    "ctn_pit_sisdr_bwd": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, _i64, _i64, _p, _p, _p, C.c_int, C.c_int, _p,
                                    _i64, _i64, _p]),
    "ctn_dwconv_generic_fwd": (C.c_int, [_p, _i64, _i64, C.c_int, C.c_int, C.c_int, _p, _p, C.c_int, C.c_int, C.c_int,
                                         C.c_int, _p, _i64, _i64, C.c_int, _p]),
    "ctn_dwconv_generic_bwd": (C.c_int, [_p, _i64, _i64, C.c_int, C.c_int, C.c_int, _p, C.c_int, C.c_int, C.c_int,
                                         C.c_int, _p, _i64, _i64, C.c_int, _p, _i64, _i64, _p, _p, _p]),
    "ctn_pair_sdr_fwd": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, _i64, _i64, _f32, C.c_int, _p, _p, _p, _p]),
    "ctn_pair_sdr_bwd": (C.c_int, [_p, _p, C.c_int, C.c_int, C.c_int, _i64, _i64, _p, _p, _p, _i64, _i64, _p]),
    "ctn_sumsq": (C.c_int, [_p, _i64, _p, _p]),
    "ctn_adam_step": (C.c_int, [_p, _p, _p, _p, _i64, _p, _f32, _f32, _f32, _f32, _f32, _f32, C.c_int, _p]),
}

_STRUCT_SIZES = {
    "ctn_sizeof_gemm_nn_args": GemmNNArgs,
    "ctn_sizeof_gemm_nt_args": GemmNTArgs,
    "ctn_sizeof_dw_fwd_args": DwFwdArgs,
    "ctn_sizeof_dw_bwd_args": DwBwdArgs,
    "ctn_sizeof_heads_bwd_args": HeadsBwdArgs,
    "ctn_sizeof_wprep_item": WPrepItem,
    "ctn_sizeof_transpose_item": TransposeItem,
}

_lib = None


class CtnError(RuntimeError):
    pass


def load():
    """Load the HIP library (once) and bind every prototype.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CtnError("{} not found: build it with `python -m speech_amd.build` (hipcc, gfx950). "
                       "There is no CPU fallback for these operators.".format(LIB_PATH))
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)       # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    for name, struct in _STRUCT_SIZES.items():
        want = getattr(lib, name)()
        if want != C.sizeof(struct):
            raise CtnError("{}: library says {} bytes, binding has {}".format(name, want, C.sizeof(struct)))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ctn_last_error().decode()
        raise CtnError("{} failed (rc={}): {}".format(what, rc, msg))
