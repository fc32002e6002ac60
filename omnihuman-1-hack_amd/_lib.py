"""ctypes binding of libomh.so — the C ABI declared in include/omh.h.

The product path has no CPU fallback: if the library cannot be loaded (and
cannot be built because hipcc is absent) importing this module raises.
"""
import ctypes as C
import os

# PyTorch must load ITS HIP runtime (libamdhip64) first: libomh.so then binds to
# that same runtime instance by soname.  Loading libomh.so before torch would
# pull in a second copy from /opt/rocm with no device/stream state in common
# (every launch then fails with hipErrorNoDevice).
import torch  # noqa: F401  (import order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
# OMH_LIB: another build of the same ABI (A/B timing of kernel revisions on one box); normally unset
_LIB_PATH = os.environ.get("OMH_LIB") or os.path.join(_HERE, "lib", "libomh.so")


class OmhError(RuntimeError):
    pass


ABI_VERSION = 12         # == OMH_ABI_VERSION of include/omh.h; checked against the loaded library below


def _load():
    if not os.environ.get("OMH_LIB"):
        # (re)build in-tree whenever the sources changed (hipcc cross-compiles without a GPU).  build() is a hash
        # check when nothing changed, takes a file lock so that the ranks of one torchrun do not race, and swaps
        # the library in atomically.  Without hipcc (a box that only received the prebuilt .so) a stale or missing
        # library is an error, not something to paper over.
        import importlib.util
        spec = importlib.util.spec_from_file_location("_omh_build", os.path.join(_HERE, "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if mod.have_hipcc():
            mod.build(verbose=False)
        elif not os.path.exists(_LIB_PATH):
            raise OmhError(f"{_LIB_PATH} is missing and hipcc is not available to build it")
        elif not mod.up_to_date():
            raise OmhError(f"{_LIB_PATH} does not match csrc/ (stale build) and hipcc is not available to rebuild it")
    try:
        return C.CDLL(_LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise OmhError(f"cannot load the HIP kernel library {_LIB_PATH}: {e}") from e


lib = _load()

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp),
                ("M", i32), ("N", i32), ("K", i32),
                ("lda", i32), ("ldb", i32), ("ldc", i32),
                ("batch", i32),
                ("strideA", i64), ("strideB", i64), ("strideC", i64),
                ("epilogue", i32), ("bias_mode", i32),
                ("bias", vp),
                ("gate0", vp), ("gate1", vp),
                ("gate1_stride", i64), ("gate_rows", i32), ("gate_const", f32),
                ("b_kmajor", i32),
                ("c_in", vp), ("aux", vp), ("ldaux", i32),
                ("workspace", vp), ("workspace_bytes", i64), ("n_split", i32)]


COLSUM_MAX = 16


class ColsumBatch(C.Structure):
    _fields_ = [("n", i32), ("x", vp * COLSUM_MAX), ("out", vp * COLSUM_MAX), ("ld", i64 * COLSUM_MAX),
                ("R", i64 * COLSUM_MAX), ("C", i32 * COLSUM_MAX), ("is_bf16", i32 * COLSUM_MAX),
                ("blocks", i32 * COLSUM_MAX)]


class GemmTnArgs(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("M", i32), ("N", i32), ("K", i32),
                ("lda", i32), ("ldb", i32), ("ldc", i32), ("accumulate", i32)]


TN_GROUP_MAX = 12


class GemmTnGroup(C.Structure):
    _fields_ = [("n", i32), ("problem", GemmTnArgs * TN_GROUP_MAX), ("first_tile", i32 * TN_GROUP_MAX), ("total_tiles", i32)]


ATTN_SHORT_KERNEL, ATTN_ALLOW_SPLIT = 1, 2       # omh_attn_args.flags (ABI v8)


class AttnArgs(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("vt", vp), ("o", vp),
                ("k_lens", vp),
                ("B", i32), ("H", i32), ("Lq", i32), ("Lk", i32),
                ("q_bs", i64), ("q_rs", i64), ("k_bs", i64), ("k_rs", i64),
                ("vt_bs", i64), ("o_bs", i64), ("o_rs", i64),
                ("ldv", i32), ("scale", f32), ("lse", vp), ("q_prescaled", i32), ("workspace", vp), ("workspace_bytes", i64),
                ("o32", vp), ("flags", i32), ("q_lens", vp), ("window_left", i32), ("window_right", i32)]

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        if len(a) < 26 and "window_left" not in k:      # unbounded band unless the caller says otherwise (0 is a bound)
            self.window_left = -1
        if len(a) < 27 and "window_right" not in k:
            self.window_right = -1


class AttnBwdArgs(C.Structure):
    _fields_ = [("q", vp), ("k", vp), ("v", vp), ("o", vp), ("dout", vp),
                ("qt", vp), ("dot", vp), ("kt", vp),
                ("lse", vp), ("delta", vp),
                ("dq", vp), ("dk", vp), ("dv", vp),
                ("k_lens", vp),
                ("B", i32), ("H", i32), ("Lq", i32), ("Lk", i32),
                ("q_bs", i64), ("q_rs", i64), ("k_bs", i64), ("k_rs", i64), ("o_bs", i64), ("o_rs", i64),
                ("dq_bs", i64), ("dq_rs", i64), ("dk_bs", i64), ("dk_rs", i64), ("qt_bs", i64), ("kt_bs", i64),
                ("ldq", i32), ("ldk", i32), ("scale", f32), ("q_prescaled", i32), ("out_bf16", i32), ("o32", vp),
                ("phase", i32), ("workspace", vp), ("workspace_bytes", i64)]


class PartialReduce(C.Structure):
    _fields_ = [("part", vp), ("nj", i32), ("np", i32), ("nb", i32), ("dim", i32), ("grid_y", i32),
                ("out", vp * 3), ("stride", i64 * 3)]


PARTIAL_REDUCE_MAX = 8


class PartialReduceBatch(C.Structure):
    _fields_ = [("n", i32), ("e", PartialReduce * PARTIAL_REDUCE_MAX)]


class LnBwdArgs(C.Structure):
    _fields_ = [("x", vp), ("dy", vp), ("dy_bf16", i32), ("dx", vp),
                ("rows", i64), ("dim", i32), ("eps", f32), ("mul_const", f32),
                ("mul0", vp), ("mul1", vp), ("mul1_stride", i64),
                ("dmul", vp), ("dadd", vp), ("dstride", i64), ("rows_per_batch", i64),
                ("dy_next", vp), ("y_next", vp), ("gate_const", f32), ("gate0", vp), ("gate1", vp), ("gate1_stride", i64),
                ("dgate", vp), ("dgate_stride", i64),
                ("workspace", vp), ("workspace_floats", i64), ("deferred", C.POINTER(PartialReduce))]


class RmsBwdArgs(C.Structure):
    _fields_ = [("x", vp), ("x_bf16", i32), ("ldx", i64), ("dy", vp), ("dy_bf16", i32), ("lddy", i64), ("dx", vp), ("lddx", i64),
                ("n_seg", i32), ("seg_x", i64), ("seg_dy", i64), ("seg_dx", i64),
                ("weight", vp * 2), ("dweight", vp * 2),
                ("rows", i64), ("dim", i32), ("eps", f32), ("do_norm", i32),
                ("rope_cos", vp), ("rope_sin", vp), ("rope_len", i32), ("head_dim", i32), ("grid", vp), ("seq_len", i32),
                ("workspace", vp), ("workspace_floats", i64), ("deferred", C.POINTER(PartialReduce))]


class ConvArgs(C.Structure):
    _fields_ = [("x", vp), ("w", vp), ("bias", vp), ("resid", vp), ("y", vp),
                ("Tin", i32), ("Hin", i32), ("Win", i32), ("Cin", i32),
                ("Tout", i32), ("Hout", i32), ("Wout", i32), ("Cout", i32),
                ("KT", i32), ("KH", i32), ("KW", i32),
                ("stride_t", i32), ("stride_hw", i32), ("pad_h", i32), ("pad_w", i32),
                ("up2", i32), ("out_f32", i32), ("split_n", i32), ("resid_f32", i32),
                ("norm_gamma", vp), ("norm_out", vp), ("norm_only", i32), ("pair", i32)]


EPI_BF16, EPI_F32, EPI_GELU_BF16, EPI_RESID, EPI_F32_ACCUM, EPI_GELU_ERF_BF16, EPI_GELU_BWD_BF16, EPI_BF16_SPLIT_T = 0, 1, 2, 3, 4, 5, 6, 7
BIAS_NONE, BIAS_N, BIAS_M = 0, 1, 2

_SIGS = {
    "omh_abi_version": (i32, []),
    "omh_set_deterministic": (i32, [i32]),
    "omh_build_arch": (C.c_char_p, []),
    "omh_set_option": (i32, [C.c_char_p, C.c_char_p]),
    "omh_get_option": (C.c_char_p, [C.c_char_p]),
    "omh_option_count": (i32, []),
    "omh_option_name": (C.c_char_p, [i32]),
    "omh_gemm_bf16": (i32, [C.POINTER(GemmArgs), vp]),
    "omh_gemm_workspace_bytes": (i64, [C.POINTER(GemmArgs)]),
    "omh_gemm_bf16_tn": (i32, [C.POINTER(GemmTnArgs), vp]),
    "omh_gemm_bf16_tn_grouped": (i32, [C.POINTER(GemmTnGroup), vp]),
    "omh_flash_attn_fwd_d128": (i32, [C.POINTER(AttnArgs), vp]),
    "omh_flash_attn_workspace_bytes": (i64, [C.POINTER(AttnArgs)]),
    "omh_flash_attn_bwd_d128": (i32, [C.POINTER(AttnBwdArgs), vp]),
    "omh_flash_attn_bwd_workspace_bytes": (i64, [C.POINTER(AttnBwdArgs)]),
    "omh_layernorm_modulate": (i32, [vp, vp, i64, i32, f32, f32, vp, vp, i64, vp, vp, i64, i64, vp]),
    "omh_rmsnorm_rope": (i32, [vp, i64, vp, i64, i32, vp, f32, i32, vp, vp, i32, i32, vp, i32, vp]),
    "omh_rmsnorm_rope_bf16": (i32, [vp, i64, vp, i64, i32, vp, f32, i32, vp, vp, i32, i32, vp, i32, f32, vp]),
    "omh_rmsnorm_rope_bf16_pair": (i32, [vp, i64, i64, vp, vp, i64, i32, vp, vp, f32, i32, vp, vp, i32, i32, vp, i32, f32, f32, vp]),
    "omh_cast_f32_bf16": (i32, [vp, vp, i64, vp]),
    "omh_patchify": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "omh_unpatchify": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "omh_dense_f32": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "omh_sinusoidal_embedding": (i32, [vp, vp, i32, i32, vp]),
    "omh_conv_cl_bf16": (i32, [C.POINTER(ConvArgs), vp]),
    "omh_rms_silu_cl": (i32, [vp, vp, vp, i64, i32, i32, vp]),
    "omh_rms_silu_cl_f32in": (i32, [vp, vp, vp, i64, i32, i32, vp]),
    "omh_relu_bf16": (i32, [vp, i64, vp]),
    "omh_relu_bwd_bf16": (i32, [vp, vp, vp, i64, vp]),
    "omh_nchw_to_cl": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp]),
    "omh_cl_to_nchw": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, f32, f32, i32, i32, vp]),
    "omh_softmax_rows": (i32, [vp, i64, vp, i64, i64, i32, f32, vp]),
    "omh_softmax_rows_f32": (i32, [vp, i64, vp, i64, i64, i32, f32, vp]),
    "omh_split3_f32": (i32, [vp, i64, vp, i64, i64, i32, i32, i32, vp]),
    "omh_rms_silu_cl_split3": (i32, [vp, vp, vp, i64, i32, i32, vp]),
    "omh_rms_silu_cl_pair": (i32, [vp, vp, vp, i64, i32, i32, vp]),
    "omh_conv_pair_supported": (i32, [C.POINTER(ConvArgs)]),
    "omh_nchw_to_cl_f32": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, i32, vp]),
    "omh_transpose_bf16": (i32, [vp, vp, i32, i32, i64, i64, i32, i64, i64, vp]),
    "omh_colsum_accum": (i32, [vp, i32, i64, vp, i64, i32, vp]),
    "omh_colsum_accum_multi": (i32, [vp, vp]),
    "omh_gelu_tanh_bf16": (i32, [vp, vp, i64, vp]),
    "omh_gelu_tanh_bwd_bf16": (i32, [vp, vp, vp, i64, vp]),
    "omh_gelu_erf_bf16": (i32, [vp, vp, i64, vp]),
    "omh_gelu_erf_bwd_bf16": (i32, [vp, vp, vp, i64, vp]),
    "omh_gated_residual_fwd": (i32, [vp, vp, vp, i64, i32, f32, vp, vp, i64, i64, vp]),
    "omh_gated_residual_bwd": (i32, [vp, vp, vp, vp, i64, i64, i32, f32, vp, vp, i64, i64, vp]),
    "omh_layernorm_modulate_bwd": (i32, [vp, vp, vp, i64, i32, f32, f32, vp, vp, i64, vp, vp, i64, i64, vp]),
    "omh_rmsnorm_rope_bwd": (i32, [vp, i64, vp, i64, vp, i64, vp, i64, i32, vp, f32, i32, vp, vp, i32, i32, vp, i32,
                                   vp]),
    "omh_rmsnorm_rope_bwd_t": (i32, [vp, i32, i64, vp, i32, i64, vp, i64, vp, i64, i32, vp, f32, i32, vp, vp, i32, i32,
                                     vp, i32, vp]),
    "omh_layernorm_modulate_bwd2_workspace": (i64, [i64, i32, i64]),
    "omh_layernorm_modulate_bwd2": (i32, [C.POINTER(LnBwdArgs), vp]),
    "omh_rmsnorm_rope_bwd2_workspace": (i64, [i64, i32, i32]),
    "omh_rmsnorm_rope_bwd2": (i32, [C.POINTER(RmsBwdArgs), vp]),
    "omh_partial_colsum_multi": (i32, [C.POINTER(PartialReduceBatch), vp]),
    "omh_softmax_bwd_rows": (i32, [vp, i64, vp, i64, vp, i64, i64, i32, f32, vp]),
    "omh_unpatchify_bwd": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "omh_dense_f32_bwd": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "omh_adamw_step": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp]),
    "omh_adamw_multi": (i32, [vp, i32, f32, f32, f32, f32, f32, i32, f32, vp]),
    "omh_adamw_pack_multi": (i32, [vp, i32, i64, f32, f32, f32, f32, f32, i32, f32, vp]),
    "omh_ema_update": (i32, [vp, vp, i64, f32, vp]),
    "omh_ema_update_multi": (i32, [vp, i32, i64, f32, vp]),
    "omh_pack_weights_multi": (i32, [vp, i32, i64, vp]),
    "omh_gather_rows_f32": (i32, [vp, vp, vp, i64, i32, i64, vp]),
    "omh_gather_rows_bf16": (i32, [vp, vp, vp, i64, i32, i64, vp]),
    "omh_rmsnorm_f32": (i32, [vp, vp, f32, vp, vp, i64, i32, vp]),
    "omh_layernorm_f32": (i32, [vp, vp, vp, f32, vp, i64, i32, vp]),
    "omh_softmax_bias_rows": (i32, [vp, i64, vp, i64, i32, i32, f32, vp, vp, i32, vp]),
    "omh_mul_bf16": (i32, [vp, vp, vp, i64, vp]),
    "omh_vit_embed": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "omh_probe_mfma_tflops": (i32, [i32, i32, vp, i64, vp, vp]),
    "omh_stream_create_cu_mask": (i32, [i32, i32, vp]),
    "omh_stream_destroy": (i32, [vp]),
    "omh_cfg_unipc_step": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, f32, f32, i32, f32, f32, f32, f32, f32,
                                 f32, f32, vp]),
}

for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)      # AttributeError here = header/library out of sync
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTED = tuple(_SIGS)

_ERR = {-1: "OMH_E_BADARG", -2: "OMH_E_ALIGN", -3: "OMH_E_SHAPE"}


def check(rc: int, what: str):
    if rc != 0:
        raise OmhError(f"{what} failed: {_ERR.get(rc, 'hipError ' + str(rc))}")


if lib.omh_abi_version() != ABI_VERSION:  # pragma: no cover
    raise OmhError(f"libomh.so reports ABI version {lib.omh_abi_version()}, this binding was written for {ABI_VERSION}")
