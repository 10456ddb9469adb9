"""ctypes binding of librba_hip.so (the C ABI declared in include/rba_hip.h).

The product has no CPU path: if the library is missing, or a wrapper is handed a tensor that is not a
contiguous fp32 tensor on a HIP device, it raises -- it never falls back to PyTorch or to the oracle.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RBA_HIP_LIB") or os.path.join(_HERE, "csrc", "librba_hip.so")   # env override: A/B runs of two builds
# the same sources built with -DRBA_TUNE_KNOBS (csrc/knobs.h): the tuning knobs as exported, writable ints.  Tests and tools only -- the product library
# (LIB_PATH) has no writable state besides rba_set_concurrent_streams' hint.  `use_library(KNOBS_LIB_PATH)` swaps it in for a block.
KNOBS_LIB_PATH = os.path.join(_HERE, "csrc", "librba_hip_knobs.so")

_c_f32p = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_vp = ctypes.c_void_p

# name -> argtypes ; every function returns int (hipError_t)
SIGNATURES = {
    "rba_hip_version": [],
    "rba_set_concurrent_streams": [_i],
    "rba_reduce_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i, _vp],
    "rba_reduce_ws_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i, _vp, _vp],
    "rba_reduce_up4_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "rba_resample_bilinear_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "rba_ms_deform_attn_fwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "rba_ms_deform_attn_fwd_f64": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "rba_msda_prepare_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "rba_resample_bilinear_nhwc_gn_f32": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i64, _vp],
    "rba_group_norm_nhwc_stats_f32": [_vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _vp],
    "rba_msda_fused_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "rba_masked_xattn_workspace_bytes": [_i, _i, _i, _i],
    "rba_masked_xattn_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "rba_mask_logits_f32": [_vp, _vp, _vp, _i, _i, _i, _i64, _vp],
    "rba_mask_logits_f16x3_f32": [_vp, _vp, _vp, _i, _i, _i, _i64, _vp],
    "rba_swin_window_attn_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "rba_swin_window_attn_split_out_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "rba_swin_bias_fragments_elems": [_i, _i],
    "rba_swin_attn_block_supported": [_i, _i],
    "rba_swin_attn_block_weight_bytes": [_i],
    "rba_swin_attn_qkv_supported": [_i, _i],
    "rba_swin_attn_qkv_split_out_f32": [_vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "rba_swin_attn_block_pack_f32": [_vp, _vp, _vp, _i, _vp],
    "rba_swin_attn_block_f32": [_vp, _vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _i, _i, _i, _i, _i, _i, _vp],
    "rba_swin_bias_fragments_f32": [_vp, _vp, _i, _i, _vp],
    "rba_skinny_linear_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "rba_skinny_linear_add_f32": [_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "rba_split_weight_bf16x3": [_vp, _vp, _i, _i, _vp],
    "rba_split_linear_f32": [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "rba_split_weight_f16x2": [_vp, _vp, _i, _i, _vp],
    "rba_split_linear_f16x3_f32": [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "rba_split_linear_f16x3_res_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "rba_split_linear_f16x3_frag_f32": [_vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "rba_swin_mlp_fused_f16x3_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "rba_swin_mlp_fused_ln_f16x3_f32": [_vp, _vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "rba_split_linear_f16x3_gelu_split_out": [_vp, _i, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "rba_split_linear_nchw_out_f32": [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "rba_split_linear_nchw_out_f16x3_f32": [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "rba_split_linear_f16x3_gn_moments_f32": [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp],
    "rba_conv3x3_nhwc_f16x3_split_in_gn_moments_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "rba_group_norm_nhwc_merge_f32": [_vp, _vp, _i, _i, _i, ctypes.c_float, _vp],
    "rba_split_linear_nchw_out_gn_f16x3_f32": [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "rba_conv3x3_nhwc_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "rba_conv3x3_nhwc_f16x3_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "rba_conv3x3_nhwc_f16x3_split_in_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "rba_resample_bilinear_nhwc_split_out_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i64, _vp],
    "rba_patch_im2col_u8": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "rba_patch_im2col_f32": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "rba_group_norm_nhwc_workspace_bytes": [_i, _i, _i, _i],
    "rba_group_norm_nhwc_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _i, _vp],
    "rba_resample_bilinear_nhwc_f32": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "rba_bn_relu_conv1x1_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _vp],
    "rba_resample_bilinear_ac_f32": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "rba_gaussian_blur_f32": [_vp, _vp, _i, _i, _i, ctypes.c_float, _vp],
    "rba_quad_mean_f32": [_vp, _vp, _i64, _i, _vp],
    "rba_softmax_drop_last_f32": [_vp, _vp, _i64, _i, _vp],
    "rba_threshold_u8": [_vp, _vp, _i64, ctypes.c_float, _vp],
    "rba_morph3x3_u8": [_vp, _vp, _i, _i, _i, _vp],
    "rba_ccl4_roots_i32": [_vp, _vp, _i, _i, _vp],
    "rba_add_layer_norm_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, ctypes.c_float, _vp],
    "rba_add_layer_norm_frag_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, ctypes.c_float, _vp],
    "rba_merge_layer_norm_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _vp],
    "rba_group_norm_workspace_bytes": [_i, _i, _i, _i],
    "rba_group_norm_f32": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _i, _vp],
    "rba_token_linear_pack_f16x2": [_vp, _vp, _i, _i, _vp],
    "rba_token_linear_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_float, _vp, _i64, _i, _i, _i, _vp],
    "rba_token_linear_multi_f32": [_vp, _vp, _i, _i64, _i, _vp],
}

class TokenLinearProblem(ctypes.Structure):
    """rba_token_linear_problem of include/rba_hip.h"""
    _fields_ = [("x_add", ctypes.c_void_p), ("packed", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("out", ctypes.c_void_p),
                ("N", ctypes.c_int), ("ld_out", ctypes.c_int), ("act", ctypes.c_int)]


EXPECTED_ABI = 190        # rba_hip_version() the argtypes above were written for (include/rba_hip.h)

_lib = None


class RbaHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises RbaHipError when the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    _lib = _open(LIB_PATH)
    return _lib


_handles = {}


def _open(path):
    if path in _handles:
        return _handles[path]
    # torch bundles its own libamdhip64; it must be the HIP runtime already resident when librba_hip.so's
    # DT_NEEDED libamdhip64.so.N is resolved, otherwise the process ends up with a second runtime (the system one)
    # that owns our code objects but not torch's device context -> launches fail with hipErrorNoDevice.
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise RbaHipError(
            f"{path} not found: build it with `python -m rba_amd.csrc.build` (hipcc --offload-arch=gfx950; --knobs for the knobs build). "
            "rba_amd has no CPU or PyTorch fallback for its HIP kernels.")
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RbaHipError(f"{path} does not export {name}; rebuild it") from e
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int64 if name.endswith(("_bytes", "_elems")) else ctypes.c_int
    abi = int(lib.rba_hip_version())
    if abi != EXPECTED_ABI:               # same names, possibly different argument lists: calling it would corrupt memory
        raise RbaHipError(f"{path} has ABI version {abi}, these bindings are written for {EXPECTED_ABI}; rebuild it "
                          "(python -m rba_amd.csrc.build)")
    _handles[path] = lib
    return lib


class use_library:
    """with use_library(path): every rba_amd.ops call of the block runs on ANOTHER build of the kernel library (same ABI) -- the knobs build for tests and
    tools that select a kernel variant.  Not thread-safe; not for product code."""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        global _lib
        self.prev = _lib
        _lib = _open(self.path)
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def knob(name):
    """ctypes int of a tuning knob of the CURRENT library -- exists only in the knobs build (csrc/knobs.h)."""
    try:
        return ctypes.c_int.in_dll(load(), name)
    except ValueError as e:
        raise RbaHipError(f"{name} is a compile-time constant in the product library: run on the knobs build (RBA_HIP_LIB={KNOBS_LIB_PATH} or "
                          "rba_amd._lib.use_library(KNOBS_LIB_PATH); python -m rba_amd.csrc.build --knobs)") from e


def hip_error_string(code: int) -> str:
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipGetErrorString.restype = ctypes.c_char_p
        hip.hipGetErrorString.argtypes = [ctypes.c_int]
        return hip.hipGetErrorString(code).decode()
    except Exception:
        return "hipError"


def check(code: int, what: str):
    if code != 0:
        raise RbaHipError(f"{what} failed: hipError {code} ({hip_error_string(code)})")
