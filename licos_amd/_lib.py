"""ctypes binding of ``liblicos_hip.so`` (the C ABI declared in include/licos_hip.h).

The HIP library is the product; there is no CPU fallback.  If the shared object
is missing this module raises immediately with the build instruction.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("LICOS_HIP_SO") or os.path.join(_HERE, "liblicos_hip.so")  # override: A/B builds of dev tools

_c = ctypes
_vp, _i, _l, _f = _c.c_void_p, _c.c_int, _c.c_long, _c.c_float

# name -> (restype, argtypes); must list every symbol of include/licos_hip.h
SIGNATURES = {
    "licos_last_error": (_c.c_char_p, []),
    "licos_abi_version": (_i, []),
    "licos_query": (_i, [_i, _vp]),
    "licos_pmf_to_quantized_cdf": (_i, [_vp, _i, _i, _vp]),
    "licos_rans_build_enc_table": (_i, [_vp, _vp, _i, _i, _vp]),
    "licos_conv2d_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "licos_deconv2d_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "licos_gdn_reparam_f32": (_i, [_vp, _vp, _f, _f, _f, _vp, _vp, _i, _vp]),
    "licos_gdn_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "licos_gdn_f32_split3_applies": (_i, [_i, _i]),
    "licos_gdn_f32_fwd_norm": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "licos_gdn_bwd_fused_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "licos_gdn_f32_split3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "licos_conv2d_wgrad_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "licos_gdn_gamma_grad_parts": (_i, [_i, _l]),
    "licos_gdn_gamma_grad_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _l, _vp]),
    "licos_gdn_gamma_grad_scaled_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _l, _vp]),
    "licos_bias_grad_f32": (_i, [_vp, _vp, _i, _i, _l, _vp]),
    "licos_gdn_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "licos_reparam_bwd_f32": (_i, [_vp, _vp, _f, _vp, _l, _vp]),
    "licos_adam_f32": (_i, [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _i, _f, _vp]),
    "licos_sumsq_f32": (_i, [_vp, _l, _vp, _vp]),
    "licos_eb_packed_size": (_i, [_vp, _i]),
    "licos_eb_pack": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "licos_eb_quantize": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _l, _i, _i, _i, _i, _vp]),
    "licos_eb_likelihood": (_i, [_vp, _vp, _vp, _i, _vp, _f, _i, _vp, _i, _i, _i, _vp]),
    "licos_eb_likelihood_bwd_slices": (_i, [_i, _i]),
    "licos_eb_likelihood_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _f, _i, _vp, _vp, _i, _i, _i, _vp]),
    "licos_gc_likelihood_bwd": (_i, [_vp, _vp, _vp, _f, _f, _vp, _vp, _l, _vp]),
    "licos_mask_mul_f32": (_i, [_vp, _vp, _vp, _l, _i, _vp]),
    "licos_eb_dequantize": (_i, [_vp, _l, _l, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "licos_reduce_sqdiff": (_i, [_vp, _vp, _l, _i, _vp, _vp]),
    "licos_ssim_stats_f32": (_i, [_vp, _vp, _i, _i, _i, _vp, _f, _f, _vp, _vp]),
    "licos_rans_encode_batch": (_i, [_vp, _vp, _l, _l, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    "licos_rans_encode_host": (_i, [_vp, _vp, _l, _l, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _l, _vp, _i, _i]),
    "licos_rans_decode_host": (_i, [_vp, _vp, _vp, _l, _l, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i]),
    "licos_rans_compact": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp]),
    "licos_rans_decode_batch": (_i, [_vp, _vp, _vp, _l, _l, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "licos_gc_encode_prepare": (_i, [_vp, _vp, _vp, _i, _f, _vp, _i, _vp, _vp, _vp, _vp, _i, _l, _vp]),
    "licos_rans_encode_records": (_i, [_vp, _vp, _l, _vp, _i, _vp, _vp, _i, _vp]),
    "licos_rans_image_budget": (_l, [_i]),
    "licos_rans_image_build": (_i, [_vp, _vp, _vp, _i, _i, _vp, _l, _vp, _vp]),
    "licos_rans_image_lookup": (_i, [_vp, _i, _i, _vp]),
    "licos_gc_decode_prepare": (_i, [_vp, _vp, _i, _f, _vp, _vp, _i, _l, _vp]),
    "licos_rans_decode_image": (_i, [_vp, _vp, _vp, _i, _l, _vp, _vp, _vp, _l, _l, _vp, _i, _vp]),
    "licos_eb_encode_prepare": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "licos_gc_likelihood": (_i, [_vp, _vp, _vp, _f, _f, _vp, _i, _i, _i, _vp]),
    "licos_gc_build_indexes": (_i, [_vp, _vp, _i, _f, _vp, _l, _l, _i, _l, _vp]),
    "licos_eb_symbols16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "licos_eb_dequantize16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "licos_rans_encode_host_sym16": (_i, [_vp, _l, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _l, _vp, _i, _i]),
    "licos_rans_decode_host_sym16": (_i, [_vp, _vp, _l, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i]),
    "licos_gc_pack_symbols": (_i, [_vp, _vp, _vp, _i, _f, _vp, _vp, _i, _l, _vp]),
    "licos_gc_build_rows8": (_i, [_vp, _vp, _i, _f, _vp, _i, _l, _vp]),
    "licos_rans_encode_host_packed": (_i, [_vp, _l, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _l, _vp, _i, _i]),
    "licos_rans_decode_host_rows8": (_i, [_vp, _vp, _vp, _l, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i]),
    "licos_dn12_to_grid8_f32": (_i, [_vp, _vp, _l, _i, _vp]),
    "licos_resample_bilinear_f32": (_i, [_vp, _vp, _l, _i, _i, _i, _i, _f, _f, _i, _vp]),
    "licos_tile_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "licos_untile_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "licos_tile_overlap_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "licos_untile_overlap_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "licos_comm_unique_id": (_i, [_vp]),
    "licos_comm_init": (_i, [_vp, _i, _i, _vp]),
    "licos_comm_destroy": (_i, [_vp]),
    "licos_allreduce_weighted": (_i, [_vp, _l, _f, _vp, _vp]),
    "licos_allreduce_weighted_direct": (_i, [_vp, _l, _l, _f, _vp, _i, _i, _vp, _vp]),
    "licos_scale_f32": (_i, [_vp, _l, _f, _vp, _vp]),
    "licos_packed_conv_w_bytes": (_c.c_size_t, [_i, _i]),
    "licos_pack_conv_w_f16": (_i, [_vp, _i, _i, _vp, _vp]),
    "licos_pack_deconv_w_f16": (_i, [_vp, _i, _i, _vp, _vp]),
    "licos_packed_gdn_bytes": (_c.c_size_t, [_i]),
    "licos_pack_gdn_bf16": (_i, [_vp, _vp, _f, _f, _f, _i, _vp, _vp]),
    "licos_packed_gdn_f32split_bytes": (_c.c_size_t, [_i]),
    "licos_pack_gdn_f32split": (_i, [_vp, _vp, _f, _f, _f, _i, _vp, _vp]),
    "licos_nchw_f32_to_s2d_blk16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "licos_pack_conv_w_s2d_f16": (_i, [_vp, _i, _i, _vp, _vp]),
    "licos_conv5x5s2_s2d_f16": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "licos_hwc_pad_f16_bytes": (_c.c_size_t, [_i, _i, _i, _i]),
    "licos_nchw_f32_to_hwc_pad_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "licos_packed_conv_w_first_bytes": (_c.c_size_t, [_i, _i]),
    "licos_pack_conv_w_first_f16": (_i, [_vp, _i, _i, _vp, _vp]),
    "licos_conv5x5s2_first_f16": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "licos_conv5x5s2_first_nchw_f16": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "licos_conv5x5s2_first16_nchw_f16": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "licos_packed_deconv_w_fewch_bytes": (_c.c_size_t, [_i, _i]),
    "licos_pack_deconv_w_fewch_f16": (_i, [_vp, _i, _i, _vp, _vp]),
    "licos_deconv5x5s2_fewch_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "licos_packed_deconv_w_scatter_bytes": (_c.c_size_t, [_i, _i]),
    "licos_pack_deconv_w_scatter_f16": (_i, [_vp, _i, _i, _vp, _vp]),
    "licos_deconv5x5s2_scatter_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "licos_packed_deconv_w_rows_bytes": (_c.c_size_t, [_i, _i]),
    "licos_pack_deconv_w_rows_f16": (_i, [_vp, _i, _i, _vp, _vp]),
    "licos_deconv5x5s2_rows_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "licos_nchw_f32_to_blk16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "licos_packed_conv1x1_w_bytes": (_c.c_size_t, [_i, _i]),
    "licos_pack_conv1x1_w_f16": (_i, [_vp, _i, _i, _vp, _vp]),
    "licos_conv1x1_f16": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "licos_gdn_pointwise_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _vp]),
    "licos_nchw_f32_split_bm8": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "licos_wgrad5x5s2_strips": (_i, [_i, _i, _i]),
    "licos_wgrad5x5s2_f16": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "licos_nchw_f32_split_blk16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "licos_nchw_f32_split3_blk16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "licos_pack_conv3x3_w_f16": (_i, [_vp, _i, _i, _vp, _vp]),
    "licos_conv3x3s1_f16": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "licos_blk16_to_nchw_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "licos_conv5x5s2_f16": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "licos_conv5x5s2_f16_symbols": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "licos_deconv5x5s2_f16": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "licos_deconv5x5s2_f16_layouts": (_i, [_i, _i, _i, _i]),
}

_lib = None


class LicosError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise LicosError(
            f"licos_amd: {SO_PATH} is missing - the HIP extension is the product and there is no CPU "
            "fallback.  Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C licos_amd/csrc`)."
        )
    lib = ctypes.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.licos_abi_version() != 1:
        raise LicosError("licos_amd: ABI version mismatch between Python host and liblicos_hip.so")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc < 0:
        msg = load().licos_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        if rc == -3:
            raise ValueError(msg)
        raise LicosError(f"{what}: {msg} (code {rc})")
    return rc
