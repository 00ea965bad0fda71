"""ctypes binding of libmnf_hip.so (the C ABI declared in include/mnf_hip.h).

There is no fallback: if the shared library is missing or a call fails, this module raises.
PyTorch is used by the callers only to own device memory and streams.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_uint64, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MNF_LIB_PATH") or os.path.join(_HERE, "libmnf_hip.so")  # override: A/B builds

ABI_VERSION = 14  # include/mnf_hip.h MNF_ABI_VERSION
MNF_OK = 0
MNF_ERR_INVALID_ARG = -1
MNF_ERR_UNSUPPORTED = -2
MNF_ERR_LAUNCH = -3
MNF_ERR_NO_DEVICE = -4
MNF_ERR_DOMAIN = -5

_intp = POINTER(c_int)
_i32p = POINTER(c_int32)
_i64p = POINTER(c_int64)
MNF_SPLIT_TAIL_WORDS = 4

# name -> (restype, argtypes); mirrors include/mnf_hip.h one to one
SIGNATURES = {
    "mnf_abi_version": (c_int, []),
    "mnf_error_string": (c_char_p, [c_int]),
    "mnf_last_hip_error": (c_int, []),
    "mnf_last_kernel": (c_char_p, []),
    "mnf_deterministic": (c_int, []),
    "mnf_device_count": (c_int, []),
    "mnf_affine_half": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                c_int, c_int, c_int, _intp, c_int, c_int, c_int, c_void_p]),
    "mnf_affine_half_sq": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                   c_int64, c_int, c_int, c_int, c_int, _intp, c_int, c_int, c_int, c_void_p]),
    "mnf_affine_half_stack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, _intp, c_int,
                                      c_int64, c_int, c_int, c_int, _intp, c_void_p]),
    "mnf_affine_half_split_layout": (c_int, [c_int, c_int, _intp, c_int, c_int, _i64p, _i64p]),
    "mnf_affine_half_split_index": (c_int, [c_int, c_int, _intp, c_int, c_int, _i32p]),
    "mnf_pack_gather_split": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "mnf_affine_half_image_floats": (c_int64, [c_int, c_int, _intp, c_int, c_int]),
    "mnf_affine_half_image_index": (c_int, [c_int, c_int, _intp, c_int, c_int, _i32p]),
    "mnf_affine_half_flat_floats": (c_int64, [c_int, c_int, _intp, c_int, c_int]),
    "mnf_pack_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mnf_pack_gather_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_void_p]),
    "mnf_pack_gather_split_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int64, c_void_p]),
    "mnf_nsf_cl": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                           c_float, c_int, c_int, _intp, c_int, c_void_p]),
    "mnf_nsf_cl_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                 c_int, c_float, c_int, c_int, _intp, c_void_p]),
    "mnf_nsf_cl_split_layout": (c_int, [c_int, c_int, c_int, _intp, _i64p, _i64p]),
    "mnf_nsf_cl_split_index": (c_int, [c_int, c_int, c_int, _intp, _i32p]),
    "mnf_nsf_cl_flat_floats": (c_int64, [c_int, c_int, c_int, _intp]),
    "mnf_nsf_cl_image_floats": (c_int64, [c_int, c_int, c_int, _intp]),
    "mnf_nsf_cl_image_index": (c_int, [c_int, c_int, c_int, _intp, _i32p]),
    "mnf_rqs": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float,
                        c_int, c_void_p]),
    "mnf_rnvp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                         c_int, _intp, c_int, c_void_p]),
    "mnf_rnvp_seeded": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                c_void_p, c_int64, c_int, c_int, _intp, c_int, c_void_p]),
    "mnf_rnvp_split_layout": (c_int, [c_int, c_int, _intp, _i64p, _i64p]),
    "mnf_rnvp_split_index": (c_int, [c_int, c_int, _intp, _i32p]),
    "mnf_rnvp_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_int, c_void_p,
                                c_void_p, c_int64, c_int, c_int, _intp, c_void_p]),
    "mnf_rnvp_mask": (c_int, [c_uint64, c_void_p, c_int64, c_int, c_void_p]),
    "mnf_rnvp_flat_floats": (c_int64, [c_int, c_int, _intp]),
    "mnf_rnvp_image_floats": (c_int64, [c_int, c_int, _intp]),
    "mnf_rnvp_image_index": (c_int, [c_int, c_int, _intp, _i32p]),
    "mnf_affine_const": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64,
                                 c_int, c_int, c_void_p]),
    "mnf_linear_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mnf_linear_rows_image_floats": (c_int64, [c_int]),
    "mnf_linear_rows_image_index": (c_int, [c_int, _i32p]),
    "mnf_linear_rows_img": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mnf_gauss_logprob": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mnf_gauss_logprob_sq": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mnf_affine_half_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                    c_int, c_int, c_int, _intp, c_int, c_int, c_void_p]),
    "mnf_affine_half_bwd_index_ints": (c_int64, [c_int, c_int, _intp, c_int, c_int]),
    "mnf_affine_half_bwd_index": (c_int, [c_int, c_int, _intp, c_int, c_int, _i32p]),
    "mnf_affine_half_bwd_mfma": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                         c_int, c_int, c_int, c_int, _intp, c_void_p]),
    "mnf_affine_half_bwd_mfma_workspace": (c_int64, [c_int64, c_int, c_int, _intp]),
    "mnf_affine_half_bwd_mfma_det": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                             c_int, c_int, c_int, c_int, _intp, c_void_p, c_int64, c_void_p]),
    "mnf_affine_half_bwd_mfma_tiles": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_int64, c_int, c_int, c_int, c_int, _intp, c_void_p, c_int, c_void_p]),
    "mnf_affine_half_bwd_split_layout": (c_int, [c_int, c_int, _intp, c_int, c_int, _i64p, _i64p]),
    "mnf_affine_half_bwd_split_index": (c_int, [c_int, c_int, _intp, c_int, c_int, _i32p]),
    "mnf_affine_half_grad_scale": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "mnf_affine_half_bwd_split_workspace": (c_int64, [c_int64, c_int, c_int, _intp]),
    "mnf_affine_half_bwd_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                          c_int, c_int, c_int, c_int, _intp, c_void_p, c_void_p, c_int, c_void_p, c_int64,
                                          c_void_p]),
    "mnf_affine_half_bwd_split_lp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_int64, c_int, c_int, c_int, c_int, _intp, c_void_p, c_void_p, c_int,
                                             c_void_p, c_int64, c_void_p]),
    "mnf_affine_half_bwd_rt": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int64, c_int, c_int, c_int, c_int, _intp, c_int, c_int, c_void_p]),
    "mnf_nsf_cl_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                               c_float, c_int, c_int, _intp, c_void_p]),
    "mnf_nsf_cl_bwd_rt": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                  c_int, c_int, c_float, c_int, c_int, _intp, c_void_p]),
    "mnf_nsf_cl_bwd_tile_supported": (c_int, [c_int, c_int, c_int, _intp]),
    "mnf_nsf_cl_bwd_tile_layout": (c_int, [c_int, c_int, c_int, _intp, _i64p, _i64p, _i64p]),
    "mnf_nsf_cl_bwd_tile_index": (c_int, [c_int, c_int, c_int, _intp, _i32p, _i32p]),
    "mnf_nsf_cl_bwd_tile_workspace": (c_int64, [c_int64, c_int, c_int, c_int, _intp]),
    "mnf_nsf_cl_bwd_tile": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                    c_int, c_float, c_int, c_int, _intp, c_void_p, c_void_p, c_int, c_void_p, c_int64,
                                    c_void_p]),
    "mnf_nsf_cl_bwd_tile_fixup": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                          c_int, c_float, c_int, c_int, _intp, c_void_p, c_int, c_void_p]),
    "mnf_rnvp_bwd": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_int64, c_int, c_int, _intp, c_void_p]),
    "mnf_rnvp_bwd_rt": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_int64, c_int, c_int, _intp, c_void_p]),
    "mnf_rnvp_bwd_mfma_workspace_bytes": (c_int64, [c_int64, c_int, c_int, _intp]),
    "mnf_rnvp_bwd_mfma_layout": (c_int, [c_int, c_int, _intp, _i64p, _i64p]),
    "mnf_rnvp_bwd_mfma_index": (c_int, [c_int, c_int, _intp, _i32p]),
    "mnf_rnvp_bwd_mfma": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, _intp, c_void_p]),
    "mnf_rnvp_bwd_mfma_phases": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, _intp,
                                         c_int, c_void_p, c_void_p]),
    "mnf_rnvp_y_floats_per_row": (c_int, [c_int, _intp]),
    "mnf_rnvp_seeded_train": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                      c_void_p, c_int64, c_int, c_int, _intp, c_int, c_void_p, POINTER(c_int), c_void_p]),
    "mnf_affine_const_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_int64, c_int, c_int, c_void_p]),
    "mnf_linear_rows_bwd_weight": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mnf_glow_actnorm_inv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                     c_void_p]),
    "mnf_glow_actnorm_inv_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mnf_glow_actnorm_inv_logprob": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                             c_int, c_void_p]),
    "mnf_glow_actnorm_inv_logprob_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                 c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mnf_glow_actnorm_inv_bwd_workspace": (c_int64, [c_int64, c_int]),
    "mnf_glow_actnorm_inv_bwd_det": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "mnf_glow_actnorm_inv_logprob_bwd_det": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                     c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64,
                                                     c_void_p]),
    "mnf_sample_z0": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mnf_sample_z0_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mnf_sample_z0_seeded": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_int64, c_int, c_void_p]),
    "mnf_sample_z0_seeded_bwd": (c_int, [c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mnf_sample_z0_noise": (c_int, [c_uint64, c_void_p, c_int64, c_int, c_void_p]),
    "mnf_sample_z0_bwd_workspace": (c_int64, [c_int64, c_int]),
    "mnf_sample_z0_bwd_det": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p,
                                      c_int64, c_void_p]),
    "mnf_adam_step_graph": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                                    c_float, c_void_p, c_void_p]),
    "mnf_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                              c_float, c_int, c_void_p]),
    "mnf_nsf_ar": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_float, c_int, c_int,
                           _intp, c_void_p]),
    "mnf_nsf_ar_flat_floats": (c_int64, [c_int, c_int, c_int, _intp]),
    "mnf_nsf_ar_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                               c_float, c_int, c_int, _intp, c_void_p]),
    "mnf_mnf_linear_split_layout": (c_int, [c_int, c_int, _i64p, _i64p]),
    "mnf_mnf_linear_split_index": (c_int, [c_int, c_int, _i32p]),
    "mnf_mnf_linear_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_float,
                                   c_void_p, c_int64, c_int, c_int, c_void_p]),
    "mnf_mnf_linear_noise": (c_int, [c_uint64, c_void_p, c_int64, c_int, c_void_p]),
    "mnf_mnf_linear_fwd_train": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_float, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "mnf_mnf_linear_bwd_workspace_bytes": (c_int64, [c_int64, c_int, c_int]),
    "mnf_mnf_linear_bwd_layout": (c_int, [c_int, c_int, _i64p, _i64p]),
    "mnf_mnf_linear_bwd_index": (c_int, [c_int, c_int, _i32p]),
    "mnf_mnf_linear_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                   c_int, c_int, c_void_p]),
    "mnf_mnf_conv_operands": (c_int, [c_void_p] * 7 + [c_int, c_int, c_void_p]),
    "mnf_mnf_conv_operands_bwd": (c_int, [c_void_p] * 11 + [c_int, c_int, c_int, c_void_p]),
    "mnf_mnf_noise": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mnf_mnf_noise_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "mnf_glow_weight": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mnf_glow_weight_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mnf_maf_flat_floats": (c_int64, [c_int, c_int, _intp]),
    "mnf_maf_mask_bytes": (c_int64, [c_int, c_int, _intp]),
    "mnf_maf": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, _intp,
                        c_void_p]),
    "mnf_maf_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                            c_int, c_int, c_int, _intp, c_void_p]),
    "mnf_rnvp_bwd_few_workspace_floats": (c_int64, [c_int64, c_int, c_int, _intp]),
    "mnf_rnvp_bwd_few": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_int64, c_int, c_int, _intp, c_void_p]),
    "mnf_rnvp_few_rows_ok": (c_int, [c_int64, c_int, c_int, _intp, c_int]),
    "mnf_mnf_kl_saved_floats": (c_int64, [c_int64]),
    "mnf_mnf_kl_grad_floats": (c_int64, [c_int]),
    "mnf_mnf_kl_param_grad_floats": (c_int64, [c_int, c_int64, c_int, c_int]),
    "mnf_mnf_kl_fwd": (c_int, [c_void_p] * 14 + [c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mnf_mnf_kl_bwd": (c_int, [c_void_p] * 13 + [c_int, c_int64, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
}

_lib = None


class MnfHipError(RuntimeError):
    """A libmnf_hip.so call returned a negative MNF_ERR_* code."""

    def __init__(self, fn: str, code: int, message: str):
        super().__init__(f"{fn} failed: {message} (code {code})")
        self.code = code


def load() -> ctypes.CDLL:
    """Load libmnf_hip.so (once) and attach the prototypes.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C torch_mnf_amd/csrc). "
            "torch_mnf_amd has no CPU or PyTorch fallback path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header / library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.mnf_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} has ABI version {lib.mnf_abi_version()}, this package binds version "
                           f"{ABI_VERSION}: rebuild it (make -C torch_mnf_amd/csrc)")
    _lib = lib
    return lib


def check(fn: str, code: int) -> None:
    if code == MNF_OK:
        return
    lib = load()
    msg = lib.mnf_error_string(code).decode()
    if code == MNF_ERR_LAUNCH:
        msg += f" (hipError_t {lib.mnf_last_hip_error()})"
    if code == MNF_ERR_DOMAIN:
        # same exception type and text as the reference (spline_flow.py:90-93)
        raise ValueError("Minimal bin width too large for the number of bins")
    raise MnfHipError(fn, code, msg)


def int_array(values) -> ctypes.Array:
    values = [int(v) for v in values]
    return (c_int * max(len(values), 1))(*values)


def last_kernel() -> str:
    """The kernel family the process's most recent layer call ran ("ahf_split_stack", "nsf_bwd_tile", ... or a
    "*_generic" name for the any-shape kernels; "" before the first launch): include/mnf_hip.h mnf_last_kernel."""
    return (load().mnf_last_kernel() or b"").decode()


def deterministic() -> bool:
    """True when the library runs its fixed-order gradient reductions (MNF_DETERMINISTIC=1 in the environment when the
    process started): include/mnf_hip.h mnf_deterministic."""
    return bool(load().mnf_deterministic())


_WARNED_GENERIC: set = set()
GENERIC_WARN_ROWS = 4096


def note_generic(layer: str, rows: int, shape: str) -> None:
    """Once per (layer type, shape): tell the caller that a batch of >= 4,096 rows ran on an any-shape kernel.  The
    specialised kernels are narrow templates (shape -> kernel table) and nothing else says which
    side of a cliff a layer landed on; the generic kernels are correct but 20-40 x slower at large batches."""
    if rows < GENERIC_WARN_ROWS:
        return
    name = last_kernel()
    if "generic" not in name or (layer, shape) in _WARNED_GENERIC:
        return
    _WARNED_GENERIC.add((layer, shape))
    import warnings
    warnings.warn(f"torch_mnf_amd: {layer}({shape}) ran on the any-shape kernel {name!r} at {rows} rows: no matrix-core "
                  "kernel for this shape (see INTEGRATION.md, 'Which kernel runs'); results are the same, the speed is not",
                  RuntimeWarning, stacklevel=3)
