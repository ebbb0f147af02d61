"""ctypes binding of libdreammat_hip.so (the C ABI declared in include/dreammat_hip.h).

The product path FAILS LOUDLY when the HIP library is missing: there is no CPU or PyTorch fallback
for any of these ops.
"""
import ctypes
import os

import torch  # noqa: F401  -- MUST precede CDLL: PyTorch-ROCm bundles its own libamdhip64; loading ours first
#                              binds the process to the system runtime and every launch fails with hipErrorNoDevice
from ctypes import POINTER, c_float, c_int, c_int32, c_longlong, c_size_t, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdreammat_hip.so")
if os.environ.get("DREAMMAT_LIB"):       # development aid (tools/grad_budget.py: the same library built with other flags)
    LIB_PATH = os.environ["DREAMMAT_LIB"]

_lib = None
ABI_VERSION = 14     # dm_abi_version() of the library this binding was written for (csrc/host.cpp, include/dreammat_hip.h)

DM_ERRORS = {-1: "DM_ERR_ARG", -2: "DM_ERR_WORKSPACE", -3: "DM_ERR_UNSUPPORTED"}


class DmError(RuntimeError):
    pass


class EnvAtlasStruct(ctypes.Structure):
    _fields_ = [("spec", c_void_p), ("diff", c_void_p), ("fg_lut", c_void_p),
                ("spec_env_stride", c_longlong), ("diff_env_stride", c_longlong),
                ("mip_off", c_longlong * 8), ("mip_res", c_int * 8),
                ("n_mips", c_int), ("diff_res", c_int), ("lut_res", c_int),
                ("min_rough_mip", c_float), ("max_rough_mip", c_float), ("texel_format", c_int),
                ("fg_pairs", c_void_p)]


class MatCfgStruct(ctypes.Structure):
    _fields_ = [("min_metallic", c_float), ("max_metallic", c_float),
                ("min_roughness", c_float), ("max_roughness", c_float)]


class McSceneStruct(ctypes.Structure):
    """dm_mc_scene (include/dreammat_hip.h): host struct with device pointers for the Monte-Carlo shading kernels."""
    _fields_ = [("bvh_nodes", c_void_p), ("bvh_tris", c_void_p), ("lights", c_void_p), ("n_env", c_int),
                ("light_h", c_int), ("light_w", c_int), ("samples_diffuse", c_void_p), ("samples_specular", c_void_p),
                ("n_diffuse", c_int), ("n_specular", c_int), ("geometry_ggx_smith", c_int), ("bvh_nodes4", c_void_p),
                ("grid", c_void_p)]


class GridStruct(ctypes.Structure):
    """dm_grid (include/dreammat_hip.h): occupancy grid of the mesh, host struct with device pointers."""
    _fields_ = [("gmin", c_float * 3), ("cell", c_float), ("inv_cell", c_float), ("dim", c_int32 * 3),
                ("n_words", c_int32), ("n_occ", c_int32), ("n_entries", c_longlong),
                ("bits", c_void_p), ("sbase", c_void_p), ("off16", c_void_p), ("dist4", c_void_p), ("occ_start", c_void_p),
                ("cell_tris", c_void_p)]


_LL = c_longlong
_SIGS = {
    "dm_abi_version": (c_int, []),
    "dm_mesh_build_topology": (c_int, [c_void_p, c_int32, c_void_p]),
    "dm_vertex_transform": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "dm_raster_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dm_rasterize": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                             c_void_p]),
    "dm_raster_overflowed": (c_int, [c_void_p, c_void_p, POINTER(c_int)]),
    "dm_interpolate": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, _LL, c_void_p, c_void_p]),
    "dm_antialias_plan": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                  c_void_p]),
    "dm_antialias_apply": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dm_antialias_grad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "dm_gbuffer_workspace_bytes": (c_size_t, [_LL]),
    "dm_gbuffer_compact": (c_int, [c_void_p, _LL, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                   _LL, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                   c_void_p]),
    "dm_gbuffer_tiled_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "dm_gbuffer_compact_tiled": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_float, _LL, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_size_t, c_void_p]),
    "dm_control_maps": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    "dm_scatter_rows": (c_int, [c_void_p, c_void_p, _LL, c_void_p, _LL, _LL, c_int, c_void_p, c_void_p]),
    "dm_gather_rows": (c_int, [c_void_p, c_void_p, _LL, c_void_p, c_int, c_void_p, _LL, _LL, c_void_p]),
    "dm_hashgrid_fwd": (c_int, [c_void_p, _LL, _LL, c_void_p, _LL, c_void_p, c_int, POINTER(c_float),
                                POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32), c_float, c_void_p, _LL, _LL,
                                c_void_p]),
    "dm_hashgrid_bwd": (c_int, [c_void_p, _LL, _LL, c_void_p, _LL, c_void_p, _LL, _LL, c_int, POINTER(c_float),
                                POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32), c_float, c_void_p, c_void_p]),
    "dm_hashgrid2d_fwd": (c_int, [c_void_p, _LL, _LL, c_void_p, _LL, c_void_p, c_int, POINTER(c_float),
                                  POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32), c_float, c_void_p, _LL, _LL,
                                  c_void_p]),
    "dm_hashgrid2d_bwd": (c_int, [c_void_p, _LL, _LL, c_void_p, _LL, c_void_p, _LL, _LL, c_int, POINTER(c_float),
                                  POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32), c_float, c_void_p, c_void_p]),
    "dm_field_mlp_fwd": (c_int, [c_void_p, _LL, _LL, c_void_p, c_void_p, c_int, c_int, c_void_p, _LL, c_void_p]),
    "dm_field_mlp_bwd": (c_int, [c_void_p, _LL, _LL, c_void_p, c_void_p, c_int, c_int, c_void_p, _LL, _LL, c_void_p, _LL, c_void_p,
                                 c_void_p, c_void_p]),
    "dm_hashgrid_bwd_workspace_bytes": (c_size_t, [_LL, c_int, POINTER(c_uint32), POINTER(c_uint32)]),
    "dm_hashgrid_bwd_binned": (c_int, [c_void_p, _LL, _LL, c_void_p, _LL, c_void_p, _LL, _LL, c_int, POINTER(c_float),
                                       POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32), c_float, c_void_p, c_void_p,
                                       c_size_t, c_void_p]),
    "dm_shade_fwd": (c_int, [POINTER(EnvAtlasStruct), POINTER(MatCfgStruct), c_void_p, _LL, _LL, c_void_p, _LL, _LL,
                             c_void_p, _LL, _LL, c_void_p, c_void_p, c_void_p, _LL, c_int, c_int, c_void_p, _LL, _LL,
                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dm_shade_bwd": (c_int, [POINTER(EnvAtlasStruct), POINTER(MatCfgStruct), c_void_p, _LL, _LL, c_void_p, _LL, _LL,
                             c_void_p, _LL, _LL, c_void_p, c_void_p, c_void_p, _LL, c_int, c_int, c_void_p, _LL, _LL,
                             c_void_p, _LL, _LL, c_void_p]),
    "dm_matreg_fwd": (c_int, [c_void_p, _LL, _LL, c_void_p, _LL, _LL, c_void_p, _LL, c_void_p, c_void_p]),
    "dm_matreg_bwd": (c_int, [c_void_p, _LL, _LL, c_void_p, _LL, _LL, c_void_p, _LL, c_float, c_void_p, _LL, _LL,
                              c_void_p, _LL, _LL, c_void_p]),
    "dm_bvh_build": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dm_bvh_collapse4": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "dm_bvh_any_hit_rays": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, _LL, c_float, c_void_p, c_void_p]),
    "dm_grid_build": (c_int, [c_void_p, c_int32, c_int32, POINTER(GridStruct), POINTER(c_void_p), POINTER(c_longlong)]),
    "dm_host_free": (None, [c_void_p]),
    "dm_grid_any_hit_rays": (c_int, [POINTER(GridStruct), c_void_p, c_void_p, _LL, c_float, c_void_p, c_void_p]),
    "dm_mc_hit_words": (c_int, [c_int, c_int]),
    "dm_mc_shade_fwd": (c_int, [POINTER(McSceneStruct), POINTER(MatCfgStruct)] + [c_void_p, _LL, _LL] * 4
                        + [c_void_p, c_void_p, c_void_p, _LL, c_int, c_void_p, c_void_p, c_void_p, c_void_p, _LL, _LL]
                        + [c_void_p] * 8),
    "dm_mc_shade_bwd": (c_int, [POINTER(McSceneStruct), POINTER(MatCfgStruct)] + [c_void_p, _LL, _LL] * 4
                        + [c_void_p, c_void_p, c_void_p, _LL, c_int, c_void_p, c_void_p, c_void_p, c_void_p, _LL, _LL,
                           c_void_p, _LL, _LL, c_void_p]),
    "dm_attention_select": (c_int, [ctypes.c_char_p]),
    "dm_attention_selected": (ctypes.c_char_p, []),
    "dm_attention_fwd_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int]
                              + [_LL] * 12 + [c_float, c_void_p]),
    "dm_attention_fwd_lse_bf16": (c_int, [c_void_p] * 5 + [c_int] * 5 + [_LL] * 9 + [c_float, c_void_p]),
    "dm_attention_bwd_bf16": (c_int, [c_void_p] * 10 + [c_int] * 5 + [_LL] * 6 + [c_float, c_void_p]),
    "dm_attention_fp8_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "dm_attention_fwd_fp8": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int] + [_LL] * 12
                             + [c_float, c_int, c_void_p, c_size_t, c_void_p]),
    "dm_conv3x3_nhwc_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p]),
    "dm_conv3x3_nhwc_bf16_fused": (c_int, [c_void_p] * 6 + [c_int] * 10 + [c_void_p]),
    "dm_conv3x3_wgrad_splits": (c_int, [c_int] * 5),
    "dm_conv3x3_wgrad_nhwc_bf16": (c_int, [c_void_p] * 3 + [c_int] * 6 + [c_void_p]),
    "dm_conv2x2_nhwc_bf16": (c_int, [c_void_p] * 4 + [c_int] * 9 + [c_void_p]),
    "dm_conv2x2_subpixel_nhwc_bf16": (c_int, [c_void_p] * 4 + [c_int] * 10 + [c_void_p]),
    "dm_conv3x3_small_nhwc_bf16": (c_int, [c_void_p] * 4 + [c_int] * 11 + [c_void_p]),
    "dm_conv3x3_small_res_nhwc_bf16": (c_int, [c_void_p] * 4 + [c_int, c_void_p] + [c_int] * 11 + [c_void_p]),
    "dm_gemm_bf16_fused": (c_int, [c_void_p] * 5 + [_LL, c_int, c_int, c_int, c_void_p]),
    "dm_gemm_bf16_batched": (c_int, [c_void_p] * 3 + [c_int, _LL, c_int, c_int, c_void_p]),
    "dm_groupnorm_workspace_floats": (c_size_t, [c_int, c_int]),
    "dm_groupnorm_nhwc_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                      c_int, c_void_p]),
    "dm_linear_small_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, _LL, c_int, c_int, c_void_p]),
    "dm_groupnorm_nhwc_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "dm_conv3x3_gn_ok": (c_int, [c_int] * 5),
    "dm_conv3x3_gn_nhwc_bf16_fused": (c_int, [c_void_p, c_void_p, c_int] + [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "dm_groupnorm_nhwc_infer": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float,
                                        c_int, c_void_p]),
    "dm_groupnorm_nhwc_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                      c_float, c_int, c_void_p]),
    "dm_groupnorm_nhwc_bwd_res": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "dm_groupnorm_affine_rows": (c_int, [c_int, c_int, c_int]),
    "dm_groupnorm_nhwc_bwd_affine": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "dm_layernorm_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, _LL, c_int, c_float, c_void_p]),
    "dm_geglu_bf16": (c_int, [c_void_p, c_void_p, _LL, c_int, c_void_p]),
    "dm_cat_add_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, _LL, c_int, c_int, c_float, c_void_p]),
    "dm_transpose_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dm_softmax_rows_bf16": (c_int, [c_void_p, c_void_p, _LL, c_int, c_float, c_void_p]),
    "dm_softmax_rows_bwd_bf16": (c_int, [c_void_p, c_void_p, c_void_p, _LL, c_int, c_float, c_void_p]),
    "dm_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, _LL, c_int, c_float, c_float, c_float, c_float,
                             c_float, c_int, c_void_p]),
}


# IEEE-half instantiations of the net kernels (csrc/dm_elem.h, built from the same sources with -DDM_F16): same signatures
for _n in ("dm_attention_fwd_bf16", "dm_attention_fwd_lse_bf16", "dm_conv3x3_nhwc_bf16", "dm_conv3x3_nhwc_bf16_fused",
           "dm_conv2x2_nhwc_bf16", "dm_conv2x2_subpixel_nhwc_bf16", "dm_conv3x3_small_nhwc_bf16", "dm_conv3x3_small_res_nhwc_bf16", "dm_gemm_bf16_fused", "dm_gemm_bf16_batched",
           "dm_layernorm_bf16", "dm_geglu_bf16", "dm_cat_add_bf16", "dm_softmax_rows_bf16", "dm_softmax_rows_bwd_bf16",
           "dm_conv3x3_gn_nhwc_bf16_fused", "dm_linear_small_bf16", "dm_transpose_bf16"):
    _SIGS[_n.replace("bf16", "f16")] = _SIGS[_n]
for _n in ("dm_groupnorm_nhwc_fwd", "dm_groupnorm_nhwc_infer", "dm_groupnorm_nhwc_bwd", "dm_groupnorm_nhwc_bwd_res", "dm_groupnorm_nhwc_stats"):
    _SIGS[_n + "_f16"] = _SIGS[_n]
del _n


def lib():
    """Loads the shared library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DmError(
                f"{LIB_PATH} is missing: build it with `python -m dreammat_amd.csrc.build` "
                "(hipcc --offload-arch=gfx950). The HIP library is mandatory: there is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.dm_abi_version() != ABI_VERSION:
            raise DmError(f"{LIB_PATH} has ABI version {L.dm_abi_version()}, this binding expects {ABI_VERSION}: rebuild it "
                          "(`python -m dreammat_amd.csrc.build`)")
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        if rc < 0:
            raise DmError(f"{what}: {DM_ERRORS.get(rc, rc)}")
        raise DmError(f"{what}: hipError {rc}")


def exported_symbols():
    return sorted(_SIGS.keys())
