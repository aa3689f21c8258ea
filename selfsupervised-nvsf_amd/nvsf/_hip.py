"""ctypes binding of libnvsf_hip.so (the C ABI declared in include/nvsf_hip.h).

This is the only place where Python touches the native library.  There is deliberately NO fallback:
if the shared object is missing or a call is made without a HIP device the caller gets an exception,
never a silently slower PyTorch/CPU path.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libnvsf_hip.so")

_P, _U, _F, _I = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_float, ctypes.c_int
_U64 = ctypes.c_uint64

# name -> argument ctypes (the trailing stream argument is appended automatically)
SIGNATURES = {
    # section 1: raymarching extension
    "nvsf_near_far_from_aabb": [_P, _P, _P, _U, _F, _P, _P],
    "nvsf_sph_from_ray": [_P, _P, _F, _U, _P],
    "nvsf_morton3D": [_P, _U, _P],
    "nvsf_morton3D_invert": [_P, _U, _P],
    "nvsf_packbits": [_P, _U, _F, _P],
    "nvsf_march_rays_train": [_P, _P, _P, _F, _F, _U, _U, _U, _U, _U, _P, _P, _P, _P, _P, _P, _P, _P],
    "nvsf_march_rays_train_passes": [_P, _P, _P, _F, _F, _U, _U, _U, _U, _U, _P, _P, _P, _P, _P, _P, _P, _P],
    "nvsf_march_rays_train_ws": [_P, _P, _P, _F, _F, _U, _U, _U, _U, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P, ctypes.c_size_t, _U],
    "nvsf_composite_rays_train_forward": [_P, _P, _P, _P, _U, _U, _F, _P, _P, _P],
    "nvsf_composite_rays_train_backward": [_P, _P, _P, _P, _P, _P, _P, _P, _U, _U, _F, _P, _P],
    "nvsf_march_rays": [_U, _U, _P, _P, _P, _P, _F, _F, _U, _U, _U, _P, _P, _P, _P, _P, _P, _P],
    "nvsf_composite_rays": [_U, _U, _F, _P, _P, _P, _P, _P, _P, _P, _P],
    # section 2: uniform sampler / compositor
    "nvsf_uniform_samples": [_P, _P, _P, _P, _P, _P, _P, _U, _U, _P, _P],
    "nvsf_composite_uniform_weights_fwd": [_P, _P, _P, _P, _U, _U, _F, _P, _P, _P],
    "nvsf_composite_uniform_weights_bwd": [_P, _P, _P, _P, _P, _P, _P, _U, _U, _F, _P],
    "nvsf_composite_uniform_image_fwd": [_P, _P, _P, _U, _U, _U, _P, _P],
    "nvsf_composite_uniform_image_bwd": [_P, _P, _P, _U, _U, _U, _P, _P, _P, _P],
    # section 3: field operators
    "nvsf_hashgrid_fwd": [_P, _U, _U, _P, _U, _P, _U, _U, _P, _P, _P, _P, _U],
    "nvsf_hashgrid_fwd_level_major": [_P, _U, _U, _P, _U, _U, _P, _P, _P, _P],
    "nvsf_hashgrid_bwd": [_P, _U, _U, _P, _U, _U, _U, _P, _P, _P, _P, _I, _U, _P],
    "nvsf_hashgrid_bwd_binned": [_P, _U, _U, _P, _U, _U, _U, _P, _P, _P, _P, _I, _U, _U, _P, _U, _U, _P, ctypes.c_size_t],
    "nvsf_freq_encode": [_P, _U, _U, _U, _P, _U],
    "nvsf_sh4_encode": [_P, _U, _P, _U],
    "nvsf_adam_prepare": [_P, _P, _F, _F],
    "nvsf_adam_update": [_P, _P, _P, _P, _U64, _F, _F, _F, _F, _P, _P, _P, _F, _P],
    "nvsf_ema_update": [_P, _P, _U64, _F],
    "nvsf_sigma_geo_bwd": [_P, _P, _P, _U, _U, _U, _P, _U, _F, _F],
    "nvsf_density_tail_grad_split": [_P, _U, _U, _P, _P, _P, _I, _I, _P, _I, _P, _F],
    "nvsf_cast_cols_f16": [_P, _I, _U, _U, _U, _P, _U],
    "nvsf_masked_sigmoid": [_P, _U, _U, _P, _U, _U, _P],
    "nvsf_sigmoid_bwd": [_P, _P, _U, _P],
    "nvsf_exp_col": [_P, _U, _U, _U, _P],
    "nvsf_count_nonzero_u8": [_P, ctypes.c_uint64, _P],
    "nvsf_repeat_rows_f16": [_P, _U, _U, _U, _U, _P, _U],
    "nvsf_heads_input_f16": [_P, _U, _U, _U, _U, _P, _I, _U, _U, _P, _U, _U],
    "nvsf_mlp_fwd": [_P, _I, _U, _U, _U, _P, _U, _U, _U, _U, _P, _U],
    "nvsf_mlp_bwd": [_P, _I, _U, _U, _U, _P, _U, _U, _U, _U, _P, _U, _U, _F, _P, _U, _P, _U, _I],
    "nvsf_mlp_fwd_prefix": [_P, _U, _U, _U, _P, _U, _U, _U, _P, _U, _U, _U, _U, _P, _U],
    "nvsf_mlp_bwd_density": [_P, _I, _U, _U, _U, _P, _U, _U, _U, _U, _P, _P, _P, _P, _U, _U, _F, _F, _F, _P, _U, _P, _U, _I],
    "nvsf_mlp_bwd_prefix": [_P, _U, _U, _U, _P, _U, _U, _U, _P, _U, _U, _U, _U, _P, _U, _U, _F, _P, _U, _P, _U, _I],
    "nvsf_planes_fwd": [_P, _U, _P, _U, _U, _P, _I, _P, _P],
    "nvsf_planes_multi_fwd": [_P, _U, _U, _P, _U, _U, _P, _U, _P, _P, _P, _P, _P, _P, _I],
    "nvsf_planes_multi_bwd": [_P, _U, _U, _P, _U, _U, _P, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "nvsf_planes_bwd": [_P, _U, _P, _U, _U, _P, _I, _P, _P, _P, _P],
    "nvsf_hashgrid4d_dynamic_fwd": [_P, _U, _P, _U, _U, _U, _P, _P, _P, _P, _P, _I, _I, _P],
    "nvsf_hashgrid4d_dynamic3_fwd": [_P, _U, _P, _U, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "nvsf_hashgrid4d_dynamic_bwd_scalar": [_P, _U, _U, _P, _P, _P, _P, _P],
    "nvsf_hashgrid4d_dynamic_bwd_scalar_t": [_P, _U, _U, _P, _P, _P, _P, _P],
    "nvsf_hashgrid4d_dynamic_bwd": [_P, _U, _U, _P, _P, _P, _P, _I, _P, _P],
    "nvsf_hashgrid3d_lagrange_fwd": [_P, _U, _U, _P, _U, _U, _P, _P, _P, _P, _P],
    "nvsf_density_dynamic_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _U, _P, _P, _P, _P, _P],
    "nvsf_lidar_losses_fwd": [_P, _P, _P, _P, _P, _P, _U, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P],
    "nvsf_lidar_losses_bwd": [_P, _P, _P, _P, _P, _P, _U, _F, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P],
    "nvsf_lidar_grad_loss_fwd": [_P, _P, _P, _P, _P, _U, _U, _U, _U, _U, _U, _F, _I, _F, _F, _I, _P, _P],
    "nvsf_lidar_grad_loss_bwd": [_P, _P, _P, _P, _P, _U, _U, _U, _U, _U, _U, _F, _I, _F, _F, _I, _P, _P, _P],
    "nvsf_lidar_ray_losses": [_P, _P, _P, _P, _P, _U, _F, _F, _F, _F, _P, _P],
    "nvsf_mse_rows": [_P, _P, _U, _U, _F, _P, _P],
    "nvsf_error_map_update": [_P, _P, _U, _U, _P, _U, _U, _F, _F, _P, _P],
    "nvsf_mse_sum_fwd": [_P, _P, _U, _F, _P],
    "nvsf_mse_sum_bwd": [_P, _P, _U, _F, _P, _P],
    "nvsf_density_dynamic_f16planes_fwd": [_P, _P, _P, _P, _P, _I, _P, _I, _U, _P, _P, _P, _P, _P],
    "nvsf_density_dynamic_lm_fwd": [_P, _P, _P, _P, _P, _I, _P, _I, _U, _P, _P, _P, _P, _P],
    "nvsf_density_dynamic_lm32_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _U, _P, _P, _P, _P, _P],
    # section 5: ray generation
    "nvsf_lidar_rays": [_P, _P, _U, _U, _U, _F, _F, _F, _P, _P],
    "nvsf_camera_rays": [_P, _P, _U, _U, _F, _F, _F, _F, _P, _P],
    # section 5: chamfer distance
    "nvsf_chamfer_forward": [_P, _P, _U, _U, _U, _P, _P, _P, _P, _P],
    "nvsf_chamfer_backward": [_P, _P, _U, _U, _U, _P, _P, _P, _P, _P, _P],
    # section 4: fused uniform-render kernels
    "nvsf_field_density_uniform_fwd": [_P, _P, _P, _P, _P, _P, _P, _F, _U, _U, _P, _U, _U, _P, _P, _P, _P, _P, _P, _P],
    "nvsf_field_density_uniform_train_fwd": [_P, _P, _P, _P, _P, _P, _P, _F, _U, _U, _P, _U, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "nvsf_field_density_uniform_sliced_fwd": [_P, _P, _P, _P, _P, _P, _P, _F, _U, _U, _P, _U, _U, _P, _P, _P, _P, _P, _P, _P, _P, _U],
    "nvsf_render_uniform_fwd": [_P, _P, _P, _P, _P, _P, _P, _F, _U, _U, _P, _U, _U, _P, _P, _P, _P, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P,
                                _P, _P],
    "nvsf_render_uniform_train_fwd": [_P, _P, _P, _P, _P, _P, _P, _F, _U, _U, _P, _U, _U, _P, _P, _P, _P, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P,
                                      _P, _P, _P, _P, _P, _P, _P],
    "nvsf_render_occupancy_fwd": [_P, _P, _P, _P, _P, _F, _F, _U, _U, _U, _U, _P, _U, _U, _P, _P, _P, _P, _I, _P, _P, _F, _F, _P, _P, _P, _P],
    "nvsf_field_heads_uniform_fwd": [_P, _P, _P, _P, _I, _P, _P, _U, _U, _F, _P, _P],
}

_lib = None


class NvsfHipError(RuntimeError):
    pass


def load():
    """Loads libnvsf_hip.so (once).  Raises NvsfHipError when the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NvsfHipError(
            f"{LIB_PATH} is missing: build it with `python selfsupervised-nvsf_amd/build.py` "
            "(or __graft_entry__.build()).  There is no CPU / PyTorch fallback for the NVSF hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == header / library mismatch
        fn.argtypes = list(argtypes) + [_P]
        fn.restype = ctypes.c_int
    lib.nvsf_scratch_pool_stats.restype = ctypes.c_int
    lib.nvsf_scratch_pool_stats.argtypes = [ctypes.POINTER(ctypes.c_uint64)] * 3
    lib.nvsf_march_rays_train_ws_bytes.restype = ctypes.c_size_t
    lib.nvsf_march_rays_train_ws_bytes.argtypes = [_U]
    lib.nvsf_hashgrid_bwd_binned_ws_bytes.restype = ctypes.c_size_t
    lib.nvsf_hashgrid_bwd_binned_ws_bytes.argtypes = [_U, _U, _U, _P, _P, _U, _U]
    lib.nvsf_version.restype = ctypes.c_char_p
    lib.nvsf_version.argtypes = []
    lib.nvsf_build_digest.restype = ctypes.c_char_p
    lib.nvsf_build_digest.argtypes = [ctypes.c_int]
    lib.nvsf_test_variant.restype = ctypes.c_int
    lib.nvsf_test_variant.argtypes = [ctypes.c_char_p, ctypes.c_int]
    _lib = lib
    return lib


def version():
    return load().nvsf_version().decode()


def build_digest(render_only=False):
    """Digest of the sources the MAPPED library was compiled from (csrc/version.hip), not of the sources on disk."""
    return load().nvsf_build_digest(1 if render_only else 0).decode()


def march_ws_bytes(n_rays):
    """Scratch bytes nvsf_march_rays_train_ws needs for `n_rays` rays."""
    return int(load().nvsf_march_rays_train_ws_bytes(int(n_rays)))


def hashgrid_bwd_ws_bytes(n_rows, spec, merge_from, fine_from):
    """Scratch bytes nvsf_hashgrid_bwd_binned needs for a grid (field_ops.GridSpec) (0: the shape has no binned form)."""
    return int(load().nvsf_hashgrid_bwd_binned_ws_bytes(int(n_rows), spec.L, spec.F, spec.h_res, spec.h_offsets, int(merge_from), int(fine_from)))


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must live on a HIP device."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NvsfHipError("NVSF HIP kernels need tensors on a HIP device (got a CPU tensor); there is no CPU fallback")
    if not t.is_contiguous():
        raise NvsfHipError("NVSF HIP kernels need contiguous tensors")
    return t.data_ptr()


def ptr_rows(t):
    """Device pointer of a 2-D tensor whose rows are dense (stride(1) == 1) but may be separated by padding / belong to a
    wider buffer: the kernels that take an explicit row stride accept such views without a copy."""
    if not t.is_cuda:
        raise NvsfHipError("NVSF HIP kernels need tensors on a HIP device (got a CPU tensor); there is no CPU fallback")
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise NvsfHipError("expected a 2-D tensor with unit column stride")
    return t.data_ptr()


def host_f32(values):
    return (ctypes.c_float * len(values))(*[float(v) for v in values])


def host_u32(values):
    return (ctypes.c_uint32 * len(values))(*[int(v) for v in values])


def host_i32(values):
    return (ctypes.c_int * len(values))(*[int(v) for v in values])


def call(name, *args):
    """Launches `name` on PyTorch's current HIP stream and raises on a non-zero status."""
    lib = load()
    stream = torch.cuda.current_stream().cuda_stream
    status = getattr(lib, name)(*args, stream)
    if status != 0:
        kind = "rejected arguments" if status < 0 else "hipError_t"
        raise NvsfHipError(f"{name} failed with status {status} ({kind})")
