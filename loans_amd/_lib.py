"""ctypes binding of ``include/loans_hip.h`` (the C-ABI drop-in boundary).

The shared library is built in-tree by ``__graft_entry__.build()`` /
``make -C loans_amd/csrc``.  There is NO fallback: if the library is missing
or a symbol is absent, importing the compute path raises, and every kernel
wrapper raises ``HipKernelError`` on a non-zero return code.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libloans_hip.so')

MAX_TAPS = 64
F_RELU_IN, F_BIAS, F_STATS, F_MASK, F_ADDEND, F_ADDEND_MASK = 1, 2, 4, 8, 16, 32
F_DENSE = 64
F_OUT_BF16, F_GY_BF16 = 128, 256
F_BNSUMS = 512
F_AFFINE_IN = 1024
TILE_AUTO, TILE_128x128, TILE_128x64, TILE_64x64, TILE_256x64 = 0, 1, 2, 3, 4
TILE_256x128, TILE_DMA = 7, 16


class HipKernelError(RuntimeError):
    pass


class IgemmDesc(C.Structure):
    """Mirror of ``loans_igemm_desc``."""
    _fields_ = [(n, C.c_int32) for n in (
        'B', 'inH', 'inW', 'Cin', 'outH', 'outW', 'Cout', 'gridH', 'gridW',
        'osy', 'osx', 'oy0', 'ox0', 'isy', 'isx', 'ntaps', 'flags', 'tile')] + [
        ('dy', C.c_int8 * MAX_TAPS), ('dx', C.c_int8 * MAX_TAPS)]


REPACK_JOB_TAPS = 16


class RepackJob(C.Structure):
    """Mirror of ``loans_repack_job``."""
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p)] + [(n, C.c_int32) for n in ('Cout', 'Cin', 'src_taps', 'ntaps')] + \
               [('tapsel', C.c_int32 * REPACK_JOB_TAPS)] + [(n, C.c_int32) for n in ('first_tile', 'tiles_co', 'tiles_ci', 'dst_bf16')]


class PwPackJob(C.Structure):
    """Mirror of ``loans_pw_pack_job``."""
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p)] + [(n, C.c_int32) for n in ('Cout', 'Cin', 'first_unit', 'reserved')]


class SmallConv(C.Structure):
    """Mirror of ``loans_small_conv``."""
    _fields_ = [(n, C.c_int32) for n in ('k', 'stride', 'pad', 'outH', 'outW')]


_p = C.c_void_p
_i32, _i64, _f32, _f64 = C.c_int32, C.c_int64, C.c_float, C.c_double

# name -> argtypes, exactly the prototypes of include/loans_hip.h
SIGNATURES = {
    'loans_igemm_bf16s_splitk': [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(IgemmDesc), C.c_int32, C.c_void_p],
    'loans_igemm_finalize_bf16': [C.c_void_p] * 6 + [C.c_int32, C.c_int64, C.c_int32, C.c_void_p],
    'loans_augment_stage_u8': [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p],
    'loans_channel_mean_f32': [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p],
    'loans_channel_mean_bf16': [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p],
    'loans_vbp_scale_f32': [C.c_void_p] * 3 + [C.c_int32] * 11 + [C.c_void_p],
    'loans_minmax_normalize_f32': [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p],
    'loans_gray_fwd_f32': [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    'loans_gray_bwd_f32': [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    'loans_crop_dgrad_f32': [C.c_void_p, C.c_void_p, C.POINTER(SmallConv), C.c_void_p, C.c_void_p, C.POINTER(SmallConv),
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    'loans_crop_dgrad_bf16_f32': [C.c_void_p, C.c_void_p, C.POINTER(SmallConv), C.c_void_p, C.c_void_p, C.POINTER(SmallConv),
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p],
    'loans_igemm_f32': [_p, _p, _p, _p, _p, _p, _p, C.POINTER(IgemmDesc), _p],
    'loans_igemm_classes_f32': [_p, C.POINTER(_p), _p, _p, _p, C.POINTER(IgemmDesc), _i32, _p],
    'loans_igemm_pair_bf16s': [_p, _p, _p, _p, _p, C.POINTER(IgemmDesc), _p],
    'loans_pw_pack_bf16': [_p, _p, C.c_int32, C.c_int32, _p],
    'loans_pw_pack_batch_f32': [_p, C.c_int32, C.c_int32, _p],
    'loans_igemm_pair_f32': [_p, _p, _p, _p, _p, _p, _p, _i32, C.POINTER(IgemmDesc), _p],
    'loans_igemm_finalize_f32': [_p, _p, _p, _p, _p, _i32, _i64, _i32, _p],
    'loans_igemm_bf16_f32': [_p, _p, _p, _p, _p, _p, _p, C.POINTER(IgemmDesc), _p],
    'loans_wgrad_f32': [_p, _p, _p, C.POINTER(IgemmDesc), _i32, _p],
    'loans_wgrad_bf16_f32': [_p, _p, _p, C.POINTER(IgemmDesc), _i32, _p],
    'loans_igemm_bf16s': [_p, _p, _p, _p, _p, _p, _p, C.POINTER(IgemmDesc), _p],
    'loans_wgrad_bf16s': [_p, _p, _p, C.POINTER(IgemmDesc), _i32, _p],
    'loans_wgrad_bf16s_ws': [_p, _p, _p, C.POINTER(IgemmDesc), _i32, _p, _i64, _p],
    'loans_wgrad_bf16s_affine_ws': [_p, _p, _p, C.POINTER(IgemmDesc), _i32, _p, _i64, _p, _p],
    'loans_fold_slabs_f32': [_p, _p, _i64, _i32, _p],
    'loans_wgrad_bf16s_ws_floats': [C.POINTER(IgemmDesc), _i32],
    'loans_cast_bf16': [_p, _p, _i64, _p],
    'loans_repack_dgrad_bf16': [_p, _p, _i32, _i32, _i32, C.POINTER(_i32), _i32, _p],
    'loans_bn_apply_bf16': [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _p],
    'loans_bn_relu_maxpool_bf16': [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_maxpool_relu_bwd_bf16': [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_bn_bwd_reduce_bf16': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p],
    'loans_bn_bwd_apply_bf16': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p],
    'loans_colsum_bf16': [_p, _p, _i64, _i32, _p],
    'loans_bn_apply_bits_f32': [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _p],
    'loans_bn_apply_bits_bf16': [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _p],
    'loans_bn_bwd_reduce_bits_f32': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p],
    'loans_bn_bwd_reduce_bits_bf16': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p],
    'loans_bn_bwd_apply_bits_f32': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p],
    'loans_bn_bwd_apply_bits_bf16': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p],
    'loans_bn_bwd_reduce_xmask_f32': [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p],
    'loans_bn_bwd_reduce_xmask_bf16': [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p],
    'loans_bn_bwd_apply_xmask_f32': [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p],
    'loans_bn_bwd_apply_xmask_bf16': [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p],
    'loans_pool_bn_bwd_reduce_f32': [_p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_pool_bn_bwd_reduce_bf16': [_p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_pool_bn_bwd_apply_f32': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_pool_bn_bwd_apply_bf16': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_gap_fwd_bf16_f32': [_p, _p, _i32, _i32, _i32, _p],
    'loans_gap_bwd_f32_bf16': [_p, _p, _i32, _i32, _i32, _p],
    'loans_dgrad_c4_bf16_f32': [_p, _p, _p, _p, _p, C.POINTER(IgemmDesc), C.POINTER(_i32), _i32, _p],
    'loans_linear_fwd_bf16': [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_linear_bwd_bf16': [_p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_dgrad_c4_f32': [_p, _p, _p, _p, _p, C.POINTER(IgemmDesc), C.POINTER(_i32), _i32, _p],
    'loans_repack_dgrad_f32': [_p, _p, _i32, _i32, _i32, C.POINTER(_i32), _i32, _p],
    'loans_repack_dgrad_batch': [_p, _i32, _i32, _p],
    'loans_resize_lanczos_u8': [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _i32, _p, _p, _i32, _p],
    'loans_resize_lanczos_u8_f32': [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _i32, _p, _p, _i32, _p],
    'loans_resize_ragged_u8_f32': [_p, _p, _p, _p, _i32, _p, _i32, _i32, _i32, _p],
    'loans_u8hwc3_to_f32chw': [_p, _p, _i32, _i32, _i32, _p],
    'loans_prep_images_f32': [_p, _p, _i32, _i32, _i32, _p],
    'loans_prep_images_dense_f32': [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_prep_images_dense_bf16': [_p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_nchw3_to_nhwc4_f32': [_p, _p, _i32, _i32, _i32, _p],
    'loans_bn_finalize_f32': [_p, _i32, _i64, _f32, _f32, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _p],
    'loans_bn_eval_coeffs_f32': [_i32, _f32, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    'loans_bn_apply_f32': [_p, _p, _p, _p, _p, _p, _p, _i64, _i32, _i32, _i32, _p],
    'loans_bn_relu_maxpool_f32': [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_bn_relu_maxpool_sel_f32': [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_bn_relu_maxpool_sel_bf16': [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_maxpool_relu_bwd_f32': [_p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_bn_bwd_reduce_f32': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p],
    'loans_bn_bwd_coeffs_f32': [_p, _i32, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    'loans_bn_bwd_coeffs_rep_f32': [_p, _i32, _i32, _i32, _i32, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    'loans_pool_bn_bwd_reduce_rep_f32': [_p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_pool_bn_bwd_reduce_rep_bf16': [_p, _p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_pool_bn_bwd_apply_rep_f32': [_p] * 10 + [_i32] * 7 + [_p],
    'loans_pool_bn_bwd_apply_rep_bf16': [_p] * 10 + [_i32] * 7 + [_p],
    'loans_fold_replicas_f32': [_p, _p, _i32, _i32, _p],
    'loans_bn_bwd_reduce_rep_f32': [_p, _p, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i64, _i32, _p],
    'loans_bn_bwd_reduce_rep_bf16': [_p, _p, _i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _i64, _i32, _p],
    'loans_bn_bwd_apply_f32': [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p],
    'loans_colsum_f32': [_p, _p, _i64, _i32, _p],
    'loans_gap_fwd_f32': [_p, _p, _i32, _i32, _i32, _p],
    'loans_gap_bwd_f32': [_p, _p, _i32, _i32, _i32, _p],
    'loans_linear_fwd_f32': [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_linear_bwd_f32': [_p, _p, _p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_mul_f32': [_p, _p, _p, _i64, _p],
    'loans_axpby_f32': [_f32, _p, _f32, _p, _i64, _p],
    'loans_st_grid_fwd_f32': [_p, _p, _i32, _i32, _i32, _p],
    'loans_st_grid_bwd_f32': [_p, _p, _i32, _i32, _i32, _p],
    'loans_st_sampler_fwd_f32': [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_st_sampler_bwd_grid_f32': [_p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _i32, _p],
    'loans_mse_fwd_f32': [_p, _p, _f32, _p, _i32, _p],
    'loans_mse_bwd_f32': [_p, _p, _f32, _p, _p, _i32, _p],
    'loans_grid_loss_fwd_f32': [_p, _p, _i32, _f32, _f32, _f32, _i32, _i32, _i32, _p],
    'loans_grid_loss_bwd_f32': [_p, _p, _p, _i32, _f32, _f32, _f32, _i32, _i32, _i32, _p],
    'loans_adam_amsgrad_f32': [_p, _p, _p, _p, _p, _i64, _f64, _f64, _f64, _f64, _f64, _f64, _f64, _p],
    'loans_adam_amsgrad_devlr_f32': [_p, _p, _p, _p, _p, _i64, _p, _f64, _f64, _f64, _f64, _f64, _f64, _p],
    'loans_adam_f32': [_p, _p, _p, _p, _i64, _f64, _p, _f64, _f64, _f64, _f64, _f64, _f64, _p],
}

# every entry point returns int (0 / LOANS_E* / hipError_t) except:
RESTYPES = {'loans_wgrad_bf16s_ws_floats': C.c_int64}

_lib = None


def load():
    """Load (once) and return the ctypes library; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipKernelError(
            "HIP kernel library not built: %s is missing. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C loans_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = RESTYPES.get(name, C.c_int)
    lib.loans_hip_version.restype = C.c_char_p
    lib.loans_hip_version.argtypes = []
    _lib = lib
    return lib


def check(rc, name):
    if rc != 0:
        kind = {-1: 'LOANS_EINVAL (rejected arguments)', -2: 'LOANS_ERANGE (size beyond 32-bit indexing)'}.get(
            rc, 'hipError_t %d' % rc)
        raise HipKernelError('%s failed: %s' % (name, kind))
