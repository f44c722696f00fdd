"""ctypes binding of libl2i_hip.so (the C ABI declared in include/l2i.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is no CPU fallback: every
op in this package raises if the library is missing or a kernel call is made without a GPU tensor.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('L2I_LIB') or os.path.join(_HERE, 'libl2i_hip.so')      # L2I_LIB: another build of the library (A/B probes)

ACT_NONE, ACT_LRELU, ACT_RELU = 0, 1, 2

c_f = ctypes.c_float
c_i = ctypes.c_int32
c_p = ctypes.c_void_p
c_l = ctypes.c_int64


class ConvParams(ctypes.Structure):
    """Mirror of ``struct l2i_conv_params`` (include/l2i.h) — field order and types must match."""
    _fields_ = [
        ('x', c_p), ('w', c_p), ('y', c_p),
        ('B', c_i), ('Cin', c_i), ('H', c_i), ('W', c_i), ('Cout', c_i), ('CoutP', c_i),
        ('KH', c_i), ('KW', c_i), ('stride', c_i), ('pad_y', c_i), ('pad_x', c_i),
        ('OH', c_i), ('OW', c_i), ('OHf', c_i), ('OWf', c_i),
        ('oy_step', c_i), ('ox_step', c_i), ('oy_off', c_i), ('ox_off', c_i),
        ('in_scale', c_p), ('in_mask', c_p), ('mask_pos', c_f), ('mask_neg', c_f),
        ('out_scale', c_p), ('noise', c_p), ('noise_w', c_f), ('bias', c_p),
        ('residual', c_p), ('res_mask', c_p), ('out_mask', c_p),
        ('act', c_i), ('act_slope', c_f), ('act_gain', c_f), ('out_gain', c_f),
        ('accumulate', c_i), ('tile_hint', c_i),
        ('w_hi', c_p), ('w_lo', c_p),
        ('ws', c_p), ('ksplit', c_i), ('res_sub', c_p), ('res_coef', c_f), ('res_coef_dev', c_p),
        ('sq_ref', c_p), ('sq_out', c_p),
        ('w_bstride', c_l), ('out_f32', c_i),
        ('in_h8', c_i), ('rgb_w', c_p), ('rgb_bias', c_p), ('rgb_out', c_p), ('pool_out', c_p), ('pool_idx', c_p),
        ('mask_out', c_p), ('mask_bits', c_i),
    ]


SQ_SLOTS = 1024          # L2I_SQ_SLOTS


class SegmvPart(ctypes.Structure):
    """Mirror of ``struct l2i_segmv_part`` (include/l2i.h)."""
    _fields_ = [('K', c_i), ('in_off_c', c_i), ('in_off_b', c_i), ('in_bstride', c_i), ('w_pitch', c_i), ('pre', c_i), ('aux_off', c_i), ('pad_', c_i),
                ('w_off', c_l)]


class SegmvSeg(ctypes.Structure):
    """Mirror of ``struct l2i_segmv_seg`` (include/l2i.h)."""
    _fields_ = [('rows', c_i), ('nparts', c_i), ('out_off_c', c_i), ('out_off_b', c_i), ('out_bstride', c_i), ('epi', c_i), ('bias_off', c_i),
                ('e_off_c', c_i), ('e_off_b', c_i), ('e_bstride', c_i), ('rgb_off_b', c_i), ('rgb_w_off', c_i), ('part', SegmvPart * 2)]


_SIGNATURES = {
    'l2i_conv2d_f32': (c_i, [ctypes.POINTER(ConvParams), c_p]),
    'l2i_conv2d_family': (c_i, [ctypes.POINTER(ConvParams)]),
    'l2i_conv_transpose2d_f32': (c_i, [ctypes.POINTER(ConvParams), c_p]),
    'l2i_conv2d_bf16x3_f32': (c_i, [ctypes.POINTER(ConvParams), c_p]),
    'l2i_conv_transpose2d_bf16x3_f32': (c_i, [ctypes.POINTER(ConvParams), c_p]),
    'l2i_conv2d_wino_f32': (c_i, [ctypes.POINTER(ConvParams), c_p]),
    'l2i_conv2d_wino4_f32': (c_i, [ctypes.POINTER(ConvParams), c_p]),
    'l2i_conv1x1_pair_f32': (c_i, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvParams), c_p]),
    'l2i_conv2d_h8': (c_i, [ctypes.POINTER(ConvParams), c_p]),
    'l2i_conv_transpose2d_h8': (c_i, [ctypes.POINTER(ConvParams), c_p]),
    'l2i_conv_img_h8': (c_i, [ctypes.POINTER(ConvParams), c_p]),
    'l2i_conv1x1_pair_h8': (c_i, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvParams), c_i, c_p]),
    'l2i_conv_chain3_h8': (c_i, [ctypes.POINTER(ConvParams), ctypes.POINTER(ConvParams), ctypes.POINTER(ConvParams), c_i, c_p]),
    'l2i_fused_bias_act_f32': (c_i, [c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_i, c_i, c_f, c_f, c_p]),
    'l2i_fused_bias_act_f16': (c_i, [c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_i, c_i, c_f, c_f, c_p]),
    'l2i_upfirdn2d_f16': (c_i, [c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'l2i_upfirdn2d_f32': (c_i, [c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                c_i, c_p, c_f, c_p, c_p, c_i, c_f, c_f, c_p]),
    'l2i_upfirdn2d_masked_f32': (c_i, [c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                                       c_i, c_p, c_f, c_p, c_p, c_i, c_f, c_f, c_p, c_f, c_f, c_p]),
    'l2i_torgb_fwd_f32': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_l, c_p]),
    'l2i_sg2_act_bwd_f32': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_f, c_p, c_p, c_p, c_i, c_i, c_l, c_p]),
    'l2i_dot_reduce_f32': (c_i, [c_p, c_p, c_p, c_l, c_l, c_p]),
    'l2i_maxpool2d_fwd_f32': (c_i, [c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'l2i_maxpool2d_bwd_f32': (c_i, [c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'l2i_maxpool2x2_bwd_add_diff_f32': (c_i, [c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_l, c_i, c_i, c_p]),
    'l2i_sqdiff_f32': (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_p, c_p]),
    'l2i_axpby_f32': (c_i, [c_p, c_p, c_p, c_f, c_f, c_l, c_p]),
    'l2i_relu_mask_f32': (c_i, [c_p, c_p, c_p, c_l, c_p]),
    'l2i_conv2d_wgrad_f32': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'l2i_bn_stats_f32': (c_i, [c_p, c_p, c_p, c_i, c_i, c_l, c_p]),
    'l2i_bn_apply_f32': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_l, c_p]),
    'l2i_bn_bwd_reduce_f32': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_l, c_p]),
    'l2i_bn_bwd_apply_f32': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_l, c_p]),
    'l2i_pixelnorm_act_f32': (c_i, [c_p, c_p, c_i, c_i, c_l, c_f, c_f, c_p]),
    'l2i_pixelnorm_act_bwd_f32': (c_i, [c_p, c_p, c_p, c_i, c_i, c_l, c_f, c_f, c_p]),
    'l2i_upsample2x_nearest_f32': (c_i, [c_p, c_p, c_l, c_i, c_i, c_f, c_p]),
    'l2i_pool2x2_f32': (c_i, [c_p, c_p, c_l, c_i, c_i, c_f, c_p]),
    'l2i_cast_f32_to_h8': (c_i, [c_p, c_p, c_i, c_i, c_i, c_l, c_p]),
    'l2i_cast_h8_to_f32': (c_i, [c_p, c_p, c_i, c_i, c_i, c_l, c_p]),
    'l2i_upfirdn2d_h8': (c_i, [c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_f, c_p, c_i, c_f, c_f, c_p, c_f, c_f, c_p, c_p, c_p, c_i, c_p]),
    'l2i_torgb_fwd_h8': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_l, c_p]),
    'l2i_sg2_act_bwd_h8': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_f, c_p, c_p, c_p, c_i, c_i, c_l, c_p]),
    'l2i_dot_reduce_h8': (c_i, [c_p, c_p, c_p, c_i, c_i, c_l, c_p]),
    'l2i_maxpool2d_fwd_h8': (c_i, [c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'l2i_maxpool2d_bwd_h8': (c_i, [c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_l, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    'l2i_sqdiff_h8': (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_p, c_p]),
    'l2i_add_zero_insert_h8': (c_i, [c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_i, c_p]),
    'l2i_mask_mul_h8': (c_i, [c_p, c_p, c_p, c_f, c_f, c_l, c_p]),
    'l2i_mask_mul_bits_h8': (c_i, [c_p, c_p, c_p, c_f, c_f, c_l, c_p]),
    'l2i_modulate_planes_h8': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'l2i_modulate_planes_multi_h8': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    'l2i_segmented_matvec_f32': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p]),
    'l2i_reg_bce_f32': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p]),
    'l2i_nonfinite_flag_f32': (c_i, [c_p, c_l, c_p, c_p]),
    'l2i_adam_guarded_f32': (c_i, [c_p, c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_i, c_p, c_p, c_f, c_f, c_i, c_f, c_i, c_p]),
    'l2i_last_error': (ctypes.c_char_p, []),
    'l2i_abi_version': (c_i, []),
    'l2i_sizeof_conv_params': (c_i, []),
}

# [r5] every h8 entry point exists twice: bf16 elements (the name as is) and IEEE fp16 elements (suffix _f16), same signatures
for _n in [k for k in _SIGNATURES if k.endswith('_h8') or k in ('l2i_cast_f32_to_h8', 'l2i_cast_h8_to_f32')]:
    _SIGNATURES[_n + '_f16'] = _SIGNATURES[_n]

ABI_VERSION = 7          # L2I_ABI_VERSION of include/l2i.h this binding mirrors

EXPORTS = tuple(_SIGNATURES)

_lib = None


BUILD_HASHES = os.path.join(_HERE, 'csrc', 'BUILD_HASHES.json')      # tools/update_build_hash.py: library sha256 of the committed sources


def committed_build():
    """What tools/update_build_hash.py recorded for the committed kernel sources, and whether the library that would be loaded IS that build:
    dict(kernel_sources_sha256_16, library_sha256, hipcc, sources_match, library_match).  The build is reproducible (csrc/Makefile), so
    library_match = False with sources_match = True means a different compiler or a hand-built library."""
    import hashlib
    import json
    try:
        with open(BUILD_HASHES) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    rec['sources_match'] = rec.get('kernel_sources_sha256_16') == source_hash()
    try:
        with open(LIB_PATH, 'rb') as f:
            rec['library_match'] = hashlib.sha256(f.read()).hexdigest() == rec.get('library_sha256')
    except OSError:
        rec['library_match'] = False
    return rec


def source_hash():
    """sha256[:16] over the kernel SOURCES (csrc/*.hip, *.h, Makefile, include/l2i.h; names + contents, sorted).  The PMC traffic summaries
    under profiles/ are keyed to this and not to the .so file: a rebuild of the same sources in another directory hashes the binary
    differently, and the figure must survive the driver's own build()."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, 'csrc')
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(('.hip', '.h')) or f == 'Makefile')      # (not BUILD_HASHES.json)
    files.append(os.path.join(os.path.dirname(_HERE), 'include', 'l2i.h'))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


class L2IError(RuntimeError):
    pass


def load():
    """Load the shared library once; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise L2IError('%s is missing: run `python __graft_entry__.py build` (hipcc --offload-arch=gfx950); '
                           'there is no CPU fallback for the hot path' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)           # AttributeError if the export is absent
            fn.restype = res
            fn.argtypes = args
        # L2I_LIB may point at a foreign build: a library with another struct layout must not be called through this mirror
        if lib.l2i_abi_version() != ABI_VERSION or lib.l2i_sizeof_conv_params() != ctypes.sizeof(ConvParams):
            raise L2IError('%s has ABI version %d / sizeof(l2i_conv_params) %d, this binding mirrors version %d / %d bytes: rebuild it '
                           '(`python __graft_entry__.py build`)' % (LIB_PATH, lib.l2i_abi_version(), lib.l2i_sizeof_conv_params(), ABI_VERSION,
                                                                    ctypes.sizeof(ConvParams)))
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise L2IError('%s failed (%d): %s' % (what, rc, load().l2i_last_error().decode()))


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous float32/uint8 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise L2IError('l2i kernels take GPU tensors only (got a %s tensor): no CPU fallback' % t.device)
    if not t.is_contiguous():
        raise L2IError('l2i kernels take contiguous tensors')
    return ctypes.c_void_p(t.data_ptr())


def fptr(t):
    if t is not None and t.dtype != torch.float32:
        raise L2IError('expected float32, got %s' % t.dtype)
    return ptr(t)
