"""ctypes binding of libunflow_hip.so (include/unflow_hip.h).

There is no CPU fallback: if the library is missing or a symbol is absent, importing the ops
fails loudly.  Signatures here are the single source of truth for the Python side and are
checked against the header by tests/test_abi.py.
"""
import ctypes
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, 'libunflow_hip.so')

_P = ctypes.c_void_p
_I = ctypes.c_int

# name -> argument types (all return int)
SIGNATURES = {
    'unflow_abi_version': [],
    'unflow_timing_reserve': [_I],
    'unflow_timing_begin': [],
    'unflow_timing_end': [],
    'unflow_timing_elapsed_us': [_I, _P],
    'unflow_timing_reset': [],
    'unflow_partials_per_sample': [_I, _I],
    'unflow_corr_fwd': [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    'unflow_corr_bwd': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'unflow_corr_bwd_ex': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'unflow_warp_fwd': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'unflow_warp_bwd': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'unflow_warp_bwd_det': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'unflow_warp_bwd_fused_supported': [_I, _I, _I, _I],
    'unflow_warp_bwd_table_bytes': [_I, _I, _I, _I],
    'unflow_warp_bwd_fused': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'unflow_warp_fwd_table': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'unflow_warp_corr_supported': [_I, _I, _I, _I],
    'unflow_warp_corr_fwd': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'unflow_warp_corr_bwd': [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'unflow_occ_weight_fwd': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    'unflow_absdiff_bwd': [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    'unflow_masked_mean_fwd': [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    'unflow_masked_mean_bwd': [_P, _P, _P, _P, _I, _I, _I, _P],
    'unflow_ssim_loss_fwd': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    'unflow_ssim_loss_bwd': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    'unflow_ssim_map': [_P, _P, _P, _I, _I, _I, _I, _P],
    'unflow_ssim_map_bwd': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    'unflow_smooth2_fwd': [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    'unflow_smooth2_bwd': [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    'unflow_consis_fwd': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    'unflow_consis_bwd': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    'unflow_bias_leaky_fwd': [_P, _P, _I, _I, _I, _I, ctypes.c_float, _P],
    'unflow_bias_leaky_partials': [_I, _I, _I, _I],
    'unflow_bias_leaky_bwd': [_P, _P, _P, _P, _P, _I, _I, _I, _I, ctypes.c_float, _P],
    'unflow_bias_leaky_bwd2': [_P, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _P, _P, _P, _I, _I, _I, _I, ctypes.c_float, _P],
    'unflow_bias_leaky_fwd_nhwc': [_P, _P, ctypes.c_longlong, _I, ctypes.c_float, _P],
    'unflow_bias_leaky_partials_nhwc': [ctypes.c_longlong, _I],
    'unflow_bias_leaky_bwd2_nhwc': [_P, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _P, _P, _P, ctypes.c_longlong, _I, ctypes.c_float, _P],
    'unflow_to_nchw_dup': [_P, _P, _I, _I, _I, _I, _P],
    'unflow_to_nchw_dup_bf16': [_P, _P, _I, _I, _I, _I, _P],
    'unflow_to_nhwc_fold': [_P, _P, _I, _I, _I, _I, _P],
    'unflow_to_nhwc_fold_bf16': [_P, _P, _I, _I, _I, _I, _P],
    'unflow_flow_head_partials': [],
    'unflow_flow_head_fwd': [_P, _P, _P, _P, _I, _I, _P],
    'unflow_flow_head_fwd_bf16': [_P, _P, _P, _P, _I, _I, _P],
    'unflow_flow_head_bwd': [_P, _P, _P, _P, _I, _I, _P],
    'unflow_flow_head_bwd_bf16': [_P, _P, _P, _P, _I, _I, _P],
    'unflow_bias_grad_finalize_batch': [_P, _P, _P, _P, _P, _I, _P],
    'unflow_adam_chunk': [],
    'unflow_loss_partial_blocks': [_I, _I, _I, _I, _I],
    'unflow_loss_finalize_batch': [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    'unflow_occ_weight_fwd_ms': [_I, _P, _P, _P, _P, _P, _P, _I, _P],
    'unflow_absdiff_bwd_ms': [_I, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    'unflow_masked_mean_fwd_ms': [_I, _P, _P, _P, _P, _P, _I, _P],
    'unflow_masked_mean_bwd_ms': [_I, _P, _P, _P, _P, _P, _P, _I, _P],
    'unflow_ssim_loss_fwd_ms': [_I, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    'unflow_ssim_loss_bwd_ms': [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    'unflow_smooth2_fwd_ms': [_I, _P, _P, _P, _P, _P, _I, _I, _P],
    'unflow_smooth2_bwd_ms': [_I, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    'unflow_consis_fwd_ms': [_I, _P, _P, _P, _P, _P, _P, _I, _P],
    'unflow_consis_bwd_ms': [_I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    'unflow_warp_fwd_ms': [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    'unflow_warp_bwd_ms': [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    'unflow_adam_multi': [_P, _P, _I, _P, _I, _P, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _P],
    'unflow_loss_combine_fwd': [_P, _I, _I, _P, _P],
    'unflow_loss_combine_bwd': [_P, _I, _P, _P],
    'unflow_weighted_mean_sum_fwd': [_P, _P, _I, _I, _P, _P],
    'unflow_weighted_mean_sum_bwd': [_P, _P, _I, _I, _P, _P],
    'unflow_upsample_scaled_fwd': [_P, _P, _I, _I, _I, _I, _I, ctypes.c_float, _P],
    'unflow_upsample_scaled_bwd': [_P, _P, _I, _I, _I, _I, _I, ctypes.c_float, _P],
    'unflow_bias_leaky_fwd_nhwc_to': [_P, _P, ctypes.c_longlong, _I, ctypes.c_float, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _P],
    'unflow_bias_leaky_bwd2_nhwc_from': [_P, ctypes.c_longlong, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _P, _P, _P, ctypes.c_longlong, _I, ctypes.c_float, _P],
    'unflow_bias_leaky_fwd_nhwc_to_bf16': [_P, _P, ctypes.c_longlong, _I, ctypes.c_float, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _P],
    'unflow_bias_leaky_bwd2_nhwc_from_bf16': [_P, ctypes.c_longlong, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _P, _P, _P, ctypes.c_longlong, _I, ctypes.c_float, _P],
    'unflow_bias_leaky_fwd_nhwc_bf16': [_P, _P, ctypes.c_longlong, _I, ctypes.c_float, _P],
    'unflow_bias_leaky_bwd2_nhwc_bf16': [_P, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _P, _P, _P, ctypes.c_longlong, _I, ctypes.c_float, _P],
    'unflow_cat_nhwc': [_P, _I, _P, _I, _P, _I, _P, _I, _I, _P],
    'unflow_split_nhwc': [_P, _P, _I, _P, _I, _P, _I, _I, _I, _P],
    'unflow_cat_nhwc_bf16': [_P, _I, _P, _I, _P, _I, _P, _I, _I, _P],
    'unflow_split_nhwc_bf16': [_P, _P, _I, _P, _I, _P, _I, _I, _I, _P],
    'unflow_bias_leaky_fwd_bf16': [_P, _P, _I, _I, _I, _I, ctypes.c_float, _P],
    'unflow_bias_leaky_bwd2_bf16': [_P, _P, ctypes.c_longlong, _P, ctypes.c_longlong, _P, _P, _P, _I, _I, _I, _I, ctypes.c_float, _P],
    'unflow_img_pyramid': [_P, _P, _P, _I, _I, _I, _P],
    'unflow_prepare_triplets': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    'unflow_png_unfilter': [_P, _I, _I, _I],
}

HEADER_PATH = os.path.join(os.path.dirname(PKG), 'include', 'unflow_hip.h')


def header_abi_version(path=HEADER_PATH):
    """``#define UNFLOW_ABI_VERSION n`` of include/unflow_hip.h -- ONE number for the library (built from that header), a C
    user and this binding."""
    import re
    with open(path) as f:
        m = re.search(r'^#define\s+UNFLOW_ABI_VERSION\s+(\d+)\s*$', f.read(), re.M)
    if m is None:
        raise UnflowLibraryError('%s does not define UNFLOW_ABI_VERSION' % path)
    return int(m.group(1))


class UnflowLibraryError(RuntimeError):
    pass


# The ABI this file's SIGNATURES table was written for.  Bump it together with UNFLOW_ABI_VERSION of include/unflow_hip.h whenever an
# entry point changes: load() wants library == header == this number, so a header bump + rebuild with a stale ctypes table is caught
# (reading the number from the header alone only detects a stale .so).
BINDING_ABI = 12


def _abi_version():
    """The header's number when the header travels with the package (the in-tree layout), else the binding's own."""
    try:
        return header_abi_version()
    except OSError:                                        # a copied / installed package without include/: checked against the library in load()
        return BINDING_ABI


ABI_VERSION = _abi_version()
_lib = None


def load():
    """Load the HIP library (once).  Raises UnflowLibraryError if it cannot be used."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UnflowLibraryError(
            'libunflow_hip.so is not built (%s). Run `python -m unopticalflow_amd.build` '
            '(needs hipcc, targets gfx950). There is no CPU fallback.' % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise UnflowLibraryError('cannot load %s: %s' % (LIB_PATH, e))
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise UnflowLibraryError('%s does not export %s (stale build?)' % (LIB_PATH, name))
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    if not (lib.unflow_abi_version() == ABI_VERSION == BINDING_ABI):
        raise UnflowLibraryError('ABI version mismatch: library %d, include/unflow_hip.h %d, ctypes binding (_lib.BINDING_ABI) %d'
                                 % (lib.unflow_abi_version(), ABI_VERSION, BINDING_ABI))
    _lib = lib
    return lib


HOST_LIB_PATH = os.path.join(PKG, 'libunflow_host.so')
_host = None


def load_host():
    """libunflow_host.so: the host-only helpers (PNG unfilter) built WITHOUT any HIP dependency.  DataLoader workers
    are forked after the parent initialised the GPU; loading the HIP fat-binary library there would run its
    registration constructors against forked HIP runtime state, which is unsupported -- this library is plain C++."""
    global _host
    if _host is not None:
        return _host
    if not os.path.exists(HOST_LIB_PATH):
        raise UnflowLibraryError('libunflow_host.so is not built (%s). Run `python -m unopticalflow_amd.build`.' % HOST_LIB_PATH)
    try:
        lib = ctypes.CDLL(HOST_LIB_PATH)
    except OSError as e:
        raise UnflowLibraryError('cannot load %s: %s' % (HOST_LIB_PATH, e))
    lib.unflow_png_unfilter.argtypes = SIGNATURES['unflow_png_unfilter']
    lib.unflow_png_unfilter.restype = ctypes.c_int
    _host = lib
    return lib


def check(status, name):
    if status != 0:
        raise RuntimeError('%s failed with status %d%s' % (
            name, status, ' (invalid argument)' if status == -22 else ' (hipError_t)'))
