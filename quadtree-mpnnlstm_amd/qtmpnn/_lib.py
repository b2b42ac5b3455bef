"""ctypes binding of libqtmpnn_hip.so (the C ABI declared in include/qtmpnn.h).

The product path has no CPU fallback: if the shared library is missing or a call
fails, an exception is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# QT_LIB_PATH: an alternative build of the same library (A/B timing of kernel variants on one box; diagnostics only)
LIB_PATH = os.environ.get('QT_LIB_PATH') or os.path.join(_HERE, 'libqtmpnn_hip.so')

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

# name -> argument types (return type is int unless listed in _RESTYPE)
_SIGNATURES = {
    'qt_abi_version': [],
    'qt_quadtree_stage1': [_P, _I, _I, _P, _I, _P, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P],
    'qt_quadtree_stage3': [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _F, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P],
    'qt_scan_i32': [_P, _P, _L, _P, _P],
    'qt_node_features': [_P, _I, _P, _I, _I, _F, _P, _P, _P],
    'qt_edges_blocks': [_I],
    'qt_scan_top': [_P, _I, _P],
    'qt_edges_count': [_P, _P, _I, _P, _I, _I, _P, _P, _P, _I, _P, _I, _P],
    'qt_edges_fill': [_P, _P, _P, _P, _I, _P, _I, _I, _F, _P, _P, _P, _P, _P],
    'qt_edges_norm': [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    'qt_tail_cap': [],
    'qt_head_dgrad': [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P],
    'qt_edges_norm_tiles': [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    'qt_cheb_tile_sync_words': [_I],
    'qt_cheb_tile_xbuf_words': [_I, _I],
    'qt_tile_cap': [_I],
    'qt_cheb_tile_fwd': [_P] * 15 + [_I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P],
    'qt_cheb_tile_bwd': [_P] * 15 + [_I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P],
    'qt_gather': [_P, _I, _P, _P, _L, _P, _P],
    'qt_pool': [_P, _I, _L, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P],
    'qt_sse_rollout': [_I, _P, _P, _P, _P, _P, _P, _P, ctypes.c_int64, ctypes.c_int64, _I, _I, _I, _P, _P],
    'qt_sse_rollout_bwd': [_I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P],
    'qt_remesh': [_P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P],
    'qt_remesh_clip_rows': [],
    'qt_pool_clip': [_P, _I, _L, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _P],
    'qt_remesh_clip': [_P, _P, _P, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _I, _P, _I, _P],
    'qt_sse': [_P, _I, _P, _P, _L, _I, _I, _I, _P, _P],
    'qt_spmm': [_P, _P, _P, _I, _P, _I, _P, _F, _P, _F, _P, _F, _P, _P],
    'qt_spmm1': [_P, _P, _P, _P, _I, _P, _P, _I, _F, _P, _I, _F, _P, _I, _F, _P, _I, _I, _I, _P, _I, _P, _P],
    'qt_dense2': [_P, _I, _P, _P, _I, _P, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _P, _P],
    'qt_spmm2': [_P, _P, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _F, _F, _F, _P, _P],
    'qt_cheb_clip_rows': [],
    'qt_cheb_clip_fwd': [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P],
    'qt_cheb_clip_bwd': [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _P, _I, _I, _P],
    'qt_dense': [_P, _P, _I, _I, _P, _P, _I, _P, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P],
    'qt_wgrad_blocks': [_I],
    'qt_wgrad': [_P, _I, _P, _P, _I, _P, _I, _I, _I, _P, _I, _P, _I, _I, _P, _I, _P, _I, _P],
    'qt_colsum': [_P, _I, _L, _P, _P],
    'qt_lstm_fwd': [_P, _P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P],
    'qt_lstm_bwd_blocks': [_I, _I],
    'qt_lstm_bwd': [_P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _P, _I, _P, _I, _P, _P, _P, _I, _P],
    'qt_sse_bwd': [_P, _I, _P, _P, _P, _I, _P, _I, _P, _P],
    'qt_wgrad_group_blocks': [_I, _P],
    'qt_wgrad_group': [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P],
    'qt_dense_lstm': [_P, _I, _P, _P, _I, _P, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    'qt_decoder_input': [_P, _I, _P, _I, _P, _P, _P],
    'qt_concat': [_P, _P, _P, _I, _I, _P, _P, _P],
    'qt_act_bwd': [_P, _P, _P, _I, _P, _I, _I, _P, _I, _P, _P, _P, _P],
    'qt_attn_blocks': [_I, _I],
    'qt_attn_edge_attrs': [_P, _P, _P, _I, _P, _P, _P, _P],
    'qt_attn_fwd': [_P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P, _F, ctypes.c_uint32, _P, _P, _P, _I, _I, _L, _L, _L, _P],
    'qt_attn_bwd': [_P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _P, _F, ctypes.c_uint32, _P, _P, _I, _P, _P, _I, _P, _P, _I, _P, _P, _I,
                    _I, _I, _L, _L, _L, _L, _P],
    'qt_lstm_dgrad_blocks': [_I],
    'qt_lstm_bwd_dgrad': [_P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _P, _I, _P, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P, _I, _P, _P],
    'qt_split_bf16': [_P, _L, _P, _P, _P],
    'qt_flat_adam': [_P, _P, _P, _P, _I, _P, _P, _F, _F, _F, _F, _F, _P, _P],
    'qt_proj_group': [_P, _I, _L, _I, _I, _P, _P, _P, _L, _I, _I, _I, _P, _I, _L, _I, _I, _P, _P],
    'qt_wgrad_groups': [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _L, _L, _I, _P, _P],
    'qt_dense_sb': [_P, _I, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P],
    'qt_num_cus': [],
    'qt_proj_bwd_blocks': [_I],
    'qt_proj_bwd': [_P, _L, _L, _P, _L, _I, _P, _L, _I, _P, _L, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P],
    'qt_lstm_fused_blocks': [],
    'qt_lstm_bwd_fused': [_P, _I, _P, _I, _P, _I, _P, _P, _I, _P, _P, _I, _P, _I, _P, _P, _I, _P, _I, _I, _I, _P, _P,
                          _P, _I, _P, _P, _I, _P, _I, _I, _I, _P, _I, _P, _I, _P],
    'qt_compose_step_fwd': [_P] * 4 + [_I] * 5 + [_P] * 3,
    'qt_compose_step_bwd': [_P] * 4 + [_I] * 5 + [_P] * 7,
    'qt_compose2_fwd': [_P] * 8 + [_I] * 6 + [_P, _P, _P, _P, _P],
    'qt_compose2_bwd': [_P] * 8 + [_I] * 6 + [_P] * 11,
    'qt_head_fwd': [_P, _I, _P, _P, _I, _P, _I, _I, _P, _P, _P],
    'qt_head_bwd': [_P, _P, _P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _I, _P],
}
_PLAIN = {'qt_proj_bwd_blocks', 'qt_abi_version', 'qt_cheb_clip_rows', 'qt_cheb_tile_sync_words', 'qt_cheb_tile_xbuf_words', 'qt_tile_cap', 'qt_remesh_clip_rows', 'qt_tail_cap', 'qt_num_cus', 'qt_lstm_fused_blocks', 'qt_wgrad_blocks', 'qt_lstm_bwd_blocks', 'qt_lstm_dgrad_blocks', 'qt_attn_blocks'}  # return a value, not an error code

_lib = None


def load():
    """Load the shared library once; raises if it has not been built (run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} is missing: build it with `make -C quadtree-mpnnlstm_amd/csrc` '
                               '(or __graft_entry__.build()); there is no CPU fallback')
        lib = ctypes.CDLL(LIB_PATH)
        lib.qt_last_error.restype = ctypes.c_char_p
        lib.qt_last_error.argtypes = []
        for name, args in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = ctypes.c_int
        _lib = lib
    return _lib


def exported_names():
    return ['qt_last_error'] + list(_SIGNATURES)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream():
    """Handle of torch's current stream on the current device (the raw getter when this torch has it: the Stream object of
    torch.cuda.current_stream() costs ~15 us to build, and an eager training step asks ~200 times)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def on_device(get):
    """Decorator for the entry points a caller reaches first (trainer methods, Seq2Seq.forward, image_to_graph ...): the library
    launches on the CURRENT device's current stream, so a model that lives on cuda:1 while the process's current device is cuda:0
    must switch for the duration of the call, like every torch op does per operand.  get(*args, **kw) -> device | tensor | None."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*a, **k):
            d = get(*a, **k)
            d = getattr(d, 'device', d)
            if d is not None and not isinstance(d, torch.device):
                d = torch.device(d)
            if d is None or d.type != 'cuda' or d.index is None or d.index == torch.cuda.current_device():
                return fn(*a, **k)
            with torch.cuda.device(d):
                return fn(*a, **k)
        return wrapper
    return deco


def call(name, *args):
    """Call an entry point on the current torch stream; raises RuntimeError on a non-zero code."""
    lib = load()
    rc = getattr(lib, name)(*args, stream())
    if rc != 0:
        raise RuntimeError(f'{name} failed ({rc}): {lib.qt_last_error().decode()}')


def value(name, *args):
    return getattr(load(), name)(*args)


def require_cuda(t, what='tensor'):
    if not t.is_cuda:
        raise RuntimeError(f'{what} must live on the GPU: the qtmpnn path has no CPU implementation '
                           '(use oracle/ only as a test checker)')
