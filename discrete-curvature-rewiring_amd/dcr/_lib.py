"""ctypes loader for libdcr_hip.so (the C ABI declared in include/dcr.h).

There is no CPU fallback: if the HIP library is missing or no GPU is visible
the product path raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_HERE), 'csrc')
# DCR_LIB: another build of the same library (A/B timing of kernel variants, tools/ab_pass.sh) without touching the default
LIB_PATH = os.environ.get('DCR_LIB') or os.path.join(CSRC, 'libdcr_hip.so')

_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
_f64 = ctypes.c_double
_vp = ctypes.c_void_p
_i32p = ctypes.POINTER(_i32)
_i64p = ctypes.POINTER(_i64)
_f64p = ctypes.POINTER(_f64)

# name -> (restype, argtypes); every symbol include/dcr.h declares
SIGNATURES = {
    'dcr_last_error': (ctypes.c_char_p, []),
    'dcr_device_count': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    'dcr_graph_create': (ctypes.c_int, [ctypes.c_int, _i64, _i64, _i64p, _i64p, ctypes.POINTER(_vp)]),
    'dcr_graph_destroy': (ctypes.c_int, [_vp]),
    'dcr_graph_num_nodes': (ctypes.c_int, [_vp, _i64p]),
    'dcr_graph_num_edges': (ctypes.c_int, [_vp, _i64p]),
    'dcr_graph_add_edge': (ctypes.c_int, [_vp, _i32, _i32]),
    'dcr_graph_remove_edge': (ctypes.c_int, [_vp, _i32, _i32]),
    'dcr_graph_has_edge': (ctypes.c_int, [_vp, _i32, _i32, ctypes.POINTER(ctypes.c_int)]),
    'dcr_graph_degree': (ctypes.c_int, [_vp, _i32, _i32p]),
    'dcr_graph_neighbors': (ctypes.c_int, [_vp, _i32, _i64, _i32p, _i64p]),
    'dcr_graph_edges': (ctypes.c_int, [_vp, _i32p, _i32p]),
    'dcr_graph_export_edge_index': (ctypes.c_int, [_vp, _i64p]),
    'dcr_curvature_pass': (ctypes.c_int, [_vp, ctypes.c_int]),
    'dcr_curvature_pass_incremental': (ctypes.c_int, [_vp, ctypes.c_int]),
    'dcr_curvature_pass_argmin': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _i32p, _i32p, _f64p]),
    'dcr_curvature_read': (ctypes.c_int, [_vp, _f64p, _i32p, _i32p]),
    'dcr_curvature_edge': (ctypes.c_int, [_vp, _i32, _i32, ctypes.c_int, _f64p]),
    'dcr_bfc_ingredients': (ctypes.c_int, [_vp, _i32, _i32, _i64p]),
    'dcr_argext': (ctypes.c_int, [_vp, ctypes.c_int, _i32, _i32, _i32p, _i32p, _f64p]),
    'dcr_improvements': (ctypes.c_int, [_vp, _i32, _i32, ctypes.c_int, ctypes.c_int, _i64p, ctypes.POINTER(_f64p),
                                        ctypes.POINTER(_i32p), ctypes.POINTER(_i32p)]),
    'dcr_improvements_argmax': (ctypes.c_int, [_vp, _i64p]),
    'dcr_candidate_at': (ctypes.c_int, [_vp, _i64, _i32p, _i32p]),
    'dcr_sdrf_tail': (ctypes.c_int, [_vp, _i32, _i32, ctypes.c_int, _f64, _i32p, _f64p]),
    'dcr_sdrf_tail_at': (ctypes.c_int, [_vp, _i64, ctypes.c_int, _f64, _i32p, _i32p, _f64p]),
    'dcr_sdrf_iteration_device_draw': (ctypes.c_int, [_vp, _i32, _i32, ctypes.c_int, _f64, _f64, ctypes.c_int, _f64, ctypes.c_int,
                                                     ctypes.POINTER(ctypes.c_int), _i64p, _i32p, _i32p, _i32p, _i32p, _f64p]),
    'dcr_sdrf_tail_at_pass_argmin': (ctypes.c_int, [_vp, _i64, ctypes.c_int, _f64, ctypes.c_int, ctypes.c_int, _i32p, _i32p, _i32p, _i32p, _f64p]),
    'dcr_profile_reset': (ctypes.c_int, [_vp]),
    'dcr_profile_read': (ctypes.c_int, [_vp, _f64p, _i64p]),
    'dcr_bfc_algorithmic_bytes': (ctypes.c_int, [_vp, _f64p]),
    'dcr_bfc_algorithmic_bytes_one_sided': (ctypes.c_int, [_vp, _f64p]),
    'dcr_pass_engine': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int)]),
    'dcr_host_cdf_from_exp': (ctypes.c_int, [_f64p, _i64, _f64, _f64p, _f64p]),
    'dcr_host_cdf_from_exp_plain': (ctypes.c_int, [_f64p, _i64, _f64, _f64p, _f64p]),
    'dcr_spmm_csr_f32_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, ctypes.c_int, _vp]),
    'dcr_spmm_csr_f32_pair_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, ctypes.c_int, _vp]),
    'dcr_spmm_csr_rows2_f32_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _vp, ctypes.c_int, _vp]),
    'dcr_spmm_csr_rows_f32_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _vp, ctypes.c_int, _vp]),
    'dcr_spmm_csr_f32_pair_split_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, ctypes.c_int, _vp]),
    'dcr_relu_dropout_bits_words': (ctypes.c_int, [_i64, _i64p]),
    'dcr_relu_dropout_fwd_f32_dev': (ctypes.c_int, [_vp, _vp, _vp, _i64, _f64, ctypes.c_uint64, ctypes.c_uint64, _vp]),
    'dcr_relu_dropout_fwd_f32_ctr_dev': (ctypes.c_int, [_vp, _vp, _vp, _i64, _f64, ctypes.c_uint64, ctypes.c_uint64, _vp, _vp]),
    'dcr_relu_dropout_bwd_f32_dev': (ctypes.c_int, [_vp, _vp, _vp, _i64, _f64, _vp]),
    'dcr_act_linear_fwd_f32_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, ctypes.c_int, ctypes.c_int, _f64, ctypes.c_uint64,
                                                   ctypes.c_uint64, _vp, _vp]),
    'dcr_nll_picked_mean_fwd_f32_dev': (ctypes.c_int, [_vp, _i64, _vp, _i64, ctypes.c_int, _vp, _vp, _vp]),
    'dcr_nll_picked_mean_bwd_f32_dev': (ctypes.c_int, [_vp, _i64, ctypes.c_int, _vp, _vp, _vp]),
    'dcr_count_argmax_equal_f32_dev': (ctypes.c_int, [_vp, _i64, _vp, _i64, ctypes.c_int, _vp, _vp]),
    'dcr_adam_step_f32_dev': (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                             _i64p, ctypes.POINTER(ctypes.c_float), _f64, _f64, _f64, _f64, _vp, _vp, _vp]),
    'dcr_head_workspace': (ctypes.c_int, [_i64p]),
    'dcr_head_fwd_f32_dev': (ctypes.c_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, ctypes.c_int, _vp, _vp, _vp, _i64, _vp]),
    'dcr_head_bwd_f32_dev': (ctypes.c_int, [_vp, _i64, _vp, _i64, ctypes.c_int, _vp, _vp, _vp, _vp, _i64, _vp]),
    'dcr_first_layer_fits': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'dcr_first_layer_fwd_f32_dev': (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, _f64, ctypes.c_uint64, ctypes.c_uint64, _vp, _vp]),
    'dcr_first_layer_fwd_workspace': (ctypes.c_int, [_i64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]),
    'dcr_first_layer_fwd_ws_f32_dev': (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, ctypes.c_int, ctypes.c_int,
                                                      ctypes.c_int, _f64, ctypes.c_uint64, ctypes.c_uint64, _vp, _vp, _i64, _vp]),
    'dcr_dropout_words_count': (ctypes.c_int, [_i64, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]),
    'dcr_dropout_words_dev': (ctypes.c_int, [_vp, _i64, ctypes.c_int, _f64, ctypes.c_uint64, ctypes.c_uint64, _vp, _vp]),
    'dcr_first_layer_bwd_workspace': (ctypes.c_int, [_i64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]),
    'dcr_first_layer_bwd_f32_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i64, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, _f64, _vp]),
    'dcr_act_linear_bwd_fused_workspace': (ctypes.c_int, [_i64, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]),
    'dcr_act_linear_bwd_fused_f32_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, ctypes.c_int, ctypes.c_int, _f64, _vp]),
    'dcr_act_linear_bwd_workspace': (ctypes.c_int, [_i64, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]),
    'dcr_act_linear_bwd_colsum_f32_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, ctypes.c_int, ctypes.c_int, _f64, _vp]),
    'dcr_act_linear_bwd_f32_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, ctypes.c_int, ctypes.c_int, _f64, _vp]),
    'dcr_bfc_dense_f32_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    'dcr_bfc_dense_post_delta_f32_dev': (ctypes.c_int, [_vp, _vp, ctypes.c_float, ctypes.c_float, _i64, _vp, ctypes.c_int32, ctypes.c_int32, _vp, _vp, _i64, _i64, _vp]),
    'dcr_atb_f32_workspace': (ctypes.c_int, [_i64, _i64, _i64, _i64p]),
    'dcr_atb_f32_dev': (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _i64, _vp]),
}

_LIB = None


class DcrError(RuntimeError):
    pass


def build():
    """Compile csrc/*.hip for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(['bash', os.path.join(CSRC, 'build.sh')])


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise DcrError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                           f'(there is no CPU fallback for the curvature / SDRF path)')
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; if this library pulled in the system
        # one first, torch.cuda would later report "No HIP GPUs are available".  Let torch load and initialise its
        # runtime first, so libdcr_hip.so binds to the copy that is already in the process.
        try:
            import torch
            if getattr(torch.version, 'hip', None):
                torch.cuda.is_available()
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        msg = lib().dcr_last_error().decode('utf-8', 'replace')
        if rc == -1:
            raise ValueError(msg)
        if rc == -5:
            raise KeyError(msg)
        raise DcrError(f'libdcr_hip error {rc}: {msg}')
