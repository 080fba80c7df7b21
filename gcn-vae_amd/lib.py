"""ctypes binding of libgcnvae_hip.so (the C ABI declared in include/gcnvae.h).

There is NO fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libgcnvae_hip.so')

_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

# name -> (restype, argtypes); mirrors include/gcnvae.h one to one (checked by tests/test_abi.py)
SIGNATURES = {
    'gv_version': (_I, []),
    'gv_last_error_string': (ctypes.c_char_p, []),
    'gv_index_caps': (None, [_L, _I, _I, _P, _P, _P]),
    'gv_index_workspace_bytes': (_L, [_L, _I]),
    'gv_build_csr': (_I, [_P, _L, _I, _I, _P, _P, _P, _I, _P, _I, _P, _L, _P]),
    'gv_build_csr_batch_workspace_bytes': (_L, [_P, _I]),
    'gv_build_csr_batch': (_I, [_P, _I, _P, _L, _P]),
    'gv_triplet_lists': (_I, [_P, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'gv_widen2_i32': (_I, [_P, _P, _L, _P, _P, _L, _P]),
    'gv_graph_index_build': (_I, [_P, _P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _I, _P, _L, _P]),
    'gv_relation_index_build': (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _L, _P]),
    'gv_triplet_index_build': (_I, [_P, _L, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I,
                                    _P, _L, _P]),
    'gv_perm_sample': (_I, [_L, _L, ctypes.c_uint64, ctypes.c_uint64, _P, _P, ctypes.c_uint32, _P, _P]),
    'gv_neighborhood_sample_workspace_bytes': (_L, [_I, _L]),
    'gv_neighborhood_sample': (_I, [_P, _P, _P, _P, _I, _L, _I, ctypes.c_uint64, ctypes.c_uint64, _P, ctypes.c_uint32, _P, _P, _L, _P]),
    'gv_relabel_workspace_bytes': (_L, [_I]),
    'gv_relabel_pairs': (_I, [_P, _P, _L, _I, _P, _I, _P, _P, _P, _P, _L, _P]),
    'gv_negative_sampling': (_I, [_P, _P, _P, _L, _I, _P, _P, _P, ctypes.c_uint64, ctypes.c_uint64, _P, ctypes.c_uint32, _P, _P, _P]),
    'gv_graph_from_triplets_workspace_bytes': (_L, [_L, _I, _I]),
    'gv_graph_from_triplets': (_I, [_P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P, _P, _L, _P]),
    'gv_segment_items_count': (_I, [_P, _I, _I, _P, _P, _P, _P]),
    'gv_segment_items_fill': (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _P]),
    'gv_rgcn_bdd_aggregate': (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I,
                                   _P, _I, _I, _P, _F, _P, _I, _P, _P]),
    'gv_rgcn_bdd_phase_plan': (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    'gv_rgcn_bdd_pack_weight_phase': (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    'gv_rgcn_bdd_aggregate_phases': (_I, [_P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I,
                                          _P, _I, _I, _P, _F, _P, _I, _P, _P]),
    'gv_rgcn_bdd_lds_plan': (_I, [_I, _I, _I, _I, _I, _P]),
    'gv_rgcn_bdd_pack_weight_lds': (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    'gv_rgcn_bdd_aggregate_lds': (_I, [_P, _I, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _I, _I,
                                       _P, _F, _P, _I, _P, _I, _P]),
    'gv_rgcn_bdd_pack_supported': (_I, [_I, _I, _I, _I]),
    'gv_rgcn_bdd_pack_weight': (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    'gv_rgcn_bdd_pack_weight_pair': (_I, [_P, _I, _I, _I, _I, _P, _P, _P]),
    'gv_rgcn_bdd_fixup': (_I, [_P, _I, _P, _I, _P, _I, _I, _P, _F, _P, _I, _P]),
    'gv_rgcn_bdd_grad_weight': (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _P]),
    'gv_rgcn_epilogue_fwd': (_I, [_P, _P, _I, _P, _F, _P, _L, _I, _P]),
    'gv_rgcn_epilogue_bwd': (_I, [_P, _P, _I, _P, _F, _P, _L, _I, _P, _P]),
    'gv_colsum_finish': (_I, [_P, _I, _I, _P, _I, _P]),
    'gv_gemm_workspace_bytes': (_L, [_I, _I, _I, _I]),
    'gv_gemm_f32': (_I, [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _P, _L, _P]),
    'gv_gemm_f32_live_rows': (_I, [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _P, _L, _P, _P]),
    'gv_gemm_f32_sparse': (_I, [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _P, _L, _P, _P, _P, _I, _P]),
    'gv_gemm_bf16': (_I, [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _P, _P, _L, _P]),
    'gv_gemm_bf16_nt_workspace_bytes': (_L, [_I, _I, _I]),
    'gv_gemm_bf16_gradw_fits': (_I, [_I, _I, _I, _I]),
    'gv_gemm_bf16_gradw_workspace_bytes': (_L, [_I, _I, _I]),
    'gv_gemm_bf16_gradw': (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _L, _P]),
    'gv_gemm_bf16_gradw_tiles': (_I, [_P, _L, _P, _L, _I, _I, _I, _P, _I, _P, _I, _P, _L, _P]),
    'gv_gemm_bf16_nt': (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _I, _P, _I, _P, _I, _I, _P, _L, _P]),
    'gv_cast_bf16': (_I, [_P, _I, _I, _I, _P, _I, _P, _I, _P]),
    'gv_rowsum_bf16_workspace_floats': (_L, [_I, _I]),
    'gv_iaf_update_fwd_bf16': (_I, [_P, _P, _I, _P, _P, _P, _P, _I, _P, _I, _L, _I, _P]),
    'gv_iaf_update_fwd_bf16_tiles': (_I, [_P, _P, _I, _P, _P, _P, _P, _I, _P, _L, _L, _I, _P]),
    'gv_iaf_update_bwd_bf16': (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _L, _I, _P]),
    'gv_gather3_i32': (_I, [_P, _L, _P, _P, _P, _P, _P, _P, _P]),
    'gv_iaf_update_bwd_row0_workspace_floats': (_L, [_I]),
    'gv_iaf_update_bwd_row0': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    'gv_iaf_update_bwd_bf16_ex': (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _L, _I, _P]),
    'gv_rowsum_bf16': (_I, [_P, _I, _I, _I, _P, _I, _P, _P]),
    'gv_rowsum_bf16_segments': (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _P, _P]),
    'gv_made_pack_weight_elems': (_L, [_I, _I]),
    'gv_made_pack_weight': (_I, [_P, _I, _I, _I, _P, _P, _P]),
    'gv_made_pack_weight_multi': (_I, [_I, _P, _P, _P, _P, _P, _P, _P]),
    'gv_made_pack_weight_multi_iaf': (_I, [_I, _P, _P, _P, _P, _P, _P, _P]),
    'gv_made_pack_weight_iaf': (_I, [_P, _I, _I, _I, _P, _P]),
    'gv_made_chain_fits': (_I, [_I, _P, _P, _I]),
    'gv_made_chain': (_I, [_P, _I, _I, _I, _P, _P]),
    'gv_made_chain_iafb': (_I, [_P, _I, _I, _P, _P]),
    'gv_made_chain_fwd': (_I, [_P, _I, _I, _I, _P, _I, _P, _P]),
    'gv_made_chain_fwd_row0': (_I, [_P, _I, _I, _P, _I, _P, _P]),
    'gv_made_chain_debug_stamps': (_I, [_P]),
    'gv_made_pack_weight_f32_elems': (_L, [_I, _I]),
    'gv_made_pack_weight_f32_multi': (_I, [_I, _P, _P, _P, _P, _P, _P, _P]),
    'gv_made_chain_f32_fits': (_I, [_I, _P, _P]),
    'gv_made_chain_f32_plan': (_I, [_I, _P, _P, _P, _P, _P, _P, _P]),
    'gv_made_chain_f32': (_I, [_P, _I, _I, _I, _P, _P, _P, _P]),
    'gv_made_passes_f32': (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P]),
    'gv_made_gradw_f32_plan_words': (_L, [_I, _I]),
    'gv_made_gradw_f32_plan': (_I, [_P, _I, _I, _I, _P, _P]),
    'gv_made_gradw_f32_workspace_floats': (_L, [_I, _I, _L]),
    'gv_made_gradw_f32': (_I, [_P, _I, _P, _I, _I, _I, _L, _P, _P, _I, _P, _P, _P, _P, _I, _I, _P, _I, _P, _L, _P]),
    'gv_made_gradw_f32_multi_workspace_floats': (_L, [_I, _P]),
    'gv_made_gradw_f32_multi': (_I, [_I, _P, _P, _L, _P]),
    'gv_made_row_fwd': (_I, [_P, _I, _P, _P]),
    'gv_made_row_bwd': (_I, [_P, _I, _P, _P, _P]),
    'gv_rel_rows_gemm': (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _I, _P, _P]),
    'gv_rel_gradw_gemm': (_I, [_P, _I, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P, _P]),
    'gv_rank_scores': (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _P]),
    'gv_colsum': (_I, [_P, _P, _L, _I, _I, _P, _P, _I, _P]),
    'gv_gather_rows': (_I, [_P, _P, _P, _L, _I, _P]),
    'gv_gather_rows_rng_tick': (_I, [_P, _P, _P, _L, _I, _P, _P]),
    'gv_rng_fill': (_I, [_P, _I, _P, _P, _P, _P, _P, _P]),
    'gv_rng_tick': (_I, [_P, _P]),
    'gv_scatter_add_rows': (_I, [_P, _P, _P, _L, _I, _P]),
    'gv_reparam_fwd': (_I, [_P, _P, _P, _P, _P, _L, _I, _P]),
    'gv_reparam_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    'gv_distmult_bce_fwd': (_I, [_P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    'gv_bce_grad': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P]),
    'gv_mean_sq': (_I, [_P, _L, _F, _P, _P, _I, _P]),
    'gv_mean_sq2': (_I, [_P, _L, _F, _P, _L, _F, _P, _P, _P, _L, _P]),
    'gv_axpby': (_I, [_L, _P, _F, _P, _F, _P, _P]),
    'gv_mul': (_I, [_L, _P, _P, _P, _P]),
    'gv_mul_multi': (_I, [_I, _P, _P, _P, _P, _P]),
    'gv_kl_workspace_bytes': (_L, [_L, _I, _I]),
    'gv_kl_fwd': (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P]),
    'gv_kl_bwd': (_I, [_P, _P, _I, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _I, _P, _I, _L, _I, _I, _P, _P]),
    'gv_reparam_kl_fwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _P]),
    'gv_reparam_kl_bwd': (_I, [_P, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _I, _P, _L, _I, _I, _P]),
    'gv_loss_combine': (_I, [_P, _L, _P, _L, _L, _P, _L, _I, _I, _P, _I, _I, _F, _F, _F, _P, _P, _P, _P]),
    'gv_lincomb4': (_I, [_P, _F, _P, _F, _P, _F, _P, _F, _P, _P]),
    'gv_mmd_fwd': (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P]),
    'gv_mmd_bwd': (_I, [_P, _P, _P, _I, _I, _I, _P, _F, _P, _P, _P]),
    'gv_prior_sample_fwd': (_I, [_P, _P, _P, _I, _I, _I, _P]),
    'gv_prior_sample_bwd': (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    'gv_iaf_update_fwd': (_I, [_P, _P, _I, _P, _P, _P, _L, _I, _P]),
    'gv_iaf_update_bwd': (_I, [_P, _P, _I, _P, _P, _P, _P, _P, _P, _L, _I, _P]),
    'gv_iaf_update_bwd_acc': (_I, [_P, _P, _I, _P, _P, _P, _P, _I, _P, _P, _L, _I, _P]),
    'gv_rowsum': (_I, [_P, _I, _I, _I, _P, _L, _P]),
    'gv_reverse_cols': (_I, [_P, _P, _L, _I, _P]),
    'gv_mean_rows_multi': (_I, [_I, _P, _L, _P, _P, _P, _P]),
    'gv_mean_rows_bwd': (_I, [_P, _L, _L, _P, _P, _P]),
    'gv_adam_step': (_I, [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _F, _P, _P]),
    'gv_clip_adam_step': (_I, [_P, _P, _P, _P, _L, _P, _P, _F, _F, _F, _F, _F, _P, _I, _P]),
}

_lib = None
_fns = {}        # name -> bound foreign function (a dict lookup per launch instead of CDLL.__getattr__)


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: the HIP extension is not built. Run `python gcn-vae_amd/_build.py` '
                '(or __graft_entry__.build()). There is no CPU fallback for this path.')
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
            _fns[name] = fn
        _lib = lib
    return _lib


def last_error():
    return load().gv_last_error_string().decode('utf-8', 'replace')


class KernelTimer:
    """Brackets selected C-ABI launches with HIP events on torch's current stream (the stream the
    kernels are enqueued on).  Used by bench.py for the roofline figure; never active otherwise."""

    def __init__(self):
        self.events = {}

    def bracket(self, key, fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.events.setdefault(key, []).append((s, e))

    def results_ms(self):
        torch.cuda.synchronize()
        return {k: [s.elapsed_time(e) for s, e in v] for k, v in self.events.items()}


TIMER = None      # set to a KernelTimer to time tagged launches


def call(name, *args, tag=None):
    """Invoke an int-returning entry point; non-zero status raises with the library's message."""
    fn = _fns.get(name)
    if fn is None:
        fn = getattr(load(), name)
    if TIMER is not None and tag is not None:
        box = []
        TIMER.bracket(tag, lambda: box.append(fn(*args)))
        rc = box[0]
    else:
        rc = fn(*args)
    if rc != 0:
        raise RuntimeError(f'{name} failed with status {rc}: {last_error()}')


def stream():
    """The caller's current stream as a raw handle (an int: ctypes converts it for the void* parameter)."""
    return torch.cuda.current_stream().cuda_stream or None


def ptr(t):
    """Raw device address of a tensor (int), None -> NULL."""
    return None if t is None else (t.data_ptr() or None)
