"""ctypes binding of libofx.so (the C ABI declared in include/ofx.h).

There is NO fallback: if the library is missing or a call returns a non-zero
status the product raises.  Tensors cross the boundary as raw device pointers
(`tensor.data_ptr()`), sizes as int64, the stream as `hipStream_t`.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('OFX_LIB', os.path.join(_HERE, 'libofx.so'))   # OFX_LIB: experiment builds

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_l = ctypes.c_int64
c_f = ctypes.c_float
c_sz = ctypes.c_size_t


class OfxTree(ctypes.Structure):
    _fields_ = [('depth', c_i), ('full_depth', c_i), ('batch_size', c_i),
                ('child_all', c_p), ('key_all', c_p), ('leafrank_all', c_p),
                ('nnum_host', c_p), ('nnum_nempty_host', c_p)]


# name -> (restype, argtypes, returns_status)
_SIGS = {
    'ofx_version': (c_i, [], False),
    'ofx_status_string': (ctypes.c_char_p, [c_i], False),
    'ofx_device_check': (c_i, [], False),
    'ofx_build_hash': (ctypes.c_char_p, [], False),
    'ofx_build_ablation': (c_i, [], False),
    'ofx_probe_mfma_sustained': (c_i, [c_i, c_i, c_p, c_p, c_p], True),
    'ofx_scan_ws_bytes': (c_sz, [c_l], False),
    'ofx_scan_i32': (c_i, [c_p, c_p, c_l, c_p, c_p], True),
    'ofx_octree_full_layer': (c_i, [c_i, c_i, c_p, c_p, c_p], True),
    'ofx_octree_split': (c_i, [c_p, c_l, c_p, c_p, c_p, c_p], True),
    'ofx_octree_grow': (c_i, [c_p, c_p, c_l, c_p, c_p, c_p], True),
    'ofx_split_small_label0': (c_i, [c_p, c_i, c_i, c_p, c_p], True),
    'ofx_split_small_label1': (c_i, [c_p, c_i, c_i, c_p, c_p, c_p], True),
    'ofx_split_large_label0': (c_i, [c_p, c_l, c_p, c_p], True),
    'ofx_split_large_label1': (c_i, [c_p, c_l, c_p, c_p, c_p], True),
    'ofx_octree2voxel_cf': (c_i, [c_p, c_l, c_i, c_i, c_i, c_p, c_p], True),
    'ofx_voxel2octree_cf': (c_i, [c_p, c_i, c_i, c_i, c_p, c_l, c_p], True),
    'ofx_points_sort_ws_bytes': (c_sz, [c_l], False),
    'ofx_points_keys': (c_i, [c_p, c_l, c_p, c_i, c_l, c_i, c_p, c_p, c_p], True),
    'ofx_points_sort': (c_i, [c_p, c_p, c_l, c_p, c_p, c_p, c_sz, c_p], True),
    'ofx_octree_label_from_points': (c_i, [c_p, c_l, c_p, c_l, c_i, c_i, c_p, c_p], True),
    'ofx_octree_point_features': (c_i, [c_p, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_l, c_i, c_p, c_p, c_p, c_p], True),
    'ofx_tree_leafrank_ws_bytes': (c_sz, [c_l], False),
    'ofx_tree_leafrank': (c_i, [c_p, c_p, c_i, c_p, c_p, c_p], True),
    'ofx_graph_nodes': (c_i, [ctypes.POINTER(OfxTree), c_i, c_p, c_p, c_p, c_p, c_p], True),
    'ofx_graph_count': (c_i, [ctypes.POINTER(OfxTree), c_i, c_p, c_p], True),
    'ofx_graph_reverse_count': (c_i, [c_p, c_p, c_l, c_p, c_p], True),
    'ofx_graph_reverse_fill': (c_i, [c_p, c_p, c_l, c_p, c_p, c_p, c_p, c_p], True),
    'ofx_graph_primary_w': (c_i, [c_p, c_p, c_p, c_l, c_p, c_p], True),
    'ofx_graph_multi_flag_w': (c_i, [c_p, c_p, c_l, c_p, c_p], True),
    'ofx_graph_primary_ext_w': (c_i, [c_p, c_p, c_p, c_l, c_p, c_p, c_p, c_p], True),
    'ofx_graphconv_bwd_data': (c_i, [c_p, c_l, c_i, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_p, c_p, c_l, c_i, c_p,
                                     c_l, c_p, c_sz, c_p], True),
    'ofx_graphconv_bwd_weight': (c_i, [c_p, c_l, c_i, c_l, c_p, c_p, c_p, c_p, c_p, c_l, c_p, c_p, c_l, c_i, c_p, c_l,
                                       c_i, c_p, c_l, c_p, c_sz, c_p], True),
    'ofx_gn_backward': (c_i, [c_p, c_l, c_p, c_l, c_l, c_i, c_p, c_i, c_p, c_i, c_f, c_p, c_p, c_p, c_p, c_i, c_p, c_p,
                              c_p, c_l, c_p, c_p, c_p], True),
    'ofx_gemm_tn_f32': (c_i, [c_p, c_l, c_p, c_l, c_l, c_l, c_l, c_p, c_p, c_sz, c_p], True),
    'ofx_table_reverse_count': (c_i, [c_p, c_l, c_i, c_l, c_p, c_p], True),
    'ofx_table_reverse_fill': (c_i, [c_p, c_l, c_i, c_l, c_p, c_p, c_p, c_p, c_p], True),
    'ofx_seg_primary_w': (c_i, [c_p, c_p, c_p, c_l, c_p, c_p], True),
    'ofx_seg_multi_flag_w': (c_i, [c_p, c_p, c_l, c_p, c_p], True),
    'ofx_seg_primary_ext_w': (c_i, [c_p, c_p, c_p, c_l, c_l, c_p, c_p, c_p, c_p], True),
    'ofx_gridconv_bwd_data': (c_i, [c_p, c_l, c_i, c_l, c_l, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_p, c_p, c_i, c_p, c_l,
                                    c_p, c_sz, c_p], True),
    'ofx_gridconv_bwd_weight': (c_i, [c_p, c_l, c_i, c_l, c_l, c_p, c_p, c_p, c_p, c_l, c_i, c_p, c_p, c_sz, c_p], True),
    'ofx_attention_bwd': (c_i, [c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_i, c_p, c_p, c_l, c_p], True),
    'ofx_adamw_step': (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_f, c_f, c_f, c_i, c_p], True),
    'ofx_ema_update': (c_i, [c_p, c_p, c_l, c_f, c_p], True),
    'ofx_gn_fused_rows': (c_i, [c_p, c_l, c_i, c_i, c_i, c_i, c_f, c_f, c_p, c_p, c_i, c_p, c_l, c_p], True),
    'ofx_set_gn_rows16': (c_i, [c_i], True),
    'ofx_mpu_eval': (c_i, [ctypes.POINTER(OfxTree), c_i, c_i, c_p, c_l, c_p, c_p, c_p, c_p], True),
    'ofx_mpu_eval_grid': (c_i, [ctypes.POINTER(OfxTree), c_i, c_i, c_p, c_i, c_f, c_f, c_i, c_l, c_l, c_p, c_p, c_p], True),
    'ofx_mpu_eval_grad': (c_i, [ctypes.POINTER(OfxTree), c_i, c_i, c_p, c_l, c_p, c_p, c_p, c_p, c_p], True),
    'ofx_mpu_backward': (c_i, [ctypes.POINTER(OfxTree), c_i, c_i, c_p, c_l, c_p, c_p, c_p, c_p, c_p], True),
    'ofx_octree_ce': (c_i, [c_p, c_l, c_p, c_l, c_f, c_p, c_p, c_l, c_p], True),
    'ofx_sdf_reg_loss': (c_i, [c_p, c_p, c_p, c_p, c_l, c_f, c_f, c_p, c_p, c_p, c_p], True),
    'ofx_kl_sample_fwd': (c_i, [c_p, c_l, c_p, c_l, c_i, c_p, c_p, c_p], True),
    'ofx_kl_sample_bwd': (c_i, [c_p, c_l, c_p, c_p, c_l, c_i, c_f, c_p, c_l, c_p], True),
    'ofx_graph_fill': (c_i, [ctypes.POINTER(OfxTree), c_i, c_p, c_p, c_p], True),
    'ofx_graph_expand': (c_i, [c_p, c_l, c_p, c_p, c_p, c_p, c_p], True),
    'ofx_graph_type_frac': (c_i, [c_p, c_p, c_p, c_l, c_i, c_p, c_l, c_p], True),
    'ofx_set_precision': (c_i, [c_i], True),
    'ofx_set_range_words': (c_i, [c_p], True),
    'ofx_get_precision': (c_i, [], False),
    'ofx_gn_apply_rows': (c_i, [], False),
    'ofx_set_attention_split': (c_i, [c_i], True),
    'ofx_gather_gemm_f32': (c_i, [c_p, c_l, c_i, c_i, c_l, c_l, c_p, c_p, c_p, c_l, c_i, c_p, c_p, c_l, c_p, c_l, c_p, c_p, c_sz, c_i, c_p], True),
    'ofx_graphconv_narrow_in_tab': (c_i, [c_p, c_l, c_i, c_l, c_p, c_p, c_p, c_p, c_l, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_l, c_p, c_l,
                                          c_p, c_sz, c_p], True),
    'ofx_graphconv_narrow_in': (c_i, [c_p, c_l, c_i, c_l, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_l, c_p, c_l, c_p, c_sz, c_p], True),
    'ofx_narrow_out_pack': (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_p], True),
    'ofx_narrow_out_type_term': (c_i, [c_p, c_l, c_i, c_l, c_p, c_i, c_i, c_p, c_p, c_p], True),
    'ofx_graphconv_narrow_out': (c_i, [c_p, c_l, c_i, c_l, c_p, c_p, c_p, c_p, c_l, c_p], True),
    'ofx_packed_floats': (c_l, [c_l, c_l], False),
    'ofx_packed_k': (c_l, [c_l], False),
    'ofx_graphconv_packed_k': (c_l, [c_i, c_i], False),
    'ofx_pack_weights': (c_i, [c_p, c_l, c_l, c_l, c_l, c_i, c_i, c_p, c_l, c_p], True),
    'ofx_gemm_f32': (c_i, [c_p, c_l, c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_p, c_l, c_p, c_l, c_p, c_p, c_sz, c_p], True),
    'ofx_gemm_f32_planes': (c_i, [c_p, c_l, c_p, c_l, c_l, c_p, c_l, c_l, c_p, c_p, c_l, c_p, c_l, c_p, c_p, c_sz, c_i, c_p], True),
    'ofx_graphconv_fwd': (c_i, [c_p, c_l, c_i, c_l, c_p, c_p, c_p, c_p, c_p, c_l, c_p, c_p, c_l, c_i, c_p, c_l,
                                c_i, c_p, c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_sz, c_p], True),
    'ofx_planes_split': (c_i, [c_p, c_l, c_l, c_i, c_i, c_i, c_p, c_l, c_p], True),
    'ofx_planes_merge': (c_i, [c_p, c_l, c_l, c_i, c_i, c_p, c_l, c_p], True),
    'ofx_gn_apply_planes': (c_i, [c_p, c_l, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_f, c_p, c_p, c_i, c_i, c_p,
                                  c_l, c_p, c_p, c_p, c_l, c_p, c_p, c_l, c_p], True),
    'ofx_gn_apply_planes_oct': (c_i, [c_p, c_l, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_f, c_p, c_p, c_i, c_i, c_p,
                                      c_l, c_l, c_p, c_p, c_p, c_l, c_i, c_p, c_p, c_l, c_p], True),
    'ofx_set_gn_left_place': (c_i, [c_i], True),
    'ofx_planes_packed_ktiles': (c_l, [c_i, c_i, c_i], False),
    'ofx_planes_packed_bytes': (c_l, [c_i, c_i, c_i, c_i], False),
    'ofx_pack_weights_planes': (c_i, [c_p, c_l, c_l, c_i, c_i, c_i, c_i, c_p, c_p], True),
    'ofx_graphconv_fwd_planes': (c_i, [c_p, c_l, c_i, c_l, c_p, c_p, c_p, c_p, c_l, c_p, c_sz, c_p, c_l, c_i, c_p, c_i,
                                       c_p, c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_p, c_l, c_p, c_sz, c_p, c_sz, c_i, c_i,
                                       c_p], True),
    'ofx_set_gconv_persistent': (c_i, [c_i], True),
    'ofx_set_gconv_xcd_contig': (c_i, [c_i], True),
    'ofx_set_gconv_cus': (c_i, [c_i], True),
    'ofx_set_gemm_bn64': (c_i, [c_i], True),
    'ofx_gemm_planes_packed_bytes': (c_l, [c_i, c_i, c_i], False),
    'ofx_pack_gemm_planes': (c_i, [c_p, c_l, c_l, c_i, c_i, c_i, c_p, c_p], True),
    'ofx_gemm_planes': (c_i, [c_p, c_l, c_l, c_l, c_i, c_p, c_p, c_i, c_p, c_p, c_l, c_i, c_p, c_sz, c_p, c_sz, c_i, c_p], False),
    'ofx_gconv3_plan': (c_i, [c_l, c_i, c_i, c_i, c_i, c_i, c_p, c_l], False),
    'ofx_set_gconv2_variant': (c_i, [c_i], True),
    'ofx_set_gconv2_debug': (c_i, [c_p], True),
    'ofx_set_gconv2_tile': (c_i, [c_i], True),
    'ofx_set_gconv2_stagger': (c_i, [c_i], True),
    'ofx_set_gconv2_prefetch': (c_i, [c_i], True),
    'ofx_graph_multi_flag': (c_i, [c_p, c_l, c_p, c_p], True),
    'ofx_graph_primary_ext': (c_i, [c_p, c_p, c_l, c_p, c_p, c_p, c_p], True),
    'ofx_graph_primary': (c_i, [c_p, c_p, c_l, c_p, c_p], True),
    'ofx_grid_conv_table': (c_i, [c_i, c_i, c_i, c_i, c_p, c_p], True),
    'ofx_conv3d_packed_k': (c_l, [c_i], False),
    'ofx_pack_conv3d': (c_i, [c_p, c_i, c_i, c_p, c_p], True),
    'ofx_gridconv_fwd': (c_i, [c_p, c_l, c_i, c_l, c_l, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_l, c_p, c_p, c_l, c_p,
                               c_l, c_p, c_sz, c_p], True),
    'ofx_attention': (c_i, [c_p, c_l, c_i, c_i, c_i, c_i, c_p, c_l, c_p], True),
    'ofx_gather_mean': (c_i, [c_p, c_l, c_i, c_l, c_p, c_p, c_p, c_p], True),
    'ofx_gn_stats': (c_i, [c_p, c_l, c_l, c_i, c_p, c_i, c_p, c_p], True),
    'ofx_gn_stats_acc': (c_i, [c_p, c_l, c_l, c_i, c_p, c_i, c_p, c_p], True),
    'ofx_gn_finalize': (c_i, [c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_p, c_p, c_p], True),
    'ofx_gn_apply': (c_i, [c_p, c_l, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_i, c_f, c_f, c_p, c_p, c_i, c_p, c_l, c_p], True),
    'ofx_rows_copy': (c_i, [c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_i, c_p], True),
    'ofx_rows_copy_planes': (c_i, [c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_i, c_i, c_p], True),
    'ofx_act': (c_i, [c_p, c_p, c_l, c_i, c_p], True),
    'ofx_timestep_embedding': (c_i, [c_p, c_i, c_i, c_f, c_p, c_p], True),
    'ofx_learned_sinusoid': (c_i, [c_p, c_p, c_i, c_i, c_p, c_p], True),
    'ofx_linear_small': (c_i, [c_p, c_l, c_i, c_i, c_p, c_l, c_i, c_p, c_p, c_l, c_i, c_i, c_p, c_l, c_p], True),
    'ofx_ddim_eps_update': (c_i, [c_p, c_p, c_p, c_p, c_l, c_p], True),
    'ofx_ddim_x0_update': (c_i, [c_p, c_p, c_p, c_p, c_l, c_p], True),
}

EXPORTS = sorted(_SIGS)


class OfxError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libofx.so (once).  Raises OfxError if it is missing -- no CPU fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OfxError('libofx.so not built: run `python -m octfusion_amd.build` '
                           '(the product has no CPU / eager fallback)')
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args, _) in _SIGS.items():
            fn = getattr(L, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _check_fresh(L)
        _lib = L
    return _lib


def _check_fresh(L):
    """Refuse a library that was not built from the sources next to it (libofx.so is git-ignored and mtime says
    nothing after a checkout / snapshot): the build embeds the sha256 of every source, header and flag."""
    if 'OFX_LIB' in os.environ:            # an explicitly chosen experiment build
        return
    from . import build
    try:
        want = {build.source_hash(), build.source_hash(build.FLAGS + ['-DOFX_ABLATION'])}
    except OSError:
        return                             # sources not shipped: nothing to compare with
    have = L.ofx_build_hash().decode()
    if have not in want:
        raise OfxError('libofx.so is stale: built from sources %s, the tree is %s -- run `python -m octfusion_amd.build`'
                       % (have, sorted(want)[0] if len(want) == 1 else build.source_hash()))


# bench.py sets PROFILE to a list to time EVERY entry-point call with HIP events on the launching stream (eager runs
# only): entries are (name, start_event, end_event, meta); `meta` = (class, algorithmic flops, algorithmic bytes, tag)
# set by the ops wrapper right before the call (ops._meta) or None.
PROFILE = None
META = None


def call(name, *args):
    """Invoke a status-returning entry point; raise OfxError on failure."""
    global META
    prof = PROFILE
    if prof is not None and _SIGS[name][2]:
        meta, META = META, None
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib(), name)(*args)
        e1.record()
        prof.append((name, e0, e1, meta))
    else:
        rc = getattr(lib(), name)(*args)
    if _SIGS[name][2] and rc != 0:
        msg = lib().ofx_status_string(rc).decode()
        raise OfxError('%s failed: %s (%d)' % (name, msg, rc))
    return rc


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_device():
    if not torch.cuda.is_available():
        raise OfxError('no HIP device: octfusion_amd has no CPU path')
    rc = lib().ofx_device_check()
    if rc != 0:
        raise OfxError('ofx_device_check: %s' % lib().ofx_status_string(rc).decode())
