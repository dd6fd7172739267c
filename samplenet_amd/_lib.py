"""ctypes binding of libsamplenet_hip.so (the C ABI declared in include/samplenet_hip.h).

There is NO fallback: if the HIP library is missing or does not export a symbol, importing
samplenet_amd fails loudly -- the product path never routes through a CPU or eager-PyTorch
substitute.
"""
import ctypes
import os

import torch  # noqa: F401  -- first: its bundled HIP runtime (libamdhip64.so.7) must be the one this process uses,
# because the streams and device pointers handed to the library come from torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SAMPLENET_AMD_LIB: another build of the SAME library (same ABI; tools/build_variant.sh -- same-box A/B of a kernel change)
LIB_PATH = os.environ.get("SAMPLENET_AMD_LIB") or os.path.join(_HERE, "lib", "libsamplenet_hip.so")

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float

# name -> argument types (restype is int unless listed in _RESTYPES); mirrors include/samplenet_hip.h (the drop-in
# boundary) and include/samplenet_hip_internal.h (the fused-step entry points behind it)
PROTOTYPES = {
    "sn_abi_version": [],
    "sn_last_error_string": [],
    "sn_workspace_bytes": [ctypes.c_char_p, _i, _i, _i, _i],
    "sn_pairscan_forward": [_i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _f, _vp],
    "sn_pairscan_workspace_bytes": [_i, _i, _i],
    "sn_pairscan_forward_ws": [_i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _f, _vp,
                               ctypes.c_longlong, _vp],
    "sn_soft_bwd_splits": [_i, _i],
    "sn_sigma_grad": [_i, _vp, _vp, _f, _vp, _vp],
    "sn_sigma_forward": [_vp, _f, _vp, _vp],
    "sn_chamfer_forward": [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_chamfer_backward": [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_simplification_loss_forward": [_i, _i, _i, _vp, _vp, _f, _vp, _vp, _vp, _vp],
    "sn_simplification_loss_backward": [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _i, _vp],
    "sn_chamfer_mean_loss_forward": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_chamfer_mean_loss_backward": [_i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_pcrnet_head_forward": [_i, _vp, _vp, _vp, _vp, _vp],
    "sn_pcrnet_head_backward": [_i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_pcrnet_head_rot_forward": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_pcrnet_head_rot_backward": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_sampler_loss_forward": [_i, _vp, _vp, _vp, _f, _f, _f, _vp, _vp],
    "sn_sampler_loss_backward": [_i, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp],
    "sn_pairscan_colmin_splits": [_i, _i, _i],
    "sn_pairscan_forward_partial": [_i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _f, _vp, ctypes.c_longlong, _vp],
    "sn_pairscan_forward_partial_fc": [_i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _f,
                                       _vp, ctypes.c_longlong, _vp],
    "sn_sampler_step_loss_forward": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "sn_sampler_step_loss_forward_direct": [_i, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _vp, _vp, _vp, _i, _vp],
    "sn_sampler_step_loss_backward": [_i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _vp, _vp, _vp,
                                      _vp, _vp, _vp, _vp],
    "sn_pairscan_forward_keys": [_i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _f, _vp, _vp,
                                 _vp, _vp],
    "sn_sampler_step_loss_keys": [_i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _f, _f, _f, _f, _vp, _vp, _vp, _vp,
                                  _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_surface_values_keys": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "sn_surface_gather_upstream": [_i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_conv_stack_set_persist_min_tiles": [_i],
    "sn_conv_stack_set_in3_blocks": [_i],
    "sn_step_tail_bytes": [],
    "sn_step_tail_set_error_words": [_vp, _vp, _vp],
    "sn_prefix_point_minima": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_prefix_simplification_loss_forward": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_prefix_simplification_loss_backward": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_prefix_pack": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sn_chamfer_forward_valid": [_i, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _vp],
    "sn_pcrnet_head_rot_forward_grouped": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_pcrnet_head_rot_backward_grouped": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_chamfer_mean_loss_forward_grouped": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_chamfer_mean_loss_backward_grouped": [_i, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_cyclic_pad_cat": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sn_cyclic_pad_cat_backward": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sn_prefix_scatter_sum": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sn_knn": [_i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _vp],
    "sn_nn_matching": [_i, _i, _i, _vp, _i, _vp, _i, _vp, _vp],
    "sn_qrot_forward": [_i, _i, _vp, _vp, _vp, _vp],
    "sn_qrot_backward": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_group_point": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sn_group_point_grad": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sn_grouping_operation": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sn_grouping_operation_grad": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "sn_soft_weights_forward": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp],
    "sn_soft_weights_backward": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_weighted_gather_forward": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sn_weighted_gather_backward": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_soft_project_backward": [_i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _f, _vp, _i, _vp, _i, _vp, _vp, _vp],
    "sn_soft_project_backward_ordered": [_i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp, _f, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp],
    "sn_soft_weights_backward_ordered": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_weighted_gather_backward_ordered": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_linear_stats_blocks": [_i],
    "sn_layer_forward_bn": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "sn_conv_forward_bn_pool": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp, _vp],
    "sn_layer_backward": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                          _vp, ctypes.c_longlong, _vp],
    "sn_conv_stack_forward_supported": [_i, _i, _i, _vp],
    "sn_conv_stack_acc_elems": [_i],
    "sn_conv_stack_z1_free_supported": [_i, _i, _i, _vp],
    "sn_conv_stack_acc_sum_elems": [_i, ctypes.c_void_p],
    "sn_conv_stack_backward_scratch_floats": [_i, _i, _i, _vp],
    "sn_conv_stack_backward": [_i, _i, _i] + [_vp] * 17,
    "sn_conv_stack_forward_bn": [_i, _i, _i] + [_vp] * 20,
    "sn_layer_backward_in3_stats_floats": [_i, _i, _i],
    "sn_layer_backward_in3": [_i, _i, _i] + [_vp] * 17 + [_vp],
    "sn_linear_forward": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_linear_forward_rows": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_bn_finalize": [_i, _i, ctypes.c_longlong, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "sn_fc_chain_forward_supported": [_i, _i, _i, _i],
    "sn_fc_chain_forward_pool_supported": [_i, _i, _i, _i, _i],
    "sn_fc_chain_forward_pool": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _i, _i] + [_vp] * 14,
    "sn_fc_chain_forward_pool_out_supported": [_i, _i, _i, _i, _i, _i],
    "sn_fc_chain_forward_pool_out": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _i, _i] + [_vp] * 13 +
                                    [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp],
    "sn_fc_chain_forward": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_fc_chain_backward_supported": [_i, _i, _vp, _vp],
    "sn_fc_chain_backward": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_fc_chain_backward_obn": [_i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp] + [_vp] * 16,
    "sn_linear_forward_maxpool_supported": [_i, _i, _i, _i],
    "sn_linear_forward_maxpool": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "sn_linear_forward_maxpool_wide_supported": [_i, _i, _i, _i],
    "sn_linear_forward_maxpool_wide_scratch_bytes": [_i, _i, _i, _i],
    "sn_linear_forward_maxpool_wide": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "sn_pointnet_narrow_forward_supported": [_i, _i, _i, _i, _i],
    "sn_pointnet_narrow_forward": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "sn_pointnet_narrow_backward_supported": [_i, _i, _i, _i, _i],
    "sn_pointnet_narrow_backward": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "sn_skinny_linear_supported": [_i, _i, _i],
    "sn_skinny_linear_scratch_bytes": [_i, _i, _i],
    "sn_skinny_linear": [_i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp],
    "sn_skinny_linear2": [_i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp],
    "sn_skinny_wgrad": [_i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "sn_pool_dgrad_sparse_supported": [_i, _i, _i, _i],
    "sn_pool_dgrad_sparse": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_bn_batch_stats_twopass": [_i, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "sn_bn_eval_coef": [_i, _vp, _vp, _f, _vp, _vp, _vp, _vp],
    "sn_layer_forward_bn_out": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_bn_output_forward": [_i, _i, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_bn_output_backward": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_pool_forward": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_pool_backward": [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_pool_backward_bn": [_i, _i, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_bn_backward_coef": [_i, _i, ctypes.c_longlong, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_linear_dgrad": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_linear_backward": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_conv_backward_partials": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_linear_wgrad_splits": [_i, _i, _i, _i],
    "sn_linear_wgrad": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_approxmatch": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "sn_matchcost": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_matchcost_grad": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_emd_loss": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_emd_loss_fast": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sn_emd_set_sweep2d": [_i],
    "sn_emd_set_segments": [_i],
}
_RESTYPES = {"sn_last_error_string": ctypes.c_char_p, "sn_workspace_bytes": ctypes.c_longlong,
             "sn_pairscan_workspace_bytes": ctypes.c_longlong, "sn_linear_forward_maxpool_wide_scratch_bytes": ctypes.c_longlong,
             "sn_skinny_linear_scratch_bytes": ctypes.c_longlong,
             "sn_layer_backward_in3_stats_floats": ctypes.c_longlong,
             "sn_conv_stack_acc_elems": ctypes.c_longlong,
             "sn_conv_stack_acc_sum_elems": ctypes.c_longlong,
             "sn_conv_stack_backward_scratch_floats": ctypes.c_longlong}


class SampleNetHipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "samplenet_amd: %s not found. Build it with `python samplenet_amd/build.py` (hipcc, gfx950). "
            "There is no CPU / eager fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise ImportError("samplenet_amd: %s does not export %s (stale build?)" % (LIB_PATH, name)) from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, _i)
    return lib


lib = _load()
if lib.sn_abi_version() != 1:
    raise ImportError("samplenet_amd: ABI version mismatch in %s" % LIB_PATH)


def check(rc, what=""):
    if rc != 0:
        msg = lib.sn_last_error_string()
        raise SampleNetHipError("%s failed (code %d): %s" % (what or "samplenet_hip call", rc, (msg or b"").decode()))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


try:  # the current stream's handle without building a torch.cuda.Stream object (~0.3 us instead of ~4 us per lookup; the eager
    from torch._C import _cuda_getCurrentRawStream as _raw_stream  # module surface issues dozens of them per step)
except ImportError:  # pragma: no cover
    _raw_stream = None


def stream_of(t):
    """hipStream_t (as an int) of torch's current stream on tensor t's device."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream
