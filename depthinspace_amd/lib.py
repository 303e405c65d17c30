"""ctypes binding of libdis_hip.so (the C ABI declared in include/dis_hip.h).

The HIP library is the product; there is NO fallback.  If the shared object is missing or a symbol
is absent, importing/using this module raises.  PyTorch is used only as the owner of device memory
and of the current HIP stream.
"""
import ctypes
import os
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DIS_HIP_LIB: another build of the same library, for A/B measurements of build flags)
LIB_PATH = os.environ.get('DIS_HIP_LIB') or os.path.join(_HERE, 'libdis_hip.so')

# signature strings: p = pointer (device or host), i = int, l = long, f = float, d = double; the return type is int
# unless listed in _RET_LONG.  Keep in sync with include/dis_hip.h (tests/test_abi.py checks the symbol set).
SIGS = {
    'dis_abi_version': '',
    'dis_lcn_fwd': 'pppiiiifp',
    'dis_photometric_fwd': 'pppiiiiiifp',
    'dis_photometric_bwd': 'ppppiiiiiifp',
    'dis_photometric_fwd_multi': 'pppiiiiiifp',
    'dis_photometric_bwd_multi': 'ppppiiiiiifp',
    'dis_pattern_warp_fwd': 'pppiiip',
    'dis_pattern_warp_bwd': 'ppppiiip',
    'dis_weighted_mean_fwd': 'pppplp',
    'dis_weighted_mean_bwd': 'pppplp',
    'dis_l1_mean_fwd': 'pppplp',
    'dis_l1_mean_bwd': 'pppplp',
    'dis_sgm_l1_fwd': 'pppfpplp',
    'dis_sgm_l1_bwd': 'pppfppplp',
    'dis_smooth_loss_fwd': 'ppppiiip',
    'dis_smooth_loss_bwd': 'pppppiiip',
    'dis_disp_to_depth_fwd': 'ppflp',
    'dis_disp_to_depth_bwd': 'pppflp',
    'dis_geo_loss_fwd': 'ppppppppppppp' + 'f' + 'ppp' + 'iiip',
    'dis_geo_loss_bwd': 'ppppppppp' + 'f' + 'ppppp' + 'iiip',
    'dis_pack4_nhwc': 'pppppiiip',
    'dis_pack4_nhwc_strided': 'plplplplpiiip',
    'dis_planar_to_nhwc': 'ppiiiip',
    'dis_nhwc_to_planar': 'ppiiiip',
    'dis_resize_bilinear_nhwc_fwd': 'ppiiiiiiip',
    'dis_resize_bilinear_nhwc_bwd': 'ppiiiiiiip',
    'dis_resize_bilinear_planar_fwd': 'ppiiiiiiffip',
    'dis_resize_bilinear_planar_bwd': 'ppiiiiiip',
    'dis_gather_warped_feat_fwd': 'pppiiiiip',
    'dis_gather_warped_feat_bwd': 'pppiiiiip',
    'dis_gather_csr_workspace': 'iiii',
    'dis_gather_csr_build': 'ppiiiip',
    'dis_gather_warped_feat_bwd_csr': 'ppppiiiiip',
    'dis_gather_warped_feat_bwd_csr_gnres': 'ppppppp' + 'iiiiiiip',
    'dis_mf_geometry': 'pppppiipiiiip',
    'dis_mf_geometry_resize': 'ppiiiiiip',
    'dis_conv2d_pack_weights': 'ppiiiiip',
    'dis_conv2d_fwd': 'pppppiiiiiiiiip',
    'dis_conv2d_pack_weights_bf16x3': 'ppiiiip',
    'dis_conv2d_fwd_bf16x3': 'pppppiiiiiiiiip',
    'dis_conv2d_fwd_bf16x3_oihw': 'ppiiiipppiiiiiiiiip',
    'dis_conv2d_pack_bf16x3_size': 'ii',
    'dis_conv2d_dgrad_bf16x3_act': 'ppipiiipiiiiiiip',
    'dis_conv2d_wgrad_bf16x3_act': 'pppipppiiiiiiiiip',
    'dis_set_conv_split': 'i',
    'dis_get_conv_split': '',
    'dis_last_kernel': 'pii',
    'dis_conv2d_fwd_bf16x3_gn': 'ppppfpiiipppiiiiiiiiip',
    'dis_conv2d_gnsums_slots': '',
    'dis_conv2d_dgrad_bf16x3_gnsums': 'ppiiipppiiiiiip',
    'dis_conv2d_dgrad_bf16x3_gnsums_res': 'ppiiippppiiiiiip',
    'dis_conv2d_dgrad_bf16x3_act_gnsums_res': 'pppiiippppiiiiiip',
    'dis_gn_bwd_from_sums': 'pppppippppilifip',
    'dis_conv2d_fwd_split_oihw': 'ppiiiipppiiiiiiiiip',
    'dis_conv2d_fwd_split_gn': 'ppppfpiiipppiiiiiiiiip',
    'dis_conv2d_dgrad_split_gnsums': 'ppiiipppiiiiiip',
    'dis_conv2d_dgrad_split_gnsums_res': 'ppiiippppiiiiiip',
    'dis_conv2d_dgrad_split_act_gnsums_res': 'pppiiippppiiiiiip',
    'dis_conv2d_dgrad_split_act': 'ppipiiipiiiiiiip',
    'dis_conv2d_wgrad_split': 'pppppiiiiiiiiip',
    'dis_conv2d_wgrad_split_act': 'pppipppiiiiiiiiip',
    'dis_conv2d_wgrad_split_gn': 'ppppfppppiiiiiiiiip',
    'dis_conv2d_fwd_f16x2_gnres': 'ppppfpppiiipppiiiiiip',
    'dis_conv2d_dgrad1x1_scaled_gnb': 'pppippppiiiiiip',
    'dis_conv2d_wgrad_k4s2_f16x2_gnb': 'ppppippppiiip',
    'dis_conv2d_fwd_k4s2_f16x2': 'pppppiiiip',
    'dis_conv2d_dgrad_k4s2_f16x2': 'pppiiiip',
    'dis_gn_bwd_coef': 'pppippppilifp',
    'dis_gn_bwd_res_sums': 'pppppiiliip',
    'dis_gn_bwd_apply_coef': 'ppppiliip',
    'dis_conv2d_dgrad_f16x2_gnb': 'pppippiiipippp' + 'iiiip',
    'dis_conv2d_bwd_fused_workspace': 'i',
    'dis_conv2d_bwd_fused_f16x2': 'pppippiiipippp' + 'pppp' + 'f' + 'ppp' + 'iiiiip',
    'dis_conv2d_wgrad_bf16x3_gn': 'ppppfppppiiiiiiiiip',
    'dis_conv2d_fwd_scaled': 'ppppppp' + 'iiiiiiiii' + 'p',
    'dis_conv2d_wgrad_scaled': 'pppppp' + 'iiiiiiiii' + 'p',
    'dis_slot_weights': 'pplip',
    'dis_conv2d_wgrad_workspace': 'iiii',
    'dis_conv2d_wgrad': 'pppppiiiiiiiiip',
    'dis_conv2d_wgrad_act': 'pppipppiiiiiiiiip',
    'dis_conv2d_wgrad_bf16x3': 'pppppiiiiiiiiip',
    'dis_conv2d_dgrad_strided': 'ppppiiiiiiiiip',
    'dis_disp_head_fwd': 'ppppiiiiffp',
    'dis_disp_head_bwd': 'pppppppp' + 'iiiifp',
    'dis_disp_head_bwd_workspace': 'iiii',
    'dis_act_bwd': 'pppilp',
    'dis_act_bwd_ld': 'pipipilip',
    'dis_act_bwd_ld_bias_workspace': 'i',
    'dis_act_bwd_ld_bias': 'pipipilippp',
    'dis_copy_channels': 'pipiliip',
    'dis_gn_stats': 'ppilp',
    'dis_gn_apply': 'ppppppiliifp',
    'dis_gn_bwd_workspace': 'ii',
    'dis_gn_apply_bwd': 'ppppp' + 'pppp' + 'pp' + 'iliifip',
    'dis_add_act_fwd': 'pppilp',
    'dis_mask_weight_slots': 'pppliiip',
    'dis_conv3d_knn_select': 'ppiiiiip',
    'dis_conv3d_knn_fwd': 'ppppppppp' + 'iiiiip',
    'dis_conv3d_knn_fwd_agg': 'ppppppp' + 'ppp' + 'iiiiip',
    'dis_conv3d_knn_bwd_det': 'ppppppp' + 'ppppppp' + 'iiiiip',
    'dis_conv3d_knn_bwd_agg': 'ppppppp' + 'ppppppp' + 'iiiiip',
    'dis_conv3d_knn_bwd_det_workspace': 'iiiii',
    'dis_geo_loss_acc_doubles': '',
    'dis_geo_loss_multi_acc_doubles': 'i',
    'dis_geo_loss_fwd_multi': 'pippfppiiip',
    'dis_geo_loss_bwd_multi': 'pippfppiiip',
    'dis_conv3d_knn_bwd_workspace': '',
    'dis_conv3d_knn_bwd': 'ppppppp' + 'pppppp' + 'iiiiip',
    'dis_conv3d_knn_bwd_csr': 'ppppppp' + 'pppppp' + 'ppi' + 'iiiiip',
    'dis_conv3d_knn_bwd_stage': 'iiiii',
    'dis_conv3d_csr_workspace': 'iiiii',
    'dis_conv3d_csr_build': 'ppiiiiip',
    'dis_convg_pack_workspace': 'iii',
    'dis_convg_splitk_workspace': 'iiiiiiiiiii',
    'dis_convg_run': 'ipiipppiip' + 'iiiiiiiiiiiii' + 'p',
    'dis_convg_wgrad_workspace': 'iiiiii',
    'dis_convg_wgrad': 'piiiiiipiiiiiipp' + 'iiiip',
    'dis_colsum_workspace': 'i',
    'dis_colsum': 'piilippp',
    'dis_sigmoid_affine_fwd': 'ppfflp',
    'dis_sigmoid_affine_bwd': 'pppflp',
    'dis_augment': 'pppppppiiip',
    'dis_convb_pack_workspace': 'iii',
    'dis_convb_splitk_workspace': 'iiiiiiiiiiii',
    'dis_convb_pack_desc_bytes': '',
    'dis_convb_pack_record': 'pip',
    'dis_convb_pack_batch': 'piip',
    'dis_convb_run': 'ipiiipppiiip' + 'iiiiiiiiiiiii' + 'p',
    'dis_convb_wgrad_workspace': 'iiiiii',
    'dis_convb_wgrad': 'piiiiiiipiiiiiiipp' + 'iiiip',
    'dis_act_bwd_bf16': 'pipipilip',
    'dis_act_bwd_bf16_f32': 'pipipilip',
    'dis_act_bwd_bf16_bias': 'pipipilippp',
    'dis_copy_channels_bf16': 'piipiliip',
    'dis_colsum_bf16_workspace': 'i',
    'dis_colsum_bf16': 'piilippp',
    'dis_adam_step': 'pppplfddfifp',
    'dis_adam_step_dev': 'pppplfddfpfp',
    'dis_allreduce_unique_id': 'p',
    'dis_allreduce_init': 'ppii',
    'dis_allreduce_sum_f32': 'pplip',
    'dis_allreduce_destroy': 'p',
}
_RET_LONG = {'dis_conv2d_bwd_fused_workspace', 'dis_convb_pack_desc_bytes', 'dis_convg_splitk_workspace', 'dis_convb_splitk_workspace', 'dis_conv2d_gnsums_slots', 'dis_conv2d_wgrad_workspace', 'dis_convg_pack_workspace', 'dis_convg_wgrad_workspace',
             'dis_colsum_workspace', 'dis_convb_pack_workspace', 'dis_convb_wgrad_workspace', 'dis_colsum_bf16_workspace', 'dis_gn_bwd_workspace', 'dis_act_bwd_ld_bias_workspace', 'dis_conv3d_knn_bwd_workspace', 'dis_geo_loss_acc_doubles', 'dis_geo_loss_multi_acc_doubles', 'dis_conv3d_knn_bwd_det_workspace', 'dis_conv3d_knn_bwd_stage', 'dis_conv3d_csr_workspace', 'dis_gather_csr_workspace',
             'dis_conv2d_pack_bf16x3_size', 'dis_disp_head_bwd_workspace'}

_CT = {'p': ctypes.c_void_p, 'i': ctypes.c_int, 'l': ctypes.c_long, 'f': ctypes.c_float, 'd': ctypes.c_double}
_lib = None


def load():
    """Load libdis_hip.so and bind every entry point; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'(or `make -C depthinspace_amd/csrc`). There is no non-HIP fallback.')
    _lib = ctypes.CDLL(LIB_PATH)
    return _lib


_bound = {}


def fn(name):
    """Bound entry point (argtypes set); AttributeError if the library does not export it."""
    f = _bound.get(name)
    if f is None:
        f = getattr(load(), name)  # missing symbol => loud failure, by design
        f.argtypes = [_CT[c] for c in SIGS[name]]
        f.restype = ctypes.c_long if name in _RET_LONG else ctypes.c_int
        _bound[name] = f
    return f


def check_all_symbols():
    """Bind every entry point declared in SIGS (== include/dis_hip.h); raises on the first missing one."""
    for name in SIGS:
        fn(name)
    return len(SIGS)


class DisHipError(RuntimeError):
    pass


_ERR = {-1: 'bad shape', -2: 'unsupported configuration', -3: 'null pointer'}


# optional per-call HIP-event timing (bench.py roofline leg): list of (name, int-args, start_event, end_event)
_profile = None
_tagbuf = ctypes.create_string_buffer(96)


_profile_ptrs = []   # per recorded call: the device address of every tensor argument (None elsewhere), aligned with profile_stop()
last_profile_ptrs = []


def profile_start():
    global _profile
    _profile = []
    del _profile_ptrs[:]
    fn('dis_last_kernel')(ctypes.cast(_tagbuf, ctypes.c_void_p), len(_tagbuf), 1)


def profile_stop():
    """-> list of (name, int_args_tuple, milliseconds, kernel_tag, n_tensor_args); synchronises the device.  kernel_tag: the
    kernel family the conv dispatchers report through dis_last_kernel ('' for the other entry points); n_tensor_args: how many
    of the call's pointer arguments were not NULL (optional operands change a launch's algorithmic bytes)."""
    global _profile
    global last_profile_ptrs
    rec, _profile = _profile, None
    last_profile_ptrs = list(_profile_ptrs)
    torch.cuda.synchronize()
    return [(n, a, e0.elapsed_time(e1), t, k) for (n, a, e0, e1, t, k) in rec]


def call_try(name, *args):
    """call(), except that DIS_ERR_UNSUPPORTED (-2: this build / mode has no kernel for the configuration) is returned as False
    instead of raised - for callers that hold a general form of the same operation (never a non-HIP fallback)."""
    return call(name, *args, _soft=True)


def call(name, *args, _soft=False):
    """Call a C-ABI entry point; tensors are passed as device pointers, None as NULL.
    Appends the current HIP stream.  Raises DisHipError on a non-zero status (like the reference's
    C++ exceptions surfacing as RuntimeError, model/ext_functions.py:123-126)."""
    f = fn(name)
    conv = []
    for a in args:
        if a is None:
            conv.append(None)
        elif isinstance(a, torch.Tensor):
            conv.append(a.data_ptr())
        elif isinstance(a, ctypes.Array):
            conv.append(ctypes.cast(a, ctypes.c_void_p))
        else:
            conv.append(a)
    conv.append(torch.cuda.current_stream().cuda_stream)
    if _profile is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = f(*conv)
        e1.record()
        # the kernel family that served the call (conv dispatchers only): `*_bf16x3*` entry points run the two-term fp16 kernels
        # by default, and bench.py selects / labels launches by what actually ran
        fn('dis_last_kernel')(ctypes.cast(_tagbuf, ctypes.c_void_p), len(_tagbuf), 1)
        tag = _tagbuf.value.decode()
        _profile.append((name, tuple(a for a in args if isinstance(a, int)), e0, e1, tag,
                         sum(1 for a in args if isinstance(a, torch.Tensor))))
        _profile_ptrs.append(tuple(a.data_ptr() if isinstance(a, torch.Tensor) else None for a in args))
    else:
        rc = f(*conv)
    if rc == -2 and _soft:
        return False
    if rc != 0:
        raise DisHipError(f'{name} failed: {_ERR.get(rc, "hipError_t " + str(rc))}')
    return True


def host_floats(vals):
    vals = [float(v) for v in vals]
    return (ctypes.c_float * len(vals))(*vals)
