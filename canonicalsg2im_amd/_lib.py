"""ctypes binding of libcsg_hip.so (C ABI: include/csg_hip.h).

The library is built in-tree by `__graft_entry__.build()` (hipcc, gfx950).  There is NO fallback:
if it is missing or a symbol cannot be resolved, importing this module raises, and every op that
receives a non-HIP tensor raises as well.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSG_HIP_LIB") or os.path.join(_HERE, "libcsg_hip.so")    # override: developer builds (tools/)

MAX_TAPS = 16
ACT_NONE, ACT_LEAKY, ACT_TANH = 0, 1, 2

c_i32, c_i64, c_f32, c_f64, c_p = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_void_p


class ConvDesc(ctypes.Structure):
    """Mirror of `csg_conv_desc` (include/csg_hip.h)."""
    _fields_ = [
        ("B", c_i32), ("IHp", c_i32), ("IWp", c_i32), ("Cin", c_i32), ("x_cs", c_i32),
        ("IHv", c_i32), ("IWv", c_i32), ("in_up", c_i32),
        ("OHg", c_i32), ("OWg", c_i32), ("OHf", c_i32), ("OWf", c_i32), ("os", c_i32), ("ooy", c_i32), ("oox", c_i32),
        ("Cout", c_i32), ("y_cs", c_i32),
        ("istride", c_i32), ("ntaps", c_i32), ("wtaps", c_i32),
        ("tap_dy", c_i32 * MAX_TAPS), ("tap_dx", c_i32 * MAX_TAPS), ("tap_w", c_i32 * MAX_TAPS),
        ("act", c_i32), ("slope", c_f32), ("accumulate", c_i32), ("res_gate", c_i32),
    ]


class WinoDesc(ctypes.Structure):
    """Mirror of `csg_wino_desc` (include/csg_hip.h)."""
    _fields_ = [("B", c_i32), ("H", c_i32), ("W", c_i32), ("Cin", c_i32), ("x_cs", c_i32), ("Cout", c_i32),
                ("y_cs", c_i32), ("act", c_i32), ("slope", c_f32)]


class WinoPackItem(ctypes.Structure):
    """Mirror of `csg_wino_pack_item` (include/csg_hip.h)."""
    _fields_ = [("w", c_p), ("s_o", c_i64), ("s_i", c_i64), ("s_h", c_i64), ("s_w", c_i64), ("Cout", c_i64), ("Cin", c_i64),
                ("backward_data", c_i32), ("variant", c_i32), ("packed", c_p)]


class HingeItem(ctypes.Structure):
    """Mirror of `csg_hinge_item` (include/csg_hip.h)."""
    _fields_ = [("x", c_p), ("dx", c_p), ("sb", c_i64), ("sh", c_i64), ("sw", c_i64), ("B", c_i64), ("H", c_i64), ("W", c_i64)]


class FewDesc(ctypes.Structure):
    """Mirror of `csg_few_desc` (include/csg_hip.h)."""
    _fields_ = [("B", c_i32), ("IH", c_i32), ("IW", c_i32), ("Cin", c_i32), ("x_cs", c_i32), ("KH", c_i32), ("KW", c_i32),
                ("pad", c_i32), ("cout_real", c_i32), ("act", c_i32), ("slope", c_f32), ("in_act", c_i32), ("in_slope", c_f32)]


class SnFwdItem(ctypes.Structure):
    """Mirror of `csg_sn_fwd_item` (include/csg_hip.h)."""
    _fields_ = [("w", c_p), ("u", c_p), ("v", c_p), ("Cout", c_i64), ("K", c_i64), ("w_eff", c_p), ("cl_Cin", c_i64),
                ("sigma", c_p), ("u_used", c_p), ("v_used", c_p), ("workspace", c_p), ("workspace_bytes", c_i64)]


class SnBwdItem(ctypes.Structure):
    """Mirror of `csg_sn_bwd_item` (include/csg_hip.h)."""
    _fields_ = [("dweff", c_p), ("Cout", c_i64), ("Cin", c_i64), ("KH", c_i64), ("KW", c_i64), ("s0", c_i64), ("s1", c_i64),
                ("s2", c_i64), ("s3", c_i64), ("w", c_p), ("u_used", c_p), ("v_used", c_p), ("sigma", c_p), ("dw", c_p),
                ("workspace", c_p), ("workspace_bytes", c_i64)]


# name -> (restype, argtypes); every symbol declared in include/csg_hip.h
class GemmDesc(ctypes.Structure):
    """csg_gemm_desc (include/csg_hip.h)."""
    _fields_ = [("M", ctypes.c_int64), ("N", ctypes.c_int64), ("K", ctypes.c_int64),
                ("lda", ctypes.c_int64), ("ldb", ctypes.c_int64), ("ldy", ctypes.c_int64), ("ldg", ctypes.c_int64),
                ("act", ctypes.c_int32), ("slope", ctypes.c_float), ("gate_slope", ctypes.c_float)]


SIGNATURES = {
    "csg_version": (c_i32, []),
    "csg_last_error": (ctypes.c_char_p, []),
    "csg_prof_enable": (c_i32, [c_i32]),
    "csg_prof_reset": (c_i32, []),
    "csg_prof_num_kernels": (c_i32, []),
    "csg_prof_kernel_name": (ctypes.c_char_p, [c_i32]),
    "csg_prof_read": (c_i32, [c_i32, ctypes.POINTER(c_f64), ctypes.POINTER(c_i64), ctypes.POINTER(c_f64)]),
    "csg_embed_fwd": (c_i32, [c_p, c_i64, c_i64, c_p, c_i64, c_i64, c_p, c_i64, c_i64, c_p]),
    "csg_embed_bwd_workspace": (c_i64, [c_i64, c_i64, c_i64]),
    "csg_embed_bwd": (c_i32, [c_p, c_i64, c_i64, c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_p]),
    "csg_real_object_mask": (c_i32, [c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "csg_graph_csr_build": (c_i32, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p]),
    "csg_gather_concat_fwd": (c_i32, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "csg_gather_concat_bwd_workspace": (c_i64, [c_i64, c_i64, c_i64, c_i64]),
    "csg_gather_concat_bwd": (c_i32, [c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_i64, c_p]),
    "csg_segment_avg_fwd_workspace": (c_i64, [c_i64, c_i64, c_i64, c_i64]),
    "csg_segment_avg_fwd": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p,
                                    c_i64, c_p]),
    "csg_segment_avg_bwd": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i32,
                                    c_p, c_p, c_p, c_p]),
    "csg_layout_fwd": (c_i32, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_i64,
                               c_i64, c_p]),
    "csg_disc_input_fwd": (c_i32, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_i64, c_i64, c_i64,
                                   c_i64, c_p, c_i64, c_p]),
    "csg_layout_bwd_workspace": (c_i64, [c_i64, c_i64, c_i64, c_i64, c_i64, c_i32, c_i32]),
    "csg_layout_bwd": (c_i32, [c_p, c_i64, c_i64, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                               c_i64, c_p, c_i32, c_p, c_p, c_p, c_i64, c_p]),
    "csg_layout_bwd_masks": (c_i32, [c_p, c_i64, c_i64, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64,
                                     c_p, c_p, c_i32, c_p]),
    "csg_layout_mass": (c_i32, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "csg_layout_paint": (c_i32, [c_p, c_p, c_p, c_i64, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_i64,
                                 c_i64, c_p]),
    "csg_conv_fwd_workspace": (c_i64, [ctypes.POINTER(ConvDesc)]),
    "csg_conv_fwd": (c_i32, [ctypes.POINTER(ConvDesc), c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_p]),
    "csg_conv_bwd_weight_workspace": (c_i64, [ctypes.POINTER(ConvDesc)]),
    "csg_conv_bwd_weight": (c_i32, [ctypes.POINTER(ConvDesc), c_p, c_p, c_p, c_p, c_p, c_i64, c_p]),
    "csg_wino_pack_bytes": (c_i64, [c_i64, c_i64]),
    "csg_wino_pack_weights": (c_i32, [c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i32, c_p, c_p, c_p]),
    "csg_conv_fwd_multi_workspace": (c_i64, [ctypes.POINTER(ConvDesc), c_i32]),
    "csg_conv_fwd_multi": (c_i32, [ctypes.POINTER(ConvDesc), c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_p]),
    "csg_gemm_supported": (c_i32, [ctypes.POINTER(GemmDesc)]),
    "csg_gemm_nt": (c_i32, [ctypes.POINTER(GemmDesc), c_p, c_p, c_p, c_p, c_p, c_p]),
    "csg_gemm_tn_workspace": (c_i64, [c_i64, c_i64, c_i64]),
    "csg_gemm_tn": (c_i32, [c_i64, c_i64, c_i64, c_p, c_i64, c_p, c_i64, c_p, c_p, c_p, c_i64, c_p]),
    "csg_conv_few_supported": (c_i32, [ctypes.POINTER(FewDesc)]),
    "csg_conv_few_fwd_workspace": (c_i64, [ctypes.POINTER(FewDesc)]),
    "csg_conv_few_fwd": (c_i32, [ctypes.POINTER(FewDesc), c_p, c_p, c_p, c_p, c_p, c_i64, c_p]),
    "csg_conv_few_bwd_data": (c_i32, [ctypes.POINTER(FewDesc), c_p, c_p, c_p, c_p]),
    "csg_conv_few_bwd_weight_workspace": (c_i64, [ctypes.POINTER(FewDesc)]),
    "csg_conv_few_bwd_weight": (c_i32, [ctypes.POINTER(FewDesc), c_p, c_p, c_p, c_p, c_p, c_i64, c_p]),
    "csg_wino_conv_workspace": (c_i64, [ctypes.POINTER(WinoDesc)]),
    "csg_wino_conv": (c_i32, [ctypes.POINTER(WinoDesc), c_p, c_p, c_p, c_p, c_p, c_f32, c_p, c_p, c_i64, c_p]),
    "csg_wino_bwd_weight_workspace": (c_i64, [ctypes.POINTER(WinoDesc)]),
    "csg_wino_bwd_weight": (c_i32, [ctypes.POINTER(WinoDesc), c_p, c_p, c_p, c_p, c_p, c_i64, c_p]),
    "csg_wino4_bwd_weight_workspace": (c_i64, [ctypes.POINTER(WinoDesc)]),
    "csg_wino4_bwd_weight": (c_i32, [ctypes.POINTER(WinoDesc), c_p, c_p, c_p, c_p, c_p, c_i64, c_p]),
    "csg_wino4_supported": (c_i32, [ctypes.POINTER(WinoDesc)]),
    "csg_wino4_persistent": (c_i32, [c_i32]),
    "csg_wino_pack_weights_multi": (c_i32, [ctypes.POINTER(WinoPackItem), c_i32, c_p]),
    "csg_wino4_pack_bytes": (c_i64, [c_i64, c_i64]),
    "csg_wino4_pack_weights": (c_i32, [c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i32, c_p, c_p, c_p]),
    "csg_wino4_conv_workspace": (c_i64, [ctypes.POINTER(WinoDesc)]),
    "csg_wino4_conv": (c_i32, [ctypes.POINTER(WinoDesc), c_p, c_p, c_p, c_p, c_p, c_f32, c_p, c_p, c_i64, c_p]),
    "csg_wino4_conv_part": (c_i32, [ctypes.POINTER(WinoDesc), c_p, c_p, c_i64, c_i64, c_p, c_p, c_p, c_i64, c_p, c_p, c_f32,
                                    c_p, c_p]),
    "csg_wino4_conv_spade_supported": (c_i32, [ctypes.POINTER(WinoDesc)]),
    "csg_wino4_conv_spade": (c_i32, [ctypes.POINTER(WinoDesc), c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_p, c_f32, c_p, c_p]),
    "csg_wino34_supported": (c_i32, [ctypes.POINTER(WinoDesc), c_i32]),
    "csg_wino34_pack_weights": (c_i32, [c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i32, c_p, c_p, c_p]),
    "csg_wino34_conv_workspace": (c_i64, [ctypes.POINTER(WinoDesc), c_i32]),
    "csg_wino34_conv": (c_i32, [ctypes.POINTER(WinoDesc), c_i32, c_p, c_p, c_p, c_p, c_p, c_f32, c_p, c_p, c_i64, c_p]),
    "csg_act_bwd": (c_i32, [c_p, c_p, c_i64, c_i32, c_f32, c_p, c_p]),
    "csg_colsum": (c_i32, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_p]),
    "csg_norm_stats": (c_i32, [c_p, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_p]),
    "csg_norm_finalize": (c_i32, [c_p, c_i64, c_i64, c_f64, c_f32, c_i32, c_p, c_p, c_p, c_p, c_f32, c_p]),
    "csg_norm_stats_finalize": (c_i32, [c_p, c_i64, c_i64, c_i64, c_p, c_i64, c_f64, c_f32, c_p, c_p, c_p, c_p, c_p, c_p, c_f32,
                                        c_p]),
    "csg_norm_apply_fwd": (c_i32, [c_p, c_p, c_p, c_p, c_f32, c_i64, c_i64, c_i64, c_p, c_p, c_f32, c_p, c_p]),
    "csg_norm_apply_bwd_reduce": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_p, c_f32, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_i64,
                                          c_i64, c_p]),
    "csg_norm_apply_bwd_dx": (c_i32, [c_p, c_p, c_p, c_p, c_p, c_f32, c_p, c_f64, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_f32,
                                      c_p, c_p, c_i64, c_p]),
    "csg_nearest_resize_fwd": (c_i32, [c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "csg_nearest_resize_bwd": (c_i32, [c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "csg_upsample2x_fwd": (c_i32, [c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "csg_upsample2x_bwd": (c_i32, [c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "csg_avgpool3s2_fwd": (c_i32, [c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "csg_avgpool3s2_bwd": (c_i32, [c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "csg_avgpool3s2_bwd_add": (c_i32, [c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_p]),
    "csg_crop_fwd": (c_i32, [c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_i64, c_p]),
    "csg_crop_bwd": (c_i32, [c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "csg_crop_bwd_boxes": (c_i32, [c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_p,
                                   c_p]),
    "csg_maxpool2_fwd": (c_i32, [c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "csg_maxpool2_bwd": (c_i32, [c_p, c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "csg_avgpool2_fwd": (c_i32, [c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "csg_avgpool2_bwd": (c_i32, [c_p, c_i64, c_i64, c_i64, c_i64, c_p, c_p]),
    "csg_l1_mean_workspace": (c_i64, [c_i64]),
    "csg_l1_mean_fwd": (c_i32, [c_p, c_p, c_i64, c_p, c_p, c_i64, c_p]),
    "csg_l1_mean_bwd": (c_i32, [c_p, c_p, c_p, c_i64, c_p, c_p]),
    "csg_hinge_mean_fwd": (c_i32, [ctypes.POINTER(HingeItem), c_i32, c_i32, c_p, c_p]),
    "csg_hinge_mean_bwd": (c_i32, [ctypes.POINTER(HingeItem), c_i32, c_i32, c_p, c_p]),
    "csg_spectral_norm_workspace": (c_i64, [c_i64, c_i64]),
    "csg_spectral_norm_fwd": (c_i32, [c_p, c_p, c_p, c_i64, c_i64, c_i32, c_f32, c_p, c_i64, c_p, c_p, c_p, c_p, c_i64,
                                      c_p]),
    "csg_spectral_norm_bwd": (c_i32, [c_p, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_p,
                                      c_p, c_p, c_i64, c_p]),
    "csg_spectral_norm_fwd_multi": (c_i32, [ctypes.POINTER(SnFwdItem), c_i32, c_i32, c_f32, c_p]),
    "csg_spectral_norm_bwd_multi": (c_i32, [ctypes.POINTER(SnBwdItem), c_i32, c_p]),
    "csg_canon_workspace": (c_i64, [c_i64]),
    "csg_canon_build": (c_i32, [c_p, c_p, c_p, c_p, c_i64, c_i64, ctypes.POINTER(c_i32), c_i64, c_i32, c_i32, c_p, c_i64,
                                c_p, c_p]),
    "csg_canon_emit": (c_i32, [c_p, c_p, c_i64, c_i64, ctypes.POINTER(c_i32), c_i64, c_i32, c_i32, c_p, c_p, c_i64, c_p,
                               c_p, c_p]),
    "csg_canon_converse": (c_i32, [c_p, c_p, c_i64, c_i64, ctypes.POINTER(c_i32), c_i64, c_i32, c_i32, c_p, c_p, c_p, c_p,
                                   c_i64, c_p, c_p, c_p]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libcsg_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "from the repository root; there is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def last_error():
    return lib.csg_last_error().decode("utf-8", "replace")


def check(rc, what=""):
    if rc != 0:
        raise RuntimeError("libcsg_hip %s failed (%d): %s" % (what, rc, last_error()))


def ptr(t):
    """Device pointer of a HIP tensor (None -> NULL).  CPU tensors are refused: no fallback."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("canonicalsg2im_amd ops need HIP (cuda) tensors; got a %s tensor — there is no CPU path"
                           % t.device)
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """The current HIP stream of the current device as a void*.  `torch.cuda.current_stream()` builds a Stream object
    through several Python layers (~13 us; a training step asks ~900 times): the raw query is one C call."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---------------------------------------------------------------- per-kernel timing
def prof_enable(on=True):
    """True / 1: time every launch; 2: only the dominant convolution kernels (k_wino_conv2, k_igemm_fwd<128>);
    3: only the streaming (HBM-bound) kernels; False / 0: off."""
    check(lib.csg_prof_enable(int(on)), "prof_enable")


def prof_reset():
    check(lib.csg_prof_reset(), "prof_reset")


def prof_read():
    """{kernel name: (ms, launches, work)} for kernels that ran."""
    out = {}
    for k in range(lib.csg_prof_num_kernels()):
        ms, n, w = c_f64(), c_i64(), c_f64()
        check(lib.csg_prof_read(k, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(w)), "prof_read")
        if n.value:
            out[lib.csg_prof_kernel_name(k).decode()] = (ms.value, n.value, w.value)
    return out
