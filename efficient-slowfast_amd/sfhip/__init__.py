"""sfhip — ctypes binding of libsfhip.so (include/sfhip.h), the MI355X kernels of the SlowFast/CMDA path.

PyTorch is used for device memory (torch.empty on the caching allocator), the current HIP stream and
parameter preprocessing only; every activation-sized computation is a libsfhip kernel.  There is NO CPU
or eager-PyTorch fallback: if the shared library is missing, or a tensor is not on a GPU, the ops raise.

Activations are NDHWC views (`Act`): a contiguous buffer [N, T, H, W, pitch] plus a channel slice
(coff, C).  Producers write straight into slices of wider buffers, so torch.cat never runs on the path.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SF_LIB: an alternative build of the library (A/B runs of compiler flags inside one GPU lease; default: the in-tree build)
_LIB_PATH = os.environ.get("SF_LIB") or os.path.join(_HERE, "libsfhip.so")
_lib = None

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SOFTMAX, ACT_HSIGMOID, ACT_RELU6 = 0, 1, 2, 3, 4, 5


def _act(relu):
    """relu: False/True (ReLU) or 6 (ReLU6)."""
    return ACT_RELU6 if relu == 6 else (ACT_RELU if relu else ACT_NONE)
SF_EINVAL, SF_EALIGN, SF_ELAUNCH, SF_ENOTTAKEN = -1, -2, -3, -4
_ERR = {-1: "SF_EINVAL (inconsistent descriptor)", -2: "SF_EALIGN", -3: "SF_ELAUNCH (hip launch failed)",
        -4: "SF_ENOTTAKEN (shape not served by this specialised entry point)"}


class SfhipError(RuntimeError):
    pass


class ConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "N", "Ti", "Hi", "Wi", "Cin", "in_cs", "in_coff", "To", "Ho", "Wo", "Cout", "out_cs", "out_coff",
        "out_cmul", "kT", "kH", "kW", "sT", "sH", "sW", "pT", "pH", "pW", "dT", "dH", "dW", "cin_pad", "act",
        "res_cs", "res_coff", "transposed", "os_T", "os_H", "os_W", "oo_T", "oo_H", "oo_W", "ob_T", "ob_H", "ob_W")]


class PoolDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "N", "Ti", "Hi", "Wi", "C", "in_cs", "in_coff", "To", "Ho", "Wo", "out_cs", "out_coff",
        "kT", "kH", "kW", "sT", "sH", "sW", "pT", "pH", "pW", "is_avg")]


EXPORTS = [
    "sf_abi_version", "sf_build_arch", "sf_ncthw_to_ndhwc", "sf_ndhwc_to_ncthw", "sf_conv_fwd", "sf_dwconv_fwd",
    "sf_pool_fwd", "sf_maxpool_fwd_arg", "sf_maxpool_bwd_arg", "sf_tmax_mean_ws_floats", "sf_tmax_mean", "sf_gate_apply", "sf_attn_fwd", "sf_head_act_mean",
    "sf_copy_channels", "sf_channel_stats_ws_floats", "sf_channel_stats", "sf_affine_fwd", "sf_bn_train_stats",
    "sf_conv_wgrad_splits", "sf_conv_wgrad", "sf_bn_bwd_ws_floats", "sf_bn_bwd_reduce", "sf_bn_bwd_apply",
    "sf_attn_bwd", "sf_maxpool_bwd", "sf_tmax_dot", "sf_eca_bwd_apply", "sf_eca_gate_bwd", "sf_bcast_add", "sf_rowdot", "sf_axpy", "sf_act_bwd",
    "sf_dwconv_dgrad", "sf_dwconv_wgrad_ws_floats", "sf_dwconv_wgrad", "sf_gather_add",
    "sf_bn_train_stats_split", "sf_affine_fwd_split", "sf_bn_bwd_reduce_split", "sf_bn_bwd_apply_split",
    "sf_clip_prologue", "sf_conv_wgrad_finish", "sf_bn_bwd_reduce_acc", "sf_row_softmax_fwd", "sf_row_softmax_bwd",
    "sf_attn_bwd_fused_ws_floats", "sf_attn_bwd_fused", "sf_pack_conv_weight", "sf_conv_fwd_ws_floats",
    "sf_conv_fwd_ws", "sf_attn_fwd_ws_floats", "sf_attn_fwd_ws", "sf_affine_fwd_mask", "sf_bn_bwd_apply_first", "sf_maxpool_bwd_first",
    "sf_conv_tune", "sf_conv_stats_ws_floats", "sf_conv_fwd_stats", "sf_bn_train_stats_merge",
    "sf_attn_products_per_fp32", "sf_pack_conv_weights", "sf_attn_bwd_variant", "sf_attn_tune",
    "sf_bx_planes_elems", "sf_bx_split", "sf_bx_split_batched", "sf_conv_rows_parts", "sf_conv_pw_ws_floats", "sf_conv_pw_stats_floats", "sf_conv_fwd_pw", "sf_conv_bx_ws_floats", "sf_conv_fwd_bx", "sf_conv_wgrad_bx_splits",
    "sf_conv_wgrad_bx_ws_floats", "sf_conv_wgrad_bx",
    "sf_conv_fwd_grouped", "sf_conv_wgrad_grouped_splits", "sf_conv_wgrad_grouped", "sf_channel_shuffle",
    "sf_dwconv_wgrad_param",
]
_LONG_RET = ("sf_tmax_mean_ws_floats", "sf_channel_stats_ws_floats", "sf_bn_bwd_ws_floats",
             "sf_dwconv_wgrad_ws_floats", "sf_attn_bwd_fused_ws_floats", "sf_conv_fwd_ws_floats",
             "sf_attn_fwd_ws_floats", "sf_conv_stats_ws_floats", "sf_bx_planes_elems", "sf_conv_bx_ws_floats",
             "sf_conv_wgrad_bx_ws_floats")


def lib_path():
    return _LIB_PATH


def lib():
    """Load libsfhip.so (built by `make -C efficient-slowfast_amd/csrc` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise SfhipError(
                "libsfhip.so not found at %s — build it with `make -C efficient-slowfast_amd/csrc` "
                "(there is no CPU fallback for the SlowFast hot path)" % _LIB_PATH)
        L = ctypes.CDLL(_LIB_PATH)
        vp, ci, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
        L.sf_abi_version.restype = ci
        L.sf_build_arch.restype = ctypes.c_char_p
        L.sf_ncthw_to_ndhwc.argtypes = [vp, vp] + [ci] * 9 + [vp]
        L.sf_ndhwc_to_ncthw.argtypes = [vp, ci, ci, vp] + [ci] * 5 + [vp]
        L.sf_conv_fwd.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 7
        L.sf_dwconv_fwd.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 7
        L.sf_pool_fwd.argtypes = [ctypes.POINTER(PoolDesc), vp, vp, vp]
        L.sf_maxpool_fwd_arg.argtypes = [ctypes.POINTER(PoolDesc), vp, vp, vp, vp]
        L.sf_maxpool_bwd_arg.argtypes = [ctypes.POINTER(PoolDesc), vp, vp, ci, ci, vp, ci, ci, ci, vp]
        L.sf_tmax_mean_ws_floats.argtypes = [ci, ci]
        L.sf_tmax_mean_ws_floats.restype = cl
        L.sf_tmax_mean.argtypes = [vp, ci, ci] + [ci] * 6 + [vp, vp, vp]
        L.sf_gate_apply.argtypes = [vp, ci, ci] + [ci] * 6 + [vp, vp, vp, vp, ci, vp, ci, ci, vp]
        L.sf_attn_fwd.argtypes = [vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, vp, ci, vp, ci, ci] + [ci] * 6 + [vp, vp, vp]
        L.sf_affine_fwd_mask.argtypes = [vp, ci, ci] + [ci] * 5 + [vp, vp, vp, ci, ci, ci, vp, ci, ci, vp, vp]
        L.sf_attn_fwd_ws_floats.argtypes = [ci, ci, ci]
        L.sf_attn_fwd_ws_floats.restype = cl
        L.sf_attn_fwd_ws.argtypes = L.sf_attn_fwd.argtypes[:-1] + [vp, vp]
        L.sf_attn_bwd.argtypes = [vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, ci, vp, ci, vp, ci, ci, ci, ci, ci, vp]
        L.sf_head_act_mean.argtypes = [vp, ci, ci, ci, ci, vp, vp]
        L.sf_copy_channels.argtypes = [vp, ci, ci, vp, ci, ci, ci, cl, ci, vp]
        L.sf_channel_shuffle.argtypes = [vp, ci, ci, vp, ci, ci, ci, cl, ci, ci, vp]
        L.sf_channel_stats_ws_floats.argtypes = [ci]
        L.sf_channel_stats_ws_floats.restype = cl
        L.sf_channel_stats.argtypes = [vp, ci, ci, cl, ci, vp, vp, vp, vp]
        L.sf_bn_train_stats.argtypes = [vp, ci, ci, cl, ci, vp, vp, ctypes.c_float, ctypes.c_float] + [vp] * 9
        L.sf_affine_fwd.argtypes = [vp, ci, ci] + [ci] * 5 + [vp, vp, vp, ci, ci, ci, ci, vp, ci, ci, ci, vp]
        cf = ctypes.c_float
        L.sf_conv_wgrad_splits.argtypes = [ctypes.POINTER(ConvDesc)]
        L.sf_conv_wgrad.argtypes = [ctypes.POINTER(ConvDesc), vp, vp, ci, ci, vp, vp]
        L.sf_conv_fwd_grouped.argtypes = [ctypes.POINTER(ConvDesc), ci, ci] + [vp] * 7
        L.sf_conv_wgrad_grouped_splits.argtypes = [ctypes.POINTER(ConvDesc), ci]
        L.sf_conv_wgrad_grouped.argtypes = [ctypes.POINTER(ConvDesc), ci, vp, vp, ci, ci, vp, vp]
        L.sf_bn_bwd_ws_floats.argtypes = [ci]
        L.sf_bn_bwd_ws_floats.restype = cl
        L.sf_bn_bwd_reduce.argtypes = [vp, ci, ci, vp, ci, ci, vp, ci, ci] + [ci] * 7 + [vp] * 5 + [vp]
        L.sf_bn_bwd_apply.argtypes = ([vp, ci, ci, vp, ci, ci, vp, ci, ci] + [ci] * 7 + [vp] * 5 +
                                      [vp, ci, ci, vp, ci, ci, vp])
        L.sf_bn_bwd_apply_first.argtypes = L.sf_bn_bwd_apply.argtypes
        L.sf_maxpool_bwd_first.argtypes = [ctypes.POINTER(PoolDesc), vp, vp, vp, ci, ci, vp, ci, ci, vp]
        L.sf_maxpool_bwd.argtypes = [ctypes.POINTER(PoolDesc), vp, vp, vp, ci, ci, vp, ci, ci, vp]
        L.sf_tmax_dot.argtypes = [vp, ci, ci] + [ci] * 6 + [vp, ci, ci, vp, vp, vp]
        L.sf_eca_bwd_apply.argtypes = [vp, ci, ci] + [ci] * 6 + [vp, ci, ci, vp, vp, vp, ci, ci, vp]
        L.sf_eca_gate_bwd.argtypes = [vp, vp, vp, ci, ci, cf, vp, vp, vp, vp]
        L.sf_bcast_add.argtypes = [vp, ci, ci, ci, cl, ci, vp, cf, vp]
        L.sf_rowdot.argtypes = [vp, ci, ci, vp, ci, ci, cl, ci, cf, vp, vp]
        L.sf_axpy.argtypes = [vp, ci, ci, cf, vp, ci, ci, cl, ci, ci, vp]
        L.sf_act_bwd.argtypes = [vp, ci, ci, vp, ci, ci, ci, vp, ci, ci, cl, ci, ci, vp]
        L.sf_dwconv_dgrad.argtypes = [ctypes.POINTER(ConvDesc), vp, ci, ci, vp, vp, ci, ci, ci, vp]
        L.sf_dwconv_wgrad_ws_floats.argtypes = [ctypes.POINTER(ConvDesc), ci]
        L.sf_dwconv_wgrad_ws_floats.restype = cl
        L.sf_dwconv_wgrad.argtypes = [ctypes.POINTER(ConvDesc), vp, vp, ci, ci, ci, vp, vp, vp]
        L.sf_dwconv_wgrad_param.argtypes = [ctypes.POINTER(ConvDesc), vp, vp, ci, ci, ci, vp, ci, vp, vp]
        L.sf_gather_add.argtypes = [vp, ci, ci, ci, vp, ci, ci, cl, ci, ci, vp]
        L.sf_clip_prologue.argtypes = [vp] + [ci] * 10 + [ctypes.POINTER(ctypes.c_float)] * 2 + [vp, ci, vp, ci, ci, ci, vp]
        L.sf_attn_bwd_fused_ws_floats.argtypes = [ci, ci, ci]
        L.sf_attn_bwd_fused_ws_floats.restype = cl
        L.sf_attn_bwd_fused.argtypes = [vp, ci, vp, ci, vp, ci, vp, ci, vp, vp, vp, vp, ci, vp, ci, vp, ci, ci, ci, ci,
                                        vp, vp]
        L.sf_conv_fwd_ws_floats.argtypes = [ctypes.POINTER(ConvDesc)]
        L.sf_conv_fwd_ws_floats.restype = cl
        L.sf_conv_stats_ws_floats.argtypes = [ctypes.POINTER(ConvDesc)]
        L.sf_conv_stats_ws_floats.restype = cl
        L.sf_conv_fwd_stats.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 7 + [ctypes.POINTER(ctypes.c_int), vp]
        L.sf_bn_train_stats_merge.argtypes = [vp, ci, ci, vp, vp, ctypes.c_float, ctypes.c_float] + [vp] * 8
        L.sf_conv_fwd_ws.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 8
        L.sf_pack_conv_weight.argtypes = [vp, ci, ci, ci, vp, ci, vp, ci, vp]
        L.sf_conv_tune.argtypes = [ci, ci]
        L.sf_attn_tune.argtypes = [ci, ci]
        L.sf_bx_planes_elems.argtypes = [cl, ci]
        L.sf_bx_planes_elems.restype = cl
        L.sf_bx_split.argtypes = [vp, ci, ci, cl, ci, vp, vp]
        L.sf_bx_split_batched.argtypes = [vp, vp, ci, ci, vp]
        L.sf_conv_pw_ws_floats.argtypes = [ctypes.POINTER(ConvDesc), ci]
        L.sf_conv_pw_ws_floats.restype = cl
        L.sf_conv_rows_parts.argtypes = [ctypes.POINTER(ConvDesc)]
        L.sf_conv_pw_stats_floats.argtypes = [ctypes.POINTER(ConvDesc)]
        L.sf_conv_pw_stats_floats.restype = cl
        L.sf_conv_fwd_pw.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 9 + [ctypes.POINTER(ctypes.c_int), vp]
        L.sf_conv_bx_ws_floats.argtypes = [ctypes.POINTER(ConvDesc), ci, ci]
        L.sf_conv_bx_ws_floats.restype = cl
        L.sf_conv_fwd_bx.argtypes = [ctypes.POINTER(ConvDesc)] + [vp] * 10
        L.sf_conv_wgrad_bx_splits.argtypes = [ctypes.POINTER(ConvDesc)]
        L.sf_conv_wgrad_bx_ws_floats.argtypes = [ctypes.POINTER(ConvDesc), ci, ci]
        L.sf_conv_wgrad_bx_ws_floats.restype = cl
        L.sf_conv_wgrad_bx.argtypes = [ctypes.POINTER(ConvDesc), vp, vp, vp, ci, ci, vp, vp, vp, vp]
        L.sf_attn_bwd_variant.argtypes = [ci, ci, ci]
        L.sf_row_softmax_fwd.argtypes = [vp, ci, ci, cl, ci, cf, vp]
        L.sf_row_softmax_bwd.argtypes = [vp, ci, ci, vp, ci, ci, cl, ci, cf, vp]
        L.sf_conv_wgrad_finish.argtypes = [vp, ci, ci, ci, ci, ci, ci, vp, ci, vp]
        L.sf_bn_bwd_reduce_acc.argtypes = [vp, ci, ci, vp, ci, ci, vp, ci, ci] + [ci] * 7 + [vp] * 5 + [vp, vp, vp]
        L.sf_bn_train_stats_split.argtypes = [vp, ci, ci, ci, cl, ci, ci, vp, vp, cf, cf] + [vp] * 9
        L.sf_affine_fwd_split.argtypes = [vp, ci, ci] + [ci] * 6 + [vp, vp, vp, ci, ci, ci, ci, vp, ci, ci, ci, vp]
        L.sf_bn_bwd_reduce_split.argtypes = [vp, ci, ci, vp, ci, ci, vp, ci, ci] + [ci] * 8 + [vp] * 5 + [vp]
        L.sf_bn_bwd_apply_split.argtypes = ([vp, ci, ci, vp, ci, ci, vp, ci, ci] + [ci] * 8 + [vp] * 5 +
                                            [vp, ci, ci, vp, ci, ci, vp])
        for name in EXPORTS:
            fn = getattr(L, name)
            if name != "sf_build_arch" and name not in _LONG_RET:
                fn.restype = ci
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise SfhipError("%s failed: %s" % (what, _ERR.get(rc, rc)))


CALLS = 0  # C-ABI calls issued by this process (every entry point takes the stream: counted there; bench.py's launch_bound block)


def _stream():
    global CALLS
    CALLS += 1
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


EVENT_TRACE = None  # bench.py sets this to a list: (tag, start_event, end_event) per traced launch


def _traced(tag, fn):
    """Run `fn` (the launches of one C-ABI call on the current stream) between two HIP events when tracing is on."""
    if EVENT_TRACE is None:
        return fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn()
    e1.record()
    EVENT_TRACE.append((tag, e0, e1))
    return rc


def _require_gpu(t, what):
    if not t.is_cuda:
        raise SfhipError("%s: tensor is on %s — the SlowFast hot path only runs on an MI355X GPU "
                         "(no CPU fallback)" % (what, t.device))
    if t.dtype != torch.float32:
        raise SfhipError("%s: expected float32, got %s" % (what, t.dtype))
    if t.device.index != torch.cuda.current_device():
        # kernels are enqueued on the CURRENT device's stream: a tensor of another GPU would be launched on the wrong one
        raise SfhipError("%s: tensor is on %s but the current device is cuda:%d — call torch.cuda.set_device(%d) (one "
                         "process per GPU)" % (what, t.device, torch.cuda.current_device(), t.device.index))


# ------------------------------------------------------------------------------------------------ Act
class Act(object):
    """NDHWC activation view: `buf` is a contiguous [N, T, H, W, pitch] tensor, the view is channels
    [coff, coff + C)."""
    __slots__ = ("buf", "coff", "C")

    def __init__(self, buf, coff=0, C=None):
        assert buf.dim() == 5 and buf.is_contiguous()
        self.buf = buf
        self.coff = coff
        self.C = buf.shape[4] - coff if C is None else C

    N = property(lambda s: s.buf.shape[0])
    T = property(lambda s: s.buf.shape[1])
    H = property(lambda s: s.buf.shape[2])
    W = property(lambda s: s.buf.shape[3])
    cs = property(lambda s: s.buf.shape[4])
    rows = property(lambda s: s.buf.shape[0] * s.buf.shape[1] * s.buf.shape[2] * s.buf.shape[3])

    def slice(self, off, C):
        assert 0 <= off and off + C <= self.C
        return Act(self.buf, self.coff + off, C)

    def ptr(self):
        return ctypes.c_void_p(self.buf.data_ptr())

    @property
    def shape_ncthw(self):
        return (self.N, self.C, self.T, self.H, self.W)

    def __repr__(self):
        return "Act(N=%d,T=%d,H=%d,W=%d,C=%d@%d/%d)" % (self.N, self.T, self.H, self.W, self.C, self.coff, self.cs)


def new_act(like_or_dev, N, T, H, W, C, before=0, after=0):
    """Allocate [N,T,H,W,before+C+after] and return the middle C-channel view."""
    dev = like_or_dev.buf.device if isinstance(like_or_dev, Act) else like_or_dev
    buf = torch.empty((N, T, H, W, before + C + after), dtype=torch.float32, device=dev)
    return Act(buf, before, C)


def from_ncthw(x, cpad=None, ph=0, pw=0, wp=None):
    """NCTHW torch tensor -> (optionally channel/border padded) NDHWC Act."""
    _require_gpu(x, "from_ncthw")
    x = x.contiguous()
    N, C, T, H, W = x.shape
    cpad = C if cpad is None else cpad
    wp = W + 2 * pw if wp is None else wp
    buf = torch.empty((N, T, H + 2 * ph, wp, cpad), dtype=torch.float32, device=x.device)
    _check(lib().sf_ncthw_to_ndhwc(_ptr(x), _ptr(buf), N, C, T, H, W, cpad, ph, pw, wp, _stream()), "sf_ncthw_to_ndhwc")
    return Act(buf, 0, cpad)


class PackedClip(object):
    """A pathway input already in the stem's layout: buf [N, T, H+2*ph, Wp, 4] fp32 (channels padded to 4, zero
    H/W borders).  Produced by `clip_prologue`; accepted by the model in place of an NCTHW tensor."""

    def __init__(self, buf, channels, H, W, ph, pw):
        self.buf, self.C, self.H, self.W, self.ph, self.pw = buf, channels, H, W, ph, pw

    @property
    def shape(self):  # the logical NCTHW shape the reference's tensor would have
        return (self.buf.shape[0], self.C, self.buf.shape[1], self.H, self.W)

    @property
    def Wp(self):
        return self.buf.shape[3]

    def to_ncthw(self):
        v = self.buf[:, :, self.ph:self.ph + self.H, self.pw:self.pw + self.W, :self.C]
        return v.permute(0, 4, 1, 2, 3).contiguous()


def clip_prologue(clip_u8, dst, new_hw, yx, crop, flip, mean, std, frame_idx=None, reverse=False, ph=0, pw=0):
    """One decoded clip (uint8 [T,H,W,3], device) -> dst (float [n_frames, crop+2ph, Wp, 4] view of a PackedClip
    buffer): normalise, bilinear short-side scale, crop, flip, frame selection in one kernel."""
    _require_gpu(dst, "clip_prologue")
    if not clip_u8.is_cuda:
        raise SfhipError("clip_prologue: the uint8 clip is on %s — copy it to the GPU first" % clip_u8.device)
    assert clip_u8.dtype == torch.uint8 and clip_u8.dim() == 4 and clip_u8.shape[3] == 3 and clip_u8.is_contiguous()
    assert dst.dtype == torch.float32 and dst.is_contiguous() and dst.shape[3] == 4
    T, H, W, _ = clip_u8.shape
    n = dst.shape[0]
    assert dst.shape[1] == crop + 2 * ph
    m3 = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s3 = (ctypes.c_float * 3)(*[float(v) for v in std])
    _check(lib().sf_clip_prologue(_ptr(clip_u8), T, H, W, int(new_hw[0]), int(new_hw[1]), int(yx[0]), int(yx[1]),
                                  int(crop), 1 if flip else 0, 1 if reverse else 0, m3, s3,
                                  _ptr(frame_idx) if frame_idx is not None else None, n, _ptr(dst), ph, pw,
                                  dst.shape[2], _stream()), "sf_clip_prologue")
    return dst


def to_ncthw(a):
    out = torch.empty(a.shape_ncthw, dtype=torch.float32, device=a.buf.device)
    _check(lib().sf_ndhwc_to_ncthw(a.ptr(), a.cs, a.coff, _ptr(out), a.N, a.C, a.T, a.H, a.W, _stream()),
           "sf_ndhwc_to_ncthw")
    return out


# ------------------------------------------------------------------------------------------------ conv
def _out_dim(i, k, s, p, d):
    return (i + 2 * p - d * (k - 1) - 1) // s + 1


def pack_conv_weight(w, cin_pad=None):
    """[Cout, Cin, kT, kH, kW] -> [Cout, kT*kH*kW, cin_pad] (zero padded), contiguous."""
    cout, cin = w.shape[0], w.shape[1]
    taps = w.shape[2] * w.shape[3] * w.shape[4]
    cin_pad = (cin + 15) // 16 * 16 if cin_pad is None else cin_pad
    wp = torch.zeros((cout, taps, cin_pad), dtype=torch.float32, device=w.device)
    wp[:, :, :cin] = w.detach().reshape(cout, cin, taps).permute(0, 2, 1)
    return wp.contiguous()


def bx_planes(t, rows, C, cs=None, coff=0):
    """The bf16 piece planes [3][rows + 1][C] (uint16 tensor) of a [rows][cs] fp32 operand: sf_bx_split."""
    planes = torch.empty((lib().sf_bx_planes_elems(rows, C),), dtype=torch.int16, device=t.device)
    _check(lib().sf_bx_split(_ptr(t), C if cs is None else cs, coff, rows, C, _ptr(planes), _stream()), "sf_bx_split")
    return planes


def _weight_planes(wp):
    """Planes of a packed conv weight [Cout][taps][cin_pad], cached ON the tensor object (pack_conv_weight_pairs
    refreshes them when it overwrites the packed weight in place)."""
    pl = wp.__dict__.get("_sf_bx")
    # torch's version counter of the packed tensor: the library's own in-place re-pack (a raw kernel launch) does not
    # move it and refreshes the planes itself; any OTHER in-place write (wp.copy_(...)) does, and the planes are re-made
    if pl is None or wp.__dict__.get("_sf_bx_ver") != wp._version:
        if pl is None:
            pl = bx_planes(wp, wp.shape[0], wp.shape[1] * wp.shape[2])
        else:  # same storage: pack_conv_weight_pairs' device-side tables keep pointing at it
            _check(lib().sf_bx_split(_ptr(wp), wp.shape[1] * wp.shape[2], 0, wp.shape[0], wp.shape[1] * wp.shape[2],
                                     _ptr(pl), _stream()), "sf_bx_split")
        wp.__dict__["_sf_bx"] = pl
        wp.__dict__["_sf_bx_ver"] = wp._version
    return pl


def act_planes(a):
    """bf16 piece planes of an activation view (all rows, its C channels): what conv_bx.hip's kernels read."""
    return bx_planes(a.buf, a.rows, a.C, cs=a.cs, coff=a.coff)


def _conv_launch(d, x_ptr, w_ptr, scale, bias, res_ptr, out_ptr, device, what, w_tensor=None, x_act=None,
                 in_planes=None, keep=None):
    """sf_conv_fwd, through the split-K schedule when the shape asks for it (workspace from the caching allocator).
    Trace tag: ("conv", output positions, taps * Cin, Cout) — 2 * product = the launch's algorithmic FLOPs.
    w_tensor: the packed weight as a tensor — lets the bf16-piece path (conv_bx.hip) keep its planes across calls.
    in_planes: act_planes of the input view when the caller already has them; keep (a dict, with x_act): the planes this
    call makes are left in keep["x"] for the layer's weight gradient."""
    tag = ("conv", d.N * d.To * d.Ho * d.Wo, d.kT * d.kH * d.kW * d.Cin, d.Cout)
    if SPLIT_K and w_tensor is not None and w_tensor.shape[2] == d.Cin:
        n = lib().sf_conv_pw_ws_floats(ctypes.byref(d), 1)
        if n > 0:  # pointwise layers: activations split in registers, only the weight's planes are kept
            planes = _weight_planes(w_tensor)
            ws = torch.empty((n,), dtype=torch.float32, device=device)
            rc = _traced(tag, lambda: lib().sf_conv_fwd_pw(ctypes.byref(d), x_ptr, w_ptr, _ptr(planes), scale, bias,
                                                           res_ptr, out_ptr, _ptr(ws), None, None, _stream()))
            if rc not in (SF_EALIGN, SF_ENOTTAKEN):  # those two: refused before any launch -> the f32 kernels below
                _check(rc, what)      # (SF_EINVAL — a genuinely inconsistent descriptor or a null pointer — raises)
                return
        if in_planes is None and keep is not None and x_act is not None and not BX_AF32 and \
                lib().sf_conv_bx_ws_floats(ctypes.byref(d), 1, 1) > 0:
            in_planes = keep["x"] = act_planes(x_act)
        n = lib().sf_conv_bx_ws_floats(ctypes.byref(d), 1 if in_planes is not None else 0, 1)
        if n > 0:
            planes = _weight_planes(w_tensor)
            ws = torch.empty((n,), dtype=torch.float32, device=device)
            rc = _traced(tag, lambda: lib().sf_conv_fwd_bx(ctypes.byref(d), x_ptr, _ptr(in_planes), w_ptr,
                                                           _ptr(planes), scale, bias, res_ptr, out_ptr, _ptr(ws),
                                                           _stream()))
            if rc not in (SF_EALIGN, SF_ENOTTAKEN):
                _check(rc, what)
                return
    n = lib().sf_conv_fwd_ws_floats(ctypes.byref(d)) if SPLIT_K else 0
    if n > 0:
        ws = torch.empty((n,), dtype=torch.float32, device=device)
        _check(_traced(tag, lambda: lib().sf_conv_fwd_ws(ctypes.byref(d), x_ptr, w_ptr, scale, bias, res_ptr, out_ptr,
                                                         _ptr(ws), _stream())), what)
    else:
        _check(_traced(tag, lambda: lib().sf_conv_fwd(ctypes.byref(d), x_ptr, w_ptr, scale, bias, res_ptr, out_ptr,
                                                      _stream())), what)


SPLIT_K = os.environ.get("SF_SPLIT_K", "1") != "0"


def pack_conv_weight_pair(w):
    """(wp [Cout][taps][cin_pad], wtp [Cin][taps][cout_pad]) of an nn.Conv3d weight on the GPU in one launch."""
    _require_gpu(w, "pack_conv_weight_pair")
    w = w.detach().contiguous()
    cout, cin = w.shape[0], w.shape[1]
    taps = w.shape[2] * w.shape[3] * w.shape[4]
    cin_pad, cout_pad = (cin + 15) // 16 * 16, (cout + 15) // 16 * 16
    wp = torch.empty((cout, taps, cin_pad), dtype=torch.float32, device=w.device)
    wtp = torch.empty((cin, taps, cout_pad), dtype=torch.float32, device=w.device)
    _check(lib().sf_pack_conv_weight(_ptr(w), cout, cin, taps, _ptr(wp), cin_pad, _ptr(wtp), cout_pad, _stream()),
           "sf_pack_conv_weight")
    return wp, wtp


def pack_grouped_weight_pair(w, groups):
    """Packed weights of nn.Conv3d(groups = G) for sf_conv_fwd_grouped: (wp [Cout][taps][pad(Cin / G)], wtp
    [Cin][taps][pad(Cout / G)]) = the G per-group (forward, transposed) packs one after the other, written in place
    into two tensors (one sf_pack_conv_weight launch per group, once per parameter version)."""
    _require_gpu(w, "pack_grouped_weight_pair")
    w = w.detach().contiguous()
    cout, cin_g = w.shape[0], w.shape[1]
    assert cout % groups == 0, (w.shape, groups)
    cout_g = cout // groups
    taps = w.shape[2] * w.shape[3] * w.shape[4]
    cin_pad, cout_pad = (cin_g + 15) // 16 * 16, (cout_g + 15) // 16 * 16
    wp = torch.empty((cout, taps, cin_pad), dtype=torch.float32, device=w.device)
    wtp = torch.empty((cin_g * groups, taps, cout_pad), dtype=torch.float32, device=w.device)
    for g in range(groups):
        _check(lib().sf_pack_conv_weight(_ptr(w[g * cout_g:(g + 1) * cout_g]), cout_g, cin_g, taps,
                                         _ptr(wp[g * cout_g:(g + 1) * cout_g]), cin_pad,
                                         _ptr(wtp[g * cin_g:(g + 1) * cin_g]), cout_pad, _stream()),
               "sf_pack_conv_weight")
    return wp, wtp


_PACK_TABLES = {}


_SPLIT_TABLES = {}


def pack_conv_weight_pairs(weights, outs):
    """pack_conv_weight_pair for many weights in ONE launch.  outs: per weight the (wp, wtp) tensors to fill (None =
    allocate).  Returns the list of (wp, wtp).  The device-side table of pointers is cached per pointer set."""
    import struct
    import numpy as np
    res, recs, blk0, nb = [], [], [0], 0
    for w, o in zip(weights, outs):
        _require_gpu(w, "pack_conv_weight_pairs")
        assert w.is_contiguous() and w.dtype == torch.float32
        cout, cin = w.shape[0], w.shape[1]
        taps = w.shape[2] * w.shape[3] * w.shape[4]
        cin_pad, cout_pad = (cin + 15) // 16 * 16, (cout + 15) // 16 * 16
        ok = (o is not None and o[0] is not None and o[1] is not None and
              all(t.device == w.device and t.dtype == torch.float32 and t.is_contiguous() for t in o) and
              tuple(o[0].shape) == (cout, taps, cin_pad) and tuple(o[1].shape) == (cin, taps, cout_pad))
        if not ok:  # e.g. the (wp, None) a CPU forward cached, or a pair left on another GPU by .to(device)
            o = (torch.empty((cout, taps, cin_pad), dtype=torch.float32, device=w.device),
                 torch.empty((cin, taps, cout_pad), dtype=torch.float32, device=w.device))
        res.append(o)
        n_wp = cout * taps * cin_pad
        total = n_wp + cin * taps * cout_pad
        recs.append((w.data_ptr(), o[0].data_ptr(), o[1].data_ptr(), cout, cin, taps, cin_pad, cout_pad, 0, n_wp, total))
        # workgroups of this weight (sfhip.h): the forward layout element-wise, the transposed one in 32 x 32 tiles
        nbw = ((cout + 7) // 8) * ((cin_pad + 31) // 32) if 1 < taps <= 9 else (n_wp + 255) // 256
        nb += nbw + ((cin * taps + 31) // 32) * ((cout_pad + 31) // 32)
        blk0.append(nb)
    key = tuple(r[:8] for r in recs)  # pointers AND dims: a freed buffer's address may come back with another shape
    dev = weights[0].device
    tab = _PACK_TABLES.get(dev)
    if (tab is None or tab[0] != key) and torch.cuda.is_current_stream_capturing():
        # a new pointer set while a hipGraph is being captured (e.g. the first training step after an eval pass re-packed
        # the weights one by one): the device-side table would need a host-to-device copy, which a capture refuses —
        # one launch per weight (and per plane set) instead, all of them capturable
        for w, o in zip(weights, res):
            cout, cin = w.shape[0], w.shape[1]
            taps = w.shape[2] * w.shape[3] * w.shape[4]
            _check(lib().sf_pack_conv_weight(_ptr(w), cout, cin, taps, _ptr(o[0]), o[0].shape[2], _ptr(o[1]),
                                             o[1].shape[2], _stream()), "sf_pack_conv_weight")
            for t in o:
                pl = t.__dict__.get("_sf_bx")
                if pl is not None:
                    c = t.shape[1] * t.shape[2]
                    _check(lib().sf_bx_split(_ptr(t), c, 0, t.shape[0], c, _ptr(pl), _stream()), "sf_bx_split")
                t.__dict__.pop("_sf_classes", None)
        return res
    if tab is None or tab[0] != key:
        raw = b"".join(struct.pack("<QQQiiiiiiqq", *r) for r in recs)
        items = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).to(dev)
        starts = torch.tensor(blk0, dtype=torch.int32).to(dev)
        tab = (key, items, starts)
        _PACK_TABLES[dev] = tab
    _check(lib().sf_pack_conv_weights(_ptr(tab[1]), _ptr(tab[2]), len(recs), nb, _stream()), "sf_pack_conv_weights")
    # packed weights overwritten in place: their bf16 piece planes (conv_bx.hip) follow, all in ONE launch
    recs, blk0, nb = [], [0], 0
    for o in res:
        for t in o:
            pl = t.__dict__.get("_sf_bx")
            if pl is not None:
                rows, c = t.shape[0], t.shape[1] * t.shape[2]
                recs.append((t.data_ptr(), pl.data_ptr(), rows, c, 0))
                nb += ((rows + 1) * (c // 8) + 255) // 256
                blk0.append(nb)
            t.__dict__.pop("_sf_classes", None)  # tap-subset copies of the previous contents (strided data gradients)
    if len(recs) == 1:
        r = recs[0]
        _check(lib().sf_bx_split(r[0], r[3], 0, r[2], r[3], r[1], _stream()), "sf_bx_split")
    elif recs:
        key = tuple(recs)
        tab = _SPLIT_TABLES.get(dev)
        if tab is None or tab[0] != key:
            raw = b"".join(struct.pack("<QQqii", *r) for r in recs)
            tab = (key, torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).to(dev),
                   torch.tensor(blk0, dtype=torch.int32).to(dev))
            _SPLIT_TABLES[dev] = tab
        _check(lib().sf_bx_split_batched(_ptr(tab[1]), _ptr(tab[2]), len(recs), nb, _stream()), "sf_bx_split_batched")
    return res


def pack_dw_weight(w):
    """depthwise [C, 1, kT, kH, kW] -> [taps, C]."""
    c = w.shape[0]
    return w.detach().reshape(c, -1).t().contiguous()


CONV_STATS = os.environ.get("SF_CONV_STATS", "1") != "0"


def conv(x, wp, kernel, stride=(1, 1, 1), padding=(0, 0, 0), dilation=(1, 1, 1), scale=None, bias=None,
         relu=False, res=None, out=None, cin=None, out_cmul=1, out_reserve=(0, 0), out_thw=None, stats=False,
         keep=None):
    """Dense conv (implicit GEMM, sf_conv_fwd).  `wp` = pack_conv_weight(...).  `out`: Act to write into
    (a slice of a wider buffer) or None to allocate [.., before + Cout + after].
    stats=True returns (out, parts) with parts = (workspace, rows) of per-tile channel statistics of the output taken
    in the conv's epilogue (sf_conv_fwd_stats) for bn_train_stats_merge, or None when this shape produces none."""
    _require_gpu(x.buf, "conv")
    cout, taps, cin_pad = wp.shape
    cin = x.C if cin is None else cin
    kT, kH, kW = kernel
    To = _out_dim(x.T, kT, stride[0], padding[0], dilation[0])
    Ho = _out_dim(x.H, kH, stride[1], padding[1], dilation[1])
    Wo = _out_dim(x.W, kW, stride[2], padding[2], dilation[2])
    if out_thw is not None:  # caller restricts the output extent (stem trick: trailing columns are padding)
        assert out_thw[0] <= To and out_thw[1] <= Ho and out_thw[2] <= Wo
        To, Ho, Wo = out_thw
    if out is None:
        out = new_act(x, x.N, To, Ho, Wo, cout, out_reserve[0], out_reserve[1])
    else:
        assert (out.N, out.T, out.H, out.W) == (x.N, To, Ho, Wo), (out, (x.N, To, Ho, Wo))
        assert out.coff + (cout - 1) * out_cmul < out.cs and (out_cmul > 1 or out.C == cout), (out, cout)
    d = ConvDesc(x.N, x.T, x.H, x.W, cin, x.cs, x.coff, To, Ho, Wo, cout, out.cs, out.coff, out_cmul,
                 kT, kH, kW, stride[0], stride[1], stride[2], padding[0], padding[1], padding[2],
                 dilation[0], dilation[1], dilation[2], cin_pad, _act(relu),
                 res.cs if res is not None else 0, res.coff if res is not None else 0, 0)
    if res is not None:
        assert res.rows == out.rows and res.C == cout
    if stats and CONV_STATS and SPLIT_K and scale is None and res is None and not relu and out_cmul == 1 and \
            wp.shape[2] == cin:
        n = lib().sf_conv_pw_stats_floats(ctypes.byref(d))
        if n > 0:  # pointwise layer on the bf16 pipe, statistics from its epilogue
            planes = _weight_planes(wp)
            ws = torch.empty((lib().sf_conv_pw_ws_floats(ctypes.byref(d), 1),), dtype=torch.float32, device=x.buf.device)
            st = torch.empty((n,), dtype=torch.float32, device=x.buf.device)
            parts = ctypes.c_int(0)
            tag = ("conv", d.N * d.To * d.Ho * d.Wo, d.Cin, d.Cout)
            _check(_traced(tag, lambda: lib().sf_conv_fwd_pw(ctypes.byref(d), x.ptr(), _ptr(wp), _ptr(planes), None,
                                                             _ptr(bias), None, out.ptr(), _ptr(ws), _ptr(st),
                                                             ctypes.byref(parts), _stream())), "sf_conv_fwd_pw")
            return out, ((st, parts.value) if parts.value > 0 else None)
    if stats:
        n = lib().sf_conv_stats_ws_floats(ctypes.byref(d)) if (CONV_STATS and scale is None and res is None
                                                              and not relu and out_cmul == 1) else 0
        if n > 0:
            ws = torch.empty((n,), dtype=torch.float32, device=x.buf.device)
            parts = ctypes.c_int(0)
            tag = ("conv", d.N * d.To * d.Ho * d.Wo, d.kT * d.kH * d.kW * d.Cin, d.Cout)
            _check(_traced(tag, lambda: lib().sf_conv_fwd_stats(ctypes.byref(d), x.ptr(), _ptr(wp), None, _ptr(bias),
                                                                None, out.ptr(), _ptr(ws), ctypes.byref(parts),
                                                                _stream())), "sf_conv_fwd_stats")
            return out, ((ws, parts.value) if parts.value > 0 else None)
    _conv_launch(d, x.ptr(), _ptr(wp), _ptr(scale), _ptr(bias), res.ptr() if res is not None else None, out.ptr(),
                 x.buf.device, "sf_conv_fwd", w_tensor=wp, x_act=x if cin == x.C else None, keep=keep)
    return (out, None) if stats else out


def dwconv(x, wp, kernel, stride=(1, 1, 1), padding=(0, 0, 0), scale=None, bias=None, relu=False, res=None,
           out=None, cout=None, out_cmul=1):
    """Depthwise conv (sf_dwconv_fwd); wp = pack_dw_weight(...) [taps, C], or [taps, pitch >= C] (the transposed
    half of pack_conv_weight_pair of the [C, 1, kT, kH, kW] parameter)."""
    _require_gpu(x.buf, "dwconv")
    taps, wpitch = wp.shape
    c = x.C
    assert wpitch >= c, (wp.shape, x)
    cout = c if cout is None else cout
    kT, kH, kW = kernel
    To = _out_dim(x.T, kT, stride[0], padding[0], 1)
    Ho = _out_dim(x.H, kH, stride[1], padding[1], 1)
    Wo = _out_dim(x.W, kW, stride[2], padding[2], 1)
    if out is None:
        out = new_act(x, x.N, To, Ho, Wo, cout)
    else:
        assert (out.N, out.T, out.H, out.W) == (x.N, To, Ho, Wo)
    d = ConvDesc(x.N, x.T, x.H, x.W, c, x.cs, x.coff, To, Ho, Wo, cout, out.cs, out.coff, out_cmul,
                 kT, kH, kW, stride[0], stride[1], stride[2], padding[0], padding[1], padding[2], 1, 1, 1,
                 wpitch, _act(relu),
                 res.cs if res is not None else 0, res.coff if res is not None else 0, 0)
    # trace tag: ("dwconv", algorithmic HBM bytes = input + output (+ residual) rows once)
    nbytes = 4 * (x.rows * c + out.rows * cout * (2 if res is not None else 1))
    _check(_traced(("dwconv", nbytes, taps, c), lambda: lib().sf_dwconv_fwd(
        ctypes.byref(d), x.ptr(), _ptr(wp), _ptr(scale), _ptr(bias), res.ptr() if res is not None else None,
        out.ptr(), _stream())), "sf_dwconv_fwd")
    return out


def pool(x, kernel, stride, padding=(0, 0, 0), avg=False, out=None, out_reserve=(0, 0), want_arg=False):
    """want_arg (max pooling): also return the byte map of window winners for maxpool_bwd_arg — (out, arg), arg None
    when the views are not float4-addressable or the window has more than 255 taps."""
    _require_gpu(x.buf, "pool")
    To = _out_dim(x.T, kernel[0], stride[0], padding[0], 1)
    Ho = _out_dim(x.H, kernel[1], stride[1], padding[1], 1)
    Wo = _out_dim(x.W, kernel[2], stride[2], padding[2], 1)
    if out is None:
        out = new_act(x, x.N, To, Ho, Wo, x.C, out_reserve[0], out_reserve[1])
    else:
        assert (out.N, out.T, out.H, out.W, out.C) == (x.N, To, Ho, Wo, x.C)
    d = PoolDesc(x.N, x.T, x.H, x.W, x.C, x.cs, x.coff, To, Ho, Wo, out.cs, out.coff,
                 kernel[0], kernel[1], kernel[2], stride[0], stride[1], stride[2],
                 padding[0], padding[1], padding[2], 1 if avg else 0)
    if want_arg and not avg:
        ok = (x.C % 4 == 0 and x.cs % 4 == 0 and x.coff % 4 == 0 and out.cs % 4 == 0 and out.coff % 4 == 0 and
              kernel[0] * kernel[1] * kernel[2] <= 255 and x.buf.data_ptr() % 16 == 0 and out.buf.data_ptr() % 16 == 0)
        if ok:
            arg = torch.empty((out.rows, x.C), dtype=torch.uint8, device=x.buf.device)
            _check(lib().sf_maxpool_fwd_arg(ctypes.byref(d), x.ptr(), out.ptr(), _ptr(arg), _stream()),
                   "sf_maxpool_fwd_arg")
            return out, arg
    _check(lib().sf_pool_fwd(ctypes.byref(d), x.ptr(), out.ptr(), _stream()), "sf_pool_fwd")
    return (out, None) if want_arg else out


def tmax_mean(x, alpha):
    """pooled[b,c] = mean_{t',h,w} max_{r<alpha} x[b,t'*alpha+r,h,w,c]  ->  torch [N, C]."""
    _require_gpu(x.buf, "tmax_mean")
    pooled = torch.empty((x.N, x.C), dtype=torch.float32, device=x.buf.device)
    ws = torch.empty((lib().sf_tmax_mean_ws_floats(x.N, x.C),), dtype=torch.float32, device=x.buf.device)
    _check(lib().sf_tmax_mean(x.ptr(), x.cs, x.coff, x.N, x.T, x.H, x.W, x.C, alpha, _ptr(pooled), _ptr(ws),
                              _stream()), "sf_tmax_mean")
    return pooled


def gate_apply(x, alpha, pooled, w3=None, scale=None, bias=None, relu=False, out=None):
    _require_gpu(x.buf, "gate_apply")
    if out is None:
        out = new_act(x, x.N, x.T // alpha, x.H, x.W, x.C)
    else:
        assert (out.N, out.T, out.H, out.W, out.C) == (x.N, x.T // alpha, x.H, x.W, x.C), (out, x)
    _check(lib().sf_gate_apply(x.ptr(), x.cs, x.coff, x.N, x.T, x.H, x.W, x.C, alpha, _ptr(pooled), _ptr(w3),
                               _ptr(scale), _ptr(bias), ACT_RELU if relu else ACT_NONE, out.ptr(), out.cs,
                               out.coff, _stream()), "sf_gate_apply")
    return out


def attention(q, k, v, x, gamma, scale=None, bias=None, relu=False, alpha=1, out=None, save=None):
    """Flash SpatialAttention + gamma-residual + (BN affine, ReLU) + nearest T-upsample x alpha.
    save: optional dict that receives 'o' ([B,N,C] tensor, O = P v) and 'lse' ([B,N]) for the backward pass."""
    _require_gpu(x.buf, "attention")
    if out is None:
        out = new_act(x, x.N, x.T * alpha, x.H, x.W, x.C)
    else:
        assert (out.N, out.T, out.H, out.W, out.C) == (x.N, x.T * alpha, x.H, x.W, x.C), (out, x)

    def base(a):  # slice base pointer: buffer pointer + channel offset
        return ctypes.c_void_p(a.buf.data_ptr() + 4 * a.coff)

    o_save = lse_save = None
    if save is not None:
        n = x.T * x.H * x.W
        o_save = save["o"] = torch.empty((x.N, n, x.C), dtype=torch.float32, device=x.buf.device)
        lse_save = save["lse"] = torch.empty((x.N, n), dtype=torch.float32, device=x.buf.device)
    # workspace for the key-range parts the launcher may cut the sweep into (sf_sweep_parts: fills the last round)
    ws = torch.empty((lib().sf_attn_fwd_ws_floats(x.N, x.T * x.H * x.W, x.C),), dtype=torch.float32,
                     device=x.buf.device)
    _check(_traced(("attn", x.N, x.T * x.H * x.W, x.C), lambda: lib().sf_attn_fwd_ws(
        base(q), q.cs, base(k), k.cs, base(v), v.cs, base(x), x.cs, _ptr(gamma), _ptr(scale),
        _ptr(bias), ACT_RELU if relu else ACT_NONE, out.ptr(), out.cs, out.coff,
        x.N, x.T, x.H, x.W, x.C, alpha, _ptr(o_save), _ptr(lse_save), _ptr(ws), _stream())), "sf_attn_fwd_ws")
    return out


def head_act_mean(logits, act):
    """logits Act [N,T,H,W,K] (dense) -> torch [N, K] = mean_{t,h,w} act(logits)."""
    assert logits.coff == 0 and logits.C == logits.cs
    out = torch.empty((logits.N, logits.C), dtype=torch.float32, device=logits.buf.device)
    _check(lib().sf_head_act_mean(logits.ptr(), logits.N, logits.T * logits.H * logits.W, logits.C, act, _ptr(out),
                                  _stream()), "sf_head_act_mean")
    return out


def copy_channels(x, out, out_cmul=1):
    assert x.rows == out.rows
    _check(lib().sf_copy_channels(x.ptr(), x.cs, x.coff, out.ptr(), out.cs, out.coff, out_cmul, x.rows, x.C,
                                  _stream()), "sf_copy_channels")
    return out


def channel_shuffle(x, out, groups, accumulate=False):
    """out[.., j*G + g] (+)= x[.., g*(C/G) + j] in one launch (sf_channel_shuffle); groups = C/G inverts it."""
    assert x.rows == out.rows and x.C == out.C and x.C % groups == 0
    _check(lib().sf_channel_shuffle(x.ptr(), x.cs, x.coff, out.ptr(), out.cs, out.coff, groups, x.rows, x.C,
                                    1 if accumulate else 0, _stream()), "sf_channel_shuffle")
    return out


def channel_stats(x):
    """Per-channel (mean, biased variance) over all N*T*H*W rows of the view -> two torch [C] tensors."""
    _require_gpu(x.buf, "channel_stats")
    dev = x.buf.device
    mean = torch.empty((x.C,), dtype=torch.float32, device=dev)
    var = torch.empty((x.C,), dtype=torch.float32, device=dev)
    ws = torch.empty((lib().sf_channel_stats_ws_floats(x.C),), dtype=torch.float32, device=dev)
    _check(lib().sf_channel_stats(x.ptr(), x.cs, x.coff, x.rows, x.C, _ptr(mean), _ptr(var), _ptr(ws), _stream()),
           "sf_channel_stats")
    return mean, var


BN_MASK = os.environ.get("SF_BN_MASK", "1") != "0"  # byte masks instead of re-reading activations in BN backward


def affine(x, scale=None, bias=None, res=None, relu=False, rep=1, out=None, out_reserve=(0, 0), out_cmul=1,
           nsplit=1, mask=None):
    """out = act(x*scale + bias + res), repeated `rep` times along T (nearest upsample).  nsplit > 1: scale/bias
    hold nsplit*C entries and sample n uses block n % nsplit (SubBatchNorm3d).
    mask: optional dict (training, ReLU / ReLU6 layers): when the layer qualifies for the flat float4 kernel it
    receives 'bytes', a uint8 tensor [rows * C/4] with one pass-through bit per channel, which bn_bwd(mask=...)
    reads instead of the activation."""
    _require_gpu(x.buf, "affine")
    if out is None:
        out = new_act(x, x.N, x.T * rep, x.H, x.W, x.C, out_reserve[0], out_reserve[1])
    else:
        assert (out.N, out.T, out.H, out.W) == (x.N, x.T * rep, x.H, x.W), (out, x, rep)
        assert out.coff + (x.C - 1) * out_cmul < out.cs and (out_cmul > 1 or out.C == x.C), (out, x)
    if res is not None:
        assert res.rows == x.rows and res.C == x.C
    if (mask is not None and BN_MASK and relu and rep == 1 and nsplit == 1 and out_cmul == 1 and x.C % 4 == 0 and
            x.cs % 4 == 0 and x.coff % 4 == 0 and out.cs % 4 == 0 and out.coff % 4 == 0 and
            x.rows * (x.C // 4) < 2 ** 31 - 1 and scale is not None and
            (res is None or (res.cs % 4 == 0 and res.coff % 4 == 0))):
        mk = torch.empty((x.rows * (x.C // 4),), dtype=torch.uint8, device=x.buf.device)
        _check(lib().sf_affine_fwd_mask(
            x.ptr(), x.cs, x.coff, x.N, x.T, x.H, x.W, x.C, _ptr(scale), _ptr(bias),
            res.ptr() if res is not None else None, res.cs if res is not None else 0,
            res.coff if res is not None else 0, _act(relu), out.ptr(), out.cs, out.coff, _ptr(mk), _stream()),
            "sf_affine_fwd_mask")
        mask["bytes"] = mk
        return out
    tail = (_ptr(scale), _ptr(bias), res.ptr() if res is not None else None, res.cs if res is not None else 0,
            res.coff if res is not None else 0, _act(relu), rep, out.ptr(), out.cs, out.coff, out_cmul, _stream())
    if nsplit == 1:
        _check(lib().sf_affine_fwd(x.ptr(), x.cs, x.coff, x.N, x.T, x.H, x.W, x.C, *tail), "sf_affine_fwd")
    else:
        _check(lib().sf_affine_fwd_split(x.ptr(), x.cs, x.coff, x.N, x.T, x.H, x.W, x.C, nsplit, *tail),
               "sf_affine_fwd_split")
    return out


def conv_dgrad(dz, wt_packed, x_like, kernel, stride=(1, 1, 1), padding=(0, 0, 0), dilation=(1, 1, 1),
               out=None, accumulate=False, dz_planes=None):
    """Data gradient of a dense conv: dL/dx[N,Ti,Hi,Wi,Cin] (+)= conv^T(dL/dz, W).  `wt_packed` =
    pack_conv_weight(W.transpose(0, 1)) i.e. [Cin][tap][Cout_pad]; `x_like` gives the forward input's dims."""
    _require_gpu(dz.buf, "conv_dgrad")
    cin, taps, cout_pad = wt_packed.shape
    if out is None:
        out = new_act(dz, x_like.N, x_like.T, x_like.H, x_like.W, cin)
        accumulate = False
    assert (out.N, out.T, out.H, out.W, out.C) == (x_like.N, x_like.T, x_like.H, x_like.W, cin), (out, x_like)
    if max(stride) > 1 and tuple(dilation) == (1, 1, 1) and (accumulate or (out.coff == 0 and out.cs == out.C)):
        return _conv_dgrad_strided(dz, wt_packed, out, kernel, stride, padding, accumulate, dz_planes)
    d = ConvDesc(dz.N, dz.T, dz.H, dz.W, dz.C, dz.cs, dz.coff, out.T, out.H, out.W, cin, out.cs, out.coff, 1,
                 kernel[0], kernel[1], kernel[2], stride[0], stride[1], stride[2], padding[0], padding[1],
                 padding[2], dilation[0], dilation[1], dilation[2], cout_pad, ACT_NONE,
                 out.cs if accumulate else 0, out.coff if accumulate else 0, 1)
    _conv_launch(d, dz.ptr(), _ptr(wt_packed), None, None, out.ptr() if accumulate else None, out.ptr(),
                 dz.buf.device, "sf_conv_fwd(transposed)", w_tensor=wt_packed, in_planes=dz_planes)
    return out



def _residue_taps(k, s, p, a):
    """Taps kk of a stride-s, pad-p, size-k kernel that reach input positions = a (mod s), as (offsets, taps):
    input position s*i + a receives dz[i + off] through tap kk, off = (a + p - kk) / s; returned in ascending
    off order so that they form a dense stride-1 kernel over dz."""
    pairs = sorted(((a + p - kk) // s, kk) for kk in range(k) if (a + p - kk) % s == 0)
    return [o for o, _ in pairs], [kk for _, kk in pairs]


_TAP_INDEX = {}


def _class_weights(wt_packed, store, kernel, stride, padding):
    """The tap-subset copies of a packed data-gradient weight [Cin][taps][cout_pad] for EVERY residue class of a strided
    layer, by ONE gather launch per weight and optimizer step: the rows (ci, tap) of all classes are gathered class by
    class into one tensor, and a class's weight [Cin][its taps][cout_pad] is a contiguous slice of it (a gather per
    class was 30 small launches per step of cfg #3, each in front of its class's conv; a torch.stack of tap slices ~190
    copy kernels).  The row index is built once per (Cin, layer geometry, device) — during warm-up, so nothing is copied
    from the host inside a hipGraph capture.  Fills `store` (key = (a_t, a_h, a_w, kernel, stride, padding))."""
    cin, taps, cout_pad = wt_packed.shape
    kT, kH, kW = kernel
    fam = (tuple(kernel), tuple(stride), tuple(padding))
    classes = []
    for at in range(stride[0]):
        tt = _residue_taps(kT, stride[0], padding[0], at)[1]
        for ah in range(stride[1]):
            th = _residue_taps(kH, stride[1], padding[1], ah)[1]
            for aw in range(stride[2]):
                tw = _residue_taps(kW, stride[2], padding[2], aw)[1]
                if tt and th and tw:
                    classes.append(((at, ah, aw) + fam, [(a * kH + b) * kW + c for a in tt for b in th for c in tw]))
    ikey = (cin, taps, fam, str(wt_packed.device))
    idx = _TAP_INDEX.get(ikey)
    if idx is None:
        rows = [ci * taps + t for _, sel in classes for ci in range(cin) for t in sel]
        idx = _TAP_INDEX[ikey] = torch.tensor(rows, dtype=torch.long, device=wt_packed.device)
    allrows = wt_packed.view(cin * taps, cout_pad).index_select(0, idx)
    off = 0
    for key, sel in classes:
        n = cin * len(sel)
        store[key] = allrows[off:off + n].view(cin, len(sel), cout_pad)
        off += n
    return store


def _conv_dgrad_strided(dz, wt_packed, out, kernel, stride, padding, accumulate, dz_planes=None):
    """Data gradient of a strided conv as one DENSE small conv over dL/dz per residue class of the input position
    (the transposed-gather formulation evaluates every tap at every input position and predicates s^2-1 of s^2
    of them away).  Class (a_t, a_h, a_w): dx[s*i + a] = sum_j dz[i - pad' + j] * W[tap_j]; stores are scattered
    to the class's positions by the conv kernel's epilogue."""
    cin, _, cout_pad = wt_packed.shape
    kT, kH, kW = kernel
    # without accumulation every class WRITES its own positions; only when some class gets no tap at all (1x1x1
    # stride-2 shortcuts: 3 of 4) does the buffer need zeros underneath
    every = all(_residue_taps(k, s, p, a)[1] and (full - a + s - 1) // s > 0
                for k, s, p, full in zip(kernel, stride, padding, (out.T, out.H, out.W)) for a in range(s))
    add = accumulate or not every
    if not accumulate and not every:
        out.buf.zero_()
    for at in range(stride[0]):
        ot, tt = _residue_taps(kT, stride[0], padding[0], at)
        for ah in range(stride[1]):
            oh, th = _residue_taps(kH, stride[1], padding[1], ah)
            for aw in range(stride[2]):
                ow, tw = _residue_taps(kW, stride[2], padding[2], aw)
                dims = [(full - a + s - 1) // s for full, a, s in zip((out.T, out.H, out.W), (at, ah, aw), stride)]
                if not (tt and th and tw) or min(dims) <= 0:
                    continue  # no tap reaches this class: its gradient is zero
                # tap-subset weights, cached ON the packed tensor (they must die with it: the next optimizer step's
                # packed copy may land at the same address)
                store = wt_packed.__dict__.setdefault("_sf_classes", {})
                key = (at, ah, aw, tuple(kernel), tuple(stride), tuple(padding))
                wsub = store.get(key)
                if wsub is None and os.environ.get("SF_CLASS_GATHER", "1") == "0":
                    taps_sel = tuple((a * kH + b) * kW + c for a in tt for b in th for c in tw)
                    ikey = (taps_sel, str(wt_packed.device))
                    idx = _TAP_INDEX.get(ikey)
                    if idx is None:
                        idx = _TAP_INDEX[ikey] = torch.tensor(taps_sel, dtype=torch.long, device=wt_packed.device)
                    wsub = store[key] = wt_packed.index_select(1, idx)
                if wsub is None:
                    wsub = _class_weights(wt_packed, store, kernel, stride, padding)[key]
                d = ConvDesc(dz.N, dz.T, dz.H, dz.W, dz.C, dz.cs, dz.coff, dims[0], dims[1], dims[2], cin, out.cs,
                             out.coff, 1, len(tt), len(th), len(tw), 1, 1, 1, -ot[0], -oh[0], -ow[0], 1, 1, 1,
                             cout_pad, ACT_NONE, out.cs, out.coff, 0, stride[0], stride[1], stride[2], at, ah, aw,
                             out.T, out.H, out.W)
                _conv_launch(d, dz.ptr(), _ptr(wsub), None, None, out.ptr() if add else None, out.ptr(),
                             dz.buf.device, "sf_conv_fwd(strided dgrad class)", w_tensor=wsub, in_planes=dz_planes)
    return out


# ------------------------------------------------------------------------------------------------ backward
# SF_CONV_BX_AF32 (default 1, read by conv_bx.hip too): forward / data-gradient launches of conv_bx.hip take their
# activation operand as fp32 rows and split it after the LDS fragment read — no activation planes on those paths; the
# weight gradient (whose transposing LDS reads need 16-bit elements) makes the planes of x and dz itself, inside its
# own launch sequence on the companion stream.  0: planes made once per tensor on the pathway's stream and shared.
BX_AF32 = os.environ.get("SF_CONV_BX_AF32", "1") != "0"


def bx_backward_wants_dz_planes(x, dz, cout, kernel, stride, padding, dilation, cin=None, cin_pad=None,
                                dgrad=True):
    """True when the layer's weight gradient or (stride-1) data gradient runs on conv_bx.hip AND takes planes from the
    caller (SF_CONV_BX_AF32=0): the caller then makes dz's planes ONCE (act_planes) and hands them to both."""
    if BX_AF32:
        return False
    cin = x.C if cin is None else cin
    cin_pad = (cin + 15) // 16 * 16 if cin_pad is None else cin_pad
    if cin_pad != cin or not SPLIT_K:
        return False
    kT, kH, kW = kernel
    d = ConvDesc(x.N, x.T, x.H, x.W, cin, x.cs, x.coff, dz.T, dz.H, dz.W, cout, 0, 0, 1,
                 kT, kH, kW, stride[0], stride[1], stride[2], padding[0], padding[1], padding[2],
                 dilation[0], dilation[1], dilation[2], cin_pad, ACT_NONE, 0, 0, 0)
    if lib().sf_conv_wgrad_bx_ws_floats(ctypes.byref(d), 1, 1) > 0:
        return True
    if dgrad and max(stride) == 1 and cout % 16 == 0:
        dt = ConvDesc(dz.N, dz.T, dz.H, dz.W, cout, dz.cs, dz.coff, x.T, x.H, x.W, cin, cin, 0, 1,
                      kT, kH, kW, 1, 1, 1, padding[0], padding[1], padding[2], dilation[0], dilation[1], dilation[2],
                      cout, ACT_NONE, 0, 0, 1)
        return lib().sf_conv_bx_ws_floats(ctypes.byref(dt), 1, 1) > 0
    return False


def conv_wgrad(x, dz, cout, kernel, stride=(1, 1, 1), padding=(0, 0, 0), dilation=(1, 1, 1), cin=None,
               cin_pad=None, finish_into=None, x_planes=None, dz_planes=None):
    """dW packed [Cout, taps, cin_pad] = sum over positions of dz (x) x (split partials summed in fixed order).
    finish_into=(dst, Cin, fold_kw): instead of returning the packed gradient, sum the partials and ACCUMULATE
    them into dst, a contiguous tensor in nn.Conv3d's [Cout, Cin, kT, kH, kW] layout (e.g. the weight's .grad)."""
    _require_gpu(x.buf, "conv_wgrad")
    cin = x.C if cin is None else cin
    cin_pad = (cin + 15) // 16 * 16 if cin_pad is None else cin_pad
    kT, kH, kW = kernel
    d = ConvDesc(x.N, x.T, x.H, x.W, cin, x.cs, x.coff, dz.T, dz.H, dz.W, cout, 0, 0, 1,
                 kT, kH, kW, stride[0], stride[1], stride[2], padding[0], padding[1], padding[2],
                 dilation[0], dilation[1], dilation[2], cin_pad, ACT_NONE, 0, 0, 0)
    nws = lib().sf_conv_wgrad_bx_ws_floats(ctypes.byref(d), 1 if x_planes is not None else 0,
                                           1 if dz_planes is not None else 0) if cin_pad == cin else 0
    if nws > 0:  # long reductions: the bf16-piece kernel of conv_bx.hip (operand planes handed in or made in ws)
        S = lib().sf_conv_wgrad_bx_splits(ctypes.byref(d))
        part = torch.empty((S, cout, kT * kH * kW, cin_pad), dtype=torch.float32, device=x.buf.device)
        ws = torch.empty((nws,), dtype=torch.float32, device=x.buf.device)
        rc = _traced(("conv", dz.rows, kT * kH * kW * cin, cout), lambda: lib().sf_conv_wgrad_bx(
            ctypes.byref(d), x.ptr(), _ptr(x_planes), dz.ptr(), dz.cs, dz.coff, _ptr(dz_planes), _ptr(part), _ptr(ws),
            _stream()))
        if rc in (SF_EALIGN, SF_ENOTTAKEN):  # a dz view the shape-only plan cannot see (odd channel offset / pitch)
            nws = 0
        else:
            _check(rc, "sf_conv_wgrad_bx")
    if nws <= 0:
        S = lib().sf_conv_wgrad_splits(ctypes.byref(d))
        part = torch.empty((S, cout, kT * kH * kW, cin_pad), dtype=torch.float32, device=x.buf.device)
        _check(_traced(("conv", dz.rows, kT * kH * kW * cin, cout), lambda: lib().sf_conv_wgrad(
            ctypes.byref(d), x.ptr(), dz.ptr(), dz.cs, dz.coff, _ptr(part), _stream())), "sf_conv_wgrad")
    if finish_into is not None:
        dst, real_cin, fold_kw = finish_into
        assert dst.is_contiguous() and dst.dtype == torch.float32
        if S > 128 and cout * kT * kH * kW * cin_pad > 65536:  # long split lists of large weights (does not occur
            part, S = part.sum(0, keepdim=True), 1  # in the shipped models): a parallel tree sum first
        assert dst.numel() == cout * real_cin * kT * kH * kW * max(fold_kw, 1), (dst.shape, cout, real_cin, kernel)
        _check(lib().sf_conv_wgrad_finish(_ptr(part), S, cout, kT * kH * kW, cin_pad, real_cin, fold_kw, _ptr(dst), 1,
                                          _stream()), "sf_conv_wgrad_finish")
        return None
    return part.sum(0) if S > 1 else part[0]


# ------------------------------------------------------------------------------------------------ grouped convs
def conv_grouped(x, wp, groups, kernel, stride=(1, 1, 1), padding=(0, 0, 0), dilation=(1, 1, 1), scale=None,
                 bias=None, relu=False, res=None, out=None, shuffle=False, out_reserve=(0, 0)):
    """nn.Conv3d(groups = G), 1 < G < channels, as ONE launch (sf_conv_fwd_grouped: group = grid z of the LDS-tiled
    kernel).  wp = the packed weight [Cout][taps][pad(Cin / G)] (pack_conv_weight_pair of the grouped parameter: the G
    per-group packs one after the other).  shuffle: group g's channel j is stored at channel j*G + g."""
    _require_gpu(x.buf, "conv_grouped")
    cout, taps, cin_pad = wp.shape
    kT, kH, kW = kernel
    To = _out_dim(x.T, kT, stride[0], padding[0], dilation[0])
    Ho = _out_dim(x.H, kH, stride[1], padding[1], dilation[1])
    Wo = _out_dim(x.W, kW, stride[2], padding[2], dilation[2])
    if out is None:
        out = new_act(x, x.N, To, Ho, Wo, cout, out_reserve[0], out_reserve[1])
    assert (out.N, out.T, out.H, out.W, out.C) == (x.N, To, Ho, Wo, cout), (out, (x.N, To, Ho, Wo, cout))
    assert x.C % groups == 0 and cout % groups == 0 and cin_pad >= x.C // groups, (x, wp.shape, groups)
    d = ConvDesc(x.N, x.T, x.H, x.W, x.C, x.cs, x.coff, To, Ho, Wo, cout, out.cs, out.coff, 1,
                 kT, kH, kW, stride[0], stride[1], stride[2], padding[0], padding[1], padding[2],
                 dilation[0], dilation[1], dilation[2], cin_pad, _act(relu),
                 res.cs if res is not None else 0, res.coff if res is not None else 0, 0)
    if res is not None:
        assert res.rows == out.rows and res.C == cout
    tag = ("conv", d.N * d.To * d.Ho * d.Wo, taps * (x.C // groups), cout)
    _check(_traced(tag, lambda: lib().sf_conv_fwd_grouped(
        ctypes.byref(d), groups, 1 if shuffle else 0, x.ptr(), _ptr(wp), _ptr(scale), _ptr(bias),
        res.ptr() if res is not None else None, out.ptr(), _stream())), "sf_conv_fwd_grouped")
    return out


def conv_dgrad_grouped(dz, wtp, groups, x_like, kernel, stride=(1, 1, 1), padding=(0, 0, 0), dilation=(1, 1, 1),
                       out=None, accumulate=False):
    """Data gradient of a grouped conv in ONE launch: wtp = [Cin][taps][pad(Cout / G)], the G transposed per-group
    packs one after the other (group g: rows [g*Cin/G, (g+1)*Cin/G), reading dz channels [g*Cout/G, (g+1)*Cout/G)).
    Strided layers run the transposed gather with its taps predicated (no residue-class launches)."""
    _require_gpu(dz.buf, "conv_dgrad_grouped")
    cin, taps, cout_pad = wtp.shape
    if out is None:
        out = new_act(dz, x_like.N, x_like.T, x_like.H, x_like.W, cin)
        accumulate = False
    assert (out.N, out.T, out.H, out.W, out.C) == (x_like.N, x_like.T, x_like.H, x_like.W, cin), (out, x_like)
    assert dz.C % groups == 0 and cin % groups == 0 and cout_pad >= dz.C // groups
    d = ConvDesc(dz.N, dz.T, dz.H, dz.W, dz.C, dz.cs, dz.coff, out.T, out.H, out.W, cin, out.cs, out.coff, 1,
                 kernel[0], kernel[1], kernel[2], stride[0], stride[1], stride[2], padding[0], padding[1],
                 padding[2], dilation[0], dilation[1], dilation[2], cout_pad, ACT_NONE,
                 out.cs if accumulate else 0, out.coff if accumulate else 0, 1)
    tag = ("conv", dz.rows, taps * (dz.C // groups), cin)
    _check(_traced(tag, lambda: lib().sf_conv_fwd_grouped(
        ctypes.byref(d), groups, 0, dz.ptr(), _ptr(wtp), None, None, out.ptr() if accumulate else None, out.ptr(),
        _stream())), "sf_conv_fwd_grouped(transposed)")
    return out


def conv_wgrad_grouped(x, dz, groups, kernel, stride=(1, 1, 1), padding=(0, 0, 0), dilation=(1, 1, 1),
                       cin_pad=None, finish_into=None):
    """Weight gradient of a grouped conv in ONE launch (+ the finish): packed [Cout][taps][cin_pad] with cin_pad =
    pad(Cin / G), or accumulated into finish_into = a contiguous tensor in nn.Conv3d's grouped layout
    [Cout][Cin / G][kT][kH][kW] (the parameter's gradient)."""
    _require_gpu(x.buf, "conv_wgrad_grouped")
    cin_g, cout = x.C // groups, dz.C
    assert x.C % groups == 0 and cout % groups == 0
    cin_pad = (cin_g + 15) // 16 * 16 if cin_pad is None else cin_pad
    kT, kH, kW = kernel
    d = ConvDesc(x.N, x.T, x.H, x.W, x.C, x.cs, x.coff, dz.T, dz.H, dz.W, cout, 0, 0, 1,
                 kT, kH, kW, stride[0], stride[1], stride[2], padding[0], padding[1], padding[2],
                 dilation[0], dilation[1], dilation[2], cin_pad, ACT_NONE, 0, 0, 0)
    S = lib().sf_conv_wgrad_grouped_splits(ctypes.byref(d), groups)
    if S <= 0:
        raise SfhipError("sf_conv_wgrad_grouped_splits: %d -> %d channels in %d groups" % (x.C, cout, groups))
    part = torch.empty((S, cout, kT * kH * kW, cin_pad), dtype=torch.float32, device=x.buf.device)
    _check(_traced(("conv", dz.rows, kT * kH * kW * cin_g, cout), lambda: lib().sf_conv_wgrad_grouped(
        ctypes.byref(d), groups, x.ptr(), dz.ptr(), dz.cs, dz.coff, _ptr(part), _stream())), "sf_conv_wgrad_grouped")
    if finish_into is not None:
        dst = finish_into
        assert dst.is_contiguous() and dst.dtype == torch.float32 and dst.numel() == cout * cin_g * kT * kH * kW
        _check(lib().sf_conv_wgrad_finish(_ptr(part), S, cout, kT * kH * kW, cin_pad, cin_g, 0, _ptr(dst), 1,
                                          _stream()), "sf_conv_wgrad_finish")
        return None
    return part.sum(0) if S > 1 else part[0]


def unpack_conv_weight_grad(dwp, shape):
    """[Cout, taps, cin_pad] -> [Cout, Cin, kT, kH, kW]."""
    cout, cin, kT, kH, kW = shape
    return dwp[:, :, :cin].permute(0, 2, 1).reshape(cout, cin, kT, kH, kW).contiguous()


def bn_bwd(dy, y, z, mean, invstd, gamma, relu, rep=1, dres=None, dz_out=None, dgamma_out=None, nsplit=1,
           sync=None, grad_sink=None, mask=None, dres_overwrite=False):
    """Training BN backward (+ReLU mask, + residual fan-out, + upsample-copy sum).  Returns (dz, dgamma, dbeta);
    dz is written over z unless dz_out is given.
    nsplit > 1 (SubBatchNorm3d): mean/invstd/gamma and the returned sums hold nsplit*C entries [split*C + c].
    sync (NaiveSyncBatchNorm3d): callable (dbeta, dgamma) -> (dbeta', dgamma') applied between the reduction and
    the normalisation pass (the cross-rank sum, already divided by the number of ranks); the LOCAL sums are
    returned for the parameter gradients."""
    C = z.C
    dev = z.buf.device
    mean, invstd, gamma = mean.contiguous(), invstd.contiguous(), gamma.detach().contiguous()
    dbeta = torch.empty((nsplit * C,), dtype=torch.float32, device=dev)
    dgamma = torch.empty((nsplit * C,), dtype=torch.float32, device=dev)
    ws = torch.empty((lib().sf_bn_bwd_ws_floats(C),), dtype=torch.float32, device=dev)
    yp, ycs, yco = (y.ptr(), y.cs, y.coff) if y is not None else (None, 0, 0)
    code = 2 if relu == 6 else (1 if relu else 0)
    if mask is not None and relu:  # the byte mask affine(mask=...) left: read instead of the activation
        assert rep == 1 and nsplit == 1 and mask.numel() == z.rows * (C // 4)
        yp, ycs, yco, code = _ptr(mask), C // 4, 0, 3
    head = (dy.ptr(), dy.cs, dy.coff, yp, ycs, yco, z.ptr(), z.cs, z.coff, z.N, z.T, z.H, z.W, C)
    tail = (rep, code, _ptr(mean), _ptr(invstd))
    split = (nsplit,) if nsplit > 1 else ()
    reduce_fn = lib().sf_bn_bwd_reduce_split if nsplit > 1 else lib().sf_bn_bwd_reduce
    apply_fn = lib().sf_bn_bwd_apply_split if nsplit > 1 else lib().sf_bn_bwd_apply
    if dres_overwrite:  # dres is an uninitialised buffer this call is the first writer of (dres = g, not +=)
        assert dres is not None and nsplit == 1
        apply_fn = lib().sf_bn_bwd_apply_first
    if grad_sink is not None:  # (weight.grad[:C], bias.grad[:C]) accumulated by the reduction's final kernel
        assert nsplit == 1 and dgamma_out is None
        _check(lib().sf_bn_bwd_reduce_acc(*head, *tail, _ptr(dbeta), _ptr(dgamma), _ptr(ws), _ptr(grad_sink[1]),
                                          _ptr(grad_sink[0]), _stream()), "sf_bn_bwd_reduce_acc")
    else:
        _check(reduce_fn(*head, *split, *tail, _ptr(dbeta), _ptr(dgamma), _ptr(ws), _stream()), "sf_bn_bwd_reduce")
    db_apply, dg_apply = (dbeta, dgamma) if sync is None else sync(dbeta, dgamma)
    out = z if dz_out is None else dz_out
    _check(apply_fn(*head, *split, *tail, _ptr(gamma), _ptr(db_apply), _ptr(dg_apply), out.ptr(), out.cs, out.coff,
                    dres.ptr() if dres is not None else None, dres.cs if dres is not None else 0,
                    dres.coff if dres is not None else 0, _stream()), "sf_bn_bwd_apply")
    if dgamma_out is not None:  # scatter into full-width parameter gradients (sliced BN, GhostModule)
        if nsplit > 1:
            dgamma_out[0][:C] = dgamma.view(nsplit, C).sum(0)
            dgamma_out[1][:C] = dbeta.view(nsplit, C).sum(0)
        else:
            dgamma_out[0][:C] = dgamma
            dgamma_out[1][:C] = dbeta
    return out, dgamma, dbeta


def maxpool_bwd(x, y, dy, dx, kernel, stride, padding=(0, 0, 0), overwrite=False):
    """dx (+)= gathered dy; overwrite: dx is an uninitialised buffer this call is the first writer of."""
    d = PoolDesc(x.N, x.T, x.H, x.W, x.C, x.cs, x.coff, y.T, y.H, y.W, y.cs, y.coff,
                 kernel[0], kernel[1], kernel[2], stride[0], stride[1], stride[2],
                 padding[0], padding[1], padding[2], 0)
    fn = lib().sf_maxpool_bwd_first if overwrite else lib().sf_maxpool_bwd
    _check(fn(ctypes.byref(d), x.ptr(), y.ptr(), dy.ptr(), dy.cs, dy.coff, dx.ptr(), dx.cs, dx.coff, _stream()),
           "sf_maxpool_bwd")
    return dx


def maxpool_bwd_arg(x_like, arg, dy, dx, kernel, stride, padding=(0, 0, 0), overwrite=False):
    """dx (+)= dy gathered through the winner map `arg` of pool(..., want_arg=True); x_like gives the input dims.
    Returns False (nothing launched) when the gradient views are not float4-addressable."""
    if not (dy.cs % 4 == 0 and dy.coff % 4 == 0 and dx.cs % 4 == 0 and dx.coff % 4 == 0 and
            dy.buf.data_ptr() % 16 == 0 and dx.buf.data_ptr() % 16 == 0):
        return False
    d = PoolDesc(x_like.N, x_like.T, x_like.H, x_like.W, x_like.C, x_like.cs, x_like.coff, dy.T, dy.H, dy.W, dy.cs,
                 dy.coff, kernel[0], kernel[1], kernel[2], stride[0], stride[1], stride[2],
                 padding[0], padding[1], padding[2], 0)
    _check(lib().sf_maxpool_bwd_arg(ctypes.byref(d), _ptr(arg), dy.ptr(), dy.cs, dy.coff, dx.ptr(), dx.cs, dx.coff,
                                    1 if overwrite else 0, _stream()), "sf_maxpool_bwd_arg")
    return True


def tmax_dot(x, alpha, dz):
    out = torch.empty((x.N, x.C), dtype=torch.float32, device=x.buf.device)
    ws = torch.empty((lib().sf_tmax_mean_ws_floats(x.N, x.C),), dtype=torch.float32, device=x.buf.device)
    _check(lib().sf_tmax_dot(x.ptr(), x.cs, x.coff, x.N, x.T, x.H, x.W, x.C, alpha, dz.ptr(), dz.cs, dz.coff,
                             _ptr(out), _ptr(ws), _stream()), "sf_tmax_dot")
    return out


def eca_bwd_apply(x, alpha, dz, gate, dpool, dx):
    _check(lib().sf_eca_bwd_apply(x.ptr(), x.cs, x.coff, x.N, x.T, x.H, x.W, x.C, alpha, dz.ptr(), dz.cs, dz.coff,
                                  _ptr(gate), _ptr(dpool), dx.ptr(), dx.cs, dx.coff, _stream()), "sf_eca_bwd_apply")
    return dx


def eca_gate_bwd(dg, pooled, w3, dpool_scale, dw3):
    """(gate, dpool) of the ECA gate's backward on [N, C] vectors; dw3 (3 floats) is accumulated in place."""
    n, c = pooled.shape
    gate = torch.empty_like(pooled)
    dpool = torch.empty_like(pooled)
    _check(lib().sf_eca_gate_bwd(_ptr(dg), _ptr(pooled), _ptr(w3), n, c, float(dpool_scale), _ptr(gate), _ptr(dpool),
                                 _ptr(dw3), _stream()), "sf_eca_gate_bwd")
    return gate, dpool


def bcast_add(g, v, scale):
    _check(lib().sf_bcast_add(g.ptr(), g.cs, g.coff, g.N, g.T * g.H * g.W, g.C, _ptr(v), float(scale), _stream()),
           "sf_bcast_add")
    return g


def rowdot(a, b, scale=1.0):
    out = torch.empty((a.rows,), dtype=torch.float32, device=a.buf.device)
    _check(lib().sf_rowdot(a.ptr(), a.cs, a.coff, b.ptr(), b.cs, b.coff, a.rows, a.C, float(scale), _ptr(out),
                           _stream()), "sf_rowdot")
    return out


def axpy(a, out, alpha=1.0, accumulate=True):
    assert a.rows == out.rows and a.C == out.C
    _check(lib().sf_axpy(a.ptr(), a.cs, a.coff, float(alpha), out.ptr(), out.cs, out.coff, a.rows, a.C,
                         1 if accumulate else 0, _stream()), "sf_axpy")
    return out


def row_softmax(x, scale=1.0):
    """x <- softmax(scale * x) over the channel dimension, in place (one wavefront per row)."""
    _require_gpu(x.buf, "row_softmax")
    _check(lib().sf_row_softmax_fwd(x.ptr(), x.cs, x.coff, x.rows, x.C, float(scale), _stream()), "sf_row_softmax_fwd")
    return x


def row_softmax_bwd(p, dp, scale=1.0):
    """dp <- scale * p * (dp - <p, dp>) in place."""
    assert p.rows == dp.rows and p.C == dp.C
    _check(lib().sf_row_softmax_bwd(p.ptr(), p.cs, p.coff, dp.ptr(), dp.cs, dp.coff, p.rows, p.C, float(scale),
                                    _stream()), "sf_row_softmax_bwd")
    return dp


def act_bwd(dy, y, relu, dx, accumulate=True):
    """dx (+)= dy * [0 < y (< 6)] — backward of a bare ReLU (relu=True) / ReLU6 (relu=6)."""
    assert dy.rows == y.rows == dx.rows and dy.C == y.C == dx.C
    _check(lib().sf_act_bwd(dy.ptr(), dy.cs, dy.coff, y.ptr(), y.cs, y.coff, _act(relu), dx.ptr(), dx.cs, dx.coff,
                            dy.rows, dy.C, 1 if accumulate else 0, _stream()), "sf_act_bwd")
    return dx


FUSED_ATTN_BWD = os.environ.get("SF_ATTN_BWD_SPLIT") != "1"  # single-sweep backward for 16 < C <= 64 (5 GB scratch at
# N = 25088); SF_ATTN_BWD_SPLIT=1 selects the two-kernel form (no scratch)


def attention_bwd(q, k, v, dz, o, lse, gamma, dq, dk, dv):
    """dq/dk/dv (Act slices, overwritten) of the flash SpatialAttention; returns dvec[i] = <dz_i, O_i>
    (its sum is dL/dgamma)."""
    B, n, C = o.shape
    o_act = Act(o.view(B, 1, 1, n, C))
    dvec = rowdot(Act(dz.buf.view(B, 1, 1, n, dz.cs), dz.coff, C), o_act)

    def base(a):
        return ctypes.c_void_p(a.buf.data_ptr() + 4 * a.coff)

    nws = lib().sf_attn_bwd_fused_ws_floats(B, n, C) if FUSED_ATTN_BWD else 0
    if nws > 0:  # one sweep: S and dP are recomputed once (5 products), dQ summed over key-block planes
        ws = torch.empty((nws,), dtype=torch.float32, device=o.device)
        _check(_traced(("attn_bwd_fused", B, n, C), lambda: lib().sf_attn_bwd_fused(
            base(q), q.cs, base(k), k.cs, base(v), v.cs, base(dz), dz.cs, _ptr(lse), _ptr(dvec), _ptr(gamma),
            base(dq), dq.cs, base(dk), dk.cs, base(dv), dv.cs, B, n, C, _ptr(ws), _stream())), "sf_attn_bwd_fused")
        return dvec
    for which, tag in ((1, "attn_bwd_dq"), (2, "attn_bwd_dkv")):
        _check(_traced((tag, B, n, C), lambda: lib().sf_attn_bwd(
            base(q), q.cs, base(k), k.cs, base(v), v.cs, base(dz), dz.cs, _ptr(lse), _ptr(dvec),
            _ptr(gamma), base(dq), dq.cs, base(dk), dk.cs, base(dv), dv.cs, B, n, C, which, _stream())),
            "sf_attn_bwd")
    return dvec


def bn_train_stats(x, gamma, beta, eps, momentum, run_mean, run_var, nsplit=1):
    """Training BN statistics of the view + scale/shift + in-place running-stat update in two launches.
    Returns (mean, invstd, scale, shift).  nsplit > 1 (SubBatchNorm3d): sample n belongs to split n % nsplit and
    every array has nsplit*C entries laid out [split*C + c]."""
    _require_gpu(x.buf, "bn_train_stats")
    dev = x.buf.device
    o = torch.empty((5, nsplit * x.C), dtype=torch.float32, device=dev)
    ws = torch.empty((lib().sf_channel_stats_ws_floats(x.C),), dtype=torch.float32, device=dev)
    if nsplit == 1:
        _check(lib().sf_bn_train_stats(x.ptr(), x.cs, x.coff, x.rows, x.C, _ptr(gamma), _ptr(beta), float(eps),
                                       float(momentum), _ptr(run_mean), _ptr(run_var), _ptr(o[0]), _ptr(o[1]),
                                       _ptr(o[2]), _ptr(o[3]), _ptr(o[4]), _ptr(ws), _stream()), "sf_bn_train_stats")
    else:
        _check(lib().sf_bn_train_stats_split(
            x.ptr(), x.cs, x.coff, x.N, x.T * x.H * x.W, x.C, nsplit, _ptr(gamma), _ptr(beta), float(eps),
            float(momentum), _ptr(run_mean), _ptr(run_var), _ptr(o[0]), _ptr(o[1]), _ptr(o[2]), _ptr(o[3]),
            _ptr(o[4]), _ptr(ws), _stream()), "sf_bn_train_stats_split")
    return o[0], o[2], o[3], o[4]


def bn_train_stats_merge(parts, C, gamma, beta, eps, momentum, run_mean, run_var):
    """bn_train_stats from the per-tile rows a conv epilogue left (conv(..., stats=True)): one small launch, no pass
    over the activation.  Returns (mean, invstd, scale, shift)."""
    ws, rows = parts
    o = torch.empty((5, C), dtype=torch.float32, device=ws.device)
    _check(lib().sf_bn_train_stats_merge(_ptr(ws), rows, C, _ptr(gamma), _ptr(beta), float(eps), float(momentum),
                                         _ptr(run_mean), _ptr(run_var), _ptr(o[0]), _ptr(o[1]), _ptr(o[2]),
                                         _ptr(o[3]), _ptr(o[4]), _stream()), "sf_bn_train_stats_merge")
    return o[0], o[2], o[3], o[4]


def _dw_desc(x, dz, kernel, stride, padding, wpitch=None):
    return ConvDesc(x.N, x.T, x.H, x.W, x.C, x.cs, x.coff, dz.T, dz.H, dz.W, x.C, 0, 0, 1,
                    kernel[0], kernel[1], kernel[2], stride[0], stride[1], stride[2], padding[0], padding[1],
                    padding[2], 1, 1, 1, x.C if wpitch is None else wpitch, ACT_NONE, 0, 0, 0)


def dwconv_bwd(x, dz, wp, kernel, stride, padding, dx=None, into=None):
    """Depthwise conv backward: returns dw [taps, C]; accumulates the data gradient into dx (if given).
    into: a contiguous tensor in the parameter's own layout [C, 1, kT, kH, kW] (its .grad) — the weight gradient is
    ACCUMULATED there by the reduction's final step (sf_dwconv_wgrad_param) and None is returned."""
    d = _dw_desc(x, dz, kernel, stride, padding, wp.shape[1])
    C = x.C
    ws = torch.empty((lib().sf_dwconv_wgrad_ws_floats(ctypes.byref(d), C),), dtype=torch.float32,
                     device=x.buf.device)
    if into is not None:
        assert into.is_contiguous() and into.dtype == torch.float32 and \
            into.numel() == C * kernel[0] * kernel[1] * kernel[2], (into.shape, C, kernel)
        dw = None
        _check(lib().sf_dwconv_wgrad_param(ctypes.byref(d), x.ptr(), dz.ptr(), dz.cs, dz.coff, C, _ptr(into), 1,
                                           _ptr(ws), _stream()), "sf_dwconv_wgrad_param")
    else:
        dw = torch.empty((kernel[0] * kernel[1] * kernel[2], C), dtype=torch.float32, device=x.buf.device)
        _check(lib().sf_dwconv_wgrad(ctypes.byref(d), x.ptr(), dz.ptr(), dz.cs, dz.coff, C, _ptr(dw), _ptr(ws),
                                     _stream()), "sf_dwconv_wgrad")
    if dx is not None:
        _check(lib().sf_dwconv_dgrad(ctypes.byref(d), dz.ptr(), dz.cs, dz.coff, _ptr(wp), dx.ptr(), dx.cs, dx.coff,
                                     C, _stream()), "sf_dwconv_dgrad")
    return dw


def dwconv_dgrad(x, dz, wp, kernel, stride, padding, dx):
    """dx += transposed depthwise conv of dz (data gradient only: constant-weight pooling)."""
    d = _dw_desc(x, dz, kernel, stride, padding, wp.shape[1])
    _check(lib().sf_dwconv_dgrad(ctypes.byref(d), dz.ptr(), dz.cs, dz.coff, _ptr(wp), dx.ptr(), dx.cs, dx.coff,
                                 x.C, _stream()), "sf_dwconv_dgrad")
    return dx


def gather_add(src, src_cmul, out, accumulate=True):
    """out[r, c] (+)= src[r, src.coff + c*src_cmul]."""
    _check(lib().sf_gather_add(src.ptr(), src.cs, src.coff, src_cmul, out.ptr(), out.cs, out.coff, out.rows, out.C,
                               1 if accumulate else 0, _stream()), "sf_gather_add")
    return out
