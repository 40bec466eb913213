"""ctypes binding of libbtorch_hip.so (include/bt_hip.h).

The library is the product: there is NO Python/ATen fallback. If the shared
object is missing or a tensor is not on a HIP device, calls raise.
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BT_LIB_PATH") or os.path.join(_HERE, "libbtorch_hip.so")   # BT_LIB_PATH: a diagnostic build (tools/stamps.py)
WORKSPACE_BYTES = 65536
KL_MAX_SEGMENTS = 64
PACK_MAX_SEGMENTS = 64
KL_RHO_IS_SIGMA = 1
KL_PRIOR_LAPLACE = 2
PRIOR_NORMAL, PRIOR_LAPLACE = 0, 1

_f32p = C.POINTER(C.c_float)
_vp = C.c_void_p


class bt_rng(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("call_base_dev", _vp), ("call", C.c_uint32), ("layer_id", C.c_uint32),
                ("sample0", C.c_uint32), ("reserved", C.c_uint32)]


class bt_params(C.Structure):
    _fields_ = [(n, _vp) for n in ("mu_w", "rho_w", "mu_b", "rho_b", "prior_mu_w", "prior_sigma_w", "prior_mu_b", "prior_sigma_b", "mu_packed", "sigma_packed")] \
        + [("prior_kind", C.c_int32), ("reserved", C.c_int32)]


class bt_draws(C.Structure):
    _fields_ = [("eps_w", _vp), ("eps_b", _vp), ("sign_in", _vp), ("sign_out", _vp), ("rng", bt_rng)]


class bt_epilogue(C.Structure):
    _fields_ = [("scale", _vp), ("shift", _vp), ("residual", _vp), ("residual_sample_stride", C.c_int64), ("relu", C.c_int32),
                ("pool", C.c_int32)]


class bt_pack_seg(C.Structure):
    _fields_ = [(n, _vp) for n in ("mu_w", "rho_w", "src_mu", "src_rho", "mu_packed", "sigma_packed", "state")] \
        + [("Co", C.c_int64), ("Ci", C.c_int64), ("taps", C.c_int64), ("force", C.c_int32), ("reserved", C.c_int32)]


class bt_conv2d_geom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "Ci", "H", "W", "Co", "kh", "kw", "sh", "sw", "ph", "pw", "dh", "dw", "groups")]


_lib = None
_lock = threading.Lock()

_FWD_TAIL = [_vp, C.c_int64, C.POINTER(bt_params), C.POINTER(bt_draws), C.POINTER(bt_epilogue), _vp, _vp, _vp, C.c_size_t, _vp]
_PROTOS = {
    "bt_version": (C.c_int, []),
    "bt_last_error_string": (C.c_char_p, []),
    "bt_last_kernel_name": (C.c_char_p, []),
    "bt_last_launch_info": (C.c_int, [C.POINTER(C.c_int64), C.c_int32]),
    "bt_fused_scratch_bytes": (C.c_size_t, [C.POINTER(bt_conv2d_geom), C.c_int32]),
    "bt_set_contraction": (C.c_int, [C.c_int]),
    "bt_conv2d_bwd_workspace": (C.c_size_t, [C.POINTER(bt_conv2d_geom), C.c_int32]),
    "bt_conv2d_bwd": (C.c_int, [C.POINTER(bt_conv2d_geom), C.c_int32, C.c_int32, _vp, C.c_int64, _vp, C.POINTER(bt_params), C.POINTER(bt_draws),
                                _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "bt_maxpool_3x3s2": (C.c_int, [_vp, _vp, C.c_int64, C.c_int32, C.c_int32, _vp]),
    "bt_conv2d_bwd_kl": (C.c_int, [C.POINTER(bt_conv2d_geom), C.c_int32, C.c_int32, _vp, C.c_int64, _vp, C.POINTER(bt_params), C.POINTER(bt_draws), _vp,
                                   _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "bt_kl_normal_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_uint32, _vp, _vp, _vp]),
    "bt_kl_normal_bwd_segs": (C.c_int, [C.c_int32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_int64), _vp, C.c_uint32,
                                        C.POINTER(_vp), C.POINTER(_vp), _vp]),
    "bt_reparam_linear_fwd": (C.c_int, [C.c_int32] * 4 + _FWD_TAIL),
    "bt_flipout_linear_fwd": (C.c_int, [C.c_int32] * 4 + _FWD_TAIL),
    "bt_reparam_conv2d_fwd": (C.c_int, [C.POINTER(bt_conv2d_geom), C.c_int32] + _FWD_TAIL),
    "bt_flipout_conv2d_fwd": (C.c_int, [C.POINTER(bt_conv2d_geom), C.c_int32] + _FWD_TAIL),
    "bt_kl_normal": (C.c_int, [C.c_int32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_int64),
                               C.POINTER(C.c_int32), C.c_uint32, _vp, _vp, C.c_size_t, _vp]),
    "bt_pack_sync": (C.c_int, [C.c_int32, C.POINTER(bt_pack_seg), _vp, C.c_size_t, _vp]),
    "bt_pack_params": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_int64, _vp, _vp, _vp]),
    "bt_rng_normal_fill": (C.c_int, [C.POINTER(bt_rng), C.c_uint32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, _vp, _vp]),
    "bt_rng_sign_fill": (C.c_int, [C.POINTER(bt_rng), C.c_uint32, C.c_int32, C.c_int64, _vp, _vp]),
    "bt_rng_philox_raw": (C.c_int, [C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "bt_mc_epilogue": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp]),
}
EXPORTS = tuple(_PROTOS)


def lib():
    """The loaded library. Raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} is missing: the HIP extension is the only implementation of this path "
                        "(no CPU/ATen fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "or `make -C bayesian_torch_amd/csrc`.")
                handle = C.CDLL(LIB_PATH)
                for name, (res, args) in _PROTOS.items():
                    fn = getattr(handle, name)
                    fn.restype, fn.argtypes = res, args
                _lib = handle
    return _lib


LAUNCH_INFO_FIELDS = ("workgroups", "m_tiles", "n_tiles", "S", "t_NI", "t_R", "t_Wt", "pixel_major", "row_tiles", "kl_slices", "groups", "n_bt", "n_rt", "n_ct", "fused_kl")


def last_launch_info():
    """Tile geometry of the calling thread's last fused-forward launch (bt_last_launch_info) as a dict."""
    v = (C.c_int64 * 16)()
    check(lib().bt_last_launch_info(v, 16))
    return dict(zip(LAUNCH_INFO_FIELDS, (int(x) for x in v)))


ERR_UNSUPPORTED = -2  # BT_ERR_UNSUPPORTED


def check(rc):
    if rc != 0:
        raise RuntimeError(f"libbtorch_hip error {rc}: {lib().bt_last_error_string().decode()}")


def ptr(t):
    return None if t is None else t.data_ptr()


def dev_f32(t, what):
    """Validate a tensor handed to the kernels: fp32, contiguous, on a HIP device."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{what} is on {t.device}: bayesian_torch_amd runs on a HIP (MI355X) device only; "
                           "there is no CPU implementation of this path")
    if t.dtype != torch.float32:
        raise TypeError(f"{what} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def stream_ptr(device=None):
    """The current torch stream OF ``device`` (not of whatever device happens to be current)."""
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)     # the handle without a Stream object (the eager path asks twice per layer)
    if raw is not None and isinstance(device, torch.device) and device.index is not None:
        return raw(device.index)
    return torch.cuda.current_stream(device).cuda_stream


def on(device):
    """Context for a C-ABI launch on tensors of ``device``: the kernels launch on HIP's current device, so make the
    tensors' device current for the call (a model on cuda:1 while cuda:0 is current would otherwise run on GPU 0
    against GPU 1's memory, on a stream that orders nothing)."""
    return torch.cuda.device(device)


_ws = {}
_ws_retired = []


def workspace(key, device, scratch=0):
    """One zero-initialised BT_WORKSPACE_BYTES buffer per (owner, device); kernels leave it zeroed. ``scratch`` more bytes behind it
    (bt_fused_scratch_bytes: the split-K forwards' slabs; contents never matter) -- the buffer grows when a call asks for more."""
    k = (key, device.index if device.index is not None else torch.cuda.current_device())
    w = _ws.get(k)
    need = WORKSPACE_BYTES + ((int(scratch) + 255) // 256) * 256
    if w is None or w.numel() < need:
        if w is not None:
            _ws_retired.append(w)    # (a captured graph may still hold the smaller buffer's address)
        w = torch.zeros(need, dtype=torch.uint8, device=device)
        _ws[k] = w
    return w


def kl_normal(segments, layer_ids=None, rho_is_sigma=False, out=None, owner="kl", laplace=False):
    """segments: list of (mu, rho, prior_mu, prior_sigma) device tensors. Returns a 0-dim tensor:
    sum over layers of (sum over the layer's segments of the element mean).  laplace: kl_div's 'laplace' branch for
    every segment (the prior tensors are not read, as in the reference)."""
    L = lib()
    dev = segments[0][0].device
    total = None
    for c0 in range(0, len(segments), KL_MAX_SEGMENTS):
        chunk = segments[c0:c0 + KL_MAX_SEGMENTS]
        n = len(chunk)
        keep = []
        arrs = [(_vp * n)() for _ in range(4)]
        numel = (C.c_int64 * n)()
        for i, seg in enumerate(chunk):
            ts = [dev_f32(t, "kl segment tensor") for t in seg]
            if not all(t.numel() == ts[0].numel() for t in ts):
                raise ValueError("kl segment tensors must have equal numel")
            keep.append(ts)
            for a, t in zip(arrs, ts):
                a[i] = t.data_ptr()
            numel[i] = ts[0].numel()
        lay = None
        if layer_ids is not None:
            lay = (C.c_int32 * n)(*layer_ids[c0:c0 + n])
        res = torch.empty((), dtype=torch.float32, device=dev) if (out is None or total is not None) else out
        with on(dev):
            check(L.bt_kl_normal(n, arrs[0], arrs[1], arrs[2], arrs[3], numel, lay, (KL_RHO_IS_SIGMA if rho_is_sigma else 0) | (KL_PRIOR_LAPLACE if laplace else 0),
                                 res.data_ptr(), workspace(owner, dev).data_ptr(), WORKSPACE_BYTES, stream_ptr(dev)))
        total = res if total is None else total + res
    return total
