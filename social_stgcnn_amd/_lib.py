"""ctypes binding of libstgcnn_hip.so (C ABI: include/stgcnn_hip.h).

Plumbing only: torch supplies device memory (`tensor.data_ptr()`) and the current HIP stream;
every compute call goes to a hand-written HIP kernel.  There is no CPU / eager fallback: if the
library is missing or a tensor is not on a GPU, the call raises.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# STG_USE_DIAG_LIB=1 (tools/ only) loads the diagnostic build `make -C csrc DIAG=1` produces; the product
# library reads no environment variable itself
DIAG = os.environ.get("STG_USE_DIAG_LIB", "0") not in ("", "0")
LIB_PATH = os.path.join(CSRC, "libstgcnn_hip_diag.so" if DIAG else "libstgcnn_hip.so")
ABI_VERSION = 7
OPT_WG_PATH, OPT_SPLIT_BF16, OPT_WAVE_PATH, OPT_BF16_STORE, OPT_F32_MFMA = 1, 2, 4, 8, 16
EUNSUPPORTED = -2            # STG_EUNSUPPORTED

c_f = ctypes.c_void_p          # device pointers travel as void*
c_i = ctypes.c_int
c_l = ctypes.c_int64


class ModelDesc(ctypes.Structure):
    """stg_model_desc (include/stgcnn_hip.h)."""
    _fields_ = [("n_stgcnn", ctypes.c_int32), ("n_txpcnn", ctypes.c_int32), ("c_in", ctypes.c_int32),
                ("c_out", ctypes.c_int32), ("t_obs", ctypes.c_int32), ("t_pred", ctypes.c_int32),
                ("kt", ctypes.c_int32), ("residual0", ctypes.c_int32), ("use_mdn", ctypes.c_int32),
                ("bn_mode", ctypes.c_int32), ("bn_eps", ctypes.c_float), ("bn_momentum", ctypes.c_float),
                ("flags", ctypes.c_int32), ("wg_waves", ctypes.c_int32)]


_SIGNATURES = {
    "stg_abi_version": (c_i, []),
    "stg_last_error": (ctypes.c_char_p, []),
    "stg_adj_build": (c_i, [c_f, c_l, c_l, c_l, c_l, c_f, c_i, c_i, c_i, c_i, c_f, c_f, c_f]),
    "stg_spatial_agg_fwd": (c_i, [c_f, c_l, c_l, c_l, c_l, c_f, c_l, c_f, c_i, c_i, c_i, c_i, c_f, c_f]),
    "stg_spatial_agg_bwd": (c_i, [c_f, c_f, c_l, c_f, c_i, c_i, c_i, c_i, c_f, c_f]),
    "stg_conv_t_fwd": (c_i, [c_f, c_l, c_l, c_l, c_l, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_f]),
    "stg_conv_t_bwd": (c_i, [c_f, c_l, c_l, c_l, c_l, c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_i, c_i,
                             c_f, c_f, c_f, c_f]),
    "stg_model_param_count": (c_l, [ctypes.POINTER(ModelDesc)]),
    "stg_model_buffer_count": (c_l, [ctypes.POINTER(ModelDesc)]),
    "stg_model_ws_floats": (c_l, [ctypes.POINTER(ModelDesc), c_i]),
    "stg_model_ws_tail_floats": (c_l, [ctypes.POINTER(ModelDesc), c_i, c_i]),
    "stg_model_stat_floats": (c_l, [ctypes.POINTER(ModelDesc)]),
    "stg_model_fwd_scratch_floats": (c_l, [ctypes.POINTER(ModelDesc), c_i, c_i]),
    "stg_model_bwd_scratch_floats": (c_l, [ctypes.POINTER(ModelDesc), c_i, c_i]),
    "stg_model_fwd": (c_i, [ctypes.POINTER(ModelDesc), c_f, c_f, c_f, c_l, c_l, c_l, c_l, c_f, c_l, c_f, c_i, c_i,
                            c_f, c_f, c_f, c_f, ctypes.POINTER(ctypes.c_void_p), c_i, c_f]),
    "stg_model_bwd": (c_i, [ctypes.POINTER(ModelDesc), c_f, c_f, c_f, c_l, c_l, c_l, c_l, c_f, c_l, c_f, c_i, c_i,
                            c_f, c_f, c_f, c_f, c_f, ctypes.POINTER(ctypes.c_void_p), c_i, c_f]),
    "stg_model_bwd_nll": (c_i, [ctypes.POINTER(ModelDesc), c_f, c_f, c_f, c_l, c_l, c_l, c_l, c_f, c_l, c_f, c_i, c_i,
                                c_f, c_f, c_f, c_f, c_f, c_f, c_f, ctypes.POINTER(ctypes.c_void_p), c_i, c_f]),
    "stg_model_bwd_step": (c_i, [ctypes.POINTER(ModelDesc), c_f, c_f, c_f, c_l, c_l, c_l, c_l, c_f, c_l, c_f, c_i, c_i,
                                 c_f, c_f, c_f, c_f, c_f, c_f, c_f, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), c_i,
                                 c_f]),
    "stg_bn_fold": (c_i, [ctypes.POINTER(ModelDesc), c_f, c_f, c_i, c_f, ctypes.POINTER(ctypes.c_void_p), c_i, c_f]),
    "stg_nll_fwd": (c_i, [c_f, c_l, c_l, c_l, c_l, c_f, c_f, c_f, c_i, c_i, c_i, c_f, c_f, c_f]),
    "stg_nll_bwd": (c_i, [c_f, c_f, c_i, c_i, c_i, c_f, c_f]),
    "stg_sgd_step": (c_i, [c_f, c_f, c_l, ctypes.c_float, c_f]),
    "stg_scene_order": (c_i, [c_f, c_i, c_i, c_f, c_f, c_f]),
    "stg_optim_step": (c_i, [c_f, c_f, c_l, c_f, ctypes.c_float, ctypes.c_float, c_f, c_f]),
    "stg_bestofk_eval": (c_i, [c_f, c_l, c_l, c_l, c_l, c_f, c_f, c_f, c_f, ctypes.c_uint64, c_i, c_i, c_i, c_i,
                               c_f, c_f, c_f]),
    "stg_gather_windows": (c_i, [c_f, c_f, c_f, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f]),
    "stg_dp_pack": (c_i, [c_f, c_f, c_f, c_f, c_i, ctypes.c_float, c_i, c_i, c_i, c_i, c_f, c_f]),
    "stg_dp_fold": (c_i, [c_f, c_f, ctypes.c_float, c_i, c_i, c_i, c_i, c_f, ctypes.POINTER(ctypes.c_void_p), c_i, c_f]),
    "stg_weighted_sum": (c_i, [c_f, c_f, c_i, c_f, c_f]),
    "stg_train_tail": (c_i, [ctypes.POINTER(ModelDesc), c_f, c_f, c_i, c_f, ctypes.POINTER(ctypes.c_void_p), c_i, c_f, c_f,
                             c_f, c_f, c_f, c_l, c_f, ctypes.c_float, ctypes.c_float, c_f, c_f]),
    "stg_selftest_mfma": (c_i, [c_f, c_f, c_i, c_f, c_f]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def build(verbose=False):
    """Compile libstgcnn_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"] + (["DIAG=1"] if DIAG else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libstgcnn_hip.so failed (exit %d)" % res.returncode)
    return LIB_PATH


def lib():
    """The loaded library (raises if it has not been built: there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "social_stgcnn_amd: %s is missing -- build it with `make -C %s` (hipcc, "
                "--offload-arch=gfx950) or `python -c 'import __graft_entry__ as g; g.build()'`. "
                "There is no CPU or eager-PyTorch fallback for this path." % (LIB_PATH, CSRC))
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        if h.stg_abi_version() != ABI_VERSION:
            raise RuntimeError("libstgcnn_hip.so ABI %d != binding ABI %d" % (h.stg_abi_version(), ABI_VERSION))
        _lib = h
    return _lib


class StepTail(ctypes.Structure):
    """stg_step_tail (include/stgcnn_hip.h)."""
    _fields_ = [("params", ctypes.c_void_p), ("lr_dev", ctypes.c_void_p), ("lr", ctypes.c_float),
                ("stats", ctypes.c_void_p), ("buffers", ctypes.c_void_p), ("nbt", ctypes.POINTER(ctypes.c_void_p)),
                ("n_bn", ctypes.c_int), ("total", ctypes.c_void_p)]


class HipEvents:
    """hipEvent_t handles for the per-kernel device timing the fused entry points offer (`events` argument of
    stg_model_fwd / stg_model_bwd): created through the HIP runtime torch already loaded, recorded by the library on
    the launch stream, read back here."""
    _hip = None

    @classmethod
    def hip(cls):
        if cls._hip is None:
            h = ctypes.CDLL("libamdhip64.so")
            h.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
            h.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
            h.hipEventSynchronize.argtypes = [ctypes.c_void_p]
            h.hipEventDestroy.argtypes = [ctypes.c_void_p]
            cls._hip = h
        return cls._hip

    def __init__(self, n):
        self.n = n
        self.arr = (ctypes.c_void_p * n)()
        for i in range(n):
            ev = ctypes.c_void_p()
            if self.hip().hipEventCreate(ctypes.byref(ev)) != 0:
                raise RuntimeError("hipEventCreate failed")
            self.arr[i] = ev

    def intervals_ms(self):
        """[t(events[k]) - t(events[k-1]) for k = 1..n-1] after the stream has drained."""
        out = []
        self.hip().hipEventSynchronize(self.arr[self.n - 1])
        for k in range(1, self.n):
            ms = ctypes.c_float()
            if self.hip().hipEventElapsedTime(ctypes.byref(ms), self.arr[k - 1], self.arr[k]) != 0:
                raise RuntimeError("hipEventElapsedTime failed")
            out.append(ms.value)
        return out

    def __del__(self):
        try:
            for i in range(self.n):
                self.hip().hipEventDestroy(self.arr[i])
        except Exception:
            pass


def check(rc, what):
    if rc != 0:
        msg = lib().stg_last_error().decode("utf-8", "replace")
        raise RuntimeError("%s failed (status %d): %s" % (what, rc, msg))


def stream_ptr():
    """The current torch HIP stream as a void* (kernels are enqueued on it)."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("social_stgcnn_amd runs on MI355X only: got a %s tensor (no CPU fallback)"
                               % t.device.type)


def as_f32(t, what):
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (got %s)" % (what, t.dtype))
    return t


def peds_arg(num_peds, n, device):
    """num_peds -> contiguous int32 device tensor of shape (N,) or None."""
    if num_peds is None:
        return None
    if not torch.is_tensor(num_peds):
        num_peds = torch.as_tensor(num_peds, dtype=torch.int32)
    num_peds = num_peds.to(device=device, dtype=torch.int32).contiguous()
    if num_peds.numel() != n:
        raise ValueError("num_peds has %d entries for a batch of %d scenes" % (num_peds.numel(), n))
    return num_peds
