"""ctypes handle of rba_amd/csrc/tune/librba_tune.so (probes, ablation builds, losing kernel variants; built by
`python -m rba_amd.csrc.build --tune`).  Tools only: nothing under rba_amd/ loads it."""
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: F401,E402  (its HIP runtime must be resident first, see rba_amd/_lib.py)

from rba_amd import _lib, ops  # noqa: E402

PATH = os.path.join(REPO, "rba_amd", "csrc", "tune", "librba_tune.so")
_h = None


def load():
    global _h
    if _h is None:
        if not os.path.exists(PATH):
            raise RuntimeError(f"{PATH} not built: python -m rba_amd.csrc.build --tune")
        _h = ctypes.CDLL(PATH)
        _h.rba_split_linear_v4_f32.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int64] + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        _h.rba_split_linear_v5_timing.argtypes = ([ctypes.c_void_p] * 4 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                                          ctypes.c_void_p, ctypes.c_void_p])
        _h.rba_reduce_f32_tune.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
        for f in (_h.rba_split_linear_v4_f32, _h.rba_split_linear_v5_timing, _h.rba_reduce_f32_tune):
            f.restype = ctypes.c_int
    return _h


def split_linear_cfg(x, planes, bias=None, act=0, cfg=401421, out_features=None):
    """K6 with an explicit tile configuration: cfg = [5|6|7|8]000000 (persistent variants) + 100000 L + 1000 RT + 100 CT + 10 G + D
    (+ 10000 PROBE for the ablation builds of 1421)."""
    K = x.shape[-1]
    M = x.numel() // K
    N = planes.shape[0] * 128 if out_features is None else int(out_features)
    out = torch.empty(tuple(x.shape[:-1]) + (N,), dtype=torch.float32, device=x.device)
    rc = load().rba_split_linear_v4_f32(x.data_ptr(), planes.data_ptr(), 0 if bias is None else bias.data_ptr(), out.data_ptr(), M, N, K,
                                        int(act), int(cfg), torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, f"rba_split_linear_v4_f32 cfg {cfg}")
    return out
