"""Guard-banded device allocations for the GPU tests ("next" #1a).

The one silent-corruption bug this repository shipped (round 4: the GroupNorm-moment epilogue of K6 wrote past its moment buffer for H*W = 128*odd) was found by
a reader, not by a test: every output the tests looked at was right, the damage was in memory nobody compared.  This helper makes that class of bug visible.

`install()` replaces the name `torch` in the product modules that allocate kernel outputs and workspaces (rba_amd.ops and the modeling files) with a proxy whose
`empty / empty_like / zeros / zeros_like` -- for HIP-device tensors only -- carve the tensor out of a larger byte buffer:

    [ HEAD bytes of 0xFF | payload (poisoned 0xFF = NaN for f32 / f16, -1 for integers; zeroed for zeros()) | TAIL bytes of 0xFF ]

and remember the buffer.  `check()` synchronises and verifies that every guard byte is still 0xFF: a kernel that wrote outside the tensor it was given fails the
test that launched it, whatever the test compares.  The NaN payload also exposes outputs a kernel did not fully write and (through NaN propagation) reads
outside an input that reach a result.  Everything else of `torch` passes through the proxy untouched; CPU allocations are not guarded.

tests/conftest.py installs it around every `-m gpu` test (opt out with @pytest.mark.no_canary: launch-count and timing tests, whose dispatch counts the fill
kernels would change).  Test infrastructure only: nothing under rba_amd/ imports it.
"""
import sys

import torch as _torch

HEAD = 4096            # bytes; keeps the payload's alignment what the caching allocator gives (512 B) and more
TAIL = 65536
FLUSH_BYTES = 24 << 30  # strong references are dropped (after a check) once this much is registered: full-size loops allocate GBs
POISON = 0xFF

GUARDED_MODULES = (
    "rba_amd.ops", "rba_amd.maskformer_model", "rba_amd.support", "rba_amd.metrics", "rba_amd.modeling.backbone.swin", "rba_amd.modeling.backbone.resnet",
    "rba_amd.modeling.pixel_decoder.msdeformattn", "rba_amd.modeling.pixel_decoder.ops.ms_deform_attn",
    "rba_amd.modeling.transformer_decoder.mask2former_transformer_decoder", "rba_amd.modeling.transformer_decoder.position_encoding",
)


class CanaryError(AssertionError):
    pass


class Registry:
    def __init__(self):
        self.entries = []       # (raw uint8 buffer, payload bytes, tag)
        self.bytes = 0
        self.allocations = 0
        self.checked = 0
        self.pending_error = None

    def add(self, raw, nbytes, tag):
        self.entries.append((raw, nbytes, tag))
        self.bytes += raw.numel()
        self.allocations += 1
        if self.bytes > FLUSH_BYTES and not _torch.cuda.is_current_stream_capturing():
            try:
                self.check()
            except CanaryError as e:          # raised from inside product code it would be swallowed or mis-attributed: keep it for the fixture
                self.pending_error = e

    def check(self):
        """synchronise, verify every guard, forget the buffers (the tensors carved out of them stay valid)"""
        entries, self.entries, self.bytes = self.entries, [], 0
        if not entries:
            if self.pending_error is not None:
                e, self.pending_error = self.pending_error, None
                raise e
            return 0
        _torch.cuda.synchronize()
        bad = None
        for raw, nbytes, _ in entries:
            b = (raw[:HEAD] != POISON).any() | (raw[HEAD + nbytes:] != POISON).any()
            bad = b if bad is None else (bad | b)
        self.checked += len(entries)
        if bool(bad.item()):
            lines = []
            for raw, nbytes, tag in entries:
                head, tail = raw[:HEAD] != POISON, raw[HEAD + nbytes:] != POISON
                if bool(head.any().item()) or bool(tail.any().item()):
                    hi = head.nonzero().flatten()
                    ti = tail.nonzero().flatten()
                    lines.append(f"  {tag}: payload {nbytes} B; {hi.numel()} guard bytes written BEFORE it (offsets {(hi[:4] - HEAD).tolist()}...), "
                                 f"{ti.numel()} AFTER it (offsets +{ti[:4].tolist()}...)")
            raise CanaryError("out-of-bounds device write(s) detected by the guard bands:\n" + "\n".join(lines[:12]))
        if self.pending_error is not None:
            e, self.pending_error = self.pending_error, None
            raise e
        return len(entries)


REGISTRY = Registry()


def _is_hip_device(device):
    if device is None:
        return False
    if isinstance(device, int):
        return True
    return _torch.device(device).type == "cuda"


def _shape(size):
    if len(size) == 1 and not isinstance(size[0], int):
        size = tuple(size[0])
    return tuple(int(s) for s in size)


def _caller():
    f = sys._getframe(3)
    return f"{f.f_code.co_filename.rsplit('/', 1)[-1]}:{f.f_lineno} {f.f_code.co_name}"


def guarded(shape, dtype, device, zero=False):
    dtype = dtype or _torch.get_default_dtype()
    n = 1
    for s in shape:
        n *= s
    item = _torch.empty((), dtype=dtype).element_size()
    nbytes = n * item
    pad = (-nbytes) % 16                      # the tail guard starts right after the payload; keep the raw size a multiple of 16
    raw = _torch.empty(HEAD + nbytes + pad + TAIL, dtype=_torch.uint8, device=device)
    raw.fill_(POISON)
    out = raw[HEAD:HEAD + nbytes].view(dtype).view(shape)
    if zero:
        out.zero_()
    REGISTRY.add(raw, nbytes, f"{_caller()} {tuple(shape)} {str(dtype).replace('torch.', '')}")
    return out


class TorchProxy:
    """stands in for the module `torch` inside a product module: allocation calls for HIP tensors are guarded, everything else passes through"""

    def __init__(self):
        self.__dict__["_real"] = _torch

    def __getattr__(self, name):
        return getattr(_torch, name)

    def _plain(self, kw):
        return any(k in kw for k in ("out", "pin_memory", "memory_format", "layout", "requires_grad", "names"))

    def empty(self, *size, dtype=None, device=None, **kw):
        if self._plain(kw) or not _is_hip_device(device):
            return _torch.empty(*size, dtype=dtype, device=device, **kw)
        return guarded(_shape(size), dtype, device)

    def zeros(self, *size, dtype=None, device=None, **kw):
        if self._plain(kw) or not _is_hip_device(device):
            return _torch.zeros(*size, dtype=dtype, device=device, **kw)
        return guarded(_shape(size), dtype, device, zero=True)

    def empty_like(self, x, dtype=None, device=None, **kw):
        device = x.device if device is None else device
        if self._plain(kw) or not _is_hip_device(device) or not x.is_contiguous():
            return _torch.empty_like(x, dtype=dtype, device=device, **kw)
        return guarded(tuple(x.shape), dtype or x.dtype, device)

    def zeros_like(self, x, dtype=None, device=None, **kw):
        device = x.device if device is None else device
        if self._plain(kw) or not _is_hip_device(device) or not x.is_contiguous():
            return _torch.zeros_like(x, dtype=dtype, device=device, **kw)
        return guarded(tuple(x.shape), dtype or x.dtype, device, zero=True)


_PROXY = TorchProxy()
_installed = []


def install():
    import importlib
    if _installed:
        return
    for name in GUARDED_MODULES:
        mod = importlib.import_module(name)
        if getattr(mod, "torch", None) is _torch:
            mod.torch = _PROXY
            _installed.append(mod)


def uninstall():
    while _installed:
        _installed.pop().torch = _torch


def empty(*size, dtype=None, device="cuda"):
    """for tests that hand a kernel a caller-allocated output: a guarded tensor"""
    return guarded(_shape(size), dtype, device)


def check():
    return REGISTRY.check()
