"""Test helper: run a callable with every public rba_amd.ops function (and the torch.nn.functional calls the NCHW / bf16x6 fallback paths still make)
check-summed, so that two runs of the same forward can be compared op by op (which op's output moved first?)."""
import contextlib
import types

import torch
import torch.nn.functional as F


def _csum(o):
    from rba_amd import ops
    if isinstance(o, torch.Tensor) and o.is_cuda and o.numel():
        t = o.detach().contiguous()
        if t.dtype in (torch.float32, torch.int32):
            return int(t.view(torch.int32).to(torch.int64).sum().item())
        return int(t.to(torch.int64).sum().item()) if not t.dtype.is_floating_point else float(t.double().sum().item())
    if isinstance(o, ops.SplitActivations):
        return _csum(o.unpack())                     # the rows that exist (the image's padding rows are never written, by design)
    if isinstance(o, (tuple, list)):
        return tuple(_csum(v) for v in o)
    return None


@contextlib.contextmanager
def traced():
    """with traced() as trace: ... -> trace = [(op name, argument shapes, output checksum), ...] in call order"""
    from rba_amd import ops
    trace, saved = [], []

    def wrap(mod, nm):
        f = getattr(mod, nm)

        def w(*args, **kw):
            out = f(*args, **kw)
            shapes = [tuple(x.shape) for x in args if isinstance(x, torch.Tensor)][:3]
            trace.append((f"{mod.__name__.split('.')[-1]}.{nm}", str(shapes), _csum(out)))
            return out
        saved.append((mod, nm, f))
        setattr(mod, nm, w)

    for n_, v_ in list(vars(ops).items()):
        if isinstance(v_, types.FunctionType) and not n_.startswith("_") and n_ not in ("split_mode", "set_concurrent_streams"):
            wrap(ops, n_)
    for n_ in ("conv2d", "linear", "interpolate", "group_norm", "layer_norm", "grid_sample"):
        wrap(F, n_)
    try:
        yield trace
    finally:
        for mod, nm, f in saved:
            setattr(mod, nm, f)


def first_difference(ta, tb):
    """(index, op, shapes, the three ops before it) of the first op whose checksum differs between two traces, or None"""
    for i, (x, y) in enumerate(zip(ta, tb)):
        if x != y:
            return i, y[0], y[1], [t[0] for t in tb[max(0, i - 3):i]]
    return None
