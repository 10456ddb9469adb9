"""Python entry points of the HIP kernels: validate (device / dtype / contiguity / shape), allocate the
output with torch (device memory + stream plumbing only) and enqueue the kernel on torch's current stream.

Error behaviour mirrors the reference's native op (pixel_decoder/ops/src/cuda/ms_deform_attn_cuda.cu:33-43:
AT_ASSERTM on contiguity and device -> RuntimeError): a bad argument raises RuntimeError (RbaHipError); there is
no fallback path.
"""
import contextlib
import ctypes
import threading
import functools
import math
import os

import torch

from . import _lib
from ._lib import RbaHipError


def _stream():
    """The HIP stream a kernel is enqueued on: torch's current stream of the CURRENT device -- which `_hip_op` has made the
    device of the call's tensors."""
    return torch.cuda.current_stream().cuda_stream


def _cuda_tensors(args):
    for a in args:
        if isinstance(a, torch.Tensor):
            if a.is_cuda:
                yield a
        elif isinstance(a, (list, tuple)):
            yield from _cuda_tensors(a)
        elif isinstance(a, SplitActivations):
            if a.data.is_cuda:
                yield a.data
        elif isinstance(a, torch.nn.Module):
            yield from (p_ for p_ in a.parameters(recurse=False) if p_.is_cuda)


def _hip_op(fn):
    """Every launch wrapper: all device tensors of a call must live on ONE HIP device (RbaHipError otherwise), and the call
    runs with that device current, so outputs, workspaces, `_stream()` and the kernel all refer to the tensors' device and its
    current stream -- also when the caller never called torch.cuda.set_device (get_model(device="cuda:1"))."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        dev = None
        for t in _cuda_tensors(args + tuple(kwargs.values())):
            if dev is None:
                dev = t.device
            elif t.device != dev:
                raise RbaHipError(f"{fn.__name__}: tensors on different devices ({dev} and {t.device})")
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapper


def _chk(t, name, dtype=torch.float32, dim=None):
    if not isinstance(t, torch.Tensor):
        raise RbaHipError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RbaHipError(f"{name} must be a HIP (cuda) tensor; rba_amd kernels have no CPU path")
    if t.dtype != dtype:
        raise RbaHipError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RbaHipError(f"{name} tensor has to be contiguous")
    if dim is not None and t.dim() != dim:
        raise RbaHipError(f"{name} must have {dim} dims, got shape {tuple(t.shape)}")
    return t


def _p(t):
    return 0 if t is None else t.data_ptr()


SCORE_MODES = {"rba": 0, "energy": 1, "neg_logit_sum": 2}


def _mode(score):
    if score not in SCORE_MODES:
        raise RbaHipError(f"unknown score {score!r}; choose from {sorted(SCORE_MODES)}")
    return SCORE_MODES[score]


_K1_WORKSPACES = {}


def _k1_workspace(device):
    """8 zeroed bytes per (device, stream) for K1's dynamic tile counter (the kernel leaves them zero)."""
    key = (device.index, _stream())                      # called under _hip_op: the current device IS `device`
    ws = _K1_WORKSPACES.get(key)
    if ws is None:
        ws = _K1_WORKSPACES[key] = torch.zeros(2, dtype=torch.int32, device=device)
    return ws


@_hip_op
def rba_reduce(mask_pred, cls_prob, want_sem_seg=False, want_argmax=False, score="rba"):
    """K1.  mask_pred [Q,H,W] full-resolution mask logits, cls_prob [Q,K] -> (rba [H,W], sem_seg [K,H,W] | None,
    argmax int32 [H,W] | None).  maskformer_model.py:381-386 + evaluate_ood.py:150 + support.py:385-388."""
    lib = _lib.load()
    _chk(mask_pred, "mask_pred", dim=3)
    _chk(cls_prob, "cls_prob", dim=2)
    Q, H, W = mask_pred.shape
    if cls_prob.shape[0] != Q:
        raise RbaHipError(f"cls_prob has {cls_prob.shape[0]} queries, mask_pred {Q}")
    K = cls_prob.shape[1]
    dev = mask_pred.device
    rba = torch.empty((H, W), dtype=torch.float32, device=dev)
    sem = torch.empty((K, H, W), dtype=torch.float32, device=dev) if want_sem_seg else None
    arg = torch.empty((H, W), dtype=torch.int32, device=dev) if want_argmax else None
    _lib.check(lib.rba_reduce_ws_f32(_p(mask_pred), _p(cls_prob), _p(rba), _p(sem), _p(arg), Q, K, H * W, _mode(score),
                                     _p(_k1_workspace(dev)), _stream()), "rba_reduce_ws_f32")
    return rba, sem, arg


@_hip_op
def rba_reduce_up4(mask_lowres, cls_prob, crop_hw, want_sem_seg=False, want_argmax=False, score="rba"):
    """K1 fused with the x4 upsample (maskformer_model.py:294-299) and the crop (:330-332).
    mask_lowres [Q,h,w]; outputs are [crop_h, crop_w] of the virtual [4h,4w] map."""
    lib = _lib.load()
    _chk(mask_lowres, "mask_lowres", dim=3)
    _chk(cls_prob, "cls_prob", dim=2)
    Q, h, w = mask_lowres.shape
    K = cls_prob.shape[1]
    ch, cw = int(crop_hw[0]), int(crop_hw[1])
    dev = mask_lowres.device
    rba = torch.empty((ch, cw), dtype=torch.float32, device=dev)
    sem = torch.empty((K, ch, cw), dtype=torch.float32, device=dev) if want_sem_seg else None
    arg = torch.empty((ch, cw), dtype=torch.int32, device=dev) if want_argmax else None
    _lib.check(lib.rba_reduce_up4_f32(_p(mask_lowres), _p(cls_prob), _p(rba), _p(sem), _p(arg), Q, K, h, w, ch, cw,
                                      _mode(score), _stream()), "rba_reduce_up4_f32")
    return rba, sem, arg


@_hip_op
def resample_bilinear(x, size, add=None):
    """F.interpolate(x, size, mode="bilinear", align_corners=False) for x [C,h,w] or [B,C,h,w]; optional fused
    `+ add` (the FPN top-down sum of msdeformattn.py:358)."""
    lib = _lib.load()
    _chk(x, "x")
    if x.dim() not in (3, 4):
        raise RbaHipError("x must be [C,h,w] or [B,C,h,w]")
    H, W = int(size[0]), int(size[1])
    lead = x.shape[:-2]
    C = 1
    for s in lead:
        C *= int(s)
    h, w = x.shape[-2:]
    out = torch.empty(tuple(lead) + (H, W), dtype=torch.float32, device=x.device)
    if add is not None:
        _chk(add, "add")
        if tuple(add.shape) != tuple(out.shape):
            raise RbaHipError("add must have the output's shape")
    _lib.check(lib.rba_resample_bilinear_f32(_p(x), _p(add), _p(out), C, h, w, H, W, _stream()),
               "rba_resample_bilinear_f32")
    return out


@_hip_op
def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
                           im2col_step=128):
    """K2.  Same signature and checks as MultiScaleDeformableAttention.ms_deform_attn_forward
    (pixel_decoder/ops/functions/ms_deform_attn_func.py:36-37, ops/src/cuda/ms_deform_attn_cuda.cu:25-85):
    value [N,S,M,D], spatial_shapes [L,2] int64, level_start_index [L] int64, sampling_locations
    [N,Lq,M,L,P,2], attention_weights [N,Lq,M,L,P] -> [N,Lq,M*D].  `im2col_step` is accepted for signature
    compatibility and only validated (the kernel needs no batch chunking)."""
    lib = _lib.load()
    # float or double, like the reference op (AT_DISPATCH_FLOATING_TYPES, ms_deform_attn_cuda.cu:69); all three tensors of one type
    dt = value.dtype if isinstance(value, torch.Tensor) and value.dtype == torch.float64 else torch.float32
    _chk(value, "value", dt, dim=4)
    _chk(spatial_shapes, "spatial_shapes", torch.int64, 2)
    _chk(level_start_index, "level_start_index", torch.int64, 1)
    _chk(sampling_locations, "sampling_loc", dt, dim=6)
    _chk(attention_weights, "attn_weight", dt, dim=5)
    N, S, M, D = value.shape
    _, Lq, M2, L, P, two = sampling_locations.shape
    if M2 != M or two != 2 or sampling_locations.shape[0] != N:
        raise RbaHipError("sampling_loc shape does not match value")
    if tuple(attention_weights.shape) != (N, Lq, M, L, P):
        raise RbaHipError("attn_weight shape does not match sampling_loc")
    if spatial_shapes.shape[0] != L or level_start_index.shape[0] != L:
        raise RbaHipError("spatial_shapes / level_start_index must have L rows")
    step = min(N, int(im2col_step))
    if N > 0 and N % step != 0:
        raise RbaHipError(f"batch({N}) must divide im2col_step({step})")
    out = torch.empty((N, Lq, M * D), dtype=dt, device=value.device)
    fn, name = ((lib.rba_ms_deform_attn_fwd_f64, "rba_ms_deform_attn_fwd_f64") if dt == torch.float64 else
                (lib.rba_ms_deform_attn_fwd_f32, "rba_ms_deform_attn_fwd_f32"))
    _lib.check(fn(_p(value), _p(spatial_shapes), _p(level_start_index), _p(sampling_locations), _p(attention_weights), _p(out),
                  N, S, M, D, L, Lq, P, _stream()), name)
    return out


@_hip_op
def masked_xattn(q, k, v, mask_logits=None, split_keys=None):
    """K3.  q [B,Q,nH,hd] (unscaled), k, v [B,S,nH,hd], mask_logits [B,Q,S] | None -> [B,Q,nH*hd].
    split_keys: use the split-key matrix-pipe path (needs a scratch buffer, allocated here); None = automatic
    (it wins from about a thousand keys; below that its three launches cost more than they save)."""
    lib = _lib.load()
    _chk(q, "q", dim=4)
    _chk(k, "k", dim=4)
    _chk(v, "v", dim=4)
    B, Q, nH, hd = q.shape
    S = k.shape[1]
    if tuple(k.shape) != (B, S, nH, hd) or tuple(v.shape) != (B, S, nH, hd):
        raise RbaHipError("k / v shape mismatch")
    if mask_logits is not None:
        _chk(mask_logits, "mask_logits", dim=3)
        if tuple(mask_logits.shape) != (B, Q, S):
            raise RbaHipError("mask_logits must be [B,Q,S]")
    out = torch.empty((B, Q, nH * hd), dtype=torch.float32, device=q.device)
    ws = None
    if split_keys is None:
        split_keys = S >= 1024
    if split_keys:
        ws = torch.empty(max(lib.rba_masked_xattn_workspace_bytes(B, Q, S, nH) // 4, 4), dtype=torch.float32, device=q.device)
    _lib.check(lib.rba_masked_xattn_f32(_p(q), _p(k), _p(v), _p(mask_logits), _p(out), _p(ws), B, Q, S, nH, hd, _stream()),
               "rba_masked_xattn_f32")
    return out


@_hip_op
def mask_logits(embed, feat, mode=None):
    """K4.  einsum("bqc,bchw->bqhw") (mask2former_transformer_decoder.py:479): embed [B,Q,C], feat [B,C,h,w].  `mode` (default: _split_mode())
    "f16x3" = three f16 matrix-pipe products per fp32 product (|x| < 65504), anything else = the exact-fp32 MFMA kernel."""
    lib = _lib.load()
    _chk(embed, "embed", dim=3)
    _chk(feat, "feat")
    if feat.dim() not in (3, 4):
        raise RbaHipError("feat must be [B,C,N] or [B,C,h,w]")
    B, Q, C = embed.shape
    if feat.shape[0] != B or feat.shape[1] != C:
        raise RbaHipError("feat shape does not match embed")
    sp = tuple(feat.shape[2:])
    N = 1
    for s in sp:
        N *= int(s)
    out = torch.empty((B, Q) + sp, dtype=torch.float32, device=embed.device)
    if (_split_mode() if mode is None else mode) == "f16x3":
        _lib.check(lib.rba_mask_logits_f16x3_f32(_p(embed), _p(feat), _p(out), B, Q, C, N, _stream()), "rba_mask_logits_f16x3_f32")
    else:
        _lib.check(lib.rba_mask_logits_f32(_p(embed), _p(feat), _p(out), B, Q, C, N, _stream()), "rba_mask_logits_f32")
    return out


@_hip_op
def swin_bias_fragments(rel_bias, window_size):
    """[nH,N,N] gathered relative-position bias -> the MFMA-fragment-ordered copy K5 reads with coalesced loads."""
    lib = _lib.load()
    _chk(rel_bias, "rel_bias", dim=3)
    nH = rel_bias.shape[0]
    n = lib.rba_swin_bias_fragments_elems(nH, window_size)
    frag = torch.empty(n, dtype=torch.float32, device=rel_bias.device)
    _lib.check(lib.rba_swin_bias_fragments_f32(_p(rel_bias), _p(frag), nH, window_size, _stream()),
               "rba_swin_bias_fragments_f32")
    return frag


def swin_window_attn_split_ok(head_dim, window_size):
    """Geometries whose window-attention kernel can write SplitActivations (the f16x3 matrix-pipe form)."""
    return head_dim == 32 and window_size == 12


@_hip_op
def swin_window_attn(qkv, qkv_bias, rel_bias, H, W, num_heads, window_size, shift, bias_frag=None, split_out=False):
    """K5.  qkv [B,H*W,3*C] = Linear(norm1(x)) on un-padded tokens, qkv_bias [3*C], rel_bias [nH,N,N] ->
    attention output [B,H*W,C] (before proj).  swin.py:131-171 + :251-284 + :413-440."""
    lib = _lib.load()
    _chk(qkv, "qkv", dim=3)
    _chk(qkv_bias, "qkv_bias", dim=1)
    _chk(rel_bias, "rel_bias", dim=3)
    B, L, C3 = qkv.shape
    C = C3 // 3
    if L != H * W or C3 != 3 * C or C % num_heads:
        raise RbaHipError("qkv shape does not match H, W, num_heads")
    hd = C // num_heads
    N = window_size * window_size
    if tuple(rel_bias.shape) != (num_heads, N, N) or qkv_bias.numel() != C3:
        raise RbaHipError("rel_bias must be [nH, ws*ws, ws*ws] and qkv_bias [3C]")
    if bias_frag is not None:
        _chk(bias_frag, "bias_frag", dim=1)
        if bias_frag.numel() != lib.rba_swin_bias_fragments_elems(num_heads, window_size):
            raise RbaHipError("bias_frag has the wrong size")
    if split_out:                                   # the proj Linear's split A operand (see swin_window_attn_split_ok)
        if not swin_window_attn_split_ok(hd, window_size) or bias_frag is None:
            raise RbaHipError("split_out needs head_dim 32, 12 x 12 windows and bias_frag")
        out = SplitActivations.empty((B, L, C), qkv.device)
        _lib.check(lib.rba_swin_window_attn_split_out_f32(_p(qkv), _p(qkv_bias), _p(bias_frag), _p(out.data), B, H, W, num_heads, hd,
                                                          window_size, shift, _stream()), "rba_swin_window_attn_split_out_f32")
        return out
    out = torch.empty((B, L, C), dtype=torch.float32, device=qkv.device)
    _lib.check(lib.rba_swin_window_attn_f32(_p(qkv), _p(qkv_bias), _p(rel_bias), _p(bias_frag), _p(out), B, H, W,
                                            num_heads, hd, window_size, shift, _stream()), "rba_swin_window_attn_f32")
    return out


SWIN_ATTN_FUSED = os.environ.get("RBA_SWIN_ATTN_FUSED", "1") != "0"      # A/B switch (tools): 0 keeps the unfused LN -> qkv -> K5 -> proj sequence


def swin_attn_block_ok(C, num_heads, window_size):
    """True when swin_attn_block() has a kernel for this geometry (K7: head_dim 32, 12 x 12 windows, f16x3 mode)."""
    return (SWIN_ATTN_FUSED and _split_mode() == "f16x3" and num_heads * 32 == C
            and bool(_lib.load().rba_swin_attn_block_supported(int(C), int(window_size))))


def swin_attn_qkv_ok(C, num_heads, window_size):
    """True when swin_attn_qkv() has a kernel for this geometry (K7 without proj: C = 128 / 256, head_dim 32, 12 x 12 windows, f16x3 mode)."""
    return (SWIN_ATTN_FUSED and _split_mode() == "f16x3" and num_heads * 32 == C
            and bool(_lib.load().rba_swin_attn_qkv_supported(int(C), int(window_size))))


@_hip_op
def swin_attn_qkv(x, norm1, image, qkv_bias, bias_frag, H, W, window_size, shift):
    """K7 without proj: window_attention(qkv(norm1(x))) before the output projection (swin.py:235-168), returned as SplitActivations -- the proj Linear's A
    operand: ``x = ops.linear(y, attn.proj, residual=x)`` finishes the half block.  x [B, H*W, C] is only read.  Check swin_attn_qkv_ok first."""
    lib = _lib.load()
    _chk(x, "x", dim=3)
    B, L, C = x.shape
    if L != H * W or not lib.rba_swin_attn_qkv_supported(int(C), int(window_size)):
        raise RbaHipError("swin_attn_qkv: check swin_attn_qkv_ok(C, num_heads, window_size) first; x must be [B, H*W, C]")
    g1, b1, eps1 = norm1
    for t, name in ((g1, "norm1.weight"), (b1, "norm1.bias")):
        _chk(t, name, dim=1)
        if t.numel() != C:
            raise RbaHipError(f"{name} must have C elements")
    _chk(qkv_bias, "qkv_bias", dim=1)
    _chk(bias_frag, "bias_frag", dim=1)
    _chk(image, "image", dtype=torch.uint8, dim=1)
    if (qkv_bias.numel() != 3 * C or image.numel() != lib.rba_swin_attn_block_weight_bytes(int(C))
            or bias_frag.numel() != lib.rba_swin_bias_fragments_elems(C // 32, int(window_size))):
        raise RbaHipError("qkv_bias must be [3C]; image / bias_frag must come from swin_attn_block_weights / swin_bias_fragments")
    out = SplitActivations.empty((B, L, C), x.device)
    _lib.check(lib.rba_swin_attn_qkv_split_out_f32(_p(x), _p(out.data), _p(g1), _p(b1), float(eps1), _p(image), _p(qkv_bias), _p(bias_frag), B, H, W, C,
                                                   int(window_size), int(shift), _stream()), "rba_swin_attn_qkv_split_out_f32")
    return out


@_hip_op
def swin_attn_block_weights(qkv_weight, proj_weight):
    """qkv.weight [3C, C] + proj.weight [C, C] -> K7's per-head image of MFMA operand fragments (uint8; once per weight load)."""
    lib = _lib.load()
    _chk(qkv_weight, "qkv_weight", dim=2)
    _chk(proj_weight, "proj_weight", dim=2)
    C = proj_weight.shape[0]
    if tuple(qkv_weight.shape) != (3 * C, C) or tuple(proj_weight.shape) != (C, C) or C % 32:
        raise RbaHipError("swin_attn_block_weights needs qkv.weight [3C, C] and proj.weight [C, C], C % 32 == 0")
    img = torch.empty(int(lib.rba_swin_attn_block_weight_bytes(C)), dtype=torch.uint8, device=qkv_weight.device)
    _lib.check(lib.rba_swin_attn_block_pack_f32(_p(qkv_weight), _p(proj_weight), _p(img), C, _stream()), "rba_swin_attn_block_pack_f32")
    return img


@_hip_op
def swin_attn_block(x, norm1, image, qkv_bias, bias_frag, proj_bias, H, W, window_size, shift, norm2=None):
    """K7: x [B, H*W, C] <- x + proj(window_attention(qkv(norm1(x)))) IN PLACE (swin.py:235-284); with ``norm2`` also returns
    y2 = norm2(x) (:284-293).  norm1 / norm2 = (weight, bias, eps); image = swin_attn_block_weights(...); bias_frag = swin_bias_fragments(...).
    Returns (x, y2 | None).  Check swin_attn_block_ok first."""
    lib = _lib.load()
    _chk(x, "x", dim=3)
    B, L, C = x.shape
    if L != H * W or not lib.rba_swin_attn_block_supported(int(C), int(window_size)):
        raise RbaHipError("swin_attn_block: check swin_attn_block_ok(C, num_heads, window_size) first; x must be [B, H*W, C]")
    g1, b1, eps1 = norm1
    for t, name in ((g1, "norm1.weight"), (b1, "norm1.bias"), (proj_bias, "proj_bias")):
        _chk(t, name, dim=1)
        if t.numel() != C:
            raise RbaHipError(f"{name} must have C elements")
    _chk(qkv_bias, "qkv_bias", dim=1)
    _chk(bias_frag, "bias_frag", dim=1)
    _chk(image, "image", dtype=torch.uint8, dim=1)
    if (qkv_bias.numel() != 3 * C or image.numel() != lib.rba_swin_attn_block_weight_bytes(int(C))
            or bias_frag.numel() != lib.rba_swin_bias_fragments_elems(C // 32, int(window_size))):
        raise RbaHipError("qkv_bias must be [3C]; image / bias_frag must come from swin_attn_block_weights / swin_bias_fragments")
    y2, g2, b2, eps2 = None, None, None, 0.0
    if norm2 is not None:
        g2, b2, eps2 = norm2
        _chk(g2, "norm2.weight", dim=1)
        _chk(b2, "norm2.bias", dim=1)
        if g2.numel() != C or b2.numel() != C:
            raise RbaHipError("norm2 weight / bias must have C elements")
        y2 = torch.empty_like(x)
    _lib.check(lib.rba_swin_attn_block_f32(_p(x), _p(y2), _p(g1), _p(b1), float(eps1), _p(image), _p(qkv_bias), _p(bias_frag), _p(proj_bias),
                                           _p(g2), _p(b2), float(eps2), B, H, W, C, int(window_size), int(shift), _stream()),
               "rba_swin_attn_block_f32")
    return x, y2


@_hip_op
def group_norm(x, num_groups, weight, bias, eps=1e-5, relu=False):
    """GroupNorm (+ReLU) of x [B,C,h,w] -- the norm/activation of Detectron2's Conv2d wrapper
    (msdeformattn.py:222-235, 278-297)."""
    lib = _lib.load()
    _chk(x, "x", dim=4)
    _chk(weight, "weight", dim=1)
    _chk(bias, "bias", dim=1)
    B, C, h, w = x.shape
    if C % num_groups or weight.numel() != C or bias.numel() != C:
        raise RbaHipError("channels must be divisible by num_groups and match weight/bias")
    nbytes = lib.rba_group_norm_workspace_bytes(B, C, h * w, num_groups)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    _lib.check(lib.rba_group_norm_f32(_p(x), _p(weight), _p(bias), _p(y), _p(ws), B, C, h * w, num_groups, float(eps),
                                      int(bool(relu)), _stream()), "rba_group_norm_f32")
    return y


@_hip_op
def add_layer_norm(x, weight, bias, eps=1e-5, residual=None, residual_bias=None, inplace_sum=False, frag=False):
    """y = LayerNorm(x + residual + residual_bias) over the last dim.  Returns (s, y) where s = the summed tensor
    (x itself when there is nothing to add; written in place over x when inplace_sum, else a new tensor).
    ``frag``: y is returned as SplitActivations (the f16x3 GEMM's A operand, already split and fragment-ordered) for linear()."""
    lib = _lib.load()
    _chk(x, "x")
    _chk(weight, "weight", dim=1)
    _chk(bias, "bias", dim=1)
    C = x.shape[-1]
    if weight.numel() != C or bias.numel() != C:
        raise RbaHipError("weight / bias must match the last dimension")
    rows = x.numel() // C if C else 0
    if residual is not None:
        _chk(residual, "residual")
        if tuple(residual.shape) != tuple(x.shape):
            raise RbaHipError("residual must have x's shape")
    if residual_bias is not None:
        _chk(residual_bias, "residual_bias", dim=1)
        if residual_bias.numel() != C:
            raise RbaHipError("residual_bias must have C elements")
    has_sum = residual is not None or residual_bias is not None
    s = x if (not has_sum or inplace_sum) else torch.empty_like(x)
    if frag:
        if C % 32:
            raise RbaHipError("add_layer_norm(frag=True) needs C % 32 == 0")
        y = SplitActivations.empty(tuple(x.shape), x.device)
        _lib.check(lib.rba_add_layer_norm_frag_f32(_p(x), _p(residual), _p(residual_bias), _p(weight), _p(bias),
                                                   _p(s) if has_sum else 0, _p(y.data), rows, C, float(eps), _stream()),
                   "rba_add_layer_norm_frag_f32")
        return s, y
    y = torch.empty_like(x)
    _lib.check(lib.rba_add_layer_norm_f32(_p(x), _p(residual), _p(residual_bias), _p(weight), _p(bias),
                                          _p(s) if has_sum else 0, _p(y), rows, C, float(eps), _stream()),
               "rba_add_layer_norm_f32")
    return s, y


@_hip_op
def msda_prepare(raw, reference_points, spatial_shapes, M, L, P):
    """raw [N, Lq, M*L*P*3] (offsets | logits of the fused sampling Linear), reference_points [N, Lq, L, 2], spatial_shapes [L,2] int64
    -> (sampling_locations [N,Lq,M,L,P,2], attention_weights [N,Lq,M,L,P]) as MSDeformAttn.forward computes them."""
    lib = _lib.load()
    _chk(raw, "raw", dim=3)
    _chk(reference_points, "reference_points", dim=4)
    _chk(spatial_shapes, "spatial_shapes", torch.int64, 2)
    N, Lq, W3 = raw.shape
    if W3 != 3 * M * L * P or tuple(reference_points.shape) != (N, Lq, L, 2) or spatial_shapes.shape[0] != L:
        raise RbaHipError("msda_prepare: shapes do not match")
    loc = torch.empty((N, Lq, M, L, P, 2), dtype=torch.float32, device=raw.device)
    attw = torch.empty((N, Lq, M, L, P), dtype=torch.float32, device=raw.device)
    _lib.check(lib.rba_msda_prepare_f32(_p(raw), _p(reference_points), _p(spatial_shapes), _p(loc), _p(attw), N * Lq, M, L, P, _stream()),
               "rba_msda_prepare_f32")
    return loc, attw


def msda_fused_ok(D, L, P, S=None, M=None):
    """Geometries of the one-launch deformable attention core (head_dim 32, 4 points, 1 or 3 levels: every released configuration); with
    S positions and M heads given, also that one image's value [S, M, 32] fp32 stays below 4 GiB (the kernel's 32-bit tap offsets)."""
    return D == 32 and P == 4 and L in (1, 3) and (S is None or M is None or S * M * 128 < (1 << 32))


@_hip_op
def msda_fused(value, spatial_shapes, level_start_index, raw, reference_points, M, L, P):
    """MSDeformAttn.forward's core in one launch: value [N,S,M,32], raw [N,Lq,M*L*P*3] (offsets | logits of the fused sampling Linear),
    reference_points [N,Lq,L,2] -> [N,Lq,M*32]; sampling locations and the softmax over the L*P logits are computed inside the gather
    kernel (bit-identical to msda_prepare + ms_deform_attn_forward)."""
    lib = _lib.load()
    _chk(value, "value", dim=4)
    _chk(spatial_shapes, "spatial_shapes", torch.int64, 2)
    _chk(level_start_index, "level_start_index", torch.int64, 1)
    _chk(raw, "raw", dim=3)
    _chk(reference_points, "reference_points", dim=4)
    N, S, M2, D = value.shape
    _, Lq, W3 = raw.shape
    if (M2 != M or not msda_fused_ok(D, L, P, S, M) or W3 != 3 * M * L * P or raw.shape[0] != N or tuple(reference_points.shape) != (N, Lq, L, 2)
            or spatial_shapes.shape[0] != L or level_start_index.shape[0] != L):
        raise RbaHipError("msda_fused: shapes do not match (head_dim 32, P = 4, L in {1, 3})")
    out = torch.empty((N, Lq, M * D), dtype=torch.float32, device=value.device)
    _lib.check(lib.rba_msda_fused_f32(_p(value), _p(spatial_shapes), _p(level_start_index), _p(raw), _p(reference_points), _p(out),
                                      N, S, M, D, L, Lq, P, _stream()), "rba_msda_fused_f32")
    return out


@_hip_op
def merge_layer_norm(x, H, W, weight, bias, eps=1e-5):
    """PatchMerging's gather + LayerNorm: x [B, H*W, C] -> [B, ceil(H/2)*ceil(W/2), 4C] = LN over the 2x2 neighbourhoods in the order
    (ee, oe, eo, oo), odd maps zero-padded -- the concatenated tensor is never materialised."""
    lib = _lib.load()
    _chk(x, "x", dim=3)
    _chk(weight, "weight", dim=1)
    _chk(bias, "bias", dim=1)
    B, L, C = x.shape
    if L != H * W or C % 4 or weight.numel() != 4 * C or bias.numel() != 4 * C:
        raise RbaHipError("merge_layer_norm needs x [B, H*W, C] (C % 4 == 0) and a LayerNorm over 4C channels")
    y = torch.empty((B, ((H + 1) // 2) * ((W + 1) // 2), 4 * C), dtype=torch.float32, device=x.device)
    _lib.check(lib.rba_merge_layer_norm_f32(_p(x), _p(weight), _p(bias), _p(y), B, H, W, C, float(eps), _stream()), "rba_merge_layer_norm_f32")
    return y


@_hip_op
def skinny_linear(x, weight, bias=None, relu=False, x_add=None, add_cols=None, segments=1, out=None):
    """F.linear(x, weight, bias) [+ ReLU] for x with at most 128 rows (the decoder's 100 queries): [..., K] -> [..., N].
    x_add: output columns n < add_cols (default: all; a multiple of 16) are computed from ``x + x_add`` -- the query position embedding of
    mask2former_transformer_decoder.py:48-58, 106-118 added inside the projection.  segments = s > 1: the N outputs are returned as s
    separately contiguous tensors [..., N / s] (q, k, v of a stacked in_proj weight) written by the one launch.
    out (segments = 1 only): a caller-owned contiguous fp32 [..., N] tensor (e.g. a row slice of a larger one) the launch writes instead of a new tensor."""
    lib = _lib.load()
    _chk(x, "x")
    _chk(weight, "weight", dim=2)
    K = x.shape[-1]
    N = weight.shape[0]
    M = x.numel() // K if K else 0
    if weight.shape[1] != K or K % 32 or M > 128:
        raise RbaHipError("skinny_linear needs weight [N,K], K % 32 == 0 and at most 128 rows")
    if bias is not None:
        _chk(bias, "bias", dim=1)
        if bias.numel() != N:
            raise RbaHipError("bias must have N elements")
    segments = int(segments)
    if segments < 1 or N % segments:
        raise RbaHipError("segments must divide N")
    if out is not None:
        _chk(out, "out")
        if segments != 1 or tuple(out.shape) != tuple(x.shape[:-1]) + (N,):
            raise RbaHipError("out must be a contiguous fp32 [..., N] tensor with x's leading shape (segments = 1)")
    if x_add is None and segments == 1:
        if out is None:
            out = torch.empty(tuple(x.shape[:-1]) + (N,), dtype=torch.float32, device=x.device)
        _lib.check(lib.rba_skinny_linear_f32(_p(x), _p(weight), _p(bias), _p(out), M, N, K, int(bool(relu)), _stream()),
                   "rba_skinny_linear_f32")
        return out
    if x_add is not None:
        _chk(x_add, "x_add")
        if tuple(x_add.shape) != tuple(x.shape):
            raise RbaHipError("x_add must have x's shape")
    add_cols = N if add_cols is None else int(add_cols)
    if add_cols % 16 or not 0 <= add_cols <= N:
        raise RbaHipError("add_cols must be a multiple of 16 in [0, N]")
    seg_n = N // segments
    if out is not None:
        _lib.check(lib.rba_skinny_linear_add_f32(_p(x), _p(x_add), add_cols, _p(weight), _p(bias), _p(out), M, N, K, int(bool(relu)), 0, _stream()),
                   "rba_skinny_linear_add_f32")
        return out
    out = torch.empty((segments,) + tuple(x.shape[:-1]) + (seg_n,), dtype=torch.float32, device=x.device)
    _lib.check(lib.rba_skinny_linear_add_f32(_p(x), _p(x_add), add_cols, _p(weight), _p(bias), _p(out), M, N, K, int(bool(relu)),
                                             seg_n if segments > 1 else 0, _stream()), "rba_skinny_linear_add_f32")
    return out[0] if segments == 1 else tuple(out[i] for i in range(segments))


class SplitActivations:
    """A [..., K] fp32 activation tensor held ONLY as the f16x3 GEMM's A operand: h = f16(x), l = f16((x - h) 2^11), stored per
    (32-row group, 32-wide block of K) as four 1 KiB pieces [h g0 | l g0 | h g1 | l g1], each [k-half][row & 31][8 f16] with
    k = 32 b + 16 half + 8 g + i (csrc/split_linear_h3.h, "PRE").  Same bytes as the fp32 tensor (rows padded to a multiple of 32,
    padding never written); produced by add_layer_norm(frag=True), consumed by linear().  ``shape`` = the logical fp32 shape."""
    __slots__ = ("data", "shape")

    def __init__(self, data, shape):
        self.data, self.shape = data, tuple(shape)

    @staticmethod
    def empty(shape, device):
        K = shape[-1]
        M = 1
        for d in shape[:-1]:
            M *= d
        return SplitActivations(torch.empty((((M + 31) // 32) * 32 * K,), dtype=torch.int32, device=device), shape)

    @property
    def device(self):
        return self.data.device

    def numel(self):
        n = 1
        for d in self.shape:
            n *= d
        return n

    @staticmethod
    def pack(x):
        """Reference packer (torch ops; tests and tools): the image the kernels write for x."""
        K = x.shape[-1]
        x2 = x.reshape(-1, K).float()
        M = x2.shape[0]
        Mp = (M + 31) // 32 * 32
        xp = torch.zeros((Mp, K), dtype=torch.float32, device=x.device)
        xp[:M] = x2
        h = xp.half()
        l = ((xp - h.float()) * 2048.0).half()
        hl = torch.stack([h, l], 0).reshape(2, Mp // 32, 32, K // 32, 2, 2, 8)          # [hl, rg, l31, b, half, g, i]
        img = hl.permute(1, 3, 5, 0, 4, 2, 6).contiguous()                              # [rg, b, g, hl, half, l31, i]
        return SplitActivations(img.reshape(-1).view(torch.int32), x.shape)

    def unpack(self):
        """fp32 values h + l 2^-11 of the image (rows beyond M dropped)."""
        K = self.shape[-1]
        M = self.numel() // K
        Mp = (M + 31) // 32 * 32
        img = self.data.view(torch.float16).reshape(Mp // 32, K // 32, 2, 2, 2, 32, 8)  # [rg, b, g, hl, half, l31, i]
        hl = img.permute(3, 0, 5, 1, 4, 2, 6).reshape(2, Mp, K).float()
        return (hl[0] + hl[1] / 2048.0)[:M].reshape(self.shape)


_SPLIT = threading.local()
_SPLIT_DEFAULT = "f16x3"
"""ops.SPLIT_MODE: arithmetic of the token Linear (K6), PER THREAD (round 5: it was a process global; a re-score on another thread, or a tool
flipping it, changed the kernels of every forward in flight).  Read it as `ops.SPLIT_MODE`, set it for a block with `ops.split_mode(mode)`.  "f16x3": two f16 pieces per operand, three f16 MFMAs per product -- as accurate as an
fp32 GEMM for |x|, |w| < 65504 (beyond f16's range the output is NaN, never silently wrong), twice the speed of "bf16x6": three
bf16 pieces, six MFMAs, fp32's full range."""


def _split_mode():
    return getattr(_SPLIT, "mode", _SPLIT_DEFAULT)


def __getattr__(name):                     # module attribute: ops.SPLIT_MODE = the calling thread's mode
    if name == "SPLIT_MODE":
        return _split_mode()
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


class _OpsModule(type(os)):
    """`ops.SPLIT_MODE = "bf16x6"` was the supported switch until round 4.  Since the mode became per thread an assignment only created a module attribute that
    SHADOWED the getter above: ops.SPLIT_MODE then read "bf16x6" (graph keys, cache names) while every kernel kept dispatching on the thread's real mode -- a NaN
    re-score written the old way was a silent no-op.  Assignment now fails loudly."""

    def __setattr__(self, name, value):
        if name == "SPLIT_MODE":
            raise AttributeError('ops.SPLIT_MODE is read-only (the arithmetic mode is per thread): run the block inside `with ops.split_mode("bf16x6"):`')
        super().__setattr__(name, value)


import sys as _sys  # noqa: E402
_sys.modules[__name__].__class__ = _OpsModule


@_hip_op
def split_weight(weight, mode=None):
    """fp32 weight [N,K] -> the packed operand planes of the token Linear, tiled as the kernel's LDS image (once per weight load):
    mode "bf16x6": three bf16 planes hi, mid, lo (sum exactly `weight`), [N/128, K/16, 3, 128, 2, 8] bf16, the 8-element half h of
    row r in slot h ^ ((r >> 3) & 1);  mode "f16x3": h = f16(w), l = f16((w - h) 2^11), [N/128, K/16, 2, 128, 2, 8] float16 with
    the k order of csrc/split_linear_h3.h.  Default: ops.SPLIT_MODE."""
    lib = _lib.load()
    _chk(weight, "weight", dim=2)
    N, K = weight.shape
    if not split_linear_supported(N, K):
        raise RbaHipError("split_weight needs weight [N,K] with K % 32 == 0")
    mode = _split_mode() if mode is None else mode
    if mode == "f16x3":
        packed = torch.empty(((N + 127) // 128, K // 16, 2, 128, 2, 8), dtype=torch.float16, device=weight.device)
        _lib.check(lib.rba_split_weight_f16x2(_p(weight), _p(packed), N, K, _stream()), "rba_split_weight_f16x2")
        return packed
    if mode != "bf16x6":
        raise RbaHipError(f"unknown split mode {mode!r}")
    packed = torch.empty(((N + 127) // 128, K // 16, 3, 128, 2, 8), dtype=torch.bfloat16, device=weight.device)
    _lib.check(lib.rba_split_weight_bf16x3(_p(weight), _p(packed), N, K, _stream()), "rba_split_weight_bf16x3")
    return packed


@_hip_op
def unpack_split_weight(packed):
    """Inverse of split_weight's tiling: -> planes [3, Np, K] bf16 or [2, Np, K] float16, Np = N rounded up to 128 (for inspection
    and tests)."""
    nt, S, P = packed.shape[:3]
    r = torch.arange(128, device=packed.device)
    flip = ((r >> 3) & 1).bool()
    un = torch.where(flip.view(1, 1, 1, 128, 1, 1), packed.flip(4), packed)          # slot -> half
    if P == 2:        # f16x3: sub-stage 2 b + g, half h holds k = 32 b + 16 h + 8 g + (0..7)
        un = un.reshape(nt, S // 2, 2, P, 128, 2, 8).permute(3, 0, 4, 1, 5, 2, 6)     # [P, nt, row, b, h, g, 8]
        return un.reshape(P, nt * 128, S * 16)
    return un.permute(2, 0, 3, 1, 4, 5).reshape(3, nt * 128, S * 16)


def split_linear_supported(N, K):
    return K % 32 == 0 and N >= 1 and K >= 32


TILES_MIN = 64


def split_linear_pays(M, N, K, gelu=False):
    """Where the bf16x6 kernel beats hipBLASLt's fp32 GEMM on MI355X (tools/gemm_v4_sweep.py, profiles/r02_split_linear.txt):
    1.3-1.7x whenever there are at least 64 tiles of 128 x 128 (below 256 tiles the library switches to 128 x 64 tiles so that
    every CU still gets work) and K >= 64.  Since round 5 this is a statement about speed only: linear() runs the library's own kernels for every shape."""
    if not split_linear_supported(N, K) or K < 64:
        return False
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    return tiles >= TILES_MIN


@_hip_op
def linear(x, lin, use_bias=True, gelu=False, relu=False, residual=None, split_out=False):
    """``F.linear(x, lin.weight, lin.bias)`` [+ exact GELU | ReLU] for an ``nn.Linear`` on a token tensor, through the split
    kernel of the current ops.SPLIT_MODE (weight planes are split once per weight load and cached on the module) -- no library GEMM for any shape.
    ``residual`` (f16x3 form only, see linear_residual_fused): returns ``(residual + x W^T) + bias`` written IN PLACE over `residual`.
    ``split_out`` (f16x3, with gelu): the result is returned as SplitActivations for the next linear() (fc1 -> fc2)."""
    w = lin.weight
    N, K = w.shape
    M = x.numel() // K if K else 0
    bias = lin.bias if use_bias else None
    if isinstance(x, SplitActivations):
        if _split_mode() != "f16x3" or x.shape[-1] != K:
            raise RbaHipError("SplitActivations feed the f16x3 Linear only (check linear_takes_split(M, N, K) before producing them)")
        return split_linear(x, _cached_planes(lin, w), bias, gelu=gelu, out_features=N, relu=relu, residual=residual, split_out=split_out)
    _chk(x, "x") if x.is_contiguous() else _chk(x.contiguous(), "x")          # HIP fp32 tensors only: no CPU path here either
    if split_out:
        if _split_mode() != "f16x3" or not split_linear_supported(N, K):
            raise RbaHipError("linear(split_out=True) needs the f16x3 mode and K % 32 == 0")
        return split_linear(x.contiguous(), _cached_planes(lin, w), bias, gelu=gelu, out_features=N, split_out=True)
    if residual is not None and not linear_residual_fused(M, N, K):
        raise RbaHipError("linear(residual=...) needs the fused f16x3 path: check linear_residual_fused(M, N, K) first")
    if split_linear_supported(N, K):
        # round 5: every shape runs the library's own GEMM (the f16x3 / bf16x6 kernels take any M and N); `split_linear_pays` only says where they beat
        # hipBLASLt, and the small launches it used to send there (tiny test nets, the bf16x6 re-score of the encoder / decoder Linears) are latency-bound anyway
        return split_linear(x.contiguous(), _cached_planes(lin, w), bias, gelu=gelu, out_features=N, relu=relu, residual=residual)
    # K not a multiple of 32 (no layer of the released architectures): zero-pad K once per weight load and per call -- still no library GEMM
    Kp = (K + 31) // 32 * 32
    xp = torch.zeros(tuple(x.shape[:-1]) + (Kp,), dtype=torch.float32, device=x.device)
    xp[..., :K] = x
    return split_linear(xp, _cached_planes(lin, w, pad_k=Kp), bias, gelu=gelu, out_features=N, relu=relu)


def _cached_planes(lin, w, pad_k=None):
    """split_weight(w) of the current split mode, cached on the module per mode (the bf16x6 planes of the non-finite fallback stay
    beside the f16x3 ones: switching modes does not re-split).  pad_k: w zero-padded to that many columns first."""
    key = (w.data_ptr(), w._version, w.device, pad_k)
    caches = getattr(lin, "_rba_planes", None)
    if not isinstance(caches, dict):
        caches = lin._rba_planes = {}
    cache = caches.get(_split_mode())
    if cache is None or cache[0] != key:
        w2 = w.detach().contiguous()
        if pad_k is not None:
            w2 = torch.nn.functional.pad(w2, (0, pad_k - w2.shape[1]))
        cache = caches[_split_mode()] = (key, split_weight(w2))
    return cache[1]


@contextlib.contextmanager
def split_mode(mode):
    """Run a block with ops.SPLIT_MODE = mode ("f16x3" | "bf16x6") ON THE CALLING THREAD, e.g. to re-score an image whose f16x3 result is NaN
    (an activation or weight beyond f16's range) on the full-range kernels.  Other threads of the process keep their own mode; launches already
    enqueued on any stream are not affected (the mode selects kernels at launch time)."""
    if mode not in ("f16x3", "bf16x6"):
        raise RbaHipError(f"unknown split mode {mode!r}")
    prev = _split_mode()
    _SPLIT.mode = mode
    try:
        yield
    finally:
        _SPLIT.mode = prev


SPLIT_ACTIVATIONS = os.environ.get("RBA_SPLIT_ACTIVATIONS", "1") != "0"      # A/B switch (tools): producers keep writing fp32 rows


# smallest K whose Linear takes its A operand as a split image.  Round 2 stopped at K > 256 (the LDS-staged fp32-row kernel served Swin stages
# 1-2); round 3 hands the operand over from K = 128 on: those launches are HBM-bound either way (+0.8 % images/s, tools: RBA_SPLIT_MIN_K)
SPLIT_MIN_K = int(os.environ.get("RBA_SPLIT_MIN_K", "128"))


def linear_takes_split(M, N, K):
    """True when linear() on this shape runs a kernel whose A operand a producer can hand over as SplitActivations: the pipelined
    128-column f16x3 kernel (K >= SPLIT_MIN_K, at least 160 tiles of 128 x 128) or, for smaller launches, the sub-tile kernel of
    csrc/split_linear_h3q.h (K % 64 == 0, K >= 256, N % 32 == 0, at least 32 tiles: Swin stage 4's proj / fc2 with 128 tiles)."""
    if not (SPLIT_ACTIVATIONS and _split_mode() == "f16x3" and K >= SPLIT_MIN_K and K % 32 == 0):
        return False
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    return tiles >= 160 or (tiles >= 32 and K % 64 == 0 and K >= 256 and N % 32 == 0)


MLP_FUSED_MIN_ROWS = 32768          # 256 workgroups of 128 rows: below that the unfused pair's 128 x 128 tiles fill the chip better


def mlp_fused_ok(M, C, hidden):
    """True when mlp_fused() applies: the one-kernel Swin MLP exists for C = 128 (Swin-B stage 1), f16x3 mode."""
    return SPLIT_ACTIVATIONS and _split_mode() == "f16x3" and C == 128 and hidden % 32 == 0 and hidden >= 64 and M >= MLP_FUSED_MIN_ROWS


@_hip_op
def mlp_fused(x, fc1, fc2, residual):
    """residual + fc2(GELU(fc1(x))) in one kernel, in place over `residual` (Mlp + residual add of a Swin block, swin.py:35-41, 293): the
    hidden tensor stays in registers.  Bit-identical to linear(split_out=True) + linear(residual=...)."""
    lib = _lib.load()
    _chk(x, "x")
    _chk(residual, "residual")
    C, hidden = fc1.weight.shape[1], fc1.weight.shape[0]
    M = x.numel() // C
    if (x.shape[-1] != C or tuple(fc2.weight.shape) != (C, hidden) or tuple(residual.shape) != tuple(x.shape) or C != 128 or hidden % 32
            or _split_mode() != "f16x3"):
        raise RbaHipError("mlp_fused needs C == 128, hidden % 32 == 0, matching fc1 / fc2 and the f16x3 mode")
    _lib.check(lib.rba_swin_mlp_fused_f16x3_f32(_p(x), _p(_cached_planes(fc1, fc1.weight)), _p(fc1.bias), _p(_cached_planes(fc2, fc2.weight)),
                                                _p(fc2.bias), _p(residual), _p(residual), M, C, hidden, _stream()),
               "rba_swin_mlp_fused_f16x3_f32")
    return residual


@_hip_op
def mlp_fused_ln(x, norm, fc1, fc2):
    """x <- x + fc2(GELU(fc1(LayerNorm(x)))) IN PLACE in one kernel (swin.py:293): mlp_fused with norm2 computed in the kernel's prologue from the rows it
    loads anyway.  norm = (weight, bias, eps).  Same conditions as mlp_fused (check mlp_fused_ok)."""
    lib = _lib.load()
    _chk(x, "x")
    g, b, eps = norm
    _chk(g, "norm.weight", dim=1)
    _chk(b, "norm.bias", dim=1)
    C, hidden = fc1.weight.shape[1], fc1.weight.shape[0]
    M = x.numel() // C
    if (x.shape[-1] != C or tuple(fc2.weight.shape) != (C, hidden) or C != 128 or hidden % 32 or _split_mode() != "f16x3" or g.numel() != C or b.numel() != C):
        raise RbaHipError("mlp_fused_ln needs C == 128, hidden % 32 == 0, matching fc1 / fc2 / norm and the f16x3 mode")
    _lib.check(lib.rba_swin_mlp_fused_ln_f16x3_f32(_p(x), _p(g), _p(b), float(eps), _p(_cached_planes(fc1, fc1.weight)), _p(fc1.bias),
                                                   _p(_cached_planes(fc2, fc2.weight)), _p(fc2.bias), M, C, hidden, _stream()),
               "rba_swin_mlp_fused_ln_f16x3_f32")
    return x


def linear_residual_fused(M, N, K):
    """True when linear(..., residual=r) runs as ONE kernel (the f16x3 GEMM with the residual add in its epilogue)."""
    return _split_mode() == "f16x3" and split_linear_pays(M, N, K)


@_hip_op
def split_linear(x, planes, bias=None, gelu=False, out_features=None, relu=False, residual=None, split_out=False):
    """F.linear(x, W, bias) [+ exact GELU] with W given as split_weight(W): fp32-accurate on the bf16 / f16 matrix pipe (the planes'
    dtype says which form they were packed for).
    ``out_features`` = N when it is not a multiple of 128 (the packed planes are padded)."""
    lib = _lib.load()
    pre = isinstance(x, SplitActivations)
    if pre:
        _chk(x.data, "x.data", dtype=torch.int32, dim=1)
    else:
        _chk(x, "x")
    f16 = planes.dtype == torch.float16
    _chk(planes, "planes", dtype=torch.float16 if f16 else torch.bfloat16, dim=6)
    K = x.shape[-1]
    N = planes.shape[0] * 128 if out_features is None else int(out_features)
    M = x.numel() // K if K else 0
    if (tuple(planes.shape[2:]) != ((2, 128, 2, 8) if f16 else (3, 128, 2, 8)) or planes.shape[1] * 16 != K
            or not split_linear_supported(N, K) or (N + 127) // 128 != planes.shape[0]):
        raise RbaHipError("split_linear needs x [..., K] and split_weight(W) of a weight [N,K] with K % 32 == 0")
    if bias is not None:
        _chk(bias, "bias", dim=1)
        if bias.numel() != N:
            raise RbaHipError("bias must have N elements")
    act = 1 if gelu else (2 if relu else 0)
    if split_out:                            # GELU(x W^T + bias) handed to the next Linear as its split A operand
        if not f16 or act != 1 or residual is not None or N % 32:
            raise RbaHipError("split_out needs f16x3 planes, gelu=True, no residual and N % 32 == 0")
        out = SplitActivations.empty(tuple(x.shape[:-1]) + (N,), x.device)
        _lib.check(lib.rba_split_linear_f16x3_gelu_split_out(_p(x.data) if pre else _p(x), 1 if pre else 0, _p(planes), _p(bias), _p(out.data),
                                                             M, N, K, _stream()), "rba_split_linear_f16x3_gelu_split_out")
        return out
    if residual is not None:                 # out = (residual + x W^T) + bias, in place over `residual` (f16x3 planes, no activation)
        _chk(residual, "residual")
        if not f16 or act or tuple(residual.shape) != tuple(x.shape[:-1]) + (N,):
            raise RbaHipError("residual needs f16x3 planes, no activation and a [..., N] tensor")
        if pre:
            _lib.check(lib.rba_split_linear_f16x3_frag_f32(_p(x.data), _p(planes), _p(bias), _p(residual), _p(residual), M, N, K, 0, _stream()),
                       "rba_split_linear_f16x3_frag_f32")
            return residual
        _lib.check(lib.rba_split_linear_f16x3_res_f32(_p(x), _p(planes), _p(bias), _p(residual), _p(residual), M, N, K, _stream()),
                   "rba_split_linear_f16x3_res_f32")
        return residual
    out = torch.empty(tuple(x.shape[:-1]) + (N,), dtype=torch.float32, device=x.device)
    if pre:
        if not f16:
            raise RbaHipError("SplitActivations need f16x3 planes")
        _lib.check(lib.rba_split_linear_f16x3_frag_f32(_p(x.data), _p(planes), _p(bias), 0, _p(out), M, N, K, act, _stream()),
                   "rba_split_linear_f16x3_frag_f32")
        return out
    if f16:
        _lib.check(lib.rba_split_linear_f16x3_f32(_p(x), _p(planes), _p(bias), _p(out), M, N, K, act, _stream()), "rba_split_linear_f16x3_f32")
    else:
        _lib.check(lib.rba_split_linear_f32(_p(x), _p(planes), _p(bias), _p(out), M, N, K, act, _stream()), "rba_split_linear_f32")
    return out


@_hip_op
def split_linear_nchw_out(x, planes, bias, rows_per_image, out_features=None):
    """x [B*P, K] (NHWC rows) -> [B, N, P]: the Linear of split_linear written channel-major (NHWC in, NCHW out); the planes' dtype says
    for which arithmetic form they were packed (float16: f16x3, bfloat16: bf16x6)."""
    lib = _lib.load()
    _chk(x, "x", dim=2)
    f16 = planes.dtype == torch.float16
    _chk(planes, "planes", dtype=torch.float16 if f16 else torch.bfloat16, dim=6)
    M, K = x.shape
    N = planes.shape[0] * 128 if out_features is None else int(out_features)
    if (tuple(planes.shape[2:]) != ((2, 128, 2, 8) if f16 else (3, 128, 2, 8)) or planes.shape[1] * 16 != K or (N + 127) // 128 != planes.shape[0]
            or rows_per_image < 1 or M % rows_per_image):
        raise RbaHipError("split_linear_nchw_out needs x [B*P, K], split_weight(W [N,K]) and M % rows_per_image == 0")
    if bias is not None:
        _chk(bias, "bias", dim=1)
        if bias.numel() != N:
            raise RbaHipError("bias must have N elements")
    out = torch.empty((M // rows_per_image, N, rows_per_image), dtype=torch.float32, device=x.device)
    fn, name = ((lib.rba_split_linear_nchw_out_f16x3_f32, "rba_split_linear_nchw_out_f16x3_f32") if f16
                else (lib.rba_split_linear_nchw_out_f32, "rba_split_linear_nchw_out_f32"))
    _lib.check(fn(_p(x), _p(planes), _p(bias), _p(out), M, N, K, rows_per_image, _stream()), name)
    return out


GN_MOMENTS = os.environ.get("RBA_GN_MOMENTS", "1") != "0"      # A/B switch (tools): 0 keeps the separate statistics passes of the FPN's GroupNorms


def _gn_moments_ok(rows_per_image, N, num_groups):
    return (GN_MOMENTS and _split_mode() == "f16x3" and rows_per_image % 128 == 0 and N % 128 == 0 and N % num_groups == 0
            and (N // num_groups) in (4, 8, 16, 32))


def linear_emits_gn_moments(M, N, K, rows_per_image, num_groups):
    """True when linear_gn_stats runs as ONE GEMM whose epilogue leaves the GroupNorm moments of its output (the LDS-staged f16x3 kernel: K <= 256, 128-column tiles)."""
    return _gn_moments_ok(rows_per_image, N, num_groups) and K <= 256 and K % 32 == 0 and ((M + 127) // 128) * ((N + 127) // 128) >= 160 and M % rows_per_image == 0


def conv3x3_emits_gn_moments(B, H, W, N, num_groups):
    """True when conv3x3_nhwc_gn_stats runs the split-image convolution with the moment epilogue."""
    return _gn_moments_ok(H * W, N, num_groups) and conv3x3_takes_split(B * H * W, N)


def _merge_gn_moments(lib, moments, B, G, splits, eps):
    mr = torch.empty((B, G, 2), dtype=torch.float32, device=moments.device)
    _lib.check(lib.rba_group_norm_nhwc_merge_f32(_p(moments), _p(mr), B, G, splits, float(eps), _stream()), "rba_group_norm_nhwc_merge_f32")
    return mr


@_hip_op
def linear_gn_stats(x, lin, num_groups, eps, rows_per_image, use_bias=True):
    """(y, mr): y = F.linear(x, lin.weight[, lin.bias]) on tokens x [B, P, K] and mr [B, G, 2] = group_norm_nhwc_stats(y) -- the statistics come out of the
    GEMM's epilogue (per-tile moments, merged in double) instead of a second pass over y.  y is bit-identical to linear(x, lin); mr equals the stats pass up
    to the summation order.  Check linear_emits_gn_moments first."""
    lib = _lib.load()
    _chk(x, "x")
    w = lin.weight
    N, K = w.shape
    M = x.numel() // K
    if not linear_emits_gn_moments(M, N, K, rows_per_image, num_groups) or x.shape[-1] != K:
        raise RbaHipError("linear_gn_stats: check linear_emits_gn_moments(M, N, K, rows_per_image, G) first")
    bias = lin.bias if use_bias else None
    planes = _cached_planes(lin, w)
    B, splits = M // rows_per_image, rows_per_image // 128
    out = torch.empty(tuple(x.shape[:-1]) + (N,), dtype=torch.float32, device=x.device)
    mom = torch.empty((B, num_groups, splits, 3), dtype=torch.float32, device=x.device)
    _lib.check(lib.rba_split_linear_f16x3_gn_moments_f32(_p(x), _p(planes), _p(bias), _p(out), M, N, K, rows_per_image, int(num_groups), _p(mom), _stream()),
               "rba_split_linear_f16x3_gn_moments_f32")
    return out, _merge_gn_moments(lib, mom, B, int(num_groups), splits, eps)


@_hip_op
def conv3x3_nhwc_gn_stats(x, planes, num_groups, eps, bias=None, out_features=None):
    """(y, mr): conv3x3_nhwc(x) of SplitActivations x [B,H,W,C] and the GroupNorm statistics of y [B, G, 2] from the convolution's own epilogue.
    Check conv3x3_emits_gn_moments first."""
    lib = _lib.load()
    if not isinstance(x, SplitActivations) or len(x.shape) != 4:
        raise RbaHipError("conv3x3_nhwc_gn_stats needs SplitActivations of logical shape [B,H,W,C]")
    _chk(x.data, "x.data", dtype=torch.int32, dim=1)
    _chk(planes, "planes", dtype=torch.float16, dim=6)
    B, H, W, C = x.shape
    N = planes.shape[0] * 128 if out_features is None else int(out_features)
    if (tuple(planes.shape[2:]) != (2, 128, 2, 8) or planes.shape[1] * 16 != 9 * C or C % 32 or (N + 127) // 128 != planes.shape[0]
            or not conv3x3_emits_gn_moments(B, H, W, N, num_groups)):
        raise RbaHipError("conv3x3_nhwc_gn_stats: check conv3x3_emits_gn_moments(B, H, W, N, G) first")
    if bias is not None:
        _chk(bias, "bias", dim=1)
        if bias.numel() != N:
            raise RbaHipError("bias must have N elements")
    splits = H * W // 128
    out = torch.empty((B, H, W, N), dtype=torch.float32, device=x.device)
    mom = torch.empty((B, num_groups, splits, 3), dtype=torch.float32, device=x.device)
    _lib.check(lib.rba_conv3x3_nhwc_f16x3_split_in_gn_moments_f32(_p(x.data), _p(planes), _p(bias), _p(out), B, H, W, C, N, int(num_groups), _p(mom),
                                                                  _stream()), "rba_conv3x3_nhwc_f16x3_split_in_gn_moments_f32")
    return out, _merge_gn_moments(lib, mom, B, int(num_groups), splits, eps)


def split_linear_nchw_out_takes_gn(planes, rows_per_image, K, num_groups):
    """True when split_linear_nchw_out_gn applies: f16x3 planes, whole 128-row tiles per image, four-channel chunks inside one group."""
    return planes.dtype == torch.float16 and rows_per_image % 128 == 0 and K % num_groups == 0 and (K // num_groups) % 4 == 0 and K % 32 == 0


@_hip_op
def split_linear_nchw_out_gn(x, mr, weight, bias_gn, num_groups, relu, planes, bias, rows_per_image, out_features=None):
    """split_linear_nchw_out(group_norm_nhwc(x) [+ ReLU]) with the normalisation applied while the rows are loaded (f16x3 planes only): x [B*P, K] is the
    RAW convolution output, mr [B, G, 2] its statistics (group_norm_nhwc_stats), weight / bias_gn the GroupNorm's affine.  Bit-identical to the two-call form."""
    lib = _lib.load()
    _chk(x, "x", dim=2)
    _chk(mr, "mr", dim=3)
    _chk(weight, "weight", dim=1)
    _chk(bias_gn, "bias_gn", dim=1)
    _chk(planes, "planes", dtype=torch.float16, dim=6)
    M, K = x.shape
    N = planes.shape[0] * 128 if out_features is None else int(out_features)
    if (tuple(planes.shape[2:]) != (2, 128, 2, 8) or planes.shape[1] * 16 != K or (N + 127) // 128 != planes.shape[0] or rows_per_image < 1
            or M % rows_per_image or not split_linear_nchw_out_takes_gn(planes, rows_per_image, K, num_groups)):
        raise RbaHipError("split_linear_nchw_out_gn needs x [B*P, K], f16x3 split_weight(W [N,K]), P % 128 == 0 and (K / G) % 4 == 0")
    B = M // rows_per_image
    if tuple(mr.shape) != (B, num_groups, 2) or weight.numel() != K or bias_gn.numel() != K:
        raise RbaHipError("mr must be [B, G, 2]; weight / bias_gn must have K elements")
    if bias is not None:
        _chk(bias, "bias", dim=1)
        if bias.numel() != N:
            raise RbaHipError("bias must have N elements")
    out = torch.empty((B, N, rows_per_image), dtype=torch.float32, device=x.device)
    _lib.check(lib.rba_split_linear_nchw_out_gn_f16x3_f32(_p(x), _p(mr), _p(weight), _p(bias_gn), int(num_groups), int(bool(relu)), _p(planes), _p(bias),
                                                          _p(out), M, N, K, rows_per_image, _stream()), "rba_split_linear_nchw_out_gn_f16x3_f32")
    return out


@_hip_op
def conv3x3_weight(weight, mode=None):
    """conv weight [N, C, 3, 3] -> split_weight (form `mode`, default ops.SPLIT_MODE) of the implicit-GEMM matrix [N, 9 C],
    k = (3 ky + kx) C + c."""
    _chk(weight, "weight", dim=4)
    N, C, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or C % 32:
        raise RbaHipError("conv3x3_weight needs a [N, C, 3, 3] weight with C % 32 == 0")
    return split_weight(weight.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous(), mode=mode)


def conv3x3_takes_split(M, N):
    """True when conv3x3_nhwc can read its input as SplitActivations (the pipelined f16x3 kernel: at least 256 tiles of 128 x 128)."""
    return SPLIT_ACTIVATIONS and _split_mode() == "f16x3" and ((M + 127) // 128) * ((N + 127) // 128) >= 256


@_hip_op
def conv3x3_nhwc(x, planes, bias=None, out_features=None):
    """3x3, stride 1, pad 1 convolution of NHWC x [B,H,W,C] with conv3x3_weight(W) -> [B,H,W,N] (implicit GEMM on K6; the planes'
    dtype says for which form they were packed)."""
    lib = _lib.load()
    if not isinstance(x, SplitActivations):
        _chk(x, "x", dim=4)
    f16 = planes.dtype == torch.float16
    _chk(planes, "planes", dtype=torch.float16 if f16 else torch.bfloat16, dim=6)
    pre = isinstance(x, SplitActivations)
    if pre:
        _chk(x.data, "x.data", dtype=torch.int32, dim=1)
        if len(x.shape) != 4:
            raise RbaHipError("conv3x3_nhwc needs SplitActivations of logical shape [B,H,W,C]")
    B, H, W, C = x.shape
    N = planes.shape[0] * 128 if out_features is None else int(out_features)
    if (tuple(planes.shape[2:]) != ((2, 128, 2, 8) if f16 else (3, 128, 2, 8)) or planes.shape[1] * 16 != 9 * C or C % 32
            or (N + 127) // 128 != planes.shape[0]):
        raise RbaHipError("conv3x3_nhwc needs x [B,H,W,C] (C % 32 == 0) and conv3x3_weight(W [N,C,3,3])")
    if bias is not None:
        _chk(bias, "bias", dim=1)
        if bias.numel() != N:
            raise RbaHipError("bias must have N elements")
    out = torch.empty((B, H, W, N), dtype=torch.float32, device=x.device)
    if pre:
        if not f16 or not conv3x3_takes_split(B * H * W, N):
            raise RbaHipError("conv3x3_nhwc: SplitActivations need f16x3 planes and conv3x3_takes_split(B*H*W, N)")
        _lib.check(lib.rba_conv3x3_nhwc_f16x3_split_in_f32(_p(x.data), _p(planes), _p(bias), _p(out), B, H, W, C, N, _stream()),
                   "rba_conv3x3_nhwc_f16x3_split_in_f32")
        return out
    fn, name = ((lib.rba_conv3x3_nhwc_f16x3_f32, "rba_conv3x3_nhwc_f16x3_f32") if f16 else (lib.rba_conv3x3_nhwc_f32, "rba_conv3x3_nhwc_f32"))
    _lib.check(fn(_p(x), _p(planes), _p(bias), _p(out), B, H, W, C, N, _stream()), name)
    return out


_CONCURRENT_STREAMS = {}        # library handle -> last value set (the hint is per loaded library)


def set_concurrent_streams(n):
    """Tell the kernel library how many HIP streams of this process run forwards at the same time (default 1): with two or more, the K6 launches
    that would fill only half the chip use whole-CU workgroups and leave the other CUs to the other streams (include/rba_hip.h).  Speed only --
    results are bit-identical.  Returns the previous setting.  Call it before the streams start launching (and before capturing hipGraphs: a
    graph replays the launch forms it was captured with; MaskFormer keys its graphs on concurrent_streams()).  Restore it in a `finally`."""
    lib = _lib.load()
    n = max(1, int(n))
    prev = int(lib.rba_set_concurrent_streams(n))
    _CONCURRENT_STREAMS[id(lib)] = n
    return prev


def concurrent_streams():
    """The hint last given to set_concurrent_streams for the current library (1 if never set)."""
    return _CONCURRENT_STREAMS.get(id(_lib.load()), 1)


# ---------------------------------------------------------------------------------------------------------------- small token Linears
TOKEN_LINEAR = os.environ.get("RBA_TOKEN_LINEAR", "1") != "0"      # A/B switch (tools): 0 sends the small Linears back to hipBLASLt + separate kernels


def token_linear_ok(N, K):
    """Shapes the row-complete token Linear serves (csrc/token_linear.hip): N <= 256 outputs, K a multiple of 32; f16x3 arithmetic only
    (under ops.split_mode("bf16x6"), the full-range fallback, these Linears run the bf16x6 K6 kernel instead -- no library GEMM on any path since round 5)."""
    return TOKEN_LINEAR and _split_mode() == "f16x3" and 1 <= N <= 256 and K >= 32 and K % 32 == 0


TOKEN_MAX_ROWS = 8192      # beyond that a 128 x 128-tiled GEMM re-reads the weight far less often than one workgroup per 16 rows does


def token_linear_pays(M, N, K):
    """True when the row-complete kernel is the launch to use for an [M, K] x [N, K]^T Linear: the encoder / decoder-memory token counts of one
    image (2 048 at C2, 4 830 at C5) -- its callers then also fold `+ pos`, sibling Linears and the post-norm residual step into the launch."""
    return token_linear_ok(N, K) and N % 4 == 0 and 1 <= M <= TOKEN_MAX_ROWS


def _token_planes(lin):
    """fragment-ordered f16 (h, l) image of lin.weight, packed once per weight load and cached on the module"""
    w = lin.weight
    key = (w.data_ptr(), w._version, w.device)
    cache = getattr(lin, "_rba_token_planes", None)
    if cache is None or cache[0] != key:
        lib = _lib.load()
        w2 = w.detach().contiguous()
        _chk(w2, "weight", dim=2)
        N, K = w2.shape
        packed = torch.empty((((N + 15) // 16) * (K // 32) * 128, 4), dtype=torch.int32, device=w.device)
        _lib.check(lib.rba_token_linear_pack_f16x2(_p(w2), _p(packed), N, K, _stream()), "rba_token_linear_pack_f16x2")
        cache = (key, packed)
        try:
            lin._rba_token_planes = cache
        except AttributeError:                      # objects with __slots__ (none today): no caching
            pass
    return cache[1]


@_hip_op
def token_linear(x, lin, x_add=None, use_bias=True, relu=False, residual=None, norm=None):
    """``F.linear(x + x_add, lin.weight, lin.bias)`` [+ ReLU] on a token tensor [..., K] with N <= 256 outputs, as one launch of the
    row-complete kernel; with ``residual`` and ``norm`` (an nn.LayerNorm over N): ``norm(residual + F.linear(x, W, b))`` -- the post-norm
    residual step of an encoder layer (msdeformattn.py:134-138) in the GEMM's epilogue.  Check token_linear_ok(N, K) first."""
    lib = _lib.load()
    _chk(x, "x")
    N, K = lin.weight.shape
    if x.shape[-1] != K or not token_linear_ok(N, K):
        raise RbaHipError("token_linear: x [..., K] with K % 32 == 0, N <= 256 and the f16x3 mode (see token_linear_ok)")
    M = x.numel() // K
    if x_add is not None:
        _chk(x_add, "x_add")
        if tuple(x_add.shape) != tuple(x.shape):
            raise RbaHipError("x_add must have x's shape")
    bias = lin.bias if use_bias else None
    if bias is not None:
        _chk(bias, "bias", dim=1)
    out = torch.empty(tuple(x.shape[:-1]) + (N,), dtype=torch.float32, device=x.device)
    if (residual is None) != (norm is None):
        raise RbaHipError("token_linear: residual and norm come together")
    if norm is not None:
        _chk(residual, "residual")
        if tuple(residual.shape) != tuple(out.shape) or N % 16 or relu or norm.weight.numel() != N:
            raise RbaHipError("token_linear(norm=...): residual [..., N], N % 16 == 0, no activation")
        _lib.check(lib.rba_token_linear_f32(_p(x), _p(x_add), _p(_token_planes(lin)), _p(bias), _p(residual), _p(norm.weight), _p(norm.bias),
                                            float(norm.eps), _p(out), M, N, K, 0, _stream()), "rba_token_linear_f32")
        return out
    _lib.check(lib.rba_token_linear_f32(_p(x), _p(x_add), _p(_token_planes(lin)), _p(bias), None, None, None, 0.0, _p(out), M, N, K,
                                        2 if relu else 0, _stream()), "rba_token_linear_f32")
    return out


@_hip_op
def token_linear_multi(x, specs):
    """Up to three Linears over the same rows in ONE launch.  specs: [(lin, x_add | None, out | None, col0, relu), ...]; problem i computes
    F.linear(x + x_add_i, W_i, b_i) [ReLU] and writes it into out_i[..., col0 : col0 + N_i] (out_i = a fresh [..., N_i] tensor when None; several
    problems may fill column slices of one wider tensor).  Returns the list of output tensors.  Check token_linear_ok(N_i, K) first."""
    lib = _lib.load()
    _chk(x, "x")
    K = x.shape[-1]
    M = x.numel() // K if K else 0
    if not 1 <= len(specs) <= 3:
        raise RbaHipError("token_linear_multi: one to three problems")
    arr = (_lib.TokenLinearProblem * len(specs))()
    outs, keep = [], []
    for i, (lin, x_add, out, col0, relu) in enumerate(specs):
        N, K_ = lin.weight.shape
        if K_ != K or not token_linear_ok(N, K) or N % 4:
            raise RbaHipError("token_linear_multi: weights [N <= 256, N % 4 == 0, K] with x's K % 32 == 0 (see token_linear_ok)")
        if x_add is not None:
            _chk(x_add, "x_add")
            if tuple(x_add.shape) != tuple(x.shape):
                raise RbaHipError("x_add must have x's shape")
        if out is None:
            out, col0 = torch.empty(tuple(x.shape[:-1]) + (N,), dtype=torch.float32, device=x.device), 0
        else:
            _chk(out, "out")
            if tuple(out.shape[:-1]) != tuple(x.shape[:-1]) or col0 % 4 or col0 + N > out.shape[-1] or out.shape[-1] % 4:
                raise RbaHipError("token_linear_multi: out [..., ld] with ld % 4 == 0 and a column slice col0 % 4 == 0 inside it")
        if lin.bias is not None:
            _chk(lin.bias, "bias", dim=1)
        planes = _token_planes(lin)
        keep.append(planes)
        arr[i] = _lib.TokenLinearProblem(_p(x_add) or None, _p(planes), _p(lin.bias) or None, out.data_ptr() + 4 * col0, N, out.shape[-1], 2 if relu else 0)
        outs.append(out)
    _lib.check(lib.rba_token_linear_multi_f32(_p(x), ctypes.byref(arr), len(specs), M, K, _stream()), "rba_token_linear_multi_f32")
    return outs


@_hip_op
def patch_im2col(image, mean, std, Hp, Wp):
    """image [3,h,w] (uint8 or fp32) -> [(Hp/4)*(Wp/4), 64] fp32: (image - mean) / std, zero padding to Hp x Wp (ImageList semantics:
    the padding is zero AFTER normalisation) and the im2col of the 4x4 / stride-4 patch convolution in one pass; column
    c*16 + ky*4 + kx, columns 48..63 zero.  `mean`, `std`: three python floats each."""
    import ctypes
    lib = _lib.load()
    if image.dtype not in (torch.uint8, torch.float32):
        raise RbaHipError("image must be uint8 or float32")
    _chk(image, "image", dtype=image.dtype, dim=3)
    if image.shape[0] != 3 or Hp % 4 or Wp % 4 or Hp < image.shape[1] or Wp < image.shape[2] or len(mean) != 3 or len(std) != 3:
        raise RbaHipError("patch_im2col needs a [3,h,w] image and Hp >= h, Wp >= w multiples of 4")
    h, w = int(image.shape[1]), int(image.shape[2])
    out = torch.empty(((Hp // 4) * (Wp // 4), 64), dtype=torch.float32, device=image.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    sd = (ctypes.c_float * 3)(*[float(v) for v in std])
    fn, name = ((lib.rba_patch_im2col_u8, "rba_patch_im2col_u8") if image.dtype == torch.uint8 else (lib.rba_patch_im2col_f32, "rba_patch_im2col_f32"))
    _lib.check(fn(_p(image), _p(out), h, w, int(Hp), int(Wp), ctypes.addressof(m), ctypes.addressof(sd), _stream()), name)
    return out


@_hip_op
def group_norm_nhwc(x, num_groups, weight, bias, eps=1e-5, relu=False):
    """GroupNorm (+ReLU) of channels-last x [B, P, C] (or [B, H, W, C]): the same operator as group_norm on the token layout."""
    lib = _lib.load()
    _chk(x, "x")
    _chk(weight, "weight", dim=1)
    _chk(bias, "bias", dim=1)
    if x.dim() not in (3, 4):
        raise RbaHipError("x must be [B,P,C] or [B,H,W,C]")
    B, C = x.shape[0], x.shape[-1]
    P = x.numel() // (B * C) if B * C else 0
    if C % num_groups or weight.numel() != C or bias.numel() != C or (C // num_groups) % 4 or C > 1024 or 256 % (C // 4):
        raise RbaHipError("group_norm_nhwc needs C % G == 0, (C/G) % 4 == 0, C <= 1024 and 256 % (C/4) == 0")
    nbytes = lib.rba_group_norm_nhwc_workspace_bytes(B, P, C, num_groups)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    _lib.check(lib.rba_group_norm_nhwc_f32(_p(x), _p(weight), _p(bias), _p(y), _p(ws), B, P, C, num_groups, float(eps),
                                           int(bool(relu)), _stream()), "rba_group_norm_nhwc_f32")
    return y


@_hip_op
def resample_bilinear_nhwc(x, size, add=None, split_into=None, image=0):
    """F.interpolate(mode="bilinear", align_corners=False) of channels-last x [h, w, C] -> [H, W, C], optional fused `+ add`.
    ``split_into``: a SplitActivations of logical shape [B, H, W, C] (SplitActivations.empty): the result becomes image `image` of it
    (the 3x3 convolution's split operand, conv3x3_takes_split) and nothing else is written; returns split_into."""
    lib = _lib.load()
    _chk(x, "x", dim=3)
    h, w, C = x.shape
    H, W = int(size[0]), int(size[1])
    if C % 4:
        raise RbaHipError("resample_bilinear_nhwc needs C % 4 == 0")
    if add is not None:
        _chk(add, "add")
        if tuple(add.shape) != (H, W, C):
            raise RbaHipError("add must have the output's shape")
    if split_into is not None:
        if (not isinstance(split_into, SplitActivations) or len(split_into.shape) != 4 or tuple(split_into.shape[1:]) != (H, W, C) or C % 32
                or not 0 <= image < split_into.shape[0]):
            raise RbaHipError("split_into must be SplitActivations of shape [B, H, W, C] (C % 32 == 0) with 0 <= image < B")
        _lib.check(lib.rba_resample_bilinear_nhwc_split_out_f32(_p(x), _p(add), _p(split_into.data), C, h, w, H, W, image * H * W, _stream()),
                   "rba_resample_bilinear_nhwc_split_out_f32")
        return split_into
    out = torch.empty((H, W, C), dtype=torch.float32, device=x.device)
    _lib.check(lib.rba_resample_bilinear_nhwc_f32(_p(x), _p(add), _p(out), C, h, w, H, W, _stream()),
               "rba_resample_bilinear_nhwc_f32")
    return out


def _gn_nhwc_ok(C, num_groups):
    return C % num_groups == 0 and (C // num_groups) % 4 == 0 and C <= 1024 and 256 % (C // 4) == 0


@_hip_op
def group_norm_nhwc_stats(x, num_groups, eps=1e-5):
    """(mean, rstd) per image and group of channels-last x [B, P, C] (or [B, H, W, C]) -> [B, G, 2]: the statistics half of group_norm_nhwc, for a
    consumer that folds the normalisation into its own loads (resample_bilinear_nhwc_gn)."""
    lib = _lib.load()
    _chk(x, "x")
    if x.dim() not in (3, 4):
        raise RbaHipError("x must be [B,P,C] or [B,H,W,C]")
    B, C = x.shape[0], x.shape[-1]
    P = x.numel() // (B * C) if B * C else 0
    if not _gn_nhwc_ok(C, num_groups):
        raise RbaHipError("group_norm_nhwc_stats needs C % G == 0, (C/G) % 4 == 0, C <= 1024 and 256 % (C/4) == 0")
    nbytes = lib.rba_group_norm_nhwc_workspace_bytes(B, P, C, num_groups)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x.device)
    mr = torch.empty((B, num_groups, 2), dtype=torch.float32, device=x.device)
    _lib.check(lib.rba_group_norm_nhwc_stats_f32(_p(x), _p(mr), _p(ws), B, P, C, num_groups, float(eps), _stream()), "rba_group_norm_nhwc_stats_f32")
    return mr


@_hip_op
def resample_bilinear_nhwc_gn(x, size, add, num_groups, x_norm=None, add_norm=None, split_into=None, image=0):
    """The FPN top-down step `GroupNorm(add) + F.interpolate(ReLU(GroupNorm(x)))` with the normalisations folded into the loads: x [h, w, C] and
    add [H, W, C] are the RAW convolution outputs; ``x_norm`` = (mr [G, 2], weight, bias, relu) or None (x is used as it is), ``add_norm`` =
    (mr, weight, bias) or None.  The arithmetic of group_norm_nhwc + resample_bilinear_nhwc(add=...) (equal up to fma contraction: one ulp).  ``split_into`` / ``image`` as there."""
    lib = _lib.load()
    _chk(x, "x", dim=3)
    _chk(add, "add", dim=3)
    h, w, C = x.shape
    H, W = int(size[0]), int(size[1])
    if tuple(add.shape) != (H, W, C) or not _gn_nhwc_ok(C, num_groups):
        raise RbaHipError("resample_bilinear_nhwc_gn: add must be [H, W, C]; C % G == 0, (C/G) % 4 == 0")

    def unpack(n, with_relu):
        if n is None:
            return 0, 0, 0, 0
        mr, wgt, b = n[0], n[1], n[2]
        _chk(mr, "mr")
        _chk(wgt, "weight", dim=1)
        _chk(b, "bias", dim=1)
        if mr.numel() != 2 * num_groups or wgt.numel() != C or b.numel() != C:
            raise RbaHipError("norm = (mr [G, 2], weight [C], bias [C])")
        return _p(mr), _p(wgt), _p(b), (int(bool(n[3])) if with_relu and len(n) > 3 else 0)
    xm, xg, xb, xr = unpack(x_norm, True)
    am, ag, ab, _ = unpack(add_norm, False)
    if split_into is not None:
        if (not isinstance(split_into, SplitActivations) or len(split_into.shape) != 4 or tuple(split_into.shape[1:]) != (H, W, C) or C % 32
                or not 0 <= image < split_into.shape[0]):
            raise RbaHipError("split_into must be SplitActivations of shape [B, H, W, C] (C % 32 == 0) with 0 <= image < B")
        _lib.check(lib.rba_resample_bilinear_nhwc_gn_f32(_p(x), xm, xg, xb, xr, _p(add), am, ag, ab, _p(split_into.data), 1, C, num_groups, h, w, H, W,
                                                         image * H * W, _stream()), "rba_resample_bilinear_nhwc_gn_f32")
        return split_into
    out = torch.empty((H, W, C), dtype=torch.float32, device=x.device)
    _lib.check(lib.rba_resample_bilinear_nhwc_gn_f32(_p(x), xm, xg, xb, xr, _p(add), am, ag, ab, _p(out), 0, C, num_groups, h, w, H, W, 0, _stream()),
               "rba_resample_bilinear_nhwc_gn_f32")
    return out


@_hip_op
def bn_relu_conv1x1(x, scale, shift, weight, bias=None):
    """conv1x1(relu(x * scale[c] + shift[c])) for x [B,C,...] -> [B,O,...] (O = weight.shape[0] in {1,2,4}): the DenseHybrid
    `ood_pred` head BNReluConv(hidden_dim, 2, k=1) in eval mode (mask2former_transformer_decoder.py:216-230, 467-468)."""
    lib = _lib.load()
    _chk(x, "x")
    _chk(scale, "scale", dim=1)
    _chk(shift, "shift", dim=1)
    _chk(weight, "weight", dim=2)
    if x.dim() < 3:
        raise RbaHipError("x must be [B,C,...]")
    B, C = x.shape[:2]
    O = weight.shape[0]
    P = x.numel() // (B * C) if B * C else 0
    if scale.numel() != C or shift.numel() != C or weight.shape[1] != C or O not in (1, 2, 4):
        raise RbaHipError("bn_relu_conv1x1 needs scale/shift [C] and weight [O,C] with O in {1,2,4}")
    if bias is not None:
        _chk(bias, "bias", dim=1)
    out = torch.empty((B, O) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    _lib.check(lib.rba_bn_relu_conv1x1_f32(_p(x), _p(scale), _p(shift), _p(weight), _p(bias), _p(out), B, C, O, P, _stream()),
               "rba_bn_relu_conv1x1_f32")
    return out


@_hip_op
def resample_bilinear_ac(x, size):
    """F.interpolate(x, size, mode="bilinear", align_corners=True) for x [C,h,w] or [B,C,h,w] (maskformer_model.py:305)."""
    lib = _lib.load()
    _chk(x, "x")
    if x.dim() not in (3, 4):
        raise RbaHipError("x must be [C,h,w] or [B,C,h,w]")
    H, W = int(size[0]), int(size[1])
    lead = x.shape[:-2]
    C = 1
    for s_ in lead:
        C *= int(s_)
    h, w = x.shape[-2:]
    out = torch.empty(tuple(lead) + (H, W), dtype=torch.float32, device=x.device)
    _lib.check(lib.rba_resample_bilinear_ac_f32(_p(x), _p(out), C, h, w, H, W, _stream()), "rba_resample_bilinear_ac_f32")
    return out


@_hip_op
def gaussian_blur(score, kernel_size=7, sigma=1.0):
    """transforms.GaussianBlur(kernel_size, sigma) of a score map [H,W] (reflect padding): the evaluator's optional smoothing."""
    lib = _lib.load()
    _chk(score, "score", dim=2)
    H, W = score.shape
    if kernel_size % 2 == 0 or not (1 <= kernel_size <= 15) or sigma <= 0 or (H and W and (H <= kernel_size // 2 or W <= kernel_size // 2)):
        raise RbaHipError("gaussian_blur needs an odd kernel_size <= 15, sigma > 0 and a map larger than the padding")
    out = torch.empty_like(score)
    _lib.check(lib.rba_gaussian_blur_f32(_p(score), _p(out), H, W, int(kernel_size), float(sigma), _stream()), "rba_gaussian_blur_f32")
    return out


@_hip_op
def quad_mean(v):
    """v [..., 4, S] -> ((v0 + v1) + (v2 + v3)) * 0.25 [..., S]: the bilinear sample at the centre of each 2 x 2 cell (rba_quad_mean_f32)"""
    lib = _lib.load()
    _chk(v, "v")
    if v.dim() < 2 or v.shape[-2] != 4:
        raise RbaHipError(f"quad_mean expects [..., 4, S], got {tuple(v.shape)}")
    S = v.shape[-1]
    out = torch.empty(v.shape[:-2] + (S,), dtype=torch.float32, device=v.device)
    _lib.check(lib.rba_quad_mean_f32(_p(v), _p(out), v.numel() // (4 * S) if S else 0, S, _stream()), "rba_quad_mean_f32")
    return out


@_hip_op
def softmax_drop_last(logits):
    """F.softmax(logits, -1)[..., :-1] as one launch (rba_softmax_drop_last_f32): [..., K + 1] -> [..., K] contiguous, K + 1 <= 64"""
    lib = _lib.load()
    _chk(logits, "logits")
    K1 = logits.shape[-1]
    if not 2 <= K1 <= 64:
        raise RbaHipError(f"softmax_drop_last supports 2..64 classes, got {K1}")
    out = torch.empty(logits.shape[:-1] + (K1 - 1,), dtype=torch.float32, device=logits.device)
    _lib.check(lib.rba_softmax_drop_last_f32(_p(logits), _p(out), logits.numel() // K1, K1, _stream()), "rba_softmax_drop_last_f32")
    return out


@_hip_op
def ood_components(score, threshold, min_dummy=None):
    """Open-set panoptic epilogue of a score map [H,W] (maskformer_model.py:454-474): binary = score > threshold, 3x3 opening
    then closing, 4-connected components.  Returns (labels int32 [H,W] with 0 = background and components numbered 1..n in
    raster order of their first pixel, n) like cv2.connectedComponents(connectivity=4) minus the background count."""
    lib = _lib.load()
    _chk(score, "score", dim=2)
    H, W = score.shape
    a = torch.empty((H, W), dtype=torch.uint8, device=score.device)
    b = torch.empty_like(a)
    st = _stream()
    _lib.check(lib.rba_threshold_u8(_p(score), _p(a), H * W, float(threshold), st), "rba_threshold_u8")
    for dilate, (src, dst) in ((0, (a, b)), (1, (b, a)), (1, (a, b)), (0, (b, a))):          # open, then close
        _lib.check(lib.rba_morph3x3_u8(_p(src), _p(dst), H, W, dilate, st), "rba_morph3x3_u8")
    roots = torch.empty((H, W), dtype=torch.int32, device=score.device)
    _lib.check(lib.rba_ccl4_roots_i32(_p(a), _p(roots), H, W, st), "rba_ccl4_roots_i32")
    flat = roots.view(-1)
    is_root = flat == torch.arange(H * W, dtype=torch.int32, device=score.device)
    rank = torch.cumsum(is_root.to(torch.int32), 0, dtype=torch.int32)                    # raster-order number of every root
    labels = torch.where(flat >= 0, rank[flat.clamp_min(0).long()], torch.zeros_like(flat))
    n = int(rank[-1].item()) if H * W else 0
    return labels.view(H, W), n
