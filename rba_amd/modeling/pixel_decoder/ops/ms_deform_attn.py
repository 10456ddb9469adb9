"""Multi-scale deformable attention: the reference's ``MSDeformAttn`` module and ``MSDeformAttnFunction`` surface
(pixel_decoder/ops/modules/ms_deform_attn.py:34-125, ops/functions/ms_deform_attn_func.py:32-49) on the HIP
kernel K2.  Differences by design: inference only (no backward), and -- unlike the reference's bare ``except``
(ms_deform_attn.py:116-121) that silently falls back to grid_sample -- a failure of the native op raises."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import ops


class MSDeformAttnFunction:
    """``MSDeformAttnFunction.apply(value, shapes, level_start_index, sampling_locations, attention_weights,
    im2col_step)`` -- same positional signature as the reference autograd Function (forward only)."""

    @staticmethod
    def apply(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
              im2col_step=128):
        return ops.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                          attention_weights, im2col_step)


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError(f"d_model must be divisible by n_heads, but got {d_model} and {n_heads}")
        self.im2col_step = 128
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)

    def _sampling_linear(self):
        """[sampling_offsets ; attention_weights] stacked along the output dimension, rebuilt when either weight changes."""
        so, aw = self.sampling_offsets, self.attention_weights
        key = (so.weight.data_ptr(), so.weight._version, aw.weight.data_ptr(), aw.weight._version, so.bias._version, aw.bias._version)
        cache = getattr(self, "_rba_sampling", None)
        if cache is None or cache[0] != key:
            from types import SimpleNamespace
            lin = SimpleNamespace(weight=torch.cat([so.weight.detach(), aw.weight.detach()], 0).contiguous(),
                                  bias=torch.cat([so.bias.detach(), aw.bias.detach()], 0).contiguous())
            cache = (key, lin)
            self._rba_sampling = cache
        return cache[1]

    def _sampling_parts(self):
        """the stacked sampling Linear cut into row chunks of <= 256 outputs for the row-complete token kernel: [(linear view, first column)]"""
        lin = self._sampling_linear()
        parts = getattr(lin, "parts", None)
        if parts is None:
            from types import SimpleNamespace
            n = lin.weight.shape[0]
            parts = lin.parts = [(SimpleNamespace(weight=lin.weight[c:c + 256], bias=lin.bias[c:c + 256]), c) for c in range(0, n, 256)]
        return parts

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None, query_pos=None, post=None):
        """query [N,Lq,C]; reference_points [N,Lq,L,2] in [0,1]; input_flatten [N,S,C] -> [N,Lq,C].
        query_pos: the attention runs on ``query + query_pos`` (the encoder's `self.with_pos_embed(src, pos)`, msdeformattn.py:133) -- handed
        over separately so that the add happens inside the Linear that consumes it.  post = (residual, LayerNorm): return
        ``norm(residual + output)`` (msdeformattn.py:134-135), in the output projection's epilogue where the row-complete kernel applies."""
        N, Lq, _ = query.shape
        S = input_flatten.shape[1]
        M, L, P = self.n_heads, self.n_levels, self.n_points
        if reference_points.shape[-1] != 2:
            raise ValueError(f"Last dim of reference_points must be 2 on this path, got {reference_points.shape[-1]}")
        C = self.d_model
        fused = ops.msda_fused_ok(C // M, L, P, S, M) and input_padding_mask is None
        parts = self._sampling_parts()
        tok = (fused and query is input_flatten and len(parts) <= 2 and query.is_contiguous() and ops.token_linear_pays(N * S, C, C)
               and all(ops.token_linear_pays(N * Lq, pl.weight.shape[0], C) for pl, _ in parts)
               and (query_pos is None or (query_pos.is_contiguous() and tuple(query_pos.shape) == tuple(query.shape))))
        if tok:
            # value = value_proj(src) and the sampling Linears of src + pos: ONE launch (was: an add and two GEMMs)
            raw = torch.empty((N, Lq, M * L * P * 3), dtype=torch.float32, device=query.device)
            outs = ops.token_linear_multi(query, [(self.value_proj, None, None, 0, False)] + [(pl, query_pos, raw, c0, False) for pl, c0 in parts])
            out = ops.msda_fused(outs[0].view(N, S, M, C // M), input_spatial_shapes, input_level_start_index, raw, reference_points.contiguous(),
                                 M, L, P)
            if post is not None and ops.token_linear_pays(N * Lq, C, C):
                return ops.token_linear(out, self.output_proj, residual=post[0].contiguous(), norm=post[1])
            return self._finish(ops.linear(out, self.output_proj), post)
        if query_pos is not None:
            query = query + query_pos
        return self._finish(self._forward_general(query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                                                  input_padding_mask), post)

    @staticmethod
    def _finish(y, post):
        if post is None:
            return y
        res, norm = post
        return ops.add_layer_norm(res, norm.weight, norm.bias, norm.eps, y.contiguous())[1]

    def _forward_general(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                         input_padding_mask=None):
        N, Lq, _ = query.shape
        S = input_flatten.shape[1]
        M, L, P = self.n_heads, self.n_levels, self.n_points
        value = ops.linear(input_flatten, self.value_proj)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], 0.0)
        value = value.view(N, S, M, self.d_model // M)
        # sampling_offsets and attention_weights as ONE Linear (rows [offsets | logits]), then one kernel for
        # loc = reference + offset / (W_l, H_l) and the softmax over the L*P logits (reference :95-115)
        raw = ops.linear(query, self._sampling_linear())
        if ops.msda_fused_ok(self.d_model // M, L, P, S, M) and input_padding_mask is None:
            # locations + softmax inside the gather kernel: one launch, no [N,Lq,M,L,P,3] round trip
            out = ops.msda_fused(value.contiguous(), input_spatial_shapes, input_level_start_index, raw.contiguous(),
                                 reference_points.contiguous(), M, L, P)
            return ops.linear(out, self.output_proj)
        loc, weights = ops.msda_prepare(raw, reference_points.contiguous(), input_spatial_shapes, M, L, P)
        out = MSDeformAttnFunction.apply(value.contiguous(), input_spatial_shapes, input_level_start_index, loc, weights,
                                         self.im2col_step)
        return ops.linear(out, self.output_proj)
