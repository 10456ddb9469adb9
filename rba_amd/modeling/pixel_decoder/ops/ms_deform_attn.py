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

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        """query [N,Lq,C]; reference_points [N,Lq,L,2] in [0,1]; input_flatten [N,S,C] -> [N,Lq,C]."""
        N, Lq, _ = query.shape
        S = input_flatten.shape[1]
        M, L, P = self.n_heads, self.n_levels, self.n_points
        value = ops.linear(input_flatten, self.value_proj)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], 0.0)
        value = value.view(N, S, M, self.d_model // M)
        if reference_points.shape[-1] != 2:
            raise ValueError(f"Last dim of reference_points must be 2 on this path, got {reference_points.shape[-1]}")
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
