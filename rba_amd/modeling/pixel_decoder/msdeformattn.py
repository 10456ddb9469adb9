"""MSDeformAttn pixel decoder, inference only (reference: pixel_decoder/msdeformattn.py:32-367).
Same parameter names; K2 (deformable attention) and the bilinear FPN top-down sum are HIP kernels, 1x1/3x3
convolutions and the encoder FFN are MFMA GEMMs/convs through rocBLAS/MIOpen."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...arch import FEATURE_NAMES, FEATURE_STRIDES, num_fpn_levels
from ...registry import SEM_SEG_HEADS_REGISTRY
from ..transformer_decoder.position_encoding import PositionEmbeddingSine
from .ops.ms_deform_attn import MSDeformAttn


class ConvNorm(nn.Module):
    """Detectron2 ``Conv2d`` wrapper as a parameter holder: ``weight`` (+ ``bias``) and an optional ``norm`` child
    (conv -> GroupNorm(32) -> optional ReLU), keys ``<name>.weight``, ``<name>.norm.{weight,bias}``."""

    def __init__(self, cin, cout, k, bias, norm, relu=False):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        self.norm = nn.GroupNorm(32, cout) if norm else None
        self.padding, self.relu = k // 2, relu

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, padding=self.padding)
        if self.norm is not None:
            return ops.group_norm(x.contiguous(), 32, self.norm.weight, self.norm.bias, self.norm.eps, relu=self.relu)
        return F.relu(x) if self.relu else x


class MSDeformAttnTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, d_ffn, n_levels, n_heads, n_points):
        super().__init__()
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.norm1 = nn.LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.norm2 = nn.LayerNorm(d_model)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index):
        """msdeformattn.py:131-140 (dropout = identity)."""
        src2 = self.self_attn(src + pos, reference_points, src, spatial_shapes, level_start_index)
        _, src = ops.add_layer_norm(src, self.norm1.weight, self.norm1.bias, self.norm1.eps, src2.contiguous())
        src2 = F.linear(F.relu(self.linear1(src)), self.linear2.weight)
        return ops.add_layer_norm(src, self.norm2.weight, self.norm2.bias, self.norm2.eps, src2, self.linear2.bias)[1]


class MSDeformAttnTransformerEncoder(nn.Module):
    def __init__(self, layer_args, num_layers):
        super().__init__()
        self.layers = nn.ModuleList(MSDeformAttnTransformerEncoderLayer(*layer_args) for _ in range(num_layers))

    @staticmethod
    def get_reference_points(spatial_shapes, device):
        """Pixel centres (i+0.5)/H per level, broadcast over levels; valid_ratios == 1 (msdeformattn.py:149-162)."""
        refs = []
        for H_, W_ in spatial_shapes:
            ry = (torch.arange(H_, dtype=torch.float32, device=device) + 0.5) / H_
            rx = (torch.arange(W_, dtype=torch.float32, device=device) + 0.5) / W_
            refs.append(torch.stack((rx[None, :].expand(H_, W_), ry[:, None].expand(H_, W_)), -1).reshape(-1, 2))
        ref = torch.cat(refs, 0)[None]
        return ref[:, :, None, :].repeat(1, 1, len(spatial_shapes), 1).contiguous()


class MSDeformAttnTransformerEncoderOnly(nn.Module):
    def __init__(self, d_model, nhead, num_encoder_layers, dim_feedforward, num_feature_levels, enc_n_points):
        super().__init__()
        self.encoder = MSDeformAttnTransformerEncoder(
            (d_model, dim_feedforward, num_feature_levels, nhead, enc_n_points), num_encoder_layers)
        self.level_embed = nn.Parameter(torch.zeros(num_feature_levels, d_model))
        self._ref_cache = {}

    def forward(self, srcs, pos_embeds):
        """msdeformattn.py:70-98; returns (memory [B,S,C], shapes list, level_start list)."""
        dev = srcs[0].device
        shapes = [tuple(int(v) for v in s.shape[-2:]) for s in srcs]
        src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
        pos = torch.cat([p.flatten(2).transpose(1, 2) + self.level_embed[l].view(1, 1, -1)
                         for l, p in enumerate(pos_embeds)], 1)
        key = (tuple(shapes), dev)
        if key not in self._ref_cache:
            sh = torch.as_tensor(shapes, dtype=torch.long, device=dev)
            lsi = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))
            self._ref_cache[key] = (sh, lsi, MSDeformAttnTransformerEncoder.get_reference_points(shapes, dev))
        sh, lsi, ref = self._ref_cache[key]
        B = src.shape[0]
        if B > 1:
            ref = ref.expand(B, -1, -1, -1).contiguous()
        out = src
        for layer in self.encoder.layers:
            out = layer(out, pos, ref, sh, lsi)
        return out, shapes


@SEM_SEG_HEADS_REGISTRY.register()
class MSDeformAttnPixelDecoder(nn.Module):
    def __init__(self, arch):
        super().__init__()
        a = arch
        d = a["conv_dim"]
        E = a["embed_dim"]
        chans = {f: E * 2 ** k for k, f in enumerate(FEATURE_NAMES)}
        self.in_features = list(FEATURE_NAMES)
        self.transformer_in_features = list(a["enc_in"])
        self.transformer_num_feature_levels = len(a["enc_in"])
        self.maskformer_num_feature_levels = len(a["enc_in"])
        self.input_proj = nn.ModuleList(
            nn.Sequential(nn.Conv2d(chans[f], d, kernel_size=1), nn.GroupNorm(32, d)) for f in a["enc_in"][::-1])
        self.transformer = MSDeformAttnTransformerEncoderOnly(
            d, a["nheads"], a["enc_layers"], a["enc_dim_feedforward"], len(a["enc_in"]), a["enc_points"])
        self.pe_layer = PositionEmbeddingSine(d // 2, normalize=True)
        self.mask_features = nn.Conv2d(d, a["mask_dim"], kernel_size=1)
        self.num_fpn_levels = num_fpn_levels(a)
        for j in range(1, self.num_fpn_levels + 1):
            self.add_module(f"adapter_{j}", ConvNorm(chans[FEATURE_NAMES[j - 1]], d, 1, bias=False, norm=True))
            self.add_module(f"layer_{j}", ConvNorm(d, d, 3, bias=False, norm=True, relu=True))

    def forward_features(self, features):
        """-> (mask_features [B,md,H/4,W/4], out[0], multi_scale_features) (msdeformattn.py:323-367)."""
        srcs, pos = [], []
        for idx, f in enumerate(self.transformer_in_features[::-1]):
            x = features[f].float()
            conv, gn = self.input_proj[idx][0], self.input_proj[idx][1]
            srcs.append(ops.group_norm(F.conv2d(x, conv.weight, conv.bias).contiguous(), 32, gn.weight, gn.bias, gn.eps))
            pos.append(self.pe_layer(x))
        y, shapes = self.transformer(srcs, pos)
        B = y.shape[0]
        outs = [z.transpose(1, 2).reshape(B, -1, shapes[i][0], shapes[i][1]).contiguous()
                for i, z in enumerate(torch.split(y, [h * w for h, w in shapes], dim=1))]
        for idx, f in enumerate(self.in_features[:self.num_fpn_levels][::-1]):
            j = self.num_fpn_levels - idx
            cur = getattr(self, f"adapter_{j}")(features[f].float())
            yy = ops.resample_bilinear(outs[-1], cur.shape[-2:], add=cur.contiguous())   # :357-358 fused sum
            outs.append(getattr(self, f"layer_{j}")(yy))
        mf = F.conv2d(outs[-1], self.mask_features.weight, self.mask_features.bias)
        return mf, outs[0], outs[:self.maskformer_num_feature_levels]
