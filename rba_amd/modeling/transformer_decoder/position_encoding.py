"""Sine position embedding (reference: transformer_decoder/position_encoding.py:12-52, normalize=True, no mask).
It depends only on (h, w), so it is computed once per shape on the device and cached."""
import math

import torch
import torch.nn as nn

from ...lru import ShapeCache


class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=True, scale=None):
        super().__init__()
        if not normalize:
            raise NotImplementedError("only normalize=True is used on the RbA path")
        self.num_pos_feats, self.temperature = num_pos_feats, temperature
        self.scale = 2 * math.pi if scale is None else scale
        self._cache = ShapeCache(8)         # bounded: one entry per (h, w, device)

    def forward(self, x, mask=None):
        """x [B,C,h,w] (only its shape/device are used) -> [1, 2*num_pos_feats, h, w]."""
        if mask is not None:
            raise NotImplementedError("padding masks are all-False on this path (msdeformattn.py:71)")
        h, w = int(x.shape[-2]), int(x.shape[-1])
        dev = x.device

        def build():
            ys = torch.arange(1, h + 1, dtype=torch.float32, device=dev) / (h + 1e-6) * self.scale
            xs = torch.arange(1, w + 1, dtype=torch.float32, device=dev) / (w + 1e-6) * self.scale
            i = torch.arange(self.num_pos_feats, dtype=torch.float32, device=dev)
            dim_t = self.temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / self.num_pos_feats)
            px, py = xs[:, None] / dim_t, ys[:, None] / dim_t                      # [w,F], [h,F]
            px = torch.stack((px[:, 0::2].sin(), px[:, 1::2].cos()), dim=2).flatten(1)
            py = torch.stack((py[:, 0::2].sin(), py[:, 1::2].cos()), dim=2).flatten(1)
            pos = torch.cat((py[:, None, :].expand(h, w, -1), px[None, :, :].expand(h, w, -1)), dim=2)
            return pos.permute(2, 0, 1).contiguous()[None]

        return self._cache.get((h, w, dev), build)
