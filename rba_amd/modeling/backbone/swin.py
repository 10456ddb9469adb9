"""Swin Transformer backbone, inference only, for MI355X.

Same parameter tree (hence state-dict keys) as the reference's ``D2SwinTransformer``
(mask2former/modeling/backbone/swin.py:498-770) but a different dataflow: the token map stays [B, H*W, C]
for the whole stage; the zero-pad to a multiple of the window, the cyclic shift, window partition / reverse,
relative-position bias, shift mask, softmax and P@V of a block are ONE HIP kernel (``ops.swin_window_attn``,
K5) reading the un-padded qkv tensor; dense projections are the repository's f16x3 MFMA GEMMs (K6).  Round 5: where K7 has a kernel
(``ops.swin_attn_block``: C = 128; ``ops.swin_attn_qkv``: C = 128 / 192 / 256) the attention half of a block -- norm1, qkv, attention[, proj, residual] --
is one launch per block, and the C = 128 MLP computes norm2 itself (``ops.mlp_fused_ln``).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...registry import BACKBONE_REGISTRY


def relative_position_index(ws: int) -> torch.Tensor:
    """[ws*ws, ws*ws] int64 index into the (2ws-1)^2 bias table (swin.py:108-121)."""
    ar = torch.arange(ws)
    gy, gx = torch.meshgrid(ar, ar, indexing="ij")
    gy, gx = gy.reshape(-1), gx.reshape(-1)
    dy = gy[:, None] - gy[None, :] + (ws - 1)
    dx = gx[:, None] - gx[None, :] + (ws - 1)
    return dy * (2 * ws - 1) + dx


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
        self.register_buffer("relative_position_index", relative_position_index(window_size))
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        self._bias_cache = None
        self._block_image = None

    def block_image(self):
        """K7's packed qkv + proj weights (ops.swin_attn_block_weights), once per weight load."""
        wq, wp = self.qkv.weight, self.proj.weight
        key = (wq.data_ptr(), wq._version, wp.data_ptr(), wp._version, wq.device)
        if self._block_image is None or self._block_image[0] != key:
            self._block_image = (key, ops.swin_attn_block_weights(wq.detach().contiguous(), wp.detach().contiguous()))
        return self._block_image[1]

    def gathered_bias(self):
        """[nH, N, N] relative-position bias (swin.py:148-155), gathered once per weight load."""
        t = self.relative_position_bias_table
        key = (t.data_ptr(), t._version, t.device)
        if self._bias_cache is None or self._bias_cache[0] != key:
            N = self.window_size ** 2
            b = t[self.relative_position_index.view(-1)].view(N, N, -1).permute(2, 0, 1).contiguous()
            frag = ops.swin_bias_fragments(b, self.window_size) if b.is_cuda else None
            self._bias_cache = (key, b, frag)
        return self._bias_cache[1], self._bias_cache[2]


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio):
        super().__init__()
        self.window_size, self.shift_size, self.num_heads = window_size, shift_size, num_heads
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size, num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x, H, W, pending=None):
        """x [B, H*W, C] residual stream; ``pending`` = (t, bias) not yet added to it (the previous block's fc2 output).
        Returns (x, pending) with this block's fc2 output pending (reference dataflow: swin.py:235-295).  The residual
        adds and projection biases are folded into the fused add+LayerNorm kernel."""
        a = self.attn
        t, tb = pending if pending is not None else (None, None)
        M, C = x.numel() // x.shape[-1], x.shape[-1]
        if x.is_cuda and x.dim() == 3 and ops.swin_attn_block_ok(C, self.num_heads, self.window_size):
            # K7: norm1 -> qkv -> (shifted-)window attention -> proj -> + shortcut -> norm2 in ONE kernel, in place over x
            if pending is not None:
                x = x + t + tb
            x = x.contiguous()
            bias_frag = a.gathered_bias()[1]
            hidden = self.mlp.fc1.out_features
            n2 = (self.norm2.weight, self.norm2.bias, self.norm2.eps)
            fused_mlp = ops.mlp_fused_ok(M, C, hidden)               # the one-kernel MLP computes norm2 itself from the residual stream: K7 then writes x only
            x, y = ops.swin_attn_block(x, (self.norm1.weight, self.norm1.bias, self.norm1.eps), a.block_image(), a.qkv.bias, bias_frag, a.proj.bias, H, W,
                                       self.window_size, self.shift_size, norm2=None if fused_mlp else n2)
            if fused_mlp:
                return ops.mlp_fused_ln(x, n2, self.mlp.fc1, self.mlp.fc2), None
            if ops.linear_residual_fused(M, C, hidden):
                y = ops.linear(y, self.mlp.fc1, gelu=True, split_out=ops.linear_takes_split(M, C, hidden))
                return ops.linear(y, self.mlp.fc2, residual=x), None
            y = ops.linear(y, self.mlp.fc1, gelu=True)
            return x, (ops.linear(y, self.mlp.fc2, use_bias=False), self.mlp.fc2.bias)
        if (x.is_cuda and x.dim() == 3 and pending is None and ops.swin_attn_qkv_ok(C, self.num_heads, self.window_size)
                and ops.linear_residual_fused(M, C, C) and ops.linear_takes_split(M, C, C)):
            # K7 without proj (C = 256, Swin-B stage 2): norm1 -> qkv -> window attention in ONE kernel, its output the proj GEMM's split operand; the residual
            # add rides in that GEMM's epilogue as before
            x = x.contiguous()
            y = ops.swin_attn_qkv(x, (self.norm1.weight, self.norm1.bias, self.norm1.eps), a.block_image(), a.qkv.bias, a.gathered_bias()[1], H, W,
                                  self.window_size, self.shift_size)
            x = ops.linear(y, a.proj, residual=x)
            hidden = self.mlp.fc1.out_features
            y = ops.add_layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, frag=ops.linear_takes_split(M, hidden, C))[1]
            y = ops.linear(y, self.mlp.fc1, gelu=True, split_out=ops.linear_takes_split(M, C, hidden))
            return ops.linear(y, self.mlp.fc2, residual=x), None
        # where the consumer is the pipelined f16x3 GEMM, the LayerNorm hands its output over already split and in MFMA fragment
        # order (ops.SplitActivations): one split per element instead of one per column tile, contiguous operand loads
        x, y = ops.add_layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, t, tb, inplace_sum=True,
                                  frag=ops.linear_takes_split(M, 3 * C, C))
        qkv = ops.linear(y, a.qkv)
        bias, bias_frag = a.gathered_bias()
        fused = ops.linear_residual_fused(M, C, C)
        y = ops.swin_window_attn(qkv, a.qkv.bias, bias, H, W, self.num_heads, self.window_size, self.shift_size, bias_frag=bias_frag,
                                 split_out=fused and bias_frag is not None and ops.linear_takes_split(M, C, C) and
                                 ops.swin_window_attn_split_ok(C // self.num_heads, self.window_size))
        if fused:
            # the residual adds ride in the GEMM epilogues (x is updated in place), the LayerNorms read one tensor and write one
            x = ops.linear(y, a.proj, residual=x)
            hidden = self.mlp.fc1.out_features
            if ops.mlp_fused_ok(M, C, hidden):
                # C = 128 (Swin-B stage 1): fc1 + GELU + fc2 + residual in ONE kernel, the 4C-wide hidden tensor never leaves the registers
                y = ops.add_layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps)[1]
                return ops.mlp_fused(y, self.mlp.fc1, self.mlp.fc2, x), None
            # fc2 reads fc1's output as its split operand wherever it runs the pipelined kernel; fc1's own input comes split from
            # the LayerNorm only where K > 256 (below, the scattered 16-byte stores cost the LayerNorm more than the GEMM gains)
            y = ops.add_layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, frag=ops.linear_takes_split(M, hidden, C))[1]
            y = ops.linear(y, self.mlp.fc1, gelu=True, split_out=ops.linear_takes_split(M, C, hidden))
            return ops.linear(y, self.mlp.fc2, residual=x), None
        t = ops.linear(y, a.proj, use_bias=False)                        # proj bias rides in the fused add+LN
        x, y = ops.add_layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps, t, a.proj.bias, inplace_sum=True)
        y = ops.linear(y, self.mlp.fc1, gelu=True)                       # exact GELU in the GEMM epilogue
        return x, (ops.linear(y, self.mlp.fc2, use_bias=False), self.mlp.fc2.bias)


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x, H, W):
        """2x2 gather in the order (ee, oe, eo, oo) -> LN -> Linear (swin.py:311-337)."""
        x = ops.merge_layer_norm(x.contiguous(), H, W, self.norm.weight, self.norm.bias, self.norm.eps)
        return ops.linear(x, self.reduction)


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio, downsample):
        super().__init__()
        self.blocks = nn.ModuleList(
            SwinTransformerBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2, mlp_ratio)
            for i in range(depth))
        self.downsample = PatchMerging(dim) if downsample else None


class PatchEmbed(nn.Module):
    def __init__(self, patch_size, in_chans, embed_dim):
        super().__init__()
        self.patch_size = patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, x):
        """[B,3,H,W] -> tokens [B, Wh*Ww, C], Wh, Ww (swin.py:479-495).  Any memory layout (channels_last, sliced batches: the gather below copies).  Arithmetic of this
        un-fused front end = the token Linears' (ops.SPLIT_MODE): f16x3 by default, i.e. fp32-GEMM accuracy for |x| < 65504 and NaN -- never a wrong number -- beyond;
        normalised pixels are |x| < 3, and `ops.split_mode("bf16x6")` gives the full fp32 range (test_fused_front_end_matches_library_path holds both paths equal)."""
        ps = self.patch_size
        _, _, H, W = x.shape
        if W % ps or H % ps:
            x = F.pad(x, (0, (ps - W % ps) % ps, 0, (ps - H % ps) % ps))
        # the stride-ps, ps x ps convolution of non-overlapping patches IS a Linear over the patch's 3 ps^2 values (k = (c, dy, dx), the weight's own order):
        # one gather + the library's GEMM (ops.linear pads K = 48 to 64) -- no MIOpen call on this path either (round 5)
        B, Cin, H, W = x.shape
        Wh, Ww = H // ps, W // ps
        cols = x.reshape(B, Cin, Wh, ps, Ww, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, Wh * Ww, Cin * ps * ps).contiguous()     # reshape: channels_last / sliced inputs too, as conv2d took them
        w = self.proj.weight
        key = (w.data_ptr(), w._version, w.device)
        c = getattr(self, "_rba_lin", None)
        if c is None or c[0] != key:
            from types import SimpleNamespace
            c = self._rba_lin = (key, SimpleNamespace(weight=w.detach().reshape(w.shape[0], -1), bias=self.proj.bias))
        x = ops.linear(cols, c[1])
        return ops.add_layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)[1], Wh, Ww

    def fused_ok(self):
        w = self.proj.weight
        return self.patch_size == 4 and w.shape[1] == 3 and w.is_cuda and w.dtype == torch.float32 and ops.SPLIT_MODE == "f16x3"

    def forward_images(self, images, mean, std, Hp, Wp):
        """Raw images [3,h_i,w_i] (uint8 / fp32) -> tokens [B, (Hp/4)*(Wp/4), C]: normalisation, ImageList zero padding and the im2col of the
        4x4 convolution in ONE kernel (ops.patch_im2col), the projection as an f16x3 GEMM over K = 48 (+16 zero columns), then the
        LayerNorm -- instead of five elementwise passes, a library convolution and a 67 MB NCHW -> token-major copy."""
        w = self.proj.weight
        key = (w.data_ptr(), w._version, w.device)
        cache = getattr(self, "_rba_planes", None)
        if cache is None or cache[0] != key:
            w64 = torch.zeros((w.shape[0], 64), dtype=torch.float32, device=w.device)
            w64[:, :48] = w.detach().reshape(w.shape[0], 48)
            cache = (key, ops.split_weight(w64, mode="f16x3"))
            self._rba_planes = cache
        cols = torch.stack([ops.patch_im2col(im, mean, std, Hp, Wp) for im in images]) if len(images) > 1 \
            else ops.patch_im2col(images[0], mean, std, Hp, Wp)[None]
        x = ops.split_linear(cols, cache[1], self.proj.bias, out_features=w.shape[0])
        return ops.add_layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)[1], Hp // 4, Wp // 4


@BACKBONE_REGISTRY.register()
class D2SwinTransformer(nn.Module):
    """forward(x [B,3,H,W]) -> {"res2".."res5": [B,C_i,H/4..H/32, W/4..W/32]} (swin.py:743-758)."""

    def __init__(self, arch):
        super().__init__()
        E, depths = arch["embed_dim"], arch["depths"]
        self.patch_embed = PatchEmbed(arch["patch_size"], 3, E)
        self.layers = nn.ModuleList(
            BasicLayer(E * 2 ** i, depths[i], arch["num_heads"][i], arch["window_size"], arch["mlp_ratio"],
                       downsample=i < len(depths) - 1) for i in range(len(depths)))
        self.num_features = [E * 2 ** i for i in range(len(depths))]
        for i, c in enumerate(self.num_features):
            self.add_module(f"norm{i}", nn.LayerNorm(c))

    @property
    def size_divisibility(self):
        return 32

    def forward_images(self, images, mean, std, Hp, Wp):
        """The same network from raw images (see PatchEmbed.forward_images); Hp, Wp = the padded size (multiples of 32)."""
        return self._stages(*self.patch_embed.forward_images(images, mean, std, Hp, Wp))

    def forward(self, x):
        if x.dim() != 4:
            raise ValueError(f"SwinTransformer takes an input of shape (N, C, H, W). Got {tuple(x.shape)} instead!")
        return self._stages(*self.patch_embed(x))

    def _stages(self, x, Wh, Ww):
        outs = {}
        for i, layer in enumerate(self.layers):
            pending = None
            for blk in layer.blocks:
                x, pending = blk(x, Wh, Ww, pending)
            norm = getattr(self, f"norm{i}")
            t, tb = pending if pending is not None else (None, None)
            x, y = ops.add_layer_norm(x, norm.weight, norm.bias, norm.eps, t, tb, inplace_sum=True)
            # [B,C,h,w] as a channels-last VIEW of the token tensor (no copy): same values and shape as the reference's
            # permute(0,3,1,2).contiguous() (swin.py:752-753); the pixel decoder consumes the token layout directly
            outs[f"res{i + 2}"] = y.view(-1, Wh, Ww, self.num_features[i]).permute(0, 3, 1, 2)
            if layer.downsample is not None:
                x = layer.downsample(x, Wh, Ww)
                Wh, Ww = (Wh + 1) // 2, (Ww + 1) // 2
        return outs
