"""Detectron2's ResNet backbone as Mask2Former uses it for BASELINE config C1 (``MODEL.BACKBONE.NAME: build_resnet_backbone``,
``MODEL.RESNETS`` of configs/cityscapes/semantic-segmentation/Base-Cityscapes-SemanticSegmentation.yaml:2-15: DEPTH 50,
STEM_OUT_CHANNELS 64, STRIDE_IN_1X1 False, NORM SyncBN, OUT_FEATURES res2..res5, RES5 dilation 1).

Detectron2 itself is not in the reference tree or the image, so this file restates ``detectron2/modeling/backbone/resnet.py``
(v0.6) from its published definition and state-dict key layout (``backbone.stem.conv1.{weight, norm.*}``,
``backbone.res{2..5}.{i}.{shortcut, conv1, conv2, conv3}.{weight, norm.*}``): BasicStem = 7x7/2 conv + BN + ReLU + 3x3/2 max-pool;
BottleneckBlock = 1x1 -> 3x3 (carries the stride when STRIDE_IN_1X1 is False) -> 1x1, each conv + BN, ReLU after the first two,
projection shortcut (1x1 conv + BN, stride) when the shape changes, ReLU after the sum.  **Parity unpinned** (DESIGN.md section 2).

This backbone is plumbing for C1, not a tuned path: convolutions are MIOpen library calls (``F.conv2d``) with the inference-mode
BatchNorm folded into the convolution weights once per weight load; everything downstream of it (pixel decoder, decoder, K1) is the
same HIP path as with Swin.  Outputs are channels-last so that the pixel decoder's token-layout path applies without a copy."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...registry import BACKBONE_REGISTRY

STAGE_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


class ConvBN(nn.Module):
    """Detectron2 ``Conv2d(..., bias=False, norm=BN)``: parameters ``weight`` and child ``norm`` (BatchNorm statistics)."""

    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")
        self.norm = nn.BatchNorm2d(cout)
        self.stride, self.padding = stride, padding
        self._folded = None

    def folded(self):
        n = self.norm
        key = (self.weight.data_ptr(), self.weight._version, n.weight._version, n.running_var._version, self.weight.device)
        if self._folded is None or self._folded[0] != key:
            scale = n.weight * torch.rsqrt(n.running_var + n.eps)
            w = (self.weight * scale.view(-1, 1, 1, 1)).contiguous(memory_format=torch.channels_last)
            self._folded = (key, w, (n.bias - n.running_mean * scale).contiguous())
        return self._folded[1], self._folded[2]

    def forward(self, x, relu=False):
        w, b = self.folded()
        y = F.conv2d(x, w, b, stride=self.stride, padding=self.padding)
        return torch.relu_(y) if relu else y


class BasicStem(nn.Module):
    def __init__(self, cout=64):
        super().__init__()
        self.conv1 = ConvBN(3, cout, 7, stride=2, padding=3)

    def forward(self, x):
        return F.max_pool2d(self.conv1(x, relu=True), kernel_size=3, stride=2, padding=1)


class BottleneckBlock(nn.Module):
    def __init__(self, cin, cout, bottleneck, stride, stride_in_1x1):
        super().__init__()
        if cin != cout:
            self.shortcut = ConvBN(cin, cout, 1, stride=stride)
        else:
            self.shortcut = None
        s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = ConvBN(cin, bottleneck, 1, stride=s1)
        self.conv2 = ConvBN(bottleneck, bottleneck, 3, stride=s3, padding=1)
        self.conv3 = ConvBN(bottleneck, cout, 1)

    def forward(self, x):
        out = self.conv3(self.conv2(self.conv1(x, relu=True), relu=True))
        out += self.shortcut(x) if self.shortcut is not None else x
        return torch.relu_(out)


class ResNet(nn.Module):
    """forward(x [B,3,H,W]) -> {"res2".."res5": [B, 256 .. 2048, H/4 .. H/32, W/4 .. W/32]} (channels-last memory format)."""

    def __init__(self, arch):
        super().__init__()
        r = arch["resnet"]
        self.stem = BasicStem(r["stem_out"])
        cin, cout, bott = r["stem_out"], r["res2_out"], r["width"]
        self.num_features = []
        for i, nblocks in enumerate(STAGE_BLOCKS[r["depth"]]):
            blocks = []
            for b in range(nblocks):
                blocks.append(BottleneckBlock(cin, cout, bott, (1 if i == 0 else 2) if b == 0 else 1, r["stride_in_1x1"]))
                cin = cout
            setattr(self, f"res{i + 2}", nn.Sequential(*blocks))
            self.num_features.append(cout)
            cout, bott = cout * 2, bott * 2

    @property
    def size_divisibility(self):
        return 32

    def forward(self, x):
        if x.dim() != 4:
            raise ValueError(f"ResNet takes an input of shape (N, C, H, W). Got {tuple(x.shape)} instead!")
        x = self.stem(x.contiguous(memory_format=torch.channels_last))
        outs = {}
        for i in range(4):
            x = getattr(self, f"res{i + 2}")(x)
            outs[f"res{i + 2}"] = x
        return outs


@BACKBONE_REGISTRY.register()
def build_resnet_backbone(arch):
    return ResNet(arch)
