"""MaskFormerHead: pixel decoder -> transformer predictor glue (reference: meta_arch/mask_former_head.py:26-146)."""
import torch.nn as nn

from ...registry import SEM_SEG_HEADS_REGISTRY, TRANSFORMER_DECODER_REGISTRY
from ..pixel_decoder import msdeformattn as _pd  # noqa: F401  (registers MSDeformAttnPixelDecoder)
from ..transformer_decoder import mask2former_transformer_decoder as _td  # noqa: F401


@SEM_SEG_HEADS_REGISTRY.register()
class MaskFormerHead(nn.Module):
    def __init__(self, arch, pixel_decoder_name="MSDeformAttnPixelDecoder",
                 transformer_decoder_name="MultiScaleMaskedTransformerDecoder"):
        super().__init__()
        self.num_classes = arch["num_classes"]
        self.pixel_decoder = SEM_SEG_HEADS_REGISTRY.get(pixel_decoder_name)(arch)
        self.predictor = TRANSFORMER_DECODER_REGISTRY.get(transformer_decoder_name)(arch)
        self.transformer_in_feature = "multi_scale_pixel_decoder"

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # v1 -> v2 key rename (reference :31-53): pixel-decoder weights used to sit directly under the head
        for k in list(state_dict.keys()):
            if k.startswith(prefix) and not k.startswith(prefix + "predictor") and not k.startswith(prefix + "pixel_decoder"):
                state_dict[k.replace(prefix, prefix + "pixel_decoder.", 1)] = state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, features, mask=None):
        return self.layers(features, mask)

    def layers(self, features, mask=None):
        mask_features, _, multi_scale_features = self.pixel_decoder.forward_features(features)
        return self.predictor(multi_scale_features, mask_features, mask)
