"""Minimal reader for the reference's Detectron2/yacs YAML configs (no detectron2 / yacs dependency).

``load_cfg(path)`` handles both the fully-resolved dumps next to released weights (``ckpts/*/config.yaml``, the
load path of evaluate_ood.py:264-265) and the training chains under ``configs/**`` (``_BASE_`` inheritance,
python/eval tags such as Base-Cityscapes-SemanticSegmentation.yaml:37 are read as plain strings and ignored).
Keys the inference path consumes but older dumps lack get the defaults of mask2former/config.py:6-244
(e.g. ckpts/swin_l_1dl/config.yaml has no DENSE_HYBRID_LOSS).
"""
import os

import yaml


class CfgNode(dict):
    """dict with attribute access (read-only use)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return CfgNode({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


class _Loader(yaml.SafeLoader):
    pass


def _any_tag(loader, suffix, node):
    if isinstance(node, yaml.ScalarNode):
        return loader.construct_scalar(node)
    if isinstance(node, yaml.SequenceNode):
        return loader.construct_sequence(node, deep=True)
    return loader.construct_mapping(node, deep=True)


_Loader.add_multi_constructor("tag:yaml.org,2002:python/", _any_tag)
_Loader.add_multi_constructor("!", _any_tag)

# defaults of the keys read by rba_amd.arch.arch_from_cfg (detectron2 defaults + add_maskformer2_config)
_DEFAULTS = {
    "MODEL": {
        "META_ARCHITECTURE": "MaskFormer",
        "PIXEL_MEAN": [123.675, 116.28, 103.53],
        "PIXEL_STD": [58.395, 57.12, 57.375],
        "BACKBONE": {"NAME": "D2SwinTransformer"},
        "SWIN": {"PRETRAIN_IMG_SIZE": 224, "PATCH_SIZE": 4, "EMBED_DIM": 96, "DEPTHS": [2, 2, 6, 2],
                 "NUM_HEADS": [3, 6, 12, 24], "WINDOW_SIZE": 7, "MLP_RATIO": 4.0, "QKV_BIAS": True, "QK_SCALE": None,
                 "DROP_RATE": 0.0, "ATTN_DROP_RATE": 0.0, "DROP_PATH_RATE": 0.3, "APE": False, "PATCH_NORM": True,
                 "OUT_FEATURES": ["res2", "res3", "res4", "res5"], "USE_CHECKPOINT": False},
        # detectron2/config/defaults.py MODEL.RESNETS (build_resnet_backbone: BASELINE config C1)
        "RESNETS": {"DEPTH": 50, "OUT_FEATURES": ["res4"], "NUM_GROUPS": 1, "NORM": "FrozenBN", "WIDTH_PER_GROUP": 64,
                    "STRIDE_IN_1X1": True, "RES5_DILATION": 1, "RES2_OUT_CHANNELS": 256, "STEM_OUT_CHANNELS": 64,
                    "DEFORM_ON_PER_STAGE": [False, False, False, False]},
        "SEM_SEG_HEAD": {"NAME": "MaskFormerHead", "IN_FEATURES": ["res2", "res3", "res4", "res5"], "NUM_CLASSES": 19,
                         "IGNORE_VALUE": 255, "LOSS_WEIGHT": 1.0, "CONVS_DIM": 256, "MASK_DIM": 256, "NORM": "GN",
                         "PIXEL_DECODER_NAME": "MSDeformAttnPixelDecoder", "TRANSFORMER_ENC_LAYERS": 6,
                         "DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES": ["res3", "res4", "res5"], "COMMON_STRIDE": 4},
        "MASK_FORMER": {"TRANSFORMER_DECODER_NAME": "MultiScaleMaskedTransformerDecoder",
                        "TRANSFORMER_IN_FEATURE": "multi_scale_pixel_decoder", "HIDDEN_DIM": 256,
                        "NUM_OBJECT_QUERIES": 100, "NHEADS": 8, "DROPOUT": 0.0, "DIM_FEEDFORWARD": 2048,
                        "DEC_LAYERS": 10, "PRE_NORM": False, "ENFORCE_INPUT_PROJ": False, "SIZE_DIVISIBILITY": 32,
                        "DENSE_HYBRID_LOSS": False,
                        "OPEN_PANOPTIC": True,
                        "TEST": {"SEMANTIC_ON": True, "INSTANCE_ON": False, "PANOPTIC_ON": False,
                                 "OBJECT_MASK_THRESHOLD": 0.0, "OVERLAP_THRESHOLD": 0.0,
                                 "SEM_SEG_POSTPROCESSING_BEFORE_INFERENCE": False}},
    },
    "SOLVER": {"FORCE_REGION_PARTITION": False},
}


def _read(path, seen=()):
    path = os.path.abspath(path)
    if path in seen:
        raise ValueError(f"_BASE_ cycle at {path}")
    with open(path) as f:
        d = yaml.load(f, Loader=_Loader) or {}
    base = d.pop("_BASE_", None)
    if base is not None:
        if not os.path.isabs(base):
            base = os.path.join(os.path.dirname(path), base)
        d = _merge(_read(base, seen + (path,)), d)
    return d


def load_cfg(path, opts=None) -> CfgNode:
    """Read ``path`` (following ``_BASE_``), overlay it on the defaults, apply ``opts`` = [KEY.PATH, value, ...]."""
    import copy

    d = _merge(copy.deepcopy(_DEFAULTS), _read(path))
    if opts:
        if len(opts) % 2:
            raise ValueError("opts must be KEY VALUE pairs")
        for k, v in zip(opts[0::2], opts[1::2]):
            node = d
            parts = k.split(".")
            for p in parts[:-1]:
                node = node.setdefault(p, {})
            node[parts[-1]] = yaml.safe_load(v) if isinstance(v, str) else v
    return _wrap(d)
