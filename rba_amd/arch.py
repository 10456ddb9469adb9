"""Architecture description of the hot path and its checkpoint contract.

``arch`` is a plain dict holding exactly the hyper-parameters the inference path consumes
(reference: ``mask2former/modeling/backbone/swin.py:690-728``,
``pixel_decoder/msdeformattn.py:303-321``, ``transformer_decoder/mask2former_transformer_decoder.py:369-396``,
``mask2former/maskformer_model.py:197-221``).  ``state_dict_shapes`` enumerates every key of the reference
``MaskFormer.state_dict()`` on this path with its shape -- the ``model_final.pth`` contract.
"""
import torch

FEATURE_NAMES = ("res2", "res3", "res4", "res5")
FEATURE_STRIDES = {"res2": 4, "res3": 8, "res4": 16, "res5": 32}
# MODEL.RESNETS of Base-Cityscapes-SemanticSegmentation.yaml:8-15 (+ Detectron2 defaults RES2_OUT_CHANNELS 256, WIDTH_PER_GROUP 64)
RESNET50 = dict(depth=50, stem_out=64, res2_out=256, width=64, stride_in_1x1=False)
RESNET_STAGE_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


def feature_channels(a: dict) -> dict:
    """channels of res2..res5 for the arch's backbone"""
    if a.get("resnet"):
        return {f: a["resnet"]["res2_out"] * 2 ** k for k, f in enumerate(FEATURE_NAMES)}
    return {f: a["embed_dim"] * 2 ** k for k, f in enumerate(FEATURE_NAMES)}


def backbone_name(a: dict) -> str:
    return "build_resnet_backbone" if a.get("resnet") else "D2SwinTransformer"

# Named architectures of BASELINE.json's configs (values from ckpts/*/config.yaml and the
# configs/cityscapes/semantic-segmentation/swin/ chains of the reference).
ARCHS = {
    "swin_b_1dl": dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=12,
                       conv_dim=256, mask_dim=256, nheads=8, num_queries=100, num_classes=19,
                       dim_feedforward=2048, enc_layers=6, dec_layers=1, enc_in=["res5"]),
    "swin_l_1dl": dict(embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], window_size=12,
                       conv_dim=256, mask_dim=256, nheads=8, num_queries=100, num_classes=19,
                       dim_feedforward=2048, enc_layers=6, dec_layers=1, enc_in=["res5"]),
    "swin_b_9dl": dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=12,
                       conv_dim=256, mask_dim=256, nheads=8, num_queries=100, num_classes=19,
                       dim_feedforward=2048, enc_layers=6, dec_layers=9, enc_in=["res3", "res4", "res5"]),
    "tiny1": dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=6,
                  conv_dim=64, mask_dim=64, nheads=2, num_queries=16, num_classes=19,
                  dim_feedforward=128, enc_layers=2, dec_layers=1, enc_in=["res5"]),
    "tiny3": dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=6,
                  conv_dim=64, mask_dim=64, nheads=2, num_queries=16, num_classes=19,
                  dim_feedforward=128, enc_layers=2, dec_layers=4, enc_in=["res3", "res4", "res5"]),
    # BASELINE config C1: ResNet-50 Mask2Former (maskformer2_R50_bs16_90k.yaml) with DEC_LAYERS 2 and the single-level encoder
    "r50_1dl": dict(resnet=RESNET50, conv_dim=256, mask_dim=256, nheads=8, num_queries=100, num_classes=19,
                    dim_feedforward=2048, enc_layers=6, dec_layers=1, enc_in=["res5"]),
    # ... and as the file defines it: 9 decoder layers, 3-level encoder
    "r50_9dl": dict(resnet=RESNET50, conv_dim=256, mask_dim=256, nheads=8, num_queries=100, num_classes=19,
                    dim_feedforward=2048, enc_layers=6, dec_layers=9, enc_in=["res3", "res4", "res5"]),
    # tiny1 with the DenseHybrid `ood_pred` head
    "tiny1_dh": dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=6,
                     conv_dim=64, mask_dim=64, nheads=2, num_queries=16, num_classes=19,
                     dim_feedforward=128, enc_layers=2, dec_layers=1, enc_in=["res5"], dense_hybrid=True),
}

_DEFAULTS = dict(patch_size=4, mlp_ratio=4.0, enc_points=4, enc_dim_feedforward=1024,
                 pixel_mean=[123.675, 116.28, 103.53], pixel_std=[58.395, 57.12, 57.375],
                 size_divisibility=32, common_stride=4,
                 # DenseHybrid `ood_pred` head on the mask features (MODEL.MASK_FORMER.DENSE_HYBRID_LOSS, config.py:220)
                 dense_hybrid=False,
                 resnet=None,                     # dict(depth, stem_out, res2_out, width, stride_in_1x1) for build_resnet_backbone
                 # panoptic inference (maskformer_model.py:202-220; config.py:57-59, 243)
                 # sem_seg_postprocess on the mask logits BEFORE semantic inference (maskformer_model.py:205-209: the flag, or implied
                 # by panoptic / instance inference); only matters when a requested output size differs from the image size
                 postprocess_before_inference=False,
                 panoptic_on=False, open_panoptic=True, object_mask_threshold=0.0, overlap_threshold=0.0,
                 thing_classes=[11, 12, 13, 14, 15, 16, 17, 18])   # Cityscapes train ids of the "thing" classes


def complete(arch: dict) -> dict:
    a = dict(_DEFAULTS)
    a.update(arch)
    a["enc_in"] = sorted(a["enc_in"], key=FEATURE_NAMES.index)
    return a


def arch_from_cfg(cfg) -> dict:
    """Pull the consumed keys out of a (fully resolved) Detectron2-style config."""
    M = cfg.MODEL
    if M.META_ARCHITECTURE != "MaskFormer":
        raise ValueError(f"unsupported META_ARCHITECTURE {M.META_ARCHITECTURE!r}")
    if M.BACKBONE.NAME not in ("D2SwinTransformer", "build_resnet_backbone"):
        raise NotImplementedError(f"backbone {M.BACKBONE.NAME!r}: D2SwinTransformer and build_resnet_backbone are provided")
    if M.SEM_SEG_HEAD.PIXEL_DECODER_NAME != "MSDeformAttnPixelDecoder":
        raise NotImplementedError(M.SEM_SEG_HEAD.PIXEL_DECODER_NAME)
    if M.MASK_FORMER.TRANSFORMER_DECODER_NAME != "MultiScaleMaskedTransformerDecoder":
        raise NotImplementedError(M.MASK_FORMER.TRANSFORMER_DECODER_NAME)
    if M.MASK_FORMER.TRANSFORMER_IN_FEATURE != "multi_scale_pixel_decoder":
        raise NotImplementedError(M.MASK_FORMER.TRANSFORMER_IN_FEATURE)
    S = M.SWIN
    unsupported = []
    resnet = None
    if M.BACKBONE.NAME == "build_resnet_backbone":
        Rn = M.RESNETS
        if Rn.DEPTH not in RESNET_STAGE_BLOCKS or Rn.get("NUM_GROUPS", 1) != 1 or Rn.get("RES5_DILATION", 1) != 1:
            unsupported.append("RESNETS: only bottleneck depths 50/101/152, one group, no dilation")
        if any(Rn.get("DEFORM_ON_PER_STAGE", [False])):
            unsupported.append("RESNETS.DEFORM_ON_PER_STAGE")
        if list(Rn.OUT_FEATURES) != list(FEATURE_NAMES):
            unsupported.append("RESNETS.OUT_FEATURES != res2..res5")
        resnet = dict(depth=Rn.DEPTH, stem_out=Rn.STEM_OUT_CHANNELS, res2_out=Rn.get("RES2_OUT_CHANNELS", 256),
                      width=Rn.get("WIDTH_PER_GROUP", 64), stride_in_1x1=bool(Rn.STRIDE_IN_1X1))
    else:
        if S.APE:
            unsupported.append("SWIN.APE")
        if not S.PATCH_NORM:
            unsupported.append("SWIN.PATCH_NORM=False")
        if not S.QKV_BIAS:
            unsupported.append("SWIN.QKV_BIAS=False")
        if S.QK_SCALE is not None:
            unsupported.append("SWIN.QK_SCALE")
    if M.MASK_FORMER.PRE_NORM:
        unsupported.append("MASK_FORMER.PRE_NORM")
    if M.MASK_FORMER.ENFORCE_INPUT_PROJ:
        unsupported.append("MASK_FORMER.ENFORCE_INPUT_PROJ")
    if M.SEM_SEG_HEAD.NORM != "GN":
        unsupported.append(f"SEM_SEG_HEAD.NORM={M.SEM_SEG_HEAD.NORM!r}")
    if M.SEM_SEG_HEAD.CONVS_DIM != M.MASK_FORMER.HIDDEN_DIM:
        unsupported.append("CONVS_DIM != HIDDEN_DIM (decoder input_proj conv)")
    if list(M.SEM_SEG_HEAD.IN_FEATURES) != list(FEATURE_NAMES):
        unsupported.append("SEM_SEG_HEAD.IN_FEATURES != res2..res5")
    T = M.MASK_FORMER.TEST
    if not T.SEMANTIC_ON or T.INSTANCE_ON:
        unsupported.append("TEST.SEMANTIC_ON=False / TEST.INSTANCE_ON (only semantic and open-set panoptic inference are provided)")
    if cfg.get("SOLVER", {}).get("FORCE_REGION_PARTITION", False):
        unsupported.append("SOLVER.FORCE_REGION_PARTITION")
    if unsupported:
        raise NotImplementedError("config uses features outside the RbA hot path: " + ", ".join(unsupported))
    return complete(dict(
        embed_dim=S.EMBED_DIM, depths=list(S.DEPTHS), num_heads=list(S.NUM_HEADS), window_size=S.WINDOW_SIZE,
        patch_size=S.PATCH_SIZE, mlp_ratio=float(S.MLP_RATIO),
        conv_dim=M.SEM_SEG_HEAD.CONVS_DIM, mask_dim=M.SEM_SEG_HEAD.MASK_DIM, nheads=M.MASK_FORMER.NHEADS,
        num_queries=M.MASK_FORMER.NUM_OBJECT_QUERIES, num_classes=M.SEM_SEG_HEAD.NUM_CLASSES,
        dim_feedforward=M.MASK_FORMER.DIM_FEEDFORWARD, enc_layers=M.SEM_SEG_HEAD.TRANSFORMER_ENC_LAYERS,
        dec_layers=M.MASK_FORMER.DEC_LAYERS - 1,
        enc_in=list(M.SEM_SEG_HEAD.DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES),
        common_stride=M.SEM_SEG_HEAD.COMMON_STRIDE, size_divisibility=M.MASK_FORMER.SIZE_DIVISIBILITY,
        pixel_mean=list(M.PIXEL_MEAN), pixel_std=list(M.PIXEL_STD),
        dense_hybrid=bool(M.MASK_FORMER.get("DENSE_HYBRID_LOSS", False)), resnet=resnet,
        postprocess_before_inference=bool(T.get("SEM_SEG_POSTPROCESSING_BEFORE_INFERENCE", False) or T.PANOPTIC_ON or T.INSTANCE_ON),
        panoptic_on=bool(T.PANOPTIC_ON), open_panoptic=bool(M.MASK_FORMER.get("OPEN_PANOPTIC", True)),
        object_mask_threshold=float(T.OBJECT_MASK_THRESHOLD), overlap_threshold=float(T.OVERLAP_THRESHOLD)))


def num_fpn_levels(a: dict) -> int:
    """Extra FPN levels below the finest encoder level (msdeformattn.py:270-272)."""
    stride = min(FEATURE_STRIDES[f] for f in a["enc_in"])
    n = 0
    while stride > a["common_stride"]:
        stride //= 2
        n += 1
    return n


def _swin_shapes(a, out, lin, norm):
    f32, i64 = torch.float32, torch.int64
    E, ws, ps = a["embed_dim"], a["window_size"], a["patch_size"]
    out["backbone.patch_embed.proj.weight"] = ((E, 3, ps, ps), f32)
    out["backbone.patch_embed.proj.bias"] = ((E,), f32)
    norm("backbone.patch_embed.norm", E)
    for i, depth in enumerate(a["depths"]):
        C = E * 2 ** i
        nH = a["num_heads"][i]
        for b in range(depth):
            p = f"backbone.layers.{i}.blocks.{b}"
            norm(p + ".norm1", C)
            out[p + ".attn.relative_position_bias_table"] = (((2 * ws - 1) ** 2, nH), f32)
            out[p + ".attn.relative_position_index"] = ((ws * ws, ws * ws), i64)
            lin(p + ".attn.qkv", 3 * C, C)
            lin(p + ".attn.proj", C, C)
            norm(p + ".norm2", C)
            lin(p + ".mlp.fc1", int(C * a["mlp_ratio"]), C)
            lin(p + ".mlp.fc2", C, int(C * a["mlp_ratio"]))
        if i < len(a["depths"]) - 1:
            lin(f"backbone.layers.{i}.downsample.reduction", 2 * C, 4 * C, bias=False)
            norm(f"backbone.layers.{i}.downsample.norm", 4 * C)
        norm(f"backbone.norm{i}", C)



def state_dict_shapes(arch: dict) -> dict:
    """{key: (shape, dtype)} of the reference MaskFormer state dict restricted to this path
    (everything except ``criterion.empty_weight``)."""
    a = complete(arch)
    f32, i64 = torch.float32, torch.int64
    out = {}

    def lin(p, o, i, bias=True):
        out[p + ".weight"] = ((o, i), f32)
        if bias:
            out[p + ".bias"] = ((o,), f32)

    def norm(p, c):
        out[p + ".weight"] = ((c,), f32)
        out[p + ".bias"] = ((c,), f32)

    if a.get("resnet"):
        r = a["resnet"]

        def convbn(p, co, ci, k):              # Detectron2 Conv2d(bias=False, norm=BN): weight + norm.{weight,bias,running_*}
            out[p + ".weight"] = ((co, ci, k, k), f32)
            norm(p + ".norm", co)
            out[p + ".norm.running_mean"] = ((co,), f32)
            out[p + ".norm.running_var"] = ((co,), f32)
            out[p + ".norm.num_batches_tracked"] = ((), i64)

        convbn("backbone.stem.conv1", r["stem_out"], 3, 7)
        cin, cout, bott = r["stem_out"], r["res2_out"], r["width"]
        for i, nblocks in enumerate(RESNET_STAGE_BLOCKS[r["depth"]]):
            for b in range(nblocks):
                p = f"backbone.res{i + 2}.{b}"
                if cin != cout:
                    convbn(p + ".shortcut", cout, cin, 1)
                convbn(p + ".conv1", bott, cin, 1)
                convbn(p + ".conv2", bott, bott, 3)
                convbn(p + ".conv3", cout, bott, 1)
                cin = cout
            cout, bott = cout * 2, bott * 2
    else:
        _swin_shapes(a, out, lin, norm)

    d, md, M = a["conv_dim"], a["mask_dim"], a["nheads"]
    L, P = len(a["enc_in"]), a["enc_points"]
    pd = "sem_seg_head.pixel_decoder"
    chans = feature_channels(a)
    for idx, f in enumerate(a["enc_in"][::-1]):
        out[f"{pd}.input_proj.{idx}.0.weight"] = ((d, chans[f], 1, 1), f32)
        out[f"{pd}.input_proj.{idx}.0.bias"] = ((d,), f32)
        norm(f"{pd}.input_proj.{idx}.1", d)
    out[f"{pd}.transformer.level_embed"] = ((L, d), f32)
    for i in range(a["enc_layers"]):
        p = f"{pd}.transformer.encoder.layers.{i}"
        lin(p + ".self_attn.sampling_offsets", M * L * P * 2, d)
        lin(p + ".self_attn.attention_weights", M * L * P, d)
        lin(p + ".self_attn.value_proj", d, d)
        lin(p + ".self_attn.output_proj", d, d)
        norm(p + ".norm1", d)
        lin(p + ".linear1", a["enc_dim_feedforward"], d)
        lin(p + ".linear2", d, a["enc_dim_feedforward"])
        norm(p + ".norm2", d)
    out[f"{pd}.mask_features.weight"] = ((md, d, 1, 1), f32)
    out[f"{pd}.mask_features.bias"] = ((md,), f32)
    for j in range(1, num_fpn_levels(a) + 1):
        out[f"{pd}.adapter_{j}.weight"] = ((d, chans[FEATURE_NAMES[j - 1]], 1, 1), f32)
        norm(f"{pd}.adapter_{j}.norm", d)
        out[f"{pd}.layer_{j}.weight"] = ((d, d, 3, 3), f32)
        norm(f"{pd}.layer_{j}.norm", d)

    pr = "sem_seg_head.predictor"
    for i in range(a["dec_layers"]):
        for kind, attn in (("cross", "multihead_attn"), ("self", "self_attn")):
            p = f"{pr}.transformer_{kind}_attention_layers.{i}"
            out[f"{p}.{attn}.in_proj_weight"] = ((3 * d, d), f32)
            out[f"{p}.{attn}.in_proj_bias"] = ((3 * d,), f32)
            lin(f"{p}.{attn}.out_proj", d, d)
            norm(p + ".norm", d)
        p = f"{pr}.transformer_ffn_layers.{i}"
        lin(p + ".linear1", a["dim_feedforward"], d)
        lin(p + ".linear2", d, a["dim_feedforward"])
        norm(p + ".norm", d)
    norm(pr + ".decoder_norm", d)
    out[pr + ".query_feat.weight"] = ((a["num_queries"], d), f32)
    out[pr + ".query_embed.weight"] = ((a["num_queries"], d), f32)
    out[pr + ".level_embed.weight"] = ((L, d), f32)
    lin(pr + ".class_embed", a["num_classes"] + 1, d)
    lin(pr + ".mask_embed.layers.0", d, d)
    lin(pr + ".mask_embed.layers.1", d, d)
    lin(pr + ".mask_embed.layers.2", md, d)
    if a["dense_hybrid"]:                  # BNReluConv(hidden_dim, 2, k=1, bias=True) (mask2former_transformer_decoder.py:216-230, 365-366)
        norm(pr + ".ood_pred.norm", d)
        out[pr + ".ood_pred.norm.running_mean"] = ((d,), f32)
        out[pr + ".ood_pred.norm.running_var"] = ((d,), f32)
        out[pr + ".ood_pred.norm.num_batches_tracked"] = ((), i64)
        out[pr + ".ood_pred.conv.weight"] = ((2, d, 1, 1), f32)
        out[pr + ".ood_pred.conv.bias"] = ((2,), f32)
    return out


def seeded_weights(arch: dict, seed: int = 0, recipe: str = None) -> dict:
    """Random-init weights of ``arch`` by the deterministic recipe of ``seeded_weights.py`` (``recipe="heavy"``: its trained-like
    stress variant)."""
    from .seeded_weights import seeded_state_dict

    a = complete(arch)
    meta = dict(n_heads=a["nheads"], n_points=a["enc_points"])
    if recipe is not None:
        meta["recipe"] = recipe
    return seeded_state_dict(state_dict_shapes(a), seed, meta=meta)
