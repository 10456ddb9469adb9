"""ORACLE -- test infrastructure, not product code.

Functional CPU restatement (torch fp32) of RbA's inference forward, driven by a flat
state dict with the reference's key names.  Each function cites the reference lines
(relative to /root/reference) it follows.  Pinned by ``tests/golden/g4_*.npz`` and
``g5_*.npz`` which the reference's own modules produced from the same seeded weights.

``arch`` keys: embed_dim, depths, num_heads, window_size, conv_dim, mask_dim, nheads,
num_queries, num_classes, dim_feedforward, enc_layers, dec_layers (= DEC_LAYERS - 1),
enc_in (DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES).
"""
import torch
import torch.nn.functional as F

from . import ref_ops as R

PIXEL_MEAN = [123.675, 116.28, 103.53]   # ckpts/swin_b_1dl/config.yaml:201-208
PIXEL_STD = [58.395, 57.12, 57.375]


def _ln(x, sd, p, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


# ----------------------------------------------------------------------------- backbone
def swin_block(x, H, W, sd, p, ws, shift, nheads, attn_mask):
    """SwinTransformerBlock.forward (backbone/swin.py:235-295)."""
    B, L, C = x.shape
    shortcut = x
    x = _ln(x, sd, p + ".norm1").view(B, H, W, C)
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
    Hp, Wp = x.shape[1], x.shape[2]
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    xw = x.view(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    aw = R.window_attention(xw, sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"],
                            sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"],
                            sd[p + ".attn.relative_position_bias_table"], ws, nheads,
                            attn_mask if shift > 0 else None)
    x = aw.view(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    x = x[:, :H, :W, :].reshape(B, H * W, C)
    x = shortcut + x
    y = _lin(F.gelu(_lin(_ln(x, sd, p + ".norm2"), sd, p + ".mlp.fc1")), sd, p + ".mlp.fc2")   # Mlp :35-41
    return x + y


def patch_merging(x, H, W, sd, p):
    """PatchMerging.forward (backbone/swin.py:311-337)."""
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    if H % 2 == 1 or W % 2 == 1:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    x = x.view(B, -1, 4 * C)
    return F.linear(_ln(x, sd, p + ".norm"), sd[p + ".reduction.weight"])


def swin_backbone(x, sd, a, p="backbone"):
    """PatchEmbed (swin.py:479-495) + SwinTransformer.forward (:651-678) + BasicLayer.forward (:406-453)."""
    ps = 4
    _, _, H, W = x.shape
    if W % ps:
        x = F.pad(x, (0, ps - W % ps))
    if H % ps:
        x = F.pad(x, (0, 0, 0, ps - H % ps))
    x = F.conv2d(x, sd[p + ".patch_embed.proj.weight"], sd[p + ".patch_embed.proj.bias"], stride=ps)
    Wh, Ww = x.shape[2], x.shape[3]
    x = _ln(x.flatten(2).transpose(1, 2), sd, p + ".patch_embed.norm")
    ws = a["window_size"]
    outs = {}
    for i, depth in enumerate(a["depths"]):
        C = a["embed_dim"] * 2 ** i
        mask = R.shift_attn_mask(Wh, Ww, ws, ws // 2)
        for b in range(depth):
            x = swin_block(x, Wh, Ww, sd, f"{p}.layers.{i}.blocks.{b}", ws, 0 if b % 2 == 0 else ws // 2,
                           a["num_heads"][i], mask)
        out = _ln(x, sd, f"{p}.norm{i}")
        outs[f"res{i + 2}"] = out.view(-1, Wh, Ww, C).permute(0, 3, 1, 2).contiguous()
        if i < len(a["depths"]) - 1:
            x = patch_merging(x, Wh, Ww, sd, f"{p}.layers.{i}.downsample")
            Wh, Ww = (Wh + 1) // 2, (Ww + 1) // 2
    return outs


def resnet_backbone(x, sd, a, p="backbone"):
    """Detectron2 v0.6 ResNet (modeling/backbone/resnet.py: BasicStem, BottleneckBlock, build_resnet_backbone) as configured by
    configs/cityscapes/semantic-segmentation/Base-Cityscapes-SemanticSegmentation.yaml:8-15, inference mode.  Detectron2 is not in
    the reference tree: restated from its published definition -- PARITY UNPINNED (no reference-generated fixture exists)."""
    r = a["resnet"]

    def convbn(x, q, stride=1, padding=0, relu=False):
        y = F.conv2d(x, sd[q + ".weight"], None, stride=stride, padding=padding)
        y = F.batch_norm(y, sd[q + ".norm.running_mean"], sd[q + ".norm.running_var"], sd[q + ".norm.weight"], sd[q + ".norm.bias"],
                         False, 0.1, 1e-5)
        return F.relu(y) if relu else y

    x = F.max_pool2d(convbn(x, p + ".stem.conv1", 2, 3, True), kernel_size=3, stride=2, padding=1)
    outs = {}
    blocks = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}[r["depth"]]
    for i, nb in enumerate(blocks):
        for b in range(nb):
            q = f"{p}.res{i + 2}.{b}"
            stride = (1 if i == 0 else 2) if b == 0 else 1
            s1, s3 = (stride, 1) if r["stride_in_1x1"] else (1, stride)
            out = convbn(x, q + ".conv1", s1, 0, True)
            out = convbn(out, q + ".conv2", s3, 1, True)
            out = convbn(out, q + ".conv3")
            short = convbn(x, q + ".shortcut", stride) if (q + ".shortcut.weight") in sd else x
            x = F.relu(out + short)
        outs[f"res{i + 2}"] = x
    return outs


# ------------------------------------------------------------------------ pixel decoder
def _gn(x, sd, p, groups=32):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def ms_deform_attn_module(query, ref_points, src, shapes, lsi, sd, p, M, L, P):
    """MSDeformAttn.forward (pixel_decoder/ops/modules/ms_deform_attn.py:82-125), 2-d reference points."""
    N, Lq, C = query.shape
    value = _lin(src, sd, p + ".value_proj").view(N, src.shape[1], M, C // M)
    off = _lin(query, sd, p + ".sampling_offsets").view(N, Lq, M, L, P, 2)
    w = F.softmax(_lin(query, sd, p + ".attention_weights").view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    normalizer = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
    loc = ref_points[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    out = R.ms_deform_attn(value, shapes, loc, w)
    return _lin(out, sd, p + ".output_proj")


def encoder_reference_points(shapes):
    """MSDeformAttnTransformerEncoder.get_reference_points with valid_ratios == 1 (msdeformattn.py:149-162)."""
    refs = []
    for H_, W_ in shapes.tolist():
        ry, rx = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_), torch.linspace(0.5, W_ - 0.5, W_), indexing="ij")
        refs.append(torch.stack((rx.reshape(-1)[None] / W_, ry.reshape(-1)[None] / H_), -1))
    ref = torch.cat(refs, 1)
    return ref[:, :, None].repeat(1, 1, len(shapes), 1)


def pixel_decoder(feats, sd, a, p="sem_seg_head.pixel_decoder"):
    """MSDeformAttnPixelDecoder.forward_features (pixel_decoder/msdeformattn.py:323-367) with
    MSDeformAttnTransformerEncoderOnly.forward (:70-98) and the encoder layer (:131-140)."""
    names = ["res2", "res3", "res4", "res5"]
    enc_in = sorted(a["enc_in"], key=names.index)
    Lv = len(enc_in)
    d = a["conv_dim"]
    srcs, poss = [], []
    for idx, f in enumerate(enc_in[::-1]):
        x = feats[f].to(torch.get_default_dtype())          # the reference says .float(); the default dtype is float32 unless a test asks for a float64 truth run
        s = F.conv2d(x, sd[f"{p}.input_proj.{idx}.0.weight"], sd[f"{p}.input_proj.{idx}.0.bias"])
        srcs.append(_gn(s, sd, f"{p}.input_proj.{idx}.1"))
        poss.append(R.position_embedding_sine(x.shape[2], x.shape[3], d // 2)[None])
    shapes = torch.as_tensor([s.shape[-2:] for s in srcs], dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
    pos = torch.cat([po.flatten(2).transpose(1, 2) + sd[f"{p}.transformer.level_embed"][l].view(1, 1, -1)
                     for l, po in enumerate(poss)], 1)
    ref = encoder_reference_points(shapes)
    out = src
    for i in range(a["enc_layers"]):
        lp = f"{p}.transformer.encoder.layers.{i}"
        s2 = ms_deform_attn_module(out + pos, ref, out, shapes, lsi, sd, lp + ".self_attn", a["nheads"], Lv, 4)
        out = _ln(out + s2, sd, lp + ".norm1")
        s2 = _lin(F.relu(_lin(out, sd, lp + ".linear1")), sd, lp + ".linear2")
        out = _ln(out + s2, sd, lp + ".norm2")
    bs = out.shape[0]
    sizes = [int(h * w) for h, w in shapes.tolist()]
    outs = [z.transpose(1, 2).reshape(bs, -1, int(shapes[i][0]), int(shapes[i][1]))
            for i, z in enumerate(torch.split(out, sizes, dim=1))]
    min_stride = min(4 * 2 ** names.index(f) for f in enc_in)
    num_fpn = {4: 0, 8: 1, 16: 2, 32: 3}[min_stride]
    for idx, f in enumerate(names[:num_fpn][::-1]):
        j = num_fpn - idx          # adapter_{j} / layer_{j}: adapter_1 = res2 (msdeformattn.py:293-301)
        x = feats[f].to(torch.get_default_dtype())          # the reference says .float(); the default dtype is float32 unless a test asks for a float64 truth run
        cur = _gn(F.conv2d(x, sd[f"{p}.adapter_{j}.weight"]), sd, f"{p}.adapter_{j}.norm")
        y = cur + F.interpolate(outs[-1], size=cur.shape[-2:], mode="bilinear", align_corners=False)
        y = F.relu(_gn(F.conv2d(y, sd[f"{p}.layer_{j}.weight"], padding=1), sd, f"{p}.layer_{j}.norm"))
        outs.append(y)
    mask_features = F.conv2d(outs[-1], sd[f"{p}.mask_features.weight"], sd[f"{p}.mask_features.bias"])
    return mask_features, outs[:Lv]


# ------------------------------------------------------------------- transformer decoder
def prediction_heads(output, mask_features, size, sd, a, p, toggle=None, tap=None):
    """forward_prediction_heads (transformer_decoder/mask2former_transformer_decoder.py:472-489).
    Test hooks (no effect by default): `tap` (a list) receives the interpolated attention-mask logits [B, Q, h*w] this call thresholds; `toggle` (flat indices
    into them) inverts the threshold decision of those entries -- tests/test_shape_sweep_gpu.py uses the pair to show that a product / reference difference is the
    reference's own discontinuity at `sigmoid(logit) < 0.5` (a logit within rounding noise of 0) and nothing else."""
    dec = _ln(output, sd, p + ".decoder_norm").transpose(0, 1)
    cls = _lin(dec, sd, p + ".class_embed")
    me = dec
    for i in range(3):
        me = _lin(me, sd, f"{p}.mask_embed.layers.{i}")
        if i < 2:
            me = F.relu(me)
    masks = torch.einsum("bqc,bchw->bqhw", me, mask_features)
    am = F.interpolate(masks, size=size, mode="bilinear", align_corners=False)
    if tap is not None:
        tap.append(am.flatten(2).clone())
    dec = am.sigmoid().flatten(2) < 0.5
    if toggle is not None and len(toggle):
        dec.view(-1)[toggle] ^= True
    am = dec.unsqueeze(1).repeat(1, a["nheads"], 1, 1).flatten(0, 1).bool()
    return cls, masks, am


def transformer_decoder(multi_scale, mask_features, sd, a, p="sem_seg_head.predictor", taps=None, toggles=None):
    """MultiScaleMaskedTransformerDecoder.forward (mask2former_transformer_decoder.py:398-470), post-norm.
    toggles: {head call index (0 = before layer 0): flat indices} for prediction_heads' test hook; taps["am_logits"] lists every call's thresholded logits."""
    toggles = toggles or {}
    am_tap = [] if taps is not None else None
    Lv = len(multi_scale)
    d, nh = a["conv_dim"], a["nheads"]
    src, pos, sizes = [], [], []
    for i, x in enumerate(multi_scale):
        sizes.append(tuple(x.shape[-2:]))
        pos.append(R.position_embedding_sine(x.shape[2], x.shape[3], d // 2)[None].flatten(2).permute(2, 0, 1))
        s = x.flatten(2) + sd[p + ".level_embed.weight"][i][None, :, None]   # input_proj = identity (:353-358)
        src.append(s.permute(2, 0, 1))
    bs = src[0].shape[1]
    qe = sd[p + ".query_embed.weight"].unsqueeze(1).repeat(1, bs, 1)
    out = sd[p + ".query_feat.weight"].unsqueeze(1).repeat(1, bs, 1)
    cls, masks, am = prediction_heads(out, mask_features, sizes[0], sd, a, p, toggles.get(0), am_tap)
    aux = [(cls, masks, am)]
    for i in range(a["dec_layers"]):
        li = i % Lv
        am = am.clone()
        am[torch.where(am.sum(-1) == am.shape[-1])] = False           # :433
        cp = f"{p}.transformer_cross_attention_layers.{i}"
        t2 = R.multihead_attention(out + qe, src[li] + pos[li], src[li], sd[cp + ".multihead_attn.in_proj_weight"],
                                   sd[cp + ".multihead_attn.in_proj_bias"], sd[cp + ".multihead_attn.out_proj.weight"],
                                   sd[cp + ".multihead_attn.out_proj.bias"], nh, am)
        out = _ln(out + t2, sd, cp + ".norm")
        sp = f"{p}.transformer_self_attention_layers.{i}"
        t2 = R.multihead_attention(out + qe, out + qe, out, sd[sp + ".self_attn.in_proj_weight"],
                                   sd[sp + ".self_attn.in_proj_bias"], sd[sp + ".self_attn.out_proj.weight"],
                                   sd[sp + ".self_attn.out_proj.bias"], nh, None)
        out = _ln(out + t2, sd, sp + ".norm")
        fp = f"{p}.transformer_ffn_layers.{i}"
        t2 = _lin(F.relu(_lin(out, sd, fp + ".linear1")), sd, fp + ".linear2")
        out = _ln(out + t2, sd, fp + ".norm")
        cls, masks, am = prediction_heads(out, mask_features, sizes[(i + 1) % Lv], sd, a, p, toggles.get(i + 1), am_tap)
        aux.append((cls, masks, am))
    if taps is not None:
        taps["aux"] = aux
        taps["am_logits"] = am_tap
    return cls, masks


def ood_pred_head(mask_features, sd, p="sem_seg_head.predictor.ood_pred"):
    """DenseHybrid head BNReluConv(hidden_dim, 2, k=1, bias=True) on the mask features, inference mode
    (mask2former_transformer_decoder.py:216-230, 365-366, 467-468): BatchNorm2d (running statistics) -> ReLU -> 1x1 conv."""
    x = F.batch_norm(mask_features, sd[p + ".norm.running_mean"], sd[p + ".norm.running_var"], sd[p + ".norm.weight"],
                     sd[p + ".norm.bias"], False, 0.01, 1e-5)
    return F.conv2d(F.relu(x), sd[p + ".conv.weight"], sd[p + ".conv.bias"])


# ------------------------------------------------------------------------------ meta arch
@torch.no_grad()
def forward(image, sd, a, taps=None, canvas=None, toggles=None):
    """MaskFormer.forward inference branch (mask2former/maskformer_model.py:255-260, 290-333) for ONE image
    [3,h,w] (uint8 or float, 0..255) followed by get_RbA (evaluate_ood.py:143-150) and the argmax of
    support.py:385-388.  Returns dict(pred_logits, pred_masks, sem_seg, rba, argmax).
    canvas = (H, W): the image as a member of a batch whose ImageList is padded to that common size (a multiple of 32 not smaller than the
    image, maskformer_model.py:257); default = the image's own size rounded up to 32."""
    mean = torch.tensor(PIXEL_MEAN).view(-1, 1, 1)
    std = torch.tensor(PIXEL_STD).view(-1, 1, 1)
    x = (image.to(torch.get_default_dtype()) - mean) / std
    h, w = x.shape[-2:]
    H, W = (h + 31) // 32 * 32, (w + 31) // 32 * 32
    if canvas is not None:
        assert canvas[0] % 32 == 0 and canvas[1] % 32 == 0 and canvas[0] >= H and canvas[1] >= W
        H, W = canvas
    x = F.pad(x, (0, W - w, 0, H - h))[None]
    feats = resnet_backbone(x, sd, a) if a.get("resnet") else swin_backbone(x, sd, a)
    mask_features, multi_scale = pixel_decoder(feats, sd, a)
    cls, masks = transformer_decoder(multi_scale, mask_features, sd, a, taps=taps, toggles=toggles)
    up = R.upsample_bilinear(masks, (H, W))[0]
    sem = R.semantic_inference(cls[0], up)[:, :h, :w]
    if taps is not None:
        taps.update(feats=feats, mask_features=mask_features, multi_scale=multi_scale)
    out = dict(pred_logits=cls[0], pred_masks=masks[0], sem_seg=sem, rba=R.rba_score(sem), argmax=sem.max(dim=0)[1])
    if a.get("dense_hybrid", False):
        # maskformer_model.py:303-305 (up-sampled to the image size with align_corners=True) + evaluate_ood.py:161-173
        ood = F.interpolate(ood_pred_head(mask_features, sd), size=(h, w), mode="bilinear", align_corners=True)
        out.update(ood_pred=ood, densehybrid=R.densehybrid_score(sem, ood))
    return out
