#!/usr/bin/env python3
"""Golden-vector generator.  RUNS ONLY IN THE BUILD CONTAINER (needs /root/reference).

It imports the reference's own hot-path modules from /root/reference
(backbone/swin.py, pixel_decoder/msdeformattn.py, pixel_decoder/ops/{modules,functions},
transformer_decoder/{mask2former_transformer_decoder,position_encoding}.py,
meta_arch/mask_former_head.py), fills their parameters with the deterministic recipe of
``rba_amd/seeded_weights.py``, runs them on CPU and stores inputs + outputs as small
``.npz`` fixtures next to this file.  Nothing of the reference's source is written out:
a fixture is arrays only.

Third-party names the reference imports but this image lacks (detectron2, timm, fvcore)
are satisfied with import-only stand-ins in ``_install_import_stubs`` so that the
reference files can be *imported*; the only arithmetic in them is Detectron2's
``Conv2d`` wrapper (conv -> norm -> activation) and ``get_norm("GN") = GroupNorm(32, C)``
(Detectron2 v0.6 ``layers/wrappers.py`` / ``layers/batch_norm.py`` behaviour).  Those two,
plus the glue of ``maskformer_model.py`` that cannot be imported (it needs
detectron2.structures / cv2), are restated here; DESIGN.md lists them as "parity
unpinned by the reference's own code".  Everything else in a fixture is produced by the
reference's code itself.

    python tests/golden/make_golden.py [--only NAME] [--full]

``--full`` additionally generates the two full-size (Swin-B) sampled fixtures (about
2 minutes of CPU).
"""
import argparse
import importlib
import os
import sys
import types
import zlib

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from rba_amd.seeded_weights import fill_state_dict_, seeded_ood_labels as parity_labels  # noqa: E402


# --------------------------------------------------------------------------------------
# import-only stand-ins for absent third-party packages
# --------------------------------------------------------------------------------------
def _install_import_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    # timm.models.layers: DropPath is identity in eval, to_2tuple, trunc_normal_ (init only)
    class DropPath(nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    def trunc_normal_(t, std=1.0, **kw):
        return t

    mod("timm")
    mod("timm.models")
    mod("timm.models.layers", DropPath=DropPath, to_2tuple=to_2tuple, trunc_normal_=trunc_normal_)

    class Registry(dict):
        def __init__(self, name=""):
            super().__init__()

        def register(self, obj=None):
            def deco(o):
                self[o.__name__] = o
                return o

            return deco if obj is None else deco(obj)

        def get(self, name):
            return self[name]

    class ShapeSpec:
        def __init__(self, channels=None, height=None, width=None, stride=None):
            self.channels, self.height, self.width, self.stride = channels, height, width, stride

    class Backbone(nn.Module):
        pass

    class Conv2d(nn.Conv2d):  # Detectron2 layers/wrappers.py: conv -> norm -> activation
        def __init__(self, *a, **kw):
            norm = kw.pop("norm", None)
            activation = kw.pop("activation", None)
            super().__init__(*a, **kw)
            self.norm = norm
            self.activation = activation

        def forward(self, x):
            x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
            if self.norm is not None:
                x = self.norm(x)
            if self.activation is not None:
                x = self.activation(x)
            return x

    def get_norm(norm, out_channels):
        if norm is None or norm == "":
            return None
        assert norm == "GN", norm
        return nn.GroupNorm(32, out_channels)

    def configurable(init_func=None, *, from_config=None):
        return init_func  # construct with explicit keyword arguments

    mod("detectron2")
    mod("detectron2.config", configurable=configurable)
    mod("detectron2.layers", Conv2d=Conv2d, ShapeSpec=ShapeSpec, get_norm=get_norm, DeformConv=None)
    mod(
        "detectron2.modeling",
        BACKBONE_REGISTRY=Registry(),
        SEM_SEG_HEADS_REGISTRY=Registry(),
        META_ARCH_REGISTRY=Registry(),
        Backbone=Backbone,
        ShapeSpec=ShapeSpec,
    )
    mod("detectron2.utils")
    mod("detectron2.utils.registry", Registry=Registry)
    mod("fvcore")
    mod("fvcore.nn")
    wi = mod("fvcore.nn.weight_init", c2_xavier_fill=lambda m: None, c2_msra_fill=lambda m: None)
    sys.modules["fvcore.nn"].weight_init = wi
    # empty native module: ms_deform_attn_func.py imports it at module scope; MSDeformAttn.forward
    # then falls through its bare except to ms_deform_attn_core_pytorch (the reference CPU path).
    mod("MultiScaleDeformableAttention")

    # namespace packages so the reference's package __init__ chain (datasets, cv2, ...) never runs
    def ns(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m

    base = os.path.join(REF, "mask2former")
    ns("mask2former", base)
    ns("mask2former.modeling", os.path.join(base, "modeling"))
    for sub in ("backbone", "pixel_decoder", "transformer_decoder", "meta_arch"):
        ns("mask2former.modeling." + sub, os.path.join(base, "modeling", sub))
    ns("mask2former.modeling.pixel_decoder.ops", os.path.join(base, "modeling", "pixel_decoder", "ops"))
    return ShapeSpec


def _ref_modules():
    ShapeSpec = _install_import_stubs()
    swin = importlib.import_module("mask2former.modeling.backbone.swin")
    msda = importlib.import_module("mask2former.modeling.pixel_decoder.msdeformattn")
    dec = importlib.import_module("mask2former.modeling.transformer_decoder.mask2former_transformer_decoder")
    pe = importlib.import_module("mask2former.modeling.transformer_decoder.position_encoding")
    fn = importlib.import_module("mask2former.modeling.pixel_decoder.ops.functions.ms_deform_attn_func")
    return types.SimpleNamespace(swin=swin, msda=msda, dec=dec, pe=pe, fn=fn, ShapeSpec=ShapeSpec)


# --------------------------------------------------------------------------------------
# model assembly from reference classes (explicit kwargs = what from_config would pass)
# --------------------------------------------------------------------------------------
ARCHS = {
    # tiny nets: every code path (window pad, shift, odd patch-merge, image pad) at toy cost
    "tiny1": dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=6,
                  conv_dim=64, mask_dim=64, nheads=2, num_queries=16, num_classes=19,
                  dim_feedforward=128, enc_layers=2, dec_layers=1, enc_in=["res5"]),
    "tiny3": dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=6,
                  conv_dim=64, mask_dim=64, nheads=2, num_queries=16, num_classes=19,
                  dim_feedforward=128, enc_layers=2, dec_layers=4, enc_in=["res3", "res4", "res5"]),
    "tiny1_dh": dict(embed_dim=32, depths=[2, 2, 2, 2], num_heads=[1, 2, 4, 8], window_size=6,
                     conv_dim=64, mask_dim=64, nheads=2, num_queries=16, num_classes=19,
                     dim_feedforward=128, enc_layers=2, dec_layers=1, enc_in=["res5"], dense_hybrid=True),
    # ckpts/swin_b_1dl/config.yaml
    "swin_b_1dl": dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=12,
                       conv_dim=256, mask_dim=256, nheads=8, num_queries=100, num_classes=19,
                       dim_feedforward=2048, enc_layers=6, dec_layers=1, enc_in=["res5"]),
    # ckpts/swin_l_1dl/config.yaml
    "swin_l_1dl": dict(embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], window_size=12,
                       conv_dim=256, mask_dim=256, nheads=8, num_queries=100, num_classes=19,
                       dim_feedforward=2048, enc_layers=6, dec_layers=1, enc_in=["res5"]),
    # configs/.../swin/all_decoder_layers/maskformer2_swin_base_IN21k_384_bs16_90k.yaml (DEC_LAYERS 10)
    "swin_b_9dl": dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=12,
                       conv_dim=256, mask_dim=256, nheads=8, num_queries=100, num_classes=19,
                       dim_feedforward=2048, enc_layers=6, dec_layers=9, enc_in=["res3", "res4", "res5"]),
}

PIXEL_MEAN = [123.675, 116.28, 103.53]
PIXEL_STD = [58.395, 57.12, 57.375]


class RefModel(nn.Module):
    """backbone + sem_seg_head(pixel_decoder, predictor) under the reference's attribute names, so
    state_dict() keys equal those of the reference MaskFormer (minus criterion.empty_weight)."""

    def __init__(self, R, a):
        super().__init__()
        self.backbone = R.swin.SwinTransformer(
            pretrain_img_size=384, patch_size=4, in_chans=3, embed_dim=a["embed_dim"], depths=a["depths"],
            num_heads=a["num_heads"], window_size=a["window_size"], mlp_ratio=4.0, qkv_bias=True,
            qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.3, ape=False,
            patch_norm=True, out_indices=(0, 1, 2, 3), use_checkpoint=False)
        shapes = {f"res{i + 2}": R.ShapeSpec(channels=a["embed_dim"] * 2 ** i, stride=4 * 2 ** i) for i in range(4)}
        head = nn.Module()
        head.pixel_decoder = R.msda.MSDeformAttnPixelDecoder(
            shapes, transformer_dropout=0.0, transformer_nheads=a["nheads"], transformer_dim_feedforward=1024,
            transformer_enc_layers=a["enc_layers"], conv_dim=a["conv_dim"], mask_dim=a["mask_dim"], norm="GN",
            transformer_in_features=a["enc_in"], common_stride=4)
        head.predictor = R.dec.MultiScaleMaskedTransformerDecoder(
            a["conv_dim"], True, num_classes=a["num_classes"], hidden_dim=a["conv_dim"],
            num_queries=a["num_queries"], nheads=a["nheads"], dim_feedforward=a["dim_feedforward"],
            dec_layers=a["dec_layers"], pre_norm=False, mask_dim=a["mask_dim"], enforce_input_project=False,
            ood_prediction=bool(a.get("dense_hybrid", False)), num_feature_levels=len(a["enc_in"]))
        self.sem_seg_head = head


@torch.no_grad()
def ref_forward(model, image, taps=None):
    """Restatement of maskformer_model.py:255-260, 290-299, 328-333, 381-386 and
    evaluate_ood.py:143-150 around the reference's own backbone/head modules."""
    mean = torch.tensor(PIXEL_MEAN).view(-1, 1, 1)
    std = torch.tensor(PIXEL_STD).view(-1, 1, 1)
    x = (image.float() - mean) / std                       # :255-256
    h, w = x.shape[-2:]
    H, W = (h + 31) // 32 * 32, (w + 31) // 32 * 32        # ImageList.from_tensors(.., 32): pad bottom/right with 0
    x = F.pad(x, (0, W - w, 0, H - h))[None]
    feats = model.backbone(x)                              # :259
    mask_features, _, multi_scale = model.sem_seg_head.pixel_decoder.forward_features(feats)
    out = model.sem_seg_head.predictor(multi_scale, mask_features, None)   # mask_former_head.py:128-133
    mask_cls, mask_pred = out["pred_logits"][0], out["pred_masks"]
    up = F.interpolate(mask_pred, size=(H, W), mode="bilinear", align_corners=False)[0]   # :294-299
    sem = torch.einsum("qc,qhw->chw", F.softmax(mask_cls, dim=-1)[..., :-1], up.sigmoid())  # :381-386
    sem = sem[:, :h, :w]                                   # sem_seg_postprocess: crop, identity resize
    rba = -sem.tanh().sum(dim=0)                           # evaluate_ood.py:150
    if taps is not None:
        taps.update(feats=feats, mask_features=mask_features, multi_scale=multi_scale, out=out)
    res = dict(pred_logits=mask_cls, pred_masks=mask_pred[0], sem_seg=sem, rba=rba, argmax=sem.argmax(0))
    if "ood_pred" in out:                                  # maskformer_model.py:303-305 + evaluate_ood.py:161-173
        ood = F.interpolate(out["ood_pred"], size=(h, w), mode="bilinear", align_corners=True)
        p1 = torch.logsumexp(sem, dim=0)
        p2 = F.softmax(ood, dim=1)[:, 1]
        res.update(ood_pred_low=out["ood_pred"][0], ood_pred=ood[0], densehybrid=((-p1) + (p2 + 1e-9).log())[0])
    return res


def rand_image(h, w, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8)


def np_(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def attn_mask_margin(model, taps):
    """min |bilinear-downsampled mask logit| over every place the decoder thresholds sigmoid(x) < 0.5
    (mask2former_transformer_decoder.py:483-486) -- a parity test can only be bit-stable if this margin is
    far above fp32 noise."""
    out = taps["out"]
    sizes = [f.shape[-2:] for f in taps["multi_scale"]]
    masks = [a["pred_masks"] for a in out["aux_outputs"]]
    m = float("inf")
    for i, pm in enumerate(masks):
        tgt = sizes[i % len(sizes)]
        lo = F.interpolate(pm, size=tgt, mode="bilinear", align_corners=False)
        m = min(m, lo.abs().min().item())
    return m


# --------------------------------------------------------------------------------------
# fixtures
# --------------------------------------------------------------------------------------
def g_rba_reduce(R):
    """G1: K1 inputs/outputs at 100 x 16 x 32 (maskformer_model.py:381-386 + evaluate_ood.py:150)."""
    g = torch.Generator().manual_seed(0)
    mask_cls = torch.randn(100, 20, generator=g) * 3
    mask_pred = torch.randn(100, 16, 32, generator=g) * 5
    p = F.softmax(mask_cls, dim=-1)[..., :-1]
    sem = torch.einsum("qc,qhw->chw", p, mask_pred.sigmoid())
    rba = -sem.tanh().sum(0)
    # x4 upsample path (maskformer_model.py:294-299) from a 4x8 low-res map
    low = torch.randn(100, 4, 8, generator=g) * 5
    up = F.interpolate(low[None], size=(16, 32), mode="bilinear", align_corners=False)[0]
    sem_u = torch.einsum("qc,qhw->chw", p, up.sigmoid())
    save("g1_rba_reduce", mask_cls=np_(mask_cls), mask_pred=np_(mask_pred), sem_seg=np_(sem), rba=np_(rba),
         argmax=np_(sem.argmax(0)).astype(np.int32), low=np_(low), up=np_(up), sem_up=np_(sem_u),
         rba_up=np_(-sem_u.tanh().sum(0)))


def g_ms_deform(R):
    """G2: ms_deform_attn_core_pytorch (ops/functions/ms_deform_attn_func.py:52-72) on (a) the shape/seed set of
    the reference's own ops/test.py:24-39 and (b) a 3-level M=8 D=32 P=4 set with out-of-range samples."""
    core = R.fn.ms_deform_attn_core_pytorch
    out = {}
    torch.manual_seed(3)  # ops/test.py:31
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    for tag in ("a64", "a32"):  # double then float check, in the order test.py calls them
        value = torch.rand(N, S, M, D) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        w = torch.rand(N, Lq, M, L, P) + 1e-5
        w /= w.sum(-1, keepdim=True).sum(-2, keepdim=True)
        if tag == "a64":
            o = core(value.double(), shapes, loc.double(), w.double())
        else:
            o = core(value, shapes, loc, w)
        out.update({f"{tag}_value": np_(value), f"{tag}_loc": np_(loc), f"{tag}_w": np_(w), f"{tag}_out": np_(o)})
    out["a_shapes"] = np_(shapes)
    g = torch.Generator().manual_seed(11)
    N, M, D, L, P = 1, 8, 32, 3, 4
    shapes = torch.as_tensor([(12, 20), (6, 10), (3, 5)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    Lq = S
    value = torch.randn(N, S, M, D, generator=g)
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g) * 1.3 - 0.15   # some samples fall outside [0,1]
    w = F.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P)
    o = core(value, shapes, loc, w)
    o64 = core(value.double(), shapes, loc.double(), w.double())
    out.update(b_value=np_(value), b_loc=np_(loc), b_w=np_(w), b_out=np_(o), b_out64=np_(o64).astype(np.float32), b_shapes=np_(shapes))
    save("g2_ms_deform_attn", **out)


def g_pos_embed(R):
    pe = R.pe.PositionEmbeddingSine(128, normalize=True)
    o = pe(torch.zeros(1, 4, 23, 40))
    pe2 = R.pe.PositionEmbeddingSine(32, normalize=True)
    o2 = pe2(torch.zeros(1, 4, 2, 3))
    save("g3_pos_embed", pe128_23x40=np_(o[0]), pe32_2x3=np_(o2[0]))


def g_swin_parts(R):
    """G3: WindowAttention / SwinTransformerBlock (non multiple-of-window grid, shifted) / PatchMerging (odd H) /
    BasicLayer shift mask / PatchEmbed (H, W not multiples of 4) of backbone/swin.py."""
    S = R.swin
    out = {}
    torch.manual_seed(0)
    wa = S.WindowAttention(32, (6, 6), 2).eval()
    fill_state_dict_(wa, 0, prefix="backbone.layers.0.blocks.0.attn.")
    x = torch.randn(4, 36, 32)
    mask = torch.zeros(4, 36, 36)
    mask[1, :18, 18:] = -100.0
    mask[1, 18:, :18] = -100.0
    mask[3, :6, 6:] = -100.0
    mask[3, 6:, :6] = -100.0
    with torch.no_grad():
        out.update(wa_x=np_(x), wa_mask=np_(mask), wa_out=np_(wa(x)), wa_out_masked=np_(wa(x, mask)))
        for k, v in wa.state_dict().items():
            out["wa_sd." + k] = np_(v)
        layer = S.BasicLayer(dim=32, depth=2, num_heads=2, window_size=6, drop_path=[0.0, 0.0],
                             downsample=S.PatchMerging).eval()
        fill_state_dict_(layer, 0, prefix="backbone.layers.1.")
        H, W = 13, 20   # pads to 18 x 24; odd H for PatchMerging
        xx = torch.randn(1, H * W, 32)
        x_out, _, _, x_down, Wh, Ww = layer(xx, H, W)
        out.update(bl_x=np_(xx), bl_out=np_(x_out), bl_down=np_(x_down), bl_hw=np.array([H, W, Wh, Ww]))
        for k, v in layer.state_dict().items():
            out["bl_sd." + k] = np_(v)
        pe = S.PatchEmbed(4, 3, 32, nn.LayerNorm).eval()
        fill_state_dict_(pe, 0, prefix="backbone.patch_embed.")
        img = torch.randn(1, 3, 30, 45)
        out.update(pe_x=np_(img), pe_out=np_(pe(img)))
        for k, v in pe.state_dict().items():
            out["pe_sd." + k] = np_(v)
    save("g3_swin_parts", **out)


def _ref_model(R, arch, seed, recipe=None):
    a = ARCHS[arch]
    torch.manual_seed(0)
    model = RefModel(R, a).eval()
    meta = dict(n_heads=a["nheads"], n_points=4)
    if recipe is not None:
        meta["recipe"] = recipe
    fill_state_dict_(model, seed, meta=meta)
    return model


def sk_metrics(score, gt):
    """AUROC / AuPRC / FPR95 exactly as support.py:247-303 computes them (flat arrays; labels 1 = OoD, 0 = inlier, rest ignored)."""
    from sklearn.metrics import roc_curve, auc, average_precision_score
    score, gt = np.asarray(score).reshape(-1), np.asarray(gt).reshape(-1)
    ood_out, ind_out = score[gt == 1], score[gt == 0]
    val_out = np.concatenate((ind_out, ood_out))
    val_label = np.concatenate((np.zeros(len(ind_out)), np.ones(len(ood_out))))
    aupr = average_precision_score(val_label, val_out)
    fpr, tpr, thr = roc_curve(val_label, val_out)
    roc_auc = auc(fpr, tpr)
    fpr_best = 0
    for i, j, k in zip(tpr, fpr, thr):
        if i > 0.95:
            fpr_best = j
            break
    return np.array([roc_auc, aupr, fpr_best])


def g_metric_parity(R, arch, h, w, seed, img_seeds, name):
    """BASELINE configs[4]'s deliverable (and its configs[1] twin): AUROC / AuPRC / FPR95 of the REFERENCE's RbA maps on seeded
    images under seeded labels, pooled over the images as evaluate_ood.py:195-235 + support.py:247-303 pool them.  Stored: the
    three statistics for two label sets (independent Bernoulli labels regenerated from their seeds by the test; score-correlated
    labels stored as packed bits), per-image score summaries, nothing else."""
    model = _ref_model(R, arch, seed)
    scores, lab_a, lab_b = [], [], []
    for s in img_seeds:
        o = ref_forward(model, rand_image(h, w, s))
        scores.append(np_(o["rba"]))
        lab_a.append(np_(parity_labels(h, w, 7000 + s)))
        lab_b.append(np_(parity_labels(h, w, 9000 + s, o["rba"])))
        print(f"  {name}: image seed {s} rba mean {o['rba'].mean():.4f}")
    sc = np.stack(scores)
    la, lb = np.stack(lab_a), np.stack(lab_b)
    ma, mb = sk_metrics(sc, la), sk_metrics(sc, lb)
    print(f"  {name}: independent labels auroc/aupr/fpr95 {ma}; score-correlated labels {mb}")
    save(name, arch=np.array(arch), hw=np.array([h, w]), seed=np.array(seed), img_seeds=np.array(img_seeds),
         label_seed_offsets=np.array([7000, 9000]), metrics_indep=ma, metrics_corr=mb,
         labels_corr_bits=np.packbits((lb == 1).reshape(len(img_seeds), -1), axis=1),
         rba_mean=sc.reshape(len(img_seeds), -1).mean(1), n_ood=np.array([(la == 1).sum(), (lb == 1).sum()]))


def g_end_to_end(R, arch, h, w, seed, img_seed, full_outputs, name, npix=4096, recipe=None):
    a = ARCHS[arch]
    model = _ref_model(R, arch, seed, recipe)
    image = rand_image(h, w, img_seed)
    taps = {}
    o = ref_forward(model, image, taps)
    margin = attn_mask_margin(model, taps)
    sem = o["sem_seg"]
    top2 = sem.topk(2, dim=0).values
    gap = (top2[0] - top2[1]).min().item()
    print(f"  {name}: rba [{o['rba'].min():.3f},{o['rba'].max():.3f}] mean {o['rba'].mean():.3f}; "
          f"pred_masks std {o['pred_masks'].std():.2f}; attn-mask margin {margin:.2e}; argmax top-2 gap {gap:.2e}")
    if recipe is not None:
        fe = taps["feats"]
        print(f"  {name} [{recipe}]: pred_masks [{o['pred_masks'].min():.1f},{o['pred_masks'].max():.1f}]; max|res2..res5| "
              + ", ".join(f"{fe[k].abs().max():.0f}" for k in sorted(fe)) + f"; near-tie pixels (<1e-4) {(int(((top2[0] - top2[1]) < 1e-4).sum()))}")
    arrs = dict(arch=np.array(arch), hw=np.array([h, w]), seed=np.array(seed), img_seed=np.array(img_seed),
                recipe=np.array(recipe or "base"), attn_mask_margin=np.array(margin), argmax_gap=np.array(gap),
                pred_logits=np_(o["pred_logits"]))
    if full_outputs:
        arrs.update(image=np_(image), pred_masks=np_(o["pred_masks"]), sem_seg=np_(sem), rba=np_(o["rba"]),
                    argmax=np_(o["argmax"]).astype(np.int32), mask_features=np_(taps["mask_features"][0]))
        for k, v in taps["feats"].items():
            arrs["feat_" + k] = np_(v[0])
        for i, v in enumerate(taps["multi_scale"]):
            arrs[f"multi_scale_{i}"] = np_(v[0])
        for i, aux in enumerate(taps["out"]["aux_outputs"]):
            arrs[f"aux{i}_pred_logits"] = np_(aux["pred_logits"][0])
        for k in ("ood_pred_low", "ood_pred", "densehybrid"):
            if k in o:
                arrs[k] = np_(o[k])
    else:
        g = torch.Generator().manual_seed(123)
        ys = torch.randint(0, h, (npix,), generator=g)
        xs = torch.randint(0, w, (npix,), generator=g)
        pm = o["pred_masks"]
        ys4 = torch.randint(0, pm.shape[1], (npix,), generator=g)
        xs4 = torch.randint(0, pm.shape[2], (npix,), generator=g)
        arrs.update(ys=np_(ys), xs=np_(xs), rba_s=np_(o["rba"][ys, xs]), sem_s=np_(sem[:, ys, xs]),
                    argmax_s=np_(o["argmax"][ys, xs]).astype(np.int32),
                    ys4=np_(ys4), xs4=np_(xs4), pred_masks_s=np_(pm[:, ys4, xs4]),
                    rba_stats=np.array([o["rba"].double().sum().item(), o["rba"].min().item(), o["rba"].max().item()]),
                    argmax_hist=np.bincount(np_(o["argmax"]).ravel(), minlength=a["num_classes"]))
        # whole-map coverage at a fixture size that stays small: a strided sub-grid of exact values, row / column sums and
        # 16x16 block sums of the score map (float64), the complete argmax map (uint8, compresses to a few KB) and the flat
        # indices of every pixel whose top-2 classes are closer than 1e-4 (the only places a flip can be forgiven)
        rba = o["rba"].double()
        gy = torch.arange(64) * (h // 64) + (h // 128)
        gx = torch.arange(128) * (w // 128) + (w // 256)
        full_gap = top2[0] - top2[1]
        hb, wb = (h // 16) * 16, (w // 16) * 16
        arrs.update(gy=np_(gy), gx=np_(gx), grid_rba=np_(o["rba"][gy][:, gx]), grid_sem=np_(sem[:, gy][:, :, gx]),
                    row_sum=np_(rba.sum(1)), col_sum=np_(rba.sum(0)),
                    blk_sum=np_(rba[:hb, :wb].reshape(hb // 16, 16, wb // 16, 16).sum((1, 3))),
                    argmax_full=np_(o["argmax"]).astype(np.uint8),
                    neartie_idx=np_(torch.nonzero(full_gap.flatten() < 1e-4).flatten()).astype(np.int64))
    # state-dict contract: key -> shape
    sd = model.state_dict()
    arrs["sd_keys"] = np.array(sorted(sd.keys()))
    arrs["sd_shapes"] = np.array([",".join(map(str, sd[k].shape)) for k in sorted(sd.keys())])
    save(name, **arrs)


def g_stage_samples(R, arch, h, w, seed, img_seed, name, recipe=None):
    """Round 5 (error attribution): strided samples of the reference's INTERMEDIATES of a full-size forward -- res2..res5, mask_features, pred_masks, class
    logits, sem_seg, rba -- so that a product-side error can be assigned to a stage (tests/test_model_gpu.py::test_error_attribution_by_stage).  A few
    hundred KB: every channel at a 4 x 8 grid of positions for the backbone maps, an 8 x 8 grid for the mask features, 512 random positions of the mask logits,
    4096 random pixels of the outputs."""
    model = _ref_model(R, arch, seed, recipe)
    image = rand_image(h, w, img_seed)
    taps = {}
    o = ref_forward(model, image, taps)
    arrs = dict(arch=np.array(arch), hw=np.array([h, w]), seed=np.array(seed), img_seed=np.array(img_seed), recipe=np.array(recipe or "base"),
                pred_logits=np_(o["pred_logits"]))
    for k, v in taps["feats"].items():
        fh, fw = v.shape[-2:]
        fy = torch.arange(4) * (fh // 4) + fh // 8
        fx = torch.arange(8) * (fw // 8) + fw // 16
        arrs.update({f"fy_{k}": np_(fy), f"fx_{k}": np_(fx), f"feat_{k}_s": np_(v[0][:, fy][:, :, fx])})
    mf = taps["mask_features"][0]
    my = torch.arange(8) * (mf.shape[1] // 8) + mf.shape[1] // 16
    mx = torch.arange(8) * (mf.shape[2] // 8) + mf.shape[2] // 16
    arrs.update(my=np_(my), mx=np_(mx), mask_features_s=np_(mf[:, my][:, :, mx]))
    g = torch.Generator().manual_seed(321)
    pm = o["pred_masks"]
    ys4 = torch.randint(0, pm.shape[1], (512,), generator=g)
    xs4 = torch.randint(0, pm.shape[2], (512,), generator=g)
    ys = torch.randint(0, h, (4096,), generator=g)
    xs = torch.randint(0, w, (4096,), generator=g)
    arrs.update(ys4=np_(ys4), xs4=np_(xs4), pred_masks_s=np_(pm[:, ys4, xs4]), ys=np_(ys), xs=np_(xs), rba_s=np_(o["rba"][ys, xs]),
                sem_s=np_(o["sem_seg"][:, ys[:512], xs[:512]]))
    print(f"  {name}: " + ", ".join(f"{k} {tuple(v.shape)}" for k, v in arrs.items() if hasattr(v, "shape") and v.ndim > 1))
    save(name, **arrs)


def g_metrics(R):
    """G6: AUROC / AuPRC / FPR95 exactly as support.py:247-303 computes them (sklearn roc_curve with its
    default drop_intermediate=True, auc, average_precision_score; FPR at the first tpr > 0.95)."""
    from sklearn.metrics import roc_curve, auc, average_precision_score

    def ref_metrics(score, gt):
        score, gt = score.squeeze(), gt.squeeze()
        ood_out, ind_out = score[gt == 1], score[gt == 0]
        val_out = np.concatenate((ind_out, ood_out))
        val_label = np.concatenate((np.zeros(len(ind_out)), np.ones(len(ood_out))))
        aupr = average_precision_score(val_label, val_out)
        fpr, tpr, thr = roc_curve(val_label, val_out)
        roc_auc = auc(fpr, tpr)
        fpr_best = 0
        for i, j, k in zip(tpr, fpr, thr):
            if i > 0.95:
                fpr_best = j
                break
        return np.array([roc_auc, aupr, fpr_best])

    out = {}
    rng = np.random.RandomState(5)
    # case a: continuous scores, 3 "images" of 40x50, labels {0,1,255}
    gt = rng.choice([0, 1, 255], size=(3, 1, 40, 50), p=[0.85, 0.05, 0.10])
    score = (rng.randn(3, 40, 50) + 1.5 * (gt[:, 0] == 1)).astype(np.float32)
    out.update(a_score=score, a_gt=gt, a_metrics=ref_metrics(score, gt))
    # case b: heavy ties (quantised scores)
    gt = rng.choice([0, 1, 255], size=(2, 1, 30, 30), p=[0.7, 0.2, 0.1])
    score = np.round(rng.randn(2, 30, 30) + 1.0 * (gt[:, 0] == 1), 1).astype(np.float32)
    out.update(b_score=score, b_gt=gt, b_metrics=ref_metrics(score, gt))
    # case c: perfectly separable; case d: rba-like range (-19, 0]
    gt = rng.choice([0, 1], size=(1, 1, 20, 20), p=[0.9, 0.1])
    score = (gt[:, 0] * 2.0 + rng.rand(1, 20, 20)).astype(np.float32)
    out.update(c_score=score, c_gt=gt, c_metrics=ref_metrics(score, gt))
    gt = rng.choice([0, 1, 255], size=(4, 1, 64, 64), p=[0.9, 0.03, 0.07])
    score = (-19 * rng.beta(5, 1, size=(4, 64, 64)) + 6 * (gt[:, 0] == 1) * rng.rand(4, 64, 64)).astype(np.float32)
    out.update(d_score=score, d_gt=gt, d_metrics=ref_metrics(score, gt))
    save("g6_metrics", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--full", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(8)
    R = _ref_modules()
    jobs = {
        "g1": lambda: g_rba_reduce(R),
        "g2": lambda: g_ms_deform(R),
        "g3pe": lambda: g_pos_embed(R),
        "g3swin": lambda: g_swin_parts(R),
        "g4tiny1": lambda: g_end_to_end(R, "tiny1", 60, 90, 0, 1234, True, "g4_tiny1_60x90"),
        "g4tiny3": lambda: g_end_to_end(R, "tiny3", 60, 90, 0, 1234, True, "g4_tiny3_60x90"),
        "g4tiny1dh": lambda: g_end_to_end(R, "tiny1_dh", 60, 90, 0, 1234, True, "g4_tiny1_dh_60x90"),
        "g6": lambda: g_metrics(R),
    }
    if args.full:
        jobs["g5c2"] = lambda: g_end_to_end(R, "swin_b_1dl", 1024, 2048, 0, 1234, False, "g5_swin_b_1dl_1024x2048")
        jobs["g5c5"] = lambda: g_end_to_end(R, "swin_b_9dl", 720, 1280, 0, 1234, False, "g5_swin_b_9dl_720x1280")
        # BASELINE C4's architecture (Swin-L, channel counts that are not multiples of 128) at a size the CPU reference finishes quickly
        jobs["g5c4"] = lambda: g_end_to_end(R, "swin_l_1dl", 512, 1024, 0, 1234, False, "g5_swin_l_1dl_512x1024")
        # ... and at C4's real size
        jobs["g5c4full"] = lambda: g_end_to_end(R, "swin_l_1dl", 1024, 2048, 0, 1234, False, "g5_swin_l_1dl_1024x2048")
        # trained-like dynamic range (seeded_weights "heavy" recipe): gammas 0.1..10, residual outlier channels, mask logits to +-40
        jobs["g5heavy"] = lambda: g_end_to_end(R, "swin_b_1dl", 512, 1024, 0, 1234, False, "g5_swin_b_1dl_heavy_512x1024", recipe="heavy")
        # metric parity (BASELINE configs[4] "AuPRC/FPR95 parity check" and its configs[1] twin): 4 images each
        jobs["g7c5"] = lambda: g_metric_parity(R, "swin_b_9dl", 720, 1280, 0, [1234, 1235, 1236, 1237], "g7_metrics_swin_b_9dl_720x1280")
        jobs["g7c2"] = lambda: g_metric_parity(R, "swin_b_1dl", 1024, 2048, 0, [1234, 1235, 1236, 1237], "g7_metrics_swin_b_1dl_1024x2048")
        # round 5: strided samples of the reference's intermediates (error attribution by stage) for C2 and the heavy recipe
        jobs["g8c2"] = lambda: g_stage_samples(R, "swin_b_1dl", 1024, 2048, 0, 1234, "g8_stages_swin_b_1dl_1024x2048")
        jobs["g8heavy"] = lambda: g_stage_samples(R, "swin_b_1dl", 512, 1024, 0, 1234, "g8_stages_swin_b_1dl_heavy_512x1024", recipe="heavy")
    for k, fn in jobs.items():
        if args.only and k != args.only:
            continue
        print(k)
        fn()


if __name__ == "__main__":
    main()
