"""CPU: the oracle (oracle/) against the fixtures the reference's own modules produced
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import ref_metrics, ref_model, ref_ops
from rba_amd import arch as A

T = torch.from_numpy


def close(a, b, atol, rtol=0.0):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    err = np.abs(a - b).max()
    assert err <= atol + rtol * np.abs(b).max(), f"max abs err {err:.3e} > {atol:.1e}"


def test_k1_rba_reduce(golden):
    g = golden("g1_rba_reduce")
    sem, rba, am = ref_ops.rba_reduce(T(g["mask_pred"]), T(g["mask_cls"]))
    close(sem, g["sem_seg"], 1e-6)
    close(rba, g["rba"], 1e-6)
    assert np.array_equal(am.numpy(), g["argmax"])
    # ordered-accumulation variant (what the HIP kernel does) differs by re-association only
    sem2, rba2, am2 = ref_ops.rba_reduce_ordered(T(g["mask_pred"]), ref_ops.class_probs(T(g["mask_cls"])))
    close(sem2, g["sem_seg"], 5e-6)
    close(rba2, g["rba"], 5e-6)
    gap = np.sort(g["sem_seg"], axis=0)
    flips = am2.numpy() != g["argmax"]
    assert not (flips & ((gap[-1] - gap[-2]) > 1e-5)).any()
    # x4 bilinear upsample in front (maskformer_model.py:294-299)
    up = ref_ops.upsample_bilinear(T(g["low"])[None], (16, 32))[0]
    close(up, g["up"], 1e-6)
    close(ref_ops.rba_reduce(up, T(g["mask_cls"]))[1], g["rba_up"], 1e-6)


def test_k2_ms_deform_attn_reference_test_set(golden):
    """shape / seed set of the reference's own ops/test.py:24-39."""
    g = golden("g2_ms_deform_attn")
    shapes = T(g["a_shapes"])
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    o64 = ref_ops.ms_deform_attn(T(g["a64_value"]).double(), shapes, T(g["a64_loc"]).double(), T(g["a64_w"]).double())
    close(o64, g["a64_out"], 1e-12)
    o32 = ref_ops.ms_deform_attn(T(g["a32_value"]), shapes, T(g["a32_loc"]), T(g["a32_w"]))
    close(o32, g["a32_out"], 1e-8)
    # scalar restatement of the CUDA kernel arithmetic agrees with the grid_sample formulation
    loops = ref_ops.ms_deform_attn_loops(T(g["a64_value"]).double(), shapes, lsi, T(g["a64_loc"]).double(),
                                         T(g["a64_w"]).double())
    close(loops, g["a64_out"], 1e-12)


def test_k2_ms_deform_attn_3level(golden):
    g = golden("g2_ms_deform_attn")
    shapes = T(g["b_shapes"])
    o = ref_ops.ms_deform_attn(T(g["b_value"]), shapes, T(g["b_loc"]), T(g["b_w"]))
    close(o, g["b_out"], 1e-6)
    close(o, g["b_out64"], 2e-5)
    # loop restatement on a slice of queries (python loops are slow)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    sl = slice(0, 12)
    loops = ref_ops.ms_deform_attn_loops(T(g["b_value"]).double(), shapes, lsi, T(g["b_loc"][:, sl]).double(),
                                         T(g["b_w"][:, sl]).double())
    close(loops, g["b_out64"][:, sl], 1e-5)


def test_position_embedding(golden):
    g = golden("g3_pos_embed")
    close(ref_ops.position_embedding_sine(23, 40, 128), g["pe128_23x40"], 1e-6)
    close(ref_ops.position_embedding_sine(2, 3, 32), g["pe32_2x3"], 1e-6)


def test_swin_parts(golden):
    g = golden("g3_swin_parts")
    sd = {k[len("wa_sd."):]: T(g[k]) for k in g.files if k.startswith("wa_sd.")}
    args = (sd["qkv.weight"], sd["qkv.bias"], sd["proj.weight"], sd["proj.bias"],
            sd["relative_position_bias_table"], 6, 2)
    assert np.array_equal(ref_ops.relative_position_index(6).numpy(), g["wa_sd.relative_position_index"])
    close(ref_ops.window_attention(T(g["wa_x"]), *args), g["wa_out"], 2e-6)
    close(ref_ops.window_attention(T(g["wa_x"]), *args, mask=T(g["wa_mask"])), g["wa_out_masked"], 2e-6)
    # BasicLayer on a 13 x 20 grid: window pad, shifted block, odd-H PatchMerging
    sd = {"L." + k[len("bl_sd."):]: T(g[k]) for k in g.files if k.startswith("bl_sd.")}
    H, W, Wh, Ww = g["bl_hw"]
    x = T(g["bl_x"])
    mask = ref_ops.shift_attn_mask(H, W, 6, 3)
    for b in range(2):
        x = ref_model.swin_block(x, H, W, sd, f"L.blocks.{b}", 6, 0 if b == 0 else 3, 2, mask)
    close(x, g["bl_out"], 5e-6)
    close(ref_model.patch_merging(x, H, W, sd, "L.downsample"), g["bl_down"], 5e-6)
    assert ((H + 1) // 2, (W + 1) // 2) == (Wh, Ww)


@pytest.mark.parametrize("name", ["g4_tiny1_60x90", "g4_tiny3_60x90", "g4_tiny1_dh_60x90"])
def test_end_to_end_tiny(golden, name):
    g = golden(name)
    a = A.complete(A.ARCHS[str(g["arch"])])
    shapes = A.state_dict_shapes(a)
    # the checkpoint contract: same keys, same shapes as the reference module tree
    assert sorted(shapes) == list(g["sd_keys"])
    assert [",".join(map(str, shapes[k][0])) for k in sorted(shapes)] == list(g["sd_shapes"])
    sd = A.seeded_weights(a, int(g["seed"]))
    taps = {}
    o = ref_model.forward(T(g["image"]), sd, a, taps)
    for k in ("res2", "res3", "res4", "res5"):
        close(taps["feats"][k][0], g["feat_" + k], 2e-5)
    close(taps["mask_features"][0], g["mask_features"], 2e-5)
    for i, v in enumerate(taps["multi_scale"]):
        close(v[0], g[f"multi_scale_{i}"], 2e-5)
    close(o["pred_logits"], g["pred_logits"], 2e-5)
    close(o["pred_masks"], g["pred_masks"], 1e-4)
    close(o["sem_seg"], g["sem_seg"], 1e-5)
    close(o["rba"], g["rba"], 1e-5)
    gap = np.sort(g["sem_seg"], axis=0)
    flips = o["argmax"].numpy() != g["argmax"]
    assert not (flips & ((gap[-1] - gap[-2]) > 1e-5)).any()
    if "densehybrid" in g.files:                       # DenseHybrid head + score (evaluate_ood.py:161-173) of the reference decoder
        close(ref_model.ood_pred_head(taps["mask_features"], sd)[0], g["ood_pred_low"], 2e-5)
        close(o["ood_pred"][0], g["ood_pred"], 2e-5)
        close(o["densehybrid"][0], g["densehybrid"], 2e-5)


@pytest.mark.parametrize("case", "abcd")
def test_metrics(golden, case):
    g = golden("g6_metrics")
    r = ref_metrics.evaluate_ood(g[case + "_score"], g[case + "_gt"])
    np.testing.assert_allclose([r["auroc"], r["aupr"], r["fpr95"]], g[case + "_metrics"], rtol=0, atol=1e-12)


def test_c_oracle_k1_k2(golden):
    """the plain-C restatement (oracle/c/rba_oracle.c) against the same fixtures"""
    from oracle import c_oracle
    g = golden("g1_rba_reduce")
    prob = ref_ops.class_probs(T(g["mask_cls"])).numpy()
    sem, rba, arg = c_oracle.rba_reduce(g["mask_pred"], prob)
    close(sem, g["sem_seg"], 5e-6)
    close(rba, g["rba"], 5e-6)
    gap = np.sort(g["sem_seg"], axis=0)
    assert not ((arg != g["argmax"]) & ((gap[-1] - gap[-2]) > 1e-5)).any()
    g = golden("g2_ms_deform_attn")
    for tag in ("a32", "b"):
        sh = g["a_shapes"] if tag == "a32" else g["b_shapes"]
        lsi = np.concatenate(([0], np.cumsum(sh.prod(1))[:-1]))
        out = c_oracle.ms_deform_attn(g[tag + "_value"], sh, lsi, g[tag + "_loc"], g[tag + "_w"])
        close(out, g[tag + "_out"], 5e-6)


@pytest.mark.parametrize("fixture,arch_name", [("g5_swin_b_1dl_1024x2048", "swin_b_1dl"),
                                               ("g5_swin_b_9dl_720x1280", "swin_b_9dl"),
                                               ("g5_swin_l_1dl_512x1024", "swin_l_1dl"),
                                               ("g5_swin_b_1dl_heavy_512x1024", "swin_b_1dl")])
def test_end_to_end_full_size(golden, fixture, arch_name):
    """BASELINE configs C2 / C5 at full size: oracle vs sampled outputs of the reference's own modules (about 10 s each)."""
    g = golden(fixture)
    a = A.complete(A.ARCHS[arch_name])
    recipe = str(g["recipe"]) if "recipe" in g else "base"            # "heavy": the trained-like dynamic-range stress recipe
    sd = A.seeded_weights(a, int(g["seed"]), recipe=None if recipe == "base" else recipe)
    h, w = (int(v) for v in g["hw"])
    gen = torch.Generator().manual_seed(int(g["img_seed"]))
    image = torch.randint(0, 256, (3, h, w), generator=gen, dtype=torch.uint8)
    o = ref_model.forward(image, sd, a)
    ys, xs, ys4, xs4 = (T(g[k]).long() for k in ("ys", "xs", "ys4", "xs4"))
    close(o["pred_logits"], g["pred_logits"], 2e-5)
    close(o["pred_masks"][:, ys4, xs4], g["pred_masks_s"], 2e-4)
    close(o["rba"][ys, xs], g["rba_s"], 2e-5)
    close(o["sem_seg"][:, ys, xs], g["sem_s"], 2e-5)
    top2 = np.sort(g["sem_s"], axis=0)
    flips = o["argmax"][ys, xs].numpy() != g["argmax_s"]
    assert not (flips & ((top2[-1] - top2[-2]) > 1e-4)).any()
    assert abs(o["rba"].double().sum().item() - g["rba_stats"][0]) < 1e-5 * h * w


@pytest.mark.parametrize("arch", ["r50_1dl", "swin_b_1dl"])
def test_c1_plumbing_cpu_forward_256x512(arch):
    """BASELINE config C1 ("ResNet-50 Mask2Former, 1 decoder layer, 100 queries, 1x256x512 random tensor, CPU reference forward +
    RbA score"): the CPU path end to end on the ResNet-50 architecture (Detectron2's backbone restated, parity unpinned) and, for
    comparison, on Swin-B 1dl."""
    a = A.complete(A.ARCHS[arch])
    sd = A.seeded_weights(a, 0)
    x = torch.randn(3, 256, 512, generator=torch.Generator().manual_seed(0))
    taps = {}
    o = ref_model.forward(x, sd, a, taps)
    if arch == "r50_1dl":
        assert [tuple(taps["feats"][f].shape[1:]) for f in ("res2", "res3", "res4", "res5")] == [(256, 64, 128), (512, 32, 64),
                                                                                                 (1024, 16, 32), (2048, 8, 16)]
    assert o["pred_logits"].shape == (100, 20) and o["pred_masks"].shape == (100, 64, 128)
    assert o["sem_seg"].shape == (19, 256, 512) and o["rba"].shape == (256, 512) and o["argmax"].shape == (256, 512)
    assert torch.isfinite(o["rba"]).all() and float(o["rba"].max()) <= 0.0 and float(o["rba"].min()) > -19.0
    assert float(o["sem_seg"].min()) >= 0.0


def test_resnet50_matches_torch_module_graph():
    """The oracle's functional ResNet-50 against an independent nn.Module graph of the same Detectron2 definition (the product's
    module tree on CPU with plain BatchNorm/conv calls): stride placement (STRIDE_IN_1X1 False -> the 3x3 conv strides), projection
    shortcuts only where the shape changes, max-pool after the stem."""
    import torch.nn.functional as F
    from rba_amd.modeling.backbone.resnet import ResNet
    a = A.complete(A.ARCHS["r50_1dl"])
    sd = A.seeded_weights(a, 1)
    net = ResNet(a).eval()
    bsd = {k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}
    missing = net.load_state_dict(bsd, strict=False)
    assert not missing.unexpected_keys and all(k.endswith("num_batches_tracked") for k in missing.missing_keys)
    x = torch.randn(1, 3, 96, 160, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        got = net(x)
    want = ref_model.resnet_backbone(x, sd, a)
    for f in want:
        assert got[f].shape == want[f].shape
        assert (got[f] - want[f]).abs().max() < 2e-4 * max(1.0, float(want[f].abs().max())), f


def test_gaussian_blur_oracle_vs_scipy():
    """The oracle's restatement of torchvision's GaussianBlur(7, sigma=1) (torchvision is not installed: parity unpinned) against
    an independent implementation of the same definition: scipy's gaussian_filter with mirror boundaries, radius 3."""
    from scipy import ndimage
    from oracle import ref_ops
    g = torch.Generator().manual_seed(4)
    for H, W in ((40, 64), (7, 9), (128, 33)):
        x = torch.randn(H, W, generator=g, dtype=torch.float64)
        want = ndimage.gaussian_filter(x.numpy(), sigma=1.0, mode="mirror", truncate=3.0)
        got = ref_ops.gaussian_blur(x, 7, 1.0).numpy()
        assert got.shape == want.shape and np.abs(got - want).max() < 1e-12
    const = ref_ops.gaussian_blur(torch.full((16, 16), 2.5), 7, 1.0)
    assert (const - 2.5).abs().max() < 1e-6                                                  # weights sum to one


def test_ood_components_oracle_vs_scipy():
    """Oracle restatement of the cv2 morphology + connected-components step (OpenCV is not installed: parity unpinned) against
    scipy.ndimage's binary_opening / binary_closing / label on random blob maps, plus a hand-checked case."""
    from scipy import ndimage
    from oracle import ref_ops
    box = np.ones((3, 3), bool)
    g = np.random.default_rng(7)
    for H, W, p in ((40, 60, 0.55), (17, 9, 0.7), (64, 64, 0.35)):
        score = ndimage.gaussian_filter(g.standard_normal((H, W)), 1.5) * 4
        thr = float(np.quantile(score, 1 - p))
        labels, n = ref_ops.ood_components(score, thr)
        b = score > thr
        # scipy: erosion with border_value=1 / dilation with border_value=0 == "out-of-image neighbours are ignored"
        opened = ndimage.binary_dilation(ndimage.binary_erosion(b, box, border_value=1), box, border_value=0)
        closed = ndimage.binary_erosion(ndimage.binary_dilation(opened, box, border_value=0), box, border_value=1)
        want, wn = ndimage.label(closed)                                   # default structure = 4-connectivity, raster order
        assert n == wn and np.array_equal(labels, want)
    m = np.zeros((7, 11))
    m[0:4, 0:4] = 1; m[0:3, 7:10] = 1; m[5, 5] = 1                         # two blobs survive (gap of 3 columns: the closing does
    labels, n = ref_ops.ood_components(m, 0.5)                             # not bridge it), the single pixel is opened away
    assert n == 2 and labels[1, 1] == 1 and labels[1, 8] == 2 and labels[5, 5] == 0
    m[0:3, 6] = 1                                                          # gap of 2 columns: the closing merges the blobs
    assert ref_ops.ood_components(m, 0.5)[1] == 1
