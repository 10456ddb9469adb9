"""GPU: the whole product path (Swin -> pixel decoder -> masked decoder -> RbA reduction, HIP kernels K1-K5) against
the golden fixtures of the reference and against the oracle run on the same seeded weights."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_model
from rba_amd import arch as A

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def build(name, seed=0, recipe=None):
    from rba_amd.checkpoint import load_checkpoint
    from rba_amd.maskformer_model import MaskFormer
    a = A.complete(A.ARCHS[name])
    sd = A.seeded_weights(a, seed, recipe=recipe)
    model = load_checkpoint(MaskFormer(a), sd).cuda().eval()
    assert model.graph_replay == "auto"       # the product default (round 6): replay only where measured to be launch-bound (test_graph_replay_policy_is_measured)
    model.graph_replay = True                 # the tests below pin the capture-at-the-third-call behaviour of the explicit mode
    return model, a, sd


def maxerr(a, b):
    return (a.detach().cpu().double() - torch.as_tensor(b).double()).abs().max().item()


def argmax_bad(arg, sem_ref, tol=1e-4):
    top2 = sem_ref.topk(2, dim=0).values
    flips = arg.cpu().long() != sem_ref.argmax(0)
    return int((flips & ((top2[0] - top2[1]) > tol)).sum()), int(flips.sum())


@pytest.mark.parametrize("name", ["g4_tiny1_60x90", "g4_tiny3_60x90"])
def test_tiny_vs_golden(golden, name):
    g = golden(name)
    model, a, _ = build(str(g["arch"]), int(g["seed"]))
    image = T(g["image"])
    feats = model.backbone(model.preprocess([{"image": image}])[0])
    for k in ("res2", "res3", "res4", "res5"):
        assert maxerr(feats[k][0], g["feat_" + k]) < 5e-5, k
    mask_cls, mask_pred, sizes, padded = model.predict([{"image": image}])
    assert sizes == [(60, 90)] and padded == (64, 96)
    assert maxerr(mask_cls[0], g["pred_logits"]) < 1e-4
    assert maxerr(mask_pred[0], g["pred_masks"]) < 5e-4
    for fused in (True, False):
        model.fused_upsample = fused
        out = model([{"image": image}], return_argmax=True)[0]
        assert out["sem_seg"].shape == (19, 60, 90)
        assert maxerr(out["sem_seg"], g["sem_seg"]) < 1e-4          # north star: scores within 1e-4
        assert maxerr(out["rba"], g["rba"]) < 1e-4
        bad, flips = argmax_bad(out["argmax"], T(g["sem_seg"]))
        assert bad == 0, (bad, flips)
        rba, arg = model.rba_scores([{"image": image}], return_argmax=True)[0]
        assert torch.equal(rba, out["rba"]) and torch.equal(arg, out["argmax"])


def test_dense_hybrid_tiny(golden):
    """DenseHybrid: `ood_pred` head (BNReluConv on the mask features), MaskFormer.forward(return_ood_pred=True) and
    get_densehybrid_score against the reference decoder built with ood_prediction=True (evaluate_ood.py:161-173)."""
    from rba_amd import evaluate_ood as E
    g = golden("g4_tiny1_dh_60x90")
    model, a, _ = build("tiny1_dh", int(g["seed"]))
    image = T(g["image"])
    out, ood = model([{"image": image}], return_ood_pred=True)
    assert ood.shape == (1, 2, 60, 90)
    assert maxerr(ood[0], g["ood_pred"]) < 5e-5
    assert maxerr(out[0]["sem_seg"], g["sem_seg"]) < 1e-4
    score = E.get_densehybrid_score(model, image[None])
    assert score.shape == (1, 60, 90) and maxerr(score[0], g["densehybrid"]) < 1e-4
    model2, _, _ = build("tiny1", int(g["seed"]))
    with pytest.raises(KeyError):
        model2([{"image": image}], return_ood_pred=True)


def test_evaluator_surface_tiny(golden, tmp_path):
    """get_model(config.yaml, model_final.pth) -> get_RbA / get_logits / OODEvaluator, as evaluate_ood.py drives them."""
    import yaml
    from rba_amd import evaluate_ood as E
    from rba_amd.support import OODEvaluator
    g = golden("g4_tiny1_60x90")
    a = A.complete(A.ARCHS["tiny1"])
    cfg = {"MODEL": {"META_ARCHITECTURE": "MaskFormer", "BACKBONE": {"NAME": "D2SwinTransformer"},
                     "SWIN": {"EMBED_DIM": 32, "DEPTHS": [2, 2, 2, 2], "NUM_HEADS": [1, 2, 4, 8], "WINDOW_SIZE": 6},
                     "SEM_SEG_HEAD": {"CONVS_DIM": 64, "MASK_DIM": 64, "TRANSFORMER_ENC_LAYERS": 2,
                                      "DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES": ["res5"]},
                     "MASK_FORMER": {"HIDDEN_DIM": 64, "NHEADS": 2, "NUM_OBJECT_QUERIES": 16, "DIM_FEEDFORWARD": 128,
                                     "DEC_LAYERS": 2}}}
    (tmp_path / "config.yaml").write_text(yaml.safe_dump(cfg))
    sd = A.seeded_weights(a, 0)
    sd["criterion.empty_weight"] = torch.ones(20)          # present in real checkpoints, must be tolerated
    torch.save({"model": sd, "iteration": 1}, tmp_path / "model_final.pth")
    model = E.get_model(str(tmp_path / "config.yaml"), str(tmp_path / "model_final.pth"))
    x = T(g["image"])[None]
    assert maxerr(E.get_RbA(model, x), g["rba"]) < 1e-4
    logits = E.get_logits(model, x)
    assert logits.shape == (1, 19, 60, 90) and maxerr(logits[0], g["sem_seg"]) < 1e-4
    # the reference's own get_RbA formula applied to our model's output dict
    out = model([{"image": x[0]}])
    assert maxerr(-out[0]["sem_seg"].tanh().sum(dim=0), g["rba"]) < 1e-4
    ev = OODEvaluator(model, E.get_logits, E.get_RbA)
    gt = torch.zeros(1, 60, 90, dtype=torch.long)
    gt[:, 10:20, 10:30] = 1
    gt[:, :2] = 255
    scores, gts, preds = ev.compute_anomaly_scores([(x, gt)], device=torch.device("cuda"), return_preds=True)
    assert scores.shape == (1, 60, 90) and gts.shape == (1, 1, 60, 90) and preds.shape == (1, 1, 60, 90)
    r = ev.evaluate_ood(scores, gts, verbose=False)
    from oracle import ref_metrics
    want = ref_metrics.evaluate_ood(scores, gts)
    assert all(abs(r[k] - want[k]) < 1e-9 for k in want)
    # optional smoothing of the score map (support.py:366-383): GaussianBlur(7, sigma=1) of the same scores
    from oracle import ref_ops
    smooth, _ = ev.compute_anomaly_scores([(x, gt)], device=torch.device("cuda"), use_gaussian_smoothing=True)
    assert smooth.shape == (1, 60, 90)
    assert maxerr(torch.as_tensor(smooth[0]), ref_ops.gaussian_blur(torch.as_tensor(scores[0]).double(), 7, 1.0)) < 1e-5


def _full_size(golden, fixture, arch_name, tol_rba):
    """Whole-map parity at a BASELINE size on BOTH K1 paths (fused x4 up-sample = the product default; materialised planes =
    the path bench.py times): sampled pixels, a strided 64x128 sub-grid, row / column / 16x16-block sums of the score map and
    the COMPLETE argmax map -- a flip is forgiven only at a pixel whose two best classes are closer than 1e-4 in the
    reference's own output (their indices are in the fixture), and the count of flips is asserted, not printed."""
    g = golden(fixture)
    recipe = str(g["recipe"]) if "recipe" in g else "base"
    model, a, sd = build(arch_name, int(g["seed"]), None if recipe == "base" else recipe)
    h, w = (int(v) for v in g["hw"])
    gen = torch.Generator().manual_seed(int(g["img_seed"]))
    image = torch.randint(0, 256, (3, h, w), generator=gen, dtype=torch.uint8)
    mask_cls, mask_pred, _, _ = model.predict([{"image": image}])
    ys, xs, ys4, xs4 = (T(g[k]).long() for k in ("ys", "xs", "ys4", "xs4"))
    gy, gx = T(g["gy"]).long(), T(g["gx"]).long()
    e_logits = maxerr(mask_cls[0], g["pred_logits"])
    e_masks = maxerr(mask_pred[0].cpu()[:, ys4, xs4], g["pred_masks_s"])
    # mask logits: 5e-4 at the base recipe's range (|x| <~ 10); the heavy recipe's reach +-34, the bound scales with the range
    tol_masks = 5e-4 * max(1.0, float(np.abs(g["pred_masks_s"]).max()) / 10.0)
    assert e_logits < 1e-4 and e_masks < tol_masks, (e_logits, e_masks, tol_masks)
    ref_arg = T(g["argmax_full"].astype(np.int64))
    tie = torch.zeros(h * w, dtype=torch.bool)
    tie[T(g["neartie_idx"])] = True
    tie = tie.view(h, w)
    hb, wb = h // 16 * 16, w // 16 * 16
    for fused in (True, False):
        model.fused_upsample = fused
        rba, arg = model.rba_scores([{"image": image}], return_argmax=True)[0]
        rba, arg = rba.cpu(), arg.cpu().long()
        assert rba.shape == (h, w) and arg.shape == (h, w)
        e_rba = max(maxerr(rba[ys, xs], g["rba_s"]), maxerr(rba[gy][:, gx], g["grid_rba"]))
        d = rba.double()
        e_row = (d.sum(1) - T(g["row_sum"])).abs().max().item() / w             # mean error per pixel of the worst row
        e_col = (d.sum(0) - T(g["col_sum"])).abs().max().item() / h
        e_blk = (d[:hb, :wb].reshape(hb // 16, 16, wb // 16, 16).sum((1, 3)) - T(g["blk_sum"])).abs().max().item() / 256
        flips = arg != ref_arg
        bad = int((flips & ~tie).sum())
        n_flips = int(flips.sum())
        hist = torch.bincount(arg.flatten(), minlength=a["num_classes"])
        d_hist = int((hist - T(g["argmax_hist"])).abs().sum())
        print(f"{fixture} fused_upsample={fused}: |d logits| {e_logits:.2e} |d masks| {e_masks:.2e} |d rba| {e_rba:.2e} "
              f"row/col/block mean err {e_row:.1e}/{e_col:.1e}/{e_blk:.1e}; argmax flips {n_flips} of {h * w} "
              f"(outside the {int(tie.sum())} near-tie pixels: {bad}); attn-mask margin of the reference run "
              f"{float(g['attn_mask_margin']):.1e}")
        assert e_rba < tol_rba
        assert max(e_row, e_col) < 2e-5 and e_blk < 5e-5, (e_row, e_col, e_blk)
        assert bad == 0 and n_flips <= int(tie.sum())
        assert d_hist <= 2 * n_flips
        # sem_seg on the same sub-grid through the evaluator-facing forward (materialises [19, H, W])
        if fused:
            out = model([{"image": image}], return_argmax=True)[0]
            assert maxerr(out["sem_seg"].cpu()[:, gy][:, :, gx], g["grid_sem"]) < 1e-4
            assert torch.equal(out["rba"].cpu(), rba) and torch.equal(out["argmax"].cpu().long(), arg)
            del out


def test_full_size_swin_b_1dl_1024x2048(golden):
    """BASELINE config C2 against values the reference's own modules produced (sampled pixels)."""
    _full_size(golden, "g5_swin_b_1dl_1024x2048", "swin_b_1dl", 1e-4)


def test_swin_l_1dl_512x1024(golden):
    """BASELINE config C4's architecture (Swin-L: channel counts 192..1536, not multiples of the GEMM tile) against the reference's
    own modules at 512x1024."""
    _full_size(golden, "g5_swin_l_1dl_512x1024", "swin_l_1dl", 1e-4)


def test_full_size_swin_l_1dl_1024x2048(golden):
    """BASELINE config C4 at its real size: Swin-L, 1 decoder layer, 1024x2048."""
    _full_size(golden, "g5_swin_l_1dl_1024x2048", "swin_l_1dl", 1e-4)


def test_full_size_swin_b_9dl_720x1280(golden):
    """BASELINE config C5 (9 decoder layers, 3-level MSDeformAttn, 720 -> 736 padding)."""
    _full_size(golden, "g5_swin_b_9dl_720x1280", "swin_b_9dl", 1e-4)


def test_heavy_tailed_weights_swin_b_512x1024(golden):
    """Trained-like dynamic range (seeded_weights "heavy": LayerNorm / GroupNorm gammas log-uniform over 0.1..10, residual-stream
    outlier channels at 3e2..1e3, mask logits to +-34, RbA in its un-saturated range) through the default f16x3 arithmetic against
    the reference's own modules: the same bars as the base recipe (|d rba| < 1e-4, no argmax flip outside the near-tie set)."""
    from rba_amd import ops
    assert ops.SPLIT_MODE == "f16x3"
    _full_size(golden, "g5_swin_b_1dl_heavy_512x1024", "swin_b_1dl", 1e-4)


@pytest.mark.parametrize("fixture", ["g8_stages_swin_b_1dl_1024x2048", "g8_stages_swin_b_1dl_heavy_512x1024"])
def test_error_attribution_by_stage(golden, fixture):
    """what does the default arithmetic cost, and where?  The reference's INTERMEDIATES of a full-size forward (strided samples generated by
    make_golden.py --full from the reference's own modules: res2..res5, mask_features, mask logits, class logits, sem_seg, rba) against the product's, stage by
    stage, in three arithmetic configurations: the default (f16x3 everywhere, fused stage-1 attention), the same without the fused attention kernel (is the
    fusion's summation order visible?) and bf16x6 GEMMs (2^-24 products: is the error f16x3's 22-bit operands or re-association?).  Per-stage bounds are
    asserted; the table goes to gpurun_out/ (docs/measurements.md quotes it)."""
    import json
    from rba_amd import ops
    g = golden(fixture)
    g5 = golden(fixture.replace("g8_stages", "g5"))                                     # the complete argmax map + near-tie set of the same forward
    recipe = str(g["recipe"])
    h, w = (int(v) for v in g["hw"])
    model, a, sd = build(str(g["arch"]), int(g["seed"]), None if recipe == "base" else recipe)
    model.graph_replay = False
    gen = torch.Generator().manual_seed(int(g["img_seed"]))
    image = torch.randint(0, 256, (3, h, w), generator=gen, dtype=torch.uint8)
    ref_arg = T(g5["argmax_full"].astype(np.int64))
    tie = torch.zeros(h * w, dtype=torch.bool)
    tie[T(g5["neartie_idx"])] = True
    tie = tie.view(h, w)
    ys, xs, ys4, xs4 = (T(g[k]).long() for k in ("ys", "xs", "ys4", "xs4"))

    def run():
        r = {}
        feats = model.backbone(model.preprocess([{"image": image}])[0])
        for k in ("res2", "res3", "res4", "res5"):
            fy, fx = T(g["fy_" + k]).long(), T(g["fx_" + k]).long()
            ref = T(g["feat_" + k + "_s"])
            got = feats[k][0].cpu()[:, fy][:, :, fx]
            r[k] = ((got - ref).abs().max().item(), ref.abs().max().item())
        mf = model.sem_seg_head.pixel_decoder.forward_features(feats)[0][0].cpu()
        ref = T(g["mask_features_s"])
        r["mask_features"] = ((mf[:, T(g["my"]).long()][:, :, T(g["mx"]).long()] - ref).abs().max().item(), ref.abs().max().item())
        mask_cls, mask_pred, _, _ = model.predict([{"image": image}])
        ref = T(g["pred_masks_s"])
        r["pred_masks"] = ((mask_pred[0].cpu()[:, ys4, xs4] - ref).abs().max().item(), ref.abs().max().item())
        r["pred_logits"] = (maxerr(mask_cls[0], g["pred_logits"]), float(np.abs(g["pred_logits"]).max()))
        out = model([{"image": image}], return_argmax=True)[0]
        ref = T(g["sem_s"])
        r["sem_seg"] = ((out["sem_seg"].cpu()[:, ys[:512], xs[:512]] - ref).abs().max().item(), ref.abs().max().item())
        ref = T(g["rba_s"])
        r["rba"] = ((out["rba"].cpu()[ys, xs] - ref).abs().max().item(), ref.abs().max().item())
        flips = out["argmax"].cpu().long() != ref_arg
        r["argmax_flips"] = (int(flips.sum()), int((flips & ~tie).sum()))
        return r

    table = {"default (f16x3, fused stage-1 attention)": run()}
    prev = ops.SWIN_ATTN_FUSED
    try:
        ops.SWIN_ATTN_FUSED = False
        table["f16x3, unfused attention (LN -> K6 -> K5 -> K6)"] = run()
    finally:
        ops.SWIN_ATTN_FUSED = prev
    with ops.split_mode("bf16x6"):
        table["bf16x6 GEMMs (2^-24 products; K5 / K4 / K1 stay f16x3)"] = run()
    stages = ["res2", "res3", "res4", "res5", "mask_features", "pred_masks", "pred_logits", "sem_seg", "rba"]
    print(f"\n{fixture} ({recipe} weights, {h}x{w}): max |product - reference| per stage (reference magnitude in brackets), near-tie pixels {int(tie.sum())}")
    for mode, r in table.items():
        print(f"  {mode}")
        print("    " + "  ".join(f"{k} {r[k][0]:.1e} [{r[k][1]:.1f}]" for k in stages) + f"  argmax flips {r['argmax_flips'][0]} (outside near-ties {r['argmax_flips'][1]})")
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"r05_attribution_{recipe}.json"), "w") as f:
        json.dump({"fixture": fixture, "near_tie_pixels": int(tie.sum()), "pixels": h * w, "table": table}, f, indent=1)
    for mode, r in table.items():
        for k in ("res2", "res3", "res4", "res5", "mask_features"):
            assert r[k][0] <= 3e-5 * max(1.0, r[k][1]), (mode, k, r[k])                   # relative to the stage's own range (heavy: residual outliers to 1e3)
        assert r["pred_logits"][0] < 1e-4 and r["pred_masks"][0] <= 5e-4 * max(1.0, r["pred_masks"][1] / 10.0), (mode, r)
        assert r["sem_seg"][0] < 1e-4 and r["rba"][0] < 1e-4, (mode, r)                    # north star: scores within 1e-4
        assert r["argmax_flips"][1] == 0 and r["argmax_flips"][0] <= int(tie.sum()), (mode, r)


@pytest.mark.parametrize("fixture, arch_name", [("g7_metrics_swin_b_9dl_720x1280", "swin_b_9dl"),
                                                ("g7_metrics_swin_b_1dl_1024x2048", "swin_b_1dl")])
def test_metric_parity_full_size(golden, fixture, arch_name):
    """BASELINE configs[4] "AuPRC/FPR95 parity check" (Swin-B 9dl, 720x1280) and its configs[1] twin: AUROC / AuPRC / FPR95 of the
    product's RbA maps on four seeded images, pooled as evaluate_ood.py:195-235 pools them, against the SAME statistics of the
    reference's maps computed with scikit-learn by the support.py:247-303 logic (tests/golden/make_golden.py::g_metric_parity),
    under score-independent Bernoulli(0.03) labels and under labels that follow the reference's score.  North star: 3 decimals."""
    from rba_amd.metrics import ood_metrics, select_labelled
    from rba_amd.seeded_weights import seeded_ood_labels
    g = golden(fixture)
    model, a, _ = build(arch_name, int(g["seed"]))
    h, w = (int(v) for v in g["hw"])
    off_a, off_b = (int(v) for v in g["label_seed_offsets"])
    scores, lab_a, lab_b = [], [], []
    for i, s in enumerate(int(v) for v in g["img_seeds"]):
        gen = torch.Generator().manual_seed(s)
        image = torch.randint(0, 256, (3, h, w), generator=gen, dtype=torch.uint8)
        rba = model.rba_scores([{"image": image}])[0]
        assert abs(float(rba.mean()) - float(g["rba_mean"][i])) < 1e-4
        scores.append(rba)
        la = seeded_ood_labels(h, w, off_a + s)
        lb = la.clone()
        ood = T(np.unpackbits(g["labels_corr_bits"][i])[: h * w].reshape(h, w))
        lb[lb != 255] = ood[lb != 255]
        lab_a.append(la)
        lab_b.append(lb)
    sc = torch.stack(scores)
    for labs, key, n_ood in ((lab_a, "metrics_indep", int(g["n_ood"][0])), (lab_b, "metrics_corr", int(g["n_ood"][1]))):
        lab = torch.stack(labs).cuda()
        assert int((lab == 1).sum()) == n_ood
        got = ood_metrics(*select_labelled(sc, lab))
        want = dict(zip(("auroc", "aupr", "fpr95"), (float(v) for v in g[key])))
        d = {k: abs(got[k] - want[k]) for k in want}
        print(f"{fixture} {key}: reference {want} product {got} |d| {d}")
        assert all(v < 5e-4 for v in d.values()), d


def test_evaluate_ood_cli_end_to_end(tmp_path, monkeypatch):
    """`python -m rba_amd.evaluate_ood` flow: models folder (config.yaml + model_final.pth) x datasets on disk ->
    results/<model>/results.pkl = {dataset: {auroc, aupr, fpr95}} (evaluate_ood.py:238-288), checked against the oracle."""
    import pickle
    import yaml
    from oracle import ref_metrics
    from rba_amd import evaluate_ood as E
    from tests.test_datasets_cpu import make_fs_laf, make_road_anomaly
    a = A.complete(A.ARCHS["tiny1"])
    sd = A.seeded_weights(a, 0)
    mdir = tmp_path / "ckpts" / "tiny"
    mdir.mkdir(parents=True)
    cfg = {"MODEL": {"SWIN": {"EMBED_DIM": 32, "DEPTHS": [2, 2, 2, 2], "NUM_HEADS": [1, 2, 4, 8], "WINDOW_SIZE": 6},
                     "SEM_SEG_HEAD": {"CONVS_DIM": 64, "MASK_DIM": 64, "TRANSFORMER_ENC_LAYERS": 2,
                                      "DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES": ["res5"]},
                     "MASK_FORMER": {"HIDDEN_DIM": 64, "NHEADS": 2, "NUM_OBJECT_QUERIES": 16, "DIM_FEEDFORWARD": 128,
                                     "DEC_LAYERS": 2}}}
    (mdir / "config.yaml").write_text(yaml.safe_dump(cfg))
    torch.save({"model": sd}, mdir / "model_final.pth")
    data = tmp_path / "data"
    # images of at least 64 x 64 after padding: with a 1 x 1 res5 map the library's 1x1 convolution (MIOpen, one pixel) is not
    # bitwise reproducible from run to run (2e-7 on its output, found with a per-op replay), which no real image size reaches
    ra_imgs, ra_labs = make_road_anomaly(str(data), h=48, w=80)
    fs_imgs, fs_labs = make_fs_laf(str(data), h=40, w=72)
    monkeypatch.chdir(tmp_path)
    argv = ["--models_folder", str(tmp_path / "ckpts"), "--datasets_folder", str(data), "--out_path", str(tmp_path / "results"),
            "--verbose", "false"]
    E.main(argv)
    with open(tmp_path / "results" / "tiny" / "results.pkl", "rb") as f:
        res = pickle.load(f)
    assert sorted(res) == ["fishyscapes_laf", "road_anomaly"]
    # oracle: CPU forward + sklearn metrics over the same images / labels
    for name, imgs, labs in (("road_anomaly", ra_imgs, [(l == 2).astype(np.int64) for l in ra_labs]),
                             ("fishyscapes_laf", fs_imgs, [l.astype(np.int64) for l in fs_labs])):
        scores = np.stack([ref_model.forward(torch.from_numpy(im.transpose(2, 0, 1).copy()), sd, a)["rba"].numpy() for im in imgs])
        want = ref_metrics.evaluate_ood(scores, np.stack(labs)[:, None])
        for k in want:
            assert abs(res[name][k] - want[k]) < 5e-4, (name, k, res[name][k], want[k])     # north star: 3 decimals
    # second run: results exist -> skipped, file untouched
    mtime = (tmp_path / "results" / "tiny" / "results.pkl").stat().st_mtime_ns
    E.main(argv)
    assert (tmp_path / "results" / "tiny" / "results.pkl").stat().st_mtime_ns == mtime
    # every pipelining mode of the scoring loop gives the same pooled metrics: decode threads / processes x HIP streams
    for k, extra in enumerate((["--num_workers", "0", "--streams", "1", "--graph", "0"], ["--num_workers", "3", "--streams", "2", "--graph", "0"],
                               ["--num_workers", "2", "--streams", "1", "--graph", "0"], ["--num_workers", "3", "--streams", "3", "--graph", "1"],
                               ["--num_workers", "0", "--streams", "1", "--graph", "1"],
                               ["--num_workers", "3", "--streams", "3", "--graph", "1", "--loader", "processes"])):
        out = tmp_path / f"results_{k}"
        E.main(argv[:4] + ["--out_path", str(out), "--verbose", "false"] + extra)
        with open(out / "tiny" / "results.pkl", "rb") as f:
            alt = pickle.load(f)
        for name in res:
            assert all(abs(alt[name][m] - res[name][m]) < 1e-12 for m in res[name]), \
                (extra, name, {m: (repr(alt[name][m]), repr(res[name][m])) for m in res[name]})


@pytest.mark.parametrize("name", ["tiny1", "tiny3"])
def test_sparse_intermediate_heads_are_exact(name):
    """evaluating the intermediate mask logits only at the 2x2 source pixels of each attention-mask cell gives
    bit-identical results to the dense einsum + bilinear down-sample"""
    model, a, _ = build(name, 1)
    g = torch.Generator().manual_seed(5)
    image = torch.randint(0, 256, (3, 60, 90), generator=g, dtype=torch.uint8)
    pred = model.sem_seg_head.predictor
    assert pred.sparse_intermediate_heads
    cls_s, msk_s, _, _ = model.predict([{"image": image}])
    pred.sparse_intermediate_heads = False
    cls_d, msk_d, _, _ = model.predict([{"image": image}])
    assert torch.equal(cls_s, cls_d) and torch.equal(msk_s, msk_d)


@pytest.mark.parametrize("name", ["tiny1", "tiny3"])
def test_cached_initial_heads_are_exact_and_follow_the_weights(name):
    """round 6: the prediction heads of the un-decoded queries (reference :427-430) see only learnt parameters -- decoder_norm, class_embed and mask_embed of that
    call are computed once per checkpoint.  Same bits as evaluating them per image (batch 1 and 2, aux pred_logits included); a weight update invalidates the
    cache."""
    model, a, _ = build(name, 1)
    g = torch.Generator().manual_seed(6)
    images = [{"image": torch.randint(0, 256, (3, 60, 90), generator=g, dtype=torch.uint8)} for _ in range(2)]
    pred = model.sem_seg_head.predictor
    assert pred.cache_initial_heads
    for batch in (images[:1], images):
        cls_c, msk_c, _, _ = model.predict(batch)
        cls_c2, msk_c2, _, _ = model.predict(batch)                    # second call: served from the cache
        pred.cache_initial_heads = False
        cls_p, msk_p, _, _ = model.predict(batch)
        pred.cache_initial_heads = True
        assert torch.equal(cls_c, cls_p) and torch.equal(msk_c, msk_p) and torch.equal(cls_c2, cls_p) and torch.equal(msk_c2, msk_p)
    feats = model.backbone(model.preprocess(images[:1])[0])
    mf, _, ms = model.sem_seg_head.pixel_decoder.forward_features(feats)
    aux_c = pred(ms, mf)["aux_outputs"][0]["pred_logits"]
    with torch.no_grad():
        pred.query_feat.weight.mul_(1.5)                              # in-place update: parameter version changes -> recomputed
    aux_new = pred(ms, mf)["aux_outputs"][0]["pred_logits"]
    pred.cache_initial_heads = False
    aux_plain = pred(ms, mf)["aux_outputs"][0]["pred_logits"]
    assert torch.equal(aux_new, aux_plain) and not torch.equal(aux_new, aux_c)


def test_batch_of_different_sizes_and_float_input(golden):
    """ImageList semantics (maskformer_model.py:255-257): each image is normalised, then zero-padded bottom/right to the
    common size (multiple of 32); every image's outputs are cropped back to its own size.  NOTE: padding an image to a
    LARGER common size changes its result (the network sees more zero border), exactly as in the reference -- so the oracle
    is run on the identically padded input."""
    import torch.nn.functional as F
    from oracle import ref_ops
    model, a, sd = build("tiny1", 0)
    g = torch.Generator().manual_seed(9)
    im0 = torch.randint(0, 256, (3, 60, 90), generator=g, dtype=torch.uint8)
    im1 = torch.randint(0, 256, (3, 33, 70), generator=g, dtype=torch.uint8).float()        # float input, 0..255
    outs = model([{"image": im0}, {"image": im1}], return_argmax=True)
    assert outs[0]["sem_seg"].shape == (19, 60, 90) and outs[1]["sem_seg"].shape == (19, 33, 70)
    ref0 = ref_model.forward(im0, sd, a)                                                       # common size 64 x 96 = im0's own pad
    assert maxerr(outs[0]["rba"], ref0["rba"]) < 1e-4 and maxerr(outs[0]["sem_seg"], ref0["sem_seg"]) < 1e-4
    # oracle for im1 inside a 64 x 96 canvas: normalise, pad, run the padded tensor through the oracle stages
    mean = torch.tensor(ref_model.PIXEL_MEAN).view(-1, 1, 1); std = torch.tensor(ref_model.PIXEL_STD).view(-1, 1, 1)
    x = F.pad((im1 - mean) / std, (0, 96 - 70, 0, 64 - 33))[None]
    feats = ref_model.swin_backbone(x, sd, a)
    mf, ms = ref_model.pixel_decoder(feats, sd, a)
    cls, masks = ref_model.transformer_decoder(ms, mf, sd, a)
    sem = ref_ops.semantic_inference(cls[0], ref_ops.upsample_bilinear(masks, (64, 96))[0])[:, :33, :70]
    assert maxerr(outs[1]["sem_seg"], sem) < 1e-4 and maxerr(outs[1]["rba"], ref_ops.rba_score(sem)) < 1e-4


def test_requested_output_resolution(golden):
    """batched_inputs[i]["height"/"width"]: sem_seg_postprocess resizes the cropped result (maskformer_model.py:312-313, 330-332)."""
    from oracle import ref_ops
    g = golden("g4_tiny1_60x90")
    model, a, sd = build("tiny1", 0)
    image = T(g["image"])
    out = model([{"image": image, "height": 120, "width": 200}])[0]
    assert out["sem_seg"].shape == (19, 120, 200)
    want = ref_ops.upsample_bilinear(T(g["sem_seg"])[None], (120, 200))[0]
    assert maxerr(out["sem_seg"], want) < 1e-4
    assert maxerr(out["rba"], ref_ops.rba_score(want)) < 1e-4
    with pytest.raises(NotImplementedError):
        model([{"image": image}], return_aux=True)
    # TEST.SEM_SEG_POSTPROCESSING_BEFORE_INFERENCE (maskformer_model.py:205-209, 316-320): the mask logits are cropped and resized
    # first, the class contraction runs at the requested resolution -- a different result (sigmoid does not commute with resizing)
    model.sem_seg_postprocess_before_inference = True
    out2 = model([{"image": image, "height": 120, "width": 200}], return_argmax=True)[0]
    up = ref_ops.upsample_bilinear(T(g["pred_masks"])[None], (64, 96))[0][:, :60, :90]
    up = ref_ops.upsample_bilinear(up[None], (120, 200))[0]
    want2 = ref_ops.semantic_inference(T(g["pred_logits"]), up)
    assert maxerr(out2["sem_seg"], want2) < 2e-4 and maxerr(out2["rba"], ref_ops.rba_score(want2)) < 2e-4
    assert (want2 - want).abs().max() > 1e-3
    same = model([{"image": image}])[0]                           # no size request: the flag changes nothing
    assert maxerr(same["sem_seg"], g["sem_seg"]) < 1e-4


@pytest.mark.parametrize("arch", ["r50_1dl", "swin_b_1dl"])
def test_c1_random_tensor_256x512_vs_oracle(arch):
    """BASELINE config C1 (ResNet-50 Mask2Former, 1 decoder layer, 100 queries, torch.randn(1,3,256,512) seed 0): the product against
    the CPU oracle -- ResNet-50 restated from Detectron2's definition (parity unpinned), and Swin-B 1dl on the same input."""
    model, a, sd = build(arch, 0)
    x = torch.randn(3, 256, 512, generator=torch.Generator().manual_seed(0))
    out = model([{"image": x}], return_argmax=True)[0]
    ref = ref_model.forward(x, sd, a)
    assert maxerr(out["sem_seg"], ref["sem_seg"]) < 1e-4 and maxerr(out["rba"], ref["rba"]) < 1e-4
    bad, flips = argmax_bad(out["argmax"], ref["sem_seg"])
    assert bad == 0, (bad, flips)


def test_channels_last_fpn_path_matches_nchw_path_and_batches():
    """The token-layout (channels-last) pixel-decoder path -- 1x1 convs as Linears, 3x3 convs as bf16x6 implicit GEMMs, NHWC
    GroupNorm / resample, NCHW mask features -- against the NCHW path (MIOpen convs) on the same Swin-B features, for a batch
    of two images; and the batch against the same images run one at a time."""
    model, a, sd = build("swin_b_1dl", 0)

    def maxerr(u, v):
        return (u.double() - v.double()).abs().max().item()

    pd = model.sem_seg_head.pixel_decoder
    g = torch.Generator().manual_seed(21)
    ims = [torch.randint(0, 256, (3, 256, 384), generator=g, dtype=torch.uint8) for _ in range(2)]
    with torch.no_grad():
        x, _ = model.preprocess([{"image": im} for im in ims])
        feats = model.backbone(x)
        assert pd._channels_last_ok(feats), "Swin must hand over channels-last views the FPN can consume"
        mf_cl, out0_cl, ms_cl = pd.forward_features(feats)
        ok = pd._channels_last_ok
        try:
            pd._channels_last_ok = lambda f: False                                              # force the NCHW path
            mf_nc, out0_nc, ms_nc = pd.forward_features(feats)
        finally:
            pd._channels_last_ok = ok
        assert mf_cl.shape == mf_nc.shape == (2, 256, 64, 96) and mf_cl.is_contiguous()
        scale = float(mf_nc.abs().max())
        assert maxerr(mf_cl, mf_nc) < 2e-5 * max(scale, 1.0), (maxerr(mf_cl, mf_nc), scale)
        assert maxerr(out0_cl, out0_nc) < 2e-5 and all(maxerr(u, v) < 2e-5 for u, v in zip(ms_cl, ms_nc))
        # batch of two == each image alone (no cross-image leakage in the implicit-GEMM borders / per-image statistics)
        for b in range(2):
            xb, _ = model.preprocess([{"image": ims[b]}])
            mf_b, _, _ = pd.forward_features(model.backbone(xb))
            assert maxerr(mf_b[0], mf_cl[b]) < 2e-5 * max(scale, 1.0)


def test_open_panoptic_inference_vs_oracle():
    """MaskFormer.panoptic_inference (closed-set merge + RbA-threshold / morphology / connected-components open-set branch) on
    synthetic queries against the oracle restatement; then through forward() with TEST.PANOPTIC_ON."""
    from oracle import ref_ops
    model, a, sd = build("tiny1", 0)
    model.object_mask_threshold, model.overlap_threshold = 0.5, 0.6
    g = torch.Generator().manual_seed(13)
    Q, K, H, W = 16, 19, 96, 160
    mask_cls = torch.randn(Q, K + 1, generator=g)
    mask_cls[:, K] -= 2.0
    cls_of = [3, 3, 11, 12, 0, 13, 8, 11]                                 # two "stuff" queries share class 3 -> merged
    for q, c in enumerate(cls_of):
        mask_cls[q, c] += 8.0
    mask_cls[len(cls_of):, K] += 9.0                                       # the rest predict void
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    mask_pred = torch.full((Q, H, W), -6.0)
    for q in range(len(cls_of)):                                           # overlapping boxes; a band on the right stays unclaimed
        y0, x0 = 8 * q, 14 * q
        mask_pred[q] = torch.where((yy >= y0) & (yy < y0 + 40) & (xx >= x0) & (xx < x0 + 36), 5.0 - 0.3 * q, -6.0)
    mask_pred += 0.05 * torch.randn(Q, H, W, generator=g)
    for open_p in (False, True):
        pan, info, rba = ref_ops.panoptic_inference(mask_cls, mask_pred, set(model.thing_classes), 0.5, 0.6, open_p, -0.3, 50)
        out = model.panoptic_inference(mask_cls.cuda(), mask_pred.cuda(), open_p, -0.3, 50, return_ood_pred=open_p)
        assert np.array_equal(out[0].cpu().numpy(), pan) and out[1] == info
        if open_p:
            assert maxerr(out[2], rba) < 1e-5 and any(s["category_id"] == 255 for s in info), "the test must exercise the open-set branch"
    assert sum(1 for s in info if s["category_id"] == 3) == 1, "stuff regions of one class merge"
    # through forward(): TEST.PANOPTIC_ON -> "panoptic_seg" next to "sem_seg"
    model.panoptic_on, model.object_mask_threshold, model.overlap_threshold = True, 0.0, 0.0
    with torch.no_grad():                                                  # make some queries prefer a non-void class
        model.sem_seg_head.predictor.class_embed.bias[:K] += 6.0
    im = torch.randint(0, 256, (3, 60, 90), generator=g, dtype=torch.uint8)
    r = model([{"image": im}], panoptic_ood_threshold=-0.3, panoptic_pixel_min=20, return_panoptic_ood=True)[0]
    assert len(r["panoptic_seg"]) == 3, "queries were kept, so the open-set branch must have run"
    pan_g, info_g, ood = r["panoptic_seg"]
    assert pan_g.shape == (60, 90) and pan_g.dtype == torch.int32 and maxerr(ood, r["rba"].cpu()) < 1e-4
    assert sorted(s["id"] for s in info_g) == list(range(1, len(info_g) + 1)) and int(pan_g.max()) <= len(info_g)


def test_rccl_executes_the_metric_exchange_in_a_one_rank_group():
    """round 5: no gpurun box has two GPUs, so RCCL never saw this repository's collectives.  A ONE-rank `nccl` process group on device 0 runs them for real:
    all_gather of the sizes + padded all_gather_into_tensor on ragged DEVICE tensors (distributed._gather_padded), the histogram all_reduce, and the
    pooled metrics -- equal to the single-process values; then bench.py's exchange leg through the same group (--rccl-one-rank: dist_backend nccl)."""
    import json
    import subprocess
    import sys
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(REPO, "tools", "rccl_one_rank.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k_, None)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert res["backend"] == "nccl"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--rccl-one-rank", "--steps", "2", "--warmup", "1", "--height", "512", "--width", "1024",
                        "--streams", "1", "--n-images", "2", "--no-cpu-baseline", "--sustain", "0.5"], capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["dist_backend"] == "nccl" and line["n_gpus"] == 1 and line["metric_exchange_ms"] > 0 and set(line["pooled_metrics"]) == {"auroc", "aupr", "fpr95"}


def test_bench_self_launch_two_ranks_share_device():
    """`python bench.py --gpus 2` OUTSIDE torchrun spawns its own two ranks (here both on device 0, gloo): one JSON line with
    n_gpus = 2, the metric exchange timed, and pooled metrics equal to the single-process metric over both ranks' images."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import bench
    from rba_amd.metrics import ood_metrics
    h, w, steps, warm, nimg = 512, 1024, 2, 1, 2
    env = dict(os.environ, RBA_BENCH_SHARE_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warm),
                        "--streams", "1", "--n-images", str(nimg), "--height", str(h), "--width", str(w), "--no-cpu-baseline",
                        "--sustain", "0.5"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == steps and res["scaling"] == "weak"
    assert res["dtype"].startswith("f32 (f16x3") and res["config"]["global_batch_per_step"] == 2
    assert res["sustained"]["seconds"] >= 0.5 and res["sustained"]["images_per_s"] > 0
    assert res["metric_exchange_ms"] > 0 and res["value"] > 0 and res["roofline"]["frac"] > 0
    assert len(res["per_rank"]["images_per_s"]["all"]) == 2 and res["n_ranks_seen"] == 2
    # single process: the last timed image of each rank (bench.py: image (warmup + steps - 1) % n_images, seed 1234 + 1000 rank + i)
    model, a, _ = build("swin_b_1dl", 0)
    model.fused_upsample = False
    idx = (warm + steps - 1) % nimg
    ss, ll = [], []
    for rank in range(2):
        g = torch.Generator().manual_seed(1234 + rank * 1000 + idx)
        image = torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8).cuda()
        rba = model.rba_scores([{"image": image}])[0]
        lab, valid = bench.exchange_inputs(rank, h, w, rba.device)
        ss.append(rba[valid])
        ll.append(lab[valid])
    want = ood_metrics(torch.cat(ss), torch.cat(ll))
    # the two bench ranks and this process are three processes: library algorithm choices (MIOpen / hipBLASLt heuristics) are per
    # process, so score maps may differ in their last bits (every op is bitwise reproducible WITHIN a process: tools/op_replay.py);
    # and across processes: tools/op_replay.py, tools/concurrency_check.py); rank statistics over ~1e6 pixels then move by a few
    # pixels' worth (observed: 0 most runs, 2.5e-9 on AUROC once, 2.2e-6 = 4 pixels on FPR95 once, after 17 other tests had run
    # in this process)
    for k in want:
        assert abs(res["pooled_metrics"][k] - want[k]) < 1e-5, (k, res["pooled_metrics"], want)


def test_bench_self_launch_eight_ranks_share_device():
    """VERDICT round 5 "next" #3: the widest multi-rank plumbing run was world 2.  `python bench.py --gpus 8 --images-per-gpu 2` (BASELINE configs[2]'s launch
    shape; a tiny net so that eight processes fit one device's time budget) with all ranks on device 0 over gloo: one line, n_gpus = n_ranks_seen = 8, every
    rank's own clock reported, every rank pinned to its own share of the CPUs, 16 images per step, the pooled metric exchange over 8 ragged shards timed."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RBA_BENCH_SHARE_DEVICE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "8", "--images-per-gpu", "2", "--steps", "2", "--warmup", "1", "--arch", "tiny1",
                        "--n-images", "2", "--height", "96", "--width", "160", "--no-cpu-baseline", "--sustain", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["n_ranks_seen"] == 8 and res["dist_backend"] == "gloo" and res["scaling"] == "weak"
    assert res["config"]["global_batch_per_step"] == 16 and res["config"]["images_per_gpu_per_step"] == 2
    pr = res["per_rank"]
    assert len(pr["images_per_s"]["all"]) == 8 and 0 < pr["images_per_s"]["min"] <= pr["images_per_s"]["max"]
    assert res["value"] <= 8 * pr["images_per_s"]["min"] * 1.0001                       # whole-job rate from the SLOWEST rank's clock
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 8:
        assert pr["cpu_affinity"]["cpus_per_rank"]["min"] >= 1 and pr["cpu_affinity"]["cpus_per_rank"]["max"] <= max(1, ncpu // 8) + 1, pr["cpu_affinity"]
        assert pr["cpu_affinity"]["rank0"]["bound"] is True
    assert res["metric_exchange_ms"] > 0 and all(0.0 <= res["pooled_metrics"][k] <= 1.0 for k in ("auroc", "aupr", "fpr95"))


def test_split_mode_switch_bf16x6_matches_f16x3(monkeypatch):
    """ops.SPLIT_MODE selects the arithmetic of the token Linears: the full-range bf16x6 form and the default f16x3 form are both
    fp32-accurate, so the whole-path score maps agree far inside the 1e-4 tolerance (and the weight planes are re-split on the switch)."""
    from rba_amd import ops
    model, a, _ = build("swin_b_1dl", 0)
    g = torch.Generator().manual_seed(21)
    image = torch.randint(0, 256, (3, 512, 1024), generator=g, dtype=torch.uint8).cuda()
    assert ops.SPLIT_MODE == "f16x3"
    r3 = model.rba_scores([{"image": image}])[0].clone()
    fc1 = model.backbone.layers[2].blocks[0].mlp.fc1
    assert fc1._rba_planes["f16x3"][1].dtype == torch.float16
    with ops.split_mode("bf16x6"):                                  # (per thread since round 5: set through the context manager, not by assignment)
        r6 = model.rba_scores([{"image": image}])[0]
    assert fc1._rba_planes["bf16x6"][1].dtype == torch.bfloat16
    assert (r3 - r6).abs().max().item() < 5e-5
    assert ops.SPLIT_MODE == "f16x3"
    assert torch.equal(model.rba_scores([{"image": image}])[0], r3)


@pytest.mark.parametrize("arch_name,score_funcs", [("tiny1", ("energy", "neg_logit_sum")), ("tiny1_dh", ("dense_hybrid",))])
def test_evaluate_ood_graph_replay_equals_eager(tmp_path, arch_name, score_funcs):
    """`--graph 1` (the default: each stream's forward replayed from a captured hipGraph) gives the metrics of `--graph 0` for every
    score function, the DenseHybrid head included."""
    import pickle
    import yaml
    from rba_amd import evaluate_ood as E
    from tests.test_datasets_cpu import make_fs_laf, make_road_anomaly
    a = A.complete(A.ARCHS[arch_name])
    mdir = tmp_path / "ckpts" / "m"
    mdir.mkdir(parents=True)
    cfg = {"MODEL": {"SWIN": {"EMBED_DIM": 32, "DEPTHS": [2, 2, 2, 2], "NUM_HEADS": [1, 2, 4, 8], "WINDOW_SIZE": 6},
                     "SEM_SEG_HEAD": {"CONVS_DIM": 64, "MASK_DIM": 64, "TRANSFORMER_ENC_LAYERS": 2,
                                      "DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES": ["res5"]},
                     "MASK_FORMER": {"HIDDEN_DIM": 64, "NHEADS": 2, "NUM_OBJECT_QUERIES": 16, "DIM_FEEDFORWARD": 128, "DEC_LAYERS": 2,
                                     "DENSE_HYBRID_LOSS": arch_name == "tiny1_dh"}}}
    (mdir / "config.yaml").write_text(yaml.safe_dump(cfg))
    torch.save({"model": A.seeded_weights(a, 0)}, mdir / "model_final.pth")
    data = tmp_path / "data"
    make_road_anomaly(str(data), n=7, h=48, w=80)
    make_fs_laf(str(data), n=7, h=40, w=72)
    for sf in score_funcs:
        res = {}
        for g in ("0", "1"):
            out = tmp_path / f"res_{sf}_{g}"
            E.main(["--models_folder", str(tmp_path / "ckpts"), "--datasets_folder", str(data), "--out_path", str(out), "--verbose", "0",
                    "--score_func", sf, "--graph", g, "--num_workers", "2", "--streams", "2"])
            with open(out / "m" / "results.pkl", "rb") as f:
                res[g] = pickle.load(f)
        for d in res["0"]:
            for k in res["0"][d]:
                assert abs(res["0"][d][k] - res["1"][d][k]) < 1e-12, (sf, d, k, res["0"][d][k], res["1"][d][k])


@pytest.mark.parametrize("name,h,w", [("tiny1", 60, 90), ("swin_b_1dl", 250, 510)])
def test_fused_front_end_matches_library_path(name, h, w):
    """MaskFormer.fused_front_end (normalise + pad + im2col kernel, patch projection on K6) against the elementwise / F.conv2d path:
    same tokens to fp32 rounding, same scores well inside the tolerance; uint8 and fp32 images, ragged sizes."""
    model, a, _ = build(name, 0)
    g = torch.Generator().manual_seed(h + w)
    for image in (torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8), torch.rand(3, h, w, generator=g) * 255):
        image = image.cuda()
        assert model.fused_front_end
        r1 = model.rba_scores([{"image": image}])[0].clone()
        model.fused_front_end = False
        r0 = model.rba_scores([{"image": image}])[0]
        model.fused_front_end = True
        assert r1.shape == r0.shape == (h, w)
        assert (r1 - r0).abs().max().item() < 2e-5



def test_evaluate_ood_two_ranks_share_device_equals_single_process(tmp_path):
    """The sharded evaluator itself (image i -> rank i mod world, pooled ranking of all ranks' labelled pixels, support.py:275-290) under
    `torch.distributed.run --nproc-per-node 2`: both ranks on device 0, metric exchange over gloo (--share_device 1; RCCL refuses two
    ranks on one device) -- results.pkl must equal the single-process file."""
    import os
    import pickle
    import subprocess
    import sys
    import yaml
    from tests.test_datasets_cpu import make_fs_laf, make_road_anomaly
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = A.complete(A.ARCHS["tiny1"])
    mdir = tmp_path / "ckpts" / "tiny"
    mdir.mkdir(parents=True)
    cfg = {"MODEL": {"SWIN": {"EMBED_DIM": 32, "DEPTHS": [2, 2, 2, 2], "NUM_HEADS": [1, 2, 4, 8], "WINDOW_SIZE": 6},
                     "SEM_SEG_HEAD": {"CONVS_DIM": 64, "MASK_DIM": 64, "TRANSFORMER_ENC_LAYERS": 2,
                                      "DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES": ["res5"]},
                     "MASK_FORMER": {"HIDDEN_DIM": 64, "NHEADS": 2, "NUM_OBJECT_QUERIES": 16, "DIM_FEEDFORWARD": 128, "DEC_LAYERS": 2}}}
    (mdir / "config.yaml").write_text(yaml.safe_dump(cfg))
    torch.save({"model": A.seeded_weights(a, 0)}, mdir / "model_final.pth")
    data = tmp_path / "data"
    make_road_anomaly(str(data), n=7, h=96, w=160)                 # odd image counts: the shards are unequal
    make_fs_laf(str(data), n=5, h=80, w=144)
    common = ["--models_folder", str(tmp_path / "ckpts"), "--datasets_folder", str(data), "--verbose", "0", "--num_workers", "2"]
    env = dict(os.environ, PYTHONPATH=repo + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r1 = subprocess.run([sys.executable, "-m", "rba_amd.evaluate_ood"] + common + ["--out_path", str(tmp_path / "single")],
                        env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-2000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                         "--master-port", "29631", "-m", "rba_amd.evaluate_ood"] + common +
                        ["--out_path", str(tmp_path / "sharded"), "--share_device", "1"],
                       env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
    with open(tmp_path / "single" / "tiny" / "results.pkl", "rb") as f:
        one = pickle.load(f)
    with open(tmp_path / "sharded" / "tiny" / "results.pkl", "rb") as f:
        two = pickle.load(f)
    assert sorted(one) == sorted(two) == ["fishyscapes_laf", "road_anomaly"]
    for d in one:
        for k in one[d]:
            assert abs(one[d][k] - two[d][k]) < 1e-12, (d, k, repr(one[d][k]), repr(two[d][k]))


def test_model_graph_replay_equals_eager_and_survives_shape_churn():
    """MaskFormer.rba_scores replays a captured hipGraph from the third call of an image shape: bit-identical to the eager launches,
    per stream, and -- ADVICE r2 -- three interleaved image shapes do not evict each other's per-shape constants (the ShapeCache
    pins what a capture read)."""
    model, a, _ = build("tiny3", 0)
    shapes = [(60, 90), (64, 96), (96, 64), (48, 80), (80, 112)]
    g = torch.Generator().manual_seed(3)
    imgs = [[torch.randint(0, 256, (3, h, w), generator=g, dtype=torch.uint8).cuda() for _ in range(3)] for h, w in shapes]
    model.graph_replay = False
    want = [[model.rba_scores([{"image": im}], return_argmax=True)[0] for im in row] for row in imgs]
    model.graph_replay = True
    for rnd in range(4):                                            # round 0 eager, round 1 captures, rounds 2-3 replay, shapes interleaved
        for si, row in enumerate(imgs):
            for ii, im in enumerate(row):
                rba, arg = model.rba_scores([{"image": im}], return_argmax=True)[0]
                assert torch.equal(rba, want[si][ii][0]) and torch.equal(arg, want[si][ii][1]), (rnd, si, ii)
    assert model.live_graphs() == len(shapes)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                                   # another stream: its own graphs
        for rnd in range(3):
            r = model.rba_scores([{"image": imgs[0][1]}])[0]
            assert torch.equal(r, want[0][1][0])
    torch.cuda.current_stream().wait_stream(side)
    assert model.live_graphs() == len(shapes) + 1
    model.cpu()
    assert model.live_graphs() == 0


def test_no_library_gemm_or_convolution_on_any_path(monkeypatch):
    """VERDICT r4 missing #3: the product still shipped library fallbacks -- hipBLASLt for a batched forward's query Linears (> 128 rows), for the small Linears
    of the bf16x6 re-score mode and for anything below K6's tile threshold, MIOpen for the NCHW pixel-decoder layout.  Round 5: with F.linear / F.conv2d /
    torch.matmul-style entry points made to raise, (1) the tiny net (NCHW fallback layout, windows of 6, Linears far below every threshold), (2) a B = 2
    Swin-B forward (200 query rows) and (3) both again in bf16x6 mode still run -- and the batch equals the two single-image forwards."""
    import torch.nn.functional as F
    from rba_amd import ops

    def boom(name):
        def f(*a, **k):
            raise AssertionError(f"library call on the product path: {name}")
        return f
    for name in ("linear", "conv2d", "bilinear"):
        monkeypatch.setattr(F, name, boom("F." + name))
    monkeypatch.setattr(torch, "matmul", boom("torch.matmul"))
    monkeypatch.setattr(torch, "bmm", boom("torch.bmm"))
    monkeypatch.setattr(torch, "einsum", boom("torch.einsum"))
    g = torch.Generator().manual_seed(17)
    tiny, _, _ = build("tiny3", 0)
    tiny.graph_replay = False
    im_t = torch.randint(0, 256, (3, 60, 90), generator=g, dtype=torch.uint8).cuda()
    big, _, _ = build("swin_b_1dl", 0)
    big.graph_replay = False
    ims = [torch.randint(0, 256, (3, 256, 512), generator=g, dtype=torch.uint8).cuda() for _ in range(2)]
    for mode in ("f16x3", "bf16x6"):
        with ops.split_mode(mode):
            t1 = tiny.rba_scores([{"image": im_t}])[0]
            assert torch.isfinite(t1).all()
            both = big.rba_scores([{"image": ims[0]}, {"image": ims[1]}])
            singles = [big.rba_scores([{"image": im}])[0] for im in ims]
            for b_, s_ in zip(both, singles):
                assert b_.shape == s_.shape and torch.isfinite(b_).all() and (b_ - s_).abs().max().item() < 2e-5, mode


def test_non_finite_f16x3_score_is_rescored_on_bf16x6(tmp_path, monkeypatch):
    """A residual-stream value beyond f16's range makes the f16x3 Linear answer NaN (never a wrong number); both evaluator loops must
    catch the non-finite map before it reaches the rank statistics and score that image again on the full-range bf16x6 kernels."""
    from rba_amd import evaluate_ood as E
    from rba_amd import ops
    from rba_amd.support import OODEvaluator
    monkeypatch.setattr(ops, "TILES_MIN", 1)                        # the tiny net's Linears on K6 too (they are below the product's tile threshold)
    model, a, sd = build("tiny3", 0)
    with torch.no_grad():                                           # one LayerNorm output channel at 1e5: beyond f16 (65504), fine in fp32 / bf16x6
        model.backbone.layers[1].blocks[0].norm1.bias[3] = 1.0e5
    g = torch.Generator().manual_seed(8)
    imgs = [torch.randint(0, 256, (3, 128, 192), generator=g, dtype=torch.uint8) for _ in range(3)]
    model.graph_replay = False
    bad = model.rba_scores([{"image": imgs[0].cuda()}])[0]
    if bool(torch.isfinite(bad).all()):
        pytest.skip("this geometry does not run the f16x3 Linear on the poisoned activation")
    with ops.split_mode("bf16x6"):
        want = [model.rba_scores([{"image": im.cuda()}])[0] for im in imgs]
    assert all(bool(torch.isfinite(w_).all()) for w_ in want)
    model.graph_replay = True
    ev = OODEvaluator(model, E.get_logits, E.get_RbA)
    gt = torch.zeros(1, 128, 192, dtype=torch.long)
    gt[:, 10:40, 10:60] = 1
    scores, gts = ev.compute_anomaly_scores([(im[None], gt) for im in imgs], device=torch.device("cuda"))
    assert np.isfinite(scores).all() and ev.bf16x6_rescored_images == [0, 1, 2]
    for k_, (s_, w_) in enumerate(zip(scores, want)):
        # the same kernels on the same input: bit-equal, no tolerance.  History: rounds 3-4 saw a last-bit difference here on some boxes and traced it to
        # MIOpen's 1 x 1 convolution on the NCHW fallback layout of this tiny net (profiles/r04_flake_cause.txt).  Since round 5 no library GEMM or convolution
        # runs on ANY path of the product (test_no_library_gemm_or_convolution_on_any_path), so every launch here is one of this repository's kernels, none of
        # which uses floating-point atomics -- and the assertion is strict again.
        assert np.array_equal(s_, w_.cpu().numpy()), (k_, float(np.abs(s_ - w_.cpu().numpy()).max()), int((s_ != w_.cpu().numpy()).sum()))
    r = ev.evaluate_ood(scores, gts, verbose=False)
    assert all(np.isfinite(v) for v in r.values())


@pytest.mark.gpu
def test_to_device_stages_shared_memory_tensors():
    """rba_amd.h2d.to_device: a DataLoader worker's item (shared-memory tensor) reaches the device through a page-locked staging buffer --
    same values, ring reuse safe across more items than slots, page-locked and device tensors pass through"""
    from rba_amd.h2d import to_device
    g = torch.Generator().manual_seed(3)
    items = [torch.randint(0, 256, (1, 3, 64 + 8 * i, 96), dtype=torch.uint8, generator=g).share_memory_() for i in range(7)]
    outs = [to_device(x, "cuda") for x in items]
    torch.cuda.synchronize()
    for x, o in zip(items, outs):
        assert o.is_cuda and o.shape == x.shape and o.dtype == x.dtype and torch.equal(o.cpu(), x)
    y = torch.arange(10, dtype=torch.float32).pin_memory()
    assert torch.equal(to_device(y, "cuda").cpu(), y)
    z = outs[0]
    assert to_device(z, "cuda") is z
    nc = torch.arange(24, dtype=torch.int64).view(4, 6).t()                      # non-contiguous
    assert torch.equal(to_device(nc, torch.device("cuda", 0)).cpu(), nc)


def test_rba_scores_soak_three_streams_graph_replay():
    """the evaluator's default scorer must be bit-stable.  Swin-B 1dl at 1024 x 2048 through MaskFormer.rba_scores with graph
    replay, 3 x 60 forwards alternating over three streams (three graphs replaying concurrently): every score map bit-equal to the first."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import k1_soak
    r = k1_soak.soak_model(forwards=60)
    assert r["live_graphs"] == 3 and r["mismatching_forwards"] == 0, r


def test_concurrent_streams_hint_changes_launch_forms_not_bits(knobs):
    """ops.set_concurrent_streams(n >= 2) makes the half-chip K6 launches use the 256 x 128 / 8-wave form (include/rba_hip.h): the score map of a
    Swin-B forward is bit-identical with and without the hint, and with the 8-wave form switched off altogether (rba_k6_rs = 1)."""
    import ctypes
    from rba_amd import _lib, ops
    model, a, _ = build("swin_b_1dl", 0)
    model.graph_replay = False
    g = torch.Generator().manual_seed(21)
    image = torch.randint(0, 256, (3, 512, 1024), generator=g, dtype=torch.uint8).cuda()
    rs = ctypes.c_int.in_dll(_lib.load(), "rba_k6_rs")
    try:
        ops.set_concurrent_streams(1)
        base = model.rba_scores([{"image": image}])[0].clone()
        assert ops.set_concurrent_streams(3) == 1 and rs.value == 0                    # the hint is its own (unexported) variable: no knob moves
        hinted = model.rba_scores([{"image": image}])[0].clone()
        rs.value = 1
        off = model.rba_scores([{"image": image}])[0].clone()
    finally:
        rs.value = 0
    assert ops.set_concurrent_streams(1) == 3 and ops.concurrent_streams() == 1
    assert torch.equal(base, hinted) and torch.equal(base, off)


def test_model_with_captured_graphs_can_be_deep_copied_and_never_thrashes():
    """hipGraphs cannot be copied -- copy.deepcopy(model) leaves them behind and the copy captures its own; keys that were only
    seen once never evict a captured graph; a model whose image shapes churn faster than graphs are replayed stops capturing."""
    import copy
    model, a, _ = build("tiny3", 0)
    g = torch.Generator().manual_seed(5)
    im = torch.randint(0, 256, (3, 64, 96), generator=g, dtype=torch.uint8).cuda()
    want = model.rba_scores([{"image": im}])[0].clone()
    for _ in range(3):
        assert torch.equal(model.rba_scores([{"image": im}])[0], want)
    assert model.live_graphs() == 1
    for k_ in range(2 * model.GRAPH_MAX):                                  # one-off shapes: placeholders only, the captured graph stays
        model.rba_scores([{"image": torch.randint(0, 256, (3, 32 + 4 * k_, 48), generator=g, dtype=torch.uint8).cuda()}])
    assert model.live_graphs() == 1
    assert torch.equal(model.rba_scores([{"image": im}])[0], want)
    twin = copy.deepcopy(model)
    assert twin.live_graphs() == 0 and model.live_graphs() == 1
    for _ in range(3):
        assert torch.equal(twin.rba_scores([{"image": im}])[0], want)
    assert twin.live_graphs() == 1
    # churn: every shape comes exactly twice (captured, replayed once, evicted before it pays) -> the model gives up capturing
    churn, _, _ = build("tiny3", 0)
    churn.GRAPH_MAX = 2
    shapes = [(32 + 4 * k_, 64) for k_ in range(2 * churn.GRAPH_MAX + churn.GRAPH_THRASH_MAX + 3)]
    for hw in shapes:
        x = torch.randint(0, 256, (3,) + hw, generator=g, dtype=torch.uint8).cuda()
        churn.rba_scores([{"image": x}])
        churn.rba_scores([{"image": x}])
    assert churn.__dict__.get("_graph_thrash", 0) >= churn.GRAPH_THRASH_MAX and churn.live_graphs() <= churn.GRAPH_MAX
    n_before = churn.live_graphs()
    x = torch.randint(0, 256, (3, 200, 64), generator=g, dtype=torch.uint8).cuda()
    churn.rba_scores([{"image": x}]); churn.rba_scores([{"image": x}]); churn.rba_scores([{"image": x}])
    assert churn.live_graphs() == n_before                                  # no new capture any more
    churn.drop_graphs()
    assert churn.__dict__.get("_graph_thrash", 0) == 0


def test_model_zoo_check_tool_end_to_end(tmp_path, monkeypatch, capsys):
    """tools/model_zoo_check.py on a models folder laid out as MODEL_ZOO.md:8-19 -- `swin_b_1dl/` holding Detectron2's model_final.pth (model + optimizer +
    scheduler + iteration), `swin_l_1dl/` holding only a model-zoo style model_final.pkl (numpy arrays) -- and synthetic RoadAnomaly / Fishyscapes-LaF trees: the
    tool runs the evaluator, reads results.pkl and compares with a table (here: the oracle's CPU forward + scikit-learn on the same files, in percent) at the
    default 0.005 percentage points; a table that is off by 0.1 fails with exit status 1.  With the released weights the same command checks MODEL_ZOO.md itself."""
    import json
    import sys
    import yaml
    from oracle import ref_metrics
    from tests.test_datasets_cpu import make_fs_laf, make_road_anomaly
    from tests.test_host_cpu import _write_d2_checkpoints
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import model_zoo_check as Z
    a = A.complete(A.ARCHS["tiny1"])
    cfg = {"MODEL": {"SWIN": {"EMBED_DIM": 32, "DEPTHS": [2, 2, 2, 2], "NUM_HEADS": [1, 2, 4, 8], "WINDOW_SIZE": 6},
                     "SEM_SEG_HEAD": {"CONVS_DIM": 64, "MASK_DIM": 64, "TRANSFORMER_ENC_LAYERS": 2, "DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES": ["res5"]},
                     "MASK_FORMER": {"HIDDEN_DIM": 64, "NHEADS": 2, "NUM_OBJECT_QUERIES": 16, "DIM_FEEDFORWARD": 128, "DEC_LAYERS": 2}}}
    data = tmp_path / "data"
    ra_imgs, ra_labs = make_road_anomaly(str(data), h=48, w=80)
    fs_imgs, fs_labs = make_fs_laf(str(data), h=40, w=72)
    expected = {}
    for model_name, seed in (("swin_b_1dl", 0), ("swin_l_1dl", 1)):
        sd = A.seeded_weights(a, seed)
        mdir = tmp_path / "ckpts" / model_name
        _write_d2_checkpoints(str(mdir), sd)
        (mdir / "config.yaml").write_text(yaml.safe_dump(cfg))
        os.remove(mdir / "tensors.pkl")
        if model_name == "swin_l_1dl":
            os.remove(mdir / "model_final.pth")
            os.rename(mdir / "numpy.pkl", mdir / "model_final.pkl")
        expected[model_name] = {}
        for name, imgs, labs in (("road_anomaly", ra_imgs, [(l == 2).astype(np.int64) for l in ra_labs]), ("fishyscapes_laf", fs_imgs, [l.astype(np.int64) for l in fs_labs])):
            scores = np.stack([ref_model.forward(torch.from_numpy(im.transpose(2, 0, 1).copy()), sd, a)["rba"].numpy() for im in imgs])
            m = ref_metrics.evaluate_ood(scores, np.stack(labs)[:, None])
            expected[model_name][name] = {"aupr": 100.0 * m["aupr"], "fpr95": 100.0 * m["fpr95"]}
    (tmp_path / "expected.json").write_text(json.dumps(expected))
    monkeypatch.chdir(tmp_path)
    argv = ["--models_folder", str(tmp_path / "ckpts"), "--datasets_folder", str(data), "--out_path", str(tmp_path / "zoo"), "--expected", str(tmp_path / "expected.json")]
    assert Z.main(argv) == 0
    txt = capsys.readouterr().out
    assert "8 of 8 published numbers reproduced within 0.005" in txt, txt
    expected["swin_l_1dl"]["road_anomaly"]["aupr"] += 0.1
    (tmp_path / "expected.json").write_text(json.dumps(expected))
    assert Z.main(argv + ["--results-only"]) == 1
    assert "DIFFERS" in capsys.readouterr().out


def test_graph_replay_policy_is_measured():
    """the product default must not be the slower path.  graph_replay = "auto": call 1 of a shape is eager, call 2 is eager between HIP
    events with the host's issue time beside them, call 3 decides -- capture only when host issue time >= LAUNCH_BOUND_RATIO x GPU span; a GPU-bound shape stays
    eager and is measured again every GRAPH_REMEASURE_EVERY calls.  Forced both ways through the ratio; every path returns the same bits."""
    from rba_amd.checkpoint import load_checkpoint
    from rba_amd.maskformer_model import MaskFormer
    a = A.complete(A.ARCHS["tiny3"])
    model = load_checkpoint(MaskFormer(a), A.seeded_weights(a, 0)).cuda().eval()
    assert model.graph_replay == "auto"
    g = torch.Generator().manual_seed(11)
    im = torch.randint(0, 256, (3, 64, 96), generator=g, dtype=torch.uint8).cuda()
    model.graph_replay = False
    want = model.rba_scores([{"image": im}])[0].clone()
    model.graph_replay = "auto"
    model.LAUNCH_BOUND_RATIO = 1e9                                   # nothing is launch-bound: stay eager whatever the host does
    model.GRAPH_REMEASURE_EVERY = 3
    for _ in range(12):
        assert torch.equal(model.rba_scores([{"image": im}])[0], want)
    assert model.live_graphs() == 0
    (dec,) = model.graph_decisions().values()
    assert dec["decision"] == "eager" and dec["host_issue_ms"] > 0 and dec["gpu_span_ms"] > 0
    model.LAUNCH_BOUND_RATIO = 0.0                                   # everything is launch-bound: the next measurement captures
    for _ in range(8):
        assert torch.equal(model.rba_scores([{"image": im}])[0], want)
    assert model.live_graphs() == 1
    (dec,) = model.graph_decisions().values()
    assert dec["decision"] == "replay"
    # the real ratio on a tiny net: ~300 launches of a few microseconds each -- launch-bound on any host
    fresh = load_checkpoint(MaskFormer(a), A.seeded_weights(a, 0)).cuda().eval()
    for _ in range(5):
        assert torch.equal(fresh.rba_scores([{"image": im}])[0], want)
    (dec,) = fresh.graph_decisions().values()
    # ~300 launches of a few microseconds each: launch-bound on every host seen so far (ratio 0.98-1.0) -- but the DECISION is a measurement, so the test holds the
    # model to its own reading rather than to a timing: a graph exists exactly when the measured ratio said launch-bound
    assert dec["host_issue_ms"] > 0 and dec["gpu_span_ms"] > 0
    assert fresh.live_graphs() == (1 if dec["decision"] == "replay" else 0), dec
    assert (dec["decision"] == "replay") == (dec["host_issue_ms"] >= fresh.LAUNCH_BOUND_RATIO * dec["gpu_span_ms"]), dec
