"""End-to-end evaluator throughput: `python -m rba_amd.evaluate_ood` (the reference CLI, README.md:60-69) on an on-disk synthetic
Fishyscapes-LostAndFound-shaped set (1024x2048 PNG images + label PNGs), Swin-B 1dl with seeded weights saved as model_final.pth.
Reports images/s of the scoring loop (decode threads -> H2D -> forward -> K1 -> labelled-pixel selection) for
the pipelined loop and for the reference-like serial loop (--num_workers 0 --streams 1).
    python tools/evaluator_bench.py [n_images] [workdir] [case,case,...]"""
import json
import os
import pickle
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch


def make_dataset(root, n, h=1024, w=2048):
    """Street-scene-like smooth images (real photographs compress about 2:1 as PNG; pure noise would not compress at all)"""
    from PIL import Image
    lab_dir, img_dir = os.path.join(root, "Fishyscapes", "fishyscapes_lostandfound"), os.path.join(root, "Fishyscapes", "laf_images")
    os.makedirs(lab_dir, exist_ok=True)
    os.makedirs(img_dir, exist_ok=True)
    rng = np.random.RandomState(0)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for i in range(n):
        base = np.stack([128 + 100 * np.sin(xx / (40 + 7 * i + 13 * c) + c) * np.cos(yy / (55 + 5 * i)) for c in range(3)], -1)
        img = np.clip(base + rng.randn(h, w, 3) * 6, 0, 255).astype(np.uint8)
        lab = np.zeros((h, w), np.uint8)
        lab[h // 2 - 60: h // 2 + 60, (100 + 37 * i) % (w - 300): (100 + 37 * i) % (w - 300) + 200] = 1
        lab[:16] = 255
        name = f"{i:04d}_04_Maurener_Weg_8_000000_{i:06d}_"
        Image.fromarray(img).save(os.path.join(img_dir, name[5:] + "leftImg8bit.png"), compress_level=1)
        Image.fromarray(lab).save(os.path.join(lab_dir, name + "labels.png"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/rba_eval_bench"
    import yaml
    from rba_amd import arch as A
    from rba_amd import evaluate_ood as E
    t0 = time.perf_counter()
    make_dataset(os.path.join(work, "data"), n)
    t_data = time.perf_counter() - t0
    mdir = os.path.join(work, "ckpts", "swin_b_1dl")
    os.makedirs(mdir, exist_ok=True)
    cfg = {"MODEL": {"META_ARCHITECTURE": "MaskFormer", "BACKBONE": {"NAME": "D2SwinTransformer"},
                     "SWIN": {"EMBED_DIM": 128, "DEPTHS": [2, 2, 18, 2], "NUM_HEADS": [4, 8, 16, 32], "WINDOW_SIZE": 12},
                     "SEM_SEG_HEAD": {"DEFORMABLE_TRANSFORMER_ENCODER_IN_FEATURES": ["res5"]},
                     "MASK_FORMER": {"DEC_LAYERS": 2}}}
    with open(os.path.join(mdir, "config.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    a = A.complete(A.ARCHS["swin_b_1dl"])
    torch.save({"model": A.seeded_weights(a, 0)}, os.path.join(mdir, "model_final.pth"))
    res = {"n_images": n, "image": "1024x2048 PNG", "dataset_write_s": round(t_data, 1)}
    only = set(sys.argv[3].split(",")) if len(sys.argv) > 3 else None          # optional: run just these cases
    for tag, extra in (("default_graph_8workers_3streams", []),
                       ("graph_16workers_3streams", ["--num_workers", "16"]),
                       ("graph_12workers_2streams", ["--num_workers", "12", "--streams", "2"]),
                       ("graph_4workers_3streams", ["--num_workers", "4"]),
                       ("eager_8workers_3streams", ["--graph", "0"]),
                       ("eager_4workers_3streams", ["--graph", "0", "--num_workers", "4"]),
                       ("eager_16workers_3streams", ["--graph", "0", "--num_workers", "16"]),
                       ("graph_10workers_3streams", ["--num_workers", "10"]),
                       ("graph_12workers_3streams", ["--num_workers", "12"]),
                       ("graph_6workers_3streams", ["--num_workers", "6"]),
                       ("graph_8processes_3streams", ["--loader", "processes"]),
                       ("graph_12processes_3streams", ["--loader", "processes", "--num_workers", "12"]),
                       ("graph_16processes_3streams", ["--loader", "processes", "--num_workers", "16"]),
                       ("graph_24processes_3streams", ["--loader", "processes", "--num_workers", "24"]),
                       ("serial_like_reference_loop", ["--num_workers", "0", "--streams", "1", "--graph", "0"])):
        if only is not None and tag not in only:
            continue
        out = os.path.join(work, "results_" + tag)
        timing = {}
        args = E.build_parser().parse_args(["--models_folder", os.path.join(work, "ckpts"), "--datasets_folder", os.path.join(work, "data"),
                                            "--dataset_mode", "selective", "--selected_datasets", "fishyscapes_laf", "--out_path", out,
                                            "--verbose", "0"] + extra)
        args.decoder = E.open_decoder(args)                      # as main() does: before the model is loaded
        model = E.get_model(os.path.join(mdir, "config.yaml"), os.path.join(mdir, "model_final.pth"))
        from rba_amd.datasets import get_dataset
        ds = get_dataset("fishyscapes_laf", args.datasets_folder)
        E.run_evaluations(model, torch.utils.data.Subset(ds, [0, 1]), "warm", "fishyscapes_laf", args)       # warm-up: plans, weight planes
        m = E.run_evaluations(model, ds, "swin_b_1dl", "fishyscapes_laf", args, timing=timing)
        res[tag] = {"images_per_s": round(timing["images_per_s"], 2), "seconds": round(timing["seconds"], 3), "metrics": m,
                    "num_workers": timing["num_workers"], "loader": timing.get("loader"), "streams": timing["streams"], "host_thread": timing.get("host_thread"), "hip_graphs": timing.get("hip_graphs")}
        if "decode_processes_ms_per_sample" in timing:
            res[tag]["decode_processes_ms_per_sample"] = timing["decode_processes_ms_per_sample"]
        if args.decoder is not None:
            args.decoder.close()
        del model
        torch.cuda.empty_cache()
    # ---- the reference's OWN loop with the drop-in classes: evaluate_ood.py:205-235 builds
    # DataLoader(dataset, batch_size, num_workers=15) and calls OODEvaluator.compute_anomaly_scores + evaluate_ood -- batch 1, one
    # stream, every score leaves the GPU.  What that loop gets from rba_amd without any change of its own: hipGraph replay inside
    # MaskFormer.rba_scores, device -> host copies through the pinned ring, no per-image stream stall.
    from torch.utils.data import DataLoader
    from rba_amd.datasets import ThreadLoader, get_dataset
    from rba_amd.support import OODEvaluator
    ds = get_dataset("fishyscapes_laf", os.path.join(work, "data"))

    def thread_loader(nthreads):
        return ThreadLoader(ds, nthreads)

    for tag, make_loader, replay in (("reference_loop_dataloader_15_workers", lambda: DataLoader(ds, shuffle=False, batch_size=1, num_workers=15, timeout=300), True),
                                     ("reference_loop_dataloader_15_workers_no_graph", lambda: DataLoader(ds, shuffle=False, batch_size=1, num_workers=15, timeout=300), False),
                                     ("reference_loop_thread_loader_8", lambda: thread_loader(8), True),
                                     ("reference_loop_no_workers", lambda: DataLoader(ds, shuffle=False, batch_size=1, num_workers=0), True)):
        if only is not None and tag not in only:
            continue
        try:
            model = E.get_model(os.path.join(mdir, "config.yaml"), os.path.join(mdir, "model_final.pth"))
            model.graph_replay = replay
            ev = OODEvaluator(model, E.get_logits, E.get_RbA)
            warm = [(ds[i][0][None], ds[i][1][None]) for i in range(3)]
            ev.compute_anomaly_scores(warm, device=torch.device("cuda"))             # plans, weight planes, graph capture
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            scores, gts = ev.compute_anomaly_scores(make_loader(), device=torch.device("cuda"), upper_limit=n)
            t1 = time.perf_counter()
            m = ev.evaluate_ood(scores, gts, verbose=False)
            t2 = time.perf_counter()
            res[tag] = {"images_per_s_scoring_loop": round(len(scores) / (t1 - t0), 2), "images_per_s_with_metrics": round(len(scores) / (t2 - t0), 2),
                        "seconds_loop": round(t1 - t0, 3), "seconds_metrics": round(t2 - t1, 3), "metrics": m, "hip_graphs": model.live_graphs()}
            del model, ev, scores, gts
            torch.cuda.empty_cache()
        except Exception as e:                                                       # noqa: BLE001 -- a bench tool: record and go on
            res[tag] = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
