"""Accuracy parity on the RELEASED weights, as one command (VERDICT round 5 "next" #2).

    python tools/model_zoo_check.py --models_folder ckpts/ --datasets_folder /data/ood [--gpus 8] [--tolerance 0.005]

`ckpts/` is the reference's layout (MODEL_ZOO.md:8-19): one folder per model holding `config.yaml` (shipped in the reference repository) and the released
`model_final.pth` (github release `model-weights`: swin_b_1dl.zip, swin_l_1dl.zip, swin_b_1dl_rba_ood_coco.zip, swin_l_1dl_rba_ood_map_coco.zip, ...).
`--datasets_folder` holds `RoadAnomaly/RoadAnomaly_jpg` and `Fishyscapes` as the reference's datasets/README.md:91-122 lays them out.  No weights and no
datasets are in this image (no network): the tool is exercised end to end on synthetic checkpoints / datasets by tests/ (a CPU test of the table, the
checkpoint formats and the comparison; a GPU test of the whole run), and is ready for the day the files exist.

What it does: for every model folder that has a row in MODEL_ZOO (below -- the reference's MODEL_ZOO.md tables as data, in the published unit: percent, two
decimals) it runs `python -m rba_amd.evaluate_ood` (the reference CLI: same flags, same results/<model>/results.pkl; under torch.distributed.run when --gpus > 1),
reads the pooled AuPRC / FPR95 and prints one line per (model, dataset, metric): published, measured, difference, verdict.  Exit status 1 when any |difference|
exceeds --tolerance (percentage points; default 0.005 = "rounds to the published number") -- `--tolerance 0.05` is BASELINE.json's "three decimals" of the
fraction.  `--results-only` compares results.pkl files that already exist (e.g. written by the reference itself on another machine).

Known sources of a legitimate last-digit difference: JPEG decode (PIL here, cv2 in the reference: +-1 LSB on some pixels), fp32 summation
order (|d rba| <= 3e-5 measured on the synthetic fixtures; metric differences <= 5.3e-6 there).
"""
import argparse
import json
import os
import pickle
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# /root/reference/MODEL_ZOO.md:54-148 -- AP = results["aupr"] * 100, FPR95 = results["fpr95"] * 100 (evaluate_ood.py:175-185 stores fractions).
# (The Swin-L Cityscapes-only row's FS-LaF FPR95 of 71.79 is what the table says.)
MODEL_ZOO = {
    "swin_b_1dl": {"road_anomaly": {"aupr": 78.45, "fpr95": 11.83}, "fishyscapes_laf": {"aupr": 60.96, "fpr95": 10.63}},               # MODEL_ZOO.md:54-62
    "swin_l_1dl": {"road_anomaly": {"aupr": 79.68, "fpr95": 15.02}, "fishyscapes_laf": {"aupr": 58.61, "fpr95": 71.79}},               # :64-72
    "swin_b_1dl_rba_ood_coco": {"road_anomaly": {"aupr": 85.42, "fpr95": 6.92}, "fishyscapes_laf": {"aupr": 70.81, "fpr95": 6.30}},     # :97-106
    "swin_b_1dl_rba_ood_map_coco": {"road_anomaly": {"aupr": 89.16, "fpr95": 4.50}, "fishyscapes_laf": {"aupr": 78.27, "fpr95": 3.98}},  # :131-140
    "swin_l_1dl_rba_ood_map_coco": {"road_anomaly": {"aupr": 90.28, "fpr95": 4.92}, "fishyscapes_laf": {"aupr": 80.35, "fpr95": 4.58}},  # :142-150
}
DATASETS = ("road_anomaly", "fishyscapes_laf")


def parse(argv=None):
    p = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    p.add_argument("--models_folder", default="ckpts/")
    p.add_argument("--datasets_folder", default="./")
    p.add_argument("--out_path", default="results_model_zoo", help="where results/<model>/results.pkl go (existing files are reused, as in the reference)")
    p.add_argument("--tolerance", type=float, default=0.005, help="allowed |measured - published| in percentage points")
    p.add_argument("--expected", default=None, help="JSON file {model: {dataset: {aupr, fpr95}}} in percent replacing the built-in MODEL_ZOO table "
                   "(tests; other checkpoints whose reference numbers you hold)")
    p.add_argument("--selected_models", nargs="*", default=None)
    p.add_argument("--selected_datasets", nargs="*", default=list(DATASETS))
    p.add_argument("--gpus", type=int, default=1, help="> 1: the evaluator runs under torch.distributed.run, images sharded over the ranks, metrics pooled by RCCL")
    p.add_argument("--results-only", action="store_true", help="do not run the evaluator: compare the results.pkl files under --out_path")
    p.add_argument("--dry-run", action="store_true", help="list what would be evaluated and compared, then stop (needs no GPU)")
    p.add_argument("--evaluator-args", nargs=argparse.REMAINDER, default=[], help="everything after this flag goes to rba_amd.evaluate_ood verbatim")
    return p.parse_args(argv)


def expected_table(args):
    if args.expected:
        with open(args.expected) as f:
            return json.load(f)
    return MODEL_ZOO


def plan(args):
    """[(model, checkpoint path | None)] for the folders under --models_folder that have a row in the table"""
    table = expected_table(args)
    if not os.path.isdir(args.models_folder):
        raise SystemExit(f"--models_folder {args.models_folder!r} is not a directory")
    out = []
    for m in sorted(os.listdir(args.models_folder)):
        d = os.path.join(args.models_folder, m)
        if not os.path.isdir(d) or m not in table or (args.selected_models is not None and m not in args.selected_models):
            continue
        ck = next((os.path.join(d, f) for f in ("model_final.pth", "model_final.pkl") if os.path.exists(os.path.join(d, f))), None)
        out.append((m, ck, os.path.exists(os.path.join(d, "config.yaml"))))
    return out


def run_evaluator(args, models):
    ev = ["--models_folder", args.models_folder, "--datasets_folder", args.datasets_folder, "--out_path", args.out_path, "--model_mode", "selective",
          "--selected_models", *models, "--dataset_mode", "selective", "--selected_datasets", *args.selected_datasets, "--verbose", "false",
          *args.evaluator_args]
    if args.gpus > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(29500 + os.getpid() % 2000), "-m", "rba_amd.evaluate_ood", *ev]
        env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
        subprocess.run(cmd, check=True, cwd=REPO, env=env)
    else:
        from rba_amd import evaluate_ood as E
        E.main(ev)


def compare(args, models):
    """-> (rows, worst): rows = [(model, dataset, metric, published, measured | None, diff | None, ok)]"""
    table = expected_table(args)
    rows, worst = [], 0.0
    for m in models:
        path = os.path.join(args.out_path, m, "results.pkl")
        res = None
        if os.path.exists(path):
            with open(path, "rb") as f:
                res = pickle.load(f)
        for ds in args.selected_datasets:
            for metric, pub in sorted(table[m].get(ds, {}).items()):
                if res is None or ds not in res or metric not in res[ds]:
                    rows.append((m, ds, metric, pub, None, None, False))
                    continue
                got = 100.0 * float(res[ds][metric])
                diff = got - pub
                worst = max(worst, abs(diff))
                rows.append((m, ds, metric, pub, got, diff, abs(diff) <= args.tolerance + 1e-12))
    return rows, worst


def report(rows, tolerance, file=sys.stdout):
    name = {"aupr": "AP", "fpr95": "FPR95", "auroc": "AUROC"}
    print(f"{'model':34s} {'dataset':16s} {'metric':6s} {'published':>9s} {'measured':>10s} {'diff':>9s}  verdict (tolerance {tolerance} pp)", file=file)
    for m, ds, metric, pub, got, diff, ok in rows:
        if got is None:
            print(f"{m:34s} {ds:16s} {name.get(metric, metric):6s} {pub:9.2f} {'-':>10s} {'-':>9s}  MISSING (no results.pkl entry)", file=file)
        else:
            three = "" if ok else ("  [inside 0.05 pp = three decimals of the fraction]" if abs(diff) <= 0.05 else "")
            print(f"{m:34s} {ds:16s} {name.get(metric, metric):6s} {pub:9.2f} {got:10.4f} {diff:+9.4f}  {'ok' if ok else 'DIFFERS'}{three}", file=file)


def main(argv=None):
    args = parse(argv)
    todo = plan(args)
    if not todo:
        print(f"no folder under {args.models_folder} has a row in the table ({sorted(expected_table(args))})")
        return 2
    for m, ck, has_cfg in todo:
        print(f"[model_zoo_check] {m}: config.yaml {'found' if has_cfg else 'MISSING'}, checkpoint {ck or 'MISSING (model_final.pth / .pkl)'}")
    if args.dry_run:
        return 0
    runnable = [m for m, ck, has_cfg in todo if ck and has_cfg]
    if not args.results_only and runnable:
        run_evaluator(args, runnable)
    rows, worst = compare(args, [m for m, _, _ in todo])
    report(rows, args.tolerance)
    bad = [r for r in rows if not r[6]]
    print(f"[model_zoo_check] {len(rows) - len(bad)} of {len(rows)} published numbers reproduced within {args.tolerance} percentage points; "
          f"largest |difference| {worst:.4f}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
