"""The evaluator-facing functions of the reference's ``evaluate_ood.py`` with the same names, arguments and
outputs: ``get_model`` (:108-124), ``get_logits`` (:127-140), ``get_RbA`` (:143-150), ``get_energy`` (:152-159), and its
command line (``python -m rba_amd.evaluate_ood --models_folder ckpts/ --datasets_folder ... --score_func rba``, same flags,
same ``results/<model>/results.pkl``), extended with image sharding when launched through torchrun."""
import argparse
import sys
import os
import pickle
from pathlib import Path

import numpy as np
import torch

from .arch import arch_from_cfg
from .checkpoint import load_checkpoint
from .config import load_cfg
from .registry import META_ARCH_REGISTRY
from . import maskformer_model as _mm  # noqa: F401  (registers MaskFormer)

DEVICE = torch.device("cuda")


def build_model(cfg):
    """Trainer.build_model(cfg) (train_net.py:75-80): META_ARCH_REGISTRY[cfg.MODEL.META_ARCHITECTURE](...)."""
    cls = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)
    return cls(arch_from_cfg(cfg), backbone_name=cfg.MODEL.BACKBONE.NAME, head_name=cfg.MODEL.SEM_SEG_HEAD.NAME)


def get_model(config_path, model_path, device=None):
    """Creates the model from a config path and a checkpoint path (``model_path`` may be None for random init)."""
    cfg = load_cfg(config_path, ["OUTPUT_DIR", "output/"])
    model = build_model(cfg)
    if model_path is not None:
        # default: a checkpoint file cannot execute code (weights_only / numpy-only unpickler); opt in for trusted files
        load_checkpoint(model, model_path, trusted=os.environ.get("RBA_TRUSTED_CHECKPOINT") == "1")
    model.to(device or DEVICE)
    return model.eval()


def get_logits(model, x, **kwargs):
    """x [1,3,H,W] -> logits [1,K,H,W]."""
    with torch.no_grad():
        out = model([{"image": x[0].to(model.device)}])
    return out[0]["sem_seg"].unsqueeze(0)


def get_RbA(model, x, **kwargs):
    """x [1,3,H,W] -> RbA score [H,W] = -sum_k tanh(sem_seg_k)."""
    if hasattr(model, "rba_scores"):
        return model.rba_scores([{"image": x[0].to(model.device)}])[0]
    with torch.no_grad():
        out = model([{"image": x[0].to(model.device)}])
    return -out[0]["sem_seg"].tanh().sum(dim=0)


def get_energy(model, x, **kwargs):
    """-logsumexp_k(sem_seg) (evaluate_ood.py:152-159)."""
    if hasattr(model, "rba_scores"):
        return model.rba_scores([{"image": x[0].to(model.device)}], score="energy")[0]
    with torch.no_grad():
        out = model([{"image": x[0].to(model.device)}])
    return -torch.logsumexp(out[0]["sem_seg"], dim=0)


def get_neg_logit_sum(model, x, **kwargs):
    """-sum_k sem_seg (support.py:115-132)."""
    if hasattr(model, "rba_scores"):
        return model.rba_scores([{"image": x[0].to(model.device)}], score="neg_logit_sum")[0]
    with torch.no_grad():
        out = model([{"image": x[0].to(model.device)}])
    return -out[0]["sem_seg"].sum(dim=0)


def get_densehybrid_score(model, x, **kwargs):
    """-logsumexp_k(sem_seg) + log(softmax(ood_pred)[:, 1] + 1e-9) -> [1,H,W] (evaluate_ood.py:161-173)."""
    with torch.no_grad():
        out, ood_pred = model([{"image": x[0].to(model.device)}], return_ood_pred=True)
    p1 = torch.logsumexp(out[0]["sem_seg"], dim=0)
    p2 = torch.softmax(ood_pred, dim=1)[:, 1]                      # p(~din | x)
    return (-p1) + (p2 + 1e-9).log()


# K1 epilogue each score function selects: lets OODEvaluator take score + argmax from ONE forward (support.py) without
# guessing from the function's name
get_RbA.rba_score_mode = "rba"
get_energy.rba_score_mode = "energy"
get_neg_logit_sum.rba_score_mode = "neg_logit_sum"


# ------------------------------------------------------------------------------------------------- command line
SCORE_FUNCS = {"rba": get_RbA, "pebal": get_energy, "energy": get_energy, "neg_logit_sum": get_neg_logit_sum,
               "dense_hybrid": get_densehybrid_score}


def build_parser():
    p = argparse.ArgumentParser(description="OOD Evaluation (rba_amd)")
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--num_workers", type=int, default=8, help="image-decoding threads / processes (the reference: DataLoader workers)")
    p.add_argument("--loader", choices=("threads", "processes"), default="threads",
                   help="decode in threads of this process, or in `--num_workers` child processes (rba_amd._decode_worker: numpy + Pillow "
                        "only, started with subprocess -- not forks of this process); same samples, same order")
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--out_path", type=str, default="results")
    p.add_argument("--verbose", type=lambda v: str(v).lower() not in ("0", "false", "no", ""), default=True)
    p.add_argument("--datasets_folder", type=str, default="./")
    p.add_argument("--models_folder", type=str, default="ckpts/")
    p.add_argument("--store_anomaly_scores", action="store_true",
                   help="store every score map as anomaly_scores/<model>/<dataset>/score_<i>.npy")
    p.add_argument("--model_mode", type=str, default="all", choices=["all", "selective"])
    p.add_argument("--selected_models", nargs="*", type=str, default=[])
    p.add_argument("--dataset_mode", type=str, default="all", choices=["all", "selective"])
    p.add_argument("--selected_datasets", nargs="*", type=str, default=[])
    p.add_argument("--score_func", type=str, default="rba", choices=sorted(SCORE_FUNCS))
    p.add_argument("--upper_limit", type=int, default=1300)
    p.add_argument("--streams", type=int, default=3, help="HIP streams the batch-1 forwards alternate on (1 = the reference's serial loop)")
    p.add_argument("--graph", type=int, default=1,
                   help="1: replay the forward of each (stream, image shape) from a captured hipGraph -- the scoring thread then issues one "
                        "launch per image instead of ~370 (it is the bottleneck beside the decode threads); falls back to eager launches "
                        "if the capture fails")
    p.add_argument("--dist_backend", type=str, default=os.environ.get("RBA_EVAL_BACKEND") or None, choices=[None, "nccl", "gloo"],
                   help="torch.distributed backend under torchrun (default: nccl = RCCL when a HIP device is visible, else gloo)")
    p.add_argument("--share_device", type=int, default=int(os.environ.get("RBA_EVAL_SHARE_DEVICE", "0")),
                   help="1: every rank uses device 0 and the metric exchange runs on gloo -- a plumbing test of the sharded evaluator on a "
                        "one-GPU box (RCCL refuses two ranks on one device), never a measurement")
    return p


class GraphedScore:
    """score_func(model, x[None]) replayed from one captured hipGraph per image shape, for ONE stream.  The first call of a shape
    runs eagerly (lazy per-shape state of the model), the second captures, later calls copy the image into the static input and
    replay.  The returned map is a fresh tensor (the static output is overwritten by the next replay)."""

    MAX_SHAPES = 4

    def __init__(self, model, score_func, stream):
        self.model, self.score_func, self.stream = model, score_func, stream
        self.graphs = {}             # shape -> None (seen once) | (graph, static_in, static_out) | False (capture failed: eager)
        # score functions that end in MaskFormer.rba_scores are replayed by the model itself (one graph per image shape and stream,
        # MaskFormer._graphed_scores): nothing to do here.  The others (DenseHybrid: forward(return_ood_pred=True) + torch ops) are
        # captured whole, below, on a capture stream of this object's own (per-stream kernel state must not be shared between the
        # graphs of different streams, which replay concurrently).
        self.delegate = (getattr(score_func, "rba_score_mode", None) is not None and hasattr(model, "_graphed_scores")
                         and getattr(model, "graph_replay", False))
        self.capture_stream = None

    def __call__(self, x):
        if self.delegate:
            return self.score_func(self.model, x[None])
        key = tuple(x.shape)
        entry = self.graphs.get(key, "new")
        if entry == "new":
            # every graph keeps its own activation pool alive: at most MAX_SHAPES image shapes are captured per stream
            self.graphs[key] = None if sum(1 for e in self.graphs.values() if e is not False) < self.MAX_SHAPES else False
            return self.score_func(self.model, x[None])
        if entry is None:
            try:
                static_in = x.clone()
                g = torch.cuda.CUDAGraph()
                if self.capture_stream is None:
                    self.capture_stream = torch.cuda.Stream(device=x.device)
                self.capture_stream.wait_stream(torch.cuda.current_stream(x.device))
                # captured on this object's own stream, replayed on self.stream; thread_local: the decode threads pin memory meanwhile
                with torch.cuda.graph(g, stream=self.capture_stream, capture_error_mode="thread_local"):
                    static_out = self.score_func(self.model, static_in[None])
                torch.cuda.current_stream(x.device).wait_stream(self.capture_stream)
                entry = self.graphs[key] = (g, static_in, static_out)
            except Exception as e:                                   # noqa: BLE001 -- an optimisation only
                print(f"[rba_amd] hipGraph capture failed for shape {key} ({type(e).__name__}: {e}); eager launches", file=sys.stderr)
                torch.cuda.synchronize()
                entry = self.graphs[key] = False
        if entry is False:
            return self.score_func(self.model, x[None])
        g, static_in, static_out = entry
        static_in.copy_(x, non_blocking=True)
        g.replay()
        return static_out.clone()


def run_evaluations(model, dataset, model_name, dataset_name, args, rank=0, world=1, timing=None):
    """(wrapper) the launch-geometry hint and the model's graph-replay flag are restored whatever happens inside the loop -- an exception must not leave the
    process-wide hint of rba_set_concurrent_streams at the evaluator's value."""
    from . import ops as _ops
    dev = next(model.parameters()).device
    on_gpu = dev.type == "cuda"
    prev_streams = _ops.set_concurrent_streams(max(1, int(getattr(args, "streams", 3)))) if on_gpu else None   # speed only, bit-identical results
    prev_replay = getattr(model, "graph_replay", None)
    try:
        return _run_evaluations(model, dataset, model_name, dataset_name, args, rank, world, timing)
    finally:
        if prev_replay is not None:
            model.graph_replay = prev_replay
        if prev_streams is not None:
            _ops.set_concurrent_streams(prev_streams)


def _run_evaluations(model, dataset, model_name, dataset_name, args, rank=0, world=1, timing=None):
    """Score this rank's shard of `dataset` (image i -> rank i mod world), pool the labelled pixels of all ranks over
    RCCL and return {"auroc","aupr","fpr95"} (reference :195-235; single process there).

    Pipeline (the reference feeds its loop from DataLoader(batch_size, num_workers), :210-211): images are decoded ahead of the GPU
    by `--num_workers` threads (datasets.prefetch), copied to the GPU and scored batch-1 on alternating HIP streams (`--streams`,
    default 3) so that the host-side launch work and the under-occupied kernels of one image overlap the other image; score maps and
    labels never leave the GPU (the metric is a GPU sort).  `--graph 1` replays each stream's forward from a captured hipGraph: the
    scoring thread, which shares the interpreter lock with the decode threads, is otherwise the bottleneck (tools/host_overhead.py)."""
    import time
    from . import distributed as D
    from .datasets import prefetch
    from .metrics import select_labelled
    score_func = SCORE_FUNCS[args.score_func]
    n = min(len(dataset), args.upper_limit)
    mine = D.shard_indices(n, rank, world)
    dev = model.device
    on_gpu = dev.type == "cuda"
    nw = max(0, int(args.num_workers))
    # batch-1 items.  On a HIP device the decode threads hand the image over as the decoder left it (uint8 [1,H,W,3]; the transpose to [3,H,W]
    # is one 6 MB copy on the GPU instead of a strided host pass per image) with uint8 labels; on the CPU path [1,3,H,W] as the reference yields
    from .datasets import decode_spec_of
    use_procs = getattr(args, "loader", "threads") == "processes" and on_gpu and nw > 0 and decode_spec_of(dataset) is not None
    proc_stats = {}
    if use_procs:
        def _from_processes():
            # main() starts the children once, before the first model is loaded (they import while it loads); a direct caller without
            # `args.decoder` pays their start-up (~0.25 s) inside its first loop
            pd = getattr(args, "decoder", None)
            own = pd is None
            if own:
                pd = open_decoder(args)
            before = dict(pd.stats)
            try:
                yield from pd.items(dataset, mine, pin=True)
            finally:
                proc_stats.update({n: v - before[n] for n, v in pd.stats.items()})
                if own:                                     # children ended, /dev/shm directory removed -- also when the loop raises
                    pd.close()
        items = _from_processes()
    else:
        items = prefetch(dataset, mine, nw, pin=on_gpu and nw > 0, label_dtype=torch.uint8, raw=on_gpu)
    loader = (tuple(t[None] for t in item) for item in items)
    n_streams = max(1, int(getattr(args, "streams", 3))) if on_gpu else 1
    main_stream = torch.cuda.current_stream(dev) if on_gpu else None
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)] if n_streams > 1 else [main_stream]
    use_graphs = on_gpu and bool(int(getattr(args, "graph", 0)))
    prev_replay = getattr(model, "graph_replay", None)       # (the wrapper restores it, and the stream hint it set)
    if prev_replay is not None:
        # MaskFormer.rba_scores replays per (image shape, stream).  Several streams fed by this one Python thread are launch-bound by construction (the thread
        # issues S forwards per GPU-forward time): always replay.  One stream: the model's measured policy ("auto": replay only where the eager launches of
        # the shape are launch-bound -- at 1024 x 2048 they are not, and eager is the faster path).
        model.graph_replay = (True if n_streams > 1 else "auto") if use_graphs else False
    graphed = ({id(st_): GraphedScore(model, score_func, st_) for st_ in streams}
               if use_graphs and not args.store_anomaly_scores else None)
    scores, labels = [], []
    pending = []
    FLUSH = 32
    fallbacks = []

    def rescored(k_img, x):
        """The f16x3 arithmetic of the token Linears answers an activation or weight beyond f16's range (|v| >= 65504) with NaN, never
        with a wrong number (csrc/split_linear_h3.h) -- here that answer is acted upon: the image is scored again, eagerly, on the
        full-range bf16x6 kernels.  A score that is still not finite is an error of the model or the image, and is raised."""
        from . import ops
        with ops.split_mode("bf16x6"):
            prev = getattr(model, "graph_replay", None)
            if prev is not None:
                model.graph_replay = False
            try:
                s2 = score_func(model, x[None])
            finally:
                if prev is not None:
                    model.graph_replay = prev
        if bool(torch.isnan(s2).any()):
            raise FloatingPointError(f"{dataset_name}: image {k_img} has a NaN anomaly score in the f16x3 AND the bf16x6 arithmetic")
        fallbacks.append(k_img)
        print(f"[rba_amd] {dataset_name}: image {k_img}: non-finite score in f16x3 arithmetic, re-scored on the bf16x6 kernels", file=sys.stderr)
        return s2.reshape(-1)

    def flush():
        if not pending:
            return
        if on_gpu:
            for st_ in {p[2] for p in pending}:
                if st_ is not main_stream:
                    main_stream.wait_stream(st_)
        # one fused check per chunk: a NaN score must never reach the sort of the rank statistics.  NaN -- not +-inf -- is the f16x3 Linear's
        # overflow answer (h = inf, l = NaN: the whole output row is NaN); an infinite score of a custom score function is passed on as
        # the reference passes it on
        nan = torch.stack([torch.isnan(p[0]).any() for p in pending]).cpu()
        for j in nan.nonzero().reshape(-1).tolist():
            s_, y_, st_, x_, k_img, shp_ = pending[j]
            pending[j] = (rescored(k_img, x_), y_, main_stream, x_, k_img, shp_)
        if args.store_anomaly_scores:                              # saved AFTER the check: the map the metrics use (a re-scored one included)
            vis = os.path.join("anomaly_scores", model_name, dataset_name)
            os.makedirs(vis, exist_ok=True)
            for p in pending:
                np.save(os.path.join(vis, f"score_{p[4]}.npy"), p[0].reshape(p[5]).cpu().numpy())
        ss, yy = select_labelled(torch.cat([p[0] for p in pending]), torch.cat([p[1] for p in pending]))
        if on_gpu:
            for p in pending:                                      # produced on a side stream, last used here on the main stream
                for t_ in (p[0], p[1], p[3]):
                    t_.record_stream(main_stream)
        scores.append(ss)
        labels.append(yy)
        pending.clear()

    seen_shapes = set()
    t0 = time.perf_counter()
    k = 0
    host = {"wait_decode_s": 0.0, "score_calls_s": 0.0, "select_s": 0.0}          # where the scoring thread's time goes (timing only)
    t_prev = t0
    for xb, yb in loader:
        host["wait_decode_s"] += time.perf_counter() - t_prev
        for j in range(xb.shape[0]):                               # the scoring functions are batch-1 (reference :143-150)
            st = streams[k % len(streams)]
            i = mine[k]
            k += 1
            hwc = xb[j].dim() == 3 and xb[j].shape[-1] == 3 and xb[j].shape[0] != 3      # a raw item: channels last
            shape = tuple(xb[j].shape)
            first_of_shape = shape not in seen_shapes
            if first_of_shape and on_gpu:
                # the model builds per-shape constants (position embeddings, gather plans, weight planes on the very first call) on
                # the stream that first needs them: do that alone on the main stream and let it finish before streams share them
                seen_shapes.add(shape)
                for s_ in streams:
                    if s_ is not main_stream:
                        main_stream.wait_stream(s_)
                st = main_stream
            if on_gpu and st is not main_stream:
                st.wait_stream(main_stream)
            ctx = torch.cuda.stream(st) if on_gpu else _nullcontext()
            with ctx:
                t_a = time.perf_counter()
                x = xb[j].to(dev, non_blocking=True)
                if hwc:
                    x = x.permute(2, 0, 1).contiguous()            # [3,H,W] as the model takes it
                y = yb[j].to(dev, non_blocking=True)
                # a graph belongs to the stream it was captured on; the first image of a shape runs on the main stream (above) and
                # is scored eagerly there
                s = graphed[id(st)](x) if graphed is not None and not first_of_shape else score_func(model, x[None])
                t_b = time.perf_counter()
                # torch.nonzero waits for the GPU (its result has a data-dependent size): compacting every image would
                # serialise host and GPU image by image -- the launches of image i + 1 could not be issued while image i runs
                # (measured: 2-3 ms of waiting per image and no overlap between streams).  Park (score, label) maps instead and
                # compact FLUSH images at a time: one wait per FLUSH images, at most FLUSH x 10 MB parked at 1024 x 2048.
                pending.append((s.reshape(-1), y.reshape(-1), st, x, i, tuple(s.shape)))
                host["score_calls_s"] += t_b - t_a
            if len(pending) >= FLUSH:
                t_c = time.perf_counter()
                flush()
                host["select_s"] += time.perf_counter() - t_c
            if first_of_shape and on_gpu:
                torch.cuda.synchronize(dev)
        t_prev = time.perf_counter()
    t_c = time.perf_counter()
    flush()
    host["select_s"] += time.perf_counter() - t_c
    if on_gpu:
        for st in streams:
            if st is not main_stream:
                main_stream.wait_stream(st)
        torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if timing is not None:
        n_graphs = (sum(1 for gs in graphed.values() for e in gs.graphs.values() if isinstance(e, tuple)) if graphed else 0)
        n_graphs += model.live_graphs() if hasattr(model, "live_graphs") else 0
        timing.update(images=k, seconds=dt, images_per_s=(k / dt if dt > 0 else 0.0), num_workers=nw, loader=("processes" if use_procs else "threads"), streams=len(streams),
                      hip_graphs=n_graphs, bf16x6_rescored_images=list(fallbacks),
                      host_thread={n: round(v, 3) for n, v in host.items()})
        if proc_stats.get("items"):
            timing["decode_processes_ms_per_sample"] = {n: round(v / proc_stats["items"] * 1e3, 2) for n, v in proc_stats.items() if n != "items"}
    s_all = torch.cat(scores) if scores else torch.empty(0, device=dev)
    y_all = torch.cat(labels) if labels else torch.empty(0, dtype=torch.bool, device=dev)
    return D.pooled_ood_metrics(s_all, y_all)


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def open_decoder(args):
    """The decode worker processes of `--loader processes` (None for the thread loader): started once per run, shared by every model x dataset"""
    if getattr(args, "loader", "threads") != "processes" or int(args.num_workers) <= 0 or args.device != "cuda":
        return None
    from .datasets import ProcessDecoder
    return ProcessDecoder(int(args.num_workers))


def main(argv=None):
    from pprint import pprint
    from . import distributed as D
    from .datasets import available_datasets, get_dataset
    args = build_parser().parse_args(argv)
    launch_local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.share_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local = D.init_from_env(args.dist_backend or ("gloo" if args.share_device else None))
    device = torch.device(args.device, local) if args.device == "cuda" else torch.device(args.device)
    if device.type == "cuda":
        torch.cuda.set_device(device)
    if world > 1:                    # one process per GPU: decode threads and the launch thread stay on the cores next to this rank's GPU
        D.bind_rank_to_gpu_numa(launch_local, int(os.environ.get("LOCAL_WORLD_SIZE", world)), device_of=(lambda r: 0) if args.share_device else None)
    names = args.selected_datasets if args.dataset_mode == "selective" else ["road_anomaly", "fishyscapes_laf"]
    if not names:
        raise ValueError("Selective Mode is chosen but number of selected datasets is 0")
    unknown = [n for n in names if n not in available_datasets()]
    if unknown:
        raise ValueError(f"unknown datasets {unknown}; available: {available_datasets()}")
    models = sorted(m for m in os.listdir(args.models_folder) if os.path.isdir(os.path.join(args.models_folder, m)))
    if args.model_mode == "selective":
        models = [m for m in models if m in args.selected_models]
    if not models:
        raise ValueError("Number of models chosen is 0, either the models folder is empty or no models were selected")
    args.decoder = open_decoder(args)
    try:
        _evaluate_models(models, names, args, rank, world, device)
    finally:
        if args.decoder is not None:
            args.decoder.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def _evaluate_models(models, names, args, rank, world, device):
    from pprint import pprint
    from .datasets import get_dataset
    for model_name in models:
        exp = os.path.join(args.models_folder, model_name)
        store = os.path.join(args.out_path, model_name)
        if os.path.exists(os.path.join(store, "results.pkl")):
            if rank == 0:
                print(f"Skipping {model_name} because results already exist, if you want to re-run, delete the results.pkl file")
            continue
        model_path = os.path.join(exp, "model_final.pth")
        if not os.path.exists(model_path):
            model_path = os.path.join(exp, "model_final.pkl")
            if not os.path.exists(model_path):
                model_path = os.path.join("model_logs", model_name, "model_final.pkl")      # reference fallback (:271-273)
                if not os.path.exists(model_path):
                    if rank == 0:
                        print("Model path does not exist, skipping")
                    continue
        model = get_model(os.path.join(exp, "config.yaml"), model_path, device=device)
        results, timings = {}, {}
        for name in names:
            timings[name] = {}
            results[name] = run_evaluations(model, get_dataset(name, args.datasets_folder), model_name, name, args, rank, world,
                                            timing=timings[name])
        if rank == 0:
            if args.verbose:
                pprint(results)
                pprint({k: {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in timings.items()})
            Path(store).mkdir(exist_ok=True, parents=True)
            with open(os.path.join(store, "results.pkl"), "wb") as f:
                pickle.dump(results, f)


if __name__ == "__main__":
    main()
