"""``MaskFormer`` meta-architecture, inference branch only (reference: mask2former/maskformer_model.py:23-386).

Drop-in surface: ``model([{"image": uint8/float [3,H,W] RGB 0..255}]) -> [{"sem_seg": [K,H,W] fp32}]`` on the model's
device, identical to the reference, so the unmodified ``get_RbA`` / ``get_logits`` of evaluate_ood.py work on it.
In addition each result dict carries ``"rba"`` ([H,W], the RbA score of evaluate_ood.py:150 produced by the same
fused kernel) and, on request, ``"argmax"``; ``model.rba_scores(...)`` skips materialising ``sem_seg`` altogether.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .h2d import to_device
from .arch import backbone_name as _backbone_name, complete
from .modeling.backbone import resnet as _resnet  # noqa: F401  (registers build_resnet_backbone)
from .modeling.backbone import swin as _swin  # noqa: F401  (registers D2SwinTransformer)
from .modeling.meta_arch import mask_former_head as _head  # noqa: F401
from .registry import BACKBONE_REGISTRY, META_ARCH_REGISTRY, SEM_SEG_HEADS_REGISTRY


def _class_prob(mask_cls):
    """F.softmax(mask_cls, -1)[..., :-1] (reference maskformer_model.py:381-383), one launch.  The kernel (2 <= classes + 1 <= 64) uses its own exp / sum order:
    equal to F.softmax to ~1e-7 (tests/test_kernels_gpu.py::test_softmax_drop_last_vs_torch), so `sem_seg` / argmax agree with the fp32 torch path except on
    near-ties -- the tolerance every end-to-end fixture already grants (top-2 gap < 1e-4)."""
    if 2 <= mask_cls.shape[-1] <= 64:
        return ops.softmax_drop_last(mask_cls.contiguous())
    return F.softmax(mask_cls, dim=-1)[..., :-1].contiguous()


@META_ARCH_REGISTRY.register()
class MaskFormer(nn.Module):
    def __init__(self, arch, backbone_name=None, head_name="MaskFormerHead"):
        super().__init__()
        a = complete(arch)
        self.arch = a
        self.backbone = BACKBONE_REGISTRY.get(backbone_name or _backbone_name(a))(a)
        self.sem_seg_head = SEM_SEG_HEADS_REGISTRY.get(head_name)(a)
        self.num_queries = a["num_queries"]
        self.size_divisibility = a["size_divisibility"] if a["size_divisibility"] > 0 else self.backbone.size_divisibility
        self.register_buffer("pixel_mean", torch.tensor(a["pixel_mean"], dtype=torch.float32).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(a["pixel_std"], dtype=torch.float32).view(-1, 1, 1), False)
        self._mean3 = tuple(float(torch.tensor(v, dtype=torch.float32)) for v in a["pixel_mean"])      # fp32-rounded, as the buffers hold them
        self._std3 = tuple(float(torch.tensor(v, dtype=torch.float32)) for v in a["pixel_std"])
        self.fused_front_end = True
        # rba_scores(): batch-1 forwards replayed from a per-shape hipGraph.  "auto" (default, round 6): per shape, only where the eager launches are MEASURED to be
        # launch-bound (host issue time vs GPU span of an eager call, see _graphed_scores) -- at 1024 x 2048 the GPU is the bottleneck and eager is 4 % faster than
        # replay on one stream (BENCH_r05: 124.7 vs 119.7 images/s), at 720 x 1280 / 9 layers and on small images replay wins.  True: always (the evaluator with
        # several streams, tests); False: never.
        self.graph_replay = "auto"
        self.fused_upsample = True      # K1 reads the low-res logits and up-samples on the fly (rba_reduce_up4)
        # panoptic inference (maskformer_model.py:202-220): off unless TEST.PANOPTIC_ON
        self.panoptic_on, self.open_panoptic = bool(a["panoptic_on"]), bool(a["open_panoptic"])
        self.sem_seg_postprocess_before_inference = bool(a["postprocess_before_inference"]) or self.panoptic_on
        self.object_mask_threshold, self.overlap_threshold = float(a["object_mask_threshold"]), float(a["overlap_threshold"])
        self.thing_classes = frozenset(int(c) for c in a["thing_classes"])
        self.eval()

    @property
    def device(self):
        return self.pixel_mean.device

    # ------------------------------------------------------------------ pre-processing (:255-257)
    def preprocess(self, batched_inputs):
        images = [(to_device(x["image"], self.device).float() - self.pixel_mean) / self.pixel_std for x in batched_inputs]
        sizes = [tuple(int(v) for v in im.shape[-2:]) for im in images]
        d = self.size_divisibility
        H = (max(s[0] for s in sizes) + d - 1) // d * d
        W = (max(s[1] for s in sizes) + d - 1) // d * d
        batch = images[0].new_zeros((len(images), images[0].shape[0], H, W))
        for i, im in enumerate(images):
            batch[i, :, : sizes[i][0], : sizes[i][1]] = im       # ImageList.from_tensors: zero pad bottom/right
        return batch, sizes

    # ------------------------------------------------------------------ network
    @torch.no_grad()
    def _predict_outputs(self, batched_inputs):
        pe = getattr(self.backbone, "patch_embed", None)
        if (self.fused_front_end and pe is not None and hasattr(self.backbone, "forward_images") and pe.fused_ok()
                and all(x["image"].dim() == 3 and x["image"].shape[0] == 3 and x["image"].dtype in (torch.uint8, torch.float32)
                        for x in batched_inputs)):
            # normalisation + ImageList padding + patch im2col in one kernel, the projection on K6 (backbone/swin.py PatchEmbed.forward_images)
            images = [to_device(x["image"], self.device).contiguous() for x in batched_inputs]
            sizes = [tuple(int(v) for v in im.shape[-2:]) for im in images]
            d = self.size_divisibility
            H = (max(s[0] for s in sizes) + d - 1) // d * d
            W = (max(s[1] for s in sizes) + d - 1) // d * d
            features = self.backbone.forward_images(images, self._mean3, self._std3, H, W)
            return self.sem_seg_head(features), sizes, (H, W)
        batch, sizes = self.preprocess(batched_inputs)
        features = self.backbone(batch)
        return self.sem_seg_head(features), sizes, tuple(batch.shape[-2:])

    @torch.no_grad()
    def predict(self, batched_inputs):
        """-> (pred_logits [B,Q,K+1], pred_masks [B,Q,H/4,W/4], image_sizes, padded (H,W))."""
        outputs, sizes, padded = self._predict_outputs(batched_inputs)
        return outputs["pred_logits"], outputs["pred_masks"], sizes, padded

    def _post(self, mask_cls, mask_pred, image_size, padded, want_sem_seg, want_argmax, score="rba"):
        """Up-sample (:294-299), semantic inference (:381-386), crop (:330-332), RbA (evaluate_ood.py:150)."""
        prob = _class_prob(mask_cls)
        H, W = padded
        exact4 = mask_pred.shape[-2] * 4 == H and mask_pred.shape[-1] * 4 == W
        if self.fused_upsample and exact4 and prob.shape[1] <= 32:
            return ops.rba_reduce_up4(mask_pred.contiguous(), prob, image_size, want_sem_seg, want_argmax, score)
        up = ops.resample_bilinear(mask_pred.contiguous(), (H, W))
        rba, sem, arg = ops.rba_reduce(up, prob, want_sem_seg, want_argmax, score)
        h, w = image_size
        if (h, w) != (H, W):
            rba = rba[:h, :w].contiguous()
            sem = sem[:, :h, :w].contiguous() if sem is not None else None
            arg = arg[:h, :w].contiguous() if arg is not None else None
        return rba, sem, arg

    # ------------------------------------------------------------------ open-set panoptic inference (:394-486)
    @torch.no_grad()
    def panoptic_inference(self, mask_cls, mask_pred, open_panoptic=False, ood_threshold=-0.1, pixel_min=300,
                           return_ood_pred=False):
        """mask_cls [Q,K+1] logits, mask_pred [Q,H,W] logits at the output resolution ->
        (panoptic_seg int32 [H,W], segments_info[, rba map]).  Closed-set part: the Mask2Former merge (:395-452) -- queries whose
        best class is not void and scores above ``object_mask_threshold`` compete per pixel by score x sigmoid(mask); a query
        keeps its pixels if enough of its own >= 0.5 area wins (``overlap_threshold``); "stuff" classes merge into one segment.
        Open-set part (:454-481): the RbA map (K1) thresholded, 3x3 opened and closed, 4-connected components of at least
        ``pixel_min`` still-unclaimed pixels become new "thing" segments of category 255."""
        Q, H, W = mask_pred.shape
        K = mask_cls.shape[-1] - 1
        prob_all = F.softmax(mask_cls, dim=-1)
        scores, labels = prob_all.max(-1)
        keep = labels.ne(K) & (scores > self.object_mask_threshold)
        panoptic_seg = torch.zeros((H, W), dtype=torch.int32, device=mask_pred.device)
        segments_info = []
        if not bool(keep.any()):
            return panoptic_seg, segments_info                        # (the reference also skips the open-set part here, :414-416)
        cur_scores, cur_classes = scores[keep], labels[keep]
        cur_masks = mask_pred[keep].sigmoid()
        cur_mask_ids = (cur_scores.view(-1, 1, 1) * cur_masks).argmax(0)
        current = 0
        stuff_ids = {}
        for k in range(cur_classes.shape[0]):
            cls = int(cur_classes[k])
            won = cur_mask_ids == k
            solid = cur_masks[k] >= 0.5
            mask = won & solid
            mask_area, original_area, inter = int(won.sum()), int(solid.sum()), int(mask.sum())
            if mask_area == 0 or original_area == 0 or inter == 0 or mask_area / original_area < self.overlap_threshold:
                continue
            isthing = cls in self.thing_classes
            if not isthing:
                if cls in stuff_ids:
                    panoptic_seg[mask] = stuff_ids[cls]
                    continue
                stuff_ids[cls] = current + 1
            current += 1
            panoptic_seg[mask] = current
            segments_info.append({"id": current, "isthing": bool(isthing), "category_id": cls})
        if not open_panoptic:
            return panoptic_seg, segments_info
        ood_mask = ops.rba_reduce(mask_pred.contiguous(), prob_all[:, :-1].contiguous())[0]      # -sum_k tanh(sem_k), K1
        comp, n = ops.ood_components(ood_mask, ood_threshold)
        if n:
            free = panoptic_seg == 0
            sizes = torch.bincount(comp[free].long(), minlength=n + 1)
            ok = sizes >= pixel_min
            ok[0] = False                                                                     # background
            new_id = current + torch.cumsum(ok.to(torch.int32), 0, dtype=torch.int32)         # in component (= raster) order
            sel = free & ok[comp.long()]
            panoptic_seg[sel] = new_id[comp.long()][sel]
            for _ in range(int(ok.sum())):
                current += 1
                segments_info.append({"id": current, "isthing": True, "category_id": 255})
        if return_ood_pred:
            return panoptic_seg, segments_info, ood_mask
        return panoptic_seg, segments_info

    @torch.no_grad()
    def forward(self, batched_inputs, include_void=False, return_separately=False, return_aux=False,
                return_ood_pred=False, return_argmax=False, panoptic_ood_threshold=-0.3, panoptic_pixel_min=200,
                return_panoptic_ood=False, **kwargs):
        if include_void or return_separately or return_aux or kwargs:
            raise NotImplementedError("only the semantic and (open-set) panoptic inference paths of MaskFormer.forward are provided")
        outputs, sizes, padded = self._predict_outputs(batched_inputs)
        mask_cls, mask_pred = outputs["pred_logits"], outputs["pred_masks"]
        ood_pred = None
        if return_ood_pred:                                     # :303-305: up-sampled to the FIRST image's size, align_corners=True
            if "ood_pred" not in outputs:
                raise KeyError("ood_pred: the model has no DenseHybrid head (MODEL.MASK_FORMER.DENSE_HYBRID_LOSS)")
            ood_pred = ops.resample_bilinear_ac(outputs["ood_pred"].contiguous(), sizes[0])
        results = []
        for i, inp in enumerate(batched_inputs):
            height, width = inp.get("height", sizes[i][0]), inp.get("width", sizes[i][1])
            if (height, width) != sizes[i] and self.sem_seg_postprocess_before_inference:
                # :316-320: crop + resize the MASK LOGITS to the requested resolution first, semantic inference on those
                up = ops.resample_bilinear(mask_pred[i].contiguous(), padded)[:, : sizes[i][0], : sizes[i][1]].contiguous()
                up = ops.resample_bilinear(up, (height, width))
                prob = _class_prob(mask_cls[i])
                rba, sem, arg = ops.rba_reduce(up, prob, True, return_argmax)
            else:
                rba, sem, arg = self._post(mask_cls[i], mask_pred[i], sizes[i], padded, True, return_argmax)
                if (height, width) != sizes[i]:   # :330-332: sem_seg_postprocess of the class maps to the requested resolution
                    sem = ops.resample_bilinear(sem, (height, width))
                    rba = -sem.tanh().sum(dim=0)
                    arg = sem.argmax(0).to(torch.int32) if return_argmax else None
            r = {"sem_seg": sem, "rba": rba}
            if return_argmax:
                r["argmax"] = arg
            if self.panoptic_on:                                   # :337-341; masks first brought to the output resolution (:318-321)
                mp = ops.resample_bilinear(mask_pred[i].contiguous(), padded)[:, : sizes[i][0], : sizes[i][1]].contiguous()
                if (height, width) != sizes[i]:
                    mp = ops.resample_bilinear(mp, (height, width))
                r["panoptic_seg"] = self.panoptic_inference(mask_cls[i], mp, self.open_panoptic, panoptic_ood_threshold,
                                                            panoptic_pixel_min, return_panoptic_ood)
            results.append(r)
        if return_ood_pred:
            return results, ood_pred                            # :350-351
        return results

    # ------------------------------------------------------------------ hipGraph replay of the scoring path
    GRAPH_MAX = 8            # live graphs per model (each keeps its own activation pool: ~1.5 GB at 1024 x 2048)
    GRAPH_THRASH_MAX = 4     # never-replayed graphs evicted before the model gives up capturing new keys
    LAUNCH_BOUND_RATIO = 0.95      # graph_replay = "auto": capture a shape when host issue time >= this fraction of the GPU span of an eager forward
    GRAPH_REMEASURE_EVERY = 32     # ... and measure a GPU-bound shape again after this many eager calls

    def graph_decisions(self):
        """graph_replay = "auto": {(image shape, dtype, return_argmax, score): {"decision": "eager" | "replay", "host_issue_ms", "gpu_span_ms"}} of the last
        measurement of every shape met so far (bench.py reports it; tests assert on it)"""
        return dict(self.__dict__.get("_graph_decisions", {}))

    def _measured_eager(self, key, batched_inputs, return_argmax, score):
        import time
        dev = self.device
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(torch.cuda.current_stream(dev))
        out = self._rba_scores_eager(batched_inputs, return_argmax, score)
        e1.record(torch.cuda.current_stream(dev))
        seen = self.__dict__.setdefault("_graph_seen", {})
        if key in seen:
            seen[key] = (time.perf_counter() - t0, e0, e1)
        return out

    def _weights_version(self):
        # the Parameter OBJECTS of a module tree are stable (load_state_dict copies in place, .to() swaps .data): walk the tree once, then sum the versions of a
        # flat list -- the generator walk cost 0.3 ms per call, which a serial caller pays in front of every image's first launch (round 6)
        flat = self.__dict__.get("_param_flat")
        if flat is None:
            flat = self.__dict__["_param_flat"] = list(self.parameters())
        v = 0
        for p_ in flat:
            v += p_._version
        return v

    def _graph_key(self, image, return_argmax, score):
        return (tuple(image.shape), image.dtype, image.device, torch.cuda.current_stream(image.device).cuda_stream, bool(return_argmax), score,
                self.fused_upsample, self.fused_front_end, ops.SPLIT_MODE, ops.SPLIT_ACTIVATIONS, ops.TILES_MIN, ops.MLP_FUSED_MIN_ROWS, ops.concurrent_streams(),
                ops.SWIN_ATTN_FUSED,
                getattr(self.sem_seg_head.predictor, "sparse_intermediate_heads", None), self._weights_version())

    def drop_graphs(self, release_constants=False):
        """Forget every captured graph (their pools are freed); the next calls run eagerly, then capture again.  release_constants: also
        un-pin the per-shape constants captures have read (lru.ShapeCache) -- only safe when no graph captured OUTSIDE the model (bench.py,
        evaluate_ood.GraphedScore) is still alive, e.g. after a device move."""
        if release_constants:
            from .lru import ShapeCache
            for m_ in self.modules():
                for v_ in vars(m_).values():
                    if isinstance(v_, ShapeCache):
                        v_.unpin()
        self.__dict__.pop("_graphs", None)
        self.__dict__.pop("_graph_seen", None)
        self.__dict__.pop("_graph_thrash", None)
        self.__dict__.pop("_graph_probe", None)
        self.__dict__.pop("_graph_eager_fast", None)

    def __getstate__(self):
        st = dict(super().__getstate__() if hasattr(super(), "__getstate__") else self.__dict__)
        for k in ("_graphs", "_graph_seen", "_graph_thrash", "_graph_probe", "_graph_decisions", "_graph_eager_fast", "_param_flat"):       # hipGraphs / events cannot be copied or pickled; a copy captures its own
            st.pop(k, None)
        return st

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in ("_graphs", "_graph_seen", "_graph_thrash", "_graph_probe", "_graph_decisions", "_graph_eager_fast", "_param_flat"):
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_param_flat", None)
        self.drop_graphs(release_constants=True)   # .to() / .cuda() / .float(): every captured graph holds the old parameter addresses
        return super()._apply(fn, *args, **kwargs)

    def _graphed_scores(self, image, return_argmax, score):
        """One batch-1 image (already on the device) -> (rba, argmax | None), replayed from a hipGraph captured per (image shape, dtype,
        stream, outputs, arithmetic mode, weight version).  A forward is ~330 kernel launches that take the Python thread 6-7 ms to
        issue against ~8 ms of GPU time; a caller that waits for every score (the reference's loop does: `.cpu()` per image,
        support.py:375) therefore pays both in series.  The first call of a key runs eagerly (per-shape constants, weight planes), the
        second captures, later ones copy the image into the graph's input and replay it on the CURRENT stream; the result is a fresh
        tensor.  Returns None when the capture failed (the caller then runs eagerly; the failure is remembered for the key)."""
        graphs = self.__dict__.setdefault("_graphs", {})
        seen = self.__dict__.setdefault("_graph_seen", {})        # keys met once, not captured yet: bounded on its own, never evicts a graph
        key = self._graph_key(image, return_argmax, score)
        entry = graphs.get(key)
        if entry is None:
            if self.__dict__.get("_graph_thrash", 0) >= self.GRAPH_THRASH_MAX:
                return None                                     # image shapes churn faster than graphs are replayed: stay eager (see below)
            if key not in seen:
                seen[key] = True
                while len(seen) > 4 * self.GRAPH_MAX:
                    seen.pop(next(iter(seen)))
                return None
            if self.graph_replay == "auto":
                # measured policy (round 6).  Call 1 of a key ran eagerly (lazy initialisation); call 2 runs eagerly between two HIP events with the host's issue
                # time taken beside them (rba_scores -> _measured_eager); call 3 reads the pair: the forward is LAUNCH-BOUND when the Python thread needed at
                # least LAUNCH_BOUND_RATIO of the GPU's own span to issue it (a launch-bound GPU span stretches to the issue time, so the ratio saturates
                # near 1) -- then, and only then, the shape is captured.  A GPU-bound shape stays eager and is measured again every GRAPH_REMEASURE_EVERY
                # calls (the host may get busier: decode threads, other ranks).
                st = seen[key]
                if st is True or (isinstance(st, int) and st <= 0):
                    self.__dict__["_graph_probe"] = key
                    return None
                if isinstance(st, int):
                    seen[key] = st - 1
                    return None
                t_issue, e0, e1 = st
                e1.synchronize()
                t_gpu = e0.elapsed_time(e1) * 1e-3
                bound = t_issue >= self.LAUNCH_BOUND_RATIO * t_gpu
                self.__dict__.setdefault("_graph_decisions", {})[key[:2] + key[4:6]] = {
                    "decision": "replay" if bound else "eager", "host_issue_ms": t_issue * 1e3, "gpu_span_ms": t_gpu * 1e3}
                if not bound:
                    seen[key] = 0                                   # the next slow-path call of this key is a measurement again ...
                    fast = self.__dict__.setdefault("_graph_eager_fast", {})
                    while len(fast) > 4 * self.GRAPH_MAX:
                        fast.pop(next(iter(fast)))
                    # ... and rba_scores serves the next GRAPH_REMEASURE_EVERY calls of the shape eagerly before it asks again
                    fast[(key[0], key[1], key[4], key[5], key[8], key[3])] = self.GRAPH_REMEASURE_EVERY
                    return None
            del seen[key]
            entry = "seen"
            while len(graphs) >= self.GRAPH_MAX:                # oldest first; a graph owns its pool, dropping it frees the memory
                old = graphs.pop(next(iter(graphs)))            # (on ROCm destroying a graph synchronises the device)
                if isinstance(old, tuple) and old[4][0] <= 1:             # 1 = only the replay that followed its capture
                    # a graph that was never replayed is being evicted: with more live shapes than GRAPH_MAX every capture costs more than it
                    # saves; after GRAPH_THRASH_MAX of these the model stops capturing new keys (drop_graphs() resets)
                    self.__dict__["_graph_thrash"] = self.__dict__.get("_graph_thrash", 0) + 1
        if entry == "seen":
            try:
                static_in = image.clone()
                cap = torch.cuda.Stream(device=image.device)       # its own capture stream: per-stream state (K1's tile counter) is this graph's alone
                cap.wait_stream(torch.cuda.current_stream(image.device))
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
                    r = self._rba_scores_eager([{"image": static_in}], return_argmax, score)[0]
                torch.cuda.current_stream(image.device).wait_stream(cap)
                entry = graphs[key] = (g, static_in, r, cap, [0])
            except Exception as e:                                   # noqa: BLE001 -- an optimisation only
                import sys
                print(f"[rba_amd] hipGraph capture failed for image shape {tuple(image.shape)} ({type(e).__name__}: {e}); eager launches",
                      file=sys.stderr)
                torch.cuda.synchronize(image.device)
                entry = graphs[key] = False
        if entry is False:
            return None
        graphs[key] = graphs.pop(key)                               # most recently used last
        g, static_in, r, _, uses = entry
        uses[0] += 1
        static_in.copy_(image, non_blocking=True)
        g.replay()
        return (r[0].clone(), r[1].clone()) if return_argmax else r.clone()

    def live_graphs(self):
        return sum(1 for e in self.__dict__.get("_graphs", {}).values() if isinstance(e, tuple))

    @torch.no_grad()
    def rba_scores(self, batched_inputs, return_argmax=False, score="rba"):
        """Fast path: anomaly-score maps (and optional int32 argmax maps) without materialising sem_seg.
        score: "rba" (evaluate_ood.py:143-150), "energy" (:152-159) or "neg_logit_sum" (support.py:115-132).
        Batch-1 calls on a HIP device are replayed from a captured hipGraph -- with ``model.graph_replay = True`` from the third call of an image shape on, with
        the default "auto" from the fourth call on and only for shapes whose eager launches are measured to be launch-bound (see _graphed_scores);
        ``model.graph_replay = False`` opts out."""
        if (self.graph_replay and len(batched_inputs) == 1 and self.device.type == "cuda"
                and not torch.cuda.is_current_stream_capturing()):
            image = batched_inputs[0]["image"]
            if torch.is_tensor(image) and image.dim() == 3 and image.dtype in (torch.uint8, torch.float32) \
                    and set(batched_inputs[0]) <= {"image"}:
                image = to_device(image, self.device).contiguous()
                if self.graph_replay == "auto":
                    # a shape measured to be GPU-bound stays eager for GRAPH_REMEASURE_EVERY calls without even building the graph key (the key walks every
                    # parameter's version: host time in front of the image's first launch, which the serial caller pays in full).  A stale entry costs
                    # speed only -- eager launches are always right.
                    fast = self.__dict__.setdefault("_graph_eager_fast", {})
                    fk = (tuple(image.shape), image.dtype, bool(return_argmax), score, ops.SPLIT_MODE, torch.cuda.current_stream(image.device).cuda_stream)
                    left = fast.get(fk, 0)
                    if left > 0:
                        fast[fk] = left - 1
                        return self._rba_scores_eager([{"image": image}], return_argmax, score)
                r = self._graphed_scores(image, return_argmax, score)
                if r is not None:
                    return [r]
                batched_inputs = [{"image": image}]
                probe = self.__dict__.pop("_graph_probe", None)
                if probe is not None:                          # graph_replay = "auto": this eager call is the one that is measured
                    return self._measured_eager(probe, batched_inputs, return_argmax, score)
        return self._rba_scores_eager(batched_inputs, return_argmax, score)

    @torch.no_grad()
    def _rba_scores_eager(self, batched_inputs, return_argmax=False, score="rba"):
        mask_cls, mask_pred, sizes, padded = self.predict(batched_inputs)
        out = []
        for i in range(len(batched_inputs)):
            rba, _, arg = self._post(mask_cls[i], mask_pred[i], sizes[i], padded, False, return_argmax, score)
            out.append((rba, arg) if return_argmax else rba)
        return out
