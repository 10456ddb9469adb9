"""ORACLE -- test infrastructure, not product code.

CPU (torch fp32/fp64) restatement of the arithmetic kernels on RbA's inference hot path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; nothing under ``rba_amd/`` does.  Every function cites the reference
lines (relative to /root/reference) it restates.  Pinned against fixtures produced by the
reference's own modules: ``tests/golden/*.npz`` (generator: ``tests/golden/make_golden.py``).
"""
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------- K1
def class_probs(mask_cls: torch.Tensor) -> torch.Tensor:
    """softmax over K+1 class logits, void column dropped -> [Q, K]
    (mask2former/maskformer_model.py:382)."""
    return F.softmax(mask_cls, dim=-1)[..., :-1]


def semantic_inference(mask_cls: torch.Tensor, mask_pred: torch.Tensor) -> torch.Tensor:
    """sem_seg[k,h,w] = sum_q softmax(cls_q)[k] * sigmoid(mask_q[h,w])
    (mask2former/maskformer_model.py:381-386)."""
    return torch.einsum("qc,qhw->chw", class_probs(mask_cls), mask_pred.sigmoid())


def rba_score(sem_seg: torch.Tensor) -> torch.Tensor:
    """RbA = -sum_k tanh(sem_seg[k]) (evaluate_ood.py:143-150; duplicate support.py:135-142)."""
    return -sem_seg.tanh().sum(dim=0)


def densehybrid_score(sem_seg: torch.Tensor, ood_pred: torch.Tensor) -> torch.Tensor:
    """get_densehybrid_score (evaluate_ood.py:161-173): -logsumexp_k(sem_seg) + log(softmax(ood_pred, 1)[:, 1] + 1e-9);
    sem_seg [K,H,W], ood_pred [1,2,H,W] -> [1,H,W]."""
    p1 = torch.logsumexp(sem_seg, dim=0)
    p2 = F.softmax(ood_pred, dim=1)[:, 1]
    return (-p1) + (p2 + 1e-9).log()


def upsample_bilinear(x: torch.Tensor, size) -> torch.Tensor:
    """F.interpolate(mode="bilinear", align_corners=False) (maskformer_model.py:294-299;
    msdeformattn.py:358; mask2former_transformer_decoder.py:483)."""
    return F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False)


def rba_reduce(mask_pred: torch.Tensor, mask_cls: torch.Tensor):
    """(sem_seg [K,H,W], rba [H,W], argmax [H,W] int64) from full-resolution mask logits [Q,H,W] and
    class logits [Q,K+1]: maskformer_model.py:381-386 + evaluate_ood.py:150 + support.py:385-388."""
    sem = semantic_inference(mask_cls, mask_pred)
    return sem, rba_score(sem), sem.max(dim=0)[1]


def rba_reduce_ordered(mask_pred: torch.Tensor, cls_prob: torch.Tensor):
    """Same contraction with a FIXED ascending-q fp32 accumulation order (sem += p[q,k] * s[q]), the
    order the HIP kernel uses.  Differs from the einsum above only by fp32 re-association (<= ~1e-6);
    used to separate 'kernel bug' from 're-association' when a near-tie argmax flips."""
    Q = mask_pred.shape[0]
    K = cls_prob.shape[1]
    sem = torch.zeros((K,) + tuple(mask_pred.shape[1:]), dtype=mask_pred.dtype)
    s = mask_pred.sigmoid()
    for q in range(Q):
        sem.addcmul_(cls_prob[q].view(K, 1, 1), s[q].unsqueeze(0))
    return sem, -sem.tanh().sum(0), sem.max(dim=0)[1]


# ------------------------------------------------------------------------------- K2
def ms_deform_attn(value, spatial_shapes, sampling_locations, attention_weights):
    """Multi-scale deformable attention core, the reference's CPU path
    (pixel_decoder/ops/functions/ms_deform_attn_func.py:52-72): per level
    grid_sample(value_l, 2*loc-1, bilinear, zeros, align_corners=False), weighted sum over L*P.
    value [N,S,M,D], locations [N,Lq,M,L,P,2] (x,y in [0,1]), weights [N,Lq,M,L,P] -> [N,Lq,M*D]."""
    N_, S_, M_, D_ = value.shape
    _, Lq_, _, L_, P_, _ = sampling_locations.shape
    shapes = [(int(h), int(w)) for h, w in spatial_shapes]
    value_list = value.split([h * w for h, w in shapes], dim=1)
    grids = 2 * sampling_locations - 1
    sampled = []
    for lid, (H_, W_) in enumerate(shapes):
        v = value_list[lid].flatten(2).transpose(1, 2).reshape(N_ * M_, D_, H_, W_)
        g = grids[:, :, :, lid].transpose(1, 2).flatten(0, 1)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    w = attention_weights.transpose(1, 2).reshape(N_ * M_, 1, Lq_, L_ * P_)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * w).sum(-1).view(N_, M_ * D_, Lq_)
    return out.transpose(1, 2).contiguous()


def ms_deform_attn_loops(value, spatial_shapes, level_start_index, loc, w):
    """Scalar restatement of the CUDA kernel's arithmetic (ops/src/cuda/ms_deform_im2col_cuda.cuh:38-89,
    242-304): h_im = y*H - 0.5, w_im = x*W - 0.5, sample iff -1 < h_im < H and -1 < w_im < W, each of the
    four taps zero outside the map.  Pure-python loops: small cases only (cross-checks the grid_sample form)."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = torch.zeros(N, Lq, M * D, dtype=value.dtype)
    for n in range(N):
        for q in range(Lq):
            for m in range(M):
                acc = torch.zeros(D, dtype=value.dtype)
                for l in range(L):
                    H, W = int(spatial_shapes[l][0]), int(spatial_shapes[l][1])
                    base = int(level_start_index[l])
                    for p in range(P):
                        x, y = loc[n, q, m, l, p, 0].item(), loc[n, q, m, l, p, 1].item()
                        h_im, w_im = y * H - 0.5, x * W - 0.5
                        if not (h_im > -1 and w_im > -1 and h_im < H and w_im < W):
                            continue
                        h0, w0 = math.floor(h_im), math.floor(w_im)
                        lh, lw = h_im - h0, w_im - w0
                        val = torch.zeros(D, dtype=value.dtype)
                        for dh, dw, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw),
                                           (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
                            hh, ww = h0 + dh, w0 + dw
                            if 0 <= hh < H and 0 <= ww < W:
                                val = val + wt * value[n, base + hh * W + ww, m]
                        acc = acc + w[n, q, m, l, p] * val
                out[n, q, m * D:(m + 1) * D] = acc
    return out


# ------------------------------------------------------------------------------- K3
def multihead_attention(query, key, value, in_w, in_b, out_w, out_b, nheads, attn_mask=None):
    """nn.MultiheadAttention forward, seq-first [L,B,E], bool attn_mask [B*h, Lq, Lk] True = blocked
    (-inf before softmax); dropout 0 (mask2former_transformer_decoder.py:25-118 call sites)."""
    Lq, B, E = query.shape
    Lk = key.shape[0]
    hd = E // nheads
    q = F.linear(query, in_w[:E], in_b[:E])
    k = F.linear(key, in_w[E:2 * E], in_b[E:2 * E])
    v = F.linear(value, in_w[2 * E:], in_b[2 * E:])
    q = q.reshape(Lq, B * nheads, hd).transpose(0, 1)
    k = k.reshape(Lk, B * nheads, hd).transpose(0, 1)
    v = v.reshape(Lk, B * nheads, hd).transpose(0, 1)
    attn = torch.bmm(q * (hd ** -0.5), k.transpose(1, 2))
    if attn_mask is not None:
        attn = attn.masked_fill(attn_mask, float("-inf"))
    attn = F.softmax(attn, dim=-1)
    o = torch.bmm(attn, v).transpose(0, 1).reshape(Lq, B, E)
    return F.linear(o, out_w, out_b)


# ------------------------------------------------------------------------------- K5
def relative_position_index(ws: int) -> torch.Tensor:
    """Pair-wise relative position index inside a ws x ws window (backbone/swin.py:108-121)."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def window_attention(x, qkv_w, qkv_b, proj_w, proj_b, bias_table, ws, nheads, mask=None):
    """WindowAttention.forward (backbone/swin.py:131-171): x [nW*B, ws*ws, C]; q*scale @ k^T + relative
    position bias (+ shift mask [nW, N, N]) -> softmax -> @ v -> proj."""
    B_, N, C = x.shape
    hd = C // nheads
    qkv = F.linear(x, qkv_w, qkv_b).reshape(B_, N, 3, nheads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = bias_table[relative_position_index(ws).view(-1)].view(N, N, -1).permute(2, 0, 1).contiguous()
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, nheads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, nheads, N, N)
    attn = F.softmax(attn, dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return F.linear(out, proj_w, proj_b)


def shift_attn_mask(H, W, ws, shift):
    """SW-MSA mask of BasicLayer.forward (backbone/swin.py:413-440): 9-region id map on the padded grid,
    window-partitioned, pairwise difference -> -100 / 0.  Returns [nW, ws*ws, ws*ws]."""
    Hp, Wp = int(math.ceil(H / ws)) * ws, int(math.ceil(W / ws)) * ws
    img = torch.zeros(1, Hp, Wp, 1)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = img.view(1, Hp // ws, ws, Wp // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


# ------------------------------------------------------------------------------- misc
def position_embedding_sine(h, w, num_pos_feats, temperature=10000.0):
    """PositionEmbeddingSine(normalize=True) for an all-valid h x w map -> [2*num_pos_feats, h, w]
    (transformer_decoder/position_encoding.py:29-52)."""
    ones = torch.ones(1, h, w)
    y_embed = ones.cumsum(1, dtype=torch.get_default_dtype())
    x_embed = ones.cumsum(2, dtype=torch.get_default_dtype())
    eps, scale = 1e-6, 2 * math.pi
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.get_default_dtype())
    dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=4).flatten(3)
    pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=4).flatten(3)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)[0]


def attention_core(q, k, v, blocked=None):
    """softmax((q * hd^-0.5) k^T, blocked -> -inf) v for q [B,Q,nH,hd], k/v [B,S,nH,hd], blocked bool [B,Q,S]
    (True = not allowed to attend) -> [B,Q,nH*hd].  The arithmetic inside nn.MultiheadAttention
    (mask2former_transformer_decoder.py:106-118) between the in- and out-projections."""
    B, Q, nH, hd = q.shape
    qq = (q * hd ** -0.5).permute(0, 2, 1, 3)
    att = qq @ k.permute(0, 2, 3, 1)
    if blocked is not None:
        att = att.masked_fill(blocked[:, None], float("-inf"))
    att = F.softmax(att, dim=-1)
    return (att @ v.permute(0, 2, 1, 3)).permute(0, 2, 1, 3).reshape(B, Q, nH * hd)


def attn_mask_from_logits(mask_logits):
    """blocked = sigmoid(x) < 0.5, rows with every key blocked are un-blocked entirely
    (mask2former_transformer_decoder.py:486 and :433)."""
    blocked = mask_logits.sigmoid() < 0.5
    blocked[torch.where(blocked.sum(-1) == blocked.shape[-1])] = False
    return blocked


def gaussian_blur(score: torch.Tensor, kernel_size: int = 7, sigma: float = 1.0) -> torch.Tensor:
    """Optional smoothing of the anomaly map, ``transforms.GaussianBlur(7, sigma=1)`` at support.py:366-383.
    PARITY UNPINNED: the algorithm lives in torchvision (a dependency of the reference that is not installed here), so this
    restates its published definition -- 1-D kernel ``exp(-0.5 (x / sigma)^2)`` on ``linspace(-(k-1)/2, (k-1)/2, k)``
    normalised to sum 1, 2-D kernel = outer product, reflect padding by k // 2, depthwise conv2d
    (torchvision/transforms/_functional_tensor.py: ``_get_gaussian_kernel1d/2d``, ``gaussian_blur``) -- and is cross-checked
    against ``scipy.ndimage.gaussian_filter(mode="mirror", truncate=3)`` in tests/test_oracle_golden.py.  score [H,W] -> [H,W]."""
    half = (kernel_size - 1) * 0.5
    x = torch.linspace(-half, half, steps=kernel_size, dtype=score.dtype)
    pdf = torch.exp(-0.5 * (x / sigma) ** 2)
    k1 = pdf / pdf.sum()
    k2 = torch.mm(k1[:, None], k1[None, :])
    p = kernel_size // 2
    img = F.pad(score[None, None], (p, p, p, p), mode="reflect")
    return F.conv2d(img, k2[None, None].to(score.dtype))[0, 0]


def ood_components(score, threshold):
    """The open-set branch of MaskFormer.panoptic_inference (maskformer_model.py:454-474) on a score map [H,W]:
    ``binary = score > threshold`` -> cv2.morphologyEx(MORPH_OPEN, 3x3 ones) -> cv2.morphologyEx(MORPH_CLOSE, 3x3 ones) ->
    cv2.connectedComponents(connectivity=4).  PARITY UNPINNED: OpenCV is a dependency of the reference that is not installed here;
    this restates its documented semantics in plain numpy -- erosion / dilation with a 3x3 box and the default border
    (the operation's neutral value: out-of-image neighbours never veto an erosion nor trigger a dilation), labels numbered in raster
    order of each component's first pixel -- and is cross-checked against scipy.ndimage in tests/test_oracle_golden.py.
    Returns (labels int32 [H,W], 0 = background, 1..n, n)."""
    import numpy as np
    b = (np.asarray(score) > threshold)
    H, W = b.shape

    def morph(m, dilate):
        pad = np.full((H + 2, W + 2), not dilate, dtype=bool)          # neutral border
        pad[1:-1, 1:-1] = m
        out = np.full((H, W), not dilate, dtype=bool)
        for dy in range(3):
            for dx in range(3):
                win = pad[dy:dy + H, dx:dx + W]
                out = (out | win) if dilate else (out & win)
        return out

    b = morph(morph(b, False), True)                                   # opening
    b = morph(morph(b, True), False)                                   # closing
    labels = np.zeros((H, W), dtype=np.int32)
    n = 0
    for y in range(H):                                                 # flood fill in raster order (small maps only)
        for x in range(W):
            if b[y, x] and labels[y, x] == 0:
                n += 1
                stack = [(y, x)]
                labels[y, x] = n
                while stack:
                    cy, cx = stack.pop()
                    for ny, nx in ((cy - 1, cx), (cy + 1, cx), (cy, cx - 1), (cy, cx + 1)):
                        if 0 <= ny < H and 0 <= nx < W and b[ny, nx] and labels[ny, nx] == 0:
                            labels[ny, nx] = n
                            stack.append((ny, nx))
    return labels, n


def panoptic_inference(mask_cls, mask_pred, thing_classes, object_mask_threshold, overlap_threshold, open_panoptic=False,
                       ood_threshold=-0.1, pixel_min=300):
    """MaskFormer.panoptic_inference (maskformer_model.py:394-486) restated for CPU tensors: mask_cls [Q,K+1] logits,
    mask_pred [Q,H,W] logits -> (panoptic_seg int32 [H,W] numpy, segments_info, rba map).  The open-set branch goes through
    ``ood_components`` above (OpenCV semantics restated: parity unpinned for that step)."""
    import numpy as np
    K = mask_cls.shape[-1] - 1
    prob = F.softmax(mask_cls.double(), dim=-1)
    scores, labels = prob.max(-1)
    mp = mask_pred.double().sigmoid()
    keep = labels.ne(K) & (scores > object_mask_threshold)
    H, W = mask_pred.shape[-2:]
    pan = np.zeros((H, W), dtype=np.int32)
    info = []
    sem = torch.einsum("qc,qhw->chw", prob[:, :-1], mp)
    rba = -(sem.tanh()).sum(0)
    if int(keep.sum()) == 0:
        return pan, info, rba
    cs, cc, cm = scores[keep], labels[keep], mp[keep]
    ids = (cs.view(-1, 1, 1) * cm).argmax(0).numpy()
    cur, stuff = 0, {}
    for k in range(cc.shape[0]):
        cls = int(cc[k])
        won, solid = ids == k, cm[k].numpy() >= 0.5
        m = won & solid
        if won.sum() > 0 and solid.sum() > 0 and m.sum() > 0:
            if won.sum() / solid.sum() < overlap_threshold:
                continue
            if cls not in thing_classes:
                if cls in stuff:
                    pan[m] = stuff[cls]
                    continue
                stuff[cls] = cur + 1
            cur += 1
            pan[m] = cur
            info.append({"id": cur, "isthing": cls in thing_classes, "category_id": cls})
    if open_panoptic:
        lab, n = ood_components(rba.numpy(), ood_threshold)
        for i in range(1, n + 1):
            m = (lab == i) & (pan == 0)
            if m.sum() < pixel_min:
                continue
            cur += 1
            pan[m] = cur
            info.append({"id": cur, "isthing": True, "category_id": 255})
    return pan, info, rba
