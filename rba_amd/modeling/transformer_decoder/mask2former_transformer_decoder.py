"""Masked-attention transformer decoder, inference only (reference:
transformer_decoder/mask2former_transformer_decoder.py:232-502, post-norm, dropout 0).  Same parameter names.

MI355X dataflow: memory is kept batch-first [B, S, C]; the cross/self attention cores are the HIP kernel K3 with
the ``sigmoid(mask) < 0.5`` threshold and the all-masked-row fix fused in (the bool [B*h, Q, S] mask of the
reference is never materialised); mask logits are the HIP kernel K4; the bilinear down-sample that feeds the
attention mask is the HIP resampler.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...lru import ShapeCache
from ...registry import TRANSFORMER_DECODER_REGISTRY
from .position_encoding import PositionEmbeddingSine


def _qlinear(x, weight, bias=None, relu=False, x_add=None):
    """Linear layer on the query tensor [B, Q, C]: the skinny exact-fp32 MFMA kernel, over blocks of at most 128 rows (one launch for a single image's
    100 queries; a batch of B images takes ceil(100 B / 128) launches -- round 5: no library GEMM for B >= 2 either).
    x_add: the Linear runs on x + x_add (the query position embedding), added inside the kernel."""
    K = x.shape[-1]
    if K % 32:
        raise ops.RbaHipError("query Linears need K % 32 == 0 (hidden_dim 256 in every released architecture)")
    M = x.numel() // K
    x2 = x.contiguous().view(M, K)
    a2 = None if x_add is None else x_add.contiguous().view(M, K)
    if M <= 128:
        return ops.skinny_linear(x2, weight, bias, relu, x_add=a2).view(tuple(x.shape[:-1]) + (weight.shape[0],))
    # more than 128 rows (a batch of B >= 2 images): 128-row blocks, every launch writing ITS rows of the one output tensor (no per-block allocation + copy)
    out = torch.empty((M, weight.shape[0]), dtype=torch.float32, device=x.device)
    for r0 in range(0, M, 128):
        ops.skinny_linear(x2[r0:r0 + 128], weight, bias, relu, x_add=None if a2 is None else a2[r0:r0 + 128], out=out[r0:r0 + 128])
    return out.view(tuple(x.shape[:-1]) + (weight.shape[0],))


class _MHAParams(nn.Module):
    """Parameter holder with nn.MultiheadAttention's names: in_proj_weight, in_proj_bias, out_proj.{weight,bias}."""

    def __init__(self, d_model, nhead):
        super().__init__()
        self.embed_dim, self.num_heads = d_model, nhead
        self.in_proj_weight = nn.Parameter(torch.zeros(3 * d_model, d_model))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d_model))
        self.out_proj = nn.Linear(d_model, d_model)

    def _kv_views(self):
        """the key / value rows of in_proj as Linear views (their packed weight images are cached on the views)"""
        w, b = self.in_proj_weight, self.in_proj_bias
        key = (w.data_ptr(), w._version, b.data_ptr(), b._version)
        c = getattr(self, "_rba_kv", None)
        if c is None or c[0] != key:
            from types import SimpleNamespace
            E = self.embed_dim
            c = self._rba_kv = (key, SimpleNamespace(weight=w[E:2 * E], bias=b[E:2 * E]), SimpleNamespace(weight=w[2 * E:], bias=b[2 * E:]))
        return c[1], c[2]

    def forward(self, query, key, value, mask_logits=None, key_add=None, query_add=None):
        """batch-first: query [B,Q,E], key/value [B,S,E]; mask_logits [B,Q,S] (blocked iff sigmoid < 0.5).  key_add / query_add: the keys /
        queries are ``key + key_add`` / ``query + query_add`` (position embeddings, :48-58, :106-118), added inside the projections."""
        E, nH = self.embed_dim, self.num_heads
        B, Q, _ = query.shape
        S = key.shape[1]
        w, b = self.in_proj_weight, self.in_proj_bias
        if (query is key and key is value and key_add is query_add and B * Q <= 128 and E % 32 == 0 and E % 16 == 0
                and w.is_contiguous() and query.is_contiguous()):
            # self-attention: q = W_q (tgt + pos), k = W_k (tgt + pos), v = W_v tgt -- ONE launch over the stacked in_proj weight (was: an add
            # and three launches); q, k, v come back separately contiguous
            q, k, v = ops.skinny_linear(query, w, b, x_add=None if query_add is None else query_add.contiguous(), add_cols=2 * E, segments=3)
            o = ops.masked_xattn(q.view(B, Q, nH, E // nH), k.view(B, S, nH, E // nH), v.view(B, S, nH, E // nH), mask_logits)
            return _qlinear(o, self.out_proj.weight)
        q = _qlinear(query, w[:E], b[:E], x_add=query_add).view(B, Q, nH, E // nH)
        if (key is value and B * S > 128 and key.is_contiguous() and ops.token_linear_pays(B * S, E, E)
                and (key_add is None or (key_add.is_contiguous() and tuple(key_add.shape) == tuple(key.shape)))):
            # cross-attention: k = W_k (memory + pos) + b_k and v = W_v memory + b_v in ONE launch of the row-complete kernel (was: an add and two
            # library GEMMs per layer)
            lk, lv = self._kv_views()
            k, v = ops.token_linear_multi(key, [(lk, key_add, None, 0, False), (lv, None, None, 0, False)])
            o = ops.masked_xattn(q, k.view(B, S, nH, E // nH), v.view(B, S, nH, E // nH), mask_logits)
            return _qlinear(o, self.out_proj.weight)
        if key_add is not None:
            key = key + key_add
        if B * S > 128:
            # a memory level beyond the row-complete kernel's range (C5's 14 400-token level): the key / value projections are ordinary token
            # Linears -> K6 where it pays (ops.linear decides; the packed planes are cached on the views), not a library GEMM
            lk, lv = self._kv_views()
            k = ops.linear(key.contiguous(), lk).view(B, S, nH, E // nH)
            v = ops.linear(value.contiguous(), lv).view(B, S, nH, E // nH)
            o = ops.masked_xattn(q, k, v, mask_logits)
            return _qlinear(o, self.out_proj.weight)
        # self-attention: key / value are the (<= 128) queries themselves -> the skinny kernel too (hipBLASLt takes 33 us for 100 x 256 x 256)
        k = _qlinear(key, w[E:2 * E], b[E:2 * E]).view(B, S, nH, E // nH)
        v = _qlinear(value, w[2 * E:], b[2 * E:]).view(B, S, nH, E // nH)
        o = ops.masked_xattn(q, k, v, mask_logits)
        return _qlinear(o, self.out_proj.weight)            # out_proj.bias is added inside the caller's fused add+LN


class SelfAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead):
        super().__init__()
        self.self_attn = _MHAParams(d_model, nhead)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, tgt, query_pos):
        tgt = tgt.contiguous()
        t2 = self.self_attn(tgt, tgt, tgt, key_add=query_pos, query_add=query_pos)      # q = k = tgt + query_pos, v = tgt
        return ops.add_layer_norm(tgt.contiguous(), self.norm.weight, self.norm.bias, self.norm.eps, t2,
                                  self.self_attn.out_proj.bias)[1]          # forward_post :48-58


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead):
        super().__init__()
        self.multihead_attn = _MHAParams(d_model, nhead)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, tgt, memory, mask_logits, pos, query_pos):
        t2 = self.multihead_attn(tgt, memory, memory, mask_logits, key_add=pos, query_add=query_pos)
        return ops.add_layer_norm(tgt.contiguous(), self.norm.weight, self.norm.bias, self.norm.eps, t2,
                                  self.multihead_attn.out_proj.bias)[1]     # forward_post :106-118


class FFNLayer(nn.Module):
    def __init__(self, d_model, dim_feedforward):
        super().__init__()
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, tgt):
        t2 = _qlinear(_qlinear(tgt, self.linear1.weight, self.linear1.bias, relu=True), self.linear2.weight)
        return ops.add_layer_norm(tgt.contiguous(), self.norm.weight, self.norm.bias, self.norm.eps, t2,
                                  self.linear2.bias)[1]                     # forward_post :171-175


class MLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = _qlinear(x, layer.weight, layer.bias, relu=i < self.num_layers - 1)
        return x


class BNReluConv(nn.Module):
    """BatchNorm2d -> ReLU -> Conv2d(k=1) with the reference's sub-module names (norm / conv; reference :216-230), evaluated in
    inference mode by one HIP kernel: the running statistics are folded into a per-channel scale and shift."""

    def __init__(self, num_maps_in, num_maps_out):
        super().__init__()
        self.norm = nn.BatchNorm2d(num_maps_in, momentum=0.01)
        self.conv = nn.Conv2d(num_maps_in, num_maps_out, kernel_size=1, bias=True)

    def forward(self, x):
        n = self.norm
        scale = n.weight * torch.rsqrt(n.running_var + n.eps)
        shift = n.bias - n.running_mean * scale
        return ops.bn_relu_conv1x1(x.contiguous(), scale.contiguous(), shift.contiguous(),
                                   self.conv.weight.flatten(1).contiguous(), self.conv.bias)


@TRANSFORMER_DECODER_REGISTRY.register()
class MultiScaleMaskedTransformerDecoder(nn.Module):
    def __init__(self, arch):
        super().__init__()
        a = arch
        d = a["conv_dim"]
        self.num_heads, self.num_layers = a["nheads"], a["dec_layers"]
        self.num_queries, self.num_feature_levels = a["num_queries"], len(a["enc_in"])
        self.pe_layer = PositionEmbeddingSine(d // 2, normalize=True)
        self.transformer_self_attention_layers = nn.ModuleList(SelfAttentionLayer(d, a["nheads"]) for _ in range(self.num_layers))
        self.transformer_cross_attention_layers = nn.ModuleList(CrossAttentionLayer(d, a["nheads"]) for _ in range(self.num_layers))
        self.transformer_ffn_layers = nn.ModuleList(FFNLayer(d, a["dim_feedforward"]) for _ in range(self.num_layers))
        self.decoder_norm = nn.LayerNorm(d)
        self.query_feat = nn.Embedding(a["num_queries"], d)
        self.query_embed = nn.Embedding(a["num_queries"], d)
        self.level_embed = nn.Embedding(self.num_feature_levels, d)
        self.class_embed = nn.Linear(d, a["num_classes"] + 1)
        self.mask_embed = MLP(d, d, a["mask_dim"], 3)
        self.ood_prediction = bool(a.get("dense_hybrid", False))
        if self.ood_prediction:                                  # DenseHybrid head (reference :365-366)
            self.ood_pred = BNReluConv(d, 2)
        # inference-only shortcut, exact: intermediate heads evaluate mask logits only where the attention mask samples them
        # (aux_outputs then carry pred_logits only)
        self.sparse_intermediate_heads = True
        self.cache_initial_heads = True          # the heads of the un-decoded queries are constants of the checkpoint (_initial_query_side)
        self._plan_cache = ShapeCache(8)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # v1 checkpoints call query_feat "static_query" (reference :237-258)
        for k in list(state_dict.keys()):
            if k.startswith(prefix) and "static_query" in k:
                state_dict[k.replace("static_query", "query_feat")] = state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _sparse_plan(self, feat_hw, target_hw, device):
        """For an even integer down-sampling factor f the align_corners=False bilinear sample of target cell (y, x) is
        0.5*(0.5*v[r0,c0] + 0.5*v[r0,c1]) + 0.5*(0.5*v[r1,c0] + 0.5*v[r1,c1]) with r0 = f*y + f/2 - 1, r1 = r0 + 1 (same for
        columns): exactly 2x2 source pixels.  Returns the flat source indices [4, h*w] or None when the factor is not an
        even integer (then the dense path is used)."""
        key = (tuple(feat_hw), tuple(target_hw), device)

        def build():
            (H, W), (h, w) = feat_hw, target_hw
            if not (h > 0 and w > 0 and H % h == 0 and W % w == 0 and (H // h) % 2 == 0 and (W // w) % 2 == 0 and 4 * h * w < H * W):
                return False
            fy, fx = H // h, W // w
            r0 = torch.arange(h, device=device) * fy + fy // 2 - 1
            c0 = torch.arange(w, device=device) * fx + fx // 2 - 1
            idx = [((r0 + dr)[:, None] * W + (c0 + dc)[None, :]).reshape(-1) for dr in (0, 1) for dc in (0, 1)]
            return torch.stack(idx).contiguous()

        plan = self._plan_cache.get(key, build)
        return None if plan is False else plan

    def _query_side_heads(self, output):
        """decoder_norm -> class_embed, mask_embed MLP on the query tensor [B,Q,C] (reference :473-476): (class logits [B,Q,K+1], mask embedding [B,Q,md])"""
        dec = ops.add_layer_norm(output.contiguous(), self.decoder_norm.weight, self.decoder_norm.bias, self.decoder_norm.eps)[1]
        return _qlinear(dec, self.class_embed.weight, self.class_embed.bias), self.mask_embed(dec).contiguous()

    def _initial_query_side(self, B, device):
        """The prediction heads BEFORE the first layer (reference :427-430) see `query_feat.weight` -- learnt parameters, not the image: decoder_norm, class_embed
        and the three mask_embed Linears of that call are constants of the checkpoint.  Computed once per (weights version, batch size) with the same kernels
        (bit-identical to evaluating them per image) and kept: five launches and the query tensor's expand + copy less per image (round 6)."""
        ps = [self.query_feat.weight, self.decoder_norm.weight, self.decoder_norm.bias, self.class_embed.weight, self.class_embed.bias]
        for ly in self.mask_embed.layers:
            ps += [ly.weight, ly.bias]
        key = (B, device, tuple((p_.data_ptr(), p_._version) for p_ in ps))
        c = self.__dict__.get("_rba_heads0")
        if c is None or c[0] != key:
            if torch.cuda.is_current_stream_capturing():
                return None                                           # never build a cache entry inside a capture: evaluate in line instead
            output = self.query_feat.weight[None].expand(B, -1, -1).contiguous()
            side = self._query_side_heads(output)
            torch.cuda.current_stream(device).synchronize()          # once per checkpoint: forwards on OTHER streams read these tensors without an event
            c = self.__dict__["_rba_heads0"] = (key, output, side)
        return c[1], c[2]

    def forward_prediction_heads(self, output, mask_features, attn_mask_target_size, need_attn_mask=True, need_masks=True,
                                 gathered=None, query_side=None):
        """output [B,Q,C] -> class logits [B,Q,K+1], mask logits [B,Q,H/4,W/4] (None unless need_masks), attention-mask
        logits [B,Q,h*w] (reference :472-489; the threshold itself happens inside K3).  When only the attention mask is
        consumed (every call but the last) the mask logits are evaluated just at the 2x2 source pixels each attention
        cell interpolates -- the same arithmetic on 4*h*w instead of H*W/16 columns."""
        if query_side is not None:
            outputs_class, mask_embed = query_side
        else:
            outputs_class, mask_embed = self._query_side_heads(output)
        plan = None
        if need_attn_mask and not need_masks and self.sparse_intermediate_heads:
            plan = self._sparse_plan(mask_features.shape[-2:], attn_mask_target_size, mask_features.device)
        if plan is not None:
            B, C = mask_features.shape[:2]
            lvl = tuple(attn_mask_target_size)
            gathered = {} if gathered is None else gathered    # per-call dict owned by forward(): the module stays re-entrant
            if lvl not in gathered:                   # mask_features is the same tensor for every layer: gather once per level
                gathered[lvl] = mask_features.flatten(2).index_select(2, plan.reshape(-1)).contiguous()   # [B,C,4hw]
            cols = gathered[lvl]
            v = ops.mask_logits(mask_embed, cols).view(B, -1, 4, plan.shape[1])
            # = 0.5 * (0.5 * v0 + 0.5 * v1) + 0.5 * (0.5 * v2 + 0.5 * v3), the bilinear sample at the centre of a 2 x 2 cell: scaling by a power
            # of two is exact, so the factors can be collected without changing a bit -- ((v0 + v1) + (v2 + v3)) * 0.25 in one launch
            return outputs_class, None, ops.quad_mean(v)
        outputs_mask = ops.mask_logits(mask_embed, mask_features)
        attn_logits = None
        if need_attn_mask:
            attn_logits = ops.resample_bilinear(outputs_mask, attn_mask_target_size).flatten(2)
        return outputs_class, outputs_mask, attn_logits

    def _level_pos(self, x, B):
        """sine position embedding of a level as tokens [B, S, C] (contiguous: it is the key projection's `x_add` operand), per (shape, batch)"""
        cache = self.__dict__.setdefault("_pos_tok_cache", ShapeCache(8))
        key = (tuple(x.shape[-2:]), B, x.device)
        return cache.get(key, lambda: self.pe_layer(x).flatten(2).transpose(1, 2).expand(B, -1, -1).contiguous())

    def forward(self, x, mask_features, mask=None):
        """x: list of [B,C,h_l,w_l]; mask_features [B,md,H/4,W/4] -> dict(pred_logits, pred_masks, aux_outputs)
        (reference :398-470)."""
        assert len(x) == self.num_feature_levels
        del mask
        src, pos, size_list = [], [], []
        B = x[0].shape[0]
        for i in range(self.num_feature_levels):
            size_list.append(tuple(int(v) for v in x[i].shape[-2:]))
            pos.append(self._level_pos(x[i], B))                                              # [B,S,C], cached per shape
            t = x[i].permute(0, 2, 3, 1)                                                      # channels-last views (pixel decoder): tokens without a copy
            tok = t.reshape(B, -1, x[i].shape[1]) if t.is_contiguous() else x[i].flatten(2).transpose(1, 2)
            src.append(tok + self.level_embed.weight[i])                                      # [B,S,C] contiguous (:424-426)
        query_embed = self.query_embed.weight[None].expand(B, -1, -1)
        first = self._initial_query_side(B, mask_features.device) if self.cache_initial_heads else None
        if first is not None:
            output, side0 = first                                     # constants of the checkpoint (never written: every layer returns new tensors)
        else:
            output, side0 = self.query_feat.weight[None].expand(B, -1, -1).contiguous(), None
        mask_features = mask_features.contiguous()
        predictions_class, predictions_mask = [], []
        gathered = {}
        cls, msk, attn_logits = self.forward_prediction_heads(output, mask_features, size_list[0], self.num_layers > 0,
                                                              need_masks=self.num_layers == 0, gathered=gathered, query_side=side0)
        predictions_class.append(cls)
        predictions_mask.append(msk)
        for i in range(self.num_layers):
            li = i % self.num_feature_levels
            output = self.transformer_cross_attention_layers[i](output, src[li], attn_logits, pos[li], query_embed)   # memory + pos: inside the key projection
            output = self.transformer_self_attention_layers[i](output, query_embed)
            output = self.transformer_ffn_layers[i](output)
            last = i == self.num_layers - 1
            cls, msk, attn_logits = self.forward_prediction_heads(
                output, mask_features, size_list[(i + 1) % self.num_feature_levels], need_attn_mask=not last, need_masks=last,
                gathered=gathered)
            predictions_class.append(cls)
            predictions_mask.append(msk)
        out = {
            "pred_logits": predictions_class[-1],
            "pred_masks": predictions_mask[-1],
            "aux_outputs": [({"pred_logits": a, "pred_masks": b} if b is not None else {"pred_logits": a})
                            for a, b in zip(predictions_class[:-1], predictions_mask[:-1])],
        }
        if self.ood_prediction:
            out["ood_pred"] = self.ood_pred(mask_features)                           # reference :467-468
        return out
