"""Algorithmic cost of one image through the hot path, stage by stage: fp32-equivalent flops (2 per multiply-add) and the bytes
each operator cannot avoid -- every input tensor, weight and output tensor of an operator counted ONCE at fp32 (4 B), operator by
operator as the reference lists them (mask2former/modeling/backbone/swin.py:131-171, 235-295, 311-337;
pixel_decoder/msdeformattn.py:101-140, 323-367; transformer_decoder/mask2former_transformer_decoder.py:398-489;
maskformer_model.py:294-299, 381-386; evaluate_ood.py:150).  No credit is taken for fusion (a fused kernel moves fewer bytes than
this table says: the table is the yardstick, not the traffic) and none for the reference's copies (window partition / roll / pad
copies and layout permutes are NOT counted: they are not algorithmic).

bench.py divides these totals by the measured time per image: `roofline_e2e` in its JSON line = achieved GB/s against the 8 TB/s
HBM peak and achieved fp32-equivalent TFLOP/s against the f16x3 ceiling (2 500 / 3 TFLOP/s: three f16 matrix-pipe products per fp32
product) -- SURVEY.md 8(d).  DESIGN.md section 5 prints the table for BASELINE configs[1].
"""
from .arch import FEATURE_NAMES, FEATURE_STRIDES, complete, feature_channels, num_fpn_levels

F32 = 4


def _ceil_div(a, b):
    return -(-a // b)


def stages(arch, h, w):
    """-> list of (stage name, flops, bytes) for ONE image of h x w pixels."""
    a = complete(arch)
    if a.get("resnet"):
        raise NotImplementedError("cost model covers the Swin configurations (BASELINE configs[1..4])")
    d = a["size_divisibility"] if a["size_divisibility"] > 0 else 32
    H, W = _ceil_div(h, d) * d, _ceil_div(w, d) * d
    out = []

    def add(name, flops, nbytes):
        out.append((name, float(flops), float(nbytes)))

    def linear(T, K, N, bias=True):
        return 2.0 * T * K * N, F32 * (T * K + N * K + T * N + (N if bias else 0))

    def ln(T, C):
        return 8.0 * T * C, F32 * (2 * T * C + 2 * C)

    # ---- front end: normalise + pad (u8 in, fp32 out), patch embedding 4x4/4 + LayerNorm
    E, ws = a["embed_dim"], a["window_size"]
    Wh, Ww = H // 4, W // 4
    T = Wh * Ww
    f, b = linear(T, 48, E)
    add("normalise + pad + patch embed", 2.0 * 3 * H * W + f + ln(T, E)[0], 3 * h * w + F32 * 3 * H * W + b + ln(T, E)[1])
    # ---- Swin stages
    N = ws * ws
    for i, depth in enumerate(a["depths"]):
        C = E * 2 ** i
        nH = a["num_heads"][i]
        hid = int(C * a["mlp_ratio"])
        Tw = _ceil_div(Wh, ws) * _ceil_div(Ww, ws) * N               # tokens incl. window padding (attention runs on padded windows)
        fl = by = 0.0
        for _ in range(depth):
            for f_, b_ in (ln(T, C), linear(T, C, 3 * C), linear(T, C, C), ln(T, C), linear(T, C, hid), linear(T, hid, C)):
                fl += f_
                by += b_
            fl += 4.0 * Tw * N * C + 5.0 * Tw * N * nH                  # QK^T and PV (2 * Tw * N * C each), softmax + bias + mask
            by += F32 * (3 * T * C + T * C + nH * N * N)                 # qkv in, attention out, relative-position bias
            fl += 2.0 * T * C + 8.0 * T * hid                           # two residual adds, GELU
            by += F32 * 2 * T * C                                        # the residual operand of each add
        add(f"swin stage {i + 1} ({depth} blocks, C={C}, {T} tokens)", fl, by)
        f_, b_ = ln(T, C)                                               # out norm of the stage
        fl2, by2 = f_, b_
        if i < len(a["depths"]) - 1:                                    # patch merging: LN over 4C, Linear 4C -> 2C
            T2 = _ceil_div(Wh, 2) * _ceil_div(Ww, 2)
            f1, b1 = ln(T2, 4 * C)
            f2, b2 = linear(T2, 4 * C, 2 * C, bias=False)
            fl2 += f1 + f2
            by2 += b1 + b2
            Wh, Ww, T = _ceil_div(Wh, 2), _ceil_div(Ww, 2), T2
        add(f"swin stage {i + 1} output norm" + (" + patch merging" if i < len(a["depths"]) - 1 else ""), fl2, by2)
    # ---- pixel decoder
    dm, md, M, P = a["conv_dim"], a["mask_dim"], a["nheads"], a["enc_points"]
    chans = feature_channels(a)
    sizes = {f_: (H // FEATURE_STRIDES[f_], W // FEATURE_STRIDES[f_]) for f_ in FEATURE_NAMES}
    L = len(a["enc_in"])
    S = sum(sizes[f_][0] * sizes[f_][1] for f_ in a["enc_in"])
    fl = by = 0.0
    for f_ in a["enc_in"]:
        t = sizes[f_][0] * sizes[f_][1]
        f1, b1 = linear(t, chans[f_], dm)
        fl += f1 + 10.0 * t * dm
        by += b1 + F32 * 2 * t * dm                                      # 1x1 conv + GroupNorm
    add(f"encoder input projections ({L} level(s), {S} tokens)", fl, by)
    dff = a["enc_dim_feedforward"]
    fl = by = 0.0
    for _ in range(a["enc_layers"]):
        for f1, b1 in (linear(S, dm, M * L * P * 3), linear(S, dm, dm), linear(S, dm, dm), ln(S, dm), linear(S, dm, dff), linear(S, dff, dm),
                       ln(S, dm)):
            fl += f1
            by += b1
        fl += S * dm + S * M * L * P * 12.0 + 2.0 * S * M * L * P * 4 * (dm // M) + 3.0 * S * dm + S * dff
        # + pos add, sampling locations + softmax, 4 bilinear taps x L*P samples per (query, head), residual adds, ReLU
        by += F32 * (S * dm + 3 * S * M * L * P + 2 * S * dm + 2 * S * dm)   # value + (loc, weights) in, output; the two residual operands
    add(f"MSDeformAttn encoder ({a['enc_layers']} layers)", fl, by)
    fl = by = 0.0
    nf = num_fpn_levels(a)
    for j in range(nf, 0, -1):
        fn = FEATURE_NAMES[j - 1]
        t = sizes[fn][0] * sizes[fn][1]
        f1, b1 = linear(t, chans[fn], dm, bias=False)                    # lateral 1x1 + GN
        f2, b2 = linear(t, 9 * dm, dm, bias=False)                       # 3x3 conv as its GEMM; its input is read once
        fl += f1 + f2 + 2 * 10.0 * t * dm + 9.0 * t * dm                 # two GroupNorms (+ReLU), bilinear upsample + add
        by += b1 + (b2 - F32 * t * 9 * dm + F32 * t * dm) + 2 * F32 * 2 * t * dm + F32 * (t * dm // 4 + 2 * t * dm)
    t4 = sizes["res2"][0] * sizes["res2"][1]
    f1, b1 = linear(t4, dm, md)
    add(f"FPN ({nf} level(s)) + mask features", fl + f1, by + b1)
    # ---- masked-attention decoder
    Q, K, dffd = a["num_queries"], a["num_classes"], a["dim_feedforward"]
    lvl = [sizes[f_] for f_ in a["enc_in"][::-1]]                         # coarse -> fine, the order the decoder cycles through
    fl = by = 0.0

    def head(cols, want_masks):
        f_, b_ = ln(Q, dm)
        for f1, b1 in (linear(Q, dm, K + 1), linear(Q, dm, dm), linear(Q, dm, dm), linear(Q, dm, md)):
            f_ += f1
            b_ += b1
        f_ += 2.0 * Q * md * cols
        b_ += F32 * (md * cols + Q * md + Q * cols)
        return f_, b_
    for i in range(a["dec_layers"]):
        s = lvl[i % L][0] * lvl[i % L][1]
        f1, b1 = head(4 * s, False)                                       # the head that feeds this layer's attention mask (2x2 source pixels per cell)
        fl += f1
        by += b1
        for f1, b1 in (linear(Q, dm, dm), linear(s, dm, dm), linear(s, dm, dm), linear(Q, dm, dm), ln(Q, dm),           # cross attention
                       linear(Q, dm, dm), linear(Q, dm, dm), linear(Q, dm, dm), linear(Q, dm, dm), ln(Q, dm),           # self attention
                       linear(Q, dm, dffd), linear(Q, dffd, dm), ln(Q, dm)):                                            # FFN
            fl += f1
            by += b1
        fl += 4.0 * Q * s * dm + 6.0 * Q * s * M + 4.0 * Q * Q * dm      # QK^T + PV over s keys (+ mask, softmax), self attention
        by += F32 * (Q * dm + 2 * s * dm + Q * s + Q * dm) + F32 * 4 * Q * dm
    f1, b1 = head(t4, True)                                               # final head: class logits + full mask logits
    add(f"masked-attention decoder ({a['dec_layers']} layer(s)) + prediction heads", fl + f1, by + b1)
    # ---- post-network: x4 bilinear upsample of the mask logits, semantic inference + RbA
    add("x4 mask upsample", 9.0 * Q * H * W, F32 * (Q * t4 + Q * H * W))
    add("K1: sigmoid, class contraction, tanh, sum (RbA)", 2.0 * Q * K * H * W + 4.0 * Q * H * W + 2.0 * K * H * W,
        F32 * (Q * H * W + Q * K + H * W))
    return out


def totals(arch, h, w):
    st = stages(arch, h, w)
    return {"flops": sum(s[1] for s in st), "bytes": sum(s[2] for s in st), "stages": st}


def table(arch, h, w):
    """markdown rows for DESIGN.md"""
    st = stages(arch, h, w)
    tf, tb = sum(s[1] for s in st), sum(s[2] for s in st)
    rows = ["| stage | GFLOP (fp32-equivalent) | MB (algorithmic) |", "|---|---:|---:|"]
    rows += [f"| {n} | {f_ / 1e9:.1f} | {b_ / 1e6:.1f} |" for n, f_, b_ in st]
    rows.append(f"| **total** | **{tf / 1e9:.1f}** | **{tb / 1e6:.1f}** |")
    return "\n".join(rows)


if __name__ == "__main__":
    import sys
    from .arch import ARCHS
    name = sys.argv[1] if len(sys.argv) > 1 else "swin_b_1dl"
    hh, ww = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1024, 2048)
    print(table(ARCHS[name], hh, ww))
