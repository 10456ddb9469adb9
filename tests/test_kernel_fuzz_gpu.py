"""GPU: seeded random shapes through the kernels whose launch FORM depends on the shape ("shape coverage is hand-picked ... no randomised size
sweep").  Every case runs under the guard-banded, NaN-poisoned allocations of tests/_guard.py (conftest's canary fixture), so a launch form that writes outside its
tensors, or leaves part of an output unwritten, fails whatever the numbers say.

K6 (`ops.linear` -> rba_split_linear_*): the kernel family with the most forms -- direct / LDS-staged / pipelined 128 x 128 / 256 x 128 (row split) / K-split / sub-tile,
fp32 rows or split images in and out, GELU / ReLU / residual epilogues, both arithmetic modes, both stream hints -- chosen by (M, N, K) tile-count rules in
csrc/split_linear_dma.hip and split_linear_h3.h.  Shapes are drawn around those rules' thresholds (tile counts 32 / 64 / 128 / 160 / 256 / 512, M = 128 k +- 1, 128 * odd,
K = 32 ... 4096) plus uniform noise.  Reference = torch float64 on the device (a plain PyTorch reference of the same op); bound = the fp32-GEMM error model the fixed-shape
tests use (4e-5 sqrt(K / 256) |x| |w| scale).  The other shape-dependent launchers (3 x 3 convolution, row-complete token Linear, skinny Linear, channels-last GroupNorm and
resample, K1 in both forms) get the same treatment with fewer cases.
"""
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    from rba_amd import ops as o
    return o


def _err(a, b):
    return (a.double() - b.double()).abs().max().item()


def _lin(N, K, g, bias=True):
    lin = torch.nn.Linear(K, N, bias=bias).cuda()
    with torch.no_grad():
        lin.weight.copy_(torch.randn(N, K, generator=g, device="cuda") * K ** -0.5)
        if bias:
            lin.bias.copy_(torch.randn(N, generator=g, device="cuda"))
    return lin


def _linear_cases(seed, n):
    """(M, N, K) around the launch rules' thresholds"""
    rnd = random.Random(seed)
    cases = []
    while len(cases) < n:
        K = rnd.choice([32, 64, 96, 128, 160, 192, 256, 384, 512, 768, 1024, 1536, 2048, 4096])
        N = rnd.choice([16, 20, 48, 64, 96, 100, 128, 192, 256, 288, 384, 512, 768, 1024, 1536, 2048, 3072])
        nt = (N + 127) // 128
        kind = rnd.random()
        if kind < 0.45:                                    # a tile count right at a rule's threshold, +- one tile
            tiles = rnd.choice([32, 64, 128, 129, 160, 192, 255, 256, 257, 320, 511, 512, 768])
            mt = max(1, tiles // nt + rnd.choice([-1, 0, 0, 1]))
            M = 128 * mt + rnd.choice([0, 0, -1, 1, -37, 64])
        elif kind < 0.7:                                   # 128 * odd rows (the round-4 out-of-bounds case), whole images of a batch
            M = 128 * (2 * rnd.randint(0, 150) + 1) * rnd.choice([1, 1, 2])
        else:
            M = int(2 ** rnd.uniform(0, 15.3))
        M = max(1, M)
        if M * N * K > 6.0e10 or M * (N + K) > 1.2e8:      # keep a case under ~0.3 s and ~0.5 GB
            continue
        cases.append((M, N, K))
    return cases


@pytest.mark.parametrize("mode", ["f16x3", "bf16x6"])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_linear_random_shapes(ops, seed, mode):
    cases = _linear_cases(1000 * seed + (7 if mode == "bf16x6" else 0), 40 if mode == "f16x3" else 16)
    prev = ops.concurrent_streams()
    try:
        with ops.split_mode(mode):
            for ci, (M, N, K) in enumerate(cases):
                g = torch.Generator(device="cuda").manual_seed(seed * 100 + ci)
                x = torch.randn(M, K, generator=g, device="cuda") * 1.5 + 0.2
                lin = _lin(N, K, g)
                tol = 4e-5 * (K / 256) ** 0.5 + 4e-6
                ref = F.linear(x.double(), lin.weight.double(), lin.bias.double())
                for hint in (1, 3):
                    ops.set_concurrent_streams(hint)
                    what = (mode, M, N, K, hint)
                    y = ops.linear(x, lin)
                    assert y.shape == (M, N) and _err(y, ref) < tol * 3, what + ("plain", _err(y, ref))
                    act = ("gelu", "relu", "nobias")[(ci + hint) % 3]
                    if act == "gelu":
                        assert _err(ops.linear(x, lin, gelu=True), F.gelu(ref)) < tol * 3, what + (act,)
                    elif act == "relu":
                        assert _err(ops.linear(x, lin, relu=True), F.relu(ref)) < tol * 3, what + (act,)
                    else:
                        assert _err(ops.linear(x, lin, use_bias=False), ref - lin.bias.double()) < tol * 3, what + (act,)
                    if mode != "f16x3":
                        continue
                    if ops.linear_takes_split(M, N, K):                       # A operand handed over as the producer's split image
                        xs = ops.SplitActivations.pack(x)
                        ys = ops.linear(xs, lin)
                        assert _err(ys, ref) < tol * 3, what + ("split in",)
                        if ops.linear_takes_split(M, K, N):                    # fc1 -> fc2 hand-over: GELU output as a split image
                            got = ops.linear(xs, lin, gelu=True, split_out=True)
                            assert isinstance(got, ops.SplitActivations) and _err(got.unpack().view(M, N), F.gelu(ref)) < tol * 3 + 2.0 ** -20 * 8, what + ("split out",)
                    if ops.linear_residual_fused(M, N, K):
                        res = torch.randn(M, N, generator=g, device="cuda")
                        want = res.double() + ref
                        out = ops.linear(x, lin, residual=res)
                        assert out.data_ptr() == res.data_ptr() and _err(out, want) < tol * 3, what + ("residual",)
    finally:
        ops.set_concurrent_streams(prev)


@pytest.mark.parametrize("seed", [0, 1])
def test_conv3x3_nhwc_random_shapes(ops, seed):
    """the implicit-GEMM 3 x 3 convolution (fp32 rows in, both tile forms; split image in where it applies) on random B, H, W, C, N"""
    rnd = random.Random(seed)
    for ci in range(12):
        B, C, N = rnd.choice([1, 1, 2, 3]), rnd.choice([32, 64, 128, 256]), rnd.choice([32, 64, 100, 128, 256])
        H, W = rnd.randint(1, 150), rnd.randint(1, 220)
        if rnd.random() < 0.35:                                                  # H * W = 128 * odd
            H, W = 8 * rnd.choice([1, 3, 5, 11]), 16 * (2 * rnd.randint(0, 9) + 1)
        g = torch.Generator(device="cuda").manual_seed(seed * 50 + ci)
        x = torch.randn(B, H, W, C, generator=g, device="cuda")
        w = torch.randn(N, C, 3, 3, generator=g, device="cuda") * (9 * C) ** -0.5
        b = torch.randn(N, generator=g, device="cuda")
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
        tol = 4e-5 * (9 * C / 256) ** 0.5 + 4e-6
        for mode in ("f16x3", "bf16x6"):
            y = ops.conv3x3_nhwc(x, ops.conv3x3_weight(w, mode=mode), b, out_features=N)
            assert y.shape == (B, H, W, N) and _err(y, ref) < 3 * tol, (B, H, W, C, N, mode, _err(y, ref))
        if ops.conv3x3_takes_split(B * H * W, N):
            xs = ops.SplitActivations.pack(x.view(B * H * W, C))
            y = ops.conv3x3_nhwc(ops.SplitActivations(xs.data, (B, H, W, C)), ops.conv3x3_weight(w, mode="f16x3"), b, out_features=N)
            assert _err(y, ref) < 3 * tol, (B, H, W, C, N, "split in")


@pytest.mark.parametrize("seed", [0, 1])
def test_token_and_skinny_linear_random_shapes(ops, seed):
    rnd = random.Random(seed)
    for ci in range(20):
        g = torch.Generator(device="cuda").manual_seed(seed * 77 + ci)
        K = rnd.choice([32, 64, 256, 512, 1024])
        N = rnd.choice([16, 20, 96, 128, 192, 256])
        M = rnd.choice([1, 15, 16, 17, 100, 127, 128, 129, 460, 2048, 4830, 8192, rnd.randint(1, 9000)])
        x, pos = torch.randn(M, K, generator=g, device="cuda"), torch.randn(M, K, generator=g, device="cuda")
        lin = _lin(N, K, g)
        tol = 4e-5 * (K / 256) ** 0.5 + 4e-6
        if ops.token_linear_ok(N, K):
            ref = F.linear((x + pos).double(), lin.weight.double(), lin.bias.double())
            assert _err(ops.token_linear(x, lin, x_add=pos), ref) < 3 * tol, ("token", M, N, K)
            assert _err(ops.token_linear(x, lin, relu=True), F.relu(F.linear(x.double(), lin.weight.double(), lin.bias.double()))) < 3 * tol, ("token relu", M, N, K)
            if N % 16 == 0:
                norm = torch.nn.LayerNorm(N).cuda()
                with torch.no_grad():
                    norm.weight.copy_(torch.rand(N, generator=g, device="cuda") + 0.5)
                    norm.bias.copy_(torch.randn(N, generator=g, device="cuda"))
                res = torch.randn(M, N, generator=g, device="cuda")
                want = F.layer_norm(res.double() + F.linear(x.double(), lin.weight.double(), lin.bias.double()), (N,), norm.weight.double(), norm.bias.double(), norm.eps)
                assert _err(ops.token_linear(x, lin, residual=res, norm=norm), want) < 2e-5 + 3 * tol, ("token LN", M, N, K)
        if M <= 128:
            ref = F.linear(x.double(), lin.weight.double(), lin.bias.double())
            assert _err(ops.skinny_linear(x, lin.weight, lin.bias), ref) < 2e-5 * (K / 256) ** 0.5 + 2e-6, ("skinny", M, N, K)


@pytest.mark.parametrize("seed", [0, 1])
def test_channels_last_norm_and_resample_random_shapes(ops, seed):
    rnd = random.Random(seed)
    for ci in range(16):
        g = torch.Generator(device="cuda").manual_seed(seed * 31 + ci)
        C = rnd.choice([32, 64, 128, 256])
        G = rnd.choice([d for d in (4, 8, 32) if C % d == 0 and C // d >= 4])
        B, P = rnd.choice([1, 2, 3]), rnd.choice([1, 5, 127, 128, 129, 1000, 4097, 128 * 209, rnd.randint(1, 40000)])
        x = torch.randn(B, P, C, generator=g, device="cuda") * 2 + 0.7
        w, b = torch.randn(C, generator=g, device="cuda"), torch.randn(C, generator=g, device="cuda")
        relu = bool(ci & 1)
        ref = F.group_norm(x.double().permute(0, 2, 1), G, w.double(), b.double(), 1e-5).permute(0, 2, 1)
        ref = F.relu(ref) if relu else ref
        assert _err(ops.group_norm_nhwc(x, G, w, b, 1e-5, relu=relu), ref) < 3e-5, ("gn", B, P, C, G)
        h, wd = rnd.randint(1, 60), rnd.randint(1, 90)
        H, W = rnd.choice([(2 * h, 2 * wd), (rnd.randint(1, 130), rnd.randint(1, 190)), (4 * h, 4 * wd)])
        src = torch.randn(h, wd, C, generator=g, device="cuda")
        add = torch.randn(H, W, C, generator=g, device="cuda") if ci % 3 else None
        # the kernel reproduces ATen's fp32 tap arithmetic (scale * (dst + 0.5) - 0.5 in fp32): against ATen's own fp32 result the difference is fma
        # contraction only; against float64 both carry the fp32 rounding of the source coordinate (~1e-7 relative of a coordinate up to ~200)
        aten = F.interpolate(src.permute(2, 0, 1)[None], size=(H, W), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
        want = F.interpolate(src.permute(2, 0, 1)[None].double(), size=(H, W), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
        if add is not None:
            aten, want = aten + add, want + add.double()
        got = ops.resample_bilinear_nhwc(src, (H, W), add=add)
        assert _err(got, aten) < 4e-6 and _err(got, want) < 1e-4, ("resample", h, wd, H, W, C, _err(got, aten), _err(got, want))


@pytest.mark.parametrize("seed", [0, 1])
def test_k1_random_shapes(ops, seed):
    """K1 in both forms (full-resolution planes; fused x4 up-sample + crop) on random Q, K, H, W with every output requested: the reduction's tile hand-out,
    tail tiles and crops"""
    from oracle import ref_ops
    rnd = random.Random(seed)
    for ci in range(8):
        g = torch.Generator().manual_seed(seed * 13 + ci)
        Q, K = rnd.choice([1, 7, 16, 50, 100]), rnd.choice([1, 5, 19, 20, 32])
        H, W = rnd.randint(1, 90), rnd.randint(1, 300)
        mp = torch.randn(Q, H, W, generator=g) * 5
        prob = F.softmax(torch.randn(Q, K + 1, generator=g) * 3, -1)[:, :-1].contiguous()
        sem_r, rba_r, _ = ref_ops.rba_reduce_ordered(mp, prob)
        rba, sem, arg = ops.rba_reduce(mp.cuda(), prob.cuda(), True, True)
        assert _err(sem.cpu(), sem_r) < 5e-6 and _err(rba.cpu(), rba_r) < 2e-5, ("k1", Q, K, H, W)
        if K >= 2:
            top2 = sem_r.topk(2, dim=0).values
            assert int(((arg.cpu().long() != sem_r.argmax(0)) & ((top2[0] - top2[1]) > 1e-5)).sum()) == 0
        h, w = rnd.randint(1, 40), rnd.randint(1, 70)
        ch, cw = rnd.randint(max(1, 4 * h - 31), 4 * h), rnd.randint(max(1, 4 * w - 31), 4 * w)
        low = torch.randn(Q, h, w, generator=g) * 5
        up = ref_ops.upsample_bilinear(low[None], (4 * h, 4 * w))[0]
        sem_u, rba_u, _ = ref_ops.rba_reduce_ordered(up, prob)
        r2, s2, a2 = ops.rba_reduce_up4(low.cuda(), prob.cuda(), (ch, cw), True, True)
        assert _err(s2.cpu(), sem_u[:, :ch, :cw]) < 1e-5 and _err(r2.cpu(), rba_u[:ch, :cw]) < 2e-5, ("k1 up4", Q, K, h, w, ch, cw)
