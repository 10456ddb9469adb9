import torch, sys
sys.path.insert(0, ".")
import torch.nn.functional as F
from rba_amd import ops
g = torch.Generator().manual_seed(1 + 131072 + 256 + 256)
B, P, K, N, G = 1, 131072, 256, 256, 32
x = torch.randn(B, P, K, generator=g) * 3 + 0.7
w, b = torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(N, generator=g)
ga, be = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g)
xd = x.cuda(); p3 = ops.split_weight(w.cuda(), mode="f16x3")
gac, bec, bc = ga.cuda(), be.cuda(), b.cuda()
relu = False
yref = F.group_norm(xd.double().permute(0, 2, 1), G, gac.double(), bec.double(), 1e-5).permute(0, 2, 1)
ref = (yref.reshape(B * P, K) @ w.cuda().double().t() + bc.double()).view(B, P, N).permute(0, 2, 1)
for it in range(4):
    y = ops.group_norm_nhwc(xd, G, gac, bec, 1e-5, relu=relu)
    two = ops.split_linear_nchw_out(y.view(B * P, K), p3, bc, P, out_features=N)
    mr = ops.group_norm_nhwc_stats(xd, G, 1e-5)
    one = ops.split_linear_nchw_out_gn(xd.view(B * P, K), mr, gac, bec, G, relu, p3, bc, P, out_features=N)
    torch.cuda.synchronize()
    e1, e2, ey = (one.double() - ref).abs(), (two.double() - ref).abs(), (y.double() - yref).abs()
    bad1 = torch.nonzero(e1.amax(dim=(0, 1)) > 1e-3).flatten()
    bad2 = torch.nonzero(e2.amax(dim=(0, 1)) > 1e-3).flatten()
    print(it, "one max err", e1.max().item(), "bad pixels", bad1.numel(), bad1[:4].tolist(), bad1[-2:].tolist(), "| two max err", e2.max().item(), "bad pixels", bad2.numel(),
          bad2[:4].tolist(), "| gn err", ey.max().item())
