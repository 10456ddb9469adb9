import torch, torch.nn.functional as F, time
torch.manual_seed(0)
x = torch.randn(8192, 512, device="cuda"); w = torch.randn(2048, 512, device="cuda") * 512 ** -0.5; b = torch.randn(2048, device="cuda") * 0.1
ref = F.gelu(F.linear(x.double(), w.double(), b.double()))
y0 = F.gelu(F.linear(x, w, b))
y1 = torch._addmm_activation(b, x, w.t(), use_gelu=True)
yt = F.gelu(F.linear(x, w, b), approximate="tanh")
print("eager exact vs fp64     ", (y0.double() - ref).abs().max().item())
print("_addmm_activation vs f64", (y1.double() - ref).abs().max().item())
print("tanh-approx vs fp64     ", (yt.double() - ref).abs().max().item())
print("_addmm_activation vs tanh-approx", (y1 - yt).abs().max().item())
for name, fn in (("linear+gelu", lambda: F.gelu(F.linear(x, w, b))), ("_addmm_activation", lambda: torch._addmm_activation(b, x, w.t(), use_gelu=True))):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize(); print(name, (time.perf_counter() - t) / 20 * 1e6, "us")
