import torch, sys
sys.path.insert(0, "/root/repo")
from rba_amd import ops
M, N, K = 8192, 2048, 512
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * K ** -0.5; b = torch.randn(N, device="cuda")
p = ops.split_weight(w, mode="f16x3"); xs = ops.SplitActivations.pack(x)
w2 = torch.randn(K, N, device="cuda") * N ** -0.5; p2 = ops.split_weight(w2, mode="f16x3"); r = torch.randn(M, K, device="cuda")
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
ys = ops.split_linear(xs, p, b, gelu=True, out_features=N, split_out=True); y = ops.split_linear(xs, p, b, gelu=True, out_features=N)
print("fc1 gelu  fp32-in  fp32-out %.1f" % t(lambda: ops.split_linear(x, p, b, gelu=True, out_features=N)))
print("fc1 gelu  split-in fp32-out %.1f" % t(lambda: ops.split_linear(xs, p, b, gelu=True, out_features=N)))
print("fc1 gelu  split-in split-out %.1f" % t(lambda: ops.split_linear(xs, p, b, gelu=True, out_features=N, split_out=True)))
print("fc1 gelu  fp32-in  split-out %.1f" % t(lambda: ops.split_linear(x, p, b, gelu=True, out_features=N, split_out=True)))
print("fc2 res   fp32-in %.1f" % t(lambda: ops.split_linear(y, p2, b[:K].contiguous(), out_features=K, residual=r)))
print("fc2 res   split-in %.1f" % t(lambda: ops.split_linear(ys, p2, b[:K].contiguous(), out_features=K, residual=r)))
