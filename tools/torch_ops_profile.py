import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch.profiler import profile, ProfilerActivity
from rba_amd import arch as A
from rba_amd.checkpoint import load_checkpoint
from rba_amd.maskformer_model import MaskFormer
a = A.complete(A.ARCHS["swin_b_1dl"])
m = load_checkpoint(MaskFormer(a), A.seeded_weights(a, 0)).cuda().eval()
img = torch.randint(0, 256, (3, 1024, 2048), dtype=torch.uint8).cuda()
for _ in range(3):
    m.rba_scores([{"image": img}])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    m.rba_scores([{"image": img}])
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.device_time_total > 0 and not e.key.startswith("void") and "Cijk" not in e.key:
        rows.append((e.device_time_total, e.count, e.key, str(e.input_shapes)[:90]))
rows.sort(reverse=True)
for t, n, k, sh in rows[:28]:
    print(f"{t:9.1f} us  x{n:3d}  {k:38s} {sh}")
