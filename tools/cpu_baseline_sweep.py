"""BASELINE.md section 3's full CPU-baseline protocol (thread-count sweep on the 256x512 forward, C1's size with the best count and with ONE
thread, the bench-size forward, the isolated reduction with all / one thread): ~50 s on 2 x EPYC 9575F.  bench.py's default is the quick leg
(one full-size forward at the thread count this sweep picks there: 16); `python bench.py --cpu-baseline full` embeds this record in the line.
python tools/cpu_baseline_sweep.py [arch] [H W]   (no GPU needed)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    arch = sys.argv[1] if len(sys.argv) > 1 else "swin_b_1dl"
    h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1024, 2048)
    print(json.dumps(bench.cpu_baseline(arch, h, w, mode="full")))
