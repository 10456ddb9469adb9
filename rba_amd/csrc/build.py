"""Build librba_hip.so (gfx950) in-tree with hipcc.  `python -m rba_amd.csrc.build [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["rba_reduce.hip", "resample.hip", "ms_deform_attn.hip", "masked_xattn.hip", "mask_logits.hip",
           "swin_window_attn.hip", "group_norm.hip", "layer_norm.hip", "skinny_linear.hip"]
HEADERS = ["common.h", os.path.join("..", "..", "include", "rba_hip.h")]
LIB = os.path.join(HERE, "librba_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast",
         "-Wno-unused-result"]


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return LIB
    cmd = [HIPCC] + FLAGS + [os.path.join(HERE, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=HERE)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
