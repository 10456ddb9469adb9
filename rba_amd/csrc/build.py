"""Build librba_hip.so (gfx950) in-tree with hipcc.  `python -m rba_amd.csrc.build [--force]`.
Each .hip file is compiled to an object in parallel (hipcc --offload-arch=gfx950 -c), then linked with hipcc -shared."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["rba_reduce.hip", "rba_reduce_tune.hip", "resample.hip", "ms_deform_attn.hip", "masked_xattn.hip", "mask_logits.hip",
           "swin_window_attn.hip", "group_norm.hip", "layer_norm.hip", "skinny_linear.hip", "split_linear.hip", "gaussian_blur.hip", "open_panoptic.hip"]
HEADERS = ["common.h", "rba_reduce_kernels.h", os.path.join("..", "..", "include", "rba_hip.h")]
LIB = os.path.join(HERE, "librba_hip.so")
OBJ = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]


def _stale(target, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def stale() -> bool:
    return _stale(LIB, [os.path.join(HERE, f) for f in SOURCES + HEADERS])


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS]

    def compile_one(src):
        s, o = os.path.join(HERE, src), os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, cwd=HERE)
        return o

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=HERE)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
