"""Build librba_hip.so (gfx950) in-tree with hipcc.  `python -m rba_amd.csrc.build [--force] [--tune]`
(--tune also builds tune/librba_tune.so, the tools-only library of probes and experimental kernel variants).
Each .hip file is compiled to an object in parallel (hipcc --offload-arch=gfx950 -c), then linked with hipcc -shared."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["rba_reduce.hip", "resample.hip", "ms_deform_attn.hip", "masked_xattn.hip", "mask_logits.hip",
           "swin_window_attn.hip", "group_norm.hip", "layer_norm.hip", "skinny_linear.hip", "split_linear.hip", "split_linear_dma.hip", "gaussian_blur.hip", "open_panoptic.hip", "dense_hybrid.hip", "patch_embed.hip"]
HEADERS = ["common.h", "rba_reduce_kernels.h", "split_linear_dma.h", "split_linear_h3.h", "mlp_fused_h3.h", "swin_window_attn_h3.h",
           os.path.join("..", "..", "include", "rba_hip.h")]
TUNE_SOURCES = [os.path.join("tune", "rba_reduce_tune.hip"), os.path.join("tune", "split_linear_tune.hip")]
TUNE_LIB = os.path.join(HERE, "tune", "librba_tune.so")
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


def build_tune(force: bool = False, verbose: bool = True) -> str:
    """librba_tune.so: probes, ablation builds and losing kernel variants, for tools/ only (never loaded by rba_amd)."""
    deps = [os.path.join(HERE, f) for f in TUNE_SOURCES + HEADERS + [os.path.join("tune", "split_linear_experiments.h"), os.path.join("tune", "rba_reduce_experiments.h")]]
    if not force and not _stale(TUNE_LIB, deps):
        return TUNE_LIB
    cmd = [HIPCC] + FLAGS + ["-shared"] + [os.path.join(HERE, f) for f in TUNE_SOURCES] + ["-o", TUNE_LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=HERE)
    return TUNE_LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    if "--tune" in sys.argv:
        print(build_tune(force="--force" in sys.argv))
