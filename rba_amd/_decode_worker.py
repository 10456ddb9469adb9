"""Image-decoding worker PROCESS of the evaluator's loader (`python -m rba_amd.evaluate_ood --loader processes`).

Started as `python -m rba_amd._decode_worker <tmpdir>` by rba_amd.datasets.ProcessDecoder; imports numpy and Pillow only (no torch, no HIP
context: the process is a plain child started with subprocess, not a fork of the process that holds the model -- forked DataLoader workers
are what made every host -> device copy of the reference's loop crawl, profiles/r03_reference_loop.txt).  Protocol, one line per sample on stdin:
``<tag>\\t<kind>\\t<image path>\\t<label path>``; the worker decodes exactly as rba_amd.datasets.read_image / read_label do (kind "road_anomaly":
label value 2 -> 1, road_anomaly.py:38-39), writes ``<tmpdir>/<pid>_<tag>.bin`` = image bytes [H,W,3] uint8 followed by label bytes [H,W]
uint8 (tmpdir is on /dev/shm: page-cache memory, no disk), and answers ``<tag>\\t<H>\\t<W>\\t<file>\\t<decode ms>\\t<write ms>`` on stdout (``<tag>\\tERR\\t<message>`` on failure)."""
import os
import sys
import time

import numpy as np
from PIL import Image


def decode(kind, image_path, label_path):
    with Image.open(image_path) as im:
        image = np.asarray(im.convert("RGB"))
    with Image.open(label_path) as im:
        label = np.asarray(im) if im.mode == "L" else np.asarray(im.convert("RGB"))[:, :, 0]
    if kind == "road_anomaly":
        label = label.copy()
        label[label == 2] = 1
    return np.ascontiguousarray(image), np.ascontiguousarray(label, dtype=np.uint8)


def main():
    tmpdir = sys.argv[1]
    pid = os.getpid()
    out = sys.stdout
    for line in sys.stdin:
        line = line.rstrip("\n")
        if not line:
            continue
        tag, kind, image_path, label_path = line.split("\t")
        try:
            t0 = time.perf_counter()
            image, label = decode(kind, image_path, label_path)
            t1 = time.perf_counter()
            h, w = label.shape
            if image.shape != (h, w, 3):
                raise ValueError(f"image {image.shape} and label {label.shape} differ in size")
            path = os.path.join(tmpdir, f"{pid}_{tag}.bin")
            with open(path, "wb") as f:
                f.write(image.data)
                f.write(label.data)
            out.write(f"{tag}\t{h}\t{w}\t{path}\t{(t1 - t0) * 1e3:.2f}\t{(time.perf_counter() - t1) * 1e3:.2f}\n")
        except Exception as e:                                       # noqa: BLE001 -- reported to the parent, which raises
            out.write(f"{tag}\tERR\t{type(e).__name__}: {e}\n".replace("\n", " ").rstrip() + "\n")
        out.flush()


if __name__ == "__main__":
    main()
