"""OoD evaluation dataset readers for the two benchmarks `evaluate_ood.py` runs by default (reference :85),
with the reference's on-disk layouts and label conventions (datasets/README.md:91-122):

* RoadAnomaly    (datasets/road_anomaly.py:14-67): ``<root>/frame_list.json`` lists ``frames/<name>.jpg``; labels at
  ``frames/<name>.labels/labels_semantic.png``; first channel, value 2 -> 1 (OoD), 0 = inlier.
* FishyscapesLAF (datasets/fishyscapes.py:19-63): labels ``<root>/fishyscapes_lostandfound/NNNN_<city...>_labels.png``,
  image ``<root>/laf_images/<name[5:-10]>leftImg8bit.png``; 0 = inlier, 1 = OoD, 255 = ignore.

Items are ``(image uint8 [3,H,W] RGB, label int64 [H,W])`` -- what the reference yields after its ToTensorV2 transform
(support.py:70-72).  Decoding uses PIL instead of OpenCV (absent here): PNGs decode identically; baseline JPEG decoders
may differ by +-1 LSB (parity of the JPEG path is therefore unpinned)."""
import json
import os

import numpy as np
import torch
from torch.utils.data import Dataset


def read_image(path) -> np.ndarray:
    """HWC uint8 RGB (reference: cv2.imread + BGR2RGB, road_anomaly.py:60-64)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"))


def read_label(path) -> np.ndarray:
    """First channel of the label image as HW uint8: what `read_image(path)[:, :, 0]` gives (a grey PNG converted to RGB repeats its
    value), without the detour through three channels when the file is single-channel 8-bit."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode == "L":
            return np.asarray(im)
        return np.asarray(im.convert("RGB"))[:, :, 0]


def _to_item(image, label):
    return torch.from_numpy(np.ascontiguousarray(image.transpose(2, 0, 1))), torch.from_numpy(label.astype(np.int64))


# PIL hands numpy a read-only buffer; the raw item is only ever READ (copied into pinned memory / to the device): silence torch's one-time
# "array is not writable" note instead of paying a 6 MB copy per image for it
import warnings as _warnings
_warnings.filterwarnings("ignore", message="The given NumPy array is not writable")


def _to_raw_item(image, label):
    """The same sample as the decoders left it: image uint8 [H,W,3] (channels last), label uint8 [H,W].  For loaders that move the
    transpose to the GPU and never need int64 labels (evaluate_ood.run_evaluations): three 6-16 MB host passes per image less."""
    return torch.from_numpy(np.ascontiguousarray(image)), torch.from_numpy(np.ascontiguousarray(label, dtype=np.uint8))


class RoadAnomaly(Dataset):
    KIND = "road_anomaly"

    def __init__(self, dataset_root):
        with open(os.path.join(dataset_root, "frame_list.json")) as f:
            names = json.load(f)
        self.images = [os.path.join(dataset_root, "frames", n) for n in names]
        self.labels = [os.path.join(dataset_root, "frames", n[:-4] + ".labels", "labels_semantic.png") for n in names]

    def __len__(self):
        return len(self.images)

    def _decode(self, index):
        image = read_image(self.images[index])
        label = read_label(self.labels[index]).copy()
        label[label == 2] = 1
        return image, label

    def __getitem__(self, index):
        return _to_item(*self._decode(index))

    def raw_item(self, index):
        return _to_raw_item(*self._decode(index))

    def decode_spec(self, index):
        """(kind, image path, label path): what a decode worker process needs to produce raw_item(index) (rba_amd._decode_worker)"""
        return self.KIND, self.images[index], self.labels[index]


class FishyscapesLAF(Dataset):
    KIND = "fishyscapes_laf"

    def __init__(self, dataset_root):
        labels_path = os.path.join(dataset_root, "fishyscapes_lostandfound")
        files = sorted(os.listdir(labels_path))
        self.labels = [os.path.join(labels_path, f) for f in files]
        self.images = [os.path.join(dataset_root, "laf_images", f[5:-10] + "leftImg8bit.png") for f in files]

    def __len__(self):
        return len(self.images)

    def _decode(self, index):
        return read_image(self.images[index]), read_label(self.labels[index])

    def __getitem__(self, index):
        return _to_item(*self._decode(index))

    def raw_item(self, index):
        return _to_raw_item(*self._decode(index))

    def decode_spec(self, index):
        """(kind, image path, label path): what a decode worker process needs to produce raw_item(index) (rba_amd._decode_worker)"""
        return self.KIND, self.images[index], self.labels[index]


_FACTORIES = {
    "road_anomaly": lambda root: RoadAnomaly(os.path.join(root, "RoadAnomaly", "RoadAnomaly_jpg")),
    "fishyscapes_laf": lambda root: FishyscapesLAF(os.path.join(root, "Fishyscapes")),
}


def available_datasets():
    return sorted(_FACTORIES)


def get_dataset(name, datasets_folder):
    """Build ONE dataset by the reference's name (support.py:43-51, 83-86).  Unlike the reference's get_datasets, which
    eagerly constructs all nine datasets and fails unless every folder exists, this is lazy."""
    if name not in _FACTORIES:
        raise KeyError(f"unknown dataset {name!r}; available: {available_datasets()}")
    return _FACTORIES[name](datasets_folder)


def decode_spec_of(dataset):
    """index -> (kind, image path, label path) for this module's readers and torch Subsets of them; None for anything else"""
    if hasattr(dataset, "decode_spec"):
        return dataset.decode_spec
    from torch.utils.data import Subset
    if isinstance(dataset, Subset):
        inner = decode_spec_of(dataset.dataset)
        if inner is not None:
            return lambda i: inner(dataset.indices[i])
    return None


class ProcessDecoder:
    """Decode worker PROCESSES for datasets that offer `decode_spec` (this module's readers): `n` children started with subprocess
    (`python -m rba_amd._decode_worker`: numpy + Pillow only -- not forks of the process that holds the model and its HIP context); a
    sample goes to whichever worker is free, comes back through a file on /dev/shm, and the samples are yielded IN ORDER as
    raw items (image uint8 [H,W,3], label uint8 [H,W]), page-locked when `pin`.  Why processes at all: decode THREADS share the interpreter
    lock with the thread that launches the GPU work -- beyond ~8 of them the evaluator gets slower, not faster (profiles/r04_evaluator_288.json)."""

    def __init__(self, n):
        import subprocess
        import sys
        import tempfile
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        self.tmp = tempfile.mkdtemp(prefix="rba_decode_", dir=base)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
        self.stats = {"items": 0, "wait_worker_s": 0.0, "round_trip_s": 0.0, "child_decode_s": 0.0, "child_write_s": 0.0, "take_s": 0.0}
        self.procs = [subprocess.Popen([sys.executable, "-m", "rba_amd._decode_worker", self.tmp], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                       text=True, bufsize=1, env=env) for _ in range(max(1, int(n)))]

    def close(self):
        import shutil
        for p in self.procs:
            try:
                p.stdin.close()
            except Exception:       # noqa: BLE001
                pass
        for p in self.procs:
            try:
                p.wait(timeout=5)
            except Exception:       # noqa: BLE001
                p.kill()
        shutil.rmtree(self.tmp, ignore_errors=True)
        self.procs = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def items(self, dataset, indices, depth=None, pin=False):
        """raw items of dataset[indices] (`dataset`: one with `decode_spec`, or a Subset of one), in order; each sample is handed to a free worker by one of this process's helper threads,
        which then reads the answer file and page-locks it (file read and memcpy run without the interpreter lock)"""
        import itertools
        import queue
        import threading
        import time
        free = queue.Queue()
        for p in self.procs:
            free.put(p)
        tags = itertools.count()
        st = self.stats
        st_lock = threading.Lock()                          # several helper threads finish samples at once

        spec = decode_spec_of(dataset)
        if spec is None:
            raise TypeError(f"{type(dataset).__name__} offers no decode_spec(index): the process loader needs file paths, not a __getitem__")

        def get(i):
            kind, ip, lp = spec(i)
            t0 = time.perf_counter()
            p = free.get()
            tag = str(next(tags))                           # unique per request: the answer file's name
            t1 = time.perf_counter()
            try:
                p.stdin.write(f"{tag}\t{kind}\t{ip}\t{lp}\n")
                p.stdin.flush()
                line = p.stdout.readline()
            finally:
                free.put(p)
            t2 = time.perf_counter()
            if not line:
                raise RuntimeError(f"decode worker exited (sample {i})")
            f = line.rstrip("\n").split("\t")
            if f[0] != tag or f[1] == "ERR":
                raise RuntimeError(f"decode worker: sample {i}: {' '.join(f[1:])}")
            h, w, path = int(f[1]), int(f[2]), f[3]
            n = 4 * h * w
            try:
                return self._take(path, n, h, w, pin, i)
            finally:
                t3 = time.perf_counter()
                with st_lock:
                    st["items"] += 1
                    st["wait_worker_s"] += t1 - t0
                    st["round_trip_s"] += t2 - t1
                    st["child_decode_s"] += float(f[4]) * 1e-3
                    st["child_write_s"] += float(f[5]) * 1e-3
                    st["take_s"] += t3 - t2

        n = len(self.procs)
        yield from _ahead(get, indices, n + max(2, n // 2), depth or 3 * n)

    @staticmethod
    def _take(path, n, h, w, pin, i):
        """the answer file's bytes as (image [H,W,3], label [H,W]) uint8 views of one buffer (page-locked when `pin`); the file is removed"""
        if pin:
            buf = torch.empty(n, dtype=torch.uint8, pin_memory=True)
            view, got = memoryview(buf.numpy()), 0
            with open(path, "rb", buffering=0) as fh:
                while got < n:                              # a raw read may return early (a signal, a disk-backed temp directory): loop until n bytes or EOF
                    k = fh.readinto(view[got:])
                    if not k:
                        break
                    got += k
            if got != n:
                raise RuntimeError(f"decode worker: sample {i}: short answer file ({got} of {n} bytes)")
        else:
            buf = torch.from_numpy(np.fromfile(path, dtype=np.uint8))
        os.unlink(path)
        return buf[: 3 * h * w].view(h, w, 3), buf[3 * h * w:].view(h, w)


def prefetch(dataset, indices, num_threads=4, depth=None, pin=False, label_dtype=None, raw=False):
    """Yield ``dataset[i]`` for i in `indices`, in order, decoded ahead of the consumer by `num_threads` threads (at most `depth`
    items in flight).  Threads, not DataLoader worker processes: Pillow's decoders release the GIL, so PNG/JPEG decoding scales
    across threads, and forking workers from a process that already holds a HIP context and the model cost more than the whole
    evaluation of a 100-image benchmark (measured: 8 worker processes 4.6 images/s, this 15-16 images/s serial,
    profiles/r02_evaluator.json)."""
    indices = list(indices)
    raw = raw and hasattr(dataset, "raw_item")     # raw: (image uint8 [H,W,3], label uint8 [H,W]) where the dataset offers it (see _to_raw_item)

    def get(i):
        if raw:
            item = dataset.raw_item(i)
            return tuple(t.pin_memory() for t in item) if pin else item
        item = dataset[i]
        if label_dtype is not None:               # the readers yield int64 labels like the reference; uint8 holds {0, 1, 255} in 1/8 the bytes
            item = (item[0], item[1].to(label_dtype)) + tuple(item[2:])
        # pinned host memory (PyTorch's caching host allocator re-uses the blocks): the H2D copy is then truly asynchronous
        return tuple(t.pin_memory() for t in item) if pin else item

    yield from _ahead(get, indices, num_threads, depth)


def _ahead(fn, keys, num_threads, depth=None):
    """fn(key) for key in keys, in order, evaluated by `num_threads` threads at most `depth` items ahead of the consumer"""
    keys = list(keys)
    if num_threads <= 0 or len(keys) <= 1:
        for k in keys:
            yield fn(k)
        return
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    depth = depth or 2 * num_threads
    with ThreadPoolExecutor(max_workers=num_threads, thread_name_prefix="rba-decode") as ex:
        pending = deque()
        it = iter(keys)
        for k in it:
            pending.append(ex.submit(fn, k))
            if len(pending) >= depth:
                break
        while pending:
            item = pending.popleft().result()
            for k in it:
                pending.append(ex.submit(fn, k))
                break
            yield item


_THREAD_SAFE_DATASETS = ()            # filled below: this module's own readers (stateless __getitem__: open, decode, close)
_warned_foreign = set()


def _thread_safe(dataset):
    """True for datasets whose __getitem__ may be called from several threads at once: this module's readers, Subsets / ConcatDatasets of
    them, and any dataset that says so itself (`thread_safe = True`)."""
    from torch.utils.data import ConcatDataset, Subset
    if getattr(dataset, "thread_safe", False) or isinstance(dataset, _THREAD_SAFE_DATASETS):
        return True
    if isinstance(dataset, Subset):
        return _thread_safe(dataset.dataset)
    if isinstance(dataset, ConcatDataset):
        return all(_thread_safe(d) for d in dataset.datasets)
    return False


def threaded(loader, max_threads=8, force=None):
    """The batches of `loader` -- a map-style ``torch.utils.data.DataLoader`` with worker PROCESSES, e.g. the reference's
    ``DataLoader(dataset, shuffle=False, batch_size=1, num_workers=15)`` (evaluate_ood.py:210-211) -- produced in the same order, through the
    same batch sampler and collate function, by decode THREADS of this process; anything else is returned unchanged.  Why: with worker
    processes forked from a process that holds a HIP context, GPU work submitted while they run crawls -- the unmodified reference loop
    scored 17-20 images/s with the 15-process loader however fast the model was (every host -> device copy or graph launch waits 50-190 ms;
    tools/refloop_probe.py), against 80 with threads.  ``OODEvaluator.compute_anomaly_scores`` passes its loader through here, so the
    reference's loop needs no change.

    The swap calls ``dataset[i]`` from up to `max_threads` threads at once, which a dataset written for worker PROCESSES need not survive
    (a shared file handle, an h5py / LMDB reader, a global-RNG transform, `get_worker_info()`): it is therefore automatic ONLY for datasets
    known to be thread-safe -- this module's readers, Subset / ConcatDataset of them, or a dataset that declares ``thread_safe = True``.
    For any other dataset the loader is returned unchanged (worker processes, the slow path) with a one-time warning; ``force=True`` or
    RBA_LOADER_THREADS=1 opts such a dataset in, RBA_LOADER_PROCESSES=1 / ``force=False`` / a `worker_init_fn` keep the processes for
    every dataset.  The view ignores the loader's `prefetch_factor` and `timeout` (it decodes 2 x threads batches ahead) and page-locks
    the batches itself."""
    import os
    try:
        from torch.utils.data import DataLoader
    except Exception:       # pragma: no cover
        return loader
    if (not isinstance(loader, DataLoader) or loader.num_workers <= 0 or loader.batch_sampler is None or loader.worker_init_fn is not None
            or not hasattr(loader.dataset, "__getitem__") or os.environ.get("RBA_LOADER_PROCESSES") == "1" or force is False):
        return loader
    if not (force or os.environ.get("RBA_LOADER_THREADS") == "1" or _thread_safe(loader.dataset)):
        name = type(loader.dataset).__name__
        if name not in _warned_foreign:
            _warned_foreign.add(name)
            import warnings
            warnings.warn(f"rba_amd: DataLoader over {name} keeps its {loader.num_workers} worker processes (not known to be thread-safe). With a HIP "
                          f"context in the parent this loop is 4x slower than decode threads; set `dataset.thread_safe = True` or "
                          f"RBA_LOADER_THREADS=1 if {name}.__getitem__ may run in several threads at once.", RuntimeWarning, stacklevel=3)
        return loader
    return _ThreadedView(loader, min(int(loader.num_workers), max_threads))


class _ThreadedView:
    def __init__(self, loader, num_threads):
        self.loader, self.num_threads = loader, num_threads
        self.dataset = loader.dataset

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        ds, collate = self.loader.dataset, self.loader.collate_fn
        pin = torch.cuda.is_available()

        def pinned(b):
            if torch.is_tensor(b):
                return b.pin_memory()
            if isinstance(b, (list, tuple)):
                return type(b)(pinned(v) for v in b)
            if isinstance(b, dict):
                return {k: pinned(v) for k, v in b.items()}
            return b

        try:
            from torch.utils.data._utils.collate import default_collate
        except Exception:       # pragma: no cover
            default_collate = None

        def batch1(item):
            """default_collate of ONE sample without its copy: tensors gain a leading 1 as views (torch.stack would copy 6 MB of image and
            16 MB of int64 label per item inside the decode thread)"""
            if torch.is_tensor(item):
                return item[None]
            if isinstance(item, (list, tuple)) and all(torch.is_tensor(t) for t in item):
                return [t[None] for t in item]                      # default_collate returns a list for sequence samples
            return None

        def get(batch_indices):
            items = [ds[i] for i in batch_indices]
            b = batch1(items[0]) if len(items) == 1 and collate is default_collate and default_collate is not None else None
            if b is None:
                b = collate(items)
            return pinned(b) if pin else b

        yield from _ahead(get, self.loader.batch_sampler, self.num_threads)


class ThreadLoader:
    """``DataLoader(dataset, shuffle=False, batch_size=1, num_workers=N)`` for the reference's scoring loop (evaluate_ood.py:210-211,
    support.py:353-399) with decode THREADS instead of worker processes: yields ``(x [1,3,H,W] uint8, y [1,H,W])`` in order, decoded
    ahead of the consumer, in pinned host memory when a HIP device is present.  Starting 15 DataLoader worker processes from a process
    that holds the model costs ~10 s (8.6 images/s on a 96-image set, tools/evaluator_bench.py); threads start instantly and Pillow's
    decoders release the interpreter lock (78 images/s through the unmodified OODEvaluator loop)."""

    def __init__(self, dataset, num_workers=8, upper_limit=None, pin=None):
        self.dataset, self.num_workers = dataset, int(num_workers)
        self.n = len(dataset) if upper_limit is None else min(len(dataset), int(upper_limit))
        self.pin = torch.cuda.is_available() if pin is None else bool(pin)

    def __len__(self):
        return self.n

    def __iter__(self):
        for item in prefetch(self.dataset, range(self.n), self.num_workers, pin=self.pin and self.num_workers > 0):
            yield tuple(t[None] for t in item[:2])


_THREAD_SAFE_DATASETS = (RoadAnomaly, FishyscapesLAF)
