"""CPU: dataset readers on synthetic on-disk datasets with the reference's layouts (datasets/README.md:91-122)."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from rba_amd import datasets as DS


def make_road_anomaly(root, n=3, h=24, w=40):
    d = os.path.join(root, "RoadAnomaly", "RoadAnomaly_jpg")
    os.makedirs(os.path.join(d, "frames"))
    rng = np.random.RandomState(0)
    names, imgs, labs = [], [], []
    for i in range(n):
        name = f"scene{i}.png"          # PNG payload under the listed name: lossless, so the check below is exact
        img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        Image.fromarray(img).save(os.path.join(d, "frames", name))
        lab = rng.choice([0, 2], size=(h, w), p=[0.9, 0.1]).astype(np.uint8)
        os.makedirs(os.path.join(d, "frames", name[:-4] + ".labels"))
        Image.fromarray(np.stack([lab, lab, lab], -1)).save(os.path.join(d, "frames", name[:-4] + ".labels", "labels_semantic.png"))
        names.append(name); imgs.append(img); labs.append(lab)
    with open(os.path.join(d, "frame_list.json"), "w") as f:
        json.dump(names, f)
    return imgs, labs


def make_fs_laf(root, n=2, h=16, w=32):
    d = os.path.join(root, "Fishyscapes")
    os.makedirs(os.path.join(d, "fishyscapes_lostandfound"))
    os.makedirs(os.path.join(d, "laf_images"))
    rng = np.random.RandomState(1)
    imgs, labs = [], []
    for i in range(n):
        stem = f"04_Maurener_Weg_8_00000{i}_000030_"
        img = rng.randint(0, 256, (h, w, 3), dtype=np.uint8)
        lab = rng.choice([0, 1, 255], size=(h, w), p=[0.8, 0.1, 0.1]).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(d, "laf_images", stem + "leftImg8bit.png"))
        Image.fromarray(lab).save(os.path.join(d, "fishyscapes_lostandfound", f"{i:04d}_" + stem + "labels.png"))
        imgs.append(img); labs.append(lab)
    return imgs, labs


def test_road_anomaly(tmp_path):
    imgs, labs = make_road_anomaly(str(tmp_path))
    ds = DS.get_dataset("road_anomaly", str(tmp_path))
    assert len(ds) == 3
    for i in range(3):
        x, y = ds[i]
        assert x.dtype == torch.uint8 and x.shape == (3, 24, 40) and y.dtype == torch.int64 and y.shape == (24, 40)
        assert np.array_equal(x.numpy().transpose(1, 2, 0), imgs[i])
        assert np.array_equal(y.numpy(), (labs[i] == 2).astype(np.int64))       # 2 -> 1 (road_anomaly.py:38-39)
        xr, yr = ds.raw_item(i)                                                   # the evaluator's loader form: channels last, uint8 labels
        assert xr.dtype == torch.uint8 and xr.shape == (24, 40, 3) and yr.dtype == torch.uint8 and xr.is_contiguous()
        assert torch.equal(xr.permute(2, 0, 1), x) and torch.equal(yr.to(torch.int64), y)


def test_fishyscapes_laf(tmp_path):
    imgs, labs = make_fs_laf(str(tmp_path))
    ds = DS.get_dataset("fishyscapes_laf", str(tmp_path))
    assert len(ds) == 2
    for i in range(2):
        x, y = ds[i]
        assert np.array_equal(x.numpy().transpose(1, 2, 0), imgs[i]) and np.array_equal(y.numpy(), labs[i].astype(np.int64))
        xr, yr = ds.raw_item(i)
        assert torch.equal(xr.permute(2, 0, 1), x) and torch.equal(yr.to(torch.int64), y) and yr.dtype == torch.uint8
    got = list(DS.prefetch(ds, range(2), 2, raw=True))                           # raw items through the decode threads, in order
    assert all(torch.equal(a[0], ds.raw_item(i)[0]) and torch.equal(a[1], ds.raw_item(i)[1]) for i, a in enumerate(got))
    with pytest.raises(KeyError):
        DS.get_dataset("cityscapes", str(tmp_path))


def test_thread_loader_matches_the_dataset_order(tmp_path):
    """ThreadLoader = the reference loop's DataLoader(batch_size=1, shuffle=False) with decode threads: same items, same order"""
    from rba_amd.datasets import ThreadLoader, get_dataset
    imgs, labs = make_road_anomaly(str(tmp_path), n=5, h=24, w=40)
    ds = get_dataset("road_anomaly", str(tmp_path))
    for workers in (0, 3):
        out = list(ThreadLoader(ds, workers, pin=False))
        assert len(out) == 5 == len(ThreadLoader(ds, workers))
        for i, (x, y) in enumerate(out):
            assert x.shape == (1, 3, 24, 40) and y.shape == (1, 24, 40)
            assert torch.equal(x[0], ds[i][0]) and torch.equal(y[0], ds[i][1])
    assert len(list(ThreadLoader(ds, 2, upper_limit=3, pin=False))) == 3


def test_threaded_view_of_a_process_dataloader_yields_the_same_batches(monkeypatch):
    """datasets.threaded: the reference's DataLoader(dataset, batch_size=1, num_workers=15) is iterated with threads -- same order, same sampler,
    same collate function; loaders without workers, with a worker_init_fn, or with RBA_LOADER_PROCESSES=1 are left alone; a dataset that is not
    known to be thread-safe keeps its worker processes unless it opts in"""
    import torch
    from torch.utils.data import DataLoader, Dataset, Subset
    from rba_amd.datasets import threaded

    class DS(Dataset):
        def __len__(self):
            return 11

        def __getitem__(self, i):
            return torch.full((3, 4, 5), i, dtype=torch.uint8), torch.full((4, 5), i % 3, dtype=torch.int64)

    ds = DS()
    foreign = DataLoader(ds, shuffle=False, batch_size=1, num_workers=15)
    with pytest.warns(RuntimeWarning, match="thread_safe"):
        assert threaded(foreign) is foreign                           # unknown dataset: processes stay, one warning
    assert type(threaded(foreign, force=True)).__name__ == "_ThreadedView"
    DS.thread_safe = True                                             # the dataset opts in
    ref = list(DataLoader(ds, shuffle=False, batch_size=1, num_workers=0))
    view = threaded(DataLoader(ds, shuffle=False, batch_size=1, num_workers=15))
    assert type(view).__name__ == "_ThreadedView" and len(view) == 11 and view.dataset is ds
    got = list(view)
    assert len(got) == len(ref) and all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(got, ref))
    # batches of 4 over a subset, custom collate
    sub = Subset(ds, [9, 2, 5, 7, 1])
    coll = lambda items: {"x": torch.stack([it[0] for it in items]), "n": len(items)}
    got = list(threaded(DataLoader(sub, batch_size=4, num_workers=2, collate_fn=coll)))
    assert [g["n"] for g in got] == [4, 1] and got[0]["x"][:, 0, 0, 0].tolist() == [9, 2, 5, 7]
    plain = DataLoader(ds, batch_size=1, num_workers=0)
    assert threaded(plain) is plain
    keep = DataLoader(ds, batch_size=1, num_workers=2, worker_init_fn=lambda i: None)
    assert threaded(keep) is keep
    assert threaded([1, 2, 3]) == [1, 2, 3]
    monkeypatch.setenv("RBA_LOADER_PROCESSES", "1")
    procs = DataLoader(ds, batch_size=1, num_workers=2)
    assert threaded(procs) is procs


def test_process_decoder_yields_the_raw_items_in_order(tmp_path):
    """`--loader processes`: child processes (rba_amd._decode_worker, numpy + Pillow only) give exactly dataset.raw_item(i), in the order
    asked, for both readers (RoadAnomaly's 2 -> 1 relabelling included); a missing file is an error in the parent, not a hang; the
    /dev/shm directory is gone afterwards."""
    import os
    import pytest
    from rba_amd import datasets as DS
    make_fs_laf(str(tmp_path), n=5)
    make_road_anomaly(str(tmp_path), n=4)
    for name in ("fishyscapes_laf", "road_anomaly"):
        ds = DS.get_dataset(name, str(tmp_path))
        order = list(range(len(ds)))[::-1] + [0, 0]
        with DS.ProcessDecoder(3) as pd:
            tmp = pd.tmp
            got = list(pd.items(ds, order))
            assert os.listdir(tmp) == []                                 # every answer file consumed
        assert not os.path.exists(tmp)
        assert len(got) == len(order)
        for i, (x, y) in zip(order, got):
            xr, yr = ds.raw_item(i)
            assert x.dtype == torch.uint8 and y.dtype == torch.uint8 and torch.equal(x, xr) and torch.equal(y, yr), (name, i)
    ds = DS.get_dataset("fishyscapes_laf", str(tmp_path))
    os.remove(ds.images[2])
    with DS.ProcessDecoder(2) as pd:
        with pytest.raises(RuntimeError, match="sample 2"):
            list(pd.items(ds, range(len(ds))))


def test_decode_worker_imports_neither_torch_nor_the_package_ops():
    """the worker process must stay a plain numpy + Pillow process: no torch import (seconds of start-up per worker), no HIP library"""
    import subprocess
    import sys
    code = "import sys, rba_amd._decode_worker; print(int('torch' in sys.modules), int('rba_amd._lib' in sys.modules))"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).stdout.split()
    assert out == ["0", "0"]
