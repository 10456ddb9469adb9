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


def _to_item(image, label):
    return torch.from_numpy(np.ascontiguousarray(image.transpose(2, 0, 1))), torch.from_numpy(label.astype(np.int64))


class RoadAnomaly(Dataset):
    def __init__(self, dataset_root):
        with open(os.path.join(dataset_root, "frame_list.json")) as f:
            names = json.load(f)
        self.images = [os.path.join(dataset_root, "frames", n) for n in names]
        self.labels = [os.path.join(dataset_root, "frames", n[:-4] + ".labels", "labels_semantic.png") for n in names]

    def __len__(self):
        return len(self.images)

    def __getitem__(self, index):
        image = read_image(self.images[index])
        label = read_image(self.labels[index])[:, :, 0].copy()
        label[label == 2] = 1
        return _to_item(image, label)


class FishyscapesLAF(Dataset):
    def __init__(self, dataset_root):
        labels_path = os.path.join(dataset_root, "fishyscapes_lostandfound")
        files = sorted(os.listdir(labels_path))
        self.labels = [os.path.join(labels_path, f) for f in files]
        self.images = [os.path.join(dataset_root, "laf_images", f[5:-10] + "leftImg8bit.png") for f in files]

    def __len__(self):
        return len(self.images)

    def __getitem__(self, index):
        image = read_image(self.images[index])
        label = read_image(self.labels[index])[:, :, 0]
        return _to_item(image, label)


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
