"""Checkpoint load path of the evaluator (reference: evaluate_ood.py:118-120 -> Detectron2
DetectionCheckpointer.resume_or_load(path, resume=False)): ``model_final.pth`` = torch.save({"model": state_dict,
...}); ``.pkl`` = pickle {"model": {name: ndarray}, "matching_heuristics": True}
(tools/convert-pretrained-swin-model-to-d2.py:25-30)."""
import pickle

import numpy as np
import torch

IGNORED_EXTRA = ("criterion.",)       # criterion.empty_weight is constructed even for eval (maskformer_model.py:148-150)


def read_state_dict(path):
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1")
    else:
        data = torch.load(path, map_location="cpu", weights_only=False)
    sd = data["model"] if isinstance(data, dict) and "model" in data else data
    out = {}
    for k, v in sd.items():
        if k.startswith("module."):
            k = k[len("module."):]
        out[k] = torch.from_numpy(v) if isinstance(v, np.ndarray) else v
    return out


def load_checkpoint(model, path_or_sd):
    """Name-matched load; tolerates the extra criterion buffers and the recomputed integer index buffers, raises on
    anything else missing / unexpected / mis-shaped."""
    sd = read_state_dict(path_or_sd) if isinstance(path_or_sd, str) else dict(path_or_sd)
    sd = {k: v for k, v in sd.items() if not k.startswith(IGNORED_EXTRA)}
    own = model.state_dict()
    bad = [f"{k}: checkpoint {tuple(v.shape)} vs model {tuple(own[k].shape)}" for k, v in sd.items()
           if k in own and tuple(v.shape) != tuple(own[k].shape)]
    if bad:
        raise RuntimeError("shape mismatch in checkpoint:\n  " + "\n  ".join(bad))
    res = model.load_state_dict(sd, strict=False)
    missing = [k for k in res.missing_keys if not k.endswith("relative_position_index")]
    if missing or res.unexpected_keys:
        raise RuntimeError(f"checkpoint does not match the model: missing {missing[:8]} ({len(missing)}), "
                           f"unexpected {res.unexpected_keys[:8]} ({len(res.unexpected_keys)})")
    return model
