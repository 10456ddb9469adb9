"""Checkpoint load path of the evaluator (reference: evaluate_ood.py:118-120 -> Detectron2
DetectionCheckpointer.resume_or_load(path, resume=False)): ``model_final.pth`` = torch.save({"model": state_dict,
...}); ``.pkl`` = pickle {"model": {name: ndarray}, "matching_heuristics": True}
(tools/convert-pretrained-swin-model-to-d2.py:25-30)."""
import pickle

import numpy as np
import torch

IGNORED_EXTRA = ("criterion.",)       # criterion.empty_weight is constructed even for eval (maskformer_model.py:148-150)


class _TensorsOnlyUnpickler(pickle.Unpickler):
    """.pkl checkpoints (convert-pretrained-swin-model-to-d2.py) hold a dict of numpy arrays and strings: allow exactly the
    globals numpy needs to rebuild an ndarray, nothing else -- a pickle can otherwise run arbitrary code on load."""
    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"),
                ("numpy._core.multiarray", "scalar"), ("collections", "OrderedDict")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"checkpoint pickle references {module}.{name}; only numpy arrays are accepted "
                                     "(pass trusted=True to load a checkpoint you trust with the full unpickler)")


def read_state_dict(path, trusted=False):
    """trusted=False (default): .pth through torch.load(weights_only=True), .pkl through an unpickler restricted to numpy
    arrays -- a checkpoint file cannot execute code.  trusted=True: the full unpicklers, for checkpoints that carry other
    objects (Detectron2 writes plain tensors, so the default reads every released model_final.pth)."""
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1") if trusted else _TensorsOnlyUnpickler(f, encoding="latin1").load()
    else:
        data = torch.load(path, map_location="cpu", weights_only=not trusted)
    sd = data["model"] if isinstance(data, dict) and "model" in data else data
    out = {}
    for k, v in sd.items():
        if k.startswith("module."):
            k = k[len("module."):]
        out[k] = torch.from_numpy(v) if isinstance(v, np.ndarray) else v
    return out


def load_checkpoint(model, path_or_sd, trusted=False):
    """Name-matched load; tolerates the extra criterion buffers and the recomputed integer index buffers, raises on
    anything else missing / unexpected / mis-shaped."""
    sd = read_state_dict(path_or_sd, trusted) if isinstance(path_or_sd, str) else dict(path_or_sd)
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
