"""Checkpoint load path of the evaluator (reference: evaluate_ood.py:118-120 -> Detectron2
DetectionCheckpointer.resume_or_load(path, resume=False)): ``model_final.pth`` = torch.save({"model": state_dict,
...}); ``.pkl`` = pickle {"model": {name: array}, ...}.  Detectron2's model-zoo ``.pkl`` files hold numpy arrays; the ones
tools/convert-pretrained-swin-model-to-d2.py:24-30 writes hold *torch tensors* (``pkl.dump({"model": torch.load(...)["model"], ...})``),
whose pickles reference ``torch._utils._rebuild_tensor_v2``: those cannot be read by a numpy-only unpickler and need the
opt-in below (``trusted=True`` / ``RBA_TRUSTED_CHECKPOINT=1``), because rebuilding a pickled tensor runs the full torch unpickler."""
import pickle

import numpy as np
import torch

_TRUST_HINT = ("If you trust this file, load it with the full unpickler: load_checkpoint(..., trusted=True) / "
               "read_state_dict(path, trusted=True), or set RBA_TRUSTED_CHECKPOINT=1 for get_model() and `python -m rba_amd.evaluate_ood`.")

IGNORED_EXTRA = ("criterion.",)       # criterion.empty_weight is constructed even for eval (maskformer_model.py:148-150)


class _TensorsOnlyUnpickler(pickle.Unpickler):
    """.pkl checkpoints in Detectron2's model-zoo format hold a dict of numpy arrays and strings: allow exactly the globals numpy
    needs to rebuild an ndarray, nothing else -- a pickle can otherwise run arbitrary code on load.  (A .pkl that holds torch
    tensors, as the reference's Swin converter writes, is refused with a hint; torch's rebuild functions are NOT allow-listed:
    `torch.storage._load_from_bytes` is a full torch.load.)"""
    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"),
                ("numpy._core.multiarray", "scalar"), ("collections", "OrderedDict")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"checkpoint pickle references {module}.{name}; only numpy arrays are accepted by default.  "
                                     + _TRUST_HINT)


def read_state_dict(path, trusted=False):
    """trusted=False (default): .pth through torch.load(weights_only=True), .pkl through an unpickler restricted to numpy
    arrays -- a checkpoint file cannot execute code.  trusted=True: the full unpicklers, for checkpoints that carry other
    objects (Detectron2 writes plain tensors, so the default reads every released model_final.pth)."""
    if path.endswith(".pkl"):
        with open(path, "rb") as f:
            data = pickle.load(f, encoding="latin1") if trusted else _TensorsOnlyUnpickler(f, encoding="latin1").load()
    else:
        try:
            data = torch.load(path, map_location="cpu", weights_only=not trusted)
        except pickle.UnpicklingError as e:
            if trusted:
                raise
            raise pickle.UnpicklingError(f"{path}: {e}\n{_TRUST_HINT}") from e
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
