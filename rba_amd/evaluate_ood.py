"""The evaluator-facing functions of the reference's ``evaluate_ood.py`` with the same names, arguments and
outputs: ``get_model`` (:108-124), ``get_logits`` (:127-140), ``get_RbA`` (:143-150), ``get_energy`` (:152-159)."""
import torch

from .arch import arch_from_cfg
from .checkpoint import load_checkpoint
from .config import load_cfg
from .registry import META_ARCH_REGISTRY
from . import maskformer_model as _mm  # noqa: F401  (registers MaskFormer)

DEVICE = torch.device("cuda")


def build_model(cfg):
    """Trainer.build_model(cfg) (train_net.py:75-80): META_ARCH_REGISTRY[cfg.MODEL.META_ARCHITECTURE](...)."""
    cls = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)
    return cls(arch_from_cfg(cfg), backbone_name=cfg.MODEL.BACKBONE.NAME, head_name=cfg.MODEL.SEM_SEG_HEAD.NAME)


def get_model(config_path, model_path, device=None):
    """Creates the model from a config path and a checkpoint path (``model_path`` may be None for random init)."""
    cfg = load_cfg(config_path, ["OUTPUT_DIR", "output/"])
    model = build_model(cfg)
    if model_path is not None:
        load_checkpoint(model, model_path)
    model.to(device or DEVICE)
    return model.eval()


def get_logits(model, x, **kwargs):
    """x [1,3,H,W] -> logits [1,K,H,W]."""
    with torch.no_grad():
        out = model([{"image": x[0].to(model.device)}])
    return out[0]["sem_seg"].unsqueeze(0)


def get_RbA(model, x, **kwargs):
    """x [1,3,H,W] -> RbA score [H,W] = -sum_k tanh(sem_seg_k)."""
    if hasattr(model, "rba_scores"):
        return model.rba_scores([{"image": x[0].to(model.device)}])[0]
    with torch.no_grad():
        out = model([{"image": x[0].to(model.device)}])
    return -out[0]["sem_seg"].tanh().sum(dim=0)


def get_energy(model, x, **kwargs):
    with torch.no_grad():
        out = model([{"image": x[0].to(model.device)}])
    return -torch.logsumexp(out[0]["sem_seg"], dim=0)
