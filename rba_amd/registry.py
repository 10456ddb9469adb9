"""Name -> class registries keyed by the same strings the reference's Detectron2 registries use
(META_ARCH_REGISTRY / BACKBONE_REGISTRY / SEM_SEG_HEADS_REGISTRY / TRANSFORMER_DECODER_REGISTRY:
maskformer_model.py:23, swin.py:686, mask_former_head.py:26, msdeformattn.py:173,
mask2former_transformer_decoder.py:232), so an unchanged config.yaml resolves to our classes."""


class Registry(dict):
    def __init__(self, name):
        super().__init__()
        self._name = name

    def register(self, cls=None):
        def deco(c):
            self[c.__name__] = c
            return c
        return deco if cls is None else deco(cls)

    def get(self, name):
        if name not in self:
            raise KeyError(f"No object named '{name}' found in '{self._name}' registry!")
        return self[name]


META_ARCH_REGISTRY = Registry("META_ARCH")
BACKBONE_REGISTRY = Registry("BACKBONE")
SEM_SEG_HEADS_REGISTRY = Registry("SEM_SEG_HEADS")
TRANSFORMER_DECODER_REGISTRY = Registry("TRANSFORMER_MODULE")
