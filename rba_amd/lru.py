"""A tiny bounded cache for per-image-size constants (position embeddings, reference points, gather plans): the values depend only
on shapes, so sharing them between calls and threads is safe; the bound keeps a long evaluation over many image sizes from growing
the cache without limit."""
from collections import OrderedDict


class ShapeCache:
    def __init__(self, maxsize=8):
        self.maxsize = maxsize
        self._d = OrderedDict()

    def get(self, key, build):
        d = self._d
        v = d.get(key)
        if v is None:
            v = build()
            d[key] = v
            while len(d) > self.maxsize:
                d.popitem(last=False)
        else:
            d.move_to_end(key)
        return v

    def __len__(self):
        return len(self._d)
