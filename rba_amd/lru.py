"""A tiny bounded cache for per-image-size constants (position embeddings, reference points, gather plans): the values depend only
on shapes, so sharing them between calls and threads is safe; the bound keeps a long evaluation over many image sizes from growing
the cache without limit.

hipGraph safety: a captured graph bakes in the ADDRESSES of the constants it read, and a replay never calls ``get`` -- so to the
cache, the entries of a captured shape look idle and would be the first to be evicted (and their memory reused under the graph's
feet).  Every entry that is looked up while the current stream is being captured is therefore PINNED: it is never evicted and does
not count against ``maxsize`` (the number of live graphs is bounded by their owners).  An entry that would have to be BUILT during
a capture is returned without being inserted: its tensors live in that graph's private pool and hold garbage until the first
replay, so no other caller may ever see them."""
from collections import OrderedDict

import torch


def _capturing() -> bool:
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


class ShapeCache:
    def __init__(self, maxsize=8):
        self.maxsize = maxsize
        self._d = OrderedDict()
        self._pinned = {}

    def get(self, key, build):
        cap = _capturing()
        v = self._pinned.get(key)
        if v is not None:
            return v
        d = self._d
        v = d.get(key)
        if v is None:
            v = build()
            if cap:
                return v                     # graph-private constants: rebuilt inside the graph, never shared
            d[key] = v
            while len(d) > self.maxsize:
                d.popitem(last=False)
        elif cap:
            self._pinned[key] = d.pop(key)   # a graph now holds its addresses
        else:
            d.move_to_end(key)
        return v

    def pinned(self):
        return len(self._pinned)

    def unpin(self):
        """Hand the pinned entries back to the bounded LRU part.  ONLY when every graph that read them is gone (MaskFormer.drop_graphs
        (release_constants=True), i.e. after a device move, which invalidates every captured graph of the model anyway)."""
        for k, v in self._pinned.items():
            self._d[k] = v
        self._pinned = {}
        while len(self._d) > self.maxsize:
            self._d.popitem(last=False)

    def __len__(self):
        return len(self._d) + len(self._pinned)
