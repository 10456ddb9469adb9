#!/usr/bin/env python3
"""What is slow about a DataLoader worker's (shared-memory) tensor on this box?  Times host copies and H2D copies of one 3 x 1024 x 2048 uint8 item."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class DS(torch.utils.data.Dataset):
    def __len__(self): return 16
    def __getitem__(self, i): return torch.full((3, 1024, 2048), i, dtype=torch.uint8)


def t(f, n=1):
    t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


def main():
    torch.zeros(1, device="cuda")
    pinned = torch.empty(3 * 1024 * 2048, dtype=torch.uint8, pin_memory=True)
    priv = torch.full((1, 3, 1024, 2048), 7, dtype=torch.uint8)
    print("torch threads", torch.get_num_threads())
    for nt in (None, 1):
        if nt: torch.set_num_threads(nt)
        print(f"--- torch.set_num_threads({nt})")
        print("private -> private clone      %.2f ms" % t(lambda: priv.clone())[0])
        print("private -> pinned copy_       %.2f ms" % t(lambda: pinned.view(priv.shape).copy_(priv))[0])
        print("private -> device .to()       %.2f ms" % t(lambda: priv.to("cuda"))[0])
        print("pinned  -> device .to()       %.2f ms" % t(lambda: pinned.to("cuda", non_blocking=True))[0])
        it = iter(torch.utils.data.DataLoader(DS(), batch_size=1, num_workers=4))
        x = next(it)
        print("shm item: is_shared", x.is_shared(), "contiguous", x.is_contiguous())
        print("shm -> private clone (1st)    %.2f ms" % t(lambda: x.clone())[0])
        print("shm -> private clone (2nd)    %.2f ms" % t(lambda: x.clone())[0])
        x = next(it)
        print("shm -> pinned copy_ (1st)     %.2f ms" % t(lambda: pinned.view(x.shape).copy_(x))[0])
        print("shm -> pinned copy_ (2nd)     %.2f ms" % t(lambda: pinned.view(x.shape).copy_(x))[0])
        x = next(it)
        print("shm -> device .to() (1st)     %.2f ms" % t(lambda: x.to("cuda"))[0])
        print("shm -> device .to() (2nd)     %.2f ms" % t(lambda: x.to("cuda"))[0])
        x = next(it)
        print("shm: x.is_pinned() (1st)      %.2f ms" % t(lambda: x.is_pinned())[0])
        print("shm: x.is_pinned() (2nd)      %.2f ms" % t(lambda: x.is_pinned())[0])
        print("shm -> numpy copy after that  %.2f ms" % t(lambda: x.numpy().copy())[0])
        x = next(it)
        import numpy as np
        print("shm -> numpy copy (1st)       %.2f ms" % t(lambda: np.array(x.numpy(), copy=True))[0])
        x = next(it)
        print("shm -> bytes() via memoryview %.2f ms" % t(lambda: bytes(memoryview(x.numpy().reshape(-1))))[0])
        del it


if __name__ == "__main__":
    main()
