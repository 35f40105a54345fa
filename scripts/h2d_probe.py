"""Pageable vs pinned host-to-device copy rate of one 1000-image fp32 batch (GPU box only)."""
import time

import torch


def main():
    n = 1000 * 3 * 227 * 227
    a = torch.empty(n, dtype=torch.float32)
    p = torch.empty(n, dtype=torch.float32, pin_memory=True)
    d = torch.empty(n, dtype=torch.float32, device="cuda")
    for name, h in (("pageable", a), ("pinned", p)):
        d.copy_(h)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(5):
            d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.time() - t) / 5
        print(name, "%.2f ms  %.1f GB/s" % (dt * 1e3, n * 4 / dt / 1e9))


if __name__ == "__main__":
    main()
