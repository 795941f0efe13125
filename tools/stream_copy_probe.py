#!/usr/bin/env python3
"""Probe: D2H / H2D rate of a 16 MiB pinned copy on the k-th stream a process creates."""
import time
import torch

dev = torch.device("cuda", 0)
d = torch.empty(16 << 20, dtype=torch.uint8, device=dev)
h = torch.empty(16 << 20, dtype=torch.uint8).pin_memory()
streams = [torch.cuda.Stream(device=dev) for _ in range(10)]
for rep in range(2):
    for k, s in enumerate(streams):
        with torch.cuda.stream(s):
            for direction in ("d2h", "h2d"):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    if direction == "d2h":
                        h.copy_(d, non_blocking=True)
                    else:
                        d.copy_(h, non_blocking=True)
                s.synchronize()
                dt = (time.perf_counter() - t0) / 10
                print(f"rep {rep} stream {k} {direction}: {16.777216 / dt / 1e3:.1f} GB/s", flush=True)
