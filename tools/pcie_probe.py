"""What can the host link of this box carry? (the host-boundary leg.) Pinned host memory <-> HBM in chunks of the size of a configs[2] bin image
(56 MB) over 8 streams, alone, in both directions at once, and under a stream of device-to-device copies that keeps HBM busy the way the sort does.
torch only (no kernel of this repo): a property of the box, printed as one JSON line."""
import json
import time

import torch

CH = 56 << 20
N_CH = 64  # 3.6 GB per direction
dev = torch.device("cuda:0")
pin_in = torch.empty(N_CH * CH, dtype=torch.uint8, pin_memory=True)
pin_out = torch.empty(N_CH * CH, dtype=torch.uint8, pin_memory=True)
d_in = torch.empty(N_CH * CH, dtype=torch.uint8, device=dev)
d_out = torch.empty(N_CH * CH, dtype=torch.uint8, device=dev)
big_a = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
big_b = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
streams = [torch.cuda.Stream() for _ in range(8)]
busy = torch.cuda.Stream()


def run(h2d, d2h, load, reps=3):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(N_CH):
            with torch.cuda.stream(streams[i % 8]):
                sl = slice(i * CH, (i + 1) * CH)
                if h2d:
                    d_in[sl].copy_(pin_in[sl], non_blocking=True)
                if d2h:
                    pin_out[sl].copy_(d_out[sl], non_blocking=True)
            if load and i % 2 == 0:
                with torch.cuda.stream(busy):
                    for _ in range(4):
                        big_b.copy_(big_a, non_blocking=True)  # 4 GB of HBM traffic each
        for s in streams:
            s.synchronize()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        best = dt if best is None else min(best, dt)
    return N_CH * CH / best / 1e9


out = {"chunk_MB": CH >> 20, "chunks": N_CH, "streams": 8,
       "h2d_GBs": run(True, False, False), "d2h_GBs": run(False, True, False), "both_directions_GBs_each": run(True, True, False),
       "h2d_GBs_under_hbm_load": run(True, False, True), "both_directions_GBs_each_under_hbm_load": run(True, True, True)}
print(json.dumps(out))
