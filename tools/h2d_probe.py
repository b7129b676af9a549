"""Raw host-to-device rates on this box: what bounds the PCIe-inclusive leg of bench.py."""
import time
import torch
n = 64 * 640 * 480
host_pin = torch.empty(n, dtype=torch.int16).pin_memory()
host_page = torch.empty(n, dtype=torch.int16)
dev = torch.empty(n, dtype=torch.int16, device="cuda")
for name, src in (("pinned", host_pin), ("pageable", host_page)):
    for chunks in (1, 64):
        c = n // chunks
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(10):
            for k in range(chunks):
                dev[k * c:(k + 1) * c].copy_(src[k * c:(k + 1) * c], non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"{name:9s} {chunks:3d} copies of {c * 2 / 1e6:6.2f} MB: {n * 2 / dt / 1e9:6.1f} GB/s  ({dt * 1e3:.2f} ms per 39 MB = {64 / dt:8.0f} VGA frames/s)")
