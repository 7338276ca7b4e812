"""What one rank of an N-GPU strong-scaled bench does, in steady state on one GPU: its row-tile shard launched back to
back without host waits (as bench.py's timed loop does), against the whole frame launched the same way, divided by N."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
sc = rtmi.Scene.rtiow(7, 1920, 1080, 1024, 50)
buf = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream


def per_launch(o, n):
    for _ in range(2):
        sc.render_device(o, buf.data_ptr(), stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        sc.render_device(o, buf.data_ptr(), stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


whole = per_launch(rtmi.Opts(seed=2023), 5)
print(f"whole frame, back to back: {whole:.2f} ms", flush=True)
for n in (2, 4, 8):
    ts = [per_launch(rtmi.Opts(seed=2023, tile_first=r, tile_stride=n), 24) for r in sorted({0, n // 2, n - 1})]
    print(f"N={n}: shards {[round(t, 2) for t in ts]} ms back to back; whole/N = {whole / n:.2f}; efficiency of the slowest "
          f"{whole / n / max(ts) * 100:.1f} %", flush=True)
