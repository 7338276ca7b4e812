"""Back-to-back launches: per-launch time without host gaps (does the per-launch 'fixed cost' survive?)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
for spp in (16, 128, 1024):
    sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
    buf = torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
    o = rtmi.Opts(seed=2023)
    stream = torch.cuda.current_stream().cuda_stream
    sc.render_device(o, buf.data_ptr(), stream); torch.cuda.synchronize()
    n = 12 if spp < 1024 else 4
    t0 = time.perf_counter()
    for _ in range(n):
        sc.render_device(o, buf.data_ptr(), stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    st = rtmi.Stats(); sc.render_device(o, buf.data_ptr(), stream, st)
    print(f"{spp} spp: back-to-back {dt:.2f} ms/launch, single timed launch {st.kernel_ms:.2f} ms", flush=True)
