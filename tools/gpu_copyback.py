"""What the host-buffer entry points add to the launch (run on the GPU box): rt_render_hip and rt_render_hip_tiles, call to
return, into a buffer that is touched already and into a fresh one, against the kernel time.  usage: gpu_copyback.py [spp] [torch]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
if len(sys.argv) > 2 and sys.argv[2] == "torch":  # as in bench.py: torch's runtime and allocator live in the process
    import torch
    keep = torch.zeros((1080, 1920, 3), device="cuda:0"); torch.cuda.synchronize()
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
lib = rtmi._lib
o = rtmi.Opts(seed=2023)
out = np.zeros((1080, 1920, 3), dtype=np.float32)
st = rtmi.Stats()
for name, call in (("rt_render_hip", lambda buf: lib.rt_render_hip(sc._h, C.byref(o), buf.ctypes.data_as(C.c_void_p), C.byref(st))),
                   ("rt_render_hip_tiles", lambda buf: lib.rt_render_hip_tiles(sc._h, C.byref(o), None, 1, buf.ctypes.data_as(C.c_void_p), C.byref(st)))):
    call(out)
    for what in ("touched buffer", "fresh buffer"):
        ts = []
        for _ in range(5):
            buf = out if what == "touched buffer" else np.empty((1080, 1920, 3), dtype=np.float32)
            t0 = time.perf_counter(); rc = call(buf); t1 = time.perf_counter()
            assert rc == 0
            ts.append(((t1 - t0) * 1e3, st.kernel_ms))
        best = min(ts)
        print(f"{name}, {what}: call {best[0]:.2f} ms, kernel {best[1]:.2f} ms, rest {best[0] - best[1]:.2f} ms", flush=True)
