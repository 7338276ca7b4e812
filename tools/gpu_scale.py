import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
for (w, h, spp) in [(1920, 1080, 256), (960, 540, 1024), (1920, 1080, 128), (1920, 135, 1024), (3840, 2160, 64), (1920, 1080, 1024), (1920, 1080, 2048)]:
    sc = rtmi.Scene.rtiow(7, w, h, spp, 50)
    best = 1e9
    for rep in range(3):
        st = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023), st); best = min(best, st.kernel_ms)
    print(f"{w}x{h}x{spp}: {best:.2f} ms  {w*h*spp/best/1e3:.0f} Msamples/s  ({w*h*spp/1e6:.0f} Msamples)", flush=True)
