"""BASELINE config 1 (3-sphere scene, 400x225x100): kernel time vs chunk size (0 = the library's choice)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
sc = rtmi.Scene.load(os.path.join(ROOT, "ray-tracing-in-cuda_amd/scenes/three_sphere.json")); sc.override(width=400, height=225, spp=100)
for rep in range(3): sc.render(rtmi.Opts(seed=2023))
for chunk in (0, 2, 4, 8, 16, 32, 64, 100):
    ts = []
    for rep in range(6):
        st = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023, spp_chunk=chunk), st); ts.append(st.kernel_ms)
    print(f"three_sphere 400x225x100 chunk {chunk}: min {min(ts):.3f} ms -> {400*225*100/min(ts)/1e3:.0f} Msamples/s", flush=True)
