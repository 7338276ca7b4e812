"""First GPU run: kernel vs CPU checker on three scenes + a first timing."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
rtmi = load_package()
import rtcheck

def compare(name, sc, seed=2023, **kw):
    st = rtmi.Stats()
    img = sc.render(rtmi.Opts(seed=seed, **kw), st)
    ref, _ = rtcheck.oracle_render(sc, seed=seed, spp_chunk=kw.get("spp_chunk", 0))
    d = np.abs(img - ref) / sc.spp
    print(f"{name}: {sc.width}x{sc.height}x{sc.spp} kernel {st.kernel_ms:.3f} ms  max|d|={d.max():.3g} "
          f"mismatching px={(d.max(axis=2) > 0).sum()} bit-identical={np.array_equal(img, ref)}", flush=True)
    return img, ref

print("devices", rtmi.device_count())
S = os.path.join(ROOT, "ray-tracing-in-cuda_amd", "scenes")
sc = rtmi.Scene.load(os.path.join(S, "three_sphere.json")); sc.override(width=96, height=54, spp=8)
compare("three_sphere", sc)
sc = rtmi.Scene.load(os.path.join(S, "three_sphere.json")); sc.override(width=100, height=57, spp=33)
compare("three_sphere odd size", sc)
compare("three_sphere chunked", sc, spp_chunk=8)
sc = rtmi.Scene.rtiow(7, 160, 90, 8, 50)
compare("rtiow", sc)
sc = rtmi.Scene.load(os.path.join(S, "mixed_emissive.json")); sc.override(width=160, height=90, spp=16)
compare("mixed", sc)
# counts
sc = rtmi.Scene.rtiow(7, 160, 90, 8, 50)
st = sc.count(rtmi.Opts(seed=2023))
_, oc = rtcheck.oracle_render(sc, seed=2023, want_counts=True)
print("counts gpu", st.as_dict()); print("counts cpu", oc)
# timing
for (w, h, spp) in [(1920, 1080, 8), (1920, 1080, 32)]:
    sc = rtmi.Scene.rtiow(7, w, h, spp, 50)
    for rep in range(2):
        st = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023), st)
        print(f"rtiow {w}x{h}x{spp}: {st.kernel_ms:.2f} ms -> {w*h*spp/st.kernel_ms/1e3:.1f} Msamples/s", flush=True)
sc = rtmi.Scene.load(os.path.join(S, "three_sphere.json")); sc.override(width=1920, height=1080, spp=64)
st = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023), st)
print(f"three_sphere 1920x1080x64: {st.kernel_ms:.2f} ms -> {1920*1080*64/st.kernel_ms/1e3:.1f} Msamples/s")
