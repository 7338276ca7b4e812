import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
sc = rtmi.Scene.dna(0.0); sc.override(width=1280, height=720, spp=256)   # BASELINE config 2b
for v in (0, 16):
    st = rtmi.Stats(); img = sc.render(rtmi.Opts(seed=2023, variant=v), st)
    print(f"dna 1280x720x256 variant {v}: {st.kernel_ms:.2f} ms -> {1280*720*256/st.kernel_ms/1e3:.0f} Msamples/s mean {img.mean()/256:.4f}", flush=True)
sc = rtmi.Scene.load(os.path.join(ROOT, "tests/golden/scenes/sample_scene.json")); sc.override(width=1920, height=1080, spp=512)
for v in (0, 16):
    st = rtmi.Stats(); img = sc.render(rtmi.Opts(seed=2023, variant=v), st)
    print(f"sample_scene 1920x1080x512 variant {v}: {st.kernel_ms:.2f} ms -> {1920*1080*512/st.kernel_ms/1e3:.0f} Msamples/s", flush=True)
sc = rtmi.Scene.load(os.path.join(ROOT, "ray-tracing-in-cuda_amd/scenes/three_sphere.json")); sc.override(width=400, height=225, spp=100)
st = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023), st); sc.render(rtmi.Opts(seed=2023), st)
print(f"three_sphere 400x225x100: {st.kernel_ms:.3f} ms -> {400*225*100/st.kernel_ms/1e3:.0f} Msamples/s")
# where the DNA frame spends its time (diagnostic kernel sections)
sc = rtmi.Scene.dna(0.0); sc.override(width=1280, height=720, spp=64)
c = sc.count(rtmi.Opts(seed=2023)); d = c.as_dict(); cy = d["cycles"]; tot = sum(cy) or 1
names = ["refill", "prefix spheres", "culled spheres/rects/cylinders", "shading", "accumulate", "loop control"]
print("dna sections: " + ", ".join(f"{n} {100*v/tot:.1f}%" for n, v in zip(names, cy)), f"; queries/sample {d['queries']/d['samples']:.2f}, lanes/wave-query {d['queries']/max(1,d['wave_queries']):.1f}")
for name in ("blue", "blue2"):
    sc = rtmi.Scene.load(os.path.join(ROOT, f"tests/golden/scenes/{name}.json")); sc.override(width=1280, height=720, spp=256)
    st = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023), st); sc.render(rtmi.Opts(seed=2023), st)
    info = sc.info
    print(f"{name} 1280x720x256 ({info.num_prims} objects): {st.kernel_ms:.2f} ms -> {1280*720*256/st.kernel_ms/1e3:.0f} Msamples/s", flush=True)
