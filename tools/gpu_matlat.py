"""Does the dependent material fetch cost anything?  The same image twice: every sphere with its own (identical)
material record vs all spheres sharing one record."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
def build(shared):
    rng = np.random.default_rng(3)
    sc = rtmi.Scene.new(1920, 1080, 64, 50)
    sc.set_background((0.7, 0.8, 1.0), sky_gradient=True, defocus_blur=True)
    sc.camera((13, 2, 3), (0, 0, 0), (0, 1, 0), 20.0, 0.0, 0.1, 10.0)
    kinds = [lambda: sc.lambertian((0.5, 0.4, 0.3)), lambda: sc.metal((0.7, 0.6, 0.5), 0.2), lambda: sc.dielectric(1.5)]
    shared_ids = [k() for k in kinds] if shared else None
    sc.sphere((0, -1000, 0), 1000.0, sc.lambertian((0.5, 0.5, 0.5)))
    for a in range(-11, 11):
        for b in range(-11, 11):
            c = (a + 0.9 * rng.random(), 0.2, b + 0.9 * rng.random())
            k = 0 if rng.random() < 0.8 else (1 if rng.random() < 0.75 else 2)
            sc.sphere(c, 0.2, shared_ids[k] if shared else kinds[k]())
    return sc
for shared in (True, False, True, False):
    sc = build(shared)
    ts = []
    for rep in range(4):
        st = rtmi.Stats(); img = sc.render(rtmi.Opts(seed=1), st); ts.append(st.kernel_ms)
    print(f"shared material records: {shared}: {sc.info.num_materials} materials, min {min(ts):.2f} ms, mean radiance {img.mean()/64:.5f}", flush=True)
