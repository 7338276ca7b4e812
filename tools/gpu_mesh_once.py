"""One scene rendered a few times (for rocprofv3 runs): gpu_mesh_once.py mesh <n> | spheres <n> | dna"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
rtmi = load_package()
what = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if what == "mesh":
    from test_gpu_grid_all import height_field
    sc = height_field(rtmi, n, 1280, 720, 16, depth=20)
elif what == "dna":
    sc = rtmi.Scene.dna(0.0); sc.override(1280, 720, 256, 50)
else:
    rng = np.random.default_rng(n)
    half = 6.0 * (n / 5000.0) ** (1.0 / 3.0)
    sc = rtmi.Scene.new(1280, 720, 16, 20)
    sc.set_background((0.7, 0.8, 1.0), sky_gradient=True, defocus_blur=False)
    sc.camera((0, half * 0.6, 3.2 * half), (0, 0, 0), (0, 1, 0), 35.0)
    mats = [sc.lambertian(sc.solid_color(tuple(rng.uniform(0.1, 0.9, 3)))) for _ in range(6)] + [sc.metal((0.7, 0.7, 0.7), 0.1), sc.dielectric(1.5)]
    sc.sphere((0, -1000 - half, 0), 1000.0, mats[0])
    cen = rng.uniform(-half, half, (n, 3)); rad = rng.uniform(0.05, 0.2, n)
    for i in range(n):
        sc.sphere(tuple(cen[i]), float(rad[i]), mats[i % len(mats)])
chunks = [int(c) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
for ch in chunks:
    for _ in range(3):
        st = rtmi.Stats(); sc.render(rtmi.Opts(seed=1, spp_chunk=ch), st)
        print(f"{what} {n} chunk {ch}: variant {st.kernel_variant} {st.kernel_ms:.2f} ms", flush=True)
