"""Sphere-only scenes that fit LDS through the compact tables (variants 2 / 6) and, with RTMI_FORCE_WIDE=1, through the wide
ones (variant 36): what a product build without the compact kernels would cost.  usage: gpu_compact_vs_wide.py"""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    rtmi = load_package()
    out = []
    for n in (100, 300, 700):
        rng = np.random.default_rng(n)
        sc = rtmi.Scene.new(1280, 720, 64, 20)
        sc.set_background((0.7, 0.8, 1.0), sky_gradient=True, defocus_blur=False)
        sc.camera((0, 3, 16), (0, 0, 0), (0, 1, 0), 35.0)
        mats = [sc.lambertian(tuple(rng.uniform(0.1, 0.9, 3))) for _ in range(5)] + [sc.metal((0.8, 0.8, 0.8), 0.1), sc.dielectric(1.5)]
        sc.sphere((0, -1004, 0), 1000.0, mats[0])
        for i in range(n):
            sc.sphere(tuple(rng.uniform(-4, 4, 3)), float(rng.uniform(0.1, 0.3)), mats[i % len(mats)])
        ts = []
        for _ in range(4):
            st = rtmi.Stats(); sc.render(rtmi.Opts(seed=1), st); ts.append(st.kernel_ms)
        out.append(f"{n} spheres v{st.kernel_variant} {min(ts[1:]):.2f} ms")
    sc = rtmi.Scene.rtiow(7, 1920, 1080, 256, 50)
    ts = []
    for _ in range(3):
        st = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023), st); ts.append(st.kernel_ms)
    out.append(f"rtiow 256 spp v{st.kernel_variant} {min(ts):.2f} ms")
    c1 = rtmi.Scene.load(os.path.join(ROOT, "ray-tracing-in-cuda_amd", "scenes", "three_sphere.json")); c1.override(400, 225, 100, 50)
    ts = []
    for _ in range(3):
        st = rtmi.Stats(); c1.render(rtmi.Opts(seed=2023), st); ts.append(st.kernel_ms)
    out.append(f"three_sphere v{st.kernel_variant} {min(ts):.2f} ms")
    print(("wide:    " if os.environ.get("RTMI_FORCE_WIDE") else "compact: ") + " | ".join(out), flush=True)
else:
    for force in (False, True):
        env = dict(os.environ)
        if force: env["RTMI_FORCE_WIDE"] = "1"
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
