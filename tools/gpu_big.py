"""Scene-size sweep: N random small spheres over a ground sphere, default kernel (LDS tables or, above the
threshold, global-memory tables) -- kernel time, which variant ran, and equality with the linear scan on
a few rows.  RTMI_GLOBAL_TABLE_BYTES=<bytes> moves the threshold (read once per process).
usage: gpu_big.py [sizes] [variants]   e.g. gpu_big.py 1000,4000 0,64,128
       gpu_big.py mesh [quads per side, ...]   height fields of 2 n^2 triangles + 200 spheres (tests/test_gpu_grid_all.py)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
W, H, SPP = 1280, 720, 16
if len(sys.argv) > 1 and sys.argv[1] == "mesh":
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_grid_all import height_field
    for n in [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["32", "100", "224"])]:
        sc = height_field(rtmi, n, W, H, SPP, depth=20)
        ts = []
        for rep in range(3):
            st = rtmi.Stats(); img = sc.render(rtmi.Opts(seed=1), st); ts.append(st.kernel_ms)
        o = rtmi.Opts(seed=1, tile_rows=4, tile_first=60, tile_stride=100000)
        cst = sc.count(o)
        same = np.array_equal(sc.render(o), sc.render(rtmi.Opts(seed=1, tile_rows=4, tile_first=60, tile_stride=100000, variant=24))) if n <= 100 else None
        print(f"mesh {2*n*n} triangles + 200 spheres, kernel variant {st.kernel_variant}: {min(ts):.2f} ms ({W*H*SPP/min(ts)/1e3:.0f} Msamples/s), "
              f"list entries tested per query {cst.lane_clusters/max(1,cst.queries):.2f}, cell steps per query {cst.lane_cands/max(1,cst.queries):.2f}, "
              f"rows equal linear scan: {same}", flush=True)
    sys.exit(0)
sizes = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1000", "2000", "4000", "8000", "20000", "100000"])]
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
for n in sizes:
    rng = np.random.default_rng(n)
    half = 6.0 * (n / 5000.0) ** (1.0 / 3.0)
    sc = rtmi.Scene.new(W, H, SPP, 20)
    sc.set_background((0.7, 0.8, 1.0), sky_gradient=True, defocus_blur=False)
    sc.camera((0, half * 0.6, 3.2 * half), (0, 0, 0), (0, 1, 0), 35.0)
    mats = [sc.lambertian(sc.solid_color(tuple(rng.uniform(0.1, 0.9, 3)))) for _ in range(6)]
    mats += [sc.metal(tuple(rng.uniform(0.5, 1.0, 3)), 0.1), sc.dielectric(1.5)]
    sc.sphere((0, -1000 - half, 0), 1000.0, mats[0])
    cen = rng.uniform(-half, half, (n, 3)); rad = rng.uniform(0.05, 0.2, n)
    for i in range(n):
        sc.sphere(tuple(cen[i]), float(rad[i]), mats[i % len(mats)])
    for variant in variants:
        ts = []
        try:
            for rep in range(3):
                st = rtmi.Stats(); img = sc.render(rtmi.Opts(seed=1, variant=variant), st); ts.append(st.kernel_ms)
        except rtmi.RtmiError as e:
            print(f"n={n} variant {variant}: {e}", flush=True)
            continue
        cst = sc.count(rtmi.Opts(seed=1, variant=variant))
        # a few rows against the linear scan (no culling): must be identical
        o = rtmi.Opts(seed=1, tile_rows=4, tile_first=60, tile_stride=100000, variant=variant)
        same = np.array_equal(sc.render(o), sc.render(rtmi.Opts(seed=1, tile_rows=4, tile_first=60, tile_stride=100000, variant=16))) if n <= 4000 else None
        print(f"n={n} variant {variant} -> {st.kernel_variant} (mode {cst.cull_mode}): {min(ts):.2f} ms ({W*H*SPP/min(ts)/1e3:.0f} Msamples/s), clusters {cst.cull_clusters} x {cst.cull_cluster_size}, "
              f"wave cluster visits/query {cst.clusters_visited/max(1,cst.wave_queries):.1f}, groups passed/query {cst.groups_visited/max(1,cst.wave_queries):.1f}, "
              f"lane tests/query {cst.lane_clusters/max(1,cst.queries):.2f}, rows equal linear scan: {same}", flush=True)
