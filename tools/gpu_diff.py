"""Full-frame comparison of kernel variants; mismatching rows are then checked against the CPU checker."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
rtmi = load_package()
import rtcheck
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
scene_seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
sc = rtmi.Scene.rtiow(scene_seed, 1920, 1080, spp, 50)
a = sc.render(rtmi.Opts(seed=2023, variant=0))
b = sc.render(rtmi.Opts(seed=2023, variant=16))
bad = np.argwhere((a != b).any(axis=2))
print("variant 0 vs 16: differing pixels:", len(bad), bad[:10].tolist())
osc = rtcheck.OracleScene(sc)
for (y, x) in bad[:6]:
    ref, _ = rtcheck.oracle_render(osc, seed=2023, rows=(int(y), int(y) + 1))
    print(f"pixel ({x},{y}): cull {a[y,x]} linear {b[y,x]} checker {ref[y,x]}  cull==checker {np.array_equal(a[y,x], ref[y,x])} linear==checker {np.array_equal(b[y,x], ref[y,x])}")
    # which sample
    for s in range(spp):
        pa = sc.render(rtmi.Opts(seed=2023, variant=0, sample_first=s, sample_count=1, tile_rows=1, tile_first=int(y), tile_stride=100000))
        pb = sc.render(rtmi.Opts(seed=2023, variant=16, sample_first=s, sample_count=1, tile_rows=1, tile_first=int(y), tile_stride=100000))
        if not np.array_equal(pa[0, x], pb[0, x]):
            rgb, q = rtcheck.oracle_sample(osc, 2023, int(x), int(y), s)
            print(f"   sample {s}: cull {pa[0,x]} linear {pb[0,x]} checker {rgb} queries {q}")
