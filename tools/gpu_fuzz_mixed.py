"""More seeds of tests/test_gpu_grid_all.py::test_random_mixed_scenes_equal_the_flat_scan and of the sphere fuzz of
tests/test_gpu_fuzz.py (run on the GPU box): gpu_fuzz_mixed.py [first seed] [seeds]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
rtmi = load_package()
import rtcheck
import test_gpu_grid_all as G
import test_gpu_fuzz as F
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bad = 0
for seed in range(first, first + count):
    for name, fn in (("mixed", G.test_random_mixed_scenes_equal_the_flat_scan), ("spheres", F.test_random_geometry_every_candidate_search_equals_the_flat_scan)):
        try:
            fn(rtmi, rtcheck, seed)
            print(f"seed {seed} {name}: ok", flush=True)
        except Exception:
            bad += 1
            print(f"seed {seed} {name}: FAILED", flush=True)
            traceback.print_exc()
print("failures:", bad)
sys.exit(1 if bad else 0)
