import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
sc = rtmi.Scene.rtiow(7, 1920, 1080, 16, 50)
for v in (0, 16):
    print("=== variant", v, flush=True)
    img = sc.render(rtmi.Opts(seed=2023, variant=v, sample_first=7, sample_count=1, tile_rows=1, tile_first=828, tile_stride=100000))
    print(img[0, 90], flush=True)
