"""Query count of every row tile of the bench frame (counting kernel, one launch per tile; run on the GPU box)
-> gpurun_out/tilecost.json, for comparing ways of dealing the tiles out to N ranks offline (tools/tile_deal.py).
usage: gpu_tilecost.py [spp] [tile_rows]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
tr = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
nt = (1080 + tr - 1) // tr
q = []
for t in range(nt):
    c = sc.count(rtmi.Opts(seed=2023, tile_rows=tr, tile_first=t, tile_stride=nt))
    q.append(int(c.queries))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump({"spp": spp, "tile_rows": tr, "queries": q}, open(os.path.join(ROOT, "gpurun_out", "tilecost.json"), "w"))
print(f"{nt} tiles, total queries {sum(q)}", flush=True)
