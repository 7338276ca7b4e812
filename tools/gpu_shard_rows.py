"""The 1/N shards cut in row tiles of different heights (run on the GPU box): kernel time and query share per shard.
usage: gpu_shard_rows.py [N] [spp] [tile_rows,...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rows = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "8,4,2,1").split(",")]
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
def best(o, n=3):
    sc.render(o)
    ts = []
    for _ in range(n):
        st = rtmi.Stats(); sc.render(o, st); ts.append(st.kernel_ms)
    return min(ts)
o = rtmi.Opts(seed=2023)
tw, cw = best(o), sc.count(o)
print(f"whole: {tw:.2f} ms", flush=True)
for tr in rows:
    ts, qs = [], []
    for r in range(N):
        o = rtmi.Opts(seed=2023, tile_rows=tr, tile_first=r, tile_stride=N, tile_rotate=1)
        ts.append(best(o)); qs.append(sc.count(o).queries / cw.queries * 100)
    print(f"tile_rows {tr}: shard ms {[round(t, 2) for t in ts]} (slowest {max(ts):.2f} = {tw / N / max(ts) * 100:.1f} % of frame/N), query share % {[round(q, 2) for q in qs]}", flush=True)
