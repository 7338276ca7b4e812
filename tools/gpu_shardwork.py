"""Work share against time share of the 1/N row-tile shards (run on the GPU box): exact query counts of the diagnostic
kernel, kernel time of the product kernel.  usage: gpu_shardwork.py [N] [spp] [spp_chunk] [deal: rt_opts.tile_rotate 0 | 1 | 2; default: what bench.py and rt_render_hip_tiles use, rt_shard_deal]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 256
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
rotate = int(sys.argv[4]) if len(sys.argv) > 4 else sc.shard_deal(None, N)
print(f"deal {rotate}", flush=True)
def best(o, n=4):
    sc.render(o)
    ts = []
    for _ in range(n):
        st = rtmi.Stats(); sc.render(o, st); ts.append(st.kernel_ms)
    return min(ts)
o = rtmi.Opts(seed=2023, spp_chunk=chunk)
cw, tw = sc.count(o), best(o)
print(f"whole: {tw:.2f} ms, queries {cw.queries}, wave-queries {cw.wave_queries}, lanes/wq {cw.queries / cw.wave_queries:.2f}, test passes/wq "
      f"{cw.clusters_visited / cw.wave_queries:.2f}, step passes/wq {cw.groups_visited / cw.wave_queries:.2f}", flush=True)
for r in range(N):
    o = rtmi.Opts(seed=2023, tile_first=r, tile_stride=N, tile_rotate=rotate, spp_chunk=chunk)
    c, t = sc.count(o), best(o)
    print(f"shard {r}/{N}: rows {sc.shard_rows(o)}, {t:.2f} ms = {t / tw * 100:.2f} % of the frame's time for {c.queries / cw.queries * 100:.2f} % of its queries "
          f"({c.wave_queries / cw.wave_queries * 100:.2f} % of its wave-queries; lanes/wq {c.queries / c.wave_queries:.2f}, test passes/wq "
          f"{c.clusters_visited / c.wave_queries:.2f}, step passes/wq {c.groups_visited / c.wave_queries:.2f}) -> {c.queries / cw.queries * tw / t * 100:.1f} % efficient", flush=True)
