"""Drain diagnostics (RTMI_DEBUG_DRAIN=1, diagnostic kernel) for the whole frame and a 1/N shard (run on the GPU box)."""
import os, sys
os.environ["RTMI_DEBUG_DRAIN"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
for name, o in (("whole", rtmi.Opts(seed=2023)), (f"shard 0/{N}", rtmi.Opts(seed=2023, tile_first=0, tile_stride=N))):
    sc.render(o); sc.render(o)
    print(f"--- {name}", file=sys.stderr, flush=True)
    c = sc.count(o)
    print(f"{name}: lanes per wave-query {c.queries / max(1, c.wave_queries):.2f}, test passes {c.clusters_visited / c.wave_queries:.2f}, "
          f"span {c.wave_span_us:.0f} us, exit spread {c.wave_end_spread_us:.0f} us", file=sys.stderr, flush=True)
