"""Diagnostic counters of one row-tile shard vs the whole frame (same spp): where does a shard lose time?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
shards = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 8]
for n in shards:
    o = rtmi.Opts(seed=2023, tile_first=0, tile_stride=n)
    st = rtmi.Stats(); sc.render(o, st)
    c = sc.count(o)
    d = c.as_dict()
    print(f"N={n}: render {st.kernel_ms:.2f} ms, count kernel {c.kernel_ms:.2f} ms; samples {d['samples']}, queries/sample {d['queries']/d['samples']:.3f}, "
          f"wave_queries {d['wave_queries']}, lanes/wave-query {d['queries']/d['wave_queries']:.2f}, clusters/wq {d['clusters_visited']/d['wave_queries']:.2f}, "
          f"start spread {d['wave_start_spread_us']:.0f} us, exit spread {d['wave_end_spread_us']:.0f} us, span {d['wave_span_us']:.0f} us; "
          f"section cycles per wave-query: {[round(c / d['wave_queries']) for c in d['cycles']]}", flush=True)
