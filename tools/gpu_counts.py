"""Diagnostic counters of the bench frame per query: candidate lanes, resolve entries per wave (run on the GPU box).  usage: gpu_counts.py [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
st = sc.count(rtmi.Opts(seed=2023))
d = st.as_dict(); q = d["queries"]
print(d)
print("per query: lane candidates %.2f, wave entries per wave-query %.1f (of %d tests), lanes per entry %.2f" % (
    d["cand_lanes"] / q, d["cand_waves"] / (q / 64), sc.info.num_prims, d["cand_lanes"] / d["cand_waves"]))
wq = d["wave_queries"]
print("culling: groups visited per wave-query %.2f (lane mean %.2f), clusters visited per wave-query %.2f (lane mean %.2f), lanes per wave-query %.1f" % (
    d["groups_visited"] / wq, d["lane_groups"] / q, d["clusters_visited"] / wq, d["lane_clusters"] / q, q / wq))
print("max over lanes of needed clusters: per visited group %.2f (union %.2f), per wave-query %.2f (union %.2f)" % (
    d["group_maxpop"] / d["groups_visited"], d["clusters_visited"] / d["groups_visited"], d["query_maxpop"] / wq, d["clusters_visited"] / wq))
cy = d["cycles"]; tot = sum(cy) or 1
names = ["refill", "prefix spheres", "culled spheres/rects/cylinders", "shading", "accumulate", "loop control"]
print("main-loop time shares (count kernel, s_memtime per wave): " + ", ".join(f"{n} {100*c/tot:.1f}%" for n, c in zip(names, cy)))
print("waves: start spread %.1f us, exit spread %.1f us, first start to last exit %.1f us (count kernel %.2f ms)" % (
    d["wave_start_spread_us"], d["wave_end_spread_us"], d["wave_span_us"], d["kernel_ms"]))
