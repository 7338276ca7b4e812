"""Candidate-search modes x cluster sizes at bench size: time, counters, equality (run on the GPU box).
usage: gpu_modes.py [spp] [variants] [clusters]   e.g. gpu_modes.py 256 0,64,16 8,16"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 64]
clusters = sys.argv[3].split(",") if len(sys.argv) > 3 else ["8", "16"]
if len(sys.argv) > 4 and sys.argv[4] == "child":
    import numpy as np, zlib
    from __graft_entry__ import load_package
    rtmi = load_package()
    sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
    for v in variants:
        o = rtmi.Opts(seed=2023, variant=v)
        sc.render(o)
        best = 1e9
        for _ in range(3):
            st = rtmi.Stats(); img = sc.render(o, st); best = min(best, st.kernel_ms)
        crc = zlib.crc32(img.tobytes())
        line = f"cluster {os.environ.get('RTMI_CLUSTER')} variant {v}: {best:.2f} ms -> {1920*1080*spp/best/1e3:.0f} Msamples/s crc {crc:08x}"
        if v in (0, 1, 40, 64, 104, 128, 136):
            c = sc.count(o).as_dict()
            q, wq = c["queries"], max(1, c["wave_queries"])
            line += (f" | per query: cands {c['lane_cands']/q:.2f} clusters {c['lane_clusters']/q:.2f} windows {c['lane_groups']/q:.2f}"
                     f" | per wave-query: refine rounds {c['groups_visited']/wq:.2f} walk rounds {c['clusters_visited']/wq:.2f} max-pop {c['query_maxpop']/wq:.2f}"
                     f" | cycles% {[round(100*x/max(1,sum(c['cycles'])),1) for x in c['cycles']]}")
        print(line, flush=True)
else:
    for cl in clusters:
        env = dict(os.environ, RTMI_CLUSTER=cl)
        subprocess.run([sys.executable, os.path.abspath(__file__), str(spp), ",".join(map(str, variants)), cl, "child"], env=env, check=True)
