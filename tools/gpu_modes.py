"""Kernel variants at bench size: time, counters, image CRC (run on the GPU box).
usage: gpu_modes.py [spp] [variants]   e.g. gpu_modes.py 256 0,6,40,1,128,64,32,16"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 6, 40, 1, 128, 64, 32, 16]
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
for v in variants:
    o = rtmi.Opts(seed=2023, variant=v)
    sc.render(o)
    best = 1e9
    for _ in range(3):
        st = rtmi.Stats(); img = sc.render(o, st); best = min(best, st.kernel_ms)
    line = f"variant {v} -> {st.kernel_variant}: {best:.2f} ms -> {1920*1080*spp/best/1e3:.0f} Msamples/s crc {zlib.crc32(img.tobytes()):08x}"
    if v in (0, 2, 6, 64, 128):
        c = sc.count(o).as_dict()
        q, wq = c["queries"], max(1, c["wave_queries"])
        line += (f" | per query: tests {c['lane_clusters']/q:.2f} steps {c['lane_cands']/q:.2f} | per wave-query: test passes "
                 f"{c['clusters_visited']/wq:.2f} step passes {c['groups_visited']/wq:.2f}")
    print(line, flush=True)
