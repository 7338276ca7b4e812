import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
sc = rtmi.Scene.dna(0.0); sc.override(width=1280, height=720, spp=256)
for v in (0, 16, 32, 40, 0, 16):
    ts = []
    for rep in range(4):
        st = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023, variant=v), st); ts.append(st.kernel_ms)
    print(f"dna 1280x720x256 variant {v}: " + " ".join(f"{t:.2f}" for t in ts), flush=True)
print("dna cluster size:", sc.count(rtmi.Opts(seed=1)).cull_cluster_size)
r = rtmi.Scene.rtiow(7, 640, 360, 4, 50); print("rtiow cluster size:", r.count(rtmi.Opts(seed=1)).cull_cluster_size)
