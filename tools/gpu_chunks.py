"""Kernel time against samples per work item, whole frame and a 1/8 shard (run on the GPU box). usage: gpu_chunks.py [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
def best(o, n=3):
    sc.render(o)
    ts = []
    for _ in range(n):
        st = rtmi.Stats(); sc.render(o, st); ts.append(st.kernel_ms)
    return min(ts)
for chunk in (0, 256, 128, 64, 32, 16, 8):
    w = best(rtmi.Opts(seed=2023, spp_chunk=chunk))
    s = best(rtmi.Opts(seed=2023, spp_chunk=chunk, tile_first=0, tile_stride=8))
    print(f"chunk {chunk}: whole {w:.2f} ms, shard 0/8 {s:.2f} ms ({w / 8 / s * 100:.1f} % of whole/8 at this chunk)", flush=True)
