"""Work-item size (spp_chunk) sweep at bench size, whole frame and one of eight row shards (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
sc.render(rtmi.Opts(seed=2023))
for chunk in (64, 128, 256, 512):
    for name, kw in (("whole", {}), ("1/8 shard", dict(tile_first=0, tile_stride=8))):
        ms = []
        for _ in range(3):
            st = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023, spp_chunk=chunk, **kw), st); ms.append(st.kernel_ms)
        print(f"chunk {chunk} {name}: best {min(ms):.2f} ms {[round(m, 2) for m in ms]}", flush=True)
