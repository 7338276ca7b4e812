"""Fixed cost of a launch: kernel time against samples per pixel, whole frame and one 1/8 shard (run on the GPU box);
the intercept of the straight line is what every launch pays however little it renders."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
def best(sc, o, n=4):
    sc.render(o)
    ts = []
    for _ in range(n):
        st = rtmi.Stats(); sc.render(o, st); ts.append(st.kernel_ms)
    return min(ts)
for name, kw in (("whole", {}), ("shard 0/8", dict(tile_first=0, tile_stride=8)), ("shard 7/8", dict(tile_first=7, tile_stride=8))):
    xs, ys = [], []
    for spp in (64, 128, 256, 512, 1024):
        sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
        ms = best(sc, rtmi.Opts(seed=2023, **kw))
        xs.append(spp); ys.append(ms)
    b, a = np.polyfit(xs[1:], ys[1:], 1)
    print(f"{name}: " + ", ".join(f"{x} spp {y:.2f} ms" for x, y in zip(xs, ys)) + f" | fit (128..1024): {a:.2f} ms + {b * 1000:.2f} us/spp", flush=True)
