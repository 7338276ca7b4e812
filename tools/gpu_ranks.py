"""Kernel time of every rank's shard of an N-GPU strong-scaled frame, timed one after the other on one GPU,
for several row-tile heights: how well do interleaved tiles balance, and what does a finer interleave cost?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc = rtmi.Scene.rtiow(7, 1920, 1080, 1024, 50)
st = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023), st); sc.render(rtmi.Opts(seed=2023), st); whole = st.kernel_ms
print(f"whole frame {whole:.2f} ms; /{n} = {whole/n:.2f}", flush=True)
for tr in (8, 4, 2, 16):
    ts = []
    for r in range(n):
        best = 1e9
        for rep in range(2):
            st = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023, tile_rows=tr, tile_first=r, tile_stride=n), st); best = min(best, st.kernel_ms)
        ts.append(best)
    print(f"tile_rows {tr}: ranks " + " ".join(f"{t:.2f}" for t in ts) + f"  max {max(ts):.2f} mean {sum(ts)/n:.2f} -> efficiency {whole/n/max(ts)*100:.1f} %", flush=True)
