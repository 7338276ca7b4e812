"""What one rank of an N-GPU strong-scaled run executes, timed on one GPU: the row-tile shard r = 0 of N
(interleaved 8-row tiles) of the bench frame, for N = 1, 2, 4, 8, against the whole frame / N."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
whole = None
ns = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1, 2, 4, 8]
for n in ns:
    ts = []
    for r in sorted({0, n - 1}):
        best = 1e9
        for rep in range(3):
            st = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023, tile_first=r, tile_stride=n, spp_chunk=chunk), st); best = min(best, st.kernel_ms)
        ts.append(best)
    t = max(ts)
    if n == 1 or whole is None: whole = t * n if n > 1 else t
    print(f"N={n}: slowest of ranks 0 and {n-1}: {t:.2f} ms; whole/N = {whole/n:.2f} ms; predicted efficiency {whole/n/t*100:.1f} %", flush=True)
