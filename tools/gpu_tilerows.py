"""Row-tile height of the N-way split: slowest shard against whole frame / N (run on the GPU box).
usage: gpu_tilerows.py [N] [spp] [tile_rows,...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
trs = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [8, 4, 2, 1]
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
def best(o, n=3):
    sc.render(o)
    ts = []
    for _ in range(n):
        st = rtmi.Stats(); sc.render(o, st); ts.append(st.kernel_ms)
    return min(ts)
whole = best(rtmi.Opts(seed=2023))
print(f"whole frame {whole:.2f} ms -> 1/{N} = {whole / N:.2f} ms", flush=True)
for tr in trs:
    ts = [best(rtmi.Opts(seed=2023, tile_rows=tr, tile_first=r, tile_stride=N)) for r in range(N)]
    print(f"tile_rows {tr}: shards {' '.join(f'{t:.2f}' for t in ts)} ms; slowest {max(ts):.2f} -> {whole / N / max(ts) * 100:.1f} % "
          f"(mean {sum(ts) / N:.2f} -> {whole / sum(ts) * 100:.1f} %)", flush=True)
