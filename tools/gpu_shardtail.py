"""Where a 1/N row-tile shard loses time against whole-frame / N (run on the GPU box): wave start / exit spreads of the
diagnostic kernel and the kernel time at several chunk sizes.  usage: gpu_shardtail.py [N] [spp]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
def best(o, n=4):
    sc.render(o)
    ts = []
    for _ in range(n):
        st = rtmi.Stats(); sc.render(o, st); ts.append(st.kernel_ms)
    return min(ts)
whole = best(rtmi.Opts(seed=2023), 3)
print(f"whole frame kernel {whole:.2f} ms -> /{N} = {whole / N:.2f}", flush=True)
for chunk in (0, 64, 32, 16, 8):
    for r in (0, N - 1):
        o = rtmi.Opts(seed=2023, tile_first=r, tile_stride=N, spp_chunk=chunk)
        ms = best(o)
        c = sc.count(o)
        print(f"N={N} shard {r} chunk {chunk}: kernel {ms:.2f} ms ({whole / N / ms * 100:.1f} %), diagnostic kernel: wave start spread "
              f"{c.wave_start_spread_us:.0f} us, exit spread {c.wave_end_spread_us:.0f} us, span {c.wave_span_us:.0f} us", flush=True)
