"""Lifetimes of the waves of one launch (the counting kernel's clocks; RTMI_DEBUG_DRAIN prints the histograms on stderr).
usage: gpu_drain.py [N] [spp] [rank]      N = 1: the whole frame"""
import os, sys
os.environ["RTMI_DEBUG_DRAIN"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
r = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
o = rtmi.Opts(seed=2023, tile_first=r, tile_stride=N, tile_rotate=1) if N > 1 else rtmi.Opts(seed=2023)
st = rtmi.Stats(); sc.render(o, st); sc.render(o, st)
print(f"product kernel: {st.kernel_ms:.2f} ms", flush=True)
c = sc.count(o)
print(f"counting kernel: start spread {c.wave_start_spread_us:.0f} us, end spread {c.wave_end_spread_us:.0f} us, span {c.wave_span_us:.0f} us", flush=True)
