"""Does a short launch run at a lower clock?  Kernel time of the 1/8 row shard (1024 spp) launched 20 times in a
row, then with 50 ms of host sleep before each launch, then right after a long (whole-frame) launch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
sc = rtmi.Scene.rtiow(7, 1920, 1080, 1024, 50)
o8 = rtmi.Opts(seed=2023, tile_first=0, tile_stride=8)
def t(o):
    st = rtmi.Stats(); sc.render(o, st); return st.kernel_ms
print("back to back :", " ".join(f"{t(o8):.2f}" for _ in range(20)), flush=True)
def slept():
    time.sleep(0.05); return t(o8)
print("50 ms sleeps :", " ".join(f"{slept():.2f}" for _ in range(8)), flush=True)
w = t(rtmi.Opts(seed=2023))
print(f"whole frame  : {w:.2f} (/8 = {w/8:.2f})", flush=True)
print("after whole  :", " ".join(f"{t(o8):.2f}" for _ in range(5)), flush=True)
