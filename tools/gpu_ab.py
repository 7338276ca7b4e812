"""A/B of two builds in ONE process order: alternate processes are needed (one library per process), so this
script times one library (RTMI_LIB) for a few repetitions; run it alternately from the shell."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
ts = []
for rep in range(4):
    st = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023), st); ts.append(st.kernel_ms)
print(os.path.basename(os.environ.get("RTMI_LIB", "librtmi.so (tree)")), " ".join(f"{t:.2f}" for t in ts), flush=True)
