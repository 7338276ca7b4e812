"""Launch-tail study: kernel time of a full-frame RTIOW render for several sample counts, chunk sizes and
tail modes (RTMI_TAIL_MODE is read per render call).  What one rank of an N-GPU run sees is the 1024/N row."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
chunks = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"])]
modes = (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1"])
for (w, h, spp) in [(1920, 1080, 128), (1920, 1080, 1024)]:
    sc = rtmi.Scene.rtiow(7, w, h, spp, 50)
    for mode in modes:
        os.environ["RTMI_TAIL_MODE"] = mode
        for chunk in chunks:
            ts = []
            for rep in range(4):
                st = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023, spp_chunk=chunk), st); ts.append(st.kernel_ms)
            ts.sort()
            print(f"tail_mode={mode} chunk={chunk} {w}x{h}x{spp}: min {ts[0]:.2f} median {ts[1]:.2f} ms", flush=True)
