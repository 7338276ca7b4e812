"""Chunk / variant sweep at full bench size (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
chunks = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 64, 128, 256, 512]
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
for v in variants:
    for ch in chunks:
        st = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023, variant=v, spp_chunk=ch), st)
        print(f"rtiow 1920x1080x{spp} variant {v} chunk {ch}: {st.kernel_ms:.2f} ms -> {1920*1080*spp/st.kernel_ms/1e3:.1f} Msamples/s", flush=True)
