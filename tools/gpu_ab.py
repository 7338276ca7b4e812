"""A/B timing of kernel variants on the bench scene (run on the GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
rtmi = load_package()
import rtcheck
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1]
chunks = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
# parity first (small)
sc = rtmi.Scene.rtiow(7, 160, 90, 8, 50)
ref, _ = rtcheck.oracle_render(sc, seed=2023)
for v in variants:
    img = sc.render(rtmi.Opts(seed=2023, variant=v))
    print(f"variant {v}: bit-identical to checker = {np.array_equal(img, ref)}", flush=True)
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
for rep in range(2):
    for v in variants:
        for ch in chunks:
            st = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023, variant=v, spp_chunk=ch), st)
            print(f"rtiow 1920x1080x{spp} variant {v} chunk {ch}: {st.kernel_ms:.2f} ms -> {1920*1080*spp/st.kernel_ms/1e3:.1f} Msamples/s", flush=True)
