"""BASELINE config 5 (RTIOW 3840x2160, 8192 spp) on ONE GPU: it is specified for 8 GPUs, this is a capacity /
overflow check of the single-device path and a timing reference."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
sc = rtmi.Scene.rtiow(7, 3840, 2160, 8192, 50)
st = rtmi.Stats(); img = sc.render(rtmi.Opts(seed=2023), st)
n = 3840 * 2160 * 8192
print(f"config 5 on 1 GPU: {st.kernel_ms/1e3:.2f} s -> {n/st.kernel_ms/1e3:.0f} Msamples/s; mean {img.mean()/8192:.5f} finite {np.isfinite(img).all()} max {img.max()/8192:.3f}")
# one shard of 8 (what each GPU of the 8-GPU run renders)
st = rtmi.Stats(); loc = sc.render(rtmi.Opts(seed=2023, tile_first=3, tile_stride=8), st)
print(f"config 5, shard 3 of 8: {st.kernel_ms/1e3:.2f} s ({loc.shape[0]} rows) -> whole-job equivalent {n/st.kernel_ms/1e3:.0f} Msamples/s")
rows = sc.shard_global_rows(rtmi.Opts(tile_first=3, tile_stride=8))
print("shard equals rows of the full frame:", np.array_equal(loc, img[rows]))
