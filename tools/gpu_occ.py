import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
sc = rtmi.Scene.rtiow(7, 1920, 1080, 256, 50)
for ch in (16, 64, 256):
    st = sc.count(rtmi.Opts(seed=2023, spp_chunk=ch))
    t = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023, spp_chunk=ch), t)
    print(f"chunk {ch}: lane occupancy {st.queries/(64*st.wave_queries):.4f} wave_queries {st.wave_queries} clusters/wq {st.clusters_visited/st.wave_queries:.2f} groups/wq {st.groups_visited/st.wave_queries:.2f} render {t.kernel_ms:.1f} ms count-kernel {st.kernel_ms:.1f} ms", flush=True)
