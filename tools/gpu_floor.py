"""Loop-floor probe: camera looks away from the RTIOW spheres so every sample is ONE closest-hit
query that misses everything: time / wave-tests = cost of the bare primitive loop."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4, 8]
spp = 256
d = json.loads(rtmi.Scene.rtiow(7, 1920, 1080, spp, 50).to_json())
d["camera"]["lookfrom"] = [0, 30, 0]; d["camera"]["lookat"] = [0, 60, 1]; d["camera"]["aperture"] = 0.0
sc = rtmi.Scene.parse(json.dumps(d))
n = sc.info.num_prims
for v in variants:
    for rep in range(2):
        st = rtmi.Stats(); sc.render(rtmi.Opts(seed=1, variant=v, spp_chunk=128), st)
    c = sc.count(rtmi.Opts(seed=1, spp_chunk=128)) if v == variants[0] and False else None
    wt = 1920 * 1080 * spp * n / 64
    cyc = st.kernel_ms * 1e-3 * 2.4e9 * 1024 / wt
    print(f"variant {v}: {st.kernel_ms:.2f} ms  {1920*1080*spp/st.kernel_ms/1e3:.0f} Msamples/s  {cyc:.1f} SIMD-cycles per wave-test ({n} prims)", flush=True)
c = sc.count(rtmi.Opts(seed=1))
print("queries/sample", c.queries / c.samples, "hits", c.hits)
