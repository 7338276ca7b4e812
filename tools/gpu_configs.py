"""Launch time of the other BASELINE configurations and of the reference's scene files (run on the GPU box).
usage: gpu_configs.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
golden = os.path.join(ROOT, "tests", "golden", "scenes")
cases = []
c1 = rtmi.Scene.load(os.path.join(ROOT, "ray-tracing-in-cuda_amd", "scenes", "three_sphere.json")); c1.override(400, 225, 100, 50)
cases.append(("config1 three_sphere 400x225x100", c1))
c2 = rtmi.Scene.dna(0.0); c2.override(1280, 720, 256, 50)
cases.append(("config2b dna 1280x720x256", c2))
c4 = rtmi.Scene.load(os.path.join(golden, "sample_scene.json")); c4.override(1920, 1080, 512, 50)
cases.append(("config4 sample_scene 1920x1080x512", c4))
for name in ("blue", "blue2"):
    b = rtmi.Scene.load(os.path.join(golden, name + ".json")); b.override(1280, 720, 128, 50)
    cases.append((f"{name}.json 1280x720x128", b))
m = rtmi.Scene.load(os.path.join(ROOT, "ray-tracing-in-cuda_amd", "scenes", "mixed_emissive.json")); m.override(1280, 720, 128, 50)
cases.append(("mixed_emissive 1280x720x128", m))
for what, sc in cases:
    ts = []
    for _ in range(reps + 1):
        st = rtmi.Stats(); sc.render(rtmi.Opts(seed=2023), st); ts.append(st.kernel_ms)
    n = sc.width * sc.height * sc.spp
    print(f"{what}: kernel variant {st.kernel_variant}, best {min(ts[1:]):.2f} ms ({n / min(ts[1:]) / 1e3:.0f} Msamples/s)", flush=True)
