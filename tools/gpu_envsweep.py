"""Environment-knob sweep on the whole frame and on 1/N shards (run on the GPU box).
usage: gpu_envsweep.py <N> <spp> "K=V,K=V" ["K=V" ...]      ('-' = no setting)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    rtmi = load_package()
    N, spp = int(sys.argv[2]), int(sys.argv[3])
    sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
    def best(o, n=4):
        sc.render(o)
        ts = []
        for _ in range(n):
            st = rtmi.Stats(); sc.render(o, st); ts.append(st.kernel_ms)
        return min(ts)
    whole = best(rtmi.Opts(seed=2023), 3)
    s0 = best(rtmi.Opts(seed=2023, tile_first=0, tile_stride=N))
    s3 = best(rtmi.Opts(seed=2023, tile_first=3, tile_stride=N))
    print(f"{sys.argv[4]:40s} whole {whole:7.2f} ms, shard 0/{N} {s0:6.2f} ms, shard 3/{N} {s3:6.2f} ms", flush=True)
else:
    for setting in sys.argv[3:]:
        env = dict(os.environ)
        if setting != "-":
            for kv in setting.split(","):
                k, v = kv.split("=")
                env[k] = v
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", sys.argv[1], sys.argv[2], setting], env=env, check=True)
