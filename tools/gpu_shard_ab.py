"""A/B of library builds (RTMI_LIB) on a 1/N row-tile shard and the whole frame (run on the GPU box).
usage: gpu_shard_ab.py <N> <spp> <lib.so|-> [...]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    rtmi = load_package()
    N, spp = int(sys.argv[2]), int(sys.argv[3])
    sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
    def best(o, n=5):
        sc.render(o)
        ts = []
        for _ in range(n):
            st = rtmi.Stats(); sc.render(o, st); ts.append(st.kernel_ms)
        return min(ts)
    whole = best(rtmi.Opts(seed=2023), 3)
    s0, s7 = best(rtmi.Opts(seed=2023, tile_first=0, tile_stride=N)), best(rtmi.Opts(seed=2023, tile_first=N - 1, tile_stride=N))
    print(f"{os.environ.get('RTMI_LIB', 'in-tree')}: whole {whole:.2f} ms, shard 0/{N} {s0:.2f} ms ({whole / N / s0 * 100:.1f} %), "
          f"shard {N - 1}/{N} {s7:.2f} ms ({whole / N / s7 * 100:.1f} %)", flush=True)
else:
    for lib in sys.argv[3:]:
        env = dict(os.environ)
        if lib != "-":
            env["RTMI_LIB"] = os.path.join(ROOT, lib)
        else:
            env.pop("RTMI_LIB", None)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", sys.argv[1], sys.argv[2]], env=env, check=True)
