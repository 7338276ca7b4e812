"""Queue-tail shape (RTMI_TAIL_FACTOR, RTMI_TAIL_DIV) against the whole frame and 1/2, 1/4, 1/8 shards (run on the GPU box).
usage: gpu_tail_sweep.py [spp] [FACTOR:DIV,...]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    rtmi = load_package()
    spp = int(sys.argv[2])
    sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
    out = []
    for N, r in ((1, 0), (2, 1), (4, 1), (8, 0), (8, 3)):
        o = rtmi.Opts(seed=2023, tile_first=r, tile_stride=N, tile_rotate=1) if N > 1 else rtmi.Opts(seed=2023)
        sc.render(o)
        ts = []
        for _ in range(4):
            st = rtmi.Stats(); sc.render(o, st); ts.append(st.kernel_ms)
        out.append(f"{r}/{N}: {min(ts):.2f}")
    print("  " + "   ".join(out), flush=True)
else:
    spp = sys.argv[1] if len(sys.argv) > 1 else "1024"
    specs = (sys.argv[2] if len(sys.argv) > 2 else "12:4,8:4,6:4,4:4,3:4,2:4,6:2,6:8,4:8").split(",")
    for spec in specs:
        f, d, *more = spec.split(":")
        env = dict(os.environ, RTMI_TAIL_FACTOR=f, RTMI_TAIL_DIV=d)
        for kv in more:  # FACTOR:DIV[:NAME=VALUE ...]
            k, v = kv.split("=")
            env[k] = v
        print(f"[factor {f} div {d} {' '.join(more)}]", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", spp], env=env, check=True)
