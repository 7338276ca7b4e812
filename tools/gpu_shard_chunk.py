"""One 1/N shard of the bench frame against the sample-chunk length and the queue's tail shape (run on the GPU box).
usage: gpu_shard_chunk.py [N] [spp] [rank] [chunk,...]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    rtmi = load_package()
    N, spp, r = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
    for chunk in [int(v) for v in sys.argv[5].split(",")]:
        o = rtmi.Opts(seed=2023, tile_first=r, tile_stride=N, tile_rotate=1, spp_chunk=chunk)
        sc.render(o)
        ts = []
        for _ in range(4):
            st = rtmi.Stats(); sc.render(o, st); ts.append(st.kernel_ms)
        print(f"  chunk {chunk}: {min(ts):.2f} ms {[round(t, 2) for t in ts]}", flush=True)
else:
    N = sys.argv[1] if len(sys.argv) > 1 else "8"
    spp = sys.argv[2] if len(sys.argv) > 2 else "1024"
    r = sys.argv[3] if len(sys.argv) > 3 else "3"
    chunks = sys.argv[4] if len(sys.argv) > 4 else "0,16,32,64,128,256"
    for spec in ["-", "RTMI_TAIL_MODE=0", "RTMI_TAIL_FACTOR=6", "RTMI_TAIL_FACTOR=24", "RTMI_ORPHAN_MAX=12", "RTMI_ORPHAN_MAX=32"]:
        env = dict(os.environ)
        if spec != "-":
            k, v = spec.split("="); env[k] = v
        print(f"[{spec}]", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", N, spp, r, chunks], env=env, check=True)
