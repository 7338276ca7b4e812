"""A/B of alternative builds of the library (RTMI_LIB) on one variant (run on the GPU box).
usage: gpu_ab.py <spp> <variant> <lib.so> [<lib.so> ...]     ('-' = the in-tree library)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    rtmi = load_package()
    spp, variant = int(sys.argv[2]), int(sys.argv[3])
    sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
    o = rtmi.Opts(seed=2023, variant=variant, spp_chunk=int(os.environ.get("RTMI_AB_CHUNK", "0")))  # (0: the library's choice)
    import zlib
    crc = zlib.crc32(sc.render(o).tobytes())
    ms = []
    for _ in range(3):
        st = rtmi.Stats(); sc.render(o, st); ms.append(st.kernel_ms)
    print(f"{os.environ.get('RTMI_LIB', 'in-tree')}: variant {variant} {spp} spp: best {min(ms):.2f} ms, all {[round(m, 2) for m in ms]}, image crc {crc:08x}", flush=True)
else:
    spp, variant = sys.argv[1], sys.argv[2]
    for lib in sys.argv[3:]:
        env = dict(os.environ)
        if lib != "-":
            env["RTMI_LIB"] = os.path.join(ROOT, lib)
        else:
            env.pop("RTMI_LIB", None)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", spp, variant], env=env, check=True)
