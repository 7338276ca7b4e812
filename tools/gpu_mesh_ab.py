"""A/B of library builds on the 20 000-triangle mesh and the DNA frame: gpu_mesh_ab.py <lib|-> ..."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from __graft_entry__ import load_package
    rtmi = load_package()
    from test_gpu_grid_all import height_field
    import zlib
    out = []
    for name, sc in (("mesh20000", height_field(rtmi, 100, 1280, 720, 16, depth=20)), ("mesh2048", height_field(rtmi, 32, 1280, 720, 16, depth=20))):
        ts = []
        for _ in range(3):
            st = rtmi.Stats(); img = sc.render(rtmi.Opts(seed=1), st); ts.append(st.kernel_ms)
        out.append(f"{name} v{st.kernel_variant} {min(ts):.2f} ms crc {zlib.crc32(img.tobytes()):08x}")
    d = rtmi.Scene.dna(0.0); d.override(1280, 720, 256, 50)
    ts = []
    for _ in range(3):
        st = rtmi.Stats(); d.render(rtmi.Opts(seed=2023), st); ts.append(st.kernel_ms)
    out.append(f"dna v{st.kernel_variant} {min(ts):.2f} ms")
    print(os.environ.get("RTMI_LIB", "in-tree"), "|", " | ".join(out), flush=True)
else:
    for lib in sys.argv[1:]:
        env = dict(os.environ)
        if lib != "-": env["RTMI_LIB"] = os.path.join(ROOT, lib)
        else: env.pop("RTMI_LIB", None)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
