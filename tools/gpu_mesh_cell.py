"""Cell-size sweep on the height-field meshes (RTMI_GRID_CELL scales the packer's choice; run on the GPU box).
usage: gpu_mesh_cell.py <quads per side,...> <factor,...>"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = sys.argv[1] if len(sys.argv) > 1 else "100,224"
for f in (sys.argv[2] if len(sys.argv) > 2 else "1.0,0.7,0.5").split(","):
    env = dict(os.environ, RTMI_GRID_CELL=f)
    print(f"[RTMI_GRID_CELL={f}]", flush=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_big.py"), "mesh", sizes], env=env, check=True)
