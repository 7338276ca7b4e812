"""A/B of environment knobs of the measurement build on the bench scene (run on the GPU box).
usage: gpu_env_ab.py <spp> <variant> NAME=VALUE[,NAME=VALUE...] ...     ('-' = no knob)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spp, variant = sys.argv[1], sys.argv[2]
for spec in sys.argv[3:]:
    env = dict(os.environ)
    if spec != "-":
        for kv in spec.split(","):
            k, v = kv.split("=")
            env[k] = v
    print(f"[{spec}]", flush=True)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_ab.py"), "child", spp, variant], env=env, check=True)
