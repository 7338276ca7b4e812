"""Randomised schedule check: frame size, sample range, chunk size, shard geometry and variant drawn at random;
the kernel must equal the CPU checker bit for bit every time (the chunk / tail / streaming machinery only
decides who renders which sample when)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package
rtmi = load_package()
import rtcheck
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    w, h = int(rng.integers(2, 200)), int(rng.integers(2, 120))
    spp = int(rng.integers(1, 150))
    kind = int(rng.integers(0, 3))
    if kind == 0:
        sc = rtmi.Scene.rtiow(int(rng.integers(1, 50)), w, h, spp, int(rng.integers(1, 30)))
    elif kind == 1:
        sc = rtmi.Scene.load(os.path.join(ROOT, "ray-tracing-in-cuda_amd/scenes/mixed_emissive.json")); sc.override(w, h, spp, int(rng.integers(1, 20)))
    else:
        sc = rtmi.Scene.dna(float(rng.uniform(0, 360))); sc.override(w, h, spp, 8)
    if rng.random() < 0.3: sc.set_russian_roulette(float(rng.uniform(0.3, 1.0)))
    first = int(rng.integers(0, 5)) if rng.random() < 0.5 else 0
    count = int(rng.integers(1, spp + 1))
    chunk = int(rng.choice([0, 0, 1, 3, 7, 8, 16, 17, 32, 50, 64, 128]))
    tile_rows = int(rng.choice([1, 3, 8, 8, 16]))
    stride = int(rng.integers(1, 6)); tf = int(rng.integers(0, stride))
    # (the compact-table variants for the sphere-only scene, the wide-table ones for the others: tests/test_gpu_fuzz.py)
    variant = int(rng.choice([0, 0, 0, 1, 2, 6, 64, 32, 40, 128] if kind == 0 else [0, 0, 0, 36, 44, 16, 64, 32, 128]))
    deal = int(rng.integers(0, 3))  # rt_opts.tile_rotate
    o = rtmi.Opts(seed=int(rng.integers(0, 2**31)), sample_first=first, sample_count=count, spp_chunk=chunk,
                  tile_rows=tile_rows, tile_first=tf, tile_stride=stride, tile_rotate=deal, variant=variant)
    rows = sc.shard_global_rows(o)
    img = sc.render(o)
    ref, _ = rtcheck.oracle_render(sc, seed=o.seed, sample_first=first, sample_count=count)
    ok = img.shape[0] == len(rows) and (len(rows) == 0 or np.array_equal(img, ref[rows]))
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: {w}x{h} spp {spp} kind {kind} first {first} count {count} chunk {chunk} tile_rows {tile_rows} shard {tf}/{stride} deal {deal} variant {variant}", flush=True)
print(f"{n_cases} random cases, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
