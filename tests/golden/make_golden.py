"""Generates the committed fixtures under tests/golden/ -- run HERE (the container that has
/root/reference), never on the GPU box:

    python tests/golden/make_golden.py

* scenes/*.json           the reference's gpu-version scene files (data, not source), re-serialised
                          through rt_scene_load_json -> rt_scene_to_json (normalised text)
* rtiow_seed7.json        the RTIOW scene of rt_scene_rtiow(7) the vectors below were made on
* ref_three_sphere.npz    64x36x16spp fp64 image + 2048 per-sample radiance triples from the
                          COMPILED REFERENCE (oracle/_ref, cmake-cpu-version sources + hooked rand())
* ref_rtiow.npz           48x27x8spp fp64 image + 2048 per-sample triples, same source
* ref_write_color.npz     write_color() (color.h:14-35) outputs for a sweep of pixel sums
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402

rtmi = load_package()
import rtcheck  # noqa: E402

REF = "/root/reference/gpu-version"
SEED = 2023


def normalise_scenes():
    os.makedirs(os.path.join(HERE, "scenes"), exist_ok=True)
    for name in ("sample_scene", "basic_scene", "blue", "blue2"):
        sc = rtmi.Scene.load(os.path.join(REF, name + ".json"))
        with open(os.path.join(HERE, "scenes", name + ".json"), "w") as f:
            f.write(sc.to_json())
        print("scene", name, sc.info.num_prims, "objects")


def ref_vectors(scene, name, n_samples):
    rs = rtcheck.RefScene(scene)
    img = rs.render(seed=SEED)
    rng = np.random.default_rng(12345)
    ids = np.stack([rng.integers(0, scene.width, n_samples), rng.integers(0, scene.height, n_samples),
                    rng.integers(0, scene.spp, n_samples)], axis=1).astype(np.int32)
    vals = np.zeros((n_samples, 3))
    draws = np.zeros(n_samples, dtype=np.int32)
    for k, (x, y, s) in enumerate(ids):
        vals[k], draws[k] = rs.sample(SEED, int(x), int(y), int(s))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), image_sum=img, seed=SEED, width=scene.width,
                        height=scene.height, spp=scene.spp, sample_ids=ids, sample_rgb=vals, sample_draws=draws)
    print(name, "image mean", img.mean() / scene.spp, "samples", n_samples)


def main():
    normalise_scenes()
    sc = rtmi.Scene.load(os.path.join(ROOT, "ray-tracing-in-cuda_amd", "scenes", "three_sphere.json"))
    sc.override(width=64, height=36, spp=16)
    ref_vectors(sc, "ref_three_sphere", 2048)

    sc = rtmi.Scene.rtiow(7, 48, 27, 8, 50)
    with open(os.path.join(HERE, "rtiow_seed7.json"), "w") as f:
        f.write(sc.to_json())
    ref_vectors(sc, "ref_rtiow", 2048)

    sums = np.concatenate([np.linspace(0, 40, 200), [1e-9, 15.99, 16.0, 16.01, 100.0]])
    spp = 16
    out = np.array([rtcheck.ref_write_color([s, s / 2, s / 3], spp) for s in sums], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "ref_write_color.npz"), sums=sums, spp=spp, out=out)
    print("write_color", out[:3].tolist(), out[-3:].tolist())


if __name__ == "__main__":
    main()
