"""Frames in flight: a row-tile shard (and the whole frame) launched back to back on ONE stream against the same launches
alternating over TWO streams (a scene clone each, so each has its own accumulators): the drain of one frame overlaps the
ramp of the next (run on the GPU box).  usage: gpu_pipeline.py [N] [spp]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sc = rtmi.Scene.rtiow(7, 1920, 1080, spp, 50)
scenes = [sc, sc.clone(), sc.clone()]
bufs = [torch.empty((1080, 1920, 3), dtype=torch.float32, device="cuda:0") for _ in scenes]
streams = [torch.cuda.Stream() for _ in scenes]


def per_launch(o, n, depth):
    def go(k):
        for i in range(k):
            j = i % depth
            scenes[j].render_device(o, bufs[j].data_ptr(), streams[j].cuda_stream)
    go(2 * depth)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, o, n in (("whole", rtmi.Opts(seed=2023), 6), (f"shard 0/{N}", rtmi.Opts(seed=2023, tile_first=0, tile_stride=N), 24)):
    ts = {d: min(per_launch(o, n, d) for _ in range(2)) for d in (1, 2, 3)}
    print(f"{name}: " + ", ".join(f"{d} in flight {t:.2f} ms" for d, t in ts.items()), flush=True)
    if name == "whole":
        whole = ts
    else:
        for d in ts:
            print(f"  {d} in flight: shard against whole/{N} at the same depth {whole[d] / N / ts[d] * 100:.1f} %, against whole/{N} at depth 1 "
                  f"{whole[1] / N / ts[d] * 100:.1f} %", flush=True)
