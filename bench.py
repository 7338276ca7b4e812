#!/usr/bin/env python3
"""bench.py -- the hot path's headline metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One STEP = one full frame of the workload: every rank renders its interleaved row tiles
with the HIP kernel (through the C ABI, into a torch tensor on torch's current stream),
then ONE gather (RCCL over xGMI) assembles the framebuffer on rank 0.  Inputs (the scene
image) are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[2], the configuration the north-star target is quoted on):
RTIOW final random-spheres scene (rt_scene_rtiow(7): ~485 spheres, checker ground, defocus
blur, sky gradient), 1920x1080, 1024 spp, depth 50.  Scaling is STRONG: the same frame is
split over N GPUs.

metric   Msamples/s = W*H*spp / step seconds / 1e6          (whole job, all ranks)
roofline fp32 vector ALU (not HBM, not MFMA: the path is a scalar-per-lane bounce loop).
         achieved = algorithmic flops of one launch / mean kernel time (HIP events on the
         launch stream, rt_stats.kernel_ms), flops from the accounting of DESIGN.md with
         EXACT event counts from the diagnostic counting kernel;  peak 157.3 TFLOP/s.
cpu_baseline  the reference's own cmake-cpu-version render() loop (oracle/_ref, "reference")
         timed on this host's cores on a bounded sample of the same scene.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md chip table: 256 CU x 4 SIMD x 32 lanes x 2 x 2.4 GHz
PEAK_HBM_GBPS = 8000.0

# algorithmic flop accounting (DESIGN.md "flop accounting"):
# add/sub/mul = 1, fma = 2, div/sqrt = 1, compares/selects/negations/integer RNG = 0
F_SAMPLE = 46          # jitter, camera ray, accumulate
F_TEST = {0: 17, 1: 9, 2: 9, 3: 9, 4: 60}  # per ray-primitive test, by rt_prim_type
F_HIT = 30             # hit record of the accepted closest hit
F_SCATTER = [40, 55, 65, 0]  # lambertian, metal, dielectric, diffuse_light
F_MISS = 25            # sky / background evaluation
F_BOX = 27             # slab test of one cluster box (6 sub, 6 mul, 6 min/max, 4 reduce, compare)


def algorithmic_flops(counts: dict, prim_types) -> float:
    per_query = sum(F_TEST[int(t)] for t in prim_types)
    return (F_SAMPLE * counts["samples"] + per_query * counts["queries"] + F_HIT * counts["hits"]
            + sum(f * b for f, b in zip(F_SCATTER, counts["scatter"])) + F_MISS * counts["misses"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--chunk", type=int, default=-1, help="spp_chunk (-1: default policy)")
    ap.add_argument("--tile-rows", type=int, default=8)
    ap.add_argument("--seed", type=int, default=2023)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-counts", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=64)
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (16 = linear scan without AABB culling)")
    ap.add_argument("--verify", action="store_true", help="rank 0 also renders the unsharded frame and checks the "
                    "gathered one against it bit for bit (outside the timed region)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) | gloo (rehearsal: ranks may share one GPU)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package

    rtmi = load_package()
    import importlib
    rdist = importlib.import_module("rtmi.dist")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the render path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and args.backend == "nccl":
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    dev_index = local_rank % ndev  # identity on a real N-GPU node; gloo rehearsals share GPUs
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    scene = rtmi.Scene.rtiow(7, args.width, args.height, args.spp, args.depth)
    chunk = args.chunk if args.chunk >= 0 else default_chunk(args.spp)
    base = rtmi.Opts(seed=args.seed, device=dev_index, tile_rows=args.tile_rows, spp_chunk=chunk, variant=args.variant)
    mine = rdist.shard_opts(base, rank, world)
    local = rdist.alloc_local(scene, base, world, device)
    full = torch.empty((scene.height, scene.width, 3), dtype=torch.float32, device=device) if rank == 0 else None
    stream = torch.cuda.current_stream().cuda_stream
    kernel_ms = []

    def step(record: bool):
        st = rtmi.Stats()
        scene.render_device(mine, local.data_ptr(), stream, st)
        if record:
            kernel_ms.append(st.kernel_ms)
        return rdist.gather_framebuffer(local, scene, base, rank, world, out=full, via_host=args.backend != "nccl")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # one-time set-up outside any timed or warm-up step: scene image upload, accumulator allocation and
    # occupancy query, and the communicator's first collective.  It is the same launch as a step, so that
    # every render_kernel row of a `rocprofv3 --stats` summary of this command is one step's launch.
    prime = rdist.shard_opts(base, rank, world)
    scene.render_device(prime, local.data_ptr(), stream, rtmi.Stats())
    rdist.gather_framebuffer(local, scene, base, rank, world, out=full, via_host=args.backend != "nccl")

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img = step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    samples_per_step = scene.width * scene.height * scene.spp
    ms_per_step = elapsed / args.steps * 1e3
    value = samples_per_step / (elapsed / args.steps) / 1e6

    result = {
        "metric": "Msamples/sec (WxHxspp)",
        "value": round(value, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"RTIOW random-spheres scene (rt_scene_rtiow seed 7, {scene.info.num_prims} spheres), "
                        f"{scene.width}x{scene.height}, {scene.spp} spp, depth {scene.max_depth}",
            "sharding": f"row tiles of {args.tile_rows} rows interleaved over {world} rank(s), one gather to rank 0 "
                        f"({args.backend})",
            "spp_chunk": chunk,
            "kernel_variant": args.variant,
            "render_seed": args.seed,
        },
    }

    if rank == 0 and args.verify:
        whole = scene.render(rtmi.Opts(seed=args.seed, device=dev_index, spp_chunk=chunk, variant=args.variant))
        same = bool(np.array_equal(whole, img.cpu().numpy()))
        result["config"]["gathered_equals_unsharded"] = same
        assert same, "gathered row tiles differ from the unsharded frame"

    if rank == 0:
        # sanity of the measured frames themselves (not a parity test: those live in tests/)
        mean = float(img.mean().item()) / scene.spp
        result["config"]["frame_mean_radiance"] = round(mean, 5)
        assert np.isfinite(mean) and 0.2 < mean < 0.8, mean

        # ---- roofline of the dominant kernel (rank 0's launch)
        roof = {"bound": "valu_fp32", "achieved": None, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": None,
                "traffic": None,
                "bound_note": "fp32 vector ALU (BASELINE.json metric and SURVEY 8(d): the path is not a contraction, "
                              "so neither the MFMA nor the HBM roofline binds; HBM share below as hbm_frac)"}
        k_ms = float(np.mean(kernel_ms))
        roof["kernel_ms_avg"] = round(k_ms, 3)
        if not args.no_counts:
            st = scene.count(mine)  # exact event counts of rank 0's shard (same seed, same samples; culled kernel)
            c = st.as_dict()
            linear_flops = algorithmic_flops(c, scene.prims()["type"])
            # the ray-primitive tests the kernel executes, counted per LANE (one ray against one sphere or box):
            # every live lane tests the always-tested prefix, every outer box, the cluster boxes of the outer
            # boxes some lane of its wave passed, and the spheres of the clusters ITS OWN ray reaches
            wq = c["wave_queries"]
            lanes = c["queries"] / max(1, wq)  # live lanes per wave-level query
            culled = args.variant & 16 == 0
            if culled:
                sphere_tests = c["queries"] * c["cull_prefix"] + c["lane_clusters"] * c["cull_cluster_size"]
                box_tests = c["queries"] * c["cull_groups"] + lanes * c["groups_visited"] * 4
            else:
                sphere_tests = c["queries"] * (c["cull_prefix"] + c["cull_clusters"] * c["cull_cluster_size"])
                box_tests = 0
            per_query_linear = sum(F_TEST[int(t)] for t in scene.prims()["type"])
            flops = 17 * sphere_tests + F_BOX * box_tests + linear_flops - per_query_linear * c["queries"]
            roof["mode"] = "aabb-culled hittable_list, per-lane cluster lists (default)" if culled else "linear hittable_list scan (variant 16)"
            if args.variant in (8, 32):  # the diagnostic kernel is the default one: its counts do not describe these
                roof["mode"] = "ablation variant with wave-level cluster votes: no flop count"
                flops = 0
            roof["achieved"] = round(flops / (k_ms * 1e-3) / 1e12, 3) if flops else None
            roof["frac"] = round(roof["achieved"] / PEAK_FP32_TFLOPS, 4) if flops else None
            roof["flops_per_launch"] = int(flops)
            roof["accounting"] = ("flops = 46 S + 17 T_sphere + 27 T_box + 30 H + 40/55/65 B + 25 M with the sphere and "
                                  "box tests the kernel's lanes execute (masked-off lanes not counted), all counted "
                                  "exactly by the diagnostic kernel")
            roof["counts"] = {k: c[k] for k in ("samples", "queries", "prim_tests", "hits", "misses", "scatter",
                                                "rng_draws", "wave_queries", "clusters_visited", "groups_visited",
                                                "lane_clusters", "lane_groups")}
            roof["tests_per_sample"] = {"sphere": round(sphere_tests / max(1, c["samples"]), 1),
                                        "box": round(box_tests / max(1, c["samples"]), 1),
                                        "reference_linear_scan": round(c["prim_tests"] / max(1, c["samples"]), 1)}
            roof["lane_occupancy_of_queries"] = round(c["queries"] / max(1, 64 * wq), 4)
            # for comparison only: what the reference's O(N) scan (T = queries x N) would have
            # to sustain to render the same frame in the same time
            roof["reference_linear_scan_equivalent"] = {
                "tflops": round(linear_flops / (k_ms * 1e-3) / 1e12, 3),
                "frac_of_peak": round(linear_flops / (k_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, 4)}
            roof["valu_issue_peak_Tlane_inst"] = 78.6
        # algorithmic HBM bytes: framebuffer write once + scene image read once per workgroup (L2-resident)
        rows0 = scene.shard_rows(mine)
        plane = rows0 * scene.width * 3
        n_chunks = -(-scene.spp // (chunk or 64))
        # per launch: clear the 64-bit accumulators, one 8-byte atomic per pixel-channel and work item,
        # read them back and write the fp32 frame
        algo_bytes = plane * (8 + 8 * n_chunks + 8 + 4)
        roof["hbm_algorithmic_bytes"] = int(algo_bytes)
        roof["hbm_achieved_GBps"] = round(algo_bytes / (k_ms * 1e-3) / 1e9, 3)
        roof["hbm_frac"] = round(roof["hbm_achieved_GBps"] / PEAK_HBM_GBPS, 6)
        traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(traffic_file):
            try:
                tf = json.load(open(traffic_file))
                key = f"{scene.width}x{scene.height}x{scene.spp}"
                if key in tf:
                    roof["traffic"] = tf[key]["bytes_per_launch"]
                    roof["traffic_source"] = tf[key].get("source")
            except Exception:
                pass
        result["roofline"] = roof

        # ---- CPU baseline (reported only): the reference's own render() on this host's cores
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(rtmi, args)

        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def default_chunk(spp: int) -> int:
    """Samples per work item (8x8 tile x chunk, one wave): 0 = the library default (64).
    Scheduling only -- the fixed-point pixel sums do not depend on it."""
    return 0


def cpu_baseline(rtmi, args):
    import rtcheck
    cores = min(len(os.sched_getaffinity(0)), 16)
    spp = args.cpu_spp
    sc = rtmi.Scene.rtiow(7, args.width, args.height, spp, args.depth)
    # bounded sample: every 4th band of 8 rows of the full-size frame at reduced spp
    # (cost per sample does not depend on spp); ~4 M samples
    bands = [(y, min(y + 8, args.height)) for y in range(0, args.height, 32)]
    n_samples = sum((b - a) for a, b in bands) * args.width * spp
    if rtcheck.have_ref():
        rs = rtcheck.RefScene(sc)
        sec = 0.0
        for a, b in bands:
            s, _ = rs.time_rows(args.seed, a, b, spp, threads=cores)
            sec += s
        kind = "reference"
        what = "cmake-cpu-version render() per pixel (oracle/_ref: the reference's sources + hooked rand())"
    else:
        osc = rtcheck.OracleScene(sc)
        t0 = time.perf_counter()
        for a, b in bands:
            rtcheck.oracle_render(osc, seed=args.seed, rows=(a, b), threads=cores)
        sec = time.perf_counter() - t0
        kind = "port"
        what = "fp32 CPU restatement (oracle/rt_oracle.c)"
    return {
        "value": round(n_samples / sec / 1e6, 4),
        "unit": "Msamples/s",
        "cores": cores,
        "kind": kind,
        "sample": f"{what}; same scene at {args.width}x{args.height}, every 4th 8-row band, {spp} spp "
                  f"({n_samples} samples, {sec:.1f} s, OpenMP over rows)",
    }


if __name__ == "__main__":
    main()
