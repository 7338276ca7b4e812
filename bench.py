#!/usr/bin/env python3
"""bench.py -- the hot path's headline metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts N rank processes itself (one per
GPU; this process never touches the GPU, each rank checks that its device exists).

`--tiles-abi` times the multi-GPU path the C ABI exports instead: ONE process, rt_render_hip_tiles over
`--gpus N` devices (one stream per device, ncclCommInitAll, ONE ncclGather, row placement, one copy to the
caller's host buffer).  At N = 1 the default mode also reports it, next to the headline, under `extra`.

One STEP = one full frame of the workload: every rank renders its interleaved row tiles
with the HIP kernel (through the C ABI, into a torch tensor on torch's current stream),
then ONE gather (RCCL over xGMI) assembles the framebuffer on rank 0.  Nothing in a step waits on
the host.  Inputs (the scene image) are resident in HBM before the timed region.  Rank 0 prints
ONE JSON line.

Workload (BASELINE.json configs[2], the configuration the north-star target is quoted on):
RTIOW final random-spheres scene (rt_scene_rtiow(7): ~485 spheres, checker ground, defocus
blur, sky gradient), 1920x1080, 1024 spp, depth 50.  Scaling is STRONG: the same frame is
split over N GPUs.

metric   Msamples/s = W*H*spp / step seconds / 1e6          (whole job, all ranks)
roofline fp32 vector ALU (not HBM, not MFMA: the path is a scalar-per-lane bounce loop).
         achieved = algorithmic flops of one launch / mean launch time (HIP events on the launch
         stream around each timed step's render launches), flops from SURVEY.md 8(d)'s accounting
         with EXACT event counts from the diagnostic counting kernel; peak 157.3 TFLOP/s.
         `roofline` describes the default (culled) kernel as executed; `roofline_linear_scan` runs the
         reference's O(N) hittable_list scan (variant 16) in the same invocation -- the algorithm 8(d)'s
         accounting was written for.
cpu_baseline  the reference's own cmake-cpu-version render() loop (oracle/_ref, "reference") on ALL of
         this host's cores on a bounded sample of the same frame; plus the stock single-thread binary
         and the fp32 restatement as `legs`.
extra    the other BASELINE.json configurations (C1, C2b, C4's scene, one C5 shard), 2 steps each,
         outside the headline timing.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md chip table: 256 CU x 4 SIMD x 32 lanes x 2 x 2.4 GHz
PEAK_HBM_GBPS = 8000.0

# algorithmic flop accounting: add/sub/mul = 1, fma = 2, div/sqrt/rcp = 1,
# compares/selects/min/max/negations/integer RNG = 0
F_SAMPLE = 46          # jitter, camera ray, accumulate
F_TEST = {0: 17, 1: 9, 2: 9, 3: 9, 4: 60, 5: 40}  # per ray-primitive test, by rt_prim_type (5 = triangle)
F_HIT = 30             # hit record of the accepted closest hit
F_SCATTER = [40, 55, 65, 0]  # lambertian, metal, dielectric, diffuse_light
F_MISS = 25            # sky / background evaluation
F_BOX = 12             # slab test of one box: 6 fma (the min/max/compare that follow count 0)
F_CULL_SETUP = 17      # per query: 3 reciprocals, margin (mul + add), 6 shifted origins (add + mul each)
F_GRID_SETUP = 8       # per query: 3 reciprocals, |o|^2 (mul + 2 fma)
F_GRID_ENTER = 27      # per lane that reaches the grid: entry point 3 fma, cell index 3 x (sub + mul), leave distances 3 x (fma + sub + mul), step lengths 3 mul
F_GRID_ENTER_SHEET = 18  # the same for a grid one cell high (walk along x and z only): 2 fma, 2 x (sub + mul), 2 x (fma + sub + mul), 2 mul
F_GRID_STEP = 2        # per cell step: one leave distance += step length, best_t x (1 + 1e-4); min / compare / select = 0
F_RANGE_LOOKUP = 24    # per window box a ray reaches: 6 fma for the clipped segment's end points, 2 axes x (2 margin + 2 offset + 2 scale)


def algorithmic_flops(counts: dict, prim_types) -> float:
    """F = 46 S + sum_prims F_test * Q + 30 H + sum_m F_m B_m + 25 M  (T = queries x N)."""
    per_query = sum(F_TEST[int(t)] for t in prim_types)
    return (F_SAMPLE * counts["samples"] + per_query * counts["queries"] + F_HIT * counts["hits"]
            + sum(f * b for f, b in zip(F_SCATTER, counts["scatter"])) + F_MISS * counts["misses"])


def parse_args(argv=None):
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
    ap.add_argument("--no-linear", action="store_true", help="skip the linear-scan (variant 16) roofline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the other BASELINE configurations")
    ap.add_argument("--cpu-spp", type=int, default=64)
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (16 = linear scan without culling)")
    ap.add_argument("--verify", action="store_true", help="rank 0 also renders the unsharded frame and checks the "
                    "gathered one against it bit for bit (outside the timed region)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) | gloo (rehearsal: ranks may share one GPU)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: every step waits for its own gather (default: a frame's "
                    "gather overlaps the next frame's render)")
    ap.add_argument("--tiles-abi", action="store_true", help="one process: rt_render_hip_tiles (C ABI) over --gpus N devices")
    return ap.parse_args(argv)


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N rank processes (fresh interpreters, so no
    process that has initialised HIP is ever re-executed) and wait for them.  The parent never touches the GPU -- it does
    not even count the devices: every rank checks LOCAL_RANK against what it sees and exits non-zero (main()), which fails
    the whole run ("refusing to run on fewer"; --backend gloo rehearses ranks that share a GPU)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.tiles_abi:
        sys.exit(tiles_abi_main(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

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
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and args.backend == "nccl" and world > 1:
        raise SystemExit(f"bench.py rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible: refusing to run "
                         f"--gpus {args.gpus} on fewer (use --backend gloo to rehearse ranks that share a GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the render path has no CPU fallback")
    dev_index = local_rank % ndev  # identity on a real N-GPU node; gloo rehearsals share GPUs
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():  # (RCCL's version banner)
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            else:
                dist.init_process_group(args.backend, rank=rank, world_size=world)

    scene = rtmi.Scene.rtiow(7, args.width, args.height, args.spp, args.depth)
    chunk = args.chunk if args.chunk >= 0 else 0  # 0 = the library's default work-item size
    base = rtmi.Opts(seed=args.seed, device=dev_index, tile_rows=args.tile_rows, spp_chunk=chunk, variant=args.variant)
    base.tile_rotate = scene.shard_deal(base, world)  # how the frame's row tiles are dealt out to the ranks (include/rtmi.h)
    mine = rdist.shard_opts(base, rank, world)
    # N > 1: two local buffers, so that the gather of one frame travels while the next frame renders (the collective runs on
    # the communicator's own stream; a frame is placed on rank 0 one step after it was rendered, the last one before the
    # closing barrier of whatever region the steps are in).  --no-overlap: one buffer, every step waits for its gather.
    overlap = world > 1 and not args.no_overlap
    serial = [not overlap]  # (set by the priming step if the overlapped form fails on this installation)
    locals_ = [rdist.alloc_local(scene, base, world, device) for _ in range(2 if overlap else 1)]
    local = locals_[0]
    full = torch.empty((scene.height, scene.width, 3), dtype=torch.float32, device=device) if rank == 0 else None
    stream = torch.cuda.current_stream().cuda_stream
    via_host = args.backend != "nccl"
    in_flight = []   # at most one PendingGather
    n_steps = [0]

    def finish_gather():
        img = None
        while in_flight:
            img = rdist.gather_end(in_flight.pop(0), scene, base, rank, world, local.shape[0], device, out=full, via_host=via_host)
        return img

    def step(events=None):
        # HIP events on the stream the kernels are launched on; read after the timed loop (no host wait here)
        buf = locals_[n_steps[0] % len(locals_)]
        n_steps[0] += 1
        if events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        scene.render_device(mine, buf.data_ptr(), stream, None)
        if events is not None:
            e1.record()
            events.append((e0, e1))
        if serial[0]:
            return rdist.gather_framebuffer(buf, scene, base, rank, world, out=full, via_host=via_host)
        started = rdist.gather_begin(buf, rank, world, via_host=via_host, slot=(n_steps[0] - 1) % 2)
        img = finish_gather()  # the previous frame: its buffer is free again once this returns
        in_flight.append(started)
        return img

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # one-time set-up outside any timed or warm-up step: scene image upload, accumulator allocation and
    # occupancy query, and the communicator's first collective.  It is the same launch as a step, so that
    # every render_kernel row of a `rocprofv3 --stats` summary of this command is one step's launch.
    with stdout_to_stderr():
        try:
            step()
            finish_gather()
        except Exception as e:  # (every rank runs the same calls: a refusal of the asynchronous gather is common to all of them)
            if serial[0]:
                raise
            print(f"rank {rank}: overlapped gather failed ({e!r}); one gather per step from here on", file=sys.stderr, flush=True)
            serial[0] = True
            in_flight.clear()
            step()
        if world > 1:
            dist.barrier()
    for _ in range(args.warmup):
        step()
    finish_gather()
    barrier()
    events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img = step(events)
    if not serial[0]:
        img = finish_gather()  # the last frame's gather and placement belong to the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    kernel_ms = [a.elapsed_time(b) for a, b in events]

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
            "sharding": f"row tiles of {args.tile_rows} rows dealt out to {world} rank(s) (rt_opts.tile_rotate = {base.tile_rotate}), "
                        f"one gather to rank 0 per frame ({args.backend}"
                        + (", overlapping the next frame's render)" if not serial[0] else ")"),
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
        k_ms = float(np.mean(kernel_ms))
        result["roofline"] = roofline_of(rtmi, scene, mine, args, chunk, world, k_ms, np)
        if world == 1 and not args.no_linear and not args.no_counts and args.variant == 0:
            result["roofline_linear_scan"] = linear_scan_leg(rtmi, scene, base, local, stream, torch, np)
        if world == 1 and not args.no_extra:
            result["extra"] = extra_configs(rtmi, dev_index, args.seed, local, stream, torch, np)
            try:  # the multi-GPU path the C ABI exports, on this one GPU: what its plumbing costs per frame
                result["extra"]["tiles_abi_1gpu"] = tiles_abi_frames(rtmi, scene, base, [dev_index], 3, np, reference=img.cpu().numpy())[0]
            except Exception as e:  # never lose the bench line over an extra
                result["extra"]["tiles_abi_1gpu"] = {"error": repr(e)}
        # ---- CPU baseline (reported only): the reference's own render() on this host's cores
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(rtmi, args)
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


class stdout_to_stderr:
    """RCCL prints a version banner on the process's stdout at the first communicator: keep stdout to the ONE JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def tiles_abi_frames(rtmi, scene, base, devices, steps, np, reference=None):
    """rt_render_hip_tiles (include/rtmi.h; csrc/tiles.hip) on `devices`: the first call (streams, buffers, ncclCommInitAll)
    apart, then `steps` frames.  Wall time per frame from call to return = render + gather + row placement + copy of the
    frame to the caller's host buffer."""
    o = rtmi.Opts(seed=base.seed, tile_rows=base.tile_rows, spp_chunk=base.spp_chunk, variant=base.variant)
    st = rtmi.Stats()
    t0 = time.perf_counter()
    with stdout_to_stderr():
        img = scene.render_tiles(devices=devices, opts=o, stats=st)
    first = time.perf_counter() - t0
    wall, kern, gath = [], [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        img = scene.render_tiles(devices=devices, opts=o, stats=st, out=img)  # (the caller's buffer, as a renderer's frame loop has one)
        wall.append((time.perf_counter() - t0) * 1e3)
        kern.append(st.kernel_ms), gath.append(st.gather_ms)
    n = scene.width * scene.height * scene.spp
    out = {"entry_point": "rt_render_hip_tiles", "devices": list(devices), "steps": steps,
           "ms_per_frame_call_to_return": round(float(np.mean(wall)), 3),
           "Msamples_per_s": round(n / (float(np.mean(wall)) * 1e-3) / 1e6, 1),
           "slowest_device_render_ms": round(float(np.mean(kern)), 3),
           "gather_placement_ms": round(float(np.mean(gath)), 3),
           "first_call_s": round(first, 3),
           "note": "host buffer out: includes the device-to-host copy of the fp32 frame; first_call_s = streams, buffers, "
                   "ncclCommInitAll and one frame"}
    if reference is not None:
        out["equals_headline_frame"] = bool(np.array_equal(img, reference))
        assert out["equals_headline_frame"], "rt_render_hip_tiles differs from the headline frame"
    return out, img


def tiles_abi_main(args) -> int:
    """`bench.py --tiles-abi --gpus N`: the bench line of the single-process C-ABI path (no torch.distributed)."""
    import numpy as np
    from __graft_entry__ import load_package
    rtmi = load_package()
    try:
        ndev = rtmi.device_count()
    except rtmi.RtmiError:  # no HIP device at all
        ndev = 0
    if ndev < args.gpus:
        print(f"bench.py --tiles-abi: --gpus {args.gpus} but only {ndev} GPU(s) visible: refusing to run on fewer", file=sys.stderr)
        return 2
    scene = rtmi.Scene.rtiow(7, args.width, args.height, args.spp, args.depth)
    chunk = args.chunk if args.chunk >= 0 else 0
    base = rtmi.Opts(seed=args.seed, tile_rows=args.tile_rows, spp_chunk=chunk, variant=args.variant)
    devices = list(range(args.gpus))
    st = rtmi.Stats()
    with stdout_to_stderr():
        img = scene.render_tiles(devices=devices, opts=base, stats=st)  # set-up: scene upload, streams, RCCL communicators
    for _ in range(args.warmup):
        img = scene.render_tiles(devices=devices, opts=base, stats=st)
    kern, gath = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        img = scene.render_tiles(devices=devices, opts=base, stats=st)  # synchronous: returns with the frame in host memory
        kern.append(st.kernel_ms), gath.append(st.gather_ms)
    elapsed = time.perf_counter() - t0
    n = scene.width * scene.height * scene.spp
    result = {
        "metric": "Msamples/sec (WxHxspp)", "value": round(n / (elapsed / args.steps) / 1e6, 3), "unit": "Msamples/s",
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"RTIOW random-spheres scene (rt_scene_rtiow seed 7, {scene.info.num_prims} spheres), "
                        f"{scene.width}x{scene.height}, {scene.spp} spp, depth {scene.max_depth}",
            "sharding": f"rt_render_hip_tiles (C ABI, one process): row tiles of {args.tile_rows} rows interleaved over "
                        f"{args.gpus} device(s), one stream per device, ONE ncclGather to device 0, row placement, one copy "
                        f"to the caller's host buffer (inside the timed region: this entry point hands over host memory)",
            "spp_chunk": chunk, "kernel_variant": args.variant, "render_seed": args.seed,
            "slowest_device_render_ms": round(float(np.mean(kern)), 3), "gather_placement_ms": round(float(np.mean(gath)), 3),
        },
    }
    if args.verify:
        whole = scene.render(rtmi.Opts(seed=args.seed, spp_chunk=chunk, variant=args.variant))
        same = bool(np.array_equal(whole, img))
        result["config"]["gathered_equals_unsharded"] = same
        assert same, "rt_render_hip_tiles differs from the unsharded frame"
    mean = float(img.mean()) / scene.spp
    result["config"]["frame_mean_radiance"] = round(mean, 5)
    assert np.isfinite(mean) and 0.2 < mean < 0.8, mean
    print(json.dumps(result), flush=True)
    return 0


def _git_head():
    """HEAD of this checkout; on a GPU box (a snapshot without .git) the value build() recorded in BUILD_HEAD."""
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              timeout=10).stdout.strip()
        if head:
            return head
    except Exception:
        pass
    try:
        with open(os.path.join(ROOT, "ray-tracing-in-cuda_amd", "BUILD_HEAD")) as f:
            return f.read().strip() or None
    except OSError:
        return None


def roofline_of(rtmi, scene, mine, args, chunk, world, k_ms, np):
    """fp32-ALU roofline of the dominant kernel (rank 0's launch)."""
    roof = {"bound": "valu_fp32", "achieved": None, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": None,
            "traffic": None,
            "bound_note": "fp32 vector ALU (BASELINE.json metric and SURVEY 8(d): the path is not a contraction, "
                          "so neither the MFMA nor the HBM roofline binds; HBM share below as hbm_frac)",
            "kernel_ms_avg": round(k_ms, 3),
            "kernel_ms_note": "HIP events on the launch stream around each timed step's launches (accumulator clear, "
                              "render_kernel, finalize)"}
    if not args.no_counts:
        st = scene.count(mine)  # exact event counts of rank 0's shard (same seed, same samples; culled kernel)
        c = st.as_dict()
        types = scene.prims()["type"]
        linear_flops = algorithmic_flops(c, types)
        per_query_linear = sum(F_TEST[int(t)] for t in types)
        n_other = sum(F_TEST[int(t)] for t in types if int(t) != 0)  # rects / cylinders / triangles: tested per query
        culled = args.variant not in (16, 17, 24)
        wq = max(1, c["wave_queries"])
        lanes = c["queries"] / wq  # live lanes per wave-level query
        if culled:
            sphere_tests, box_tests, setup = executed_tests(c, lanes)
        else:
            sphere_tests = c["queries"] * (c["cull_prefix"] + c["cull_clusters"] * c["cull_cluster_size"])
            box_tests, setup = 0, 0.0
        shading = linear_flops - per_query_linear * c["queries"]  # 46 S + 30 H + scatter + 25 M
        flops_strict = 17 * sphere_tests + n_other * c["queries"] + shading            # box tests = accelerator overhead
        flops = flops_strict + F_BOX * box_tests + setup
        search = {5: ("uniform grid one cell high, per-lane x-z DDA over two-tier cell lists (default kernel; counted by the 3-D "
                      "walk's diagnostic kernel: same cells, same tests)") if c.get("grid_sheet") and args.variant in (0, 2)
                     else "uniform grid, per-lane 3-D DDA over two-tier cell lists (default kernel)",
                  7: "uniform grid over every primitive type, per-lane 3-D DDA over the wide cell lists",
                  3: "candidate clusters from range tables, per-lane cluster lists",
                  2: "per-lane cluster lists through the two-level box hierarchy"}.get(c.get("cull_mode"), "clusters")
        roof["mode"] = f"culled hittable_list: {search}" if culled else "linear hittable_list scan (variant 16)"
        if args.variant == 32:  # the diagnostic kernel is the default one: its counts do not describe this one
            roof["mode"] = "ablation variant with wave-level cluster votes: no flop count"
            flops = flops_strict = 0
        if flops:
            roof["achieved"] = round(flops / (k_ms * 1e-3) / 1e12, 3)
            roof["frac"] = round(roof["achieved"] / PEAK_FP32_TFLOPS, 4)
            roof["frac_excluding_box_tests"] = round(flops_strict / (k_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, 4)
        roof["flops_per_launch"] = int(flops)
        roof["accounting"] = ("SURVEY 8(d): flops = 46 S + 17 T_sphere + 12 T_box (6 fma; min/max/compare = 0) + per-query "
                              "culling set-up (grid: 8 per query + 27 per lane that enters + 2 per cell step) "
                              "+ 30 H + 40/55/65 B + 25 M, with the sphere and box tests the kernel's lanes "
                              "EXECUTE (masked-off lanes not counted), all counted exactly by the diagnostic kernel; "
                              "frac_excluding_box_tests counts the accelerator's own work as 0")
        roof["frac_note"] = ("the fraction counts the flops of the tests the kernel EXECUTES; each round's candidate search executes "
                             "fewer of them per frame (round 1 box hierarchy 5.2e12, round 2 range tables 2.9e12, the grid walk "
                             "fewer again) in less time, so the fraction falls while Msamples/s rises; roofline_linear_scan is the "
                             "same kernel made to execute the reference's O(N) scan")
        roof["counts"] = {k: c[k] for k in ("samples", "queries", "prim_tests", "hits", "misses", "scatter",
                                            "rng_draws", "wave_queries", "clusters_visited", "groups_visited",
                                            "lane_clusters", "lane_groups", "lane_cands") if k in c}
        roof["tests_per_sample"] = {"sphere": round(sphere_tests / max(1, c["samples"]), 1),
                                    "box": round(box_tests / max(1, c["samples"]), 1),
                                    "reference_linear_scan": round(c["prim_tests"] / max(1, c["samples"]), 1)}
        roof["lane_occupancy_of_queries"] = round(c["queries"] / max(1, 64 * wq), 4)
        roof["valu_issue_peak_Tlane_inst"] = 78.6
    # HBM: SURVEY 8(d)'s algorithmic bytes are ONE framebuffer of W x H x 16 B; the kernel's own minimum adds the
    # 64-bit accumulator plane (clear, one flush per work item, read-back)
    rows0 = scene.shard_rows(mine)
    plane = rows0 * scene.width * 3
    n_chunks = -(-scene.spp // (chunk or 128))  # 0 = the library default, 128 samples per work item
    algo_bytes = plane * (8 + 8 * n_chunks + 8 + 4)
    roof["hbm_survey_bytes"] = int(rows0 * scene.width * 16)
    roof["hbm_algorithmic_bytes"] = int(algo_bytes)
    roof["hbm_achieved_GBps"] = round(algo_bytes / (k_ms * 1e-3) / 1e9, 3)
    roof["hbm_frac"] = round(roof["hbm_achieved_GBps"] / PEAK_HBM_GBPS, 6)
    # measured HBM traffic: rocprofv3 PMC passes (tools/profile.sh) of THIS command line, kept under profiles/;
    # static (not collected in this run), so only quoted when workload, variant, chunk and world match the profiled run
    traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(traffic_file) and world == 1 and args.variant == 0 and chunk == 0:
        try:
            tf = json.load(open(traffic_file))
            key = f"{scene.width}x{scene.height}x{scene.spp}"
            if key in tf:
                roof["traffic"] = tf[key]["bytes_per_launch"]
                roof["traffic_ratio_vs_survey_bytes"] = round(tf[key]["bytes_per_launch"] / roof["hbm_survey_bytes"], 2)
                if tf[key].get("valu_insts") and roof.get("tests_per_sample"):
                    # share of the issued vector lane slots that are SURVEY 8(d) sphere tests (12 instructions each): what the
                    # 8(d) fraction hides -- tracked per round next to the lane utilisation
                    tests = roof["tests_per_sample"]["sphere"] * roof["counts"]["samples"]
                    roof["useful_valu_share"] = round(12.0 * tests / (64.0 * tf[key]["valu_insts"]), 4)
                    roof["valu_lane_utilization"] = tf[key].get("valu_lane_utilization")
                roof["traffic_source"] = {"kind": "static: PMC passes of this command, not re-collected in this run",
                                          "file": "profiles/hbm_traffic.json", "detail": tf[key].get("source"),
                                          "profiled_at_head": tf[key].get("head"), "this_head": _git_head()}
        except Exception:
            pass
    return roof


def executed_tests(c: dict, lanes: float):
    """Ray-primitive and box tests the default kernel's lanes execute, from the diagnostic counters.
    Returns (sphere tests, box tests, per-query set-up flops)."""
    q = c["queries"]
    if c.get("cull_mode", 2) in (5, 7):
        # uniform grid: every live lane tests the always-tested prefix and clips its ray against the grid's bounds (one box
        # test); lanes that reach it compute their entry cell and walk: one sphere test per list entry of every cell
        # visited (lane_clusters counts single spheres here), one step per further cell
        return (q * c["cull_prefix"] + c["lane_clusters"], q,
                F_GRID_SETUP * q + (F_GRID_ENTER_SHEET if c.get("grid_sheet") else F_GRID_ENTER) * c["lane_groups"]
                + F_GRID_STEP * c["lane_cands"])
    if c.get("cull_mode", 2) == 3:
        # range tables: every live lane tests the always-tested prefix and clips its ray against every window box;
        # where it reaches one it looks its candidate clusters up (segment end points + slab indices: 24 flops),
        # tests the box of every candidate and the spheres of the clusters ITS OWN ray reaches
        sphere_tests = q * c["cull_prefix"] + c["lane_clusters"] * c["cull_cluster_size"]
        box_tests = q * c["cull_windows"] + c["lane_cands"]
        return sphere_tests, box_tests, F_CULL_SETUP * q + F_RANGE_LOOKUP * c["lane_groups"]
    # box hierarchy: every live lane tests the always-tested prefix, every outer box, the cluster boxes of the outer
    # boxes some lane of its wave passed, and the spheres of the clusters ITS OWN ray reaches
    sphere_tests = q * c["cull_prefix"] + c["lane_clusters"] * c["cull_cluster_size"]
    box_tests = q * c["cull_groups"] + lanes * c["groups_visited"] * 4
    return sphere_tests, box_tests, F_CULL_SETUP * q


def _timed_launches(scene, opts, local, stream, torch, steps):
    """mean launch time (ms) of `steps` renders after one untimed one, HIP events on the launch stream"""
    scene.render_device(opts, local.data_ptr(), stream, None)
    ev = []
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        scene.render_device(opts, local.data_ptr(), stream, None)
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / len(ev)


def linear_scan_leg(rtmi, scene, base, local, stream, torch, np, steps=2):
    """SURVEY 8(d)'s literal accounting on the algorithm it was written for: the reference's linear
    hittable_list scan (rt_opts.variant 16: every sphere tested for every query), same frame, same seed,
    bit-identical framebuffer."""
    o = rtmi.Opts(seed=base.seed, device=base.device, tile_rows=base.tile_rows, spp_chunk=base.spp_chunk, variant=16)
    ms = _timed_launches(scene, o, local, stream, torch, steps)
    c = scene.count(rtmi.Opts(seed=base.seed, device=base.device, tile_rows=base.tile_rows)).as_dict()
    flops = algorithmic_flops(c, scene.prims()["type"])  # T = queries x N, nothing else
    tf = flops / (ms * 1e-3) / 1e12
    return {"kernel_variant": 16, "steps": steps, "kernel_ms": round(ms, 3),
            "Msamples_per_s": round(scene.width * scene.height * scene.spp / (ms * 1e-3) / 1e6, 1),
            "flops_per_launch": int(flops), "achieved": round(tf, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / PEAK_FP32_TFLOPS, 4),
            "accounting": "46 S + 17 (queries x N spheres) + 30 H + 40/55/65 B + 25 M; the kernel also executes the "
                          "table's padding slots, which are not counted"}


def extra_configs(rtmi, dev, seed, scratch, stream, torch, np, steps=2):
    """The other BASELINE.json configurations on this GPU, outside the headline timing: mean launch time of
    `steps` launches each (HIP events), whole frame unless stated."""
    out = {}
    scenes_dir = os.path.join(ROOT, "ray-tracing-in-cuda_amd", "scenes")
    golden = os.path.join(ROOT, "tests", "golden", "scenes")

    def run(name, sc, opts, what):
        rows = sc.shard_rows(opts)
        buf = torch.empty((rows, sc.width, 3), dtype=torch.float32, device=scratch.device)
        ms = _timed_launches(sc, opts, buf, stream, torch, steps)
        n = rows * sc.width * sc.spp
        out[name] = {"workload": what, "kernel_ms": round(ms, 3), "Msamples_per_s": round(n / (ms * 1e-3) / 1e6, 1),
                     "samples": n, "steps": steps}

    c1 = rtmi.Scene.load(os.path.join(scenes_dir, "three_sphere.json"))
    c1.override(width=400, height=225, spp=100, max_depth=50)
    run("config1_three_sphere", c1, rtmi.Opts(seed=seed, device=dev), "3-sphere scene 400x225, 100 spp, depth 50")
    c2 = rtmi.Scene.dna(0.0)
    c2.override(width=1280, height=720, spp=256, max_depth=50)
    run("config2b_dna", c2, rtmi.Opts(seed=seed, device=dev),
        "basic_scene.json filled with dna.py frame 0 (60 emissive spheres + 30 emissive cylinders), 1280x720, 256 spp")
    c4 = rtmi.Scene.load(os.path.join(golden, "sample_scene.json"))
    c4.override(width=1920, height=1080, spp=512, max_depth=50)
    run("config4_sample_scene", c4, rtmi.Opts(seed=seed, device=dev),
        "sample_scene.json 1920x1080 at 512 of its 4096 spp (cost per sample does not depend on spp), whole frame")
    c5 = rtmi.Scene.rtiow(7, 3840, 2160, 8192, 50)
    run("config5_one_of_8_shards", c5, rtmi.Opts(seed=seed, device=dev, tile_first=0, tile_stride=8),
        "RTIOW 3840x2160, 8192 spp: rank 0's row tiles of the 8-GPU split (1/8 of the frame) on this GPU")
    return out


def cgroup_cpu_quota():
    """CPU quota of this container in cores (cgroup v2 cpu.max, v1 cfs quota), None when unlimited."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def cpu_baseline(rtmi, args):
    """Reported-only baseline, SURVEY 8(d): (ii) the reference's render() on all host cores [the headline `value`],
    (i) the stock single-thread binary, (iii) the fp32 restatement -- each on a bounded sample."""
    import ctypes as C
    import rtcheck
    affinity = len(os.sched_getaffinity(0))
    nproc = os.cpu_count()
    quota = cgroup_cpu_quota()
    # threads = the cores this process may actually use: its affinity mask, capped by the container's CPU quota
    # (a box may show every core of the host and still be throttled to a share of them)
    cores = affinity if quota is None else max(1, min(affinity, int(quota + 0.999)))
    spp = args.cpu_spp
    period, band = 32, 8
    sc = rtmi.Scene.rtiow(7, args.width, args.height, spp, args.depth)
    legs = {}
    # bounded sample: every 4th band of 8 rows of the full-size frame at reduced spp (cost per sample does not
    # depend on spp).  Systematic in y, so sky, spheres and ground are represented in the frame's proportions.
    what = None
    if rtcheck.have_ref():
        lib = rtcheck.ref_lib()
        lib.ref_time_sample.restype = C.c_double
        lib.ref_time_sample.argtypes = [C.c_void_p, C.c_uint64] + [C.c_int] * 9 + [C.POINTER(C.c_double),
                                                                                  C.POINTER(C.c_longlong)]
        rs = rtcheck.RefScene(sc)
        chk, px = C.c_double(), C.c_longlong()
        # sampling error: the same estimator on the three other phases of the systematic sample at 1/8 of the spp
        cpu0 = time.process_time()
        sec = lib.ref_time_sample(rs.h, args.seed, args.width, args.height, 0, args.height, period, band, spp,
                                  args.depth, cores, C.byref(chk), C.byref(px))
        busy = (time.process_time() - cpu0) / sec  # CPU seconds per wall second = cores the leg really had
        n_samples = px.value * spp
        rate = n_samples / sec / 1e6
        phase_rates = []
        for ph in (8, 16, 24):
            s2 = lib.ref_time_sample(rs.h, args.seed, args.width, args.height, ph, args.height, period, band,
                                     max(1, spp // 8), args.depth, cores, C.byref(chk), C.byref(px))
            phase_rates.append(px.value * max(1, spp // 8) / s2 / 1e6)
        spread = (max(phase_rates + [rate]) - min(phase_rates + [rate])) / rate
        kind = "reference"
        what = ("cmake-cpu-version render() per pixel (oracle/_ref: the reference's own sources + hooked rand()), "
                f"OpenMP over (row, 64-pixel span) on {cores} threads")
        legs["ii_reference_all_cores"] = {"Msamples_per_s": round(rate, 4), "threads": cores, "seconds": round(sec, 2),
                                         "cores_busy": round(busy, 1),
                                         "samples": n_samples,
                                         "sampling_spread": round(spread, 4),
                                         "sampling_note": "max-min over the four phases of the every-4th-band sample, "
                                                          "relative (the other three phases at 1/8 of the spp)"}
        stock = os.path.join(ROOT, "oracle", "_ref", "ref_stock")
        if os.path.exists(stock):
            try:
                w1, h1 = 960, 540
                with tempfile.TemporaryDirectory() as td:
                    subprocess.run([stock, "-w", str(w1), "-h", str(h1), "-spp", "1", "-d", str(args.depth)], cwd=td,
                                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
                    log = open(os.path.join(td, "gpu-version-time.log")).read()
                m = re.search(r"time:\s*([0-9.eE+-]+)\s*s", log)
                t1 = float(m.group(1))
                legs["i_stock_single_thread"] = {
                    "Msamples_per_s_per_core": round(w1 * h1 / t1 / 1e6, 4), "seconds": round(t1, 2),
                    "what": f"cmake-cpu-version/main.cpp as shipped (-DMT_RANDOM_GENERATOR, its own random_scene and "
                            f"mt19937 stream, one thread, whole program incl. PPM text output), {w1}x{h1}, 1 spp, "
                            f"time from its own log line"}
            except Exception as e:  # the leg is optional: never lose the bench line over it
                legs["i_stock_single_thread"] = {"error": repr(e)}
    else:
        kind = "port"
    # (iii) the restatement (oracle/rt_oracle.c), same sample
    try:
        olib = rtcheck.oracle_lib()
        olib.rto_time_sample.restype = C.c_double
        olib.rto_time_sample.argtypes = [C.c_void_p, C.c_uint64] + [C.c_int] * 6 + [C.POINTER(C.c_double),
                                                                                    C.POINTER(C.c_longlong)]
        osc = rtcheck.OracleScene(sc)
        chk, px = C.c_double(), C.c_longlong()
        sec3 = olib.rto_time_sample(C.cast(C.byref(osc.c), C.c_void_p), args.seed, 0, args.height, period, band, spp,
                                    cores, C.byref(chk), C.byref(px))
        legs["iii_restatement_all_cores"] = {"Msamples_per_s": round(px.value * spp / sec3 / 1e6, 4), "threads": cores,
                                             "seconds": round(sec3, 2),
                                             "what": "fp32 restatement oracle/rt_oracle.c, same sample"}
        if kind == "port":
            rate, sec, n_samples = px.value * spp / sec3 / 1e6, sec3, px.value * spp
            what = f"fp32 CPU restatement (oracle/rt_oracle.c), OpenMP on {cores} threads"
    except Exception as e:
        legs["iii_restatement_all_cores"] = {"error": repr(e)}
    return {
        "value": round(rate, 4),
        "unit": "Msamples/s",
        "cores": cores,
        "nproc": nproc,
        "affinity_cores": affinity,
        "cgroup_cpu_quota": quota,
        "kind": kind,
        "sample": f"{what}; same scene at {args.width}x{args.height}, every 4th 8-row band, {spp} spp "
                  f"({n_samples} samples, {sec:.1f} s)",
        "legs": legs,
    }


if __name__ == "__main__":
    main()
