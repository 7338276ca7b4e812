"""CPU-side checks of bench.py: the rank launcher's refusal path, the flop accounting constants and the
CPU-baseline leg (the only part of bench.py that may use oracle/)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_gpus_n_without_launcher_refuses_fewer_devices():
    """`python bench.py --gpus N` starts N ranks itself; with fewer than N GPUs it must fail, not fall back to one."""
    import torch
    n = torch.cuda.device_count() + 1
    if n < 2:
        n = 2
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], capture_output=True, text=True,
                       env=env, timeout=300)
    assert p.returncode != 0
    assert "GPU(s) visible" in p.stderr and "refusing" in p.stderr
    assert p.stdout.strip() == ""  # no JSON line that could be mistaken for a measurement


def test_tiles_abi_mode_refuses_fewer_devices():
    """`bench.py --tiles-abi --gpus N` (one process, rt_render_hip_tiles) on fewer than N devices: an error, no JSON line."""
    import torch
    n = max(2, torch.cuda.device_count() + 1)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tiles-abi", "--gpus", str(n)], capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "refusing" in p.stderr and p.stdout.strip() == ""


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_flop_accounting_follows_survey_8d():
    b = _bench()
    # compares / selects / min / max count 0: the slab test is its 6 fma
    assert b.F_BOX == 12 and b.F_TEST[0] == 17 and b.F_SAMPLE == 46 and b.F_HIT == 30 and b.F_MISS == 25
    counts = {"samples": 10, "queries": 27, "hits": 17, "misses": 10, "scatter": [9, 4, 3, 1]}
    prim_types = [0] * 5 + [1, 4]
    want = 46 * 10 + (5 * 17 + 9 + 60) * 27 + 30 * 17 + 40 * 9 + 55 * 4 + 65 * 3 + 25 * 10
    assert b.algorithmic_flops(counts, prim_types) == want
    # cluster-box kernel: prefix + own clusters' spheres; every outer box + the cluster boxes of visited groups
    c = {"queries": 640, "cull_prefix": 4, "lane_clusters": 1000, "cull_cluster_size": 16, "cull_groups": 8,
         "groups_visited": 30}
    s, bx, setup = b.executed_tests(c, lanes=64.0)
    assert s == 640 * 4 + 1000 * 16 and bx == 640 * 8 + 64 * 30 * 4 and setup == b.F_CULL_SETUP * 640
    # uniform grid (the default): one bounds clip per query, one sphere test per list entry of the cells a lane visits
    c = {"queries": 640, "cull_prefix": 4, "lane_clusters": 1800, "cull_cluster_size": 8, "cull_mode": 5, "cull_windows": 1,
         "lane_cands": 450, "lane_groups": 470}
    s, bx, setup = b.executed_tests(c, lanes=64.0)
    assert s == 640 * 4 + 1800 and bx == 640 and setup == b.F_GRID_SETUP * 640 + b.F_GRID_ENTER * 470 + b.F_GRID_STEP * 450
    # range tables: one window-box clip per query and window, one box test per candidate cluster
    c = {"queries": 640, "cull_prefix": 4, "lane_clusters": 1200, "cull_cluster_size": 8, "cull_mode": 3, "cull_windows": 1,
         "lane_cands": 2000, "lane_groups": 470}
    s, bx, setup = b.executed_tests(c, lanes=64.0)
    assert s == 640 * 4 + 1200 * 8 and bx == 640 + 2000 and setup == b.F_CULL_SETUP * 640 + b.F_RANGE_LOOKUP * 470


def test_cpu_baseline_leg_runs_on_all_cores(rtmi):
    b = _bench()
    args = b.parse_args(["--width", "96", "--height", "64", "--cpu-spp", "2", "--depth", "50"])
    r = b.cpu_baseline(rtmi, args)
    assert r["affinity_cores"] == len(os.sched_getaffinity(0)) and r["nproc"] == os.cpu_count()
    q = b.cgroup_cpu_quota()
    assert r["cores"] == (r["affinity_cores"] if q is None else max(1, min(r["affinity_cores"], int(q + 0.999))))
    assert r["value"] > 0 and r["unit"] == "Msamples/s" and r["kind"] in ("reference", "port")
    legs = r["legs"]
    assert legs["iii_restatement_all_cores"]["Msamples_per_s"] > 0
    if r["kind"] == "reference":
        assert legs["ii_reference_all_cores"]["samples"] == 96 * 16 * 2  # rows 0-7 and 32-39 of 64
        assert "sampling_spread" in legs["ii_reference_all_cores"]
        if "error" not in legs.get("i_stock_single_thread", {"error": 1}):
            assert legs["i_stock_single_thread"]["Msamples_per_s_per_core"] > 0
