"""Turn a tools/profile.sh output directory into the committed summary under profiles/."""
import collections, csv, glob, json, os, subprocess, sys
src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
try:
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    head = None
out = {"tag": tag, "source": "rocprofv3 (ROCm 7.2) on MI355X via tools/profile.sh", "head": head, "kernel_stats": [], "pmc": {}}
for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        out["kernel_stats"].append({k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        # the headline kernel: render_kernel<COUNT = false, POOL, SCALAR = false, CULL = 6 (x-z walk) or 5, EXT = false, SPH = true>
        if "render_kernel<false, true, false, 6, false, true>" in r["Kernel_Name"] or "render_kernel<false, true, false, 5, false, true>" in r["Kernel_Name"]:
            out["render_dispatch"] = {k: r[k] for k in ("Kernel_Name", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Workgroup_Size_X", "Grid_Size_X")}
            break
agg = collections.defaultdict(float); launches = collections.defaultdict(int)
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "render_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); launches[r["Counter_Name"]] += 1
out["pmc_workload"] = "bench.py --steps 1 --warmup 0 (the bench workload: RTIOW 1920x1080x1024, one render_kernel launch per pass)"
out["pmc"] = {k: agg[k] / max(1, launches[k]) for k in sorted(agg)}
p = out["pmc"]
d = {}
if "SQ_THREAD_CYCLES_VALU" in p and "SQ_ACTIVE_INST_VALU" in p:
    d["valu_lane_utilization"] = p["SQ_THREAD_CYCLES_VALU"] / (p["SQ_ACTIVE_INST_VALU"] * 64)
if "GRBM_GUI_ACTIVE" in p and "SQ_INSTS_VALU" in p:
    cyc = p["GRBM_GUI_ACTIVE"] / 8  # summed over 8 XCDs
    d["kernel_cycles"] = cyc
    d["valu_busy"] = p["SQ_INSTS_VALU"] / 1024 * 2 / cyc   # wave64 VALU = 2 cycles on a SIMD-32, 1024 SIMDs
    if "SQ_WAVE_CYCLES" in p: d["avg_waves_per_simd"] = p["SQ_WAVE_CYCLES"] * 4 / (1024 * cyc)
if "SQ_LDS_BANK_CONFLICT" in p and "SQ_LDS_IDX_ACTIVE" in p:
    d["lds_bank_conflict_ratio"] = p["SQ_LDS_BANK_CONFLICT"] / p["SQ_LDS_IDX_ACTIVE"]
if "SQ_WAVE_CYCLES" in p:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in p: d[k.lower() + "_frac_of_wave_cycles"] = p[k] / p["SQ_WAVE_CYCLES"]
# useful VALU share: the 17-flop sphere tests of SURVEY 8(d), 12 vector instructions per lane each, over
# every lane slot the kernel issued (64 x SQ_INSTS_VALU) -- the tests per launch come from the bench line of the trace pass
try:
    line = [l for l in open(os.path.join(src, "trace.log")) if l.startswith('{"metric"')][-1]
    roof = json.loads(line)["roofline"]
    tests = roof["tests_per_sample"]["sphere"] * roof["counts"]["samples"]
    if "SQ_INSTS_VALU" in p:
        d["sphere_tests_per_launch"] = tests
        d["useful_valu_share"] = 12.0 * tests / (64.0 * p["SQ_INSTS_VALU"])
        d["valu_insts_per_launch"] = p["SQ_INSTS_VALU"]
except Exception as e:
    d["useful_valu_share_error"] = repr(e)
if "FETCH_SIZE" in p: d["hbm_read_bytes_per_launch"] = p["FETCH_SIZE"] * 1024 * 2  # gfx950: FETCH_SIZE reads 1/2 (MI355X_MICROARCH.md HBM)
if "WRITE_SIZE" in p: d["hbm_write_bytes_per_launch"] = p["WRITE_SIZE"] * 1024
out["derived"] = d
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_rocprof_summary.json"), "w"), indent=1)
if "hbm_read_bytes_per_launch" in d and "hbm_write_bytes_per_launch" in d:
    tf = {"1920x1080x1024": {"bytes_per_launch": int(d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]),
                             "read_bytes": int(d["hbm_read_bytes_per_launch"]), "write_bytes": int(d["hbm_write_bytes_per_launch"]),
                             "head": head,
                             "valu_insts": d.get("valu_insts_per_launch"), "valu_lane_utilization": d.get("valu_lane_utilization"),
                             "useful_valu_share": d.get("useful_valu_share"),
                             "source": f"profiles/{tag}_rocprof_summary.json: rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) and WRITE_SIZE, separate passes, render_kernel only"}}
    json.dump(tf, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out["derived"], indent=1)); print(out.get("render_dispatch")); 
for k in out["kernel_stats"][:4]: print(k["Name"][:70], k["Calls"], k["AverageNs"])
