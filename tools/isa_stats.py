#!/usr/bin/env python3
"""Static picture of ONE render_kernel instance (no GPU needed): registers, scratch, instruction mix.
usage: tools/isa_stats.py [--inst "false,true,false,6,false,true"] [--waves 7] [--keep out.s] [-- extra hipcc flags]
The default instance is the headline kernel (sphere-only x-z grid walk, variant 0 -> 2 on RTIOW)."""
import argparse, collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ray-tracing-in-cuda_amd", "csrc", "render_kernel.hip")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inst", default="false,true,false,6,false,true")
    ap.add_argument("--waves", type=int, default=7)
    ap.add_argument("--keep", default=None)
    ap.add_argument("extra", nargs="*")
    a = ap.parse_args()
    out = a.keep or os.path.join(tempfile.gettempdir(), "rtmi_isa_%d.s" % os.getpid())
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           "-fno-slp-vectorize", "-DRT_WAVES_PER_SIMD=%d" % a.waves, "-DRT_GROUP=4", "-DRT_ISA_ONLY=" + a.inst,
           "--offload-device-only", "-S", "-o", out, SRC] + a.extra
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    body = text[text.index("render_kernel"):]
    meta = {}
    for key in ("next_free_vgpr", "next_free_sgpr", "private_segment_fixed_size", "group_segment_fixed_size", "accum_offset"):
        m = re.search(r"amdhsa_%s (\d+)" % key, text)
        meta[key] = int(m.group(1)) if m else None
    for key in ("sgpr_spill_count", "vgpr_spill_count"):
        m = re.search(r"\.%s:\s+(\d+)" % key, text)
        meta[key] = int(m.group(1)) if m else None
    mix = collections.Counter()
    names = collections.Counter()
    for line in body.splitlines():
        m = re.match(r"\s+([a-z_0-9]+)\s", line + " ")
        if not m:
            continue
        op = m.group(1)
        if op.startswith("v_"):
            mix["valu"] += 1
        elif op.startswith("s_cbranch") or op == "s_branch":
            mix["branch"] += 1
        elif op.startswith("s_waitcnt") or op == "s_nop":
            mix["wait/nop"] += 1
        elif op.startswith("s_"):
            mix["salu"] += 1
        elif op.startswith("ds_"):
            mix["lds"] += 1
        elif op.startswith("scratch_"):
            mix["scratch"] += 1
        elif op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"):
            mix["vmem"] += 1
        else:
            continue
        names[op] += 1
    print("instance <%s> at %d waves/SIMD" % (a.inst, a.waves))
    print("  registers:", meta)
    print("  static instruction mix:", dict(mix), "total", sum(mix.values()))
    hot = ["v_readlane_b32", "v_writelane_b32", "v_mov_b32_e32", "v_cndmask_b32_e32", "v_cndmask_b32_e64", "scratch_load_dword",
           "scratch_store_dword", "ds_bpermute_b32", "v_mul_hi_u32", "v_mul_lo_u32", "v_sqrt_f32_e32", "v_rcp_f32_e32"]
    print("  selected:", {k: names[k] for k in hot if names[k]})
    if not a.keep:
        os.unlink(out)


if __name__ == "__main__":
    main()
