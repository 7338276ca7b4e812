"""Ways of dealing row tiles out to N ranks, compared on measured per-tile costs (gpurun_out/tilecost.json from
tools/gpu_tilecost.py): share of the work of the busiest rank relative to 1/N.  No GPU needed."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "gpurun_out", "tilecost.json")))
q = d["queries"]; T = len(q); tot = sum(q)
def report(name, owner, N):
    w = [0] * N
    for t in range(T):
        w[owner(t)] += q[t]
    print(f"  {name:28s} busiest rank {max(w) / tot * N * 100 - 100:+.2f} %, idlest {min(w) / tot * N * 100 - 100:+.2f} %")
for N in (2, 4, 8):
    print(f"N = {N}")
    report("interleave t mod N", lambda t: t % N, N)
    report("rotated (t + t/N) mod N", lambda t: (t + t // N) % N, N)
    report("snake (period 2N)", lambda t: (t % (2 * N)) if (t % (2 * N)) < N else 2 * N - 1 - (t % (2 * N)), N)
    def snake_rot(t):
        p = t % (2 * N); r = p if p < N else 2 * N - 1 - p
        return (r + t // (2 * N)) % N
    report("snake + rotation per period", snake_rot, N)
    # cost-aware bound: longest processing time first
    order = sorted(range(T), key=lambda t: -q[t]); w = [0] * N
    for t in order:
        w[w.index(min(w))] += q[t]
    print(f"  {'(cost-aware LPT bound)':28s} busiest rank {max(w) / tot * N * 100 - 100:+.2f} %")
print("cost profile (queries per tile / mean):", " ".join(f"{v / tot * T:.2f}" for v in q))
