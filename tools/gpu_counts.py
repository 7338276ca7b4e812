import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
rtmi = load_package()
sc = rtmi.Scene.rtiow(7, 1920, 1080, 16, 50)
st = sc.count(rtmi.Opts(seed=2023))
d = st.as_dict(); q = d["queries"]
print(d)
print("per query: lane candidates %.2f, wave entries per wave-query %.1f (of %d tests), lanes per entry %.2f" % (
    d["cand_lanes"] / q, d["cand_waves"] / (q / 64), sc.info.num_prims, d["cand_lanes"] / d["cand_waves"]))
