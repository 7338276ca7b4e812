"""Russian roulette (朴素光线追踪/4_0_path_tracing.py:43-46, 88: p_RR = 0.9 survival per bounce,
throughput / p_RR after every scatter).  A non-parity fast mode: the estimate stays unbiased, the draws differ."""
import json
import os

import numpy as np
import pytest

SEED = 2023


def _three(rtmi, scenes_dir, w=40, h=24, spp=300):
    sc = rtmi.Scene.load(os.path.join(scenes_dir, "three_sphere.json"))
    sc.override(width=w, height=h, spp=spp, max_depth=50)
    return sc


def test_scene_schema_and_setter(rtmi, scenes_dir):
    sc = _three(rtmi, scenes_dir)
    assert sc.info.russian_roulette == 0.0
    assert "russian_roulette" not in json.loads(sc.to_json())
    sc.set_russian_roulette(0.9)
    assert sc.info.russian_roulette == np.float32(0.9)
    j = json.loads(sc.to_json())
    assert abs(j["russian_roulette"] - 0.9) < 1e-7
    again = rtmi.Scene.parse(json.dumps(j))
    assert again.info.russian_roulette == np.float32(0.9)
    assert sc.clone().info.russian_roulette == np.float32(0.9)
    for bad in (-0.1, 1.5, float("nan")):
        with pytest.raises(rtmi.RtmiError):
            sc.set_russian_roulette(bad)
    j["russian_roulette"] = 2
    with pytest.raises(rtmi.RtmiError, match="probability"):
        rtmi.Scene.parse(json.dumps(j))
    j["russian_roulette"] = "often"
    with pytest.raises(rtmi.RtmiError):
        rtmi.Scene.parse(json.dumps(j))


def test_checker_estimate_is_unbiased_and_shorter(rtmi, rtcheck, scenes_dir):
    """Same scene with and without roulette in the CPU checker: the image means agree within the
    Monte-Carlo error, fewer closest-hit queries are traced, and p = 1 only adds the survival draws."""
    sc = _three(rtmi, scenes_dir)
    plain, c0 = rtcheck.oracle_render(sc, seed=SEED, want_counts=True)
    sc.set_russian_roulette(0.9)
    rr, c1 = rtcheck.oracle_render(sc, seed=SEED, want_counts=True)
    n = sc.spp
    m0, m1 = plain.astype(np.float64).mean() / n, rr.astype(np.float64).mean() / n
    assert abs(m0 - m1) < 0.01 * m0, (m0, m1)          # 288 k samples: sigma of the mean ~ 0.1 %
    assert c1["queries"] < 0.93 * c0["queries"]              # paths are shorter
    assert not np.array_equal(plain, rr)                # ... and the draws differ
    sc.set_russian_roulette(1.0)                        # always survives: same paths? no -- one more draw each
    one, c2 = rtcheck.oracle_render(sc, seed=SEED, want_counts=True)
    assert c2["rng_draws"] > c0["rng_draws"]
    assert abs(one.astype(np.float64).mean() / n - m0) < 0.01 * m0


@pytest.mark.gpu
@pytest.mark.parametrize("name,p", [("three_sphere", 0.9), ("rtiow", 0.9), ("rtiow", 0.5), ("mixed_emissive", 0.8)])
def test_kernel_equals_checker_with_roulette(rtmi, rtcheck, scenes_dir, name, p):
    if name == "rtiow":
        sc = rtmi.Scene.rtiow(7, 80, 45, 6, 50)
    else:
        sc = rtmi.Scene.load(os.path.join(scenes_dir, name + ".json"))
        sc.override(width=72, height=40, spp=6, max_depth=30)
    sc.set_russian_roulette(p)
    img = sc.render(rtmi.Opts(seed=SEED))
    ref, cnt = rtcheck.oracle_render(sc, seed=SEED, want_counts=True)
    assert np.array_equal(img, ref)
    st = sc.count(rtmi.Opts(seed=SEED))
    assert (st.samples, st.queries, st.hits, st.misses, st.rng_draws) == tuple(cnt[k] for k in ("samples", "queries", "hits", "misses", "rng_draws"))
    # sharded and chunked renders stay bit-identical with roulette on
    out = np.zeros_like(img)
    for r in range(3):
        o = rtmi.Opts(seed=SEED, tile_first=r, tile_stride=3, spp_chunk=4)
        sc.scatter_rows(o, sc.render(o), out)
    assert np.array_equal(out, img)
    assert np.array_equal(sc.render(rtmi.Opts(seed=SEED, variant=16)), img)
