"""The product build (make ABLATIONS=0 -> librtmi_product.so): the product kernels alone -- no measurement variants, no
counting kernels, no RTMI_* environment knobs -- behind the same C ABI.  The default library (librtmi.so), which the rest
of the tests and bench.py load, carries all of those."""
import os
import re
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ray-tracing-in-cuda_amd")
DEFAULT, PRODUCT = os.path.join(PKG, "librtmi.so"), os.path.join(PKG, "librtmi_product.so")


def _nm(lib):
    return subprocess.run(["nm", "-C", "--defined-only", lib], capture_output=True, text=True, check=True).stdout


def _kernels(lib):
    return sorted(set(re.findall(r"__device_stub__render_kernel<([^>]*)>", _nm(lib))))


def test_product_library_has_the_same_abi_and_only_the_product_kernels():
    assert os.path.exists(PRODUCT), "build() makes it: make -C ray-tracing-in-cuda_amd product"
    exported = lambda lib: sorted(set(re.findall(r" T (rt_\w+)$", _nm(lib), flags=re.M)))
    assert exported(PRODUCT) == exported(DEFAULT) and "rt_render_hip_count" in exported(PRODUCT)
    prod, full = _kernels(PRODUCT), _kernels(DEFAULT)
    # <COUNT, POOL, SCALAR, CULL, EXT, SPH>: the sheet and 3-D walks over compact tables, the wide-table walk from LDS and from
    # global memory and the linear scan, the last three also with triangles / image textures
    assert prod == sorted(["false, true, false, 6, false, true", "false, true, false, 5, false, true",
                           "false, true, false, 7, false, false", "false, true, true, 7, false, false",
                           "false, true, false, 7, true, false", "false, true, true, 7, true, false",
                           "false, true, false, 0, false, false", "false, true, false, 0, true, false"])
    assert set(prod) < set(full) and len(full) >= len(prod) + 10
    assert "getenv" not in subprocess.run(["nm", "-D", "--undefined-only", PRODUCT], capture_output=True, text=True).stdout


_SCRIPT = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    from __graft_entry__ import load_package
    rtmi = load_package()
    import rtcheck
    assert not rtmi.has_ablations()
    cases = []
    cases.append((rtmi.Scene.rtiow(7, 96, 54, 3, 20), 2))
    sc = rtmi.Scene.load(os.path.join(%r, "ray-tracing-in-cuda_amd", "scenes", "mixed_emissive.json")); sc.override(64, 36, 3)
    cases.append((sc, 16))
    sc = rtmi.Scene.dna(30.0); sc.override(64, 36, 2)
    cases.append((sc, 36))
    from test_gpu_grid_all import height_field
    cases.append((height_field(rtmi, 6, 64, 36, 2, spheres=40, extent=6.0), 36))
    cases.append((height_field(rtmi, 40, 64, 36, 2, spheres=100), 44))
    for sc, want in cases:
        st = rtmi.Stats()
        img = sc.render(rtmi.Opts(seed=5), st)
        ref, _ = rtcheck.oracle_render(sc, seed=5)
        assert st.kernel_variant == want, (st.kernel_variant, want)
        assert np.array_equal(img, ref), want
    sc = cases[0][0]
    for v in (1, 40, 17, 24, 32, 64, 128):
        try:
            sc.render(rtmi.Opts(variant=v)); raise SystemExit("variant %%d accepted" %% v)
        except rtmi.RtmiError as e:
            assert e.status == 1 and "ABLATIONS" in str(e), str(e)
    try:
        sc.count(); raise SystemExit("count accepted")
    except rtmi.RtmiError as e:
        assert e.status == 6 and "ABLATIONS" in str(e), str(e)
    print("product build ok")
""") % (ROOT, ROOT, ROOT)


@pytest.mark.gpu
def test_product_library_renders_every_scene_class_and_refuses_the_rest():
    env = dict(os.environ, RTMI_LIB=PRODUCT, RTMI_GLOBAL_TABLE_BYTES="64")  # (the knob must be inert in this build)
    p = subprocess.run([sys.executable, "-c", _SCRIPT], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "product build ok" in p.stdout, p.stdout + p.stderr
