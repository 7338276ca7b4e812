"""include/rtmi.hpp: the reference's constructor argument lists over the C ABI (host side only,
no GPU): compile a small program with g++, build the 3-sphere scene, compare with the JSON one."""
import json
import os
import subprocess

SRC = r'''
#include <cstdio>
#include "rtmi.hpp"
int main() {
    rtmi::scene sc(400, 225, 100, 50);
    sc.set_background({0.5f, 0.7f, 1.0f}, true, true);
    sc.set_camera(rtmi::camera({-2, 2, 1}, {0, 0, -1}, {0, 1, 0}, 20.0f, 0.0f, 0.0f, 0.0f));
    auto glass = rtmi::dielectric(1.5f);
    sc.add(rtmi::sphere({0, 0, -1}, 0.5f, rtmi::lambertian(rtmi::color(0.1f, 0.2f, 0.5f))));
    sc.add(rtmi::sphere({0, -100.5f, -1}, 100.0f, rtmi::lambertian(rtmi::color(0.8f, 0.8f, 0.0f))));
    sc.add(rtmi::sphere({1, 0, -1}, 0.5f, rtmi::metal(rtmi::color(0.8f, 0.6f, 0.2f), 0.0f)));
    sc.add(rtmi::sphere({-1, 0, -1}, 0.5f, glass));
    sc.add(rtmi::sphere({-1, 0, -1}, -0.45f, glass));   // same material object -> same table entry
    auto tube = rtmi::cylinder(0.25f, -1.0f, 1.0f, rtmi::diffuse_light(rtmi::checker_texture({1, 1, 1}, {4, 0, 0})));
    tube.rotate({0, 1, 0}, 3.14159265358979f / 2);
    tube.translate({1, 2, 3});
    sc.add(tube);
    sc.add(rtmi::xz_rect(-1, 1, -2, 2, 0.5f, rtmi::metal(rtmi::color(1, 1, 1), 7.0f)));
    fputs(sc.to_json().c_str(), stdout);
    try { rtmi::scene bad("/nonexistent.json"); } catch (const rtmi::error &e) { fprintf(stderr, "caught: %s\n", e.what()); return 0; }
    return 1;
}
'''


def test_cpp_wrappers_build_the_same_tables(rtmi, scenes_dir, tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.dirname(rtmi.LIB_PATH)
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                    "-L", pkg, "-lrtmi", f"-Wl,-rpath,{pkg}"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "cannot open" in r.stderr
    got = rtmi.Scene.parse(r.stdout)
    want = rtmi.Scene.load(os.path.join(scenes_dir, "three_sphere.json"))
    assert got.prims()[:5].tobytes() == want.prims().tobytes()
    assert got.materials()[:4].tobytes() == want.materials().tobytes()   # glass registered once
    assert bytes(got.get_camera()) == bytes(want.get_camera())
    p, m, t = got.prims(), got.materials(), got.textures()
    assert len(p) == 7 and list(p["type"][5:]) == [4, 2]
    assert m["type"][p["material"][5]] == 3 and t["type"][m["texture"][p["material"][5]]] == 1
    assert m["fuzz"][p["material"][6]] == 1.0  # clamped
    d = json.loads(r.stdout)
    cyl = d["object"]["data"][5]
    assert abs(cyl["rotate"]["angle"] - 90.0) < 1e-3 and cyl["translate"] == [1, 2, 3]
