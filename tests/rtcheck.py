"""ctypes wrappers of the CHECKERS under oracle/ (test infrastructure only):

* ``oracle/librt_oracle.so``   -- fp32 CPU restatement (rt_oracle.c)
* ``oracle/_ref/libref_cpu.so`` -- the reference's own cmake-cpu-version sources compiled
  with a hooked rand() (ref_harness.cpp); spheres + lambertian/metal/dielectric only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "librt_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_cpu.so")


class _RtoCamera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("lower_left", C.c_float * 3), ("horizontal", C.c_float * 3),
                ("vertical", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
                ("lens_radius", C.c_float)]


class _RtoCameraParams(C.Structure):
    _fields_ = [("lookfrom", C.c_double * 3), ("lookat", C.c_double * 3), ("vup", C.c_double * 3),
                ("vfov", C.c_double), ("aspect", C.c_double), ("aperture", C.c_double), ("focus_dist", C.c_double)]


class _RtoImage(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("rgb", C.c_void_p)]


class _RtoScene(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_depth", C.c_int32), ("flags", C.c_uint32),
                ("background", C.c_float * 3), ("cam", _RtoCamera),
                ("prims", C.c_void_p), ("num_prims", C.c_int32),
                ("mats", C.c_void_p), ("num_mats", C.c_int32),
                ("texs", C.c_void_p), ("num_texs", C.c_int32), ("rr_p", C.c_float),
                ("images", C.c_void_p), ("num_images", C.c_int32)]


class _RtoCounts(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("queries", C.c_uint64), ("prim_tests", C.c_uint64), ("hits", C.c_uint64),
                ("misses", C.c_uint64), ("scatter", C.c_uint64 * 4), ("rng_draws", C.c_uint64)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "scatter"}
        d["scatter"] = list(self.scatter)
        return d


_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            raise FileNotFoundError(f"{ORACLE_SO} missing: run `make -C oracle`")
        lib = C.CDLL(ORACLE_SO)
        lib.rto_sample.restype = C.c_int
        lib.rto_sample.argtypes = [C.POINTER(_RtoScene), C.c_uint64, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(C.c_float), C.POINTER(_RtoCounts)]
        lib.rto_render.restype = C.c_int
        lib.rto_render.argtypes = [C.POINTER(_RtoScene), C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.POINTER(_RtoCounts), C.c_int]
        lib.rto_render_rect.restype = C.c_int
        lib.rto_render_rect.argtypes = [C.POINTER(_RtoScene), C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_void_p, C.POINTER(_RtoCounts), C.c_int]
        lib.rto_derive_camera.restype = None
        lib.rto_derive_camera.argtypes = [C.POINTER(_RtoCameraParams), C.POINTER(_RtoCamera)]
        lib.rto_philox4x32_10.restype = None
        lib.rto_philox4x32_10.argtypes = [C.POINTER(C.c_uint32)] * 3
        lib.rto_sample_stream.restype = None
        lib.rto_sample_stream.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_int]
        lib.rto_aabb_hit.restype = C.c_int
        lib.rto_aabb_hit.argtypes = [C.POINTER(C.c_float)] * 4 + [C.c_float, C.c_float]
        lib.rto_quantize.restype = C.c_int
        lib.rto_quantize.argtypes = [C.c_float, C.c_int, C.c_int]
        lib.rto_num_threads.restype = C.c_int
        _oracle = lib
    return _oracle


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def ref_lib():
    global _ref
    if _ref is None:
        if not have_ref():
            raise FileNotFoundError(f"{REF_SO} missing: run `make -C oracle` where /root/reference exists")
        lib = C.CDLL(REF_SO)
        d3 = C.POINTER(C.c_double)
        lib.ref_scene_new.restype = C.c_void_p
        lib.ref_scene_free.argtypes = [C.c_void_p]
        lib.ref_scene_set_camera.argtypes = [C.c_void_p, d3, d3, d3, C.c_double, C.c_double, C.c_double, C.c_double]
        lib.ref_scene_add_sphere.restype = C.c_int
        lib.ref_scene_add_sphere.argtypes = [C.c_void_p, d3, C.c_double, C.c_int, C.c_int, d3, d3, C.c_double,
                                             C.c_double]
        lib.ref_scene_num_objects.argtypes = [C.c_void_p]
        lib.ref_sample.restype = C.c_int
        lib.ref_sample.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, d3]
        lib.ref_render.restype = C.c_int
        lib.ref_render.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_void_p, C.c_int]
        lib.ref_time_rows.restype = C.c_double
        lib.ref_time_rows.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, d3]
        lib.ref_write_color.argtypes = [d3, C.c_int, C.POINTER(C.c_int)]
        lib.ref_num_threads.restype = C.c_int
        _ref = lib
    return _ref


# ------------------------------------------------------------------ oracle (fp32 restatement)
class OracleScene:
    """rto_scene built from the PRODUCT's exported tables (rt_scene_get_*)."""

    def __init__(self, scene):
        self.prims = scene.prims()
        self.mats = scene.materials()
        self.texs = scene.textures()
        assert self.prims.dtype.itemsize == 128 and self.mats.dtype.itemsize == 28 and self.texs.dtype.itemsize == 28
        info = scene.info
        cam = scene.get_camera()
        s = _RtoScene()
        s.width, s.height, s.max_depth, s.flags = info.width, info.height, info.max_depth, info.flags
        for i in range(3):
            s.background[i] = info.background[i]
            for name in ("origin", "lower_left", "horizontal", "vertical", "u", "v", "w"):
                getattr(s.cam, name)[i] = getattr(cam, name)[i]
        s.cam.lens_radius = cam.lens_radius
        s.prims, s.num_prims = self.prims.ctypes.data, len(self.prims)
        s.mats, s.num_mats = self.mats.ctypes.data, len(self.mats)
        s.texs, s.num_texs = self.texs.ctypes.data, len(self.texs)
        s.rr_p = info.russian_roulette
        # pixels of the image textures (texture record: c0 = {image index, rows, cols})
        self.images = {}
        for ti, t in enumerate(self.texs):
            if t["type"] == 2:
                self.images[int(t["c0"][0])] = np.ascontiguousarray(scene.get_image(ti))
        n_img = (max(self.images) + 1) if self.images else 0
        self.image_recs = (_RtoImage * max(1, n_img))()
        for k, px in self.images.items():
            self.image_recs[k].rows, self.image_recs[k].cols, self.image_recs[k].rgb = px.shape[0], px.shape[1], px.ctypes.data
        s.images, s.num_images = C.cast(self.image_recs, C.c_void_p), n_img
        self.c = s
        self.width, self.height, self.spp = info.width, info.height, info.samples_per_pixel


def oracle_render(scene, seed=2023, rows=None, sample_first=0, sample_count=None, spp_chunk=0, threads=0,
                  want_counts=False):
    """Full-image (H, W, 3) fp32 sums; only rows [y0, y1) are filled when rows=(y0, y1)."""
    lib = oracle_lib()
    osc = scene if isinstance(scene, OracleScene) else OracleScene(scene)
    y0, y1 = rows if rows is not None else (0, osc.height)
    n = osc.spp if sample_count is None else sample_count
    out = np.zeros((osc.height, osc.width, 3), dtype=np.float32)
    counts = _RtoCounts()
    rc = lib.rto_render(C.byref(osc.c), seed, y0, y1, sample_first, n, spp_chunk, out.ctypes.data,
                        C.byref(counts) if want_counts else None, threads)
    assert rc == 0
    return out, (counts.as_dict() if want_counts else None)


def oracle_render_rect(scene, seed, x0, x1, y0, y1, sample_first=0, sample_count=None, threads=0):
    """fp32 sums of the pixel window [x0, x1) x [y0, y1) only, shape (y1 - y0, x1 - x0, 3)."""
    lib = oracle_lib()
    osc = scene if isinstance(scene, OracleScene) else OracleScene(scene)
    n = osc.spp if sample_count is None else sample_count
    out = np.zeros((osc.height, osc.width, 3), dtype=np.float32)
    rc = lib.rto_render_rect(C.byref(osc.c), seed, x0, x1, y0, y1, sample_first, n, out.ctypes.data, None, threads)
    assert rc == 0
    return out[y0:y1, x0:x1].copy()


def oracle_hit_uv(scene, origin, direction):
    """Closest hit of one ray in the checker: (hit?, (u, v), t, prim index) -- the hit record's texture coordinates."""
    lib = oracle_lib()
    osc = scene if isinstance(scene, OracleScene) else OracleScene(scene)
    lib.rto_hit_uv.restype = C.c_int
    lib.rto_hit_uv.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                               C.POINTER(C.c_float), C.POINTER(C.c_int)]
    o, d = (C.c_float * 3)(*origin), (C.c_float * 3)(*direction)
    uv, t, prim = (C.c_float * 2)(), C.c_float(), C.c_int()
    hit = lib.rto_hit_uv(C.cast(C.byref(osc.c), C.c_void_p), o, d, uv, C.byref(t), C.byref(prim))
    return bool(hit), (uv[0], uv[1]), t.value, prim.value


def oracle_sample(scene, seed, x, y, sample):
    lib = oracle_lib()
    osc = scene if isinstance(scene, OracleScene) else OracleScene(scene)
    rgb = (C.c_float * 3)()
    q = lib.rto_sample(C.byref(osc.c), seed, x, y, sample, rgb, None)
    return np.array(rgb[:], dtype=np.float32), q


def oracle_trace_sample(scene, seed, x, y, sample, max_queries=64):
    """The closest-hit queries of one sample in the checker: rows of (origin xyz, direction xyz, hit t or -1, hit)."""
    lib = oracle_lib()
    osc = scene if isinstance(scene, OracleScene) else OracleScene(scene)
    rgb = (C.c_float * 3)()
    buf = np.zeros((max_queries, 8), dtype=np.float32)
    lib.rto_trace_sample.restype = C.c_int
    lib.rto_trace_sample.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    n = lib.rto_trace_sample(C.cast(C.byref(osc.c), C.c_void_p), seed, x, y, sample, C.cast(rgb, C.c_void_p),
                             buf.ctypes.data_as(C.c_void_p), max_queries)
    return np.array(rgb[:], dtype=np.float32), buf[:n]


def oracle_derive_camera(lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist):
    lib = oracle_lib()
    p = _RtoCameraParams()
    for i in range(3):
        p.lookfrom[i], p.lookat[i], p.vup[i] = lookfrom[i], lookat[i], vup[i]
    p.vfov, p.aspect, p.aperture, p.focus_dist = vfov, aspect, aperture, focus_dist
    out = _RtoCamera()
    lib.rto_derive_camera(C.byref(p), C.byref(out))
    return out


def camera_params_of(scene):
    """fp64 camera parameters as the scene file holds them (through rt_scene_to_json)."""
    j = json.loads(scene.to_json())
    cam = j["camera"]
    aspect = j["width"] / j["height"]
    d = [a - b for a, b in zip(cam["lookfrom"], cam["lookat"])]
    focus = cam.get("focus_dist", math.sqrt(sum(v * v for v in d)))
    return dict(lookfrom=cam["lookfrom"], lookat=cam["lookat"], vup=cam["vup"], vfov=cam["vfov"], aspect=aspect,
                aperture=cam["aperture"], focus_dist=focus)


# ------------------------------------------------------------------ reference build (fp64)
def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


class RefScene:
    """The scene rebuilt with the reference's own constructors (oracle/_ref)."""

    def __init__(self, scene):
        lib = ref_lib()
        info = scene.info
        if not (info.flags & 1) or not (info.flags & 2):
            raise ValueError("the reference CPU renderer always uses the sky gradient and defocus blur")
        prims, mats, texs = scene.prims(), scene.materials(), scene.textures()
        self.h = lib.ref_scene_new()
        self.lib = lib
        for p in prims:
            if p["type"] != 0:
                raise ValueError("cmake-cpu-version's hittable_list holds spheres only")
            m = mats[p["material"]]
            c0, c1, tex_type = [0, 0, 0], [0, 0, 0], 0
            if m["type"] == 0:
                t = texs[m["texture"]]
                tex_type, c0, c1 = int(t["type"]), t["c0"], t["c1"]
            elif m["type"] == 1:
                c0 = m["albedo"]
            rc = lib.ref_scene_add_sphere(self.h, _d3(p["f"][:3]), float(p["f"][3]), int(m["type"]), tex_type,
                                          _d3(c0), _d3(c1), float(m["fuzz"]), float(m["ir"]))
            if rc != 0:
                raise ValueError("material not expressible in cmake-cpu-version")
        cp = camera_params_of(scene)
        lib.ref_scene_set_camera(self.h, _d3(cp["lookfrom"]), _d3(cp["lookat"]), _d3(cp["vup"]), cp["vfov"],
                                 cp["aspect"], cp["aperture"], cp["focus_dist"])
        self.width, self.height, self.spp, self.max_depth = info.width, info.height, info.samples_per_pixel, info.max_depth

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_scene_free(self.h)
            self.h = None

    def sample(self, seed, x, y, s):
        rgb = (C.c_double * 3)()
        draws = self.lib.ref_sample(self.h, seed, x, y, s, self.width, self.height, self.max_depth, rgb)
        return np.array(rgb[:]), draws

    def render(self, seed=2023, rows=None, sample_first=0, spp=None, threads=0):
        y0, y1 = rows if rows is not None else (0, self.height)
        out = np.zeros((self.height, self.width, 3), dtype=np.float64)
        rc = self.lib.ref_render(self.h, seed, self.width, self.height, y0, y1, sample_first,
                                 self.spp if spp is None else spp, self.max_depth, out.ctypes.data, threads)
        assert rc == 0
        return out

    def time_rows(self, seed, y0, y1, spp, threads=0):
        chk = C.c_double()
        sec = self.lib.ref_time_rows(self.h, seed, self.width, self.height, y0, y1, spp, self.max_depth, threads,
                                     C.byref(chk))
        return sec, chk.value


def ref_write_color(rgb_sum, spp):
    out = (C.c_int * 3)()
    ref_lib().ref_write_color(_d3(rgb_sum), spp, out)
    return list(out)
