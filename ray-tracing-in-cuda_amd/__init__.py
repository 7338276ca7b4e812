"""rtmi -- Python host mirror of the reference's scene-JSON -> render -> PPM interface.

Thin ctypes binding over the C ABI of ``librtmi.so`` (``include/rtmi.h``).  Names
follow the reference (``gpu-version/parser.hpp:504`` ``parse_scene``,
``gpu-version/main.cu:359`` ``output_image``, the ``camera`` / ``sphere`` /
``xy_rect`` / ``cylinder`` / ``lambertian`` / ``metal`` / ``dielectric`` /
``diffuse_light`` / ``solid_color`` / ``checker_texture`` constructors).

There is NO CPU rendering path in this package: if ``librtmi.so`` is missing the
import fails, and if no gfx950 device is usable every ``render*`` call raises
``RtmiError``.

The directory name (``ray-tracing-in-cuda_amd``) is not a valid Python identifier;
load the package with ``__graft_entry__.load_package()`` (module name ``rtmi``).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

try:  # torch first, so that one HIP runtime (torch's bundled one) serves both
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for the binding itself
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
# RTMI_LIB: alternative build of the same library (kernel tuning A/B); default is the in-tree one
_LIB_PATH = os.environ.get("RTMI_LIB") or os.path.join(_HERE, "librtmi.so")
if not os.path.exists(_LIB_PATH):
    raise ImportError(
        f"{_LIB_PATH} not found: build it with `make -C {_HERE}` (or __graft_entry__.build()); "
        "this package has no fallback implementation"
    )
_lib = C.CDLL(_LIB_PATH)

RT_OK = 0
FLAG_SKY_GRADIENT = 1
FLAG_DEFOCUS_BLUR = 2
PRIM_SPHERE, PRIM_XY_RECT, PRIM_XZ_RECT, PRIM_YZ_RECT, PRIM_CYLINDER, PRIM_TRIANGLE = range(6)
MAT_LAMBERTIAN, MAT_METAL, MAT_DIELECTRIC, MAT_DIFFUSE_LIGHT = range(4)
TEX_SOLID, TEX_CHECKER, TEX_IMAGE = range(3)

# numpy views of the table records (layouts of include/rtmi.h)
PRIM_DTYPE = np.dtype(
    [("type", "<i4"), ("material", "<i4"), ("f", "<f4", (6,)), ("m", "<f4", (12,)), ("m_inv", "<f4", (12,))]
)
MATERIAL_DTYPE = np.dtype(
    [("type", "<i4"), ("texture", "<i4"), ("albedo", "<f4", (3,)), ("fuzz", "<f4"), ("ir", "<f4")]
)
TEXTURE_DTYPE = np.dtype([("type", "<i4"), ("c0", "<f4", (3,)), ("c1", "<f4", (3,))])


class RtmiError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        msg = _lib.rt_last_error().decode(errors="replace")
        kind = _lib.rt_status_string(status).decode()
        super().__init__(f"{where}: {kind} ({status}): {msg}")


class _Camera(C.Structure):
    _fields_ = [
        ("lookfrom", C.c_float * 3), ("lookat", C.c_float * 3), ("vup", C.c_float * 3),
        ("vfov", C.c_float), ("aspect", C.c_float), ("aperture", C.c_float), ("focus_dist", C.c_float),
        ("origin", C.c_float * 3), ("lower_left", C.c_float * 3), ("horizontal", C.c_float * 3),
        ("vertical", C.c_float * 3), ("u", C.c_float * 3), ("v", C.c_float * 3), ("w", C.c_float * 3),
        ("lens_radius", C.c_float),
    ]


class _Info(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("samples_per_pixel", C.c_int32), ("max_depth", C.c_int32),
        ("num_prims", C.c_int32), ("num_materials", C.c_int32), ("num_textures", C.c_int32),
        ("flags", C.c_uint32), ("background", C.c_float * 3), ("russian_roulette", C.c_float),
    ]


class Opts(C.Structure):
    """rt_opts (include/rtmi.h)."""
    _fields_ = [
        ("seed", C.c_uint64), ("device", C.c_int32), ("tile_rows", C.c_int32), ("tile_first", C.c_int32),
        ("tile_stride", C.c_int32), ("tile_rotate", C.c_int32), ("spp_chunk", C.c_int32), ("sample_first", C.c_int32),
        ("sample_count", C.c_int32), ("variant", C.c_uint32),
    ]

    def __init__(self, **kw):
        super().__init__()
        _lib.rt_opts_default(C.byref(self))
        for k, v in kw.items():
            if not hasattr(self, k):
                raise TypeError(f"rt_opts has no field {k!r}")
            setattr(self, k, v)


class Stats(C.Structure):
    """rt_stats (include/rtmi.h)."""
    _fields_ = [
        ("kernel_ms", C.c_double), ("upload_ms", C.c_double), ("launches", C.c_int32), ("local_rows", C.c_int32),
        ("samples", C.c_uint64), ("queries", C.c_uint64), ("prim_tests", C.c_uint64), ("hits", C.c_uint64),
        ("misses", C.c_uint64), ("scatter", C.c_uint64 * 4), ("rng_draws", C.c_uint64),
        ("cand_lanes", C.c_uint64), ("cand_waves", C.c_uint64), ("clusters_visited", C.c_uint64),
        ("wave_queries", C.c_uint64), ("groups_visited", C.c_uint64), ("lane_clusters", C.c_uint64),
        ("lane_groups", C.c_uint64), ("group_maxpop", C.c_uint64), ("query_maxpop", C.c_uint64), ("cycles", C.c_uint64 * 6),
        ("cull_prefix", C.c_int32),
        ("cull_clusters", C.c_int32), ("cull_groups", C.c_int32), ("cull_cluster_size", C.c_int32),
        ("wave_start_spread_us", C.c_double), ("wave_end_spread_us", C.c_double), ("wave_span_us", C.c_double),
        ("lane_cands", C.c_uint64), ("cull_mode", C.c_int32), ("cull_windows", C.c_int32), ("gather_ms", C.c_double), ("devices_used", C.c_int32), ("grid_sheet", C.c_int32),
        ("kernel_variant", C.c_int32),
    ]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n not in ("scatter", "cycles")}
        d["scatter"] = list(self.scatter)
        d["cycles"] = list(self.cycles)
        return d


class TableInfo(C.Structure):
    """rt_table_info (include/rtmi.h): the device tables the host builds for a scene."""
    _fields_ = [
        ("image_floats", C.c_int32), ("grid_wide", C.c_int32), ("grid_sheet", C.c_int32), ("grid_cells", C.c_int32),
        ("grid_n", C.c_int32 * 3), ("grid_min", C.c_float * 3), ("grid_size", C.c_float * 3),
        ("ob_near2", C.c_float), ("ob_far2", C.c_float),
        ("ns", C.c_int32), ("np", C.c_int32), ("ncl", C.c_int32), ("nr", C.c_int32), ("nc", C.c_int32), ("nt", C.c_int32),
        ("nr_a", C.c_int32), ("nc_a", C.c_int32), ("nt_a", C.c_int32),
        ("off_grid_cells", C.c_int32), ("off_grid_items", C.c_int32),
        ("off_sph_cold", C.c_int32), ("off_rect_cold", C.c_int32), ("off_cyl_cold", C.c_int32), ("off_tri_cold", C.c_int32),
        ("off_rect_hot", C.c_int32), ("off_cyl_hot", C.c_int32), ("off_tri_hot", C.c_int32),
        ("hot_bytes_grid", C.c_int32), ("kernel_variant", C.c_int32),
    ]


def _sig(name, restype, *argtypes):
    fn = getattr(_lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


_p = C.c_void_p
_f3 = C.POINTER(C.c_float)
_sig("rt_last_error", C.c_char_p)
_sig("rt_status_string", C.c_char_p, C.c_int)
_sig("rt_abi_version", C.c_int)
_sig("rt_has_ablations", C.c_int)
_sig("rt_device_count", C.c_int)
_sig("rt_struct_size", C.c_size_t, C.c_int)
_sig("rt_opts_default", None, C.POINTER(Opts))
_sig("rt_scene_load_json", _p, C.c_char_p)
_sig("rt_scene_parse_json", _p, C.c_char_p, C.c_size_t)
_sig("rt_scene_rtiow", _p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int)
_sig("rt_scene_to_json", C.c_size_t, _p, C.c_char_p, C.c_size_t)
_sig("rt_scene_free", None, _p)
_sig("rt_scene_new", _p, C.c_int, C.c_int, C.c_int, C.c_int)
_sig("rt_scene_set_background", C.c_int, _p, _f3, C.c_uint32)
_sig("rt_scene_set_camera", C.c_int, _p, _f3, _f3, _f3, C.c_float, C.c_float, C.c_float, C.c_float)
_sig("rt_scene_add_solid_color", C.c_int, _p, _f3)
_sig("rt_scene_add_checker", C.c_int, _p, _f3, _f3)
_sig("rt_scene_add_lambertian", C.c_int, _p, C.c_int)
_sig("rt_scene_add_metal", C.c_int, _p, _f3, C.c_float)
_sig("rt_scene_add_dielectric", C.c_int, _p, C.c_float)
_sig("rt_scene_add_diffuse_light", C.c_int, _p, C.c_int)
_sig("rt_scene_add_sphere", C.c_int, _p, _f3, C.c_float, C.c_int)
_sig("rt_scene_add_rect", C.c_int, _p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int)
_sig("rt_scene_add_cylinder", C.c_int, _p, C.c_float, C.c_float, C.c_float, C.c_int, _f3, C.c_float, _f3)
_sig("rt_scene_add_image_texture", C.c_int, _p, C.c_int, C.c_int, _p)
_sig("rt_scene_add_image_texture_file", C.c_int, _p, C.c_char_p)
_sig("rt_scene_get_image", C.c_int, _p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), _p, C.c_size_t)
_sig("rt_scene_add_triangle", C.c_int, _p, _f3, _f3, _f3, _f3, _f3, _f3, C.c_int)
_sig("rt_scene_add_obj", C.c_int, _p, C.c_char_p, C.c_int, C.c_float, _f3, _f3)
_sig("rt_scene_override", C.c_int, _p, C.c_int, C.c_int, C.c_int, C.c_int)
_sig("rt_scene_rotate_cylinders", C.c_int, _p, C.c_double)
_sig("rt_scene_set_output_file", C.c_int, _p, C.c_char_p)
_sig("rt_scene_dna", _p, _p, C.c_double)
_sig("rt_scene_clone", _p, _p)
_sig("rt_scene_get_info", C.c_int, _p, C.POINTER(_Info))
_sig("rt_scene_get_camera", C.c_int, _p, C.POINTER(_Camera))
_sig("rt_scene_get_prims", C.c_int, _p, _p, C.c_int)
_sig("rt_scene_get_materials", C.c_int, _p, _p, C.c_int)
_sig("rt_scene_get_textures", C.c_int, _p, _p, C.c_int)
_sig("rt_shard_rows", C.c_int, _p, C.POINTER(Opts))
_sig("rt_shard_global_row", C.c_int, _p, C.POINTER(Opts), C.c_int)
_sig("rt_shard_deal", C.c_int, _p, C.POINTER(Opts), C.c_int)
_sig("rt_render_hip_device", C.c_int, _p, C.POINTER(Opts), _p, _p, C.POINTER(Stats))
_sig("rt_render_hip", C.c_int, _p, C.POINTER(Opts), _p, C.POINTER(Stats))
_sig("rt_render_hip_tiles", C.c_int, _p, C.POINTER(Opts), C.POINTER(C.c_int), C.c_int, _p, C.POINTER(Stats))
_sig("rt_tiles_shutdown", None)
_sig("rt_shard_place_rows_device", C.c_int, _p, C.POINTER(Opts), C.c_int, C.c_int, _p, _p, _p)
_sig("rt_scene_set_russian_roulette", C.c_int, _p, C.c_float)
_sig("rt_render_hip_count", C.c_int, _p, C.POINTER(Opts), _p, C.POINTER(Stats))
_sig("rt_scene_table_info", C.c_int, _p, C.POINTER(TableInfo))
_sig("rt_scene_table_image", C.c_int, _p, _p, C.c_int)
_sig("rt_render_hip_accumulate", C.c_int, _p, C.POINTER(Opts), _p, _p, C.POINTER(Stats))
_sig("rt_acc_to_rgb", None, _p, _p, C.c_size_t)
_sig("rt_shard_scatter_rows", C.c_int, _p, C.POINTER(Opts), _p, _p)
_sig("rt_write_ppm", C.c_int, C.c_char_p, _p, C.c_int, C.c_int, C.c_int)
_sig("rt_quantize_rgb8", C.c_int, _p, C.c_int, C.c_int, C.c_int, C.c_int, _p)
_sig("rt_write_png", C.c_int, C.c_char_p, _p, C.c_int, C.c_int, C.c_int, C.c_int)
_sig("rt_scene_output_file", C.c_char_p, _p)
_sig("rt_philox4x32_10", None, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32))
_sig("rt_aabb_hit", C.c_int, _f3, _f3, _f3, _f3, C.c_float, C.c_float)
_sig("rt_sample_stream", None, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_int)

C_SYMBOLS = [
    "rt_last_error", "rt_status_string", "rt_abi_version", "rt_struct_size", "rt_device_count", "rt_has_ablations", "rt_opts_default",
    "rt_scene_table_info", "rt_scene_table_image",
    "rt_scene_load_json", "rt_scene_parse_json", "rt_scene_rtiow", "rt_scene_to_json", "rt_scene_free",
    "rt_scene_new", "rt_scene_set_background", "rt_scene_set_camera", "rt_scene_add_solid_color",
    "rt_scene_add_checker", "rt_scene_add_lambertian", "rt_scene_add_metal", "rt_scene_add_dielectric",
    "rt_scene_add_diffuse_light", "rt_scene_add_sphere", "rt_scene_add_rect", "rt_scene_add_cylinder",
    "rt_scene_add_image_texture", "rt_scene_add_image_texture_file", "rt_scene_get_image", "rt_scene_add_triangle",
    "rt_scene_add_obj",
    "rt_scene_override", "rt_scene_get_info", "rt_scene_get_camera", "rt_scene_get_prims",
    "rt_scene_get_materials", "rt_scene_get_textures", "rt_shard_rows", "rt_shard_global_row", "rt_shard_deal",
    "rt_render_hip_device", "rt_render_hip", "rt_render_hip_tiles", "rt_tiles_shutdown", "rt_shard_place_rows_device", "rt_render_hip_count", "rt_render_hip_accumulate", "rt_scene_set_russian_roulette",
    "rt_acc_to_rgb", "rt_shard_scatter_rows", "rt_write_ppm",
    "rt_quantize_rgb8", "rt_philox4x32_10", "rt_aabb_hit", "rt_sample_stream", "rt_write_png",
    "rt_scene_output_file", "rt_scene_rotate_cylinders", "rt_scene_set_output_file", "rt_scene_dna", "rt_scene_clone",
]


def _v3(v):
    a = (C.c_float * 3)(*[float(x) for x in v])
    return a


def _check(status: int, where: str):
    if status != RT_OK:
        raise RtmiError(status, where)


def _check_id(rc: int, where: str) -> int:
    if rc < 0:
        raise RtmiError(-rc, where)
    return rc


class Scene:
    """gpu-version/parser.hpp:16-32 ``struct scene``: the flattened scene tables."""

    def __init__(self, handle):
        if not handle:
            raise RtmiError(4, "scene")
        self._h = C.c_void_p(handle)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:  # (module globals are already cleared when the interpreter shuts down)
            _lib.rt_scene_free(h)
            self._h = None

    # ---- construction ------------------------------------------------------
    @classmethod
    def load(cls, path: str) -> "Scene":
        h = _lib.rt_scene_load_json(os.fsencode(path))
        if not h:
            raise RtmiError(_guess_status(), f"parse_scene({path!r})")
        return cls(h)

    @classmethod
    def parse(cls, text: str) -> "Scene":
        b = text.encode()
        h = _lib.rt_scene_parse_json(b, len(b))
        if not h:
            raise RtmiError(_guess_status(), "parse_scene(<text>)")
        return cls(h)

    @classmethod
    def rtiow(cls, seed=7, width=400, height=225, spp=100, max_depth=50) -> "Scene":
        """random_scene() + camera of cmake-cpu-version/main.cpp:89-94, 125-172."""
        h = _lib.rt_scene_rtiow(seed, width, height, spp, max_depth)
        if not h:
            raise RtmiError(_guess_status(), "rt_scene_rtiow")
        return cls(h)

    @classmethod
    def new(cls, width, height, spp, max_depth=50) -> "Scene":
        return cls(_lib.rt_scene_new(width, height, spp, max_depth))

    # ---- builders (reference constructor argument lists) --------------------
    def set_background(self, rgb=(0, 0, 0), sky_gradient=False, defocus_blur=True):
        flags = (FLAG_SKY_GRADIENT if sky_gradient else 0) | (FLAG_DEFOCUS_BLUR if defocus_blur else 0)
        _check(_lib.rt_scene_set_background(self._h, _v3(rgb), flags), "set_background")

    def set_russian_roulette(self, p: float):
        """Survival probability per bounce (4_0_path_tracing.py's p_RR); 0 switches it off."""
        _check(_lib.rt_scene_set_russian_roulette(self._h, float(p)), "set_russian_roulette")

    def camera(self, lookfrom, lookat, vup, vfov, aspect_ratio=0.0, aperture=0.0, focus_dist=0.0):
        _check(_lib.rt_scene_set_camera(self._h, _v3(lookfrom), _v3(lookat), _v3(vup), vfov, aspect_ratio,
                                        aperture, focus_dist), "camera")

    def solid_color(self, rgb) -> int:
        return _check_id(_lib.rt_scene_add_solid_color(self._h, _v3(rgb)), "solid_color")

    def checker_texture(self, even, odd) -> int:
        return _check_id(_lib.rt_scene_add_checker(self._h, _v3(even), _v3(odd)), "checker_texture")

    def image_texture(self, pixels) -> int:
        """Image texture (taichi-version/material.py:137-144) from a (rows, cols, 3) uint8 array or a PPM file."""
        if isinstance(pixels, (str, bytes, os.PathLike)):
            return _check_id(_lib.rt_scene_add_image_texture_file(self._h, os.fsencode(pixels)), "image_texture")
        px = np.ascontiguousarray(pixels, dtype=np.uint8)
        if px.ndim != 3 or px.shape[2] != 3:
            raise ValueError("image texture pixels must have shape (rows, cols, 3)")
        return _check_id(_lib.rt_scene_add_image_texture(self._h, px.shape[0], px.shape[1], px.ctypes.data_as(C.c_void_p)),
                         "image_texture")

    def get_image(self, texture: int) -> np.ndarray:
        rows, cols = C.c_int(), C.c_int()
        _check_id(_lib.rt_scene_get_image(self._h, texture, C.byref(rows), C.byref(cols), None, 0), "rt_scene_get_image")
        out = np.empty((rows.value, cols.value, 3), dtype=np.uint8)
        _check_id(_lib.rt_scene_get_image(self._h, texture, None, None, out.ctypes.data_as(C.c_void_p), out.nbytes),
                  "rt_scene_get_image")
        return out

    def triangle(self, v1, v2, v3, material, u1=(0, 0), u2=(0, 0), u3=(0, 0)) -> int:
        """Triangle(v1, v2, v3, u1, u2, u3, material), taichi-version/hittable.py:95-110."""
        uv = [(C.c_float * 3)(float(u[0]), float(u[1]), 0.0) for u in (u1, u2, u3)]
        return _check_id(_lib.rt_scene_add_triangle(self._h, _v3(v1), _v3(v2), _v3(v3), uv[0], uv[1], uv[2], material),
                         "triangle")

    def add_obj(self, path, material, scale=1.0, matrix=None, translate=None) -> int:
        """readobj + placement, taichi-version/main.py:23-41, 110-118; returns the number of triangles added."""
        m = (C.c_float * 9)(*[float(x) for x in np.asarray(matrix, dtype=np.float64).reshape(9)]) if matrix is not None else None
        t = _v3(translate) if translate is not None else None
        return _check_id(_lib.rt_scene_add_obj(self._h, os.fsencode(path), material, scale, m, t), "add_obj")

    def lambertian(self, texture_or_color) -> int:
        tex = texture_or_color if isinstance(texture_or_color, int) else self.solid_color(texture_or_color)
        return _check_id(_lib.rt_scene_add_lambertian(self._h, tex), "lambertian")

    def metal(self, albedo, fuzz) -> int:
        return _check_id(_lib.rt_scene_add_metal(self._h, _v3(albedo), fuzz), "metal")

    def dielectric(self, index_of_refraction) -> int:
        return _check_id(_lib.rt_scene_add_dielectric(self._h, index_of_refraction), "dielectric")

    def diffuse_light(self, texture_or_color) -> int:
        tex = texture_or_color if isinstance(texture_or_color, int) else self.solid_color(texture_or_color)
        return _check_id(_lib.rt_scene_add_diffuse_light(self._h, tex), "diffuse_light")

    def sphere(self, center, radius, material) -> int:
        return _check_id(_lib.rt_scene_add_sphere(self._h, _v3(center), radius, material), "sphere")

    def xy_rect(self, x0, x1, y0, y1, k, material) -> int:
        return _check_id(_lib.rt_scene_add_rect(self._h, 0, x0, x1, y0, y1, k, material), "xy_rect")

    def xz_rect(self, x0, x1, z0, z1, k, material) -> int:
        return _check_id(_lib.rt_scene_add_rect(self._h, 1, x0, x1, z0, z1, k, material), "xz_rect")

    def yz_rect(self, y0, y1, z0, z1, k, material) -> int:
        return _check_id(_lib.rt_scene_add_rect(self._h, 2, y0, y1, z0, z1, k, material), "yz_rect")

    def cylinder(self, radius, zmin, zmax, material, rotate=None, translate=None) -> int:
        """rotate = (axis, degrees); applied before translate (parser.hpp:423-440)."""
        axis = _v3(rotate[0]) if rotate else None
        deg = float(rotate[1]) if rotate else 0.0
        off = _v3(translate) if translate is not None else None
        return _check_id(_lib.rt_scene_add_cylinder(self._h, radius, zmin, zmax, material, axis, deg, off), "cylinder")

    # ---- animation (blue.py / blue2.py / dna.py) ---------------------------------------------
    def rotate_cylinders(self, degrees: float) -> int:
        return _check_id(_lib.rt_scene_rotate_cylinders(self._h, degrees), "rotate_cylinders")

    def set_output_file(self, path: str):
        _check(_lib.rt_scene_set_output_file(self._h, os.fsencode(path)), "set_output_file")

    def clone(self) -> "Scene":
        return Scene(_lib.rt_scene_clone(self._h))

    @classmethod
    def dna(cls, angle_degrees: float, base: "Scene | None" = None) -> "Scene":
        """dna.py:17-98: the DNA animation frame at this angle over `base` (default: basic_scene.json)."""
        h = _lib.rt_scene_dna(base._h if base is not None else None, angle_degrees)
        if not h:
            raise RtmiError(_guess_status(), "rt_scene_dna")
        return cls(h)

    def override(self, width=0, height=0, spp=0, max_depth=0):
        _check(_lib.rt_scene_override(self._h, width, height, spp, max_depth), "override")

    # ---- read-back -----------------------------------------------------------
    @property
    def info(self) -> _Info:
        i = _Info()
        _check(_lib.rt_scene_get_info(self._h, C.byref(i)), "get_info")
        return i

    @property
    def width(self):
        return self.info.width

    @property
    def height(self):
        return self.info.height

    @property
    def spp(self):
        return self.info.samples_per_pixel

    @property
    def max_depth(self):
        return self.info.max_depth

    def get_camera(self) -> _Camera:
        c = _Camera()
        _check(_lib.rt_scene_get_camera(self._h, C.byref(c)), "get_camera")
        return c

    def _table(self, fn, dtype):
        n = _check_id(fn(self._h, None, 0), fn.__name__)
        arr = np.zeros(n, dtype=dtype)
        if n:
            fn(self._h, arr.ctypes.data_as(C.c_void_p), n)
        return arr

    def prims(self) -> np.ndarray:
        return self._table(_lib.rt_scene_get_prims, PRIM_DTYPE)

    def materials(self) -> np.ndarray:
        return self._table(_lib.rt_scene_get_materials, MATERIAL_DTYPE)

    def textures(self) -> np.ndarray:
        return self._table(_lib.rt_scene_get_textures, TEXTURE_DTYPE)

    @property
    def output_file(self) -> str:
        return _lib.rt_scene_output_file(self._h).decode()

    def to_json(self) -> str:
        n = _lib.rt_scene_to_json(self._h, None, 0)
        buf = C.create_string_buffer(n)
        _lib.rt_scene_to_json(self._h, buf, n)
        return buf.value.decode()

    # ---- shard geometry ---------------------------------------------------------
    def shard_rows(self, opts: Opts | None = None) -> int:
        opts = opts or Opts()
        return _check_id(_lib.rt_shard_rows(self._h, C.byref(opts)), "rt_shard_rows")

    def shard_deal(self, opts: Opts | None, n_ranks: int) -> int:
        """rt_opts.tile_rotate that rt_render_hip_tiles uses to cut this frame into n_ranks shards (rt_shard_deal)."""
        opts = opts or Opts()
        return _check_id(_lib.rt_shard_deal(self._h, C.byref(opts), n_ranks), "rt_shard_deal")

    def shard_global_rows(self, opts: Opts | None = None) -> np.ndarray:
        opts = opts or Opts()
        n = self.shard_rows(opts)
        return np.array([_lib.rt_shard_global_row(self._h, C.byref(opts), r) for r in range(n)], dtype=np.int64)

    # ---- render (HIP only) --------------------------------------------------------
    def render(self, opts: Opts | None = None, stats: Stats | None = None) -> np.ndarray:
        """render<<<>>> + copy back (main.cu:505-513): (local_rows, W, 3) fp32 SUMS, row 0 = bottom."""
        opts = opts or Opts()
        rows = self.shard_rows(opts)
        out = np.empty((rows, self.width, 3), dtype=np.float32)
        st = stats if stats is not None else Stats()
        _check(_lib.rt_render_hip(self._h, C.byref(opts), out.ctypes.data_as(C.c_void_p), C.byref(st)), "rt_render_hip")
        return out

    def render_tiles(self, devices=None, opts: Opts | None = None, stats: Stats | None = None, n: int | None = None,
                     out: np.ndarray | None = None):
        """One frame over several GPUs of this node: row tiles dealt out to `devices` (ordinals; None =
        0..n-1), one ncclGather to devices[0] (rt_render_hip_tiles).  Returns the (H, W, 3) fp32 sums (in `out` when given:
        a C-contiguous float32 array of that shape)."""
        opts = opts or Opts()
        if devices is None:
            count = int(n if n is not None else 1)
            arr = None
        else:
            count = len(devices)
            arr = (C.c_int * count)(*[int(d) for d in devices])
        if out is None:
            out = np.empty((self.height, self.width, 3), dtype=np.float32)
        elif out.shape != (self.height, self.width, 3) or out.dtype != np.float32 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("render_tiles: out must be a C-contiguous float32 array of shape (height, width, 3)")
        _check(_lib.rt_render_hip_tiles(self._h, C.byref(opts), arr, count, out.ctypes.data_as(C.c_void_p),
                                        C.byref(stats) if stats is not None else None), "rt_render_hip_tiles")
        return out

    def place_rows_device(self, opts: Opts, n_ranks: int, pad_rows: int, gathered_ptr: int, full_ptr: int, stream: int = 0):
        """Gathered [n_ranks][pad_rows][W][3] device buffer -> full [H][W][3] device buffer (one kernel)."""
        _check(_lib.rt_shard_place_rows_device(self._h, C.byref(opts), n_ranks, pad_rows, C.c_void_p(gathered_ptr),
                                               C.c_void_p(full_ptr), C.c_void_p(stream)), "rt_shard_place_rows_device")

    def render_device(self, opts: Opts, device_ptr: int, stream: int = 0, stats: Stats | None = None):
        """Render into a device buffer (e.g. a torch tensor's data_ptr()) on a HIP stream."""
        _check(_lib.rt_render_hip_device(self._h, C.byref(opts), C.c_void_p(device_ptr), C.c_void_p(stream),
                                         C.byref(stats) if stats is not None else None), "rt_render_hip_device")

    def table_info(self) -> TableInfo:
        """The device tables the host builds for this scene (no GPU needed)."""
        t = TableInfo()
        _check(_lib.rt_scene_table_info(self._h, C.byref(t)), "rt_scene_table_info")
        return t

    def table_image(self) -> np.ndarray:
        """The packed device image as float32 records [n][4] (view it as uint32 / uint16 for the index tables)."""
        n = _lib.rt_scene_table_image(self._h, None, 0)
        if n < 0:
            raise RtmiError(-n, "rt_scene_table_image")
        out = np.empty(n, dtype=np.float32)
        _lib.rt_scene_table_image(self._h, out.ctypes.data_as(C.c_void_p), n)
        return out.reshape(-1, 4)

    def count(self, opts: Opts | None = None, want_image=False):
        """Diagnostic launch with exact event counters (roofline flops accounting)."""
        opts = opts or Opts()
        st = Stats()
        out = None
        ptr = None
        if want_image:
            out = np.empty((self.shard_rows(opts), self.width, 3), dtype=np.float32)
            ptr = out.ctypes.data_as(C.c_void_p)
        _check(_lib.rt_render_hip_count(self._h, C.byref(opts), ptr, C.byref(st)), "rt_render_hip_count")
        return (st, out) if want_image else st

    def accumulate(self, acc: np.ndarray | None = None, opts: Opts | None = None, stats: Stats | None = None,
                   want_image=True):
        """Progressive rendering: add the samples [opts.sample_first, +sample_count) to the exact pixel sums
        `acc` (int64 [local_rows, width, 3], 2^-24 units; None = start from zero).  Returns (acc, image)."""
        opts = opts or Opts()
        rows = self.shard_rows(opts)
        if acc is None:
            acc = np.zeros((rows, self.width, 3), dtype=np.int64)
        if acc.dtype != np.int64 or acc.shape != (rows, self.width, 3) or not acc.flags.c_contiguous:
            raise ValueError(f"acc must be a C-contiguous int64 array of shape {(rows, self.width, 3)}")
        out = np.empty((rows, self.width, 3), dtype=np.float32) if want_image else None
        _check(_lib.rt_render_hip_accumulate(self._h, C.byref(opts), acc.ctypes.data_as(C.c_void_p),
                                             out.ctypes.data_as(C.c_void_p) if want_image else None,
                                             C.byref(stats) if stats is not None else None),
               "rt_render_hip_accumulate")
        return acc, out

    def scatter_rows(self, opts: Opts, local: np.ndarray, full: np.ndarray):
        local = np.ascontiguousarray(local, dtype=np.float32)
        assert full.dtype == np.float32 and full.flags.c_contiguous
        _check(_lib.rt_shard_scatter_rows(self._h, C.byref(opts), local.ctypes.data_as(C.c_void_p),
                                          full.ctypes.data_as(C.c_void_p)), "rt_shard_scatter_rows")


def acc_to_rgb(acc: np.ndarray) -> np.ndarray:
    """fp32 framebuffer values of exact int64 pixel sums (2^-24 units)."""
    acc = np.ascontiguousarray(acc, dtype=np.int64)
    out = np.empty(acc.shape, dtype=np.float32)
    _lib.rt_acc_to_rgb(acc.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), acc.size)
    return out


def _guess_status() -> int:
    msg = _lib.rt_last_error().decode(errors="replace")
    if msg.startswith("JSON error"):
        return 3
    if msg.startswith("cannot open"):
        return 2
    return 4


def parse_scene(filename: str) -> Scene:
    """gpu-version/parser.hpp:504 ``parse_scene(filename)``."""
    return Scene.load(filename)


def output_image(image: np.ndarray, samples_per_pixel: int, filename: str = "main.ppm"):
    """gpu-version/main.cu:359-372 ``output_image``: P3 PPM, gamma 2, rows top to bottom."""
    img = np.ascontiguousarray(image, dtype=np.float32)
    h, w = img.shape[0], img.shape[1]
    _check(_lib.rt_write_ppm(os.fsencode(filename), img.ctypes.data_as(C.c_void_p), w, h, samples_per_pixel),
           "rt_write_ppm")


def write_image(image: np.ndarray, samples_per_pixel: int, filename: str = "main.png", gamma: bool = False):
    """gpu-version/color.cuh:15-35 ``write_image``: 8-bit RGB PNG of the linear means."""
    img = np.ascontiguousarray(image, dtype=np.float32)
    h, w = img.shape[0], img.shape[1]
    _check(_lib.rt_write_png(os.fsencode(filename), img.ctypes.data_as(C.c_void_p), w, h, samples_per_pixel, int(gamma)),
           "rt_write_png")


def quantize_rgb8(image: np.ndarray, samples_per_pixel: int, gamma: bool = True) -> np.ndarray:
    img = np.ascontiguousarray(image, dtype=np.float32)
    h, w = img.shape[0], img.shape[1]
    out = np.empty((h, w, 3), dtype=np.uint8)
    _check(_lib.rt_quantize_rgb8(img.ctypes.data_as(C.c_void_p), w, h, samples_per_pixel, int(gamma),
                                 out.ctypes.data_as(C.c_void_p)), "rt_quantize_rgb8")
    return out


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    _lib.rt_philox4x32_10(c, k, o)
    return list(o)


def sample_stream(seed, pixel, sample, n):
    out = (C.c_uint32 * n)()
    _lib.rt_sample_stream(seed, pixel, sample, out, n)
    return list(out)


def aabb_hit(bmin, bmax, orig, direction, t_min, t_max) -> bool:
    return bool(_lib.rt_aabb_hit(_v3(bmin), _v3(bmax), _v3(orig), _v3(direction), t_min, t_max))


def device_count() -> int:
    return _check_id(_lib.rt_device_count(), "rt_device_count")


def struct_size(which: int) -> int:
    return int(_lib.rt_struct_size(which))


def abi_version() -> int:
    return _lib.rt_abi_version()


def has_ablations() -> bool:
    """Does the loaded library carry the measurement variants and the counting kernels (the default build)?"""
    return bool(_lib.rt_has_ablations())


LIB_PATH = _LIB_PATH
