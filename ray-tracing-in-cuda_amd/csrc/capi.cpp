// C ABI glue (include/rtmi.h): scene I/O and building, table read-back, PPM output.
// The render entry points live in render_host.hip.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <new>
#include <sstream>
#include <string>

#include "philox.h"
#include "scene.hpp"

using namespace rtmi;

namespace {

rt_scene *finish(rt_scene *s, int rc) {
    if (rc != RT_OK) {
        delete s;
        return nullptr;
    }
    return s;
}

bool bad_scene(const rt_scene *s, const char *fn) {
    if (!s) {
        set_error("%s: null scene", fn);
        return true;
    }
    return false;
}

}  // namespace

extern "C" {

int rt_abi_version(void) { return RTMI_ABI_VERSION; }

size_t rt_struct_size(int which) {
    switch (which) {
    case 0: return sizeof(rt_opts);
    case 1: return sizeof(rt_stats);
    case 2: return sizeof(rt_prim);
    case 3: return sizeof(rt_material);
    case 4: return sizeof(rt_texture);
    case 5: return sizeof(rt_camera);
    case 6: return sizeof(rt_scene_info);
    case 7: return sizeof(rt_table_info);
    default: return 0;
    }
}

const char *rt_last_error(void) { return get_error(); }

const char *rt_status_string(int status) {
    switch (status) {
    case RT_OK: return "ok";
    case RT_ERR_ARG: return "invalid argument";
    case RT_ERR_IO: return "I/O error";
    case RT_ERR_JSON: return "malformed JSON";
    case RT_ERR_SCENE: return "invalid scene";
    case RT_ERR_HIP: return "HIP runtime or device error";
    case RT_ERR_LIMIT: return "scene exceeds a kernel limit";
    default: return "unknown status";
    }
}

void rt_opts_default(rt_opts *o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->seed = 2023;
    o->tile_rows = 8;
    o->tile_stride = 1;
}

// ---- scene I/O ------------------------------------------------------------------
rt_scene *rt_scene_parse_json(const char *text, size_t len) {
    if (!text) {
        set_error("rt_scene_parse_json: null text");
        return nullptr;
    }
    rt_scene *s = new (std::nothrow) rt_scene();
    if (!s) {
        set_error("out of memory");
        return nullptr;
    }
    return finish(s, scene_from_json(text, len, s->s));
}

rt_scene *rt_scene_load_json(const char *path) {
    if (!path) {
        set_error("rt_scene_load_json: null path");
        return nullptr;
    }
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        set_error("cannot open scene file '%s'", path);
        return nullptr;
    }
    std::stringstream ss;
    ss << f.rdbuf();
    std::string text = ss.str();
    rt_scene *s = new (std::nothrow) rt_scene();
    if (!s) {
        set_error("out of memory");
        return nullptr;
    }
    // "file" entries of image textures and meshes are relative to the scene file
    std::string dir(path);
    const size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
    return finish(s, scene_from_json(text.data(), text.size(), s->s, dir.c_str()));
}

rt_scene *rt_scene_rtiow(uint32_t seed, int width, int height, int spp, int max_depth) {
    rt_scene *s = new (std::nothrow) rt_scene();
    if (!s) {
        set_error("out of memory");
        return nullptr;
    }
    scene_rtiow(s->s, seed, width, height, spp, max_depth);
    return finish(s, scene_validate(s->s));
}

size_t rt_scene_to_json(const rt_scene *s, char *out, size_t cap) {
    if (bad_scene(s, "rt_scene_to_json")) return 0;
    std::string j = scene_to_json(s->s);
    if (out && cap) {
        size_t n = j.size() < cap - 1 ? j.size() : cap - 1;
        memcpy(out, j.data(), n);
        out[n] = 0;
    }
    return j.size() + 1;
}

void rt_scene_free(rt_scene *s) { delete s; }

// ---- building -------------------------------------------------------------------
rt_scene *rt_scene_new(int width, int height, int spp, int max_depth) {
    rt_scene *s = new (std::nothrow) rt_scene();
    if (!s) {
        set_error("out of memory");
        return nullptr;
    }
    s->s.width = width, s->s.height = height, s->s.spp = spp, s->s.max_depth = max_depth;
    s->s.flags = RT_FLAG_SKY_GRADIENT | RT_FLAG_DEFOCUS_BLUR;  // cmake-cpu-version semantics
    return s;
}

int rt_scene_set_russian_roulette(rt_scene *s, float p) {
    if (bad_scene(s, "rt_scene_set_russian_roulette")) return RT_ERR_ARG;
    if (!(p >= 0.0f && p <= 1.0f)) {
        set_error("rt_scene_set_russian_roulette: p = %g is not a probability", (double)p);
        return RT_ERR_ARG;
    }
    s->s.rr_p = p;
    s->s.touch();
    return RT_OK;
}

int rt_scene_set_background(rt_scene *s, const float rgb[3], uint32_t flags) {
    if (bad_scene(s, "rt_scene_set_background")) return RT_ERR_ARG;
    if (rgb) memcpy(s->s.background, rgb, 3 * sizeof(float));
    s->s.flags = flags & (RT_FLAG_SKY_GRADIENT | RT_FLAG_DEFOCUS_BLUR);
    s->s.touch();
    return RT_OK;
}

int rt_scene_set_camera(rt_scene *s, const float lookfrom[3], const float lookat[3], const float vup[3], float vfov,
                        float aspect, float aperture, float focus_dist) {
    if (bad_scene(s, "rt_scene_set_camera")) return RT_ERR_ARG;
    if (!lookfrom || !lookat || !vup) {
        set_error("rt_scene_set_camera: null vector");
        return RT_ERR_ARG;
    }
    CameraParams &c = s->s.cam;
    for (int i = 0; i < 3; ++i) c.lookfrom[i] = lookfrom[i], c.lookat[i] = lookat[i], c.vup[i] = vup[i];
    c.vfov = vfov;
    c.aspect = aspect > 0 ? (double)aspect : 0.0;
    c.aperture = aperture;
    c.focus_dist = focus_dist > 0 ? (double)focus_dist : 0.0;
    s->s.touch();
    return RT_OK;
}

static int add_texture(rt_scene *s, int type, const float a[3], const float b[3]) {
    if (bad_scene(s, "rt_scene_add_texture")) return -RT_ERR_ARG;
    if (!a || !b) {
        set_error("texture colour is null");
        return -RT_ERR_ARG;
    }
    rt_texture t;
    memset(&t, 0, sizeof t);
    t.type = type;
    memcpy(t.c0, a, sizeof t.c0);
    memcpy(t.c1, b, sizeof t.c1);
    s->s.texs.push_back(t);
    s->s.touch();
    return (int)s->s.texs.size() - 1;
}
int rt_scene_add_solid_color(rt_scene *s, const float rgb[3]) { return add_texture(s, RT_TEX_SOLID, rgb, rgb); }
int rt_scene_add_checker(rt_scene *s, const float even[3], const float odd[3]) {
    return add_texture(s, RT_TEX_CHECKER, even, odd);
}

static int add_material(rt_scene *s, int type, int tex, const float *albedo, float fuzz, float ir) {
    if (bad_scene(s, "rt_scene_add_material")) return -RT_ERR_ARG;
    if ((type == RT_MAT_LAMBERTIAN || type == RT_MAT_DIFFUSE_LIGHT) && (tex < 0 || tex >= (int)s->s.texs.size())) {
        set_error("material references texture %d (have %zu)", tex, s->s.texs.size());
        return -RT_ERR_SCENE;
    }
    rt_material m;
    memset(&m, 0, sizeof m);
    m.type = type;
    m.texture = tex;
    if (albedo) memcpy(m.albedo, albedo, sizeof m.albedo);
    m.fuzz = fuzz < 1 ? fuzz : 1;  // material.cuh:61
    m.ir = ir;
    s->s.mats.push_back(m);
    s->s.touch();
    return (int)s->s.mats.size() - 1;
}
int rt_scene_add_lambertian(rt_scene *s, int texture) { return add_material(s, RT_MAT_LAMBERTIAN, texture, nullptr, 0, 0); }
int rt_scene_add_metal(rt_scene *s, const float albedo[3], float fuzz) {
    if (!albedo) {
        set_error("metal albedo is null");
        return -RT_ERR_ARG;
    }
    return add_material(s, RT_MAT_METAL, -1, albedo, fuzz, 0);
}
int rt_scene_add_dielectric(rt_scene *s, float ir) { return add_material(s, RT_MAT_DIELECTRIC, -1, nullptr, 0, ir); }
int rt_scene_add_diffuse_light(rt_scene *s, int texture) {
    return add_material(s, RT_MAT_DIFFUSE_LIGHT, texture, nullptr, 0, 0);
}

static bool bad_material(rt_scene *s, int material) {
    if (material < 0 || material >= (int)s->s.mats.size()) {
        set_error("object references material %d (have %zu)", material, s->s.mats.size());
        return true;
    }
    return false;
}

int rt_scene_add_sphere(rt_scene *s, const float center[3], float radius, int material) {
    if (bad_scene(s, "rt_scene_add_sphere")) return -RT_ERR_ARG;
    if (!center) {
        set_error("sphere center is null");
        return -RT_ERR_ARG;
    }
    if (bad_material(s, material)) return -RT_ERR_SCENE;
    if (radius == 0.0f) {
        set_error("sphere radius is zero");
        return -RT_ERR_SCENE;
    }
    rt_prim p;
    memset(&p, 0, sizeof p);
    p.type = RT_PRIM_SPHERE, p.material = material;
    p.f[0] = center[0], p.f[1] = center[1], p.f[2] = center[2], p.f[3] = radius;
    s->s.prims.push_back(p);
    s->s.xforms.emplace_back();
    s->s.touch();
    return (int)s->s.prims.size() - 1;
}

int rt_scene_add_image_texture(rt_scene *s, int rows, int cols, const uint8_t *rgb) {
    if (bad_scene(s, "rt_scene_add_image_texture")) return -RT_ERR_ARG;
    return add_image_texture(s->s, rows, cols, rgb, "");
}

int rt_scene_add_image_texture_file(rt_scene *s, const char *path) {
    if (bad_scene(s, "rt_scene_add_image_texture_file")) return -RT_ERR_ARG;
    if (!path) {
        set_error("rt_scene_add_image_texture_file: null path");
        return -RT_ERR_ARG;
    }
    return add_image_texture_file(s->s, path);
}

int rt_scene_get_image(const rt_scene *s, int texture, int *rows, int *cols, uint8_t *out, size_t cap) {
    if (bad_scene(s, "rt_scene_get_image")) return -RT_ERR_ARG;
    if (texture < 0 || texture >= (int)s->s.texs.size() || s->s.texs[texture].type != RT_TEX_IMAGE) {
        set_error("texture %d is not an image texture", texture);
        return -RT_ERR_ARG;
    }
    const SceneImage &im = s->s.images[(size_t)s->s.texs[texture].c0[0]];
    if (rows) *rows = im.rows;
    if (cols) *cols = im.cols;
    if (out) {
        if (cap < im.rgb.size()) {
            set_error("rt_scene_get_image: buffer of %zu bytes, image needs %zu", cap, im.rgb.size());
            return -RT_ERR_ARG;
        }
        memcpy(out, im.rgb.data(), im.rgb.size());
    }
    return RT_OK;
}

int rt_scene_add_triangle(rt_scene *s, const float v1[3], const float v2[3], const float v3[3], const float uv1[2],
                          const float uv2[2], const float uv3[2], int material) {
    if (bad_scene(s, "rt_scene_add_triangle")) return -RT_ERR_ARG;
    if (!v1 || !v2 || !v3) {
        set_error("triangle vertex is null");
        return -RT_ERR_ARG;
    }
    if (bad_material(s, material)) return -RT_ERR_SCENE;
    return add_triangle(s->s, v1, v2, v3, uv1, uv2, uv3, material);
}

int rt_scene_add_obj(rt_scene *s, const char *path, int material, float scale, const float matrix[9],
                     const float translate[3]) {
    if (bad_scene(s, "rt_scene_add_obj")) return -RT_ERR_ARG;
    if (!path) {
        set_error("rt_scene_add_obj: null path");
        return -RT_ERR_ARG;
    }
    if (bad_material(s, material)) return -RT_ERR_SCENE;
    return add_obj(s->s, path, material, scale, matrix, translate);
}

int rt_scene_add_rect(rt_scene *s, int axis, float a0, float a1, float b0, float b1, float k, int material) {
    if (bad_scene(s, "rt_scene_add_rect")) return -RT_ERR_ARG;
    if (axis < 0 || axis > 2) {
        set_error("rect axis must be 0 (xy), 1 (xz) or 2 (yz)");
        return -RT_ERR_ARG;
    }
    if (bad_material(s, material)) return -RT_ERR_SCENE;
    rt_prim p;
    memset(&p, 0, sizeof p);
    p.type = RT_PRIM_XY_RECT + axis, p.material = material;
    p.f[0] = a0, p.f[1] = a1, p.f[2] = b0, p.f[3] = b1, p.f[4] = k;
    s->s.prims.push_back(p);
    s->s.xforms.emplace_back();
    s->s.touch();
    return (int)s->s.prims.size() - 1;
}

int rt_scene_add_cylinder(rt_scene *s, float radius, float zmin, float zmax, int material, const float rot_axis[3],
                          float rot_degrees, const float translate[3]) {
    if (bad_scene(s, "rt_scene_add_cylinder")) return -RT_ERR_ARG;
    if (bad_material(s, material)) return -RT_ERR_SCENE;
    double ax[3], off[3];
    if (rot_axis)
        for (int i = 0; i < 3; ++i) ax[i] = rot_axis[i];
    if (translate)
        for (int i = 0; i < 3; ++i) off[i] = translate[i];
    return add_cylinder(s->s, radius, zmin, zmax, material, rot_axis ? ax : nullptr, rot_degrees,
                        translate ? off : nullptr);
}

// ---- animation ---------------------------------------------------------------------------
rt_scene *rt_scene_clone(const rt_scene *src) {
    if (bad_scene(src, "rt_scene_clone")) return nullptr;
    rt_scene *s = new (std::nothrow) rt_scene();
    if (!s) {
        set_error("out of memory");
        return nullptr;
    }
    s->s = src->s;
    s->s.dev.reset();  // device residency is per scene object
    s->s.touch();
    return s;
}

int rt_scene_set_output_file(rt_scene *s, const char *path) {
    if (bad_scene(s, "rt_scene_set_output_file") || !path) return RT_ERR_ARG;
    s->s.output_file = path;
    return RT_OK;
}

int rt_scene_rotate_cylinders(rt_scene *s, double degrees) {
    if (bad_scene(s, "rt_scene_rotate_cylinders")) return -RT_ERR_ARG;
    Scene &sc = s->s;
    int changed = 0;
    for (size_t i = 0; i < sc.prims.size(); ++i) {
        if (sc.prims[i].type != RT_PRIM_CYLINDER || !sc.xforms[i].has_rotate) continue;
        CylinderXform xf = sc.xforms[i];
        xf.degrees += degrees;
        // rebuild through the same path the parser uses (parser.hpp:423-440)
        Scene tmp;
        tmp.mats.resize(sc.mats.size());
        int rc = add_cylinder(tmp, sc.prims[i].f[0], sc.prims[i].f[1], sc.prims[i].f[2], sc.prims[i].material, xf.axis,
                              xf.degrees, xf.has_translate ? xf.offset : nullptr);
        if (rc < 0) return rc;
        sc.prims[i] = tmp.prims[0];
        sc.xforms[i] = tmp.xforms[0];
        ++changed;
    }
    sc.touch();
    return changed;
}

rt_scene *rt_scene_dna(const rt_scene *base, double angle) {
    rt_scene *s = new (std::nothrow) rt_scene();
    if (!s) {
        set_error("out of memory");
        return nullptr;
    }
    Scene &sc = s->s;
    if (base) {
        sc = base->s;
        sc.dev.reset();
        sc.prims.clear(), sc.xforms.clear(), sc.mats.clear(), sc.texs.clear();
    } else {  // gpu-version/basic_scene.json
        sc.width = 1600, sc.height = 900, sc.spp = 100, sc.max_depth = 50;
        sc.background[0] = 0.0005f, sc.background[1] = 0.0007f, sc.background[2] = 0.00099f;
        sc.flags = 0;  // constant background, no lens sampling: what gpu-version renders from that file
        sc.cam = CameraParams();
        sc.cam.lookfrom[0] = 0, sc.cam.lookfrom[1] = 0, sc.cam.lookfrom[2] = -20;
        sc.cam.vfov = 23, sc.cam.aperture = 0.1;
    }
    const double pi = std::acos(-1.0);
    const int num_object = 5;
    const double space = 5;
    // dna.py:29-53: three solid colours and three diffuse_light materials per i in range(-15, 15)
    for (int i = 0; i < num_object * 6; ++i) {
        const double cols[3][3] = {{232 / 256.0, 209 / 256.0, 209 / 256.0},
                                   {232 / 256.0, 209 / 256.0, 209 / 256.0},
                                   {202 / 256.0, 202 / 256.0, 224 / 256.0}};
        for (int k = 0; k < 3; ++k) {
            rt_texture t;
            memset(&t, 0, sizeof t);
            t.type = RT_TEX_SOLID;
            for (int c = 0; c < 3; ++c) t.c0[c] = t.c1[c] = (float)cols[k][c];
            sc.texs.push_back(t);
        }
        for (int k = 0; k < 3; ++k) {
            rt_material m;
            memset(&m, 0, sizeof m);
            m.type = RT_MAT_DIFFUSE_LIGHT;
            m.texture = i * 3 + k;
            sc.mats.push_back(m);
        }
    }
    // dna.py:55-84
    for (int offset = 0; offset < 3; ++offset) {
        for (int i = 0; i < 2 * num_object; ++i) {
            const int id = i - num_object;
            const double theta = (36.0 * (id + num_object) + angle) / 180.0 * pi;
            const double xoffset = offset * space - space;
            const double zoffset = std::fabs(offset - 1.0) * -20 + 20;
            auto add_sphere = [&](double th, int mat) {
                rt_prim p;
                memset(&p, 0, sizeof p);
                p.type = RT_PRIM_SPHERE, p.material = mat;
                p.f[0] = (float)(2.5 * std::cos(th) + xoffset), p.f[1] = (float)id;
                p.f[2] = (float)(2.5 * std::sin(th) + zoffset), p.f[3] = 0.5f;
                sc.prims.push_back(p);
                sc.xforms.emplace_back();
            };
            add_sphere(theta, i * 3 + 0);
            add_sphere(theta + pi, i * 3 + 1);
            const double axis[3] = {0, 1, 0}, off[3] = {xoffset, (double)id, zoffset};
            int rc = add_cylinder(sc, 0.3f, -2.18f, 2.18f, i * 3 + 2, axis, 36.0 * -(id + num_object) + 90 + angle, off);
            if (rc < 0) return finish(s, RT_ERR_SCENE);
        }
    }
    sc.touch();
    return finish(s, scene_validate(sc));
}

int rt_scene_override(rt_scene *s, int width, int height, int spp, int max_depth) {
    if (bad_scene(s, "rt_scene_override")) return RT_ERR_ARG;
    int w = s->s.width, h = s->s.height, p = s->s.spp, d = s->s.max_depth;
    if (width > 0) s->s.width = width;
    if (height > 0) s->s.height = height;
    if (spp > 0) s->s.spp = spp;
    if (max_depth > 0) s->s.max_depth = max_depth;
    int rc = scene_validate(s->s);
    if (rc != RT_OK) s->s.width = w, s->s.height = h, s->s.spp = p, s->s.max_depth = d;
    s->s.touch();
    return rc;
}

// ---- read-back --------------------------------------------------------------------
int rt_scene_get_info(const rt_scene *s, rt_scene_info *out) {
    if (bad_scene(s, "rt_scene_get_info") || !out) return RT_ERR_ARG;
    out->width = s->s.width, out->height = s->s.height;
    out->samples_per_pixel = s->s.spp, out->max_depth = s->s.max_depth;
    out->num_prims = (int)s->s.prims.size();
    out->num_materials = (int)s->s.mats.size();
    out->num_textures = (int)s->s.texs.size();
    out->flags = s->s.flags;
    memcpy(out->background, s->s.background, sizeof out->background);
    out->russian_roulette = s->s.rr_p;
    return RT_OK;
}

int rt_scene_get_camera(const rt_scene *s, rt_camera *out) {
    if (bad_scene(s, "rt_scene_get_camera") || !out) return RT_ERR_ARG;
    derive_camera(s->s, out);
    return RT_OK;
}

int rt_scene_get_prims(const rt_scene *s, rt_prim *out, int cap) {
    if (bad_scene(s, "rt_scene_get_prims")) return -RT_ERR_ARG;
    int n = (int)s->s.prims.size();
    if (out)
        for (int i = 0; i < n && i < cap; ++i) out[i] = s->s.prims[i];
    return n;
}
int rt_scene_get_materials(const rt_scene *s, rt_material *out, int cap) {
    if (bad_scene(s, "rt_scene_get_materials")) return -RT_ERR_ARG;
    int n = (int)s->s.mats.size();
    if (out)
        for (int i = 0; i < n && i < cap; ++i) out[i] = s->s.mats[i];
    return n;
}
int rt_scene_get_textures(const rt_scene *s, rt_texture *out, int cap) {
    if (bad_scene(s, "rt_scene_get_textures")) return -RT_ERR_ARG;
    int n = (int)s->s.texs.size();
    if (out)
        for (int i = 0; i < n && i < cap; ++i) out[i] = s->s.texs[i];
    return n;
}

const char *rt_scene_output_file(const rt_scene *s) { return s ? s->s.output_file.c_str() : ""; }

// ---- output -------------------------------------------------------------------------
// write_color(FILE*, color, spp), gpu-version/color.cuh:70-95: fp32 throughout
static inline int quantize(float sum, int spp, int gamma) {
    float v;
    if (gamma) {
        float scale = 1.0f / (float)spp;
        v = std::sqrt(sum * scale);
    } else {
        v = sum / (float)spp;  // write_image, color.cuh:24-26
    }
    if (!(v == v)) return 0;  // NaN: the reference's cast is undefined; write black
    if (v < 0.0f) v = 0.0f;
    if (v > 0.999f) v = 0.999f;
    return (int)(256.0f * v);
}

void rt_acc_to_rgb(const int64_t *acc, float *rgb_sum, size_t n_values) {
    if (!acc || !rgb_sum) return;
    // same expression as the device's finalize step (render_kernel.hip): exact in double, one rounding to float
    for (size_t i = 0; i < n_values; ++i) rgb_sum[i] = (float)((double)acc[i] * (1.0 / 16777216.0));
}

int rt_quantize_rgb8(const float *rgb_sum, int width, int height, int spp, int gamma, uint8_t *out) {
    if (!rgb_sum || !out || width <= 0 || height <= 0 || spp <= 0) {
        set_error("rt_quantize_rgb8: bad argument");
        return RT_ERR_ARG;
    }
    size_t k = 0;
    for (int j = height - 1; j >= 0; --j)
        for (int i = 0; i < width; ++i)
            for (int c = 0; c < 3; ++c) out[k++] = (uint8_t)quantize(rgb_sum[((size_t)j * width + i) * 3 + c], spp, gamma);
    return RT_OK;
}

int rt_write_ppm(const char *path, const float *rgb_sum, int width, int height, int spp) {
    if (!path || !rgb_sum || width <= 0 || height <= 0 || spp <= 0) {
        set_error("rt_write_ppm: bad argument");
        return RT_ERR_ARG;
    }
    FILE *fp = fopen(path, "w");
    if (!fp) {
        set_error("cannot open '%s' for writing", path);
        return RT_ERR_IO;
    }
    std::string buf;
    buf.reserve((size_t)width * 12 + 64);
    fprintf(fp, "P3\n%d %d\n255\n", width, height);  // main.cu:363
    char line[48];
    for (int j = height - 1; j >= 0; --j) {
        buf.clear();
        for (int i = 0; i < width; ++i) {
            const float *p = rgb_sum + ((size_t)j * width + i) * 3;
            int n = snprintf(line, sizeof line, "%d %d %d\n", quantize(p[0], spp, 1), quantize(p[1], spp, 1),
                             quantize(p[2], spp, 1));
            buf.append(line, (size_t)n);
        }
        if (fwrite(buf.data(), 1, buf.size(), fp) != buf.size()) {
            fclose(fp);
            set_error("short write to '%s'", path);
            return RT_ERR_IO;
        }
    }
    if (fclose(fp) != 0) {
        set_error("error closing '%s'", path);
        return RT_ERR_IO;
    }
    return RT_OK;
}

// write_image(), gpu-version/color.cuh:15-35: 8-bit RGB PNG of the LINEAR means (no gamma), rows top
// to bottom.  The reference encodes with stb_image_write; here a minimal encoder (zlib "stored"
// blocks, no compression) keeps the library dependency-free.  gamma != 0 applies write_color's sqrt.
namespace {
uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return crc;
}
void put_be32(std::string &o, uint32_t v) {
    o.push_back((char)(v >> 24)), o.push_back((char)(v >> 16)), o.push_back((char)(v >> 8)), o.push_back((char)v);
}
void put_chunk(std::string &png, const char type[4], const std::string &data) {
    put_be32(png, (uint32_t)data.size());
    std::string body(type, 4);
    body += data;
    png += body;
    put_be32(png, crc32_update(0xFFFFFFFFu, (const uint8_t *)body.data(), body.size()) ^ 0xFFFFFFFFu);
}
}  // namespace

int rt_write_png(const char *path, const float *rgb_sum, int width, int height, int spp, int gamma) {
    if (!path || !rgb_sum || width <= 0 || height <= 0 || spp <= 0) {
        set_error("rt_write_png: bad argument");
        return RT_ERR_ARG;
    }
    // raw scanlines: filter byte 0 + RGB bytes, top row first
    std::string raw;
    raw.reserve((size_t)height * ((size_t)width * 3 + 1));
    for (int j = height - 1; j >= 0; --j) {
        raw.push_back(0);
        for (int i = 0; i < width; ++i)
            for (int c = 0; c < 3; ++c) raw.push_back((char)quantize(rgb_sum[((size_t)j * width + i) * 3 + c], spp, gamma));
    }
    std::string z;
    z.push_back(0x78), z.push_back(0x01);  // zlib header, no preset dictionary
    uint32_t a = 1, b = 0;                 // adler32
    for (unsigned char ch : raw) a = (a + ch) % 65521u, b = (b + a) % 65521u;
    size_t pos = 0;
    do {
        size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        z.push_back(pos + n == raw.size() ? 1 : 0);  // BFINAL, BTYPE = 00 (stored)
        z.push_back((char)(n & 0xFF)), z.push_back((char)(n >> 8));
        z.push_back((char)(~n & 0xFF)), z.push_back((char)((~n >> 8) & 0xFF));
        z.append(raw, pos, n);
        pos += n;
    } while (pos < raw.size());
    put_be32(z, (b << 16) | a);
    std::string png("\x89PNG\r\n\x1a\n", 8);
    std::string ihdr;
    put_be32(ihdr, (uint32_t)width), put_be32(ihdr, (uint32_t)height);
    ihdr.push_back(8), ihdr.push_back(2), ihdr.push_back(0), ihdr.push_back(0), ihdr.push_back(0);  // 8-bit RGB
    put_chunk(png, "IHDR", ihdr);
    put_chunk(png, "IDAT", z);
    put_chunk(png, "IEND", std::string());
    FILE *fp = fopen(path, "wb");
    if (!fp) {
        set_error("cannot open '%s' for writing", path);
        return RT_ERR_IO;
    }
    bool ok = fwrite(png.data(), 1, png.size(), fp) == png.size();
    ok = (fclose(fp) == 0) && ok;
    if (!ok) {
        set_error("short write to '%s'", path);
        return RT_ERR_IO;
    }
    return RT_OK;
}

// ---- misc ----------------------------------------------------------------------------
void rt_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    Philox4 p = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    for (int i = 0; i < 4; ++i) out[i] = p.v[i];
}

// first n raw words of the stream of one (pixel, sample): known-answer tests
void rt_sample_stream(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t *out, int n) {
    Xor128 g = xor128_seed(pixel, sample, (uint32_t)seed, (uint32_t)(seed >> 32));
    for (int i = 0; i < n; ++i) out[i] = xor128_next(g);
}

// aabb::hit, gpu-version/aabb.hpp:15-29
int rt_aabb_hit(const float bmin[3], const float bmax[3], const float orig[3], const float dir[3], float t_min_f,
                float t_max_f) {
    double t_min = t_min_f, t_max = t_max_f;
    for (int a = 0; a < 3; ++a) {
        float invD = 1.0f / dir[a];
        float t0 = (bmin[a] - orig[a]) * invD;
        float t1 = (bmax[a] - orig[a]) * invD;
        if (invD < 0.0f) {
            float tmp = t0;
            t0 = t1;
            t1 = tmp;
        }
        t_min = t0 > t_min ? t0 : t_min;
        t_max = t1 < t_max ? t1 : t_max;
        if (t_max <= t_min) return 0;
    }
    return 1;
}

}  // extern "C"
