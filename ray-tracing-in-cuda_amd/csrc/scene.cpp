// Scene construction: JSON -> tables, RTIOW generator, camera derivation, JSON out.
#include "scene.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "json.hpp"
#include "philox.h"

namespace rtmi {

// ---------------------------------------------------------------- errors
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *get_error() { return g_err; }

// ---------------------------------------------------------------- camera
// camera::camera, gpu-version/camera.cuh:9-29 / cmake-cpu-version/camera.h:9-31.
// Evaluated in fp64 from the stored parameters and rounded to fp32 once, so the
// kernel, the fp32 checker and the fp64 reference build all start from one frame.
static inline void v3sub(const double *a, const double *b, double *o) {
    o[0] = a[0] - b[0], o[1] = a[1] - b[1], o[2] = a[2] - b[2];
}
static inline void v3cross(const double *a, const double *b, double *o) {
    double x = a[1] * b[2] - a[2] * b[1];
    double y = a[2] * b[0] - a[0] * b[2];
    double z = a[0] * b[1] - a[1] * b[0];
    o[0] = x, o[1] = y, o[2] = z;
}
static inline double v3len(const double *a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
static inline void v3unit(double *a) {
    double inv = 1.0 / v3len(a);
    a[0] *= inv, a[1] *= inv, a[2] *= inv;
}

void derive_camera(const Scene &s, rt_camera *out) {
    const CameraParams &c = s.cam;
    const double pi = std::acos(-1.0);
    double aspect = c.aspect > 0 ? c.aspect : (double)s.width / (double)s.height;
    double d[3];
    v3sub(c.lookfrom, c.lookat, d);
    double focus = c.focus_dist > 0 ? c.focus_dist : v3len(d);

    double theta = c.vfov * pi / 180.0;
    double h = std::tan(theta / 2);
    double vh = 2.0 * h, vw = aspect * vh;
    double w[3] = {d[0], d[1], d[2]}, u[3], v[3];
    v3unit(w);
    v3cross(c.vup, w, u);
    v3unit(u);
    v3cross(w, u, v);
    for (int i = 0; i < 3; ++i) {
        double hor = focus * vw * u[i];
        double ver = focus * vh * v[i];
        double llc = c.lookfrom[i] - hor / 2 - ver / 2 - focus * w[i];
        out->lookfrom[i] = (float)c.lookfrom[i];
        out->lookat[i] = (float)c.lookat[i];
        out->vup[i] = (float)c.vup[i];
        out->origin[i] = (float)c.lookfrom[i];
        out->horizontal[i] = (float)hor;
        out->vertical[i] = (float)ver;
        out->lower_left[i] = (float)llc;
        out->u[i] = (float)u[i];
        out->v[i] = (float)v[i];
        out->w[i] = (float)w[i];
    }
    out->vfov = (float)c.vfov;
    out->aspect = (float)aspect;
    out->aperture = (float)c.aperture;
    out->focus_dist = (float)focus;
    out->lens_radius = (float)(c.aperture / 2);
}

// ---------------------------------------------------------------- cylinders
// cylinder::rotate/translate (object.cuh:225-231) with rotate() = Rodrigues form of
// vec3.cuh:396-418 and translate() of vec3.cuh:384-394, composed as the parser
// does (parser.hpp:423-440: rotate first, then translate => o2w = T * R).
// m = [R | t], m_inv = [R^T | -R^T t]; built in fp64, rounded once.
int add_cylinder(Scene &s, float radius, float zmin, float zmax, int material, const double *axis,
                 double degrees, const double *offset) {
    rt_prim p;
    memset(&p, 0, sizeof p);
    p.type = RT_PRIM_CYLINDER;
    p.material = material;
    p.f[0] = radius, p.f[1] = zmin, p.f[2] = zmax;
    double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    double t[3] = {0, 0, 0};
    CylinderXform xf;
    if (axis) {
        double a[3] = {axis[0], axis[1], axis[2]};
        double len = v3len(a);
        if (!(len > 0)) {
            set_error("cylinder rotate.axis has zero length");
            return -RT_ERR_SCENE;
        }
        v3unit(a);
        const double pi = std::acos(-1.0);
        double th = degrees / 180.0 * pi;
        double sn = std::sin(th), cs = std::cos(th);
        R[0][0] = a[0] * a[0] + (1 - a[0] * a[0]) * cs;
        R[0][1] = a[0] * a[1] * (1 - cs) - a[2] * sn;
        R[0][2] = a[0] * a[2] * (1 - cs) + a[1] * sn;
        R[1][0] = a[0] * a[1] * (1 - cs) + a[2] * sn;
        R[1][1] = a[1] * a[1] + (1 - a[1] * a[1]) * cs;
        R[1][2] = a[1] * a[2] * (1 - cs) - a[0] * sn;
        R[2][0] = a[0] * a[2] * (1 - cs) - a[1] * sn;
        R[2][1] = a[1] * a[2] * (1 - cs) + a[0] * sn;
        R[2][2] = a[2] * a[2] + (1 - a[2] * a[2]) * cs;
        xf.has_rotate = true;
        xf.axis[0] = axis[0], xf.axis[1] = axis[1], xf.axis[2] = axis[2];
        xf.degrees = degrees;
    }
    if (offset) {
        t[0] = offset[0], t[1] = offset[1], t[2] = offset[2];
        xf.has_translate = true;
        xf.offset[0] = t[0], xf.offset[1] = t[1], xf.offset[2] = t[2];
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            p.m[i * 4 + j] = (float)R[i][j];
            p.m_inv[i * 4 + j] = (float)R[j][i];
        }
        p.m[i * 4 + 3] = (float)t[i];
        p.m_inv[i * 4 + 3] = (float)(-(R[0][i] * t[0] + R[1][i] * t[1] + R[2][i] * t[2]));
    }
    s.prims.push_back(p);
    s.xforms.push_back(xf);
    s.touch();
    return (int)s.prims.size() - 1;
}

// ---------------------------------------------------------------- validation
int scene_validate(const Scene &s) {
    if (s.width < 2 || s.height < 2) {
        // u = (x + xi) / (W - 1), main.cu:97-98: W = 1 divides by zero
        set_error("width and height must be >= 2 (got %dx%d)", s.width, s.height);
        return RT_ERR_SCENE;
    }
    if (s.width > 65536 || s.height > 65536) {
        set_error("image larger than 65536 on a side (%dx%d)", s.width, s.height);
        return RT_ERR_SCENE;
    }
    if (s.spp < 1 || s.max_depth < 0) {
        set_error("samples_per_pixel must be >= 1 and max_depth >= 0 (got %d, %d)", s.spp, s.max_depth);
        return RT_ERR_SCENE;
    }
    for (size_t i = 0; i < s.mats.size(); ++i) {
        const rt_material &m = s.mats[i];
        if (m.type == RT_MAT_LAMBERTIAN || m.type == RT_MAT_DIFFUSE_LIGHT) {
            if (m.texture < 0 || m.texture >= (int)s.texs.size()) {
                set_error("material %zu references texture %d (have %zu)", i, m.texture, s.texs.size());
                return RT_ERR_SCENE;
            }
        }
    }
    for (size_t i = 0; i < s.prims.size(); ++i) {
        const rt_prim &p = s.prims[i];
        if (p.material < 0 || p.material >= (int)s.mats.size()) {
            set_error("object %zu references material %d (have %zu)", i, p.material, s.mats.size());
            return RT_ERR_SCENE;
        }
        if (p.type == RT_PRIM_SPHERE && p.f[3] == 0.0f) {
            set_error("object %zu: sphere radius is zero", i);
            return RT_ERR_SCENE;
        }
    }
    double d[3];
    v3sub(s.cam.lookfrom, s.cam.lookat, d);
    if (!(v3len(d) > 0)) {
        set_error("camera lookfrom equals lookat");
        return RT_ERR_SCENE;
    }
    return RT_OK;
}

// ---------------------------------------------------------------- JSON in
namespace {

struct Reader {
    bool ok = true;
    void fail(const char *fmt, ...) __attribute__((format(printf, 2, 3))) {
        if (!ok) return;
        ok = false;
        char buf[400];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        set_error("%s", buf);
    }
    double num(const JsonValue &o, const char *key, const char *where) {
        const JsonValue *v = o.find(key);
        if (!v || !v->is_number()) {
            fail("%s: missing or non-numeric \"%s\"", where, key);
            return 0;
        }
        return v->num;
    }
    int integer(const JsonValue &o, const char *key, const char *where) {
        double d = num(o, key, where);
        if (ok && (d != std::floor(d) || std::fabs(d) > 1e9)) fail("%s: \"%s\" must be an integer", where, key);
        return (int)d;
    }
    void vec3(const JsonValue &o, const char *key, const char *where, double out[3]) {
        const JsonValue *v = o.find(key);
        if (!v || !v->is_array() || v->arr.size() != 3 || !v->arr[0].is_number() || !v->arr[1].is_number() ||
            !v->arr[2].is_number()) {
            fail("%s: \"%s\" must be an array of 3 numbers", where, key);
            out[0] = out[1] = out[2] = 0;
            return;
        }
        for (int i = 0; i < 3; ++i) out[i] = v->arr[i].num;
    }
    const JsonValue *data_array(const JsonValue &root, const char *key) {
        // {"object": {"data": [...]}} (parser.hpp:519-521)
        const JsonValue *o = root.find(key);
        if (!o || !o->is_object()) {
            fail("top level: missing object \"%s\"", key);
            return nullptr;
        }
        const JsonValue *d = o->find("data");
        if (!d || !d->is_array()) {
            fail("\"%s\": missing array \"data\"", key);
            return nullptr;
        }
        return d;
    }
};

}  // namespace

int scene_from_json(const char *text, size_t len, Scene &s) {
    JsonValue root;
    std::string perr;
    JsonParser parser(text, len);
    if (!parser.parse(root, perr)) {
        set_error("%s", perr.c_str());
        return RT_ERR_JSON;
    }
    if (!root.is_object()) {
        set_error("top level of a scene file must be an object");
        return RT_ERR_SCENE;
    }
    Reader r;
    char where[96];

    // scalars, parser.hpp:512-517
    double bg[3];
    r.vec3(root, "background", "top level", bg);
    s.background[0] = (float)bg[0], s.background[1] = (float)bg[1], s.background[2] = (float)bg[2];
    s.max_depth = r.integer(root, "max_depth", "top level");
    s.spp = r.integer(root, "samples_per_pixel", "top level");
    s.width = r.integer(root, "width", "top level");
    s.height = r.integer(root, "height", "top level");
    if (const JsonValue *of = root.find("output_file")) {
        if (!of->is_string()) r.fail("top level: \"output_file\" must be a string");
        else s.output_file = of->str;
    }
    // The JSON interface is gpu-version's, whose camera::get_ray has the lens sample commented out
    // (camera.cuh:33-34: rd = 0, no draws): a scene file that does not say otherwise renders without defocus
    // blur, like `parallel_compute -f scene.json`.  "defocus_blur": true selects cmake-cpu-version/camera.h:34.
    s.flags = 0;
    if (const JsonValue *f = root.find("sky_gradient")) {
        if (f->kind != JsonValue::Bool) r.fail("top level: \"sky_gradient\" must be a boolean");
        else if (f->b) s.flags |= RT_FLAG_SKY_GRADIENT;
    }
    if (const JsonValue *f = root.find("defocus_blur")) {
        if (f->kind != JsonValue::Bool) r.fail("top level: \"defocus_blur\" must be a boolean");
        else if (f->b) s.flags |= RT_FLAG_DEFOCUS_BLUR;
    }
    if (root.find("russian_roulette")) {
        const double p = r.num(root, "russian_roulette", "top level");
        if (!(p >= 0.0 && p <= 1.0)) r.fail("top level: \"russian_roulette\" must be a probability in [0, 1]");
        else s.rr_p = (float)p;
    }

    // camera, parser.hpp:113-141
    const JsonValue *cam = root.find("camera");
    if (!cam || !cam->is_object()) r.fail("top level: missing object \"camera\"");
    else {
        r.vec3(*cam, "lookfrom", "camera", s.cam.lookfrom);
        r.vec3(*cam, "lookat", "camera", s.cam.lookat);
        r.vec3(*cam, "vup", "camera", s.cam.vup);
        s.cam.vfov = r.num(*cam, "vfov", "camera");
        s.cam.aperture = r.num(*cam, "aperture", "camera");
        s.cam.aspect = 0.0;      // width / height (parser.hpp:122)
        s.cam.focus_dist = 0.0;  // |lookfrom - lookat| (parser.hpp:124)
        if (const JsonValue *fd = cam->find("focus_dist")) {
            if (fd->is_number() && fd->num > 0) s.cam.focus_dist = fd->num;
            else r.fail("camera: \"focus_dist\" must be a positive number");
        }
    }
    if (!r.ok) return RT_ERR_SCENE;

    // textures, parser.hpp:143-184 (+ checker, texture.cuh:33-57)
    if (const JsonValue *arr = r.data_array(root, "texture")) {
        for (size_t i = 0; i < arr->arr.size() && r.ok; ++i) {
            const JsonValue &t = arr->arr[i];
            snprintf(where, sizeof where, "texture %zu", i);
            const JsonValue *ty = t.find("type");
            if (!t.is_object() || !ty || !ty->is_string()) {
                r.fail("%s: missing string \"type\"", where);
                break;
            }
            rt_texture rec;
            memset(&rec, 0, sizeof rec);
            double c[3];
            if (ty->str == "solid_color") {
                rec.type = RT_TEX_SOLID;
                r.vec3(t, "color", where, c);
                for (int k = 0; k < 3; ++k) rec.c0[k] = rec.c1[k] = (float)c[k];
            } else if (ty->str == "checker") {
                rec.type = RT_TEX_CHECKER;
                r.vec3(t, "even", where, c);
                for (int k = 0; k < 3; ++k) rec.c0[k] = (float)c[k];
                r.vec3(t, "odd", where, c);
                for (int k = 0; k < 3; ++k) rec.c1[k] = (float)c[k];
            } else {
                r.fail("%s: unknown type \"%s\"", where, ty->str.c_str());
            }
            s.texs.push_back(rec);
        }
    }
    if (!r.ok) return RT_ERR_SCENE;

    // materials, parser.hpp:186-281
    if (const JsonValue *arr = r.data_array(root, "material")) {
        for (size_t i = 0; i < arr->arr.size() && r.ok; ++i) {
            const JsonValue &m = arr->arr[i];
            snprintf(where, sizeof where, "material %zu", i);
            const JsonValue *ty = m.find("type");
            if (!m.is_object() || !ty || !ty->is_string()) {
                r.fail("%s: missing string \"type\"", where);
                break;
            }
            rt_material rec;
            memset(&rec, 0, sizeof rec);
            rec.texture = -1;
            if (ty->str == "lambertian") {
                rec.type = RT_MAT_LAMBERTIAN;
                rec.texture = r.integer(m, "texture", where);
            } else if (ty->str == "metal") {
                rec.type = RT_MAT_METAL;
                double c[3];
                r.vec3(m, "albedo", where, c);
                for (int k = 0; k < 3; ++k) rec.albedo[k] = (float)c[k];
                float f = (float)r.num(m, "fuzz", where);
                rec.fuzz = f < 1 ? f : 1;  // material.cuh:61
            } else if (ty->str == "dielectric") {
                rec.type = RT_MAT_DIELECTRIC;
                rec.ir = (float)r.num(m, "index_of_refraction", where);
            } else if (ty->str == "diffuse_light") {
                rec.type = RT_MAT_DIFFUSE_LIGHT;
                rec.texture = r.integer(m, "texture", where);
            } else {
                r.fail("%s: unknown type \"%s\"", where, ty->str.c_str());
            }
            s.mats.push_back(rec);
        }
    }
    if (!r.ok) return RT_ERR_SCENE;

    // objects, parser.hpp:283-478
    if (const JsonValue *arr = r.data_array(root, "object")) {
        for (size_t i = 0; i < arr->arr.size() && r.ok; ++i) {
            const JsonValue &o = arr->arr[i];
            snprintf(where, sizeof where, "object %zu", i);
            const JsonValue *ty = o.find("type");
            if (!o.is_object() || !ty || !ty->is_string()) {
                r.fail("%s: missing string \"type\"", where);
                break;
            }
            rt_prim rec;
            memset(&rec, 0, sizeof rec);
            const std::string &t = ty->str;
            if (t == "sphere") {
                rec.type = RT_PRIM_SPHERE;
                double c[3];
                r.vec3(o, "center", where, c);
                rec.f[0] = (float)c[0], rec.f[1] = (float)c[1], rec.f[2] = (float)c[2];
                rec.f[3] = (float)r.num(o, "radius", where);
                rec.material = r.integer(o, "material", where);
                s.prims.push_back(rec);
                s.xforms.emplace_back();
            } else if (t == "xy_rect" || t == "xz_rect" || t == "yz_rect") {
                const char *k0, *k1, *k2, *k3;
                if (t == "xy_rect") rec.type = RT_PRIM_XY_RECT, k0 = "x0", k1 = "x1", k2 = "y0", k3 = "y1";
                else if (t == "xz_rect") rec.type = RT_PRIM_XZ_RECT, k0 = "x0", k1 = "x1", k2 = "z0", k3 = "z1";
                else rec.type = RT_PRIM_YZ_RECT, k0 = "y0", k1 = "y1", k2 = "z0", k3 = "z1";
                rec.f[0] = (float)r.num(o, k0, where);
                rec.f[1] = (float)r.num(o, k1, where);
                rec.f[2] = (float)r.num(o, k2, where);
                rec.f[3] = (float)r.num(o, k3, where);
                rec.f[4] = (float)r.num(o, "k", where);
                rec.material = r.integer(o, "material", where);
                s.prims.push_back(rec);
                s.xforms.emplace_back();
            } else if (t == "cylinder") {
                float radius = (float)r.num(o, "radius", where);
                float zmin = (float)r.num(o, "zmin", where);
                float zmax = (float)r.num(o, "zmax", where);
                int mat = r.integer(o, "material", where);
                double axis[3], off[3], deg = 0;
                const double *pa = nullptr, *po = nullptr;
                if (const JsonValue *rot = o.find("rotate")) {
                    if (!rot->is_object()) r.fail("%s: \"rotate\" must be an object", where);
                    else {
                        r.vec3(*rot, "axis", where, axis);
                        deg = r.num(*rot, "angle", where);
                        pa = axis;
                    }
                }
                if (o.find("translate")) {
                    r.vec3(o, "translate", where, off);
                    po = off;
                }
                if (r.ok) {
                    int rc = add_cylinder(s, radius, zmin, zmax, mat, pa, deg, po);
                    if (rc < 0) return RT_ERR_SCENE;
                }
            } else {
                r.fail("%s: unknown type \"%s\"", where, t.c_str());
            }
        }
    }
    if (!r.ok) return RT_ERR_SCENE;
    s.touch();
    return scene_validate(s);
}

// ---------------------------------------------------------------- JSON out
static void put_vec3(std::string &o, const float *v) {
    o += "[" + json_float(v[0]) + ", " + json_float(v[1]) + ", " + json_float(v[2]) + "]";
}
static std::string json_double(double d) {
    char buf[64];
    snprintf(buf, sizeof buf, "%.17g", d);
    return buf;
}
static void put_vec3d(std::string &o, const double *v) {
    o += "[" + json_double(v[0]) + ", " + json_double(v[1]) + ", " + json_double(v[2]) + "]";
}

std::string scene_to_json(const Scene &s) {
    std::string o = "{\n";
    o += "  \"output_file\": " + json_escape(s.output_file) + ",\n";
    o += "  \"background\": ";
    put_vec3(o, s.background);
    o += ",\n";
    o += "  \"max_depth\": " + std::to_string(s.max_depth) + ",\n";
    o += "  \"samples_per_pixel\": " + std::to_string(s.spp) + ",\n";
    o += "  \"width\": " + std::to_string(s.width) + ",\n";
    o += "  \"height\": " + std::to_string(s.height) + ",\n";
    o += std::string("  \"sky_gradient\": ") + ((s.flags & RT_FLAG_SKY_GRADIENT) ? "true" : "false") + ",\n";
    o += std::string("  \"defocus_blur\": ") + ((s.flags & RT_FLAG_DEFOCUS_BLUR) ? "true" : "false") + ",\n";
    if (s.rr_p > 0.0f) o += "  \"russian_roulette\": " + json_double((double)s.rr_p) + ",\n";
    o += "  \"camera\": {\"lookfrom\": ";
    put_vec3d(o, s.cam.lookfrom);
    o += ", \"lookat\": ";
    put_vec3d(o, s.cam.lookat);
    o += ", \"vup\": ";
    put_vec3d(o, s.cam.vup);
    o += ", \"vfov\": " + json_double(s.cam.vfov) + ", \"aperture\": " + json_double(s.cam.aperture);
    if (s.cam.focus_dist > 0) o += ", \"focus_dist\": " + json_double(s.cam.focus_dist);
    o += "},\n";

    o += "  \"object\": {\"data\": [";
    for (size_t i = 0; i < s.prims.size(); ++i) {
        const rt_prim &p = s.prims[i];
        o += i ? ",\n    " : "\n    ";
        switch (p.type) {
        case RT_PRIM_SPHERE:
            o += "{\"type\": \"sphere\", \"center\": ";
            put_vec3(o, p.f);
            o += ", \"radius\": " + json_float(p.f[3]);
            break;
        case RT_PRIM_XY_RECT:
        case RT_PRIM_XZ_RECT:
        case RT_PRIM_YZ_RECT: {
            const char *names[3][5] = {{"xy_rect", "x0", "x1", "y0", "y1"},
                                       {"xz_rect", "x0", "x1", "z0", "z1"},
                                       {"yz_rect", "y0", "y1", "z0", "z1"}};
            const char **n = names[p.type - RT_PRIM_XY_RECT];
            o += std::string("{\"type\": \"") + n[0] + "\"";
            for (int k = 0; k < 4; ++k) o += std::string(", \"") + n[k + 1] + "\": " + json_float(p.f[k]);
            o += ", \"k\": " + json_float(p.f[4]);
            break;
        }
        case RT_PRIM_CYLINDER: {
            const CylinderXform &xf = s.xforms[i];
            o += "{\"type\": \"cylinder\", \"radius\": " + json_float(p.f[0]) + ", \"zmin\": " + json_float(p.f[1]) +
                 ", \"zmax\": " + json_float(p.f[2]);
            if (xf.has_rotate) {
                o += ", \"rotate\": {\"axis\": ";
                put_vec3d(o, xf.axis);
                o += ", \"angle\": " + json_double(xf.degrees) + "}";
            }
            if (xf.has_translate) {
                o += ", \"translate\": ";
                put_vec3d(o, xf.offset);
            }
            break;
        }
        default: break;
        }
        o += ", \"material\": " + std::to_string(p.material) + "}";
    }
    o += "\n  ]},\n";

    o += "  \"material\": {\"data\": [";
    for (size_t i = 0; i < s.mats.size(); ++i) {
        const rt_material &m = s.mats[i];
        o += i ? ",\n    " : "\n    ";
        switch (m.type) {
        case RT_MAT_LAMBERTIAN: o += "{\"type\": \"lambertian\", \"texture\": " + std::to_string(m.texture) + "}"; break;
        case RT_MAT_METAL:
            o += "{\"type\": \"metal\", \"albedo\": ";
            put_vec3(o, m.albedo);
            o += ", \"fuzz\": " + json_float(m.fuzz) + "}";
            break;
        case RT_MAT_DIELECTRIC:
            o += "{\"type\": \"dielectric\", \"index_of_refraction\": " + json_float(m.ir) + "}";
            break;
        case RT_MAT_DIFFUSE_LIGHT:
            o += "{\"type\": \"diffuse_light\", \"texture\": " + std::to_string(m.texture) + "}";
            break;
        default: break;
        }
    }
    o += "\n  ]},\n";

    o += "  \"texture\": {\"data\": [";
    for (size_t i = 0; i < s.texs.size(); ++i) {
        const rt_texture &t = s.texs[i];
        o += i ? ",\n    " : "\n    ";
        if (t.type == RT_TEX_SOLID) {
            o += "{\"type\": \"solid_color\", \"color\": ";
            put_vec3(o, t.c0);
            o += "}";
        } else {
            o += "{\"type\": \"checker\", \"even\": ";
            put_vec3(o, t.c0);
            o += ", \"odd\": ";
            put_vec3(o, t.c1);
            o += "}";
        }
    }
    o += "\n  ]}\n}\n";
    return o;
}

// ---------------------------------------------------------------- RTIOW scene
// random_scene(), cmake-cpu-version/main.cpp:125-172 with the camera of :89-94.
// The reference draws from rand() seeded by srand(7) in an order that depends on
// the compiler (argument evaluation order is unspecified); here
// the draws come from Philox keyed by `seed` in a fixed order: choose_mat,
// center.x, center.z, then the material's draws, x before y before z.
namespace {
struct SceneRng {
    uint32_t k0, k1, block = 0, pos = 4;
    Philox4 buf;
    explicit SceneRng(uint32_t seed) : k0(seed), k1(0x52544957u /* "RTIW" */) {}
    double next() {  // [0,1), 24 bits
        if (pos == 4) {
            buf = philox4x32_10(block++, 0, 0, 0, k0, k1);
            pos = 0;
        }
        return (double)(buf.v[pos++] >> 8) * (1.0 / 16777216.0);
    }
    double range(double lo, double hi) { return lo + (hi - lo) * next(); }
};
}  // namespace

void scene_rtiow(Scene &s, uint32_t seed, int width, int height, int spp, int max_depth) {
    s = Scene();
    s.width = width, s.height = height, s.spp = spp, s.max_depth = max_depth;
    s.flags = RT_FLAG_SKY_GRADIENT | RT_FLAG_DEFOCUS_BLUR;
    s.background[0] = 0.5f, s.background[1] = 0.7f, s.background[2] = 1.0f;
    s.output_file = "rtiow.png";
    // camera, main.cpp:89-94
    s.cam.lookfrom[0] = 13, s.cam.lookfrom[1] = 2, s.cam.lookfrom[2] = 3;
    s.cam.lookat[0] = s.cam.lookat[1] = s.cam.lookat[2] = 0;
    s.cam.vup[0] = 0, s.cam.vup[1] = 1, s.cam.vup[2] = 0;
    s.cam.vfov = 20;
    s.cam.aperture = 0.1;
    s.cam.aspect = 0;
    s.cam.focus_dist = 0;

    auto add_tex_solid = [&](double r, double g, double b) {
        rt_texture t;
        memset(&t, 0, sizeof t);
        t.type = RT_TEX_SOLID;
        t.c0[0] = t.c1[0] = (float)r, t.c0[1] = t.c1[1] = (float)g, t.c0[2] = t.c1[2] = (float)b;
        s.texs.push_back(t);
        return (int)s.texs.size() - 1;
    };
    auto add_mat = [&](int type, int tex, double r, double g, double b, double fuzz, double ir) {
        rt_material m;
        memset(&m, 0, sizeof m);
        m.type = type, m.texture = tex;
        m.albedo[0] = (float)r, m.albedo[1] = (float)g, m.albedo[2] = (float)b;
        float f = (float)fuzz;
        m.fuzz = f < 1 ? f : 1;
        m.ir = (float)ir;
        s.mats.push_back(m);
        return (int)s.mats.size() - 1;
    };
    auto add_sphere = [&](double x, double y, double z, double rad, int mat) {
        rt_prim p;
        memset(&p, 0, sizeof p);
        p.type = RT_PRIM_SPHERE, p.material = mat;
        p.f[0] = (float)x, p.f[1] = (float)y, p.f[2] = (float)z, p.f[3] = (float)rad;
        s.prims.push_back(p);
        s.xforms.emplace_back();
    };

    // ground: checker_texture(color(0.2,0.3,0.1), color(0.9,0.9,0.9)), main.cpp:131-132
    {
        rt_texture t;
        memset(&t, 0, sizeof t);
        t.type = RT_TEX_CHECKER;
        t.c0[0] = 0.2f, t.c0[1] = 0.3f, t.c0[2] = 0.1f;  // even
        t.c1[0] = t.c1[1] = t.c1[2] = 0.9f;              // odd
        s.texs.push_back(t);
        add_sphere(0, -1000, 0, 1000, add_mat(RT_MAT_LAMBERTIAN, 0, 0, 0, 0, 0, 0));
    }
    SceneRng rng(seed);
    for (int a = -11; a < 11; ++a) {
        for (int b = -11; b < 11; ++b) {
            double choose = rng.next();
            double cx = a + 0.9 * rng.next();
            double cz = b + 0.9 * rng.next();
            double dx = cx - 4, dy = 0.2 - 0.2, dz = cz - 0;
            if (std::sqrt(dx * dx + dy * dy + dz * dz) > 0.9) {  // main.cpp:139
                if (choose < 0.8) {  // diffuse: albedo = random() * random()
                    double c1[3], c2[3];
                    for (double &c : c1) c = rng.next();
                    for (double &c : c2) c = rng.next();
                    int tex = add_tex_solid(c1[0] * c2[0], c1[1] * c2[1], c1[2] * c2[2]);
                    add_sphere(cx, 0.2, cz, 0.2, add_mat(RT_MAT_LAMBERTIAN, tex, 0, 0, 0, 0, 0));
                } else if (choose < 0.95) {  // metal: albedo = random(0.5,1), fuzz = random(0,0.5)
                    double c[3];
                    for (double &v : c) v = rng.range(0.5, 1);
                    double fuzz = rng.range(0, 0.5);
                    add_sphere(cx, 0.2, cz, 0.2, add_mat(RT_MAT_METAL, -1, c[0], c[1], c[2], fuzz, 0));
                } else {  // glass
                    add_sphere(cx, 0.2, cz, 0.2, add_mat(RT_MAT_DIELECTRIC, -1, 0, 0, 0, 0, 1.5));
                }
            }
        }
    }
    add_sphere(0, 1, 0, 1.0, add_mat(RT_MAT_DIELECTRIC, -1, 0, 0, 0, 0, 1.5));
    add_sphere(-4, 1, 0, 1.0, add_mat(RT_MAT_LAMBERTIAN, add_tex_solid(0.4, 0.2, 0.1), 0, 0, 0, 0, 0));
    add_sphere(4, 1, 0, 1.0, add_mat(RT_MAT_METAL, -1, 0.7, 0.6, 0.5, 0.0, 0));
    s.touch();
}

}  // namespace rtmi
