// Scene construction: JSON -> tables, RTIOW generator, camera derivation, JSON out.
#include "scene.hpp"

#include <array>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "json.hpp"
#include "png_read.hpp"
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
    for (size_t i = 0; i < s.texs.size(); ++i) {
        const rt_texture &t = s.texs[i];
        if (t.type == RT_TEX_IMAGE) {
            const int im = (int)t.c0[0];
            if (im < 0 || im >= (int)s.images.size() || s.images[im].rows != (int)t.c0[1] || s.images[im].cols != (int)t.c0[2] ||
                s.images[im].rgb.size() != (size_t)s.images[im].rows * s.images[im].cols * 3) {
                set_error("texture %zu references image %d, which is missing or has another size", i, im);
                return RT_ERR_SCENE;
            }
        } else if (t.type != RT_TEX_SOLID && t.type != RT_TEX_CHECKER) {
            set_error("texture %zu has unknown type %d", i, t.type);
            return RT_ERR_SCENE;
        }
    }
    for (size_t i = 0; i < s.prims.size(); ++i) {
        const rt_prim &p = s.prims[i];
        if (p.type < RT_PRIM_SPHERE || p.type > RT_PRIM_TRIANGLE) {
            set_error("object %zu has unknown type %d", i, p.type);
            return RT_ERR_SCENE;
        }
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

// ---------------------------------------------------------------- image textures, triangles, meshes
int add_image_texture(Scene &s, int rows, int cols, const uint8_t *rgb, const std::string &file) {
    if (rows < 1 || cols < 1 || rows > 16384 || cols > 16384 || !rgb) {
        set_error("image texture: rows and cols must be in 1..16384 and the pixels not null (got %d x %d)", rows, cols);
        return -RT_ERR_ARG;
    }
    SceneImage im;
    im.rows = rows, im.cols = cols, im.file = file;
    im.rgb.assign(rgb, rgb + (size_t)rows * cols * 3);
    s.images.push_back(std::move(im));
    rt_texture t;
    memset(&t, 0, sizeof t);
    t.type = RT_TEX_IMAGE;
    t.c0[0] = (float)(s.images.size() - 1), t.c0[1] = (float)rows, t.c0[2] = (float)cols;
    s.texs.push_back(t);
    s.touch();
    return (int)s.texs.size() - 1;
}

// P6 (binary) or P3 (text) PPM with maxval 255; '#' comments allowed in the header
static int read_ppm(const char *path, int &rows, int &cols, std::vector<uint8_t> &rgb) {
    FILE *fp = fopen(path, "rb");
    if (!fp) {
        set_error("cannot open %s", path);
        return RT_ERR_IO;
    }
    std::vector<unsigned char> buf;
    unsigned char chunk[65536];
    size_t n;
    while ((n = fread(chunk, 1, sizeof chunk, fp)) > 0) buf.insert(buf.end(), chunk, chunk + n);
    fclose(fp);
    size_t pos = 0;
    auto token = [&](long &out) -> bool {  // next unsigned integer of the header / a P3 body
        for (;;) {
            while (pos < buf.size() && isspace(buf[pos])) ++pos;
            if (pos < buf.size() && buf[pos] == '#') {
                while (pos < buf.size() && buf[pos] != '\n') ++pos;
                continue;
            }
            break;
        }
        if (pos >= buf.size() || !isdigit(buf[pos])) return false;
        long v = 0;
        while (pos < buf.size() && isdigit(buf[pos])) {
            v = v * 10 + (buf[pos++] - '0');
            if (v > 100000000) return false;
        }
        out = v;
        return true;
    };
    if (buf.size() < 2 || buf[0] != 'P' || (buf[1] != '6' && buf[1] != '3')) {
        set_error("%s is not a PPM file (P6 or P3)", path);
        return RT_ERR_IO;
    }
    const bool binary = buf[1] == '6';
    pos = 2;
    long w = 0, h = 0, mx = 0;
    if (!token(w) || !token(h) || !token(mx) || w < 1 || h < 1 || w > 16384 || h > 16384 || mx != 255) {
        set_error("%s: unsupported PPM header (need width, height <= 16384 and maxval 255)", path);
        return RT_ERR_IO;
    }
    rows = (int)h, cols = (int)w;
    const size_t need = (size_t)w * h * 3;
    rgb.resize(need);
    if (binary) {
        ++pos;  // the single whitespace byte after maxval
        if (buf.size() - pos < need) {
            set_error("%s is truncated", path);
            return RT_ERR_IO;
        }
        memcpy(rgb.data(), buf.data() + pos, need);
    } else {
        for (size_t i = 0; i < need; ++i) {
            long v = 0;
            if (!token(v) || v > 255) {
                set_error("%s: bad or missing sample %zu", path, i);
                return RT_ERR_IO;
            }
            rgb[i] = (uint8_t)v;
        }
    }
    return RT_OK;
}

int add_image_texture_file(Scene &s, const char *path) {
    int rows = 0, cols = 0;
    std::vector<uint8_t> rgb;
    // PNG by signature (8-bit, non-interlaced), else PPM
    bool is_png = false;
    if (FILE *fp = fopen(path, "rb")) {
        unsigned char sig[8] = {0};
        is_png = fread(sig, 1, 8, fp) == 8 && sig[0] == 137 && sig[1] == 'P' && sig[2] == 'N' && sig[3] == 'G';
        std::vector<uint8_t> file;
        if (is_png) {
            fseek(fp, 0, SEEK_SET);
            unsigned char chunk[65536];
            size_t n;
            while ((n = fread(chunk, 1, sizeof chunk, fp)) > 0) file.insert(file.end(), chunk, chunk + n);
        }
        fclose(fp);
        if (is_png) {
            std::string err;
            if (!png::read(file, rows, cols, rgb, err)) {
                set_error("%s: %s", path, err.c_str());
                return -RT_ERR_IO;
            }
            return add_image_texture(s, rows, cols, rgb.data(), path);
        }
    }
    int rc = read_ppm(path, rows, cols, rgb);
    if (rc) return -rc;
    return add_image_texture(s, rows, cols, rgb.data(), path);
}

// Triangle.__init__, taichi-version/hittable.py:95-110: the unit normal (v2 - v1) x (v3 - v1) / |..| is a
// constructor value there too; derived in fp64 from the fp32 vertices and rounded once
int add_triangle(Scene &s, const float v1[3], const float v2[3], const float v3[3], const float uv1[2], const float uv2[2],
                 const float uv3[2], int material) {
    double a[3], b[3], n[3];
    for (int k = 0; k < 3; ++k) a[k] = (double)v2[k] - (double)v1[k], b[k] = (double)v3[k] - (double)v1[k];
    v3cross(a, b, n);
    const double len = v3len(n);
    if (!(len > 0) || !std::isfinite(len)) {
        set_error("triangle with zero area (or non-finite vertices)");
        return -RT_ERR_SCENE;
    }
    rt_prim p;
    memset(&p, 0, sizeof p);
    p.type = RT_PRIM_TRIANGLE;
    p.material = material;
    for (int k = 0; k < 3; ++k) {
        p.m[k] = v1[k], p.m[3 + k] = v2[k], p.m[6 + k] = v3[k];
        p.m[9 + k] = (float)(n[k] / len);
    }
    const float *uv[3] = {uv1, uv2, uv3};
    for (int c = 0; c < 3; ++c)
        if (uv[c]) p.m_inv[2 * c] = uv[c][0], p.m_inv[2 * c + 1] = uv[c][1];
    s.prims.push_back(p);
    s.xforms.emplace_back();
    s.touch();
    return (int)s.prims.size() - 1;
}

// readobj(), taichi-version/main.py:23-41, and the placement of main.py:110-118 (scale * Rot @ x + dis)
int add_obj(Scene &s, const char *path, int material, float scale, const float matrix[9], const float translate[3]) {
    FILE *fp = fopen(path, "r");
    if (!fp) {
        set_error("cannot open %s", path);
        return -RT_ERR_IO;
    }
    std::vector<std::array<float, 3>> pts;
    std::vector<std::array<float, 2>> vts;
    struct Corner {
        long v, t;
    };
    std::vector<std::array<Corner, 3>> faces;
    char line[1024];
    int lineno = 0;
    bool bad = false;
    while (fgets(line, sizeof line, fp)) {
        ++lineno;
        char *p = line;
        while (*p == ' ' || *p == '\t') ++p;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            double x, y, z;
            if (sscanf(p + 1, "%lf %lf %lf", &x, &y, &z) != 3) bad = true;
            else pts.push_back({(float)x, (float)y, (float)z});
        } else if (p[0] == 'v' && p[1] == 't' && (p[2] == ' ' || p[2] == '\t')) {
            double u, v;
            if (sscanf(p + 2, "%lf %lf", &u, &v) != 2) bad = true;
            else vts.push_back({(float)u, (float)v});
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            std::array<Corner, 3> f;
            char *q = p + 1;
            int got = 0;
            for (; got < 3; ++got) {
                while (*q == ' ' || *q == '\t') ++q;
                char *e;
                long vi = strtol(q, &e, 10), ti = 0;
                if (e == q) break;
                q = e;
                if (*q == '/') {  // a/t or a/t/n or a//n
                    ++q;
                    ti = strtol(q, &e, 10);
                    q = e;
                    if (*q == '/') {
                        ++q;
                        (void)strtol(q, &e, 10);
                        q = e;
                    }
                }
                f[got] = {vi, ti ? ti : vi};  // without vt indices the reference takes texids[face corner]
            }
            if (got != 3) bad = true;
            else faces.push_back(f);
        }
        if (bad) break;
    }
    fclose(fp);
    if (bad) {
        set_error("%s:%d: cannot parse this line", path, lineno);
        return -RT_ERR_IO;
    }
    int added = 0;
    for (const auto &f : faces) {
        float v[3][3], uv[3][2];
        for (int c = 0; c < 3; ++c) {
            const long vi = f[c].v, ti = f[c].t;
            if (vi < 1 || vi > (long)pts.size()) {
                set_error("%s: face refers to vertex %ld (have %zu)", path, vi, pts.size());
                return -RT_ERR_SCENE;
            }
            const auto &x = pts[vi - 1];
            for (int r = 0; r < 3; ++r) {
                double m = matrix ? (double)matrix[3 * r] * x[0] + (double)matrix[3 * r + 1] * x[1] + (double)matrix[3 * r + 2] * x[2]
                                  : (double)x[r];
                v[c][r] = (float)((double)scale * m + (translate ? (double)translate[r] : 0.0));
            }
            uv[c][0] = uv[c][1] = 0.0f;
            if (ti >= 1 && ti <= (long)vts.size()) uv[c][0] = vts[ti - 1][0], uv[c][1] = vts[ti - 1][1];
        }
        int rc = add_triangle(s, v[0], v[1], v[2], uv[0], uv[1], uv[2], material);
        if (rc < 0) return rc;
        ++added;
    }
    return added;
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

int scene_from_json(const char *text, size_t len, Scene &s, const char *base_dir) {
    auto resolve = [&](const std::string &file) {  // relative "file" entries: next to the scene file
        if (file.empty() || file[0] == '/' || !base_dir || !*base_dir) return file;
        return std::string(base_dir) + "/" + file;
    };
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
            } else if (ty->str == "image") {
                // taichi-version/material.py:137-144 (one 100 x 100 image there, read with cv2 at hittable.py:165-172)
                int id = -1;
                if (const JsonValue *f = t.find("file")) {
                    if (!f->is_string()) r.fail("%s: \"file\" must be a string", where);
                    else {
                        id = add_image_texture_file(s, resolve(f->str).c_str());
                        if (id >= 0) s.images.back().file = f->str;  // serialised as given
                    }
                } else {
                    const int rows = r.integer(t, "rows", where), cols = r.integer(t, "cols", where);
                    const JsonValue *d = t.find("data");
                    if (r.ok && (!d || !d->is_array() || rows < 1 || cols < 1 || d->arr.size() != (size_t)rows * cols * 3))
                        r.fail("%s: \"data\" must be an array of rows * cols * 3 bytes", where);
                    if (r.ok) {
                        std::vector<uint8_t> px(d->arr.size());
                        for (size_t k = 0; k < px.size() && r.ok; ++k) {
                            const JsonValue &v = d->arr[k];
                            if (!v.is_number() || v.num < 0 || v.num > 255 || v.num != std::floor(v.num))
                                r.fail("%s: \"data\"[%zu] is not a byte", where, k);
                            else px[k] = (uint8_t)v.num;
                        }
                        if (r.ok) id = add_image_texture(s, rows, cols, px.data(), "");
                    }
                }
                if (r.ok && id < 0) return -id;
                continue;  // add_image_texture appended the texture record
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
            } else if (t == "triangle") {  // taichi-version/hittable.py:95-110
                double v[3][3], u[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
                r.vec3(o, "v1", where, v[0]), r.vec3(o, "v2", where, v[1]), r.vec3(o, "v3", where, v[2]);
                const char *uk[3] = {"u1", "u2", "u3"};
                for (int c = 0; c < 3; ++c)
                    if (const JsonValue *uv = o.find(uk[c])) {
                        if (!uv->is_array() || uv->arr.size() < 2 || !uv->arr[0].is_number() || !uv->arr[1].is_number())
                            r.fail("%s: \"%s\" must be an array of 2 numbers", where, uk[c]);
                        else u[c][0] = uv->arr[0].num, u[c][1] = uv->arr[1].num;
                    }
                const int mat = r.integer(o, "material", where);
                if (r.ok) {
                    float vf[3][3], uf[3][2];
                    for (int c = 0; c < 3; ++c) {
                        for (int k = 0; k < 3; ++k) vf[c][k] = (float)v[c][k];
                        uf[c][0] = (float)u[c][0], uf[c][1] = (float)u[c][1];
                    }
                    if (add_triangle(s, vf[0], vf[1], vf[2], uf[0], uf[1], uf[2], mat) < 0) return RT_ERR_SCENE;
                }
            } else if (t == "mesh") {  // readobj + placement, taichi-version/main.py:23-41, 110-118
                const JsonValue *f = o.find("file");
                if (!f || !f->is_string()) r.fail("%s: missing string \"file\"", where);
                const int mat = r.integer(o, "material", where);
                float scale = 1.0f, M[9], T[3];
                const float *pm = nullptr, *pt = nullptr;
                if (o.find("scale")) scale = (float)r.num(o, "scale", where);
                if (const JsonValue *m = o.find("matrix")) {
                    if (!m->is_array() || m->arr.size() != 9) r.fail("%s: \"matrix\" must be an array of 9 numbers", where);
                    else {
                        for (int k = 0; k < 9; ++k) {
                            if (!m->arr[k].is_number()) r.fail("%s: \"matrix\" must be an array of 9 numbers", where);
                            M[k] = (float)m->arr[k].num;
                        }
                        pm = M;
                    }
                }
                if (o.find("translate")) {
                    double d[3];
                    r.vec3(o, "translate", where, d);
                    T[0] = (float)d[0], T[1] = (float)d[1], T[2] = (float)d[2];
                    pt = T;
                }
                if (r.ok) {
                    const int rc = add_obj(s, resolve(f->str).c_str(), mat, scale, pm, pt);
                    if (rc < 0) return -rc;
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
        case RT_PRIM_TRIANGLE: {
            o += "{\"type\": \"triangle\"";
            const char *vk[3] = {"v1", "v2", "v3"}, *uk[3] = {"u1", "u2", "u3"};
            for (int c = 0; c < 3; ++c) {
                o += std::string(", \"") + vk[c] + "\": ";
                put_vec3(o, p.m + 3 * c);
            }
            for (int c = 0; c < 3; ++c)
                o += std::string(", \"") + uk[c] + "\": [" + json_float(p.m_inv[2 * c]) + ", " + json_float(p.m_inv[2 * c + 1]) + "]";
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
        } else if (t.type == RT_TEX_IMAGE) {
            const SceneImage &im = s.images[(size_t)t.c0[0]];
            if (!im.file.empty()) {
                o += "{\"type\": \"image\", \"file\": " + json_escape(im.file) + "}";
            } else {
                o += "{\"type\": \"image\", \"rows\": " + std::to_string(im.rows) + ", \"cols\": " + std::to_string(im.cols) +
                     ", \"data\": [";
                for (size_t k = 0; k < im.rgb.size(); ++k) o += (k ? "," : "") + std::to_string((int)im.rgb[k]);
                o += "]}";
            }
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
