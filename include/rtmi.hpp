// rtmi.hpp -- header-only C++ view of the C ABI (rtmi.h) with the reference's class names and
// constructor argument lists, so host code written against the reference's object model
// (gpu-version/{camera,object,material,texture}.cuh) builds a scene the same way:
//
//     rtmi::scene sc(400, 225, 100, 50);
//     sc.set_camera(rtmi::camera({-2,2,1}, {0,0,-1}, {0,1,0}, 20, 16.0f/9, 0.0f, 0.0f));
//     auto ground = rtmi::lambertian(rtmi::color(0.8f, 0.8f, 0.0f));
//     sc.add(rtmi::sphere({0,-100.5f,-1}, 100, ground));
//     auto tube = rtmi::cylinder(0.25f, -1, 1, rtmi::dielectric(1.5f));
//     tube.rotate({0,1,0}, 3.14159265f / 2);  tube.translate({0,0,0});       // object.cuh:225-231
//     sc.add(tube);
//     std::vector<float> image = sc.render();                                 // main.cu:505-513
//     rtmi::output_image(image, sc, "main.ppm");                              // main.cu:359-372
//
// Objects are value descriptors (no device pointers, no virtual calls); textures and materials
// are registered in the scene tables when the object that uses them is added.
#pragma once
#include <cmath>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "rtmi.h"

namespace rtmi {

struct vec3 {  // vec3.cuh:9-79
    float e[3];
    vec3() : e{0, 0, 0} {}
    vec3(float x, float y, float z) : e{x, y, z} {}
    float x() const { return e[0]; }
    float y() const { return e[1]; }
    float z() const { return e[2]; }
};
using point3 = vec3;
using color = vec3;

struct error : std::runtime_error {
    int status;
    error(int st, const std::string &where)
        : std::runtime_error(where + ": " + rt_status_string(st) + ": " + rt_last_error()), status(st) {}
};

// ---- textures (texture.cuh) --------------------------------------------------------------
struct mytexture {
    int type = RT_TEX_SOLID;
    color c0, c1;
    // image texture (taichi-version/material.py:96-110, 137-144): rows x cols texels, R G B bytes
    int rows = 0, cols = 0;
    std::vector<uint8_t> rgb;
};
inline std::shared_ptr<mytexture> solid_color(color c) {  // texture.cuh:18-19
    auto t = std::make_shared<mytexture>();
    t->type = RT_TEX_SOLID, t->c0 = c, t->c1 = c;
    return t;
}
inline std::shared_ptr<mytexture> checker_texture(color even, color odd) {  // texture.cuh:40-42
    auto t = std::make_shared<mytexture>();
    t->type = RT_TEX_CHECKER, t->c0 = even, t->c1 = odd;
    return t;
}

inline std::shared_ptr<mytexture> image_texture(int rows, int cols, std::vector<uint8_t> rgb) {
    auto t = std::make_shared<mytexture>();
    t->type = RT_TEX_IMAGE, t->rows = rows, t->cols = cols, t->rgb = std::move(rgb);
    return t;
}

// ---- materials (material.cuh) --------------------------------------------------------------
struct material {
    int type = RT_MAT_LAMBERTIAN;
    std::shared_ptr<mytexture> tex;  // lambertian albedo / diffuse_light emit
    color albedo;
    float fuzz = 0, ir = 1;
};
using material_ptr = std::shared_ptr<material>;
inline material_ptr lambertian(std::shared_ptr<mytexture> a) {  // material.cuh:34-35
    auto m = std::make_shared<material>();
    m->type = RT_MAT_LAMBERTIAN, m->tex = std::move(a);
    return m;
}
inline material_ptr lambertian(color a) { return lambertian(solid_color(a)); }  // material.cuh:31-32
inline material_ptr metal(color a, float f) {                                     // material.cuh:60-61
    auto m = std::make_shared<material>();
    m->type = RT_MAT_METAL, m->albedo = a, m->fuzz = f;
    return m;
}
inline material_ptr dielectric(float index_of_refraction) {  // material.cuh:91-92
    auto m = std::make_shared<material>();
    m->type = RT_MAT_DIELECTRIC, m->ir = index_of_refraction;
    return m;
}
inline material_ptr diffuse_light(std::shared_ptr<mytexture> a) {  // material.cuh:163-164
    auto m = std::make_shared<material>();
    m->type = RT_MAT_DIFFUSE_LIGHT, m->tex = std::move(a);
    return m;
}
inline material_ptr diffuse_light(color c) { return diffuse_light(solid_color(c)); }  // material.cuh:166-167

// ---- hittables (object.cuh) ------------------------------------------------------------------
struct hittable {
    int type = RT_PRIM_SPHERE;
    float f[6] = {0, 0, 0, 0, 0, 0};
    material_ptr mat;
    // triangle only (taichi-version/hittable.py:95-110): corners and their texture coordinates
    float v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, uv[6] = {0, 0, 0, 0, 0, 0};
    // cylinder only: rotate (axis, radians) then translate, as cylinder::rotate/translate compose
    bool has_rotate = false, has_translate = false;
    vec3 axis{0, 0, 1}, offset;
    float radians = 0;
    void rotate(vec3 a, float rad) {  // object.cuh:225-227
        if (type != RT_PRIM_CYLINDER || has_rotate || has_translate)
            throw std::logic_error("rotate: only once, on a cylinder, before translate");
        has_rotate = true, axis = a, radians = rad;
    }
    void translate(vec3 o) {  // object.cuh:229-231
        if (type != RT_PRIM_CYLINDER || has_translate) throw std::logic_error("translate: only once, on a cylinder");
        has_translate = true, offset = o;
    }
};
inline hittable sphere(point3 cen, float r, material_ptr m) {  // object.cuh:44-45
    hittable h;
    h.type = RT_PRIM_SPHERE, h.f[0] = cen.x(), h.f[1] = cen.y(), h.f[2] = cen.z(), h.f[3] = r, h.mat = std::move(m);
    return h;
}
inline hittable make_rect(int type, float a0, float a1, float b0, float b1, float k, material_ptr m) {
    hittable h;
    h.type = type, h.f[0] = a0, h.f[1] = a1, h.f[2] = b0, h.f[3] = b1, h.f[4] = k, h.mat = std::move(m);
    return h;
}
inline hittable xy_rect(float x0, float x1, float y0, float y1, float k, material_ptr m) {  // object.cuh:100-103
    return make_rect(RT_PRIM_XY_RECT, x0, x1, y0, y1, k, std::move(m));
}
inline hittable xz_rect(float x0, float x1, float z0, float z1, float k, material_ptr m) {  // object.cuh:138-141
    return make_rect(RT_PRIM_XZ_RECT, x0, x1, z0, z1, k, std::move(m));
}
inline hittable yz_rect(float y0, float y1, float z0, float z1, float k, material_ptr m) {  // object.cuh:170-173
    return make_rect(RT_PRIM_YZ_RECT, y0, y1, z0, z1, k, std::move(m));
}
// Triangle(v1, v2, v3, u1, u2, u3, material), taichi-version/hittable.py:95-110 (u = texture coordinates, 2 each)
inline hittable triangle(point3 v1, point3 v2, point3 v3, material_ptr m, const float *u1 = nullptr, const float *u2 = nullptr,
                         const float *u3 = nullptr) {
    hittable h;
    h.type = RT_PRIM_TRIANGLE, h.mat = std::move(m);
    const point3 *c[3] = {&v1, &v2, &v3};
    const float *u[3] = {u1, u2, u3};
    for (int k = 0; k < 3; ++k) {
        h.v[3 * k] = c[k]->x(), h.v[3 * k + 1] = c[k]->y(), h.v[3 * k + 2] = c[k]->z();
        if (u[k]) h.uv[2 * k] = u[k][0], h.uv[2 * k + 1] = u[k][1];
    }
    return h;
}
inline hittable cylinder(float radius, float zmin, float zmax, material_ptr m) {  // object.cuh:220-223
    hittable h;
    h.type = RT_PRIM_CYLINDER, h.f[0] = radius, h.f[1] = zmin, h.f[2] = zmax, h.mat = std::move(m);
    return h;
}

struct camera {  // camera.cuh:9-15
    point3 lookfrom, lookat;
    vec3 vup;
    float vfov, aspect_ratio, aperture, focus_dist;
    camera(point3 from, point3 at, vec3 up, float fov, float aspect, float ap, float focus)
        : lookfrom(from), lookat(at), vup(up), vfov(fov), aspect_ratio(aspect), aperture(ap), focus_dist(focus) {}
};

// ---- scene = parser.hpp:16-32 `struct scene` + hittable_list ------------------------------------
class scene {
public:
    scene(int width, int height, int samples_per_pixel, int max_depth)
        : s_(rt_scene_new(width, height, samples_per_pixel, max_depth)) {
        if (!s_) throw error(RT_ERR_ARG, "rt_scene_new");
    }
    explicit scene(const std::string &json_file) : s_(rt_scene_load_json(json_file.c_str())) {  // parse_scene
        if (!s_) throw error(RT_ERR_SCENE, "parse_scene(" + json_file + ")");
    }
    ~scene() { rt_scene_free(s_); }
    scene(const scene &) = delete;
    scene &operator=(const scene &) = delete;

    void set_background(color c, bool sky_gradient, bool defocus_blur = true) {
        check(rt_scene_set_background(s_, c.e, (sky_gradient ? RT_FLAG_SKY_GRADIENT : 0u) |
                                                   (defocus_blur ? RT_FLAG_DEFOCUS_BLUR : 0u)),
              "set_background");
    }
    void set_camera(const camera &c) {
        check(rt_scene_set_camera(s_, c.lookfrom.e, c.lookat.e, c.vup.e, c.vfov, c.aspect_ratio, c.aperture,
                                  c.focus_dist),
              "camera");
    }
    // hittable_list::add
    int add(const hittable &h) {
        if (!h.mat) throw std::logic_error("hittable without a material");
        const int m = material_id(h.mat);
        int id;
        switch (h.type) {
        case RT_PRIM_SPHERE: id = rt_scene_add_sphere(s_, h.f, h.f[3], m); break;
        case RT_PRIM_TRIANGLE: id = rt_scene_add_triangle(s_, h.v, h.v + 3, h.v + 6, h.uv, h.uv + 2, h.uv + 4, m); break;
        case RT_PRIM_CYLINDER: {
            const float deg = h.radians * 180.0f / 3.14159265358979323846f;
            id = rt_scene_add_cylinder(s_, h.f[0], h.f[1], h.f[2], m, h.has_rotate ? h.axis.e : nullptr, deg,
                                       h.has_translate ? h.offset.e : nullptr);
            break;
        }
        default: id = rt_scene_add_rect(s_, h.type - RT_PRIM_XY_RECT, h.f[0], h.f[1], h.f[2], h.f[3], h.f[4], m); break;
        }
        if (id < 0) throw error(-id, "add");
        return id;
    }
    // readobj + placement (taichi-version/main.py:23-41, 110-118): triangles added
    int add_obj(const std::string &path, const material_ptr &m, float scale = 1.0f, const float *matrix9 = nullptr,
                const float *translate3 = nullptr) {
        int n = rt_scene_add_obj(s_, path.c_str(), material_id(m), scale, matrix9, translate3);
        if (n < 0) throw error(-n, "add_obj");
        return n;
    }
    void override_size(int w, int h, int spp, int depth) { check(rt_scene_override(s_, w, h, spp, depth), "override"); }

    rt_scene_info info() const {
        rt_scene_info i;
        check(rt_scene_get_info(s_, &i), "info");
        return i;
    }
    std::string to_json() const {
        std::string out(rt_scene_to_json(s_, nullptr, 0), '\0');
        rt_scene_to_json(s_, &out[0], out.size());
        out.resize(out.size() - 1);
        return out;
    }
    // render<<<>>> + synchronise + copy back, main.cu:505-513: W*H*3 sums, row 0 = bottom
    std::vector<float> render(const rt_opts *opts = nullptr, rt_stats *stats = nullptr) const {
        rt_opts o;
        if (opts) o = *opts;
        else rt_opts_default(&o);
        const rt_scene_info i = info();
        std::vector<float> img((size_t)rt_shard_rows(s_, &o) * i.width * 3);
        check(rt_render_hip(s_, &o, img.data(), stats), "render");
        return img;
    }
    // Russian roulette (4_0_path_tracing.py's p_RR): survival probability per bounce, 0 = off
    void set_russian_roulette(float p) { check(rt_scene_set_russian_roulette(s_, p), "set_russian_roulette"); }
    // progressive rendering: adds samples [first, first + count) to the caller's exact pixel sums
    // (resized and zeroed when empty) and returns the framebuffer of the updated sums
    std::vector<float> accumulate(std::vector<int64_t> &acc, int first, int count, const rt_opts *opts = nullptr,
                                  rt_stats *stats = nullptr) const {
        rt_opts o;
        if (opts) o = *opts;
        else rt_opts_default(&o);
        o.sample_first = first, o.sample_count = count;
        const rt_scene_info i = info();
        const size_t n = (size_t)rt_shard_rows(s_, &o) * i.width * 3;
        if (acc.empty()) acc.assign(n, 0);
        if (acc.size() != n) throw std::runtime_error("rtmi: accumulate: accumulator size does not match the shard");
        std::vector<float> img(n);
        check(rt_render_hip_accumulate(s_, &o, acc.data(), img.data(), stats), "accumulate");
        return img;
    }
    rt_scene *handle() const { return s_; }

private:
    rt_scene *s_;
    // registered descriptors are kept alive so an address is never reused for another one
    std::vector<std::pair<material_ptr, int>> mats_;
    std::vector<std::pair<std::shared_ptr<mytexture>, int>> texs_;
    static void check(int st, const char *where) {
        if (st != RT_OK) throw error(st, where);
    }
    int texture_id(const std::shared_ptr<mytexture> &t) {
        for (auto &kv : texs_)
            if (kv.first == t) return kv.second;
        int id = t->type == RT_TEX_CHECKER ? rt_scene_add_checker(s_, t->c0.e, t->c1.e)
                 : t->type == RT_TEX_IMAGE ? rt_scene_add_image_texture(s_, t->rows, t->cols, t->rgb.data())
                                           : rt_scene_add_solid_color(s_, t->c0.e);
        if (id < 0) throw error(-id, "texture");
        texs_.emplace_back(t, id);
        return id;
    }
    int material_id(const material_ptr &m) {
        for (auto &kv : mats_)
            if (kv.first == m) return kv.second;
        int id;
        switch (m->type) {
        case RT_MAT_LAMBERTIAN: id = rt_scene_add_lambertian(s_, texture_id(m->tex)); break;
        case RT_MAT_METAL: id = rt_scene_add_metal(s_, m->albedo.e, m->fuzz); break;
        case RT_MAT_DIELECTRIC: id = rt_scene_add_dielectric(s_, m->ir); break;
        default: id = rt_scene_add_diffuse_light(s_, texture_id(m->tex)); break;
        }
        if (id < 0) throw error(-id, "material");
        mats_.emplace_back(m, id);
        return id;
    }
};

// output_image(image, W, H, spp, filename), main.cu:359-372
inline void output_image(const std::vector<float> &image, const scene &sc, const std::string &filename) {
    const rt_scene_info i = sc.info();
    int st = rt_write_ppm(filename.c_str(), image.data(), i.width, i.height, i.samples_per_pixel);
    if (st != RT_OK) throw error(st, "output_image");
}

}  // namespace rtmi
