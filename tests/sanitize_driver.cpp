// Host-side exercise of the C ABI under AddressSanitizer + UBSan (CPU build of scene.cpp + capi.cpp
// only: no HIP in this binary).  Built and run by tests/test_sanitize.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rtmi.h"

#define CHECK(c)                                                              \
    do {                                                                      \
        if (!(c)) {                                                           \
            fprintf(stderr, "CHECK failed line %d: %s (%s)\n", __LINE__, #c, rt_last_error()); \
            return 1;                                                         \
        }                                                                     \
    } while (0)

static std::string to_json(const rt_scene *s) {
    std::string out(rt_scene_to_json(s, nullptr, 0), '\0');
    rt_scene_to_json(s, &out[0], out.size());
    out.resize(out.size() - 1);
    return out;
}

int main(int argc, char **argv) {
    CHECK(argc >= 3);
    const std::string tmp = argv[1];
    for (int i = 2; i < argc; ++i) {  // scene files: parse, serialise, re-parse, compare tables
        rt_scene *a = rt_scene_load_json(argv[i]);
        CHECK(a);
        std::string j = to_json(a);
        rt_scene *b = rt_scene_parse_json(j.data(), j.size());
        CHECK(b);
        rt_scene_info ia, ib;
        CHECK(rt_scene_get_info(a, &ia) == RT_OK && rt_scene_get_info(b, &ib) == RT_OK);
        CHECK(ia.num_prims == ib.num_prims && ia.num_materials == ib.num_materials);
        std::vector<rt_prim> pa(ia.num_prims + 1), pb(ib.num_prims + 1);
        CHECK(rt_scene_get_prims(a, pa.data(), ia.num_prims) == ia.num_prims);
        CHECK(rt_scene_get_prims(b, pb.data(), ib.num_prims) == ib.num_prims);
        CHECK(memcmp(pa.data(), pb.data(), sizeof(rt_prim) * ia.num_prims) == 0);
        CHECK(rt_scene_rotate_cylinders(a, 17.5) >= 0);
        rt_scene *c = rt_scene_clone(a);
        CHECK(c);
        CHECK(rt_scene_override(c, 64, 36, 2, 5) == RT_OK);
        CHECK(rt_scene_override(c, 1, 0, 0, 0) != RT_OK);
        rt_camera cam;
        CHECK(rt_scene_get_camera(c, &cam) == RT_OK);
        rt_scene_free(a), rt_scene_free(b), rt_scene_free(c);
    }
    // generators
    rt_scene *r = rt_scene_rtiow(7, 48, 27, 4, 50);
    CHECK(r);
    rt_scene *d = rt_scene_dna(r, 33.0);
    CHECK(d);
    rt_scene *d0 = rt_scene_dna(nullptr, 0.0);
    CHECK(d0);
    rt_scene_free(r), rt_scene_free(d), rt_scene_free(d0);
    // malformed inputs never crash
    const char *bad[] = {"", "{", "[1,2", "{\"a\":}", "{\"background\":[1,2]}", "\"\\u12\"", "{\"object\":{\"data\":[{}]}}",
                         "nul", "-", "1e", "{\"a\":1,}", "[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[["};
    for (const char *t : bad) CHECK(rt_scene_parse_json(t, strlen(t)) == nullptr);
    CHECK(rt_scene_load_json("/nonexistent/x.json") == nullptr);
    CHECK(rt_scene_parse_json(nullptr, 0) == nullptr);
    // builders with bad arguments
    rt_scene *s = rt_scene_new(16, 9, 1, 5);
    CHECK(s);
    float v[3] = {1, 2, 3};
    CHECK(rt_scene_add_sphere(s, v, 1.0f, 0) < 0);       // no material yet
    CHECK(rt_scene_add_lambertian(s, 3) < 0);            // no texture
    int t = rt_scene_add_checker(s, v, v);
    int m = rt_scene_add_diffuse_light(s, t);
    CHECK(t == 0 && m == 0);
    CHECK(rt_scene_add_sphere(s, v, 0.0f, m) < 0);       // zero radius
    CHECK(rt_scene_add_rect(s, 5, 0, 1, 0, 1, 0, m) < 0);
    float zero[3] = {0, 0, 0};
    CHECK(rt_scene_add_cylinder(s, 1, 0, 1, m, zero, 10, nullptr) < 0);
    CHECK(rt_scene_add_cylinder(s, 1, 0, 1, m, v, 10, v) >= 0);
    CHECK(rt_scene_set_camera(s, v, v, v, 20, 0, 0, 0) == RT_OK);  // lookfrom == lookat: caught by validate
    CHECK(rt_scene_override(s, 0, 0, 0, 0) != RT_OK);
    rt_scene_free(s);
    // writers
    std::vector<float> img(7 * 5 * 3);
    for (size_t i = 0; i < img.size(); ++i) img[i] = (float)i * 0.37f - 3.0f;  // incl. negatives
    img[5] = 0.0f / 1.0f * 0.0f - 0.0f;
    img[6] = __builtin_nanf("");
    img[7] = __builtin_inff();
    CHECK(rt_write_ppm((tmp + "/a.ppm").c_str(), img.data(), 7, 5, 3) == RT_OK);
    CHECK(rt_write_png((tmp + "/a.png").c_str(), img.data(), 7, 5, 3, 0) == RT_OK);
    CHECK(rt_write_png((tmp + "/b.png").c_str(), img.data(), 7, 5, 3, 1) == RT_OK);
    CHECK(rt_write_ppm("/nonexistent/dir/a.ppm", img.data(), 7, 5, 3) == RT_ERR_IO);
    std::vector<uint8_t> q(7 * 5 * 3);
    CHECK(rt_quantize_rgb8(img.data(), 7, 5, 3, 1, q.data()) == RT_OK);
    uint32_t ctr[4] = {1, 2, 3, 4}, key[2] = {5, 6}, out[4], words[16];
    rt_philox4x32_10(ctr, key, out);
    rt_sample_stream(123, 4, 5, words, 16);
    float lo[3] = {-1, -1, -1}, hi[3] = {1, 1, 1}, o[3] = {0, 0, 5}, dir[3] = {0, 0, -1};
    CHECK(rt_aabb_hit(lo, hi, o, dir, 0.001f, 100.0f) == 1);
    puts("sanitize driver ok");
    return 0;
}
