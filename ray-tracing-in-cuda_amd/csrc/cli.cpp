// rtmi -- command-line renderer over the C ABI (include/rtmi.h).
//
// Drop-in for both reference executables:
//   gpu-version   `parallel_compute -f scene.json`            (main.cu:455-460) -> main.ppm
//   cmake-cpu     `ray_tracing -w W -h H -d DEPTH -spp N`     (main.cpp:71-81)  -> main.ppm
// plus --rtiow [--scene-seed S] (the hard-coded random_scene() of main.cpp:125-172),
// --seed, -o, --device, --chunk, --no-blur, --sky, --dump-json.  Timing goes to stderr
// like the reference's when() markers (rtweekend.cuh:40).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtmi.h"

static double now_s() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

static int usage(const char *argv0) {
    fprintf(stderr,
            "usage: %s [-f scene.json | --rtiow] [-w W] [-h H] [-d DEPTH] [-spp N] [-o out.ppm]\n"
            "          [--seed S] [--scene-seed S] [--device N] [--chunk N] [--dump-json file] [--count] [--no-png]\n",
            argv0);
    return 2;
}

int main(int argc, char **argv) {
    std::string scene_file = "sample_scene.json";  // main.cu:456 default
    std::string out_file = "main.ppm";             // main.cu:512
    std::string dump_json;
    bool rtiow = false, have_file = false, count = false, no_png = false;
    int w = 0, h = 0, depth = 0, spp = 0, device = 0, chunk = 0;
    unsigned long long seed = 2023;
    unsigned scene_seed = 7;  // srand(7), main.cpp:119
    for (int i = 1; i < argc; ++i) {
        auto need = [&](const char *flag) -> const char * {
            if (i + 1 >= argc) {
                fprintf(stderr, "%s needs a value\n", flag);
                exit(2);
            }
            return argv[++i];
        };
        if (!strcmp(argv[i], "-f")) scene_file = need("-f"), have_file = true;
        else if (!strcmp(argv[i], "-w")) w = atoi(need("-w"));
        else if (!strcmp(argv[i], "-h")) h = atoi(need("-h"));
        else if (!strcmp(argv[i], "-d")) depth = atoi(need("-d"));
        else if (!strcmp(argv[i], "-spp")) spp = atoi(need("-spp"));
        else if (!strcmp(argv[i], "-o")) out_file = need("-o");
        else if (!strcmp(argv[i], "--seed")) seed = strtoull(need("--seed"), nullptr, 0);
        else if (!strcmp(argv[i], "--scene-seed")) scene_seed = (unsigned)strtoul(need("--scene-seed"), nullptr, 0);
        else if (!strcmp(argv[i], "--device")) device = atoi(need("--device"));
        else if (!strcmp(argv[i], "--chunk")) chunk = atoi(need("--chunk"));
        else if (!strcmp(argv[i], "--dump-json")) dump_json = need("--dump-json");
        else if (!strcmp(argv[i], "--rtiow")) rtiow = true;
        else if (!strcmp(argv[i], "--count")) count = true;
        else if (!strcmp(argv[i], "--no-png")) no_png = true;
        else if (!strcmp(argv[i], "--help")) return usage(argv[0]);
        else {
            fprintf(stderr, "unknown argument '%s'\n", argv[i]);
            return usage(argv[0]);
        }
    }
    double t0 = now_s();
    rt_scene *sc;
    if (rtiow && !have_file) sc = rt_scene_rtiow(scene_seed, w > 0 ? w : 400, h > 0 ? h : 225, spp > 0 ? spp : 50,
                                                  depth > 0 ? depth : 50);  // defaults: main.cpp:64-68
    else sc = rt_scene_load_json(scene_file.c_str());
    if (!sc) {
        fprintf(stderr, "rtmi: %s\n", rt_last_error());
        return 1;
    }
    if (rt_scene_override(sc, w, h, spp, depth) != RT_OK) {
        fprintf(stderr, "rtmi: %s\n", rt_last_error());
        return 1;
    }
    rt_scene_info info;
    rt_scene_get_info(sc, &info);
    fprintf(stderr, "scene: %dx%d, %d spp, depth %d, %d objects, %d materials, %d textures\n", info.width,
            info.height, info.samples_per_pixel, info.max_depth, info.num_prims, info.num_materials,
            info.num_textures);
    if (!dump_json.empty()) {
        size_t n = rt_scene_to_json(sc, nullptr, 0);
        std::vector<char> buf(n);
        rt_scene_to_json(sc, buf.data(), n);
        FILE *fp = fopen(dump_json.c_str(), "w");
        if (!fp) {
            fprintf(stderr, "rtmi: cannot write %s\n", dump_json.c_str());
            return 1;
        }
        fwrite(buf.data(), 1, n - 1, fp);
        fclose(fp);
    }
    rt_opts o;
    rt_opts_default(&o);
    o.seed = seed;
    o.device = device;
    o.spp_chunk = chunk;
    rt_stats st;
    std::vector<float> img((size_t)info.width * info.height * 3);
    int rc = count ? rt_render_hip_count(sc, &o, img.data(), &st) : rt_render_hip(sc, &o, img.data(), &st);
    if (rc != RT_OK) {
        fprintf(stderr, "rtmi: render failed: %s: %s\n", rt_status_string(rc), rt_last_error());
        return 1;
    }
    double samples = (double)info.width * info.height * info.samples_per_pixel;
    fprintf(stderr, "render: %.3f ms kernel (%.1f Msamples/s), %.3f ms upload\n", st.kernel_ms,
            samples / (st.kernel_ms * 1e3), st.upload_ms);
    if (count)
        fprintf(stderr, "counts: samples %llu queries %llu prim_tests %llu hits %llu misses %llu draws %llu\n",
                (unsigned long long)st.samples, (unsigned long long)st.queries, (unsigned long long)st.prim_tests,
                (unsigned long long)st.hits, (unsigned long long)st.misses, (unsigned long long)st.rng_draws);
    if (rt_write_ppm(out_file.c_str(), img.data(), info.width, info.height, info.samples_per_pixel) != RT_OK) {
        fprintf(stderr, "rtmi: %s\n", rt_last_error());
        return 1;
    }
    // write_image(..., data["output_file"]), main.cu:514: linear PNG next to the PPM (skipped when the
    // directory of output_file does not exist, which the reference would crash on)
    if (!no_png) {
        const char *png = rt_scene_output_file(sc);
        if (rt_write_png(png, img.data(), info.width, info.height, info.samples_per_pixel, 0) != RT_OK)
            fprintf(stderr, "rtmi: PNG not written: %s\n", rt_last_error());
    }
    fprintf(stderr, "Program finish, cost: %f s\n", now_s() - t0);  // main.cu:519-520
    rt_scene_free(sc);
    return 0;
}
