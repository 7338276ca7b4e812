// rtmi -- command-line renderer over the C ABI (include/rtmi.h).
//
// Drop-in for both reference executables:
//   gpu-version   `parallel_compute -f scene.json`            (main.cu:455-460) -> main.ppm
//   cmake-cpu     `ray_tracing -w W -h H -d DEPTH -spp N`     (main.cpp:71-81)  -> main.ppm
// plus --rtiow [--scene-seed S] (the hard-coded random_scene() of main.cpp:125-172),
// --seed, -o, --device, --chunk, --dump-json.  Timing goes to stderr like the reference's
// when() markers (rtweekend.cuh:40).
// --gpus N renders ONE frame on the first N GPUs of the node: row tiles interleaved over the devices, one
// RCCL gather (rt_render_hip_tiles); the image is the same bytes as with one GPU.  (The reference's blue.py
// instead starts one process per GPU per animation frame; rtmi-frames is that shape.)
// Resumable rendering: --acc-out FILE saves the exact pixel sums, --acc-in FILE continues from them
// (-spp is then the number of samples to ADD; --spp-begin overrides the first sample index).  Any
// split of a sample range into runs writes the same main.ppm as one run over the whole range.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtmi.h"

// exact pixel sums on disk: 40-byte header + int64[height][width][3], little endian
struct AccHeader {
    char magic[8];  // "RTMIACC2" (version 1 held sums in units of 2^-32)
    int32_t width, height;
    int64_t samples_done;  // samples [0, samples_done) are in the sums
    uint64_t seed;
    int64_t fix_bits;      // fractional bits of the sums (RT_ACC_FIX_BITS): a file in another scale is refused
};
static_assert(sizeof(AccHeader) == 40, "accumulator file header");

static double now_s() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

static int usage(const char *argv0) {
    fprintf(stderr,
            "usage: %s [-f scene.json | --rtiow] [-w W] [-h H] [-d DEPTH] [-spp N] [-o out.ppm]\n"
            "          [--seed S] [--scene-seed S] [--device N] [--chunk N] [--dump-json file] [--count] [--no-png]\n"
            "          [--acc-in sums.bin] [--acc-out sums.bin] [--spp-begin FIRST] [--rr SURVIVAL_PROBABILITY]\n"
            "          [--gpus N] [--tile-rows R]\n",
            argv0);
    return 2;
}

int main(int argc, char **argv) {
    std::string scene_file = "sample_scene.json";  // main.cu:456 default
    std::string out_file = "main.ppm";             // main.cu:512
    std::string dump_json, acc_in, acc_out;
    long long spp_begin = -1;
    double rr = -1.0;  // Russian roulette: keep the scene file's setting
    bool rtiow = false, have_file = false, count = false, no_png = false;
    int w = 0, h = 0, depth = 0, spp = 0, device = 0, chunk = 0, gpus = 0, tile_rows = 0;
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
        else if (!strcmp(argv[i], "--gpus")) gpus = atoi(need("--gpus"));
        else if (!strcmp(argv[i], "--tile-rows")) tile_rows = atoi(need("--tile-rows"));
        else if (!strcmp(argv[i], "--dump-json")) dump_json = need("--dump-json");
        else if (!strcmp(argv[i], "--acc-in")) acc_in = need("--acc-in");
        else if (!strcmp(argv[i], "--acc-out")) acc_out = need("--acc-out");
        else if (!strcmp(argv[i], "--spp-begin")) spp_begin = atoll(need("--spp-begin"));
        else if (!strcmp(argv[i], "--rr")) rr = atof(need("--rr"));
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
    if (rr >= 0.0 && rt_scene_set_russian_roulette(sc, (float)rr) != RT_OK) {
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
    if (tile_rows > 0) o.tile_rows = tile_rows;
    rt_stats st;
    std::vector<float> img((size_t)info.width * info.height * 3);
    int total_spp = info.samples_per_pixel;  // divisor of the written image
    int rc;
    const bool progressive = !acc_in.empty() || !acc_out.empty() || spp_begin >= 0;
    if (progressive && gpus > 0) {
        fprintf(stderr, "rtmi: --gpus cannot be combined with --acc-in/--acc-out/--spp-begin\n");
        return 2;
    }
    if (progressive) {
        if (count) {
            fprintf(stderr, "rtmi: --count cannot be combined with --acc-in/--acc-out/--spp-begin\n");
            return 2;
        }
        std::vector<int64_t> acc(img.size(), 0);
        long long done = 0;
        if (!acc_in.empty()) {
            FILE *fp = fopen(acc_in.c_str(), "rb");
            AccHeader h;
            if (!fp || fread(&h, sizeof h, 1, fp) != 1 || memcmp(h.magic, "RTMIACC", 7) != 0) {
                fprintf(stderr, "rtmi: %s is not an accumulator file\n", acc_in.c_str());
                return 1;
            }
            if (h.magic[7] != '2' || h.fix_bits != RT_ACC_FIX_BITS) {
                fprintf(stderr, "rtmi: %s holds pixel sums in another fixed-point scale (format '%c', %lld fractional bits; this build: "
                        "'2', %d): it cannot be continued\n", acc_in.c_str(), h.magic[7], (long long)h.fix_bits, RT_ACC_FIX_BITS);
                return 1;
            }
            if (h.width != info.width || h.height != info.height || h.seed != seed) {
                fprintf(stderr, "rtmi: %s holds %dx%d sums of seed %llu, this run is %dx%d seed %llu\n", acc_in.c_str(),
                        h.width, h.height, (unsigned long long)h.seed, info.width, info.height, seed);
                return 1;
            }
            if (fread(acc.data(), sizeof(int64_t), acc.size(), fp) != acc.size()) {
                fprintf(stderr, "rtmi: %s is truncated\n", acc_in.c_str());
                return 1;
            }
            fclose(fp);
            done = h.samples_done;
        }
        if (spp_begin < 0) spp_begin = done;
        if (spp_begin != done) {
            fprintf(stderr, "rtmi: the sums hold samples [0, %lld) but --spp-begin is %lld\n", done, spp_begin);
            return 1;
        }
        if (spp_begin + info.samples_per_pixel > 0x7fffffffLL) {
            fprintf(stderr, "rtmi: sample index overflow\n");
            return 1;
        }
        o.sample_first = (int)spp_begin;
        o.sample_count = info.samples_per_pixel;
        rc = rt_render_hip_accumulate(sc, &o, acc.data(), img.data(), &st);
        total_spp = (int)(spp_begin + info.samples_per_pixel);
        if (rc == RT_OK && !acc_out.empty()) {
            AccHeader h = {{'R', 'T', 'M', 'I', 'A', 'C', 'C', '2'}, info.width, info.height, total_spp, seed, RT_ACC_FIX_BITS};
            FILE *fp = fopen(acc_out.c_str(), "wb");
            if (!fp || fwrite(&h, sizeof h, 1, fp) != 1 ||
                fwrite(acc.data(), sizeof(int64_t), acc.size(), fp) != acc.size() || fclose(fp) != 0) {
                fprintf(stderr, "rtmi: cannot write %s\n", acc_out.c_str());
                return 1;
            }
        }
        if (rc == RT_OK)
            fprintf(stderr, "progressive: samples [%lld, %d) added, image holds %d spp\n", spp_begin, total_spp, total_spp);
    } else if (gpus > 0) {
        if (count) {
            fprintf(stderr, "rtmi: --count is a single-device diagnostic; drop --gpus\n");
            return 2;
        }
        rc = rt_render_hip_tiles(sc, &o, nullptr, gpus, img.data(), &st);
        if (rc == RT_OK)
            fprintf(stderr, "tiles: %d device(s), slowest render %.3f ms, gather + placement %.3f ms\n", st.devices_used,
                    st.kernel_ms, st.gather_ms);
    } else {
        rc = count ? rt_render_hip_count(sc, &o, img.data(), &st) : rt_render_hip(sc, &o, img.data(), &st);
    }
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
    if (rt_write_ppm(out_file.c_str(), img.data(), info.width, info.height, total_spp) != RT_OK) {
        fprintf(stderr, "rtmi: %s\n", rt_last_error());
        return 1;
    }
    // write_image(..., data["output_file"]), main.cu:514: linear PNG next to the PPM (skipped when the
    // directory of output_file does not exist, which the reference would crash on)
    if (!no_png) {
        const char *png = rt_scene_output_file(sc);
        if (rt_write_png(png, img.data(), info.width, info.height, total_spp, 0) != RT_OK)
            fprintf(stderr, "rtmi: PNG not written: %s\n", rt_last_error());
    }
    fprintf(stderr, "Program finish, cost: %f s\n", now_s() - t0);  // main.cu:519-520
    rt_scene_free(sc);
    return 0;
}
