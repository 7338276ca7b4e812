// rtmi-frames -- the reference's animation harness (gpu-version/blue.py, blue2.py, dna.py) as one
// native program over the C ABI.
//
// The Python scripts rewrite a JSON scene per frame (every cylinder's rotate.angle += step,
// blue.py:16-19; or the DNA helix at angle k, dna.py:17-98), dump it to ./build/scene/..., and
// every 8th frame launch eight `parallel_compute` processes pinned with CUDA_VISIBLE_DEVICES=k
// (blue.py:23-32).  Here: one host thread per device, frame k rendered on device k mod N from an
// in-memory scene (no JSON round trip, no process per frame); output naming follows the scripts.
//
//   rtmi-frames --template blue.json --frames 360 --step 1 --out './build/output/blue/frame_%03d.png'
//   rtmi-frames --dna [--template basic_scene.json] --frames 360 --out './build/output/output_%03d.png'
//   common: [--devices N] [--first K] [-w W -h H -spp S -d D] [--seed S] [--scene-out 'dir/%03d.json'] [--ppm]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtmi.h"

struct Job {
    std::string tmpl, out_pattern = "frame_%03d.png", scene_pattern;
    bool dna = false, ppm = false;
    int frames = 1, first = 0, devices = 0;
    double step = 1.0;
    int w = 0, h = 0, spp = 0, depth = 0;
    unsigned long long seed = 2023;
};

static std::string fmt(const std::string &pattern, int k) {
    char buf[1024];
    snprintf(buf, sizeof buf, pattern.c_str(), k);
    return buf;
}

static bool render_frame(const Job &job, rt_scene *base, int k, int device, std::string &err) {
    rt_scene *sc;
    if (job.dna) {
        sc = rt_scene_dna(base, (double)k);  // angle = frame index (dna.py:17: range(0, 360, 1))
    } else {
        sc = rt_scene_clone(base);
        // frame k carries k+1 increments: the scripts bump the angle before writing frame k
        if (sc && rt_scene_rotate_cylinders(sc, job.step * (k + 1)) < 0) {
            rt_scene_free(sc);
            sc = nullptr;
        }
    }
    if (!sc) {
        err = rt_last_error();
        return false;
    }
    const std::string out = fmt(job.out_pattern, k);
    rt_scene_set_output_file(sc, out.c_str());
    if (!job.scene_pattern.empty()) {  // the per-frame scene file the scripts leave behind
        size_t n = rt_scene_to_json(sc, nullptr, 0);
        std::vector<char> buf(n);
        rt_scene_to_json(sc, buf.data(), n);
        if (FILE *fp = fopen(fmt(job.scene_pattern, k).c_str(), "w")) {
            fwrite(buf.data(), 1, n - 1, fp);
            fclose(fp);
        }
    }
    rt_scene_info info;
    rt_scene_get_info(sc, &info);
    std::vector<float> img((size_t)info.width * info.height * 3);
    rt_opts o;
    rt_opts_default(&o);
    o.seed = job.seed;
    o.device = device;
    rt_stats st;
    int rc = rt_render_hip(sc, &o, img.data(), &st);
    if (rc == RT_OK)
        rc = job.ppm ? rt_write_ppm(out.c_str(), img.data(), info.width, info.height, info.samples_per_pixel)
                     : rt_write_png(out.c_str(), img.data(), info.width, info.height, info.samples_per_pixel, 0);
    if (rc != RT_OK) err = rt_last_error();
    else fprintf(stderr, "frame %d on device %d: %.2f ms -> %s\n", k, device, st.kernel_ms, out.c_str());
    rt_scene_free(sc);
    return rc == RT_OK;
}

int main(int argc, char **argv) {
    Job job;
    for (int i = 1; i < argc; ++i) {
        auto need = [&](const char *flag) -> const char * {
            if (i + 1 >= argc) {
                fprintf(stderr, "%s needs a value\n", flag);
                exit(2);
            }
            return argv[++i];
        };
        if (!strcmp(argv[i], "--template")) job.tmpl = need("--template");
        else if (!strcmp(argv[i], "--dna")) job.dna = true;
        else if (!strcmp(argv[i], "--ppm")) job.ppm = true;
        else if (!strcmp(argv[i], "--frames")) job.frames = atoi(need("--frames"));
        else if (!strcmp(argv[i], "--first")) job.first = atoi(need("--first"));
        else if (!strcmp(argv[i], "--step")) job.step = atof(need("--step"));
        else if (!strcmp(argv[i], "--devices")) job.devices = atoi(need("--devices"));
        else if (!strcmp(argv[i], "--out")) job.out_pattern = need("--out");
        else if (!strcmp(argv[i], "--scene-out")) job.scene_pattern = need("--scene-out");
        else if (!strcmp(argv[i], "--seed")) job.seed = strtoull(need("--seed"), nullptr, 0);
        else if (!strcmp(argv[i], "-w")) job.w = atoi(need("-w"));
        else if (!strcmp(argv[i], "-h")) job.h = atoi(need("-h"));
        else if (!strcmp(argv[i], "-spp")) job.spp = atoi(need("-spp"));
        else if (!strcmp(argv[i], "-d")) job.depth = atoi(need("-d"));
        else {
            fprintf(stderr,
                    "usage: %s (--template scene.json | --dna [--template base.json]) --frames N [--first K] [--step DEG]\n"
                    "          --out 'dir/frame_%%03d.png' [--scene-out 'dir/%%03d.json'] [--devices N] [--ppm]\n"
                    "          [-w W -h H -spp S -d D] [--seed S]\n",
                    argv[0]);
            return 2;
        }
    }
    rt_scene *base = nullptr;
    if (!job.tmpl.empty()) {
        base = rt_scene_load_json(job.tmpl.c_str());
        if (!base) {
            fprintf(stderr, "rtmi-frames: %s\n", rt_last_error());
            return 1;
        }
    } else if (!job.dna) {
        fprintf(stderr, "rtmi-frames: --template is required unless --dna\n");
        return 2;
    } else {
        base = rt_scene_dna(nullptr, 0.0);  // carries basic_scene.json's camera, background and size
    }
    if (rt_scene_override(base, job.w, job.h, job.spp, job.depth) != RT_OK) {
        fprintf(stderr, "rtmi-frames: %s\n", rt_last_error());
        return 1;
    }
    int ndev = rt_device_count();
    if (ndev < 1) {
        fprintf(stderr, "rtmi-frames: %s\n", ndev < 0 ? rt_last_error() : "no HIP device");
        return 1;
    }
    if (job.devices > 0 && job.devices < ndev) ndev = job.devices;
    const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    std::atomic<int> failures{0};
    std::vector<std::thread> workers;
    for (int dev = 0; dev < ndev; ++dev) {
        workers.emplace_back([&, dev]() {
            // frame k -> device k mod N, in order: the batches of 8 of blue.py:23-32
            for (int k = job.first + dev; k < job.first + job.frames; k += ndev) {
                std::string err;
                if (!render_frame(job, base, k, dev, err)) {
                    fprintf(stderr, "rtmi-frames: frame %d failed: %s\n", k, err.c_str());
                    failures++;
                    return;
                }
            }
        });
    }
    for (auto &t : workers) t.join();
    const double t1 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    fprintf(stderr, "total time: %fs for %d frame(s) on %d device(s)\n", t1 - t0, job.frames, ndev);  // dna.py:112
    rt_scene_free(base);
    return failures ? 1 : 0;
}
