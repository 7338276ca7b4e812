// Host scene model: the reference's object graph (gpu-version/parser.hpp:16-32
// `struct scene` + the hittable/material/mytexture class hierarchy) flattened to
// POD tables that one hipMemcpy can upload.  The reference instead re-`new`s every
// object on the device from a <<<1,1>>> kernel to get device vtables
// (gpu-version/main.cu:374-446); there are no virtual calls here, so that step is
// designed away.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/rtmi.h"

namespace rtmi {

struct CameraParams {
    double lookfrom[3] = {0, 0, 1}, lookat[3] = {0, 0, 0}, vup[3] = {0, 1, 0};
    double vfov = 40.0;
    double aspect = 0.0;      // <= 0: width / height
    double aperture = 0.0;
    double focus_dist = 0.0;  // <= 0: |lookfrom - lookat|
};

// how a cylinder's transform was specified (kept for JSON serialisation)
struct CylinderXform {
    bool has_rotate = false, has_translate = false;
    double axis[3] = {0, 0, 1};
    double degrees = 0.0;
    double offset[3] = {0, 0, 0};
};

// pixels of an image texture (taichi-version/material.py:96-110): rows x cols texels, R G B bytes
struct SceneImage {
    int rows = 0, cols = 0;
    std::vector<uint8_t> rgb;
    std::string file;  // where it was read from ("" = given inline), for serialisation
};

struct DeviceSceneCache;  // owned by the render module

struct Scene {
    int width = 400, height = 225, spp = 100, max_depth = 50;
    float background[3] = {0, 0, 0};
    uint32_t flags = 0;
    float rr_p = 0.0f;  // Russian-roulette survival probability per bounce, 0 = off
    std::string output_file = "main.png";  // parser.hpp:566-567 default
    CameraParams cam;
    std::vector<rt_prim> prims;
    std::vector<CylinderXform> xforms;  // parallel to prims (meaningful for cylinders)
    std::vector<rt_material> mats;
    std::vector<rt_texture> texs;
    std::vector<SceneImage> images;  // referenced by RT_TEX_IMAGE textures (c0[0] = index)
    uint64_t version = 1;  // bumped on every mutation; invalidates device caches
    std::shared_ptr<DeviceSceneCache> dev;

    void touch() { ++version; }
};

// derive the camera frame (camera.cuh:9-29) in fp64, round once to fp32
void derive_camera(const Scene &s, rt_camera *out);

// all return RT_OK or an rt_status, message via set_error()
// base_dir: directory that relative "file" entries (image textures, meshes) are resolved against (NULL: the cwd)
int scene_from_json(const char *text, size_t len, Scene &out, const char *base_dir = nullptr);
std::string scene_to_json(const Scene &s);
void scene_rtiow(Scene &out, uint32_t seed, int width, int height, int spp, int max_depth);
int scene_validate(const Scene &s);

int add_cylinder(Scene &s, float radius, float zmin, float zmax, int material, const double *axis,
                 double degrees, const double *offset);

// -> texture id / triangles added / prim id, or -rt_status
int add_image_texture(Scene &s, int rows, int cols, const uint8_t *rgb, const std::string &file);
int add_image_texture_file(Scene &s, const char *path);
int add_triangle(Scene &s, const float v1[3], const float v2[3], const float v3[3], const float uv1[2], const float uv2[2],
                 const float uv3[2], int material);
int add_obj(Scene &s, const char *path, int material, float scale, const float matrix[9], const float translate[3]);

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
const char *get_error();

}  // namespace rtmi

struct rt_scene {
    rtmi::Scene s;
};
