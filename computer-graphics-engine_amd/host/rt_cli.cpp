// rt_cli.cpp — headless batch renderer on top of the C ABI, the counterpart of the
// reference's command-line mode (src/main.cpp:478-535: load scene, build BVH, render every
// configured camera, write one BMP each, print the timings). Options replace the TOML file:
//
//   rt_cli --data DIR --scene cornell|monkey|teapot|cube|cube_textured|single_triangle|spheres|custom
//          | --obj FILE [--normalize]
//          [--size W H] [--depth N] [--features shading,recursive,hard_shadow,soft_shadow,
//            normal_interp,texture,bvh,sah,bilinear] [--camera FOV DIST RX RY]... [--out PREFIX]
//
// Several --camera options render several images (the reference spawns one thread per
// camera; here the frames go to the GPU one after the other).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_whitted.h"

extern "C" {
typedef struct rt_loaded_scene rt_loaded_scene;
}

static int die(const char* what, const char* msg)
{
    std::fprintf(stderr, "rt_cli: %s: %s\n", what, msg);
    return 1;
}

int main(int argc, char** argv)
{
    std::string data = ".", scene = "", obj = "", out = "render";
    bool normalize = false;
    int W = 800, H = 800, depth = 5; // config.h: windowSize 800x800; render.cpp:318: depth 5
    rt_features f;
    memset(&f, 0, sizeof(f));
    f.enableShading = f.enableAccelStructure = 1;
    struct Cam {
        float fov, dist, rx, ry;
    };
    std::vector<Cam> cams;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto need = [&](int n) { return i + n < argc; };
        if (a == "--data" && need(1)) data = argv[++i];
        else if (a == "--scene" && need(1)) scene = argv[++i];
        else if (a == "--obj" && need(1)) obj = argv[++i];
        else if (a == "--normalize") normalize = true;
        else if (a == "--out" && need(1)) out = argv[++i];
        else if (a == "--depth" && need(1)) depth = std::atoi(argv[++i]);
        else if (a == "--size" && need(2)) W = std::atoi(argv[++i]), H = std::atoi(argv[++i]);
        else if (a == "--camera" && need(4)) {
            Cam c { (float)std::atof(argv[i + 1]), (float)std::atof(argv[i + 2]), (float)std::atof(argv[i + 3]), (float)std::atof(argv[i + 4]) };
            cams.push_back(c);
            i += 4;
        } else if (a == "--features" && need(1)) {
            memset(&f, 0, sizeof(f));
            std::string s = argv[++i];
            s += ',';
            for (size_t p = 0, q; (q = s.find(',', p)) != std::string::npos; p = q + 1) {
                const std::string k = s.substr(p, q - p);
                if (k == "shading") f.enableShading = 1;
                else if (k == "recursive") f.enableRecursive = 1;
                else if (k == "hard_shadow") f.enableHardShadow = 1;
                else if (k == "soft_shadow") f.enableSoftShadow = 1;
                else if (k == "normal_interp") f.enableNormalInterp = 1;
                else if (k == "texture") f.enableTextureMapping = 1;
                else if (k == "bvh") f.enableAccelStructure = 1;
                else if (k == "sah") f.enableBvhSahBinning = 1;
                else if (k == "bilinear") f.enableBilinearTextureFiltering = 1;
                else if (!k.empty()) return die("unknown feature", k.c_str());
            }
        } else {
            return die("unknown or incomplete option", a.c_str());
        }
    }
    if (cams.empty())
        cams.push_back(Cam { 50.0f, 3.0f, 20.0f, 20.0f }); // CameraConfig defaults, src/config.h:16-21

    static const char* names[] = { "single_triangle", "cube", "cube_textured", "cornell", "cornell_parallelogram",
        "monkey", "teapot", "dragon", "spheres", "custom" };
    rt_loaded_scene* loaded = nullptr;
    int rc;
    if (!obj.empty()) {
        rt_light_desc light; // loadSceneFromFile gets the config's lights; default: the presets' usual point light
        memset(&light, 0, sizeof(light));
        light.kind = RT_LIGHT_POINT;
        light.v[0][0] = -1, light.v[0][1] = 1, light.v[0][2] = -1, light.v[1][0] = light.v[1][1] = light.v[1][2] = 1;
        rc = rt_load_obj(obj.c_str(), normalize ? 1 : 0, &light, 1, &loaded);
    } else {
        int type = -1;
        for (int t = 0; t < 10; t++)
            if (scene == names[t])
                type = t;
        if (type < 0)
            return die("unknown --scene", scene.c_str());
        rc = rt_load_preset(type, data.c_str(), &loaded);
    }
    if (rc != RT_OK)
        return die("loading the scene failed", rt_io_last_error());
    rt_scene_desc desc;
    rt_loaded_scene_desc(loaded, &desc);
    rt_scene* sc = nullptr;
    rt_bvh* bvh = nullptr;
    using clock = std::chrono::high_resolution_clock;
    const auto t0 = clock::now();
    if (rt_scene_create(&desc, &sc) != RT_OK || rt_bvh_build(sc, &f, &bvh) != RT_OK)
        return die("scene/BVH", rt_last_error());
    const auto t1 = clock::now();
    std::printf("BVH generation took %lld ms.\n", (long long)std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count());
    std::printf("levels %d, leaves %d\n", rt_bvh_num_levels(bvh), rt_bvh_num_leaves(bvh));

    const float deg = 0.01745329251994329576923690768489f; // glm::radians
    std::vector<float> frame((size_t)W * H * 3);
    const rt_render_params params { depth, 5, 25, 0, 3, 5, 3.0f, 0.005f, 0.3f, 0.4f, 0, 3, 40, 1.0f }; // the reference's defaults
    const auto r0 = clock::now();
    for (size_t i = 0; i < cams.size(); i++) {
        rt_camera cam;
        const float look[3] = { 0, 0, 0 }, rot[3] = { cams[i].rx * deg, cams[i].ry * deg, 0.0f };
        rt_camera_from_trackball(cams[i].fov * deg, float(W) / float(H), look, cams[i].dist, rot, &cam);
        if (rt_render(bvh, &cam, &f, &params, W, H, 0, H, frame.data(), nullptr) != RT_OK)
            return die("rt_render", rt_last_error());
        const std::string path = out + "_cam_" + std::to_string(i) + ".bmp";
        if (rt_write_bmp(path.c_str(), frame.data(), W, H) != RT_OK)
            return die("rt_write_bmp", rt_io_last_error());
        std::printf("Image %zu saved to %s\n", i, path.c_str());
    }
    const auto r1 = clock::now();
    std::printf("Rendering took %lld ms, %zu images rendered.\n",
        (long long)std::chrono::duration_cast<std::chrono::milliseconds>(r1 - r0).count(), cams.size());
    rt_bvh_destroy(bvh);
    rt_scene_destroy(sc);
    rt_loaded_scene_destroy(loaded);
    return 0;
}
