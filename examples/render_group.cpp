// render_group.cpp -- the frontend of render_scene.cpp on ALL GPUs of the node, one process: tiles sharded round-robin
// over the devices, one exchange step per frame (the members' tiles gathered over RCCL) onto device 0 (include/rayzen_hip.h: rz_group_*).  Also renders the
// same frame on a single context and checks that the two images are bit-identical (they must be: disjoint tiles,
// zeros elsewhere).
//
//   g++ -std=c++17 -O2 -Iinclude -Irayzen_amd/csrc/host examples/render_group.cpp
//       -Lrayzen_amd/lib -lrayzen_host -lrayzen_hip -Wl,-rpath,$PWD/rayzen_amd/lib -o render_group
//   ./render_group [ndev width height spp]        (ndev 0 = every visible device)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "RayZenScene.h"
#include "Renderer.h"
#include "rayzen_host.h"

using namespace rayzen;

int main(int argc, char** argv) {
    int ndev = argc > 1 ? std::atoi(argv[1]) : 0;
    const int W = argc > 2 ? std::atoi(argv[2]) : 640, H = argc > 3 ? std::atoi(argv[3]) : 360;
    const int spp = argc > 4 ? std::atoi(argv[4]) : 8;

    Scene scene;
    scene.camera = Camera(vec3(0.0f, 2.5f, 10.0f), vec3(0.0f, 0.0f, -1.0f), vec3(0.0f, 1.0f, 0.0f), 70.0f,
                          float(W) / float(H), 0.1f, 100.0f);
    scene.materials = {Material(vec3(0.8f, 0.3f, 0.3f), 0.0f, 1.0f, 0.0f, 0.0f, 1.5f),
                       Material(vec3(0.6f, 0.4f, 0.2f), 0.0f, 0.9f, 0.2f, 0.0f, 1.5f)};
    scene.lights.push_back(Light(vec4{5.0f, 5.0f, 5.0f, 1.0f}, vec3(1.0f), 300.0f));
    scene.lights.push_back(Light(vec4{0.8f, 1.4f, 0.3f, 0.0f}, vec3(1.0f), 2.0f));
    auto floor = std::make_shared<Mesh>(), blob = std::make_shared<Mesh>();
    floor->triangles.resize(12);
    rzh_make_cube(1, reinterpret_cast<rz_triangle*>(floor->triangles.data()), 12);
    blob->triangles.resize((size_t)12 * 24 * 24);
    rzh_make_blob(24, 2.8f, 1u, 0, reinterpret_cast<rz_triangle*>(blob->triangles.data()), (int)blob->triangles.size());
    scene.gameObjects.push_back(GameObject{floor, translate(scale(mat4(1.0f), vec3(8.0f, 0.5f, 8.0f)), vec3(0.0f, -3.0f, 0.0f))});
    scene.gameObjects.push_back(GameObject{blob, translate(mat4(1.0f), vec3(0.0f, 2.0f, 0.0f))});

    try {
        if (ndev <= 0) ndev = rz_device_count();
        if (ndev <= 0) { std::fprintf(stderr, "error: no HIP device\n"); return 1; }
        int ver = 0;
        rz_group_rccl_version(&ver);
        std::printf("%d device(s), RCCL %d\n", ndev, ver);
        GroupRenderer group(ndev);
        group.initializeSSBOs(scene);
        group.sendSceneDataToShader(scene, W, H, 4, spp);
        group.draw();
        group.finish();
        std::vector<float> sharded = group.readFrame();

        Renderer single(0);
        single.initializeSSBOs(scene);
        single.sendSceneDataToShader(scene, W, H, 4, spp);
        single.draw();
        std::vector<float> whole = single.readAccum();
        const bool same = std::memcmp(sharded.data(), whole.data(), whole.size() * sizeof(float)) == 0;
        std::printf("%dx%d, %d spp over %d rank(s): reduced frame %s the single-GPU frame\n", W, H, spp, group.size(),
                    same ? "is bit-identical to" : "DIFFERS from");
        return same ? 0 : 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}
