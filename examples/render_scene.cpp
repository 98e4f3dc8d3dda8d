// render_scene.cpp -- a C++ frontend written against the host API exactly the way RayZen's main.cpp is written
// against its own (src/main.cpp:327-388 scene literal, :388 initializeSSBOs, :572 updateDynamicBVHAndSSBOs,
// :601 sendSceneDataToShader, :637 glDrawArrays), with OpenGL replaced by the rayzen_hip C-ABI.
//
//   g++ -std=c++17 -O2 -Iinclude -Irayzen_amd/csrc/host examples/render_scene.cpp
//       -Lrayzen_amd/lib -lrayzen_host -lrayzen_hip -Wl,-rpath,$PWD/rayzen_amd/lib -o render_scene
//   ./render_scene out.ppm [width height spp frames [device]]     (device: BLAS built and kept on the GPU, rz_build_geometry)
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>

#include "RayZenScene.h"
#include "Renderer.h"
#include "rayzen_host.h"

using namespace rayzen;

static std::shared_ptr<Mesh> cubeMesh(int material) {
    auto m = std::make_shared<Mesh>();
    m->triangles.resize(12);
    rzh_make_cube(material, reinterpret_cast<rz_triangle*>(m->triangles.data()), 12);
    return m;
}
static std::shared_ptr<Mesh> blobMesh(int n, float radius, int material) {
    auto m = std::make_shared<Mesh>();
    m->triangles.resize((size_t)12 * n * n);
    rzh_make_blob(n, radius, 1u, material, reinterpret_cast<rz_triangle*>(m->triangles.data()), (int)m->triangles.size());
    return m;
}

int main(int argc, char** argv) {
    const char* out = argc > 1 ? argv[1] : "out.ppm";
    const int W = argc > 2 ? std::atoi(argv[2]) : 640, H = argc > 3 ? std::atoi(argv[3]) : 360;
    const int spp = argc > 4 ? std::atoi(argv[4]) : 16, frames = argc > 5 ? std::atoi(argv[5]) : 3;
    const bool onDevice = argc > 6 && std::string(argv[6]) == "device";

    Scene scene;
    scene.camera = Camera(vec3(0.0f, 2.5f, 10.0f), vec3(0.0f, 0.0f, -1.0f), vec3(0.0f, 1.0f, 0.0f), 70.0f,
                          float(W) / float(H), 0.1f, 100.0f);
    scene.materials = {Material(vec3(0.8f, 0.3f, 0.3f), 0.0f, 1.0f, 0.0f, 0.0f, 1.5f),     // main.cpp:342-353
                       Material(vec3(0.1f, 0.7f, 0.1f), 1.0f, 0.35f, 0.3f, 0.0f, 1.5f),
                       Material(vec3(1.0f), 1.0f, 0.05f, 1.0f, 0.0f, 1.5f),
                       Material(vec3(0.85f, 0.95f, 1.0f), 0.0f, 0.02f, 0.05f, 0.94f, 1.5f),
                       Material(vec3(0.6f, 0.4f, 0.2f), 0.0f, 0.9f, 0.2f, 0.0f, 1.5f)};
    scene.lights.push_back(Light(vec4{5.0f, 5.0f, 5.0f, 1.0f}, vec3(1.0f), 300.0f));     // main.cpp:356-357
    scene.lights.push_back(Light(vec4{0.8f, 1.4f, 0.3f, 0.0f}, vec3(1.0f), 2.0f));
    auto floor = cubeMesh(4), bunny = blobMesh(40, 2.8f, 0), glass = blobMesh(12, 1.2f, 3);
    scene.gameObjects.push_back(GameObject{floor, translate(scale(mat4(1.0f), vec3(8.0f, 0.5f, 8.0f)), vec3(0.0f, -3.0f, 0.0f))});
    scene.gameObjects.push_back(GameObject{bunny, translate(mat4(1.0f), vec3(0.0f, 2.0f, 0.0f))});
    scene.gameObjects.push_back(GameObject{glass, translate(mat4(1.0f), vec3(4.5f, 0.6f, 3.0f))});

    try {
        Renderer renderer(0);
        if (onDevice) renderer.initializeSSBOsOnDevice(scene);             // the same, geometry half on the GPU
        else renderer.initializeSSBOs(scene, /*shareMeshes=*/true);        // main.cpp:388
        for (int frame = 0; frame < frames; ++frame) {                     // main.cpp:408 render loop
            scene.gameObjects[2].transform = translate(mat4(1.0f), vec3(4.5f - 0.5f * frame, 0.6f, 3.0f));
            renderer.updateDynamicBVHAndSSBOs(scene);                      // main.cpp:572
            renderer.sendSceneDataToShader(scene, W, H, 5, spp);           // main.cpp:601
            renderer.draw();                                               // main.cpp:637
            renderer.finish();
            std::printf("frame %d: %.3f ms on the GPU, %.1f Msamples/s\n", frame, renderer.lastRenderMs(),
                        double(W) * H * spp / (renderer.lastRenderMs() * 1e3));
        }
        std::vector<uint8_t> px = renderer.resolveRGBA8();
        FILE* f = std::fopen(out, "wb");
        if (!f) { std::perror(out); return 1; }
        std::fprintf(f, "P6\n%d %d\n255\n", W, H);
        for (int y = H - 1; y >= 0; --y)                                   // row 0 is the bottom row
            for (int x = 0; x < W; ++x) std::fwrite(&px[((size_t)y * W + x) * 4], 1, 3, f);
        std::fclose(f);
        std::printf("wrote %s\n", out);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
