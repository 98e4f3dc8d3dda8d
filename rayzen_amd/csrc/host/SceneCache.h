// SceneCache.h -- RayZen's on-disk cache formats (bvh_cache/v2), so scenes preprocessed by RayZen can be consumed here
// and vice versa.  Format of every file (RayZen/src/main.cpp:94-115): a native `size_t` element count followed by
// count * sizeof(T) raw bytes of the POD array.  File set written by initializeSSBOs (main.cpp:897-1052):
//     <dir>/ssbo_v2_triangles.bin  Triangle[]      <dir>/ssbo_v2_blasnodes.bin  BVHNode[]
//     <dir>/ssbo_v2_blastris.bin   int[]           <dir>/ssbo_v2_instances.bin  BVHInstance[]
//     <dir>/ssbo_v2_tlasnodes.bin  BVHNode[]       <dir>/ssbo_v2_tlastris.bin   int[]
// plus per mesh  <dir>/mesh<i>.nodes.bin / .tris.bin  and  <dir>/scene_tlas.nodes.bin / .tris.bin, <dir>/instances.bin.
#pragma once
#include <cstdio>
#include <string>
#include <vector>

#include "RayZenScene.h"

namespace rayzen {

template <typename T>
bool saveVectorToFile(const std::string& filename, const std::vector<T>& vec) {
    FILE* f = std::fopen(filename.c_str(), "wb");
    if (!f) return false;
    size_t n = vec.size();
    bool ok = std::fwrite(&n, sizeof n, 1, f) == 1 && (n == 0 || std::fwrite(vec.data(), sizeof(T), n, f) == n);
    return (std::fclose(f) == 0) && ok;
}

template <typename T>
bool loadVectorFromFile(const std::string& filename, std::vector<T>& vec) {
    FILE* f = std::fopen(filename.c_str(), "rb");
    if (!f) return false;
    size_t n = 0;
    bool ok = std::fread(&n, sizeof n, 1, f) == 1;
    if (ok) {
        // refuse counts the file cannot hold (the reference trusts the header)
        long here = std::ftell(f);
        std::fseek(f, 0, SEEK_END);
        long end = std::ftell(f);
        std::fseek(f, here, SEEK_SET);
        ok = here >= 0 && end >= here && n <= (size_t)(end - here) / sizeof(T);
    }
    if (ok) {
        vec.resize(n);
        ok = n == 0 || std::fread(static_cast<void*>(vec.data()), sizeof(T), n, f) == n;
    }
    std::fclose(f);
    return ok;
}

// The six ssbo_v2_* files <-> SceneBuffers (depths and root boxes are recomputed on load).
bool saveSceneCache(const std::string& dir, const SceneBuffers& b);
bool loadSceneCache(const std::string& dir, SceneBuffers& b);

}  // namespace rayzen
