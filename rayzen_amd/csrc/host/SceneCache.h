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

// saveBVHToFile / loadBVHFromFile (main.cpp:117-125): <base>.nodes.bin + <base>.tris.bin
bool saveBVHToFile(const std::string& base, const BVH& bvh);
bool loadBVHFromFile(const std::string& base, BVH& bvh);

// What initializeSSBOs (main.cpp:897-1060) found on disk.
struct CacheReport {
    bool ssboLoaded = false;        // the six ssbo_v2_* files were used (main.cpp:914-939)
    bool ssboInvalidated = false;   // ... they were there, but the object count had changed (main.cpp:929-934)
    int blasLoaded = 0;             // meshes whose mesh<i>.nodes.bin / .tris.bin were used (main.cpp:956-961)
    int blasBuilt = 0;              // meshes built from scratch (and written back, main.cpp:962-967)
    bool tlasLoaded = false;        // scene_tlas.* + instances.bin were used (main.cpp:1012-1018)
};

// initializeSSBOs WITH RayZen's disk cache, step for step (main.cpp:897-1060), including its quirks:
//  * the ssbo_v2_* set is invalidated only by a changed object COUNT; when it is used, transforms / inverses / mesh
//    indices are refreshed from the scene (main.cpp:1054-1060) but the cached TLAS is kept as it is;
//  * a BLAS is cached per OBJECT index (mesh<i>), every object gets its own copy of its mesh (no sharing);
//  * scene_tlas.* + instances.bin are used only if every BLAS came from the cache, and loading instances.bin replaces
//    the instance records just assembled -- transforms included -- with the cached ones;
//  * whatever was built is written back, then the ssbo_v2_* set is written.
// `dir` plays the role of "bvh_cache/v2/" (created if missing).  false only on a failed BLAS builder.
bool initializeSSBOsCached(const Scene& scene, const std::string& dir, bool forceRebuildBVH, SceneBuffers& out,
                           CacheReport* report = nullptr);

}  // namespace rayzen
