// Scene.cpp -- Scene -> the six geometry arrays of RayZen's SSBOs.
//   SceneBuffers::build          <- RayZen/src/main.cpp:941-1035 (initializeSSBOs, no disk cache)
//   SceneBuffers::updateDynamic  <- RayZen/src/main.cpp:1138-1194 (updateDynamicBVHAndSSBOs)
//   worldRootNode                <- RayZen/src/main.cpp:974-993
//   Camera::rotate               <- RayZen/include/Camera.h:72-90
#include "RayZenScene.h"

#include <algorithm>
#include <map>

namespace rayzen {

void Camera::rotate(float offsetX, float offsetY) {
    yaw += offsetX * sensitivity;
    pitch += offsetY * sensitivity;
    if (pitch > 89.0f) pitch = 89.0f;
    if (pitch < -89.0f) pitch = -89.0f;
    vec3 direction;
    direction.x = std::cos(radians(yaw)) * std::cos(radians(pitch));
    direction.y = std::sin(radians(pitch));
    direction.z = std::sin(radians(yaw)) * std::cos(radians(pitch));
    target = normalize(direction);
    vec3 right = normalize(cross(target, vec3(0.0f, 1.0f, 0.0f)));
    up = normalize(cross(right, target));
    updateViewMatrix();
}

BVHNode worldRootNode(const BVHNode& meshRoot, const mat4& transform) {
    const vec3& lo = meshRoot.boundsMin;
    const vec3& hi = meshRoot.boundsMax;
    vec3 bmin(1e30f), bmax(-1e30f);
    for (int c = 0; c < 8; ++c) {
        vec4 corner{(c & 4) ? hi.x : lo.x, (c & 2) ? hi.y : lo.y, (c & 1) ? hi.z : lo.z, 1.0f};
        vec4 t = transform * corner;
        vec3 tc(t.x, t.y, t.z);
        bmin = vmin(bmin, tc);
        bmax = vmax(bmax, tc);
    }
    BVHNode r = meshRoot;
    r.boundsMin = bmin;
    r.boundsMax = bmax;
    return r;
}

bool SceneBuffers::build(const Scene& scene, bool shareMeshes) {
    allTriangles.clear(); allBLASNodes.clear(); allBLASTriIndices.clear();
    meshInstances.clear(); tlasNodes.clear(); tlasTriIndices.clear(); blasRoots.clear();
    maxBLASDepth = 1;

    struct Placed { int nodeOffset, triOffset, triBase; BVHNode root; };
    std::map<const Mesh*, BVH> built;          // a BLAS depends only on the mesh: build each once
    std::map<const Mesh*, Placed> placed;      // shareMeshes: where the single copy lives
    std::vector<BVHNode> worldRootNodes;
    worldRootNodes.reserve(scene.gameObjects.size());

    for (size_t i = 0; i < scene.gameObjects.size(); ++i) {
        const GameObject& obj = scene.gameObjects[i];
        const Mesh* mesh = obj.mesh.get();
        static const Mesh kEmpty;
        if (!mesh) mesh = &kEmpty;
        auto it = built.find(mesh);
        if (it == built.end()) {
            it = built.emplace(mesh, BVH{}).first;
            if (blasBuilder) { if (!blasBuilder(*mesh, it->second)) return false; }
            else it->second.buildBLAS(mesh->triangles);
            maxBLASDepth = std::max(maxBLASDepth, it->second.depth());
        }
        const BVH& blas = it->second;
        Placed where;
        auto pit = placed.find(mesh);
        if (shareMeshes && pit != placed.end()) {
            where = pit->second;
        } else {
            where.nodeOffset = (int)allBLASNodes.size();
            where.triOffset = (int)allBLASTriIndices.size();
            where.triBase = (int)allTriangles.size();
            where.root = blas.nodes[0];
            allTriangles.insert(allTriangles.end(), mesh->triangles.begin(), mesh->triangles.end());
            allBLASNodes.insert(allBLASNodes.end(), blas.nodes.begin(), blas.nodes.end());
            allBLASTriIndices.insert(allBLASTriIndices.end(), blas.triIndices.begin(), blas.triIndices.end());
            placed[mesh] = where;
        }
        worldRootNodes.push_back(worldRootNode(where.root, obj.transform));
        blasRoots.push_back(where.root);
        BVHInstance inst;
        inst.blasNodeOffset = where.nodeOffset;
        inst.blasTriOffset = where.triOffset;
        inst.globalTriOffset = where.triBase;
        inst.meshIndex = (int)i;
        inst.transform = obj.transform;
        inst.inverseTransform = inverse(obj.transform);
        meshInstances.push_back(inst);
    }
    BVH tlas;
    tlas.buildTLAS(meshInstances, worldRootNodes);
    tlasNodes = tlas.nodes;
    tlasTriIndices = tlas.triIndices;
    tlasDepth = tlas.depth();
    return true;
}

void SceneBuffers::updateDynamic(const Scene& scene) {
    size_t n = std::min(meshInstances.size(), scene.gameObjects.size());
    std::vector<BVHNode> worldRootNodes(meshInstances.size());
    for (size_t i = 0; i < meshInstances.size(); ++i) {
        if (i < n) {
            meshInstances[i].transform = scene.gameObjects[i].transform;
            meshInstances[i].inverseTransform = inverse(scene.gameObjects[i].transform);
        }
        worldRootNodes[i] = worldRootNode(blasRoots[i], meshInstances[i].transform);
    }
    BVH tlas;
    tlas.buildTLAS(meshInstances, worldRootNodes);
    tlasNodes = tlas.nodes;
    tlasTriIndices = tlas.triIndices;
    tlasDepth = tlas.depth();
}

}  // namespace rayzen
