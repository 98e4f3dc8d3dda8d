// RayZenScene.h -- host-side mirror of RayZen's scene API (GLM-free).
//
// Same class names, member names and meaning as the reference headers, so a
// frontend written against RayZen reads the same here:
//   Triangle, Mesh        <- RayZen/include/Mesh.h:9-23
//   BVHNode, BVHInstance,
//   BVHSplitMethod, BVH   <- RayZen/include/BVH.h:7-43
//   Material              <- RayZen/include/Material.h:6-18
//   Light                 <- RayZen/include/Light.h:6-33
//   Camera                <- RayZen/include/Camera.h:7-101
//   GameObject            <- RayZen/include/GameObject.h:6-10
//   Scene                 <- RayZen/include/Scene.h:11-20
// The POD element types ARE the C-ABI structs of include/rayzen_hip.h
// (static_asserted below), i.e. the SSBO byte layouts of the reference.
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "rayzen_hip.h"
#include "rz_linalg.h"

namespace rayzen {

struct alignas(16) Triangle {
    vec3 v0; float pad0 = 0.0f;
    vec3 v1; float pad1 = 0.0f;
    vec3 v2; float pad2 = 0.0f;
    int materialIndex = 0;
};
static_assert(sizeof(Triangle) == 64 && sizeof(Triangle) == sizeof(rz_triangle), "Triangle must be the 64-B SSBO element");

class Mesh {
public:
    std::vector<Triangle> triangles;
    // Mesh.cpp:6-50.  Returns false (and leaves the mesh empty) if the file cannot be opened.
    bool loadFromOBJ(const std::string& filename, int materialIndex);
};

struct BVHNode {
    vec3 boundsMin; int leftFirst = 0;
    vec3 boundsMax; int count = 0;
};
static_assert(sizeof(BVHNode) == 32 && sizeof(BVHNode) == sizeof(rz_bvh_node), "BVHNode must be the 32-B SSBO element");

struct BVHInstance {
    int blasNodeOffset = 0;
    int blasTriOffset = 0;
    int meshIndex = 0;
    int globalTriOffset = 0;
    mat4 transform;
    mat4 inverseTransform;
};
static_assert(sizeof(BVHInstance) == 144 && sizeof(BVHInstance) == sizeof(rz_bvh_instance), "BVHInstance must be the 144-B SSBO element");

enum class BVHSplitMethod { Midpoint, SAH };

class BVH {
public:
    std::vector<BVHNode> nodes;
    std::vector<int> triIndices;
    std::vector<BVHInstance> instances;
    BVHSplitMethod splitMethod = BVHSplitMethod::SAH;
    // BVH.cpp:99-175.  Output arrays are byte-identical to the reference
    // algorithm's; the construction here pre-sorts once per axis and
    // partitions stably down the tree (O(N log N) instead of O(N log^2 N)).
    void buildBLAS(const std::vector<Triangle>& tris);
    void buildBLAS(const Triangle* tris, int n);
    // BVH.cpp:178-240.
    void buildTLAS(const std::vector<BVHInstance>& meshInstances, const std::vector<BVHNode>& meshRootNodes);
    // Longest root-to-leaf path, in nodes (root alone = 1).  The render
    // library sizes its LDS traversal stacks from this.
    int depth() const;
    // (BVH::saveToFile / loadFromFile, BVH.cpp:242-265, are dead code in the reference -- never called -- and are
    //  intentionally not mirrored; the cache formats RayZen does use live in SceneCache.h.)
};

struct Material {
    vec3 albedo;
    float metallic;
    float roughness;
    float reflectivity;
    float transparency;
    float ior;
    Material(const vec3& albedo_, float metallic_, float roughness_, float reflectivity_ = 0.0f,
             float transparency_ = 0.0f, float ior_ = 1.5f)
        : albedo(albedo_), metallic(metallic_), roughness(roughness_), reflectivity(reflectivity_),
          transparency(transparency_), ior(ior_) {}
};
static_assert(sizeof(Material) == 32 && sizeof(Material) == sizeof(rz_material), "Material must be the 32-B SSBO element");

class Light {
public:
    vec4 positionOrDirection;   // w == 1: point light, else directional
    vec3 color;
    float power;
    Light(const vec4& pd, const vec3& c, float p) : positionOrDirection(pd), color(c), power(p) {}
    bool isPointLight() const { return positionOrDirection.w == 1.0f; }
    vec3 getPosition() const { return {positionOrDirection.x, positionOrDirection.y, positionOrDirection.z}; }
    vec3 getDirection() const { return isPointLight() ? vec3(0.0f) : normalize(getPosition()); }
    vec3 getColor() const { return color; }
};
static_assert(sizeof(Light) == 32 && sizeof(Light) == sizeof(rz_light), "Light must be the 32-B SSBO element");

class Camera {
public:
    vec3 position{0.0f, 0.0f, 3.0f};
    vec3 target{0.0f, 0.0f, -1.0f};    // a DIRECTION: the view looks at position + target (Camera.h:43)
    vec3 up{0.0f, 1.0f, 0.0f};
    mat4 viewMatrix;
    mat4 projectionMatrix;
    float fov = 45.0f;                 // degrees
    float aspectRatio = 800.0f / 600.0f;
    float nearClip = 0.1f;
    float farClip = 100.0f;
    float speed = 1.0f;
    float sensitivity = 0.1f;
    float yaw = -90.0f;
    float pitch = 0.0f;

    Camera() { updateViewMatrix(); updateProjectionMatrix(); }
    Camera(vec3 position_, vec3 target_, vec3 up_, float fov_, float aspect_, float near_, float far_)
        : position(position_), target(target_), up(up_), fov(fov_), aspectRatio(aspect_), nearClip(near_), farClip(far_) {
        updateViewMatrix(); updateProjectionMatrix();
    }
    void updateViewMatrix() { viewMatrix = lookAt(position, position + target, up); }
    void updateProjectionMatrix() { projectionMatrix = perspective(radians(fov), aspectRatio, nearClip, farClip); }
    void moveForward(float dt) { position = position + target * (speed * dt); }
    void moveBackward(float dt) { position = position - target * (speed * dt); }
    void moveLeft(float dt) { position = position - normalize(cross(target, up)) * (speed * dt); }
    void moveRight(float dt) { position = position + normalize(cross(target, up)) * (speed * dt); }
    void rotate(float offsetX, float offsetY);
};

struct GameObject {
    std::shared_ptr<Mesh> mesh;
    mat4 transform;
};

class Scene {
public:
    Camera camera;
    std::vector<Material> materials;
    std::vector<Light> lights;
    std::vector<GameObject> gameObjects;
};

// The six geometry arrays main.cpp builds for its SSBOs, plus how to keep
// them current.  build() restates initializeSSBOs (main.cpp:941-1035, without
// the disk cache); updateDynamic() restates updateDynamicBVHAndSSBOs
// (main.cpp:1138-1194): instance transforms re-read, world AABBs and the
// TLAS rebuilt; BLAS and triangles untouched.
struct SceneBuffers {
    std::vector<Triangle> allTriangles;       // binding 0
    std::vector<BVHNode> allBLASNodes;        // binding 7
    std::vector<int> allBLASTriIndices;       // binding 8
    std::vector<BVHInstance> meshInstances;   // binding 9
    std::vector<BVHNode> tlasNodes;           // binding 5
    std::vector<int> tlasTriIndices;          // binding 6
    std::vector<BVHNode> blasRoots;           // object-space root box of each instance's BLAS
    int maxBLASDepth = 1;
    int tlasDepth = 1;

    // shareMeshes = false reproduces the reference exactly (one BLAS copy and
    // one triangle copy per GameObject, main.cpp:951-1007); true stores one
    // BLAS per distinct Mesh and points every instance of it at that copy.
    // false only when `blasBuilder` is set and fails (nothing is built then; the host builder is NOT used instead).
    bool build(const Scene& scene, bool shareMeshes = false);
    void updateDynamic(const Scene& scene);

    // Optional replacement for BVH::buildBLAS, e.g. rz_build_blas of include/rayzen_hip.h bound to a context (the
    // device builder, byte-identical output).  Fills `out.nodes` / `out.triIndices`; returns false on failure.
    std::function<bool(const Mesh& mesh, BVH& out)> blasBuilder;
};

// World AABB of a BLAS root under a transform (main.cpp:974-993).
BVHNode worldRootNode(const BVHNode& meshRoot, const mat4& transform);

}  // namespace rayzen
