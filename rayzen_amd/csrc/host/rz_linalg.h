// rz_linalg.h -- the handful of vector/matrix operations RayZen's host code
// takes from GLM (include/Camera.h:42-48, src/main.cpp:378-384,1001), written
// out so the host side builds without GLM.  Column-major mat4, like GLM/GLSL:
// m[c*4 + r].  float32 throughout.
//
// The render path's C-ABI receives finished matrices, so the rounding of
// these helpers is not part of the compared path.
#pragma once
#include <cmath>
#include <cstring>

namespace rayzen {

struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float X, float Y, float Z) : x(X), y(Y), z(Z) {}
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator/(vec3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline vec3 operator-(vec3 a) { return {-a.x, -a.y, -a.z}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
inline vec3 normalize(vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
// glm::min/max: (b < a) ? b : a, component-wise
inline vec3 vmin(vec3 a, vec3 b) { return {b.x < a.x ? b.x : a.x, b.y < a.y ? b.y : a.y, b.z < a.z ? b.z : a.z}; }
inline vec3 vmax(vec3 a, vec3 b) { return {a.x < b.x ? b.x : a.x, a.y < b.y ? b.y : a.y, a.z < b.z ? b.z : a.z}; }

struct vec4 { float x = 0, y = 0, z = 0, w = 0; };

struct mat4 {
    float m[16];
    mat4() { std::memset(m, 0, sizeof m); m[0] = m[5] = m[10] = m[15] = 1.0f; }
    explicit mat4(float d) { std::memset(m, 0, sizeof m); m[0] = m[5] = m[10] = m[15] = d; }
    float& at(int col, int row) { return m[col * 4 + row]; }
    float at(int col, int row) const { return m[col * 4 + row]; }
    const float* data() const { return m; }
};

inline vec4 operator*(const mat4& a, vec4 v) {
    vec4 r;
    r.x = a.m[0] * v.x + a.m[4] * v.y + a.m[8] * v.z + a.m[12] * v.w;
    r.y = a.m[1] * v.x + a.m[5] * v.y + a.m[9] * v.z + a.m[13] * v.w;
    r.z = a.m[2] * v.x + a.m[6] * v.y + a.m[10] * v.z + a.m[14] * v.w;
    r.w = a.m[3] * v.x + a.m[7] * v.y + a.m[11] * v.z + a.m[15] * v.w;
    return r;
}
inline mat4 operator*(const mat4& a, const mat4& b) {
    mat4 r(0.0f);
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += a.at(k, row) * b.at(c, k);
            r.at(c, row) = s;
        }
    return r;
}

inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

// glm::translate(m, v): m * T(v)
inline mat4 translate(const mat4& m, vec3 v) {
    mat4 r = m;
    for (int row = 0; row < 4; ++row)
        r.at(3, row) = m.at(0, row) * v.x + m.at(1, row) * v.y + m.at(2, row) * v.z + m.at(3, row);
    return r;
}
// glm::scale(m, v): m * S(v)
inline mat4 scale(const mat4& m, vec3 v) {
    mat4 r = m;
    for (int row = 0; row < 4; ++row) {
        r.at(0, row) = m.at(0, row) * v.x;
        r.at(1, row) = m.at(1, row) * v.y;
        r.at(2, row) = m.at(2, row) * v.z;
    }
    return r;
}
// glm::rotate(m, angle, axis): m * R
inline mat4 rotate(const mat4& m, float angle, vec3 axis) {
    float c = std::cos(angle), s = std::sin(angle);
    vec3 a = normalize(axis);
    vec3 t = a * (1.0f - c);
    mat4 R;
    R.at(0, 0) = c + t.x * a.x; R.at(0, 1) = t.x * a.y + s * a.z; R.at(0, 2) = t.x * a.z - s * a.y;
    R.at(1, 0) = t.y * a.x - s * a.z; R.at(1, 1) = c + t.y * a.y; R.at(1, 2) = t.y * a.z + s * a.x;
    R.at(2, 0) = t.z * a.x + s * a.y; R.at(2, 1) = t.z * a.y - s * a.x; R.at(2, 2) = c + t.z * a.z;
    return m * R;
}
// glm::lookAt (right-handed)
inline mat4 lookAt(vec3 eye, vec3 center, vec3 up) {
    vec3 f = normalize(center - eye);
    vec3 s = normalize(cross(f, up));
    vec3 u = cross(s, f);
    mat4 r;
    r.at(0, 0) = s.x; r.at(1, 0) = s.y; r.at(2, 0) = s.z;
    r.at(0, 1) = u.x; r.at(1, 1) = u.y; r.at(2, 1) = u.z;
    r.at(0, 2) = -f.x; r.at(1, 2) = -f.y; r.at(2, 2) = -f.z;
    r.at(3, 0) = -dot(s, eye); r.at(3, 1) = -dot(u, eye); r.at(3, 2) = dot(f, eye);
    return r;
}
// glm::perspective (right-handed, depth -1..1), fovy in radians
inline mat4 perspective(float fovy, float aspect, float zNear, float zFar) {
    float th = std::tan(fovy / 2.0f);
    mat4 r(0.0f);
    r.at(0, 0) = 1.0f / (aspect * th);
    r.at(1, 1) = 1.0f / th;
    r.at(2, 2) = -(zFar + zNear) / (zFar - zNear);
    r.at(2, 3) = -1.0f;
    r.at(3, 2) = -(2.0f * zFar * zNear) / (zFar - zNear);
    return r;
}
// General 4x4 inverse by the adjugate (2x2 sub-determinants), like glm::inverse.
inline mat4 inverse(const mat4& a) {
    const float* m = a.m;
    float s0 = m[0] * m[5] - m[1] * m[4], s1 = m[0] * m[6] - m[2] * m[4], s2 = m[0] * m[7] - m[3] * m[4];
    float s3 = m[1] * m[6] - m[2] * m[5], s4 = m[1] * m[7] - m[3] * m[5], s5 = m[2] * m[7] - m[3] * m[6];
    float c5 = m[10] * m[15] - m[11] * m[14], c4 = m[9] * m[15] - m[11] * m[13], c3 = m[9] * m[14] - m[10] * m[13];
    float c2 = m[8] * m[15] - m[11] * m[12], c1 = m[8] * m[14] - m[10] * m[12], c0 = m[8] * m[13] - m[9] * m[12];
    float det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    float id = 1.0f / det;
    mat4 r(0.0f);
    r.m[0] = (m[5] * c5 - m[6] * c4 + m[7] * c3) * id;
    r.m[1] = (-m[1] * c5 + m[2] * c4 - m[3] * c3) * id;
    r.m[2] = (m[13] * s5 - m[14] * s4 + m[15] * s3) * id;
    r.m[3] = (-m[9] * s5 + m[10] * s4 - m[11] * s3) * id;
    r.m[4] = (-m[4] * c5 + m[6] * c2 - m[7] * c1) * id;
    r.m[5] = (m[0] * c5 - m[2] * c2 + m[3] * c1) * id;
    r.m[6] = (-m[12] * s5 + m[14] * s2 - m[15] * s1) * id;
    r.m[7] = (m[8] * s5 - m[10] * s2 + m[11] * s1) * id;
    r.m[8] = (m[4] * c4 - m[5] * c2 + m[7] * c0) * id;
    r.m[9] = (-m[0] * c4 + m[1] * c2 - m[3] * c0) * id;
    r.m[10] = (m[12] * s4 - m[13] * s2 + m[15] * s0) * id;
    r.m[11] = (-m[8] * s4 + m[9] * s2 - m[11] * s0) * id;
    r.m[12] = (-m[4] * c3 + m[5] * c1 - m[6] * c0) * id;
    r.m[13] = (m[0] * c3 - m[1] * c1 + m[2] * c0) * id;
    r.m[14] = (-m[12] * s3 + m[13] * s1 - m[14] * s0) * id;
    r.m[15] = (m[8] * s3 - m[9] * s1 + m[10] * s0) * id;
    return r;
}

}  // namespace rayzen
