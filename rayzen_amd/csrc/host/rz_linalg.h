// rz_linalg.h -- the handful of vector/matrix operations RayZen's host code
// takes from GLM (include/Camera.h:42-48, src/main.cpp:378-384,1001), written
// out so the host side builds without GLM.  Column-major mat4, like GLM/GLSL:
// m[c*4 + r].  float32 throughout.
//
// The render path's C-ABI receives finished matrices, so the rounding of
// these helpers is not part of the compared path --
// but the arrays a frontend built on these classes uploads should be the bytes
// a RayZen build would upload, so every function follows GLM 0.9.9.8's
// published evaluation order (inverse and mat4 * vec4 are the two where that
// order is not the obvious left-to-right one).
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

// GLM 0.9.9.8 (the libglm-dev RayZen's install_requirements.sh pulls in; GLM is header-only, un-vendored and not in this
// image, so its published algorithm is restated here, glm/detail/type_mat4x4.inl): mat4 * vec4 adds the four column
// products PAIRWISE, (m[0]*v.x + m[1]*v.y) + (m[2]*v.z + m[3]*v.w) -- not left to right.  RayZen calls it for the eight
// corners of every instance's world box (src/main.cpp:986, 1183).
inline vec4 operator*(const mat4& a, vec4 v) {
    vec4 r;
    r.x = (a.m[0] * v.x + a.m[4] * v.y) + (a.m[8] * v.z + a.m[12] * v.w);
    r.y = (a.m[1] * v.x + a.m[5] * v.y) + (a.m[9] * v.z + a.m[13] * v.w);
    r.z = (a.m[2] * v.x + a.m[6] * v.y) + (a.m[10] * v.z + a.m[14] * v.w);
    r.w = (a.m[3] * v.x + a.m[7] * v.y) + (a.m[11] * v.z + a.m[15] * v.w);
    return r;
}
// mat4 * mat4 (same file): column c of the product = ((A[0]*B[c][0] + A[1]*B[c][1]) + A[2]*B[c][2]) + A[3]*B[c][3]
inline mat4 operator*(const mat4& a, const mat4& b) {
    mat4 r(0.0f);
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row)
            r.at(c, row) = ((a.at(0, row) * b.at(c, 0) + a.at(1, row) * b.at(c, 1)) + a.at(2, row) * b.at(c, 2)) + a.at(3, row) * b.at(c, 3);
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
// glm::rotate(m, angle, axis) (glm/ext/matrix_transform.inl): the 3x3 rotation block applied to m's first three columns
// with three-term sums, Result[c] = (m[0]*R[c][0] + m[1]*R[c][1]) + m[2]*R[c][2]; the fourth column is copied.
inline mat4 rotate(const mat4& m, float angle, vec3 axis) {
    float c = std::cos(angle), s = std::sin(angle);
    vec3 a = normalize(axis);
    vec3 t = (1.0f - c) * a;
    float R[3][3];
    R[0][0] = c + t.x * a.x; R[0][1] = t.x * a.y + s * a.z; R[0][2] = t.x * a.z - s * a.y;
    R[1][0] = t.y * a.x - s * a.z; R[1][1] = c + t.y * a.y; R[1][2] = t.y * a.z + s * a.x;
    R[2][0] = t.z * a.x + s * a.y; R[2][1] = t.z * a.y - s * a.x; R[2][2] = c + t.z * a.z;
    mat4 r(0.0f);
    for (int col = 0; col < 3; ++col)
        for (int row = 0; row < 4; ++row)
            r.at(col, row) = (m.at(0, row) * R[col][0] + m.at(1, row) * R[col][1]) + m.at(2, row) * R[col][2];
    for (int row = 0; row < 4; ++row) r.at(3, row) = m.at(3, row);
    return r;
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
// glm::inverse(mat4) as GLM 0.9.9.8 publishes it (glm/detail/func_matrix.inl, compute_inverse<4, 4>), restated operation
// for operation -- RayZen fills BVHInstance::inverseTransform with it (src/main.cpp:1001, 1058, 1151) and the shader
// transforms every ray by that matrix, so its rounding is part of what a RayZen build hands to binding 9:
//   * eighteen 2x2 sub-determinants of rows 1..3 ("Coef"), each  a*b - c*d;
//   * six coefficient vectors Fac0..5 = (Coef, Coef, Coef', Coef'') and four vectors Vec0..3 made of row 0 / row 1 entries;
//   * the cofactor columns  Inv_k = (Vec_a * Fac_i - Vec_b * Fac_j) + Vec_c * Fac_l, signed (+,-,+,-) / (-,+,-,+);
//   * the determinant from the first row of the cofactor matrix: Dot0 = column0(m) * Row0, Dot1 = (x + y) + (z + w);
//   * every cofactor multiplied by 1 / Dot1 (a singular matrix gives inf / NaN entries, as in GLM: no test, no throw).
// m[c * 4 + r] is GLM's m[c][r].
inline mat4 inverse(const mat4& a) {
    const float* m = a.m;
    auto M = [m](int c, int r) { return m[c * 4 + r]; };
    const float c00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3), c02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3), c03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
    const float c04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3), c06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3), c07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
    const float c08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2), c10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2), c11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
    const float c12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3), c14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3), c15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
    const float c16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2), c18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2), c19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
    const float c20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1), c22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1), c23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
    const float fac[6][4] = {{c00, c00, c02, c03}, {c04, c04, c06, c07}, {c08, c08, c10, c11},
                             {c12, c12, c14, c15}, {c16, c16, c18, c19}, {c20, c20, c22, c23}};
    float vec[4][4];                                    // Vec_k = (m[1][k], m[0][k], m[0][k], m[0][k])
    for (int k = 0; k < 4; ++k) { vec[k][0] = M(1, k); vec[k][1] = vec[k][2] = vec[k][3] = M(0, k); }
    // column k of the cofactor matrix uses the three Vec that are NOT Vec_(k') and three Fac, in GLM's pairing
    static const int va[4] = {1, 0, 0, 0}, vb[4] = {2, 2, 1, 1}, vc[4] = {3, 3, 3, 2};
    static const int fa[4] = {0, 0, 1, 2}, fb[4] = {1, 3, 3, 4}, fc[4] = {2, 4, 5, 5};
    mat4 r(0.0f);
    for (int k = 0; k < 4; ++k)
        for (int j = 0; j < 4; ++j) {
            const float inv = (vec[va[k]][j] * fac[fa[k]][j] - vec[vb[k]][j] * fac[fb[k]][j]) + vec[vc[k]][j] * fac[fc[k]][j];
            const float sign = ((k + j) & 1) ? -1.0f : 1.0f;            // SignA = (+,-,+,-) on columns 0 and 2, SignB on 1 and 3
            r.m[k * 4 + j] = inv * sign;
        }
    const float d0 = m[0] * r.m[0], d1 = m[1] * r.m[4], d2 = m[2] * r.m[8], d3 = m[3] * r.m[12];     // m[0] * Row0
    const float oneOverDet = 1.0f / ((d0 + d1) + (d2 + d3));
    for (int k = 0; k < 16; ++k) r.m[k] = r.m[k] * oneOverDet;
    return r;
}

}  // namespace rayzen
