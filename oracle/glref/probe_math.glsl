#version 430 core
// probe_math.glsl -- NOT RayZen's shader: a table-maker for the built-in functions of the GL implementation that runs
// RayZen's shader in oracle/glref (Mesa llvmpipe).  Pixel i reads (x, y) = in[2i], in[2i+1] from binding 0 and writes
// four function values chosen by `numTriangles` (reused as the mode).  tests/test_glref.py uses it to show which built-ins
// differ from the oracle's pinned ones, and by how much, at the arguments RayZen's hash (FS:188-190) really uses.
uniform vec2 resolution;
uniform int numTriangles;
layout(std430, binding = 0) buffer In { float xs[]; };
out vec4 FragColor;
void main() {
    int i = int(gl_FragCoord.y) * int(resolution.x) + int(gl_FragCoord.x);
    float x = xs[2 * i], y = xs[2 * i + 1];
    if (numTriangles == 0)      FragColor = vec4(sin(x), cos(x), acos(y), fract(sin(x) * 43758.5453));
    else if (numTriangles == 1) FragColor = vec4(x / y, sqrt(x), inversesqrt(x), pow(x, y));
    else if (numTriangles == 2) FragColor = vec4(dot(vec2(x, y), vec2(12.9898, 78.233)), x * y + x, fract(x), length(vec3(x, y, 1.0)));
    else                        FragColor = vec4(normalize(vec3(x, y, 1.0)), 1.0 / x);
}
