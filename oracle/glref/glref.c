/*
 * glref.c -- runs RayZen's OWN shaders (read at run time from /root/reference/RayZen/shaders/, never copied) on the
 * OpenGL implementation this image ships: Mesa 23.2 llvmpipe (swrast_dri.so), a conformant OpenGL 4.5 core software
 * renderer.  TEST INFRASTRUCTURE ONLY (the pin of oracle/rz_oracle.c; see oracle/glref/README.md): nothing in the product
 * path includes, links or runs it.
 *
 * The image has no X server, no EGL and no OSMesa, so there is no public way to ask for a context.  What it does have is the
 * driver itself and Mesa's own loader interface header (GL/internal/dri_interface.h, package mesa-common-dev): this file is
 * a DRI software-rasteriser LOADER -- the job libGLX_mesa / the X server do -- of ~100 lines: dlopen the driver, hand it
 * the swrast-loader callbacks (an off-screen drawable: nothing is ever presented), create a screen, a 4.3 core context
 * (RayZen's `#version 430 core`, src/main.cpp:215-217) and a dummy drawable, make it current, and fetch GL entry points
 * from Mesa's dispatch library (libglapi).  The GL implementation, the GLSL compiler and the rasteriser are Mesa's; nothing
 * of OpenGL is stood in for.
 *
 * What it replaces of the reference: the frame loop's draw (src/main.cpp:600-640): the SSBO uploads
 * (main.cpp:1072-1119, bindings 0,1,2,5,6,7,8,9), sendSceneDataToShader's uniforms (main.cpp:1356-1379, 1325-1343) and
 * glDrawArrays(GL_TRIANGLE_FAN, 0, 4) over the full-screen quad (main.cpp:1395-1409), into an RGBA32F colour attachment
 * instead of the window so that FragColor comes back unquantised.
 *
 * usage: glref <scene.blob> <out.f32> [shader_dir]
 *   scene.blob (little endian), written by oracle/glref/glref.py:
 *     char magic[4] = "RZGL"; int32 version = 2;
 *     int32 width, height, bounceBudget, numLights, numTriangles;
 *     int32 debugShowLights, debugShowBVH, debugBVHMode, debugSelectedBLAS, debugSelectedTri;
 *     int32 numSamples;   -- 1: the shader as it is.  N > 1: the ONE edit this harness can make to the text it loads --
 *                            FS:676's constant `int numSamples = 1; // increase for better quality` becomes N (the product
 *                            exposes that constant as `spp`); the fixtures say which they are
 *     float fps; float view[16], proj[16], invView[16], invProj[16]; float camPos[3];
 *     then 10 x { uint64 bytes; data }  -- the buffers of bindings 0..9 (bytes = 0: nothing is bound there)
 *   out.f32: width*height*4 floats, FragColor, row 0 = the bottom row (gl_FragCoord's origin).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>

#define DIE(...) do { fprintf(stderr, "glref: " __VA_ARGS__); fprintf(stderr, "\n"); exit(2); } while (0)

/* ---- the swrast loader's callbacks: an off-screen drawable of 16 x 16 that is never shown ---- */
static void cb_get_drawable_info(__DRIdrawable* d, int* x, int* y, int* w, int* h, void* priv) { *x = 0; *y = 0; *w = 16; *h = 16; }
static void cb_put_image(__DRIdrawable* d, int op, int x, int y, int w, int h, char* data, void* priv) {}
static void cb_get_image(__DRIdrawable* d, int x, int y, int w, int h, char* data, void* priv) { memset(data, 0, (size_t)w * h * 4); }
static void cb_put_image2(__DRIdrawable* d, int op, int x, int y, int w, int h, int stride, char* data, void* priv) {}
static void cb_get_image2(__DRIdrawable* d, int x, int y, int w, int h, int stride, char* data, void* priv) { for (int r = 0; r < h; ++r) memset(data + (size_t)r * stride, 0, (size_t)w * 4); }

static const __DRIswrastLoaderExtension swrast_loader = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = cb_get_drawable_info,
    .putImage = cb_put_image,
    .getImage = cb_get_image,
    .putImage2 = cb_put_image2,
    .getImage2 = cb_get_image2,
};
static const __DRIextension* loader_extensions[] = {&swrast_loader.base, NULL};

static void* (*get_proc)(const char*);
#define GLFN(type, name) type name = (type)get_proc(#name); if (!name) DIE("no entry point %s", #name)

static char* read_text(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) DIE("cannot open %s", path);
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* s = malloc((size_t)n + 1);
    if (fread(s, 1, (size_t)n, f) != (size_t)n) DIE("short read of %s", path);
    s[n] = 0;
    fclose(f);
    return s;
}

typedef struct {
    char magic[4];
    int32_t version, width, height, bounceBudget, numLights, numTriangles;
    int32_t debugShowLights, debugShowBVH, debugBVHMode, debugSelectedBLAS, debugSelectedTri, numSamples;
    float fps, view[16], proj[16], invView[16], invProj[16], camPos[3];
} blob_header;

int main(int argc, char** argv) {
    if (argc < 3) DIE("usage: glref <scene.blob> <out.f32> [shader_dir]");
    const char* shader_dir = argc > 3 ? argv[3] : "/root/reference/RayZen/shaders";
    const char* driver = getenv("GLREF_DRIVER") ? getenv("GLREF_DRIVER") : "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so";

    /* ---- the scene ---- */
    FILE* bf = fopen(argv[1], "rb");
    if (!bf) DIE("cannot open %s", argv[1]);
    blob_header H;
    if (fread(&H, sizeof H, 1, bf) != 1 || memcmp(H.magic, "RZGL", 4) != 0 || H.version != 2) DIE("bad blob header");
    void* bufs[10];
    uint64_t bytes[10];
    for (int b = 0; b < 10; ++b) {
        if (fread(&bytes[b], 8, 1, bf) != 1) DIE("short blob");
        bufs[b] = malloc(bytes[b] ? bytes[b] : 1);
        if (bytes[b] && fread(bufs[b], 1, bytes[b], bf) != bytes[b]) DIE("short blob (binding %d)", b);
    }
    fclose(bf);

    /* ---- Mesa's software driver, loaded the way libGLX_mesa loads it ---- */
    void* drv = dlopen(driver, RTLD_NOW | RTLD_GLOBAL);
    if (!drv) DIE("dlopen %s: %s", driver, dlerror());
    const __DRIextension** (*get_ext)(void) = (const __DRIextension** (*)(void))dlsym(drv, __DRI_DRIVER_GET_EXTENSIONS "_swrast");
    if (!get_ext) DIE("driver has no %s_swrast", __DRI_DRIVER_GET_EXTENSIONS);
    const __DRIextension** exts = get_ext();
    const __DRIcoreExtension* core = NULL;
    const __DRIswrastExtension* swrast = NULL;
    for (int i = 0; exts[i]; ++i) {
        if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension*)exts[i];
        if (!strcmp(exts[i]->name, __DRI_SWRAST)) swrast = (const __DRIswrastExtension*)exts[i];
    }
    if (!core || !swrast || swrast->base.version < 4) DIE("driver lacks DRI_Core / DRI_SWRast >= 4");
    const __DRIconfig** configs = NULL;
    __DRIscreen* screen = swrast->createNewScreen2(0, loader_extensions, exts, &configs, NULL);
    if (!screen || !configs || !configs[0]) DIE("createNewScreen2 failed");
    const uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 3};
    unsigned err = 0;
    __DRIcontext* ctx = swrast->createContextAttribs(screen, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &err, NULL);
    if (!ctx) DIE("no OpenGL 4.3 core context (error %u)", err);
    __DRIdrawable* draw = swrast->createNewDrawable(screen, configs[0], NULL);
    if (!draw) DIE("createNewDrawable failed");
    if (!core->bindContext(ctx, draw, draw)) DIE("bindContext failed");

    void* glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    if (!glapi) DIE("dlopen libglapi.so.0: %s", dlerror());
    get_proc = (void* (*)(const char*))dlsym(glapi, "_glapi_get_proc_address");
    if (!get_proc) DIE("no _glapi_get_proc_address");

    GLFN(PFNGLGETSTRINGPROC, glGetString);
    GLFN(PFNGLGETERRORPROC, glGetError);
    fprintf(stderr, "glref: %s | %s | GLSL %s\n", (const char*)glGetString(GL_RENDERER), (const char*)glGetString(GL_VERSION),
            (const char*)glGetString(GL_SHADING_LANGUAGE_VERSION));
    if (getenv("GLREF_PROBE")) return 0;

    GLFN(PFNGLCREATESHADERPROC, glCreateShader);
    GLFN(PFNGLSHADERSOURCEPROC, glShaderSource);
    GLFN(PFNGLCOMPILESHADERPROC, glCompileShader);
    GLFN(PFNGLGETSHADERIVPROC, glGetShaderiv);
    GLFN(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog);
    GLFN(PFNGLCREATEPROGRAMPROC, glCreateProgram);
    GLFN(PFNGLATTACHSHADERPROC, glAttachShader);
    GLFN(PFNGLLINKPROGRAMPROC, glLinkProgram);
    GLFN(PFNGLGETPROGRAMIVPROC, glGetProgramiv);
    GLFN(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog);
    GLFN(PFNGLUSEPROGRAMPROC, glUseProgram);
    GLFN(PFNGLGETUNIFORMLOCATIONPROC, glGetUniformLocation);
    GLFN(PFNGLUNIFORM1IPROC, glUniform1i);
    GLFN(PFNGLUNIFORM1FPROC, glUniform1f);
    GLFN(PFNGLUNIFORM2FPROC, glUniform2f);
    GLFN(PFNGLUNIFORM3FVPROC, glUniform3fv);
    GLFN(PFNGLUNIFORMMATRIX4FVPROC, glUniformMatrix4fv);
    GLFN(PFNGLGENBUFFERSPROC, glGenBuffers);
    GLFN(PFNGLBINDBUFFERPROC, glBindBuffer);
    GLFN(PFNGLBUFFERDATAPROC, glBufferData);
    GLFN(PFNGLBINDBUFFERBASEPROC, glBindBufferBase);
    GLFN(PFNGLGENVERTEXARRAYSPROC, glGenVertexArrays);
    GLFN(PFNGLBINDVERTEXARRAYPROC, glBindVertexArray);
    GLFN(PFNGLVERTEXATTRIBPOINTERPROC, glVertexAttribPointer);
    GLFN(PFNGLENABLEVERTEXATTRIBARRAYPROC, glEnableVertexAttribArray);
    GLFN(PFNGLGENFRAMEBUFFERSPROC, glGenFramebuffers);
    GLFN(PFNGLBINDFRAMEBUFFERPROC, glBindFramebuffer);
    GLFN(PFNGLFRAMEBUFFERTEXTURE2DPROC, glFramebufferTexture2D);
    GLFN(PFNGLCHECKFRAMEBUFFERSTATUSPROC, glCheckFramebufferStatus);
    GLFN(PFNGLGENTEXTURESPROC, glGenTextures);
    GLFN(PFNGLBINDTEXTUREPROC, glBindTexture);
    GLFN(PFNGLTEXIMAGE2DPROC, glTexImage2D);
    GLFN(PFNGLVIEWPORTPROC, glViewport);
    GLFN(PFNGLCLEARCOLORPROC, glClearColor);
    GLFN(PFNGLCLEARPROC, glClear);
    GLFN(PFNGLDRAWARRAYSPROC, glDrawArrays);
    GLFN(PFNGLFINISHPROC, glFinish);
    GLFN(PFNGLREADPIXELSPROC, glReadPixels);
    GLFN(PFNGLREADBUFFERPROC, glReadBuffer);
    GLFN(PFNGLDISABLEPROC, glDisable);
    GLFN(PFNGLCLAMPCOLORPROC, glClampColor);

    /* ---- RayZen's program, from RayZen's files ---- */
    char path[4096];
    GLuint sh[2];
    const char* names[2] = {"vertex_shader.glsl", "fragment_shader.glsl"};
    const GLenum kinds[2] = {GL_VERTEX_SHADER, GL_FRAGMENT_SHADER};
    GLuint prog = glCreateProgram();
    for (int i = 0; i < 2; ++i) {
        snprintf(path, sizeof path, "%s/%s", shader_dir, names[i]);
        /* GLREF_FRAGMENT: another fragment shader behind the same interface -- oracle/glref/probe_math.glsl, which tabulates
         * llvmpipe's sin / cos / acos / pow so that the tests can tell its built-ins from the oracle's (never used for a frame) */
        if (i == 1 && getenv("GLREF_FRAGMENT")) snprintf(path, sizeof path, "%s", getenv("GLREF_FRAGMENT"));
        const char* src = read_text(path);
        if (i == 1 && H.numSamples > 1) {       /* FS:676's constant, see the header comment */
            const char* key = "int numSamples = 1;";
            char* at = strstr(src, key);
            if (!at || strstr(at + 1, key)) DIE("%s: expected exactly one `%s`", path, key);
            char* patched = malloc(strlen(src) + 32);
            const int head = (int)(at - src);
            sprintf(patched, "%.*sint numSamples = %d;%s", head, src, H.numSamples, at + strlen(key));
            src = patched;
        }
        sh[i] = glCreateShader(kinds[i]);
        glShaderSource(sh[i], 1, &src, NULL);
        glCompileShader(sh[i]);
        GLint ok = 0;
        glGetShaderiv(sh[i], GL_COMPILE_STATUS, &ok);
        if (!ok) { char log[8192]; glGetShaderInfoLog(sh[i], sizeof log, NULL, log); DIE("%s does not compile:\n%s", path, log); }
        glAttachShader(prog, sh[i]);
    }
    glLinkProgram(prog);
    { GLint ok = 0; glGetProgramiv(prog, GL_LINK_STATUS, &ok); if (!ok) { char log[8192]; glGetProgramInfoLog(prog, sizeof log, NULL, log); DIE("link failed:\n%s", log); } }
    glUseProgram(prog);

    /* ---- uniforms (main.cpp:1356-1379, 1325-1343, 625-633) ---- */
    glUniform2f(glGetUniformLocation(prog, "resolution"), (float)H.width, (float)H.height);
    glUniform1i(glGetUniformLocation(prog, "numTriangles"), H.numTriangles);
    glUniform1i(glGetUniformLocation(prog, "numLights"), H.numLights);
    glUniform1i(glGetUniformLocation(prog, "uniformBounceBudget"), H.bounceBudget);
    glUniformMatrix4fv(glGetUniformLocation(prog, "camera.viewMatrix"), 1, GL_FALSE, H.view);
    glUniformMatrix4fv(glGetUniformLocation(prog, "camera.projectionMatrix"), 1, GL_FALSE, H.proj);
    glUniform3fv(glGetUniformLocation(prog, "camera.position"), 1, H.camPos);
    glUniformMatrix4fv(glGetUniformLocation(prog, "camera.invViewMatrix"), 1, GL_FALSE, H.invView);
    glUniformMatrix4fv(glGetUniformLocation(prog, "camera.invProjectionMatrix"), 1, GL_FALSE, H.invProj);
    glUniform1i(glGetUniformLocation(prog, "debugShowLights"), H.debugShowLights);
    glUniform1i(glGetUniformLocation(prog, "debugShowBVH"), H.debugShowBVH);
    glUniform1i(glGetUniformLocation(prog, "debugBVHMode"), H.debugBVHMode);
    glUniform1i(glGetUniformLocation(prog, "debugSelectedBLAS"), H.debugSelectedBLAS);
    glUniform1i(glGetUniformLocation(prog, "debugSelectedTri"), H.debugSelectedTri);
    glUniform1f(glGetUniformLocation(prog, "uniformFps"), H.fps);

    /* ---- SSBOs (main.cpp:1072-1119) ---- */
    for (int b = 0; b < 10; ++b) {
        if (!bytes[b]) continue;
        GLuint id;
        glGenBuffers(1, &id);
        glBindBuffer(GL_SHADER_STORAGE_BUFFER, id);
        glBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)bytes[b], bufs[b], GL_STATIC_DRAW);
        glBindBufferBase(GL_SHADER_STORAGE_BUFFER, (GLuint)b, id);
    }

    /* ---- the quad (main.cpp:1395-1409) and an RGBA32F target ---- */
    const float quad[12] = {-1.0f, -1.0f, 0.0f, 1.0f, -1.0f, 0.0f, 1.0f, 1.0f, 0.0f, -1.0f, 1.0f, 0.0f};
    GLuint vao, vbo, fbo, tex;
    glGenVertexArrays(1, &vao);
    glBindVertexArray(vao);
    glGenBuffers(1, &vbo);
    glBindBuffer(GL_ARRAY_BUFFER, vbo);
    glBufferData(GL_ARRAY_BUFFER, sizeof quad, quad, GL_STATIC_DRAW);
    glVertexAttribPointer(0, 3, GL_FLOAT, GL_FALSE, 3 * sizeof(float), (void*)0);
    glEnableVertexAttribArray(0);
    glGenTextures(1, &tex);
    glBindTexture(GL_TEXTURE_2D, tex);
    glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA32F, H.width, H.height, 0, GL_RGBA, GL_FLOAT, NULL);
    glGenFramebuffers(1, &fbo);
    glBindFramebuffer(GL_FRAMEBUFFER, fbo);
    glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tex, 0);
    if (glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) DIE("framebuffer incomplete");
    glViewport(0, 0, H.width, H.height);
    glDisable(GL_DEPTH_TEST);
    glDisable(GL_BLEND);
    glClampColor(GL_CLAMP_READ_COLOR, GL_FALSE);
    glClearColor(0.0f, 0.0f, 0.0f, 0.0f);
    glClear(GL_COLOR_BUFFER_BIT);
    glDrawArrays(GL_TRIANGLE_FAN, 0, 4);
    glFinish();
    float* out = malloc((size_t)H.width * H.height * 16);
    glReadBuffer(GL_COLOR_ATTACHMENT0);
    glReadPixels(0, 0, H.width, H.height, GL_RGBA, GL_FLOAT, out);
    const GLenum e = glGetError();
    if (e != GL_NO_ERROR) DIE("GL error 0x%x", e);
    FILE* of = fopen(argv[2], "wb");
    if (!of || fwrite(out, 16, (size_t)H.width * H.height, of) != (size_t)H.width * H.height) DIE("cannot write %s", argv[2]);
    fclose(of);
    return 0;
}
