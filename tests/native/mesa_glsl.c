// mesa_glsl.c -- test infrastructure: runs a GLSL 4.50 compute shader on Mesa's software rasteriser (llvmpipe, the swrast DRI driver
// of the image), without an X server: the driver is opened through the DRI interface (GL/internal/dri_interface.h of the same Mesa), a
// 4.5 core context is made current on a drawable nobody looks at, and the shader is dispatched over images / a uniform block / storage blocks read
// from files.  An INDEPENDENT GLSL implementation (Mesa's compiler front end and its CPU back end): tests/test_glsl_mesa.py compares what
// it computes for shaders/*.comp and for the language-construct cases with the oracle and with librfhip's own translation of the same text.
// Nothing of the product links or runs this.  (-D_POSIX_C_SOURCE not needed: gcc's default is gnu17.)
//
//   mesa_glsl <shader.comp> <W> <H> <groups_x> <groups_y> <job file>        job file, one resource per line:
//     image   <binding> <rgba32f|rgba8> <in.raw|-> <out.raw|->      a W x H texture bound as an image (imageLoad / imageStore)
//     sampler <binding> <rgba32f|rgba8> <in.raw>                     ... bound as a sampler2D: LINEAR, S clamp-to-edge, T repeat (reforge's one sampler)
//     ubo     <binding> <in.bin>                                     a uniform block
//     ssbo    <binding> <bytes> <in.bin|-> <out.bin|->               a storage block (zero-filled without an input file)
//   exit 0 = ran; 2 = Mesa not usable here (the test skips); 3 = the shader does not compile or link (log on stderr); 1 = anything else.
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <GL/glcorearb.h>
#include <GL/internal/dri_interface.h>

static void get_drawable_info(__DRIdrawable* d, int* x, int* y, int* w, int* h, void* p) { (void)d; (void)p; *x = *y = 0; *w = *h = 16; }
static void put_image(__DRIdrawable* d, int op, int x, int y, int w, int h, char* data, void* p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)data; (void)p; }
static void get_image(__DRIdrawable* d, int x, int y, int w, int h, char* data, void* p) { (void)d; (void)x; (void)y; (void)p; memset(data, 0, (size_t)w * h * 4); }
static void put_image2(__DRIdrawable* d, int op, int x, int y, int w, int h, int stride, char* data, void* p) { (void)d; (void)op; (void)x; (void)y; (void)w; (void)h; (void)stride; (void)data; (void)p; }
static void get_image2(__DRIdrawable* d, int x, int y, int w, int h, int stride, char* data, void* p) { (void)d; (void)x; (void)y; (void)w; (void)p; memset(data, 0, (size_t)stride * h); }

static const __DRIswrastLoaderExtension swrast_loader = {
    .base = {__DRI_SWRAST_LOADER, 3},
    .getDrawableInfo = get_drawable_info,
    .putImage = put_image,
    .getImage = get_image,
    .putImage2 = put_image2,
    .getImage2 = get_image2,
};
static const __DRIextension* loader_extensions[] = {&swrast_loader.base, NULL};

typedef void* (*get_proc_fn)(const char*);
static get_proc_fn get_proc;
#define GLF(type, name) type name = (type)get_proc(#name); if (!name) { fprintf(stderr, "mesa_glsl: no %s\n", #name); return 2; }

static void* read_file(const char* path, size_t* n)
{
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "mesa_glsl: cannot read %s\n", path); exit(1); }
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* buf = malloc((size_t)len + 1);
    if (fread(buf, 1, (size_t)len, f) != (size_t)len) { fprintf(stderr, "mesa_glsl: short read of %s\n", path); exit(1); }
    buf[len] = 0;
    fclose(f);
    if (n) *n = (size_t)len;
    return buf;
}
static void write_file(const char* path, const void* data, size_t n)
{
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(data, 1, n, f) != n) { fprintf(stderr, "mesa_glsl: cannot write %s\n", path); exit(1); }
    fclose(f);
}

#define SCRATCH_UNIT 31
struct Out { int kind; GLuint id; int rgba8; size_t bytes; char path[1024]; };

int main(int argc, char** argv)
{
    if (argc != 7) { fprintf(stderr, "usage: mesa_glsl shader.comp W H groups_x groups_y job\n"); return 1; }
    const int W = atoi(argv[2]), H = atoi(argv[3]), gx = atoi(argv[4]), gy = atoi(argv[5]);
    const char* driver_path = getenv("RF_MESA_SWRAST") ? getenv("RF_MESA_SWRAST") : "/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so";
    void* drv = dlopen(driver_path, RTLD_NOW | RTLD_GLOBAL);
    if (!drv) { fprintf(stderr, "mesa_glsl: %s\n", dlerror()); return 2; }
    const __DRIextension** (*get_exts)(void) = (const __DRIextension** (*)(void))dlsym(drv, __DRI_DRIVER_GET_EXTENSIONS "_swrast");
    if (!get_exts) { fprintf(stderr, "mesa_glsl: the driver has no %s_swrast\n", __DRI_DRIVER_GET_EXTENSIONS); return 2; }
    const __DRIextension** driver_extensions = get_exts();
    const __DRIcoreExtension* core = NULL;
    const __DRIswrastExtension* swrast = NULL;
    for (int i = 0; driver_extensions[i]; ++i) {
        if (!strcmp(driver_extensions[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension*)driver_extensions[i];
        if (!strcmp(driver_extensions[i]->name, __DRI_SWRAST)) swrast = (const __DRIswrastExtension*)driver_extensions[i];
    }
    if (!core || !swrast || swrast->base.version < 4) { fprintf(stderr, "mesa_glsl: DRI_Core / DRI_SWRast (v4) not offered\n"); return 2; }
    const __DRIconfig** configs = NULL;
    __DRIscreen* screen = swrast->createNewScreen2(0, loader_extensions, driver_extensions, &configs, NULL);
    if (!screen || !configs || !configs[0]) { fprintf(stderr, "mesa_glsl: createNewScreen2 failed\n"); return 2; }
    const uint32_t attribs[] = {__DRI_CTX_ATTRIB_MAJOR_VERSION, 4, __DRI_CTX_ATTRIB_MINOR_VERSION, 5};
    unsigned error = 0;
    __DRIcontext* ctx = swrast->createContextAttribs(screen, __DRI_API_OPENGL_CORE, configs[0], NULL, 2, attribs, &error, NULL);
    if (!ctx) { fprintf(stderr, "mesa_glsl: no OpenGL 4.5 core context (error %u)\n", error); return 2; }
    __DRIdrawable* draw = swrast->createNewDrawable(screen, configs[0], NULL);
    if (!core->bindContext(ctx, draw, draw)) { fprintf(stderr, "mesa_glsl: bindContext failed\n"); return 2; }

    void* glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
    get_proc = glapi ? (get_proc_fn)dlsym(glapi, "_glapi_get_proc_address") : NULL;
    if (!get_proc) { fprintf(stderr, "mesa_glsl: no _glapi_get_proc_address\n"); return 2; }
    GLF(PFNGLGETSTRINGPROC, glGetString) GLF(PFNGLGETERRORPROC, glGetError)
    GLF(PFNGLCREATESHADERPROC, glCreateShader) GLF(PFNGLSHADERSOURCEPROC, glShaderSource) GLF(PFNGLCOMPILESHADERPROC, glCompileShader)
    GLF(PFNGLGETSHADERIVPROC, glGetShaderiv) GLF(PFNGLGETSHADERINFOLOGPROC, glGetShaderInfoLog) GLF(PFNGLCREATEPROGRAMPROC, glCreateProgram)
    GLF(PFNGLATTACHSHADERPROC, glAttachShader) GLF(PFNGLLINKPROGRAMPROC, glLinkProgram) GLF(PFNGLGETPROGRAMIVPROC, glGetProgramiv)
    GLF(PFNGLGETPROGRAMINFOLOGPROC, glGetProgramInfoLog) GLF(PFNGLUSEPROGRAMPROC, glUseProgram) GLF(PFNGLGENTEXTURESPROC, glGenTextures)
    GLF(PFNGLBINDTEXTUREPROC, glBindTexture) GLF(PFNGLTEXSTORAGE2DPROC, glTexStorage2D) GLF(PFNGLTEXSUBIMAGE2DPROC, glTexSubImage2D)
    GLF(PFNGLBINDIMAGETEXTUREPROC, glBindImageTexture) GLF(PFNGLGETTEXIMAGEPROC, glGetTexImage) GLF(PFNGLTEXPARAMETERIPROC, glTexParameteri)
    GLF(PFNGLACTIVETEXTUREPROC, glActiveTexture) GLF(PFNGLGENBUFFERSPROC, glGenBuffers) GLF(PFNGLBINDBUFFERPROC, glBindBuffer)
    GLF(PFNGLBUFFERDATAPROC, glBufferData) GLF(PFNGLBINDBUFFERBASEPROC, glBindBufferBase) GLF(PFNGLGETBUFFERSUBDATAPROC, glGetBufferSubData)
    GLF(PFNGLDISPATCHCOMPUTEPROC, glDispatchCompute) GLF(PFNGLMEMORYBARRIERPROC, glMemoryBarrier) GLF(PFNGLFINISHPROC, glFinish)
    GLF(PFNGLPIXELSTOREIPROC, glPixelStorei)
    fprintf(stderr, "mesa_glsl: %s / %s / GLSL %s\n", (const char*)glGetString(GL_RENDERER), (const char*)glGetString(GL_VERSION), (const char*)glGetString(GL_SHADING_LANGUAGE_VERSION));

    const char* text = read_file(argv[1], NULL);
    GLuint sh = glCreateShader(GL_COMPUTE_SHADER);
    glShaderSource(sh, 1, &text, NULL);
    glCompileShader(sh);
    GLint ok = 0;
    char log[16384];
    glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
    if (!ok) { glGetShaderInfoLog(sh, sizeof log, NULL, log); fprintf(stderr, "mesa_glsl: compile failed:\n%s\n", log); return 3; }
    GLuint prog = glCreateProgram();
    glAttachShader(prog, sh);
    glLinkProgram(prog);
    glGetProgramiv(prog, GL_LINK_STATUS, &ok);
    if (!ok) { glGetProgramInfoLog(prog, sizeof log, NULL, log); fprintf(stderr, "mesa_glsl: link failed:\n%s\n", log); return 3; }
    glUseProgram(prog);
    if (getenv("RF_MESA_REFLECT")) {      // the layout of the program's blocks as Mesa computed it: one line per block and per member, then exit
        GLF(PFNGLGETPROGRAMINTERFACEIVPROC, glGetProgramInterfaceiv) GLF(PFNGLGETPROGRAMRESOURCEIVPROC, glGetProgramResourceiv)
        GLF(PFNGLGETPROGRAMRESOURCENAMEPROC, glGetProgramResourceName)
        const GLenum block_kinds[2] = {GL_UNIFORM_BLOCK, GL_SHADER_STORAGE_BLOCK}, var_kinds[2] = {GL_UNIFORM, GL_BUFFER_VARIABLE};
        for (int k = 0; k < 2; ++k) {
            GLint n = 0;
            glGetProgramInterfaceiv(prog, block_kinds[k], GL_ACTIVE_RESOURCES, &n);
            for (GLint i = 0; i < n; ++i) {
                const GLenum props[2] = {GL_BUFFER_BINDING, GL_BUFFER_DATA_SIZE};
                GLint v[2] = {0, 0};
                char name[256] = "";
                glGetProgramResourceiv(prog, block_kinds[k], (GLuint)i, 2, props, 2, NULL, v);
                glGetProgramResourceName(prog, block_kinds[k], (GLuint)i, sizeof name, NULL, name);
                printf("block %s %s binding %d bytes %d\n", k ? "storage" : "uniform", name, v[0], v[1]);
            }
            glGetProgramInterfaceiv(prog, var_kinds[k], GL_ACTIVE_RESOURCES, &n);
            for (GLint i = 0; i < n; ++i) {
                const GLenum props[5] = {GL_BLOCK_INDEX, GL_OFFSET, GL_ARRAY_STRIDE, GL_MATRIX_STRIDE, GL_ARRAY_SIZE};
                GLint v[5] = {0, 0, 0, 0, 0};
                char name[256] = "";
                glGetProgramResourceiv(prog, var_kinds[k], (GLuint)i, 5, props, 5, NULL, v);
                if (v[0] < 0) continue;      // a uniform outside blocks (an image, a sampler)
                glGetProgramResourceName(prog, var_kinds[k], (GLuint)i, sizeof name, NULL, name);
                printf("member %s %s offset %d array_stride %d matrix_stride %d count %d\n", k ? "storage" : "uniform", name, v[1], v[2], v[3], v[4]);
            }
        }
        fflush(NULL);
        _Exit(0);
    }
    glPixelStorei(GL_PACK_ALIGNMENT, 1);
    glPixelStorei(GL_UNPACK_ALIGNMENT, 1);

    struct Out outs[128];
    int n_out = 0;
    FILE* job = fopen(argv[6], "r");
    if (!job) { fprintf(stderr, "mesa_glsl: cannot read %s\n", argv[6]); return 1; }
    char kind[32];
    while (fscanf(job, "%31s", kind) == 1) {
        if (!strcmp(kind, "image") || !strcmp(kind, "sampler")) {
            int binding;
            char fmt[32], in[1024], out[1024] = "-";
            const int sampler = !strcmp(kind, "sampler");
            if (fscanf(job, "%d %31s %1023s", &binding, fmt, in) != 3 || (!sampler && fscanf(job, "%1023s", out) != 1)) { fprintf(stderr, "mesa_glsl: bad job line\n"); return 1; }
            if (sampler && (binding < 0 || binding >= SCRATCH_UNIT)) { fprintf(stderr, "mesa_glsl: sampler binding %d\n", binding); return 1; }
            const int rgba8 = !strcmp(fmt, "rgba8");
            const size_t bytes = (size_t)W * H * (rgba8 ? 4 : 16);
            GLuint tex;
            glGenTextures(1, &tex);
            glActiveTexture(GL_TEXTURE0 + (sampler ? binding : SCRATCH_UNIT));      // a sampler2D's binding is a texture unit; everything else is bound on a unit no sampler uses
            glBindTexture(GL_TEXTURE_2D, tex);
            glTexStorage2D(GL_TEXTURE_2D, 1, rgba8 ? GL_RGBA8 : GL_RGBA32F, W, H);
            void* data = NULL;
            if (strcmp(in, "-")) {
                size_t n = 0;
                data = read_file(in, &n);
                if (n != bytes) { fprintf(stderr, "mesa_glsl: %s has %zu bytes, the image %zu\n", in, n, bytes); return 1; }
            } else {
                data = calloc(1, bytes);
            }
            glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, W, H, GL_RGBA, rgba8 ? GL_UNSIGNED_BYTE : GL_FLOAT, data);
            free(data);
            if (sampler) {
                glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_LINEAR);
                glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_LINEAR);
                glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
                glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_REPEAT);
            } else {
                glBindImageTexture((GLuint)binding, tex, 0, GL_FALSE, 0, GL_READ_WRITE, rgba8 ? GL_RGBA8 : GL_RGBA32F);
            }
            if (strcmp(out, "-")) {
                struct Out* o = &outs[n_out++];
                o->kind = 0; o->id = tex; o->rgba8 = rgba8; o->bytes = bytes;
                snprintf(o->path, sizeof o->path, "%s", out);
            }
        } else if (!strcmp(kind, "ubo")) {
            int binding;
            char in[1024];
            if (fscanf(job, "%d %1023s", &binding, in) != 2) { fprintf(stderr, "mesa_glsl: bad job line\n"); return 1; }
            size_t n = 0;
            void* data = read_file(in, &n);
            GLuint buf;
            glGenBuffers(1, &buf);
            glBindBuffer(GL_UNIFORM_BUFFER, buf);
            glBufferData(GL_UNIFORM_BUFFER, (GLsizeiptr)n, data, GL_STATIC_DRAW);
            glBindBufferBase(GL_UNIFORM_BUFFER, (GLuint)binding, buf);
            free(data);
        } else if (!strcmp(kind, "ssbo")) {
            int binding;
            long bytes;
            char in[1024], out[1024];
            if (fscanf(job, "%d %ld %1023s %1023s", &binding, &bytes, in, out) != 4) { fprintf(stderr, "mesa_glsl: bad job line\n"); return 1; }
            void* data = NULL;
            if (strcmp(in, "-")) {
                size_t n = 0;
                data = read_file(in, &n);
                if ((long)n < bytes) { fprintf(stderr, "mesa_glsl: %s is smaller than the block\n", in); return 1; }
            } else {
                data = calloc(1, (size_t)bytes);
            }
            GLuint buf;
            glGenBuffers(1, &buf);
            glBindBuffer(GL_SHADER_STORAGE_BUFFER, buf);
            glBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)bytes, data, GL_DYNAMIC_COPY);
            glBindBufferBase(GL_SHADER_STORAGE_BUFFER, (GLuint)binding, buf);
            free(data);
            if (strcmp(out, "-")) {
                struct Out* o = &outs[n_out++];
                o->kind = 1; o->id = buf; o->rgba8 = 0; o->bytes = (size_t)bytes;
                snprintf(o->path, sizeof o->path, "%s", out);
            }
        } else {
            fprintf(stderr, "mesa_glsl: unknown resource `%s`\n", kind);
            return 1;
        }
    }
    fclose(job);

    glDispatchCompute((GLuint)gx, (GLuint)gy, 1);
    glMemoryBarrier(GL_ALL_BARRIER_BITS);
    glFinish();
    if (getenv("RF_MESA_REPEAT")) {      // scripts/mesa_baseline.py: the time of a dispatch on this machine's cores (the shader must not accumulate)
        const int n = atoi(getenv("RF_MESA_REPEAT"));
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int i = 0; i < n; ++i) {
            glDispatchCompute((GLuint)gx, (GLuint)gy, 1);
            glMemoryBarrier(GL_ALL_BARRIER_BITS);
        }
        glFinish();
        clock_gettime(CLOCK_MONOTONIC, &t1);
        fprintf(stderr, "mesa_glsl: dispatch_ms %.4f\n", ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6) / (n > 0 ? n : 1));
    }
    for (int i = 0; i < n_out; ++i) {
        void* data = malloc(outs[i].bytes);
        if (outs[i].kind == 0) {
            glActiveTexture(GL_TEXTURE0 + SCRATCH_UNIT);
            glBindTexture(GL_TEXTURE_2D, outs[i].id);
            glGetTexImage(GL_TEXTURE_2D, 0, GL_RGBA, outs[i].rgba8 ? GL_UNSIGNED_BYTE : GL_FLOAT, data);
        } else {
            glBindBuffer(GL_SHADER_STORAGE_BUFFER, outs[i].id);
            glGetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, (GLsizeiptr)outs[i].bytes, data);
        }
        write_file(outs[i].path, data, outs[i].bytes);
        free(data);
    }
    const GLenum e = glGetError();
    if (e != GL_NO_ERROR) { fprintf(stderr, "mesa_glsl: GL error 0x%x\n", e); return 1; }
    fflush(NULL);
    _Exit(0);      // no teardown: the process ends here
}
