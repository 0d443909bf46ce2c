// Loads a gfx950 code object and launches its crop_resize_norm through the HIP module API (tools/hw/pkisa_run.py).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
extern "C" void* pkmod_load(const char* path, const char* kernel) {
    hipModule_t m;
    if (hipModuleLoad(&m, path) != hipSuccess) { fprintf(stderr, "hipModuleLoad failed: %s\n", path); return nullptr; }
    hipFunction_t f;
    if (hipModuleGetFunction(&f, m, kernel) != hipSuccess) { fprintf(stderr, "no kernel %s\n", kernel); return nullptr; }
    return (void*)f;
}
extern "C" int pkmod_crop(void* fn, const uint8_t* frames, int nframes, int H, int W, const float* boxes, const int32_t* counts,
                          int cap, int size, float* out, void* stream) {
    void* args[] = {&frames, &H, &W, &boxes, &counts, &cap, &size, &out};
    return (int)hipModuleLaunchKernel((hipFunction_t)fn, nframes * cap, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, args, nullptr);
}
