// probe_rect.hip -- rate of rectangular D2H copies (hipMemcpy2DAsync into pinned memory): is a
// sub-rectangle of a 4096-wide f64 plane moved at the rate of a contiguous copy? (measurement tool)
//   hipcc -O3 --offload-arch=gfx950 -o probe_rect probe_rect.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

int main()
{
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t nx = 4096, ny = 4096;
    char *d, *h;
    CK(hipMalloc(&d, nx * ny * 8));
    CK(hipMemset(d, 1, nx * ny * 8));
    CK(hipHostMalloc((void **)&h, nx * ny * 8, hipHostMallocDefault));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t widths[] = {4096, 3872, 3000, 2048, 1024, 256};
    const size_t rows[] = {4096, 512, 128};
    for (size_t w : widths)
        for (size_t r : rows) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0, s));
                for (size_t y = 0; y + r <= ny; y += r)
                    CK(hipMemcpy2DAsync(h + y * w * 8, w * 8, d + (y * nx + (nx - w) / 2) * 8, nx * 8, w * 8, r, hipMemcpyDeviceToHost, s));
                CK(hipEventRecord(e1, s));
                CK(hipStreamSynchronize(s));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            printf("{\"probe\": \"2D D2H\", \"width_px\": %zu, \"rows_per_copy\": %zu, \"MB\": %.1f, \"ms\": %.3f, \"GBps\": %.1f}\n", w, r,
                   w * ny * 8 / 1e6, best, w * ny * 8 / (best * 1e-3) / 1e9);
        }
    return 0;
}
