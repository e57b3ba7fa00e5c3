// probe_host_path.hip -- what the host <-> HBM leg of the path costs on this box (measurement tool,
// not product code): pinned allocation / registration, DMA rates, pageable copies, threaded
// copies into fresh pages, duplex transfers, and kernels that read / write pinned host memory
// directly (zero-copy gathers). Output: one JSON object per line.
//
//   hipcc -O3 --offload-arch=gfx950 -o probe_host_path probe_host_path.hip -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void par_copy(char *dst, const char *src, size_t bytes, int threads)
{
    std::vector<std::thread> th;
    size_t per = ((bytes / threads) + 4095) & ~(size_t)4095;
    for (int t = 0; t < threads; t++) {
        size_t a = (size_t)t * per, b = a + per > bytes ? bytes : a + per;
        if (a >= bytes) break;
        th.emplace_back([=] { memcpy(dst + a, src + a, b - a); });
    }
    for (auto &t : th) t.join();
}

// gather: lane i of plane pl reads 2 x 16 B at a pseudo-random pixel of its plane (the footprint of
// a bilinear sample), sums them, stores 8 B coalesced
__global__ void k_gather(const double *cube, size_t plane_elems, int nx, int n_map, double *out)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    const int pl = blockIdx.y;
    if (m >= n_map) return;
    // cells of a map row walk along an image row, ~5 px apart; rows ~6 px apart
    const int row = m / 360, col = m % 360;
    const size_t x = 60 + (size_t)col * 5 / 2, y = 40 + (size_t)row * 5;
    const double *p = cube + (size_t)pl * plane_elems + y * nx + x;
    out[(size_t)pl * n_map + m] = p[0] + p[1] + p[nx] + p[nx + 1];
}

__global__ void k_fill(double *out, size_t n, double v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = v + (double)i;
}

int main(int argc, char **argv)
{
    const size_t MB = 1 << 20;
    CK(hipSetDevice(0));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    const size_t big = 1024 * MB;
    char *dev, *dev2;
    CK(hipMalloc(&dev, big));
    CK(hipMalloc(&dev2, big));
    CK(hipMemset(dev, 1, big));
    CK(hipDeviceSynchronize());

    // ---- pinned allocation cost
    for (unsigned flags : {0u, (unsigned)hipHostMallocNonCoherent}) {
        for (size_t sz : {128 * MB, 1024 * MB}) {
            void *h;
            double t0 = now();
            CK(hipHostMalloc(&h, sz, flags));
            double t1 = now();
            memset(h, 0, sz);
            double t2 = now();
            CK(hipHostFree(h));
            double t3 = now();
            printf("{\"probe\": \"hipHostMalloc\", \"flags\": %u, \"MB\": %zu, \"alloc_ms\": %.2f, \"first_touch_ms\": %.2f, \"free_ms\": %.2f}\n",
                   flags, sz / MB, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
        }
    }
    // ---- registration of fresh pageable memory
    {
        size_t sz = 128 * MB;
        char *h = (char *)aligned_alloc(4096, sz);
        double t0 = now();
        CK(hipHostRegister(h, sz, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(h, dev, sz, hipMemcpyDeviceToHost, s0));
        CK(hipStreamSynchronize(s0));
        double t2 = now();
        CK(hipHostUnregister(h));
        double t3 = now();
        printf("{\"probe\": \"hipHostRegister fresh\", \"MB\": %zu, \"register_ms\": %.2f, \"d2h_ms\": %.2f, \"unregister_ms\": %.2f}\n",
               sz / MB, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
        free(h);
    }
    // ---- DMA rates, pinned
    char *pin, *pin2;
    CK(hipHostMalloc((void **)&pin, big, 0));
    CK(hipHostMalloc((void **)&pin2, big, hipHostMallocNonCoherent));
    memset(pin, 2, big);
    memset(pin2, 3, big);
    for (size_t sz : {1 * MB, 8 * MB, 32 * MB, 128 * MB, 1024 * MB}) {
        for (int dir = 0; dir < 2; dir++) {
            int reps = sz >= 128 * MB ? 4 : 16;
            CK(hipMemcpyAsync(dir ? pin : dev, dir ? dev : pin, sz, dir ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, s0));
            CK(hipStreamSynchronize(s0));
            double t0 = now();
            for (int r = 0; r < reps; r++)
                CK(hipMemcpyAsync(dir ? pin : dev, dir ? dev : pin, sz, dir ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, s0));
            CK(hipStreamSynchronize(s0));
            double dt = (now() - t0) / reps;
            printf("{\"probe\": \"dma pinned\", \"dir\": \"%s\", \"MB\": %zu, \"GBps\": %.2f, \"ms\": %.3f}\n", dir ? "d2h" : "h2d",
                   sz / MB, sz / dt / 1e9, dt * 1e3);
        }
    }
    // duplex: h2d on s0 and d2h on s1 at once
    {
        size_t sz = 512 * MB;
        double t0 = now();
        for (int r = 0; r < 4; r++) {
            CK(hipMemcpyAsync(dev, pin, sz, hipMemcpyHostToDevice, s0));
            CK(hipMemcpyAsync(pin2, dev2, sz, hipMemcpyDeviceToHost, s1));
        }
        CK(hipStreamSynchronize(s0));
        CK(hipStreamSynchronize(s1));
        double dt = (now() - t0) / 4;
        printf("{\"probe\": \"dma duplex\", \"MB_each\": %zu, \"GBps_each\": %.2f}\n", sz / MB, sz / dt / 1e9);
    }
    // two h2d streams at once (do two SDMA engines add up?)
    {
        size_t sz = 512 * MB;
        double t0 = now();
        for (int r = 0; r < 4; r++) {
            CK(hipMemcpyAsync(dev, pin, sz, hipMemcpyHostToDevice, s0));
            CK(hipMemcpyAsync(dev2, pin2, sz, hipMemcpyHostToDevice, s1));
        }
        CK(hipStreamSynchronize(s0));
        CK(hipStreamSynchronize(s1));
        double dt = (now() - t0) / 4;
        printf("{\"probe\": \"dma 2x h2d\", \"MB_each\": %zu, \"GBps_total\": %.2f}\n", sz / MB, 2 * sz / dt / 1e9);
    }
    // ---- pageable copies through the runtime
    for (int fresh = 0; fresh < 2; fresh++) {
        size_t sz = 128 * MB;
        char *h = (char *)aligned_alloc(4096, sz);
        if (!fresh) memset(h, 1, sz);
        double t0 = now();
        CK(hipMemcpyAsync(h, dev, sz, hipMemcpyDeviceToHost, s0));
        CK(hipStreamSynchronize(s0));
        double dt = now() - t0;
        printf("{\"probe\": \"pageable d2h\", \"fresh_pages\": %d, \"MB\": %zu, \"GBps\": %.2f}\n", fresh, sz / MB, sz / dt / 1e9);
        t0 = now();
        CK(hipMemcpyAsync(dev, h, sz, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        dt = now() - t0;
        printf("{\"probe\": \"pageable h2d\", \"MB\": %zu, \"GBps\": %.2f}\n", sz / MB, sz / dt / 1e9);
        free(h);
    }
    // ---- threaded copies pinned -> pageable (fresh pages and touched pages) and back
    for (int threads : {1, 2, 4, 8, 16}) {
        size_t sz = 512 * MB;
        char *h = (char *)aligned_alloc(2 * MB, sz);
        madvise(h, sz, 14 /* MADV_HUGEPAGE */);
        double t0 = now();
        par_copy(h, pin, sz, threads);
        double t1 = now();
        par_copy(h, pin, sz, threads);
        double t2 = now();
        par_copy(pin, h, sz, threads);
        double t3 = now();
        printf("{\"probe\": \"threaded memcpy\", \"threads\": %d, \"MB\": %zu, \"to_fresh_GBps\": %.2f, \"to_touched_GBps\": %.2f, \"to_pinned_GBps\": %.2f}\n",
               threads, sz / MB, sz / (t1 - t0) / 1e9, sz / (t2 - t1) / 1e9, sz / (t3 - t2) / 1e9);
        free(h);
    }
    // ---- kernels on pinned host memory: coalesced stores, gathers
    for (int which = 0; which < 2; which++) {
        double *hp = (double *)(which ? pin2 : pin);
        size_t n = 512 * MB / 8;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s0, hp, n, 1.0);
        CK(hipStreamSynchronize(s0));
        CK(hipEventRecord(e0, s0));
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s0, hp, n, 2.0);
        CK(hipEventRecord(e1, s0));
        CK(hipStreamSynchronize(s0));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"probe\": \"kernel store to pinned\", \"noncoherent\": %d, \"MB\": %zu, \"GBps\": %.2f}\n", which, n * 8 / MB,
               n * 8 / (ms * 1e-3) / 1e9);
        // gather: 128 planes of 1024^2 f64 (1 GiB), 64800 cells
        const int nx = 1024, n_map = 64800, planes = 128;
        double *out = (double *)dev2;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0, s0));
            hipLaunchKernelGGL(k_gather, dim3((n_map + 255) / 256, planes), dim3(256), 0, s0, hp, (size_t)nx * nx, nx, n_map, out);
            CK(hipEventRecord(e1, s0));
            CK(hipStreamSynchronize(s0));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("{\"probe\": \"kernel gather from pinned\", \"noncoherent\": %d, \"rep\": %d, \"planes\": %d, \"ms\": %.3f, \"ms_per_GiB_of_cube\": %.3f, \"sample_GBps\": %.2f}\n",
                   which, rep, planes, ms, ms, (double)planes * n_map * 32 / (ms * 1e-3) / 1e9);
        }
        // the same gather from HBM for comparison
        if (which == 0) {
            CK(hipEventRecord(e0, s0));
            hipLaunchKernelGGL(k_gather, dim3((n_map + 255) / 256, planes), dim3(256), 0, s0, (const double *)dev, (size_t)nx * nx, nx, n_map, out);
            CK(hipEventRecord(e1, s0));
            CK(hipStreamSynchronize(s0));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("{\"probe\": \"kernel gather from HBM\", \"planes\": %d, \"ms\": %.3f}\n", planes, ms);
        }
    }
    // ---- do a zero-copy gather and a DMA copy share the link, or add up? (same direction: host -> GPU)
    {
        const int nx = 1024, n_map = 64800, planes = 128;
        double *out = (double *)dev2;
        hipEvent_t e0, e1, f0, f1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
        for (int with_dma = 0; with_dma < 2; with_dma++) {
            CK(hipDeviceSynchronize());
            double t0 = now();
            CK(hipEventRecord(e0, s0));
            for (int r = 0; r < 4; r++)
                hipLaunchKernelGGL(k_gather, dim3((n_map + 255) / 256, planes), dim3(256), 0, s0, (const double *)pin2, (size_t)nx * nx, nx, n_map, out);
            CK(hipEventRecord(e1, s0));
            if (with_dma) {
                CK(hipEventRecord(f0, s1));
                CK(hipMemcpyAsync(dev, pin, 1024 * MB, hipMemcpyHostToDevice, s1));
                CK(hipEventRecord(f1, s1));
            }
            CK(hipStreamSynchronize(s0));
            CK(hipStreamSynchronize(s1));
            double wall = now() - t0;
            float gms, dms = 0;
            CK(hipEventElapsedTime(&gms, e0, e1));
            if (with_dma) CK(hipEventElapsedTime(&dms, f0, f1));
            printf("{\"probe\": \"gather 4 x 128 planes from pinned %s\", \"gather_ms\": %.2f, \"dma_1GiB_ms\": %.2f, \"wall_ms\": %.2f}\n",
                   with_dma ? "WITH a concurrent 1 GiB H2D DMA" : "alone", gms, dms, wall * 1e3);
        }
    }
    (void)argc;
    (void)argv;
    return 0;
}
