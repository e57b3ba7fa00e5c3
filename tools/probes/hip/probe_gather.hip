// probe_gather.hip -- what sparse reads cost on this chip (measurement tool, not product code).
//
// The reprojection gathers 2 x 16 bytes per (map cell, plane) from a cube far larger than the
// caches: a 1 deg map touches about a third of the 64-byte sectors of each 1024^2 plane, in short
// runs along image rows. This probe reads a large buffer in runs of `run` bytes separated by `gap`
// bytes (16 bytes per lane, consecutive lanes on consecutive pieces of a run) and reports the rate of
// the bytes actually requested - from HBM and from pinned host memory (the zero-copy path) - so
// that the kernel's numbers can be held against what the memory system gives for that shape.
//
//   hipcc -O3 --offload-arch=gfx950 -o probe_gather probe_gather.hip
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

typedef double d2 __attribute__((ext_vector_type(2)));

// lane q reads the 16-byte piece q of the run sequence; `sink` keeps the loads alive
__global__ __launch_bounds__(256) void k_runs(const char *buf, size_t n_pieces, unsigned pieces_per_run, size_t period,
                                              double *sink)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n_pieces) return;
    const size_t run_id = q / pieces_per_run, within = q % pieces_per_run;
    const d2 v = *(const d2 *)(buf + run_id * period + within * 16);
    if (v.x == 1.2345e300 && v.y == 5.4321e-300) sink[0] = v.x;
}

static float time_runs(const char *buf, size_t buf_bytes, unsigned run, unsigned gap, double *sink, hipStream_t s,
                       size_t *bytes_read)
{
    const size_t period = (size_t)run + gap;
    const size_t n_runs = buf_bytes / period;
    const unsigned ppr = run / 16;
    const size_t n_pieces = n_runs * ppr;
    *bytes_read = n_pieces * 16;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((n_pieces + 255) / 256);
    hipLaunchKernelGGL(k_runs, dim3(grid), dim3(256), 0, s, buf, n_pieces, ppr, period, sink);
    CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(k_runs, dim3(grid), dim3(256), 0, s, buf, n_pieces, ppr, period, sink);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return best;
}

int main()
{
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double *sink;
    CK(hipMalloc(&sink, 64));
    const size_t hbm_bytes = (size_t)4 << 30, host_bytes = (size_t)1 << 30;
    char *hbm, *host;
    CK(hipMalloc(&hbm, hbm_bytes));
    CK(hipMemset(hbm, 0, hbm_bytes));
    CK(hipHostMalloc((void **)&host, host_bytes, hipHostMallocNonCoherent));
    for (size_t i = 0; i < host_bytes; i += 4096) host[i] = 0;
    const unsigned shapes[][2] = {{4096, 0},  {64, 64},   {64, 128},  {64, 192},  {128, 128}, {128, 256},
                                  {256, 256}, {256, 512}, {512, 512}, {1024, 1024}, {2048, 2048}, {16, 48},  {32, 32}};
    for (auto &sh : shapes) {
        size_t nb;
        float ms = time_runs(hbm, hbm_bytes, sh[0], sh[1], sink, s, &nb);
        // sectors of 64 bytes touched (a 16- or 32-byte run still costs its sector)
        const size_t per_run = ((size_t)sh[0] + 63) / 64 * 64;
        const size_t sector_bytes = nb / sh[0] * per_run;
        printf("{\"probe\": \"runs from HBM\", \"run\": %u, \"gap\": %u, \"density\": %.3f, \"ms\": %.3f, \"requested_GBps\": %.1f, \"sector_GBps\": %.1f}\n",
               sh[0], sh[1], (double)sh[0] / (sh[0] + sh[1]), ms, nb / (ms * 1e-3) / 1e9, sector_bytes / (ms * 1e-3) / 1e9);
    }
    for (auto &sh : shapes) {
        size_t nb;
        float ms = time_runs(host, host_bytes, sh[0], sh[1], sink, s, &nb);
        const size_t per_run = ((size_t)sh[0] + 63) / 64 * 64;
        const size_t sector_bytes = nb / sh[0] * per_run;
        printf("{\"probe\": \"runs from pinned host\", \"run\": %u, \"gap\": %u, \"density\": %.3f, \"ms\": %.3f, \"requested_GBps\": %.2f, \"sector_GBps\": %.2f}\n",
               sh[0], sh[1], (double)sh[0] / (sh[0] + sh[1]), ms, nb / (ms * 1e-3) / 1e9, sector_bytes / (ms * 1e-3) / 1e9);
    }
    return 0;
}
