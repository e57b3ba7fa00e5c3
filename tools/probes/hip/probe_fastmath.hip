// probe_fastmath.hip -- measures, on the GPU, the error of every function of
// planetmapper_amd/csrc/pm_fastmath.hip.h against the device libm / IEEE operations.
//
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/hip/probe_fastmath.hip -o tools/probe_fastmath
//   ./tools/probes/hip/probe_fastmath            # prints one JSON line per function
//
// Errors of sqrt / rsqrt / rcp / division are in ulp of the result; the angle functions in
// absolute radians (their results feed degrees with a parity bar of 1e-9 deg = 1.7e-11 rad).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../planetmapper_amd/csrc/pm_fastmath.hip.h"

namespace {

__device__ uint64_t rng(uint64_t &s)
{
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
}
__device__ double uniform(uint64_t &s) { return (double)(rng(s) >> 11) * 0x1.0p-53; }
// log-uniform magnitude over [1e-6, 1e12]: km, seconds, unit-vector components, squared distances
__device__ double magnitude(uint64_t &s) { return exp2(-20.0 + 60.0 * uniform(s)) * (1.0 + uniform(s)); }

__device__ double ulps(double got, double want)
{
    const double u = fabs(want) * 0x1.0p-52;
    return fabs(got - want) / u;
}
__device__ void amax(double *slot, double v)
{
    // non-negative doubles order like their bit patterns
    atomicMax((unsigned long long *)slot, (unsigned long long)__double_as_longlong(v));
}

enum { F_RCP_SEED, F_RSQ_SEED, F_RCP, F_DIV, F_SQRT, F_SQRT_SEED, F_RSQRT, F_ASIN, F_ATAN2, F_SIN_TINY, F_COS_TINY, F_SIN_SMALL,
       F_COS_SMALL, F_SIN_MED, F_COS_MED, F_COUNT };

__global__ void k_probe(double *worst, int per_thread)
{
    uint64_t s = 0x9E3779B97F4A7C15ull * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    double w[F_COUNT];
    for (int i = 0; i < F_COUNT; i++) w[i] = 0.0;
    for (int it = 0; it < per_thread; it++) {
        const double a = magnitude(s) * ((rng(s) & 1) ? 1.0 : -1.0), b = magnitude(s);
        w[F_RCP_SEED] = fmax(w[F_RCP_SEED], ulps(__builtin_amdgcn_rcp(b), 1.0 / b));
        w[F_RSQ_SEED] = fmax(w[F_RSQ_SEED], ulps(__builtin_amdgcn_rsq(b), 1.0 / sqrt(b)));
        w[F_RCP] = fmax(w[F_RCP], ulps(pm::rcp_fast(b), 1.0 / b));
        w[F_DIV] = fmax(w[F_DIV], ulps(pm::div_fast(a, b), a / b));
        w[F_SQRT] = fmax(w[F_SQRT], ulps(pm::sqrt_fast(b), sqrt(b)));
        w[F_SQRT_SEED] = fmax(w[F_SQRT_SEED], ulps(pm::sqrt_seed_pos(b), sqrt(b)));
        w[F_RSQRT] = fmax(w[F_RSQRT], ulps(pm::rsqrt_fast(b), 1.0 / sqrt(b)));
        const double h = uniform(s) - 0.5;
        w[F_ASIN] = fmax(w[F_ASIN], fabs(pm::asin_half(h) - asin(h)));
        const double sx = (rng(s) & 1) ? 1.0 : -1.0;
        w[F_ATAN2] = fmax(w[F_ATAN2], fabs(pm::atan2_fast(a, b * sx) - atan2(a, b * sx)));
        double sn, cs;
        const double t = 2e-3 * h;
        pm::sincos_tiny(t, sn, cs);
        w[F_SIN_TINY] = fmax(w[F_SIN_TINY], fabs(sn - sin(t)));
        w[F_COS_TINY] = fmax(w[F_COS_TINY], fabs(cs - cos(t)));
        const double q = 0.5 * h;
        pm::sincos_small(q, sn, cs);
        w[F_SIN_SMALL] = fmax(w[F_SIN_SMALL], fabs(sn - sin(q)));
        w[F_COS_SMALL] = fmax(w[F_COS_SMALL], fabs(cs - cos(q)));
        const double m = 2e5 * h;
        pm::sincos_medium(m, sn, cs);
        w[F_SIN_MED] = fmax(w[F_SIN_MED], fabs(sn - sin(m)));
        w[F_COS_MED] = fmax(w[F_COS_MED], fabs(cs - cos(m)));
    }
    for (int i = 0; i < F_COUNT; i++) amax(&worst[i], w[i]);
}

}  // namespace

int main()
{
    static const char *names[F_COUNT] = {"v_rcp_f64 seed", "v_rsq_f64 seed", "rcp_fast", "div_fast", "sqrt_fast", "sqrt_seed",
                                         "rsqrt_fast", "asin_half", "atan2_fast", "sin_tiny", "cos_tiny",
                                         "sin_small", "cos_small", "sin_medium", "cos_medium"};
    double *d = nullptr, h[F_COUNT];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) {
        fprintf(stderr, "no GPU\n");
        return 1;
    }
    hipMemset(d, 0, sizeof(h));
    const int per_thread = 4096, blocks = 1024, threads = 256;
    hipLaunchKernelGGL(k_probe, dim3(blocks), dim3(threads), 0, 0, d, per_thread);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) {
        fprintf(stderr, "kernel failed\n");
        return 1;
    }
    for (int i = 0; i < F_COUNT; i++)
        printf("{\"function\": \"%s\", \"max_error\": %.4g, \"unit\": \"%s\", \"samples\": %lld}\n", names[i], h[i],
               i <= F_RSQRT ? "ulp" : "rad", (long long)per_thread * blocks * threads);
    hipFree(d);
    return 0;
}
