/*
 * pm_oracle_quad.c -- the CPU oracle's image-plane path evaluated in IEEE binary128.
 *
 * TEST INFRASTRUCTURE ONLY (see the header of pm_oracle.c; the same rules apply: only tests/
 * may load this library). It is not a second restatement: it compiles pm_oracle.c itself with
 * every `double` turned into `__float128` and every libm call into its libquadmath form, so the
 * arithmetic is the oracle's formulation of the reference's per-pixel chain, free of binary64
 * rounding (113-bit significand: rounding noise 1e-34 instead of 1e-16). Inputs (geometry block,
 * disc) and outputs stay binary64; constants such as numpy's rad2deg factor keep their binary64
 * values because they are part of the reference's formulation.
 *
 * What it is for: the parity bars of tests/parity.py are argued from a noise floor - two correct
 * binary64 evaluations of this formulation differ by up to 1e-7 deg at the limb. With a binary128
 * evaluation as the truth that argument becomes a measurement: tests/test_truth_f128.py checks
 * that the HIP engine is as close to the truth as the binary64 oracle is, plane by plane.
 * (Epochs: the reference forms `et - lt` in binary64; that quantisation, 3e-8 s, is part of what
 * separates any binary64 evaluation from this one.)
 *
 * Build: gcc -O2 -fopenmp -fPIC -shared pm_oracle_quad.c -lquadmath -lm (oracle/Makefile).
 */
#include <math.h>
#include <quadmath.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double f64; /* the real binary64, for the boundary below */

/* ---- from here on `double` is binary128 and the oracle's entry points get private names */
#define double __float128
#define pm_geometry pmq_geometry
#define pm_disc pmq_disc
#define acos acosq
#define asin asinq
#define atan2 atan2q
#define ceil ceilq
#define cos cosq
#define fabs fabsq
#define floor floorq
#define fmax fmaxq
#define fmin fminq
#define fmod fmodq
#define nearbyint nearbyintq
#define pow powq
#define sin sinq
#define sqrt sqrtq
#undef isnan
#undef isfinite
#define isnan(x) isnanq(x)
#define isfinite(x) finiteq(x)
#define pmo_backplanes_img pmoq_raw_backplanes_img
#define pmo_backplanes_map pmoq_raw_backplanes_map
#define pmo_xy_map pmoq_raw_xy_map
#define pmo_clean_nans pmoq_raw_clean_nans
#define pmo_map_cube pmoq_raw_map_cube
#define pmo_map_cube_spline pmoq_raw_map_cube_spline
#define pmo_regrid_smooth pmoq_raw_regrid_smooth
#define pmo_map_cube_spline_smooth pmoq_raw_map_cube_spline_smooth
#define pmo_map_cube_smooth pmoq_raw_map_cube_smooth
#define pmo_pchip pmoq_raw_pchip
#define pmo_transform pmoq_raw_transform
#define pmo_radec_query pmoq_raw_radec_query
#define pmo_sizeof_geometry pmoq_raw_sizeof_geometry
#define pmo_sizeof_disc pmoq_raw_sizeof_disc
#define pmo_set_num_threads pmoq_raw_set_num_threads
#define pmo_get_max_threads pmoq_raw_get_max_threads

#include "pm_oracle.c"

#undef double

/* ---- binary64 boundary: pm_geometry / pm_disc as the callers hold them (all-f64 fields followed by
 * int32 fields, include/planetmapper_hip.h) converted field by field */
typedef struct disc64 {
    f64 x0, y0, r0, rotation_rad;
    int32_t nx, ny, optimize_speed, reserved;
} disc64;

int pmoq_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#endif
    (void)n;
    return 0;
}

/*
 * Image rows [row_begin, row_begin + n_rows) of the requested planes, evaluated in binary128 and
 * rounded to binary64. geometry64 / geometry64_bytes: the caller's pm_geometry (binary64 layout);
 * out[p]: n_rows * nx doubles for every requested plane p. The block is computed as a frame of its
 * own with the disc centre shifted by row_begin rows - the same pixels, exactly, in this arithmetic.
 */
int pmoq_backplanes_img_rows(const void *geometry64, int geometry64_bytes, const void *disc64_ptr, f64 alt, uint64_t mask,
                             int row_begin, int n_rows, f64 *const *out)
{
    const int k = (geometry64_bytes - 8) / 8; /* f64 fields before the two int32 fields */
    if (k <= 0 || (size_t)(16 * k) != offsetof(pmq_geometry, west_positive)) return PM_ERR_INVALID_ARGUMENT;
    pmq_geometry g;
    memset(&g, 0, sizeof(g));
    const f64 *src = (const f64 *)geometry64;
    __float128 *dst = (__float128 *)&g;
    for (int i = 0; i < k; i++) dst[i] = (__float128)src[i];
    memcpy(&g.west_positive, (const char *)geometry64 + 8 * k, 8);
    const disc64 *d64 = (const disc64 *)disc64_ptr;
    if (row_begin < 0 || n_rows < 0 || row_begin + n_rows > d64->ny) return PM_ERR_INVALID_ARGUMENT;
    pmq_disc d;
    d.x0 = d64->x0;
    d.y0 = (__float128)d64->y0 - (__float128)row_begin;
    d.r0 = d64->r0;
    d.rotation_rad = d64->rotation_rad;
    d.nx = d64->nx;
    d.ny = n_rows;
    d.optimize_speed = d64->optimize_speed;
    d.reserved = 0;
    if (n_rows == 0 || d.nx <= 0) return PM_OK;
    const size_t n = (size_t)n_rows * d.nx;
    __float128 *tmp[PM_NUM_PLANES];
    int rc = PM_OK;
    for (int p = 0; p < PM_NUM_PLANES; p++) tmp[p] = NULL;
    for (int p = 0; p < PM_NUM_PLANES; p++)
        if ((mask >> p) & 1) {
            tmp[p] = (__float128 *)malloc(n * sizeof(__float128));
            if (!tmp[p]) rc = PM_ERR_ALLOC;
        }
    if (rc == PM_OK) rc = pmoq_raw_backplanes_img(&g, &d, (__float128)alt, mask, tmp);
    if (rc == PM_OK)
        for (int p = 0; p < PM_NUM_PLANES; p++)
            if (tmp[p])
                for (size_t i = 0; i < n; i++) out[p][i] = (f64)tmp[p][i];
    for (int p = 0; p < PM_NUM_PLANES; p++) free(tmp[p]);
    return rc;
}

/*
 * Map-space planes of a lon/lat grid (n0 * n1 cells) in binary128, rounded to binary64: the chain
 * pm_backplanes_map / pm_xy_map replace (body_xy.py:3227-3300, 3419-3491, 3667). out[p]: n0 * n1
 * doubles for every requested plane.
 */
int pmoq_backplanes_map(const void *geometry64, int geometry64_bytes, const void *disc64_ptr, f64 alt, uint64_t mask,
                        const f64 *lon_deg, const f64 *lat_deg, int n0, int n1, f64 *const *out)
{
    const int k = (geometry64_bytes - 8) / 8;
    if (k <= 0 || (size_t)(16 * k) != offsetof(pmq_geometry, west_positive) || n0 < 0 || n1 < 0) return PM_ERR_INVALID_ARGUMENT;
    pmq_geometry g;
    memset(&g, 0, sizeof(g));
    const f64 *src = (const f64 *)geometry64;
    __float128 *dst = (__float128 *)&g;
    for (int i = 0; i < k; i++) dst[i] = (__float128)src[i];
    memcpy(&g.west_positive, (const char *)geometry64 + 8 * k, 8);
    const disc64 *d64 = (const disc64 *)disc64_ptr;
    pmq_disc d;
    d.x0 = d64->x0;
    d.y0 = d64->y0;
    d.r0 = d64->r0;
    d.rotation_rad = d64->rotation_rad;
    d.nx = d64->nx;
    d.ny = d64->ny;
    d.optimize_speed = d64->optimize_speed;
    d.reserved = 0;
    const size_t n = (size_t)n0 * n1;
    if (n == 0) return PM_OK;
    int rc = PM_OK;
    __float128 *lon = (__float128 *)malloc(n * sizeof(__float128)), *lat = (__float128 *)malloc(n * sizeof(__float128));
    __float128 *tmp[PM_NUM_PLANES];
    for (int p = 0; p < PM_NUM_PLANES; p++) tmp[p] = NULL;
    if (!lon || !lat) rc = PM_ERR_ALLOC;
    for (int p = 0; p < PM_NUM_PLANES && rc == PM_OK; p++)
        if ((mask >> p) & 1) {
            tmp[p] = (__float128 *)malloc(n * sizeof(__float128));
            if (!tmp[p]) rc = PM_ERR_ALLOC;
        }
    if (rc == PM_OK) {
        for (size_t i = 0; i < n; i++) {
            lon[i] = lon_deg[i];
            lat[i] = lat_deg[i];
        }
        rc = pmoq_raw_backplanes_map(&g, &d, (__float128)alt, mask, lon, lat, n0, n1, tmp);
    }
    if (rc == PM_OK)
        for (int p = 0; p < PM_NUM_PLANES; p++)
            if (tmp[p])
                for (size_t i = 0; i < n; i++) out[p][i] = (f64)tmp[p][i];
    for (int p = 0; p < PM_NUM_PLANES; p++) free(tmp[p]);
    free(lon);
    free(lat);
    return rc;
}
