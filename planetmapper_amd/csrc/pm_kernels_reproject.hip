// pm_kernels_reproject.hip -- gfx950 kernels of Observation.get_mapped_data / BodyXY.map_img:
//
//   k_reproject<T>         bilinear / nearest reprojection of a cube onto the map grid, NaN
//                          propagation and on-the-fly NaN pre-clean
//   k_median_hist / pick   per-plane nanmedian (radix select) for the pre-clean
//   k_spline_*             interpolating tensor-product splines (degrees 1..5)
//   k_reproject_smooth_*   'smooth' (PCHIP-oversampled) reprojection, evaluated on the fly (4 x 4 window form + gap-aware form)
//
// One lane per (map cell, plane): blockIdx.y = plane, so a wave gathers neighbouring map cells
// from ONE plane and stores coalesced along the map row. (Keeping a lane's cell for a few planes - the cell's coordinates
// are two thirds of a plane's traffic on a fine map - pays in the smooth kernel only: k_reproject ran no faster with 4
// planes a lane, 29.4 us a plane of a 0.1 deg map, and slower with the loop at one.)
#include "pm_device.hip.h"

namespace pm {

// ------------------------------------------------------------------ reprojection
template <typename T>
__device__ __forceinline__ double load_as_f64(const T *p, size_t i)
{
    return (double)p[i];
}

// p[i .. i+3] in as few load instructions as the element size allows (the address is aligned to the element only: gfx950
// takes global loads of 16 bytes at any dword address). The gathers of the fine-map kernels are bound by the number of
// load instructions the texture path takes in, not by bytes.
template <typename T>
__device__ __forceinline__ void load4_as_f64(const T *p, size_t i, double &a, double &b, double &c, double &d)
{
    typedef T Quad __attribute__((ext_vector_type(4), aligned(sizeof(T) < 4 ? sizeof(T) : 4)));
    const Quad v = *(const Quad *)(p + i);
    a = (double)v.x;
    b = (double)v.y;
    c = (double)v.z;
    d = (double)v.w;
}
template <typename T>
__device__ __forceinline__ void load2_as_f64(const T *p, size_t i, double &a, double &b)
{
    typedef T Pair __attribute__((ext_vector_type(2), aligned(sizeof(T) < 4 ? sizeof(T) : 4)));
    const Pair v = *(const Pair *)(p + i);
    a = (double)v.x;
    b = (double)v.y;
}

// Value the reference would interpolate from at pixel (i, j): the pixel itself if finite,
// else the mean of the finite pixels of its clipped 3x3 window, else the plane's nanmedian
// (BodyXY._replace_nans_with_interpolated_values body_xy.py:1871-1904; the reflect-mode
// `uniform_filter(bad, size=3)` test there is equivalent to "the clipped window holds no
// finite pixel" because reflection only repeats pixels of the window).
// cleaned value of pixel (i, j) whose raw value `v` has been loaded already
template <typename T>
__device__ __forceinline__ double cleaned_value(const T *img, double v, long i, long j, int ny, int nx, double median,
                                                bool &needs_median)
{
    if (isfinite(v)) return v;
    if (!img) {
        // a block table filled by the host, no plane behind it: the plane is redone in full
        needs_median = true;
        return median;
    }
    double sum = 0.0;
    int cnt = 0;
    for (long ii = (i > 0 ? i - 1 : 0); ii <= i + 1 && ii < ny; ii++)
        for (long jj = (j > 0 ? j - 1 : 0); jj <= j + 1 && jj < nx; jj++) {
            const double w = load_as_f64(img, (size_t)ii * nx + jj);
            if (isfinite(w)) {
                sum += w;
                cnt++;
            }
        }
    if (cnt > 0) return sum / (double)cnt;
    needs_median = true;
    return median;
}
template <typename T>
__device__ __forceinline__ double cleaned_at(const T *img, long i, long j, int ny, int nx, double median,
                                             bool &needs_median)
{
    return cleaned_value(img, load_as_f64(img, (size_t)i * nx + j), i, j, ny, nx, median, needs_median);
}
// One (map cell, plane) sample of BodyXY.map_img for 'nearest' / 'linear' (body_xy.py:1633-1702,
// 1855-1904): the value at pixel coordinates (x, y) of plane `pl`, NaN where the reference gives NaN.
//
// `ld(i)` loads pixel i of the plane as a double: straight from the plane (PlaneLoader) or through
// the table of fetched blocks (BlockLoader, the sparse host path below). `img` is the
// plane itself, which the rare NaN pre-clean reads around a non-finite pixel.
template <typename T>
struct PlaneLoader {
    const T *img;
    __device__ __forceinline__ double operator()(size_t i) const { return (double)img[i]; }
    // pixels i and i + 1 with ONE load instruction (element-aligned only; the hardware takes that)
    __device__ __forceinline__ void pair(size_t i, double &lo, double &hi) const
    {
        struct P2 {
            T a, b;
        } v;
        __builtin_memcpy(&v, img + i, sizeof(P2));
        lo = (double)v.a;
        hi = (double)v.b;
    }
};

template <typename T, typename L>
__device__ __forceinline__ double reproject_sample_from(const ReprojectArgs &a, int pl, const T *img, const L &ld, double x,
                                                        double y)
{
    const double nan = __builtin_nan("");
    const int nx = a.nx, ny = a.ny;
    double val = nan;
    if (!isnan(x)) {
        if (a.interpolation == PM_INTERP_NEAREST) {
            // _do_nearest_interpolation body_xy.py:1633: np.round = half to even
            long xi = (long)rint(x), yi = (long)rint(y);
            if (xi < 0) xi += nx;
            if (yi < 0) yi += ny;
            // maps made by pm_xy_map are always inside the frame; a caller-supplied map that is
            // not (the reference raises IndexError there) must not read outside the plane
            if (xi >= 0 && xi < nx && yi >= 0 && yi < ny && !isnan(y)) val = ld((size_t)yi * nx + xi);
        } else {
            const bool have_stats = a.plane_stats != nullptr;
            bool skip = have_stats && a.plane_stats[pl].all_nan;  // body_xy.py:1668-1670
            // _should_propagate_nan_to_map body_xy.py:1855-1866, first half: outside the hull of
            // the pixel centres
            if (a.propagate_nan && (x < 0.0 || y < 0.0 || x > nx - 1 || y > ny - 1)) skip = true;
            if (!skip) {
                // RectBivariateSpline(kx=ky=1, s=0).ev == bilinear; FITPACK clamps the
                // evaluation point to the knot range.
                double xc = fmin(fmax(x, 0.0), nx - 1.0), yc = fmin(fmax(y, 0.0), ny - 1.0);
                long x0 = (long)floor(xc), y0 = (long)floor(yc);
                if (x0 > nx - 2) x0 = nx - 2;
                if (y0 > ny - 2) y0 = ny - 2;
                if (x0 < 0) x0 = 0;
                if (y0 < 0) y0 = 0;
                long x1 = x0 + 1 < nx ? x0 + 1 : x0, y1 = y0 + 1 < ny ? y0 + 1 : y0;
                double fx = xc - (double)x0, fy = yc - (double)y0;
                // Corners with zero weight contribute exactly 0 in the reference (their cleaned
                // value is finite) and are not touched. The corners WITH weight are exactly the
                // floor / ceil pixels of the reference's NaN test (second half of
                // _should_propagate_nan_to_map): one set of loads serves both.
                const bool u00 = fx != 1.0 && fy != 1.0, u01 = fx != 0.0 && fy != 1.0;
                const bool u10 = fx != 1.0 && fy != 0.0, u11 = fx != 0.0 && fy != 0.0;
                // (the two pixels of a row are neighbours in memory: one load each row where both count)
                double r00 = 0.0, r01 = 0.0, r10 = 0.0, r11 = 0.0;
                const bool adj = x1 == x0 + 1;
                if (u00 && u01 && adj) {
                    ld.pair((size_t)y0 * nx + x0, r00, r01);
                } else {
                    if (u00) r00 = ld((size_t)y0 * nx + x0);
                    if (u01) r01 = ld((size_t)y0 * nx + x1);
                }
                if (u10 && u11 && adj) {
                    ld.pair((size_t)y1 * nx + x0, r10, r11);
                } else {
                    if (u10) r10 = ld((size_t)y1 * nx + x0);
                    if (u11) r11 = ld((size_t)y1 * nx + x1);
                }
                if (a.propagate_nan && (isnan(r00) || isnan(r01) || isnan(r10) || isnan(r11))) {
                    skip = true;
                } else {
                    const double med = have_stats ? a.plane_stats[pl].median : 0.0;
                    bool nm = false;
                    const double v00 = u00 ? cleaned_value(img, r00, y0, x0, ny, nx, med, nm) : 0.0;
                    const double v01 = u01 ? cleaned_value(img, r01, y0, x1, ny, nx, med, nm) : 0.0;
                    const double v10 = u10 ? cleaned_value(img, r10, y1, x0, ny, nx, med, nm) : 0.0;
                    const double v11 = u11 ? cleaned_value(img, r11, y1, x1, ny, nx, med, nm) : 0.0;
                    {
                        // no fused multiply-adds here: every instantiation of this function (resident,
                        // zero-copy, block table) then blends to the same bits whatever the compiler
                        // makes of the code around it
#pragma clang fp contract(off)
                        const double gx = 1.0 - fx, gy = 1.0 - fy;
                        const double top = gx * v00 + fx * v01;
                        const double bot = gx * v10 + fx * v11;
                        val = gy * top + fy * bot;
                    }
                    if (nm && !have_stats) atomicMax(&a.plane_flags[pl], a.seq);
                }
            }
        }
    }
    return val;
}

template <typename T>
__device__ __forceinline__ double reproject_sample(const ReprojectArgs &a, int pl, double x, double y)
{
    const T *img = (const T *)a.cube + (size_t)pl * a.ny * a.nx;
    return reproject_sample_from<T>(a, pl, img, PlaneLoader<T>{img}, x, y);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_reproject(const ReprojectArgs a)
{
    const int m = blockIdx.x * kBlock + threadIdx.x;
    const int pl = blockIdx.y;
    if (m >= a.n_map) return;
    a.out[(size_t)pl * a.n_map + m] = reproject_sample<T>(a, pl, a.x_map[m], a.y_map[m]);
}

// A plane per XCD (planes in whole groups of 8; the rest take k_reproject). Workgroups are dealt round-robin to the 8 XCDs;
// here the workgroups an XCD receives are those of ONE plane at a time (plane = 8 * (j / chunks) + xcd): the 64-byte
// lines that neighbouring map rows share are then found in that XCD's L2 instead of being fetched by several of them.
// Round 6, config 5 (512 planes of 1024^2 -> 1 deg): 0.459 -> 0.436 ms (same box, alternating runs); same samples, same bits.
// (Two / four planes of the same cell per lane - more requests in flight per wave - were slower: 0.461 -> 0.478 / 0.494 ms.)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_reproject_xcd(const ReprojectArgs a, int chunks)
{
    const unsigned wg = blockIdx.x, xcd = wg & 7u, j = wg >> 3;
    const int pl = (int)((j / (unsigned)chunks) * 8u + xcd);
    const int m = (int)(j % (unsigned)chunks) * kBlock + threadIdx.x;
    if (pl >= a.n_planes || m >= a.n_map) return;
    a.out[(size_t)pl * a.n_map + m] = reproject_sample<T>(a, pl, a.x_map[m], a.y_map[m]);
}

// ------------------------------------------------------------------ sparse host path: block table
// A cube in HOST memory mapped onto a coarse grid: what counts is how many bytes cross PCIe and in
// which request sizes (tools/probes/hip/probe_gather.hip: isolated 128-byte lines 41 GB/s, runs of 256 bytes
// and more 55 GB/s; the in-place gather of k_reproject - uncached, so neighbouring waves fetch the
// same lines again - 32 GB/s of lines). The map touches a fraction of each plane, the SAME blocks in
// every plane. k_mark_blocks runs the sampling arithmetic once and flags those blocks; k_blocks_*
// number them (block <-> row of the table); then either CPU threads collect the 16-byte blocks of each chunk
// of planes into pinned staging and one DMA brings the dense table over, or k_fetch_blocks pulls
// 128-byte blocks (PM_OPT_FETCH_BLOCK_BYTES) from pinned memory, each once, 8 lanes on one block; k_reproject_blocks samples
// the table.

// the "load" of the marking pass: flags the block of pixel i, returns a finite value (so that the
// sampler takes its ordinary path and touches exactly the corners the real pass will load)
template <typename T>
struct MarkLoader {
    unsigned char *flags;
    int shift;
    __device__ __forceinline__ double operator()(size_t i) const
    {
        flags[(i * sizeof(T)) >> shift] = 1;
        return 0.0;
    }
    __device__ __forceinline__ void pair(size_t i, double &lo, double &hi) const
    {
        lo = (*this)(i);
        hi = (*this)(i + 1);
    }
};

template <typename T>
__global__ __launch_bounds__(kBlock) void k_mark_blocks(const ReprojectArgs a, unsigned char *flags, int shift)
{
    const int m = blockIdx.x * kBlock + threadIdx.x;
    if (m >= a.n_map) return;
    const double v = reproject_sample_from<T>(a, 0, (const T *)nullptr, MarkLoader<T>{flags, shift}, a.x_map[m], a.y_map[m]);
    (void)v;
}

// ---- numbering the flagged blocks (block -> row of the table, row -> block), on the device: the
// host only needs the count. A tile is 4096 flags: 256 lanes x 16 flags (flags are 0 / 1 bytes, read
// as four 32-bit words; the flag array is padded with zeros to whole tiles).
constexpr int kNumberTile = 4096;

__device__ __forceinline__ int flags16(const unsigned char *flags, size_t tile, unsigned lane, uint4 &w)
{
    w = *(const uint4 *)(flags + tile * kNumberTile + (size_t)lane * 16);
    return __builtin_popcount(w.x) + __builtin_popcount(w.y) + __builtin_popcount(w.z) + __builtin_popcount(w.w);
}

// inclusive prefix sum of v over the 256 lanes of the workgroup (wave shuffles + one LDS hop)
__device__ __forceinline__ int block_inclusive_scan(int v, int *wave_tot /* LDS, 4 ints */)
{
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(v, d, 64);
        if (lane >= (unsigned)d) v += up;
    }
    if (lane == 63u) wave_tot[wave] = v;
    __syncthreads();
    int base = 0;
    for (unsigned w = 0; w < wave; w++) base += wave_tot[w];
    __syncthreads();
    return v + base;
}

__global__ __launch_bounds__(kBlock) void k_blocks_count(const unsigned char *__restrict__ flags, int *__restrict__ tile_sums)
{
    __shared__ int wave_tot[4];
    uint4 w;
    const int c = flags16(flags, blockIdx.x, threadIdx.x, w);
    const int incl = block_inclusive_scan(c, wave_tot);
    if (threadIdx.x == kBlock - 1) tile_sums[blockIdx.x] = incl;
}

// one workgroup: tile_sums -> exclusive offsets (in place), *total = number of flagged blocks
__global__ __launch_bounds__(kBlock) void k_blocks_offsets(int *__restrict__ tile_sums, int n_tiles, int *__restrict__ total)
{
    __shared__ int wave_tot[4];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int t0 = 0; t0 < n_tiles; t0 += kBlock) {
        const int i = t0 + (int)threadIdx.x;
        const int v = i < n_tiles ? tile_sums[i] : 0;
        const int incl = block_inclusive_scan(v, wave_tot);
        const int carry = carry_s;
        if (i < n_tiles) tile_sums[i] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == kBlock - 1) carry_s = carry + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

__global__ __launch_bounds__(kBlock) void k_blocks_number(const unsigned char *__restrict__ flags, const int *__restrict__ tile_offsets,
                                                          size_t n_blk, int *__restrict__ blkmap, int *__restrict__ blklist)
{
    __shared__ int wave_tot[4];
    uint4 w;
    const int c = flags16(flags, blockIdx.x, threadIdx.x, w);
    int row = tile_offsets[blockIdx.x] + block_inclusive_scan(c, wave_tot) - c;
    const size_t first = (size_t)blockIdx.x * kNumberTile + (size_t)threadIdx.x * 16;
    const unsigned words[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const size_t i = first + k;
        if (i >= n_blk) break;
        const bool set = (words[k >> 2] >> ((k & 3) * 8)) & 0xffu;
        blkmap[i] = set ? row : -1;
        if (set) blklist[row++] = (int)i;
    }
}

// 128-bit fingerprint of an x/y map (the key under which a context keeps its block table between
// calls, as the reference keeps _get_xy_map, body_xy.py:3478): two independent sums of a 64-bit
// finaliser over (bits of x, bits of y, cell index). out[2] is zeroed by the caller.
__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(kBlock) void k_hash_maps(const double *__restrict__ x_map, const double *__restrict__ y_map, int n,
                                                      unsigned long long *__restrict__ out)
{
    unsigned long long h0 = 0, h1 = 0;
    for (int m = blockIdx.x * kBlock + threadIdx.x; m < n; m += gridDim.x * kBlock) {
        const unsigned long long bx = (unsigned long long)__double_as_longlong(x_map[m]);
        const unsigned long long by = (unsigned long long)__double_as_longlong(y_map[m]);
        const unsigned long long i = (unsigned long long)m;
        h0 += mix64(bx + 0x9e3779b97f4a7c15ull * (2 * i + 1)) ^ mix64(by ^ (0xd1b54a32d192ed03ull * (i + 1)));
        h1 += mix64((bx ^ 0x8cb92ba72f3d8dd7ull) * (2 * i + 3) + by) + mix64(by + (i << 32 | (i >> 3)) + 0x2545f4914f6cdd1dull);
    }
    for (int d = 32; d > 0; d >>= 1) {
        h0 += __shfl_down(h0, d, 64);
        h1 += __shfl_down(h1, d, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(out, h0);
        atomicAdd(out + 1, h1);
    }
}

// blockIdx.y = plane of the chunk; 16 bytes per lane: 16 lanes per 256-byte block (4 blocks per wave), 8 per 128-byte block
__global__ __launch_bounds__(kBlock) void k_fetch_blocks(const char *__restrict__ cube, const BlockTable t)
{
    const unsigned q = blockIdx.x * kBlock + threadIdx.x;
    const int lane_shift = t.shift - 4;
    const unsigned row = q >> lane_shift, piece = q & ((1u << lane_shift) - 1u);
    if (row >= t.n_list) return;
    const size_t pl = blockIdx.y;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *src = (const u32x4 *)(cube + pl * t.plane_bytes + ((size_t)t.blklist[row] << t.shift)) + piece;
    u32x4 *dst = (u32x4 *)(t.table + ((pl * t.n_list + row) << t.shift)) + piece;
    *dst = *src;
}

template <typename T>
struct BlockLoader {
    const T *img;       // the plane in host memory (blocks the list does not hold), or null
    const int *blkmap;
    const char *rows;   // this plane's rows of the table
    int shift;
    __device__ __forceinline__ double operator()(size_t i) const
    {
        const size_t byte = i * sizeof(T);
        const int row = blkmap[byte >> shift];
        if (row < 0 && !img) return __builtin_inf();  // (not reached: the marking pass ran this very code; inf has the plane redone)
        const T *p = row >= 0 ? (const T *)(rows + ((size_t)row << shift) + (byte & ((1u << shift) - 1))) : img + i;
        return (double)*p;
    }
    __device__ __forceinline__ void pair(size_t i, double &lo, double &hi) const
    {
        lo = (*this)(i);
        hi = (*this)(i + 1);
    }
};

template <typename T>
__global__ __launch_bounds__(kBlock) void k_reproject_blocks(const ReprojectArgs a, const BlockTable t)
{
    const int m = blockIdx.x * kBlock + threadIdx.x;
    const int pl = blockIdx.y;
    if (m >= a.n_map) return;
    const T *img = a.cube ? (const T *)a.cube + (size_t)pl * a.ny * a.nx : nullptr;
    const BlockLoader<T> ld{img, t.blkmap, t.table + (((size_t)pl * t.n_list) << t.shift), t.shift};
    a.out[(size_t)pl * a.n_map + m] = reproject_sample_from<T>(a, pl, img, ld, a.x_map[m], a.y_map[m]);
}

// Observation.get_mapped_data in ONE launch for a handful of planes (observation.py:826-905: the x/y
// map of the grid, then map_img of every plane): one lane per map cell computes its pixel
// coordinates (map_cell_xy), stores them (the x/y map stays available, e.g. for the replay of planes
// that need their nanmedian) and samples the planes one after the other. Saves the launch boundary
// and the x/y-map round trip through memory between k_map_xy and k_reproject - a fifth of the time
// of those two small kernels in the frame benchmark; for cubes of many planes the two-kernel form
// (planes on the grid's y axis) has the parallelism this form lacks.
template <typename T>
__global__ __launch_bounds__(kSphBlock) void k_mapped_data(const Params p, const ReprojectArgs a, const double *__restrict__ lon_in,
                                                           const double *__restrict__ lat_in, double *__restrict__ x_map_out,
                                                           double *__restrict__ y_map_out)
{
    const KParams kp = kernarg_params();
    const int m = blockIdx.x * kSphBlock + threadIdx.x;
    if (m >= a.n_map) return;
    double x, y;
    map_cell_xy(kp, lon_in[m], lat_in[m], x, y);
    x_map_out[m] = x;
    y_map_out[m] = y;
    for (int pl = 0; pl < a.n_planes; pl++) a.out[(size_t)pl * a.n_map + m] = reproject_sample<T>(a, pl, x, y);
}

// ------------------------------------------------------------------ spline reprojection
// scipy RectBivariateSpline(kx, ky, s=0).ev of BodyXY._do_spline_interpolation
// (body_xy.py:1651-1702) for 'quadratic', 'cubic' and (k0, k1): interpolating tensor-product
// B-spline. Pipeline per chunk of planes: k_spline_clean (NaN-cleaned float64 copy; k_median_* for the planes whose clean
// values need the plane nanmedian, pm_launch_clean_lazy) -> k_spline_solve_cols (axis 0), k_spline_solve_rows (axis 1)
// (banded LU substitution, in place: samples -> coefficients) -> k_spline_eval.

// 'smooth' interpolation (BodyXY._do_smooth_interpolation / _pchip_grid_interp2d
// body_xy.py:1704-1853). The reference materialises the whole oversampled image (up to
// 10000 x 10000 per plane) and then samples it bilinearly at the map cells. PCHIP is local
// (a piece depends on four samples), so here each (cell, plane) lane evaluates just the four
// fine-grid nodes around its sample: node (r, k) = column PCHIP at ys[r] over the rows whose
// row PCHIP at xs[k] is finite, each of those a PCHIP over the finite pixels of the row.
// Work scales with the map, not with the oversampled image, and nothing is staged in HBM.
// Two kernels. k_reproject_smooth_window does the common case, some five times cheaper than the general one: the fine grid
// is finer than the pixels, so the two fine nodes of an axis lie in one pixel cell [j, j+1] (the second possibly ON its far
// edge, where a PCHIP returns the sample itself), and the 4 x 4 pixels around the cell are finite and inside the trimmed
// range. Then all four nodes are built from the same four row pieces (coefficients once per row, evaluated at both xq)
// and two column pieces - 6 sets of coefficients instead of 20, with the arithmetic of pchip_piece. A lane keeps its cell
// for sm.planes_per_lane planes (blockIdx.y counts groups of planes): where the cell lies on the fine grid - two interval
// searches, four nodes, the bilinear weights - is the same for every plane. A (cell, plane) it cannot do (gaps, the
// range's edge) gets kSmoothRedo, a NaN no arithmetic produces, and the workgroup's byte of `redo` is set;
// k_reproject_smooth_gaps, the gap-aware form, then visits the flagged workgroups and computes those.
// (One kernel with both forms ran at the registers of the general one - 80, and 203 with the loop over planes.)
constexpr unsigned long long kSmoothRedo = 0x7ff85a5a5a5a5a5aull;

struct SmoothCell {
    double xk0, xk1, yr0, yr1, fx, fy;
    bool inside;
    __device__ __forceinline__ SmoothCell(const ReprojectArgs &a, const SmoothArgs &sm, double x, double y)
    {
        // propagate_nan: NaN where the sample lies outside the image (or one of the up to four pixels around it is NaN)
        const bool skip = isnan(x) || (a.propagate_nan && (x < 0.0 || y < 0.0 || x > a.nx - 1 || y > a.ny - 1));
        // RegularGridInterpolator(bounds_error=False, fill_value=nan)
        inside = !skip && x >= (double)sm.x.first && x <= (double)sm.x.last && y >= (double)sm.y.first && y <= (double)sm.y.last;
        const int k = inside ? smooth_interval(sm.x, x) : 0, r = inside ? smooth_interval(sm.y, y) : 0;
        xk0 = smooth_grid(sm.x, k), xk1 = smooth_grid(sm.x, k + 1);
        yr0 = smooth_grid(sm.y, r), yr1 = smooth_grid(sm.y, r + 1);
        fx = (x - xk0) / (xk1 - xk0), fy = (y - yr0) / (yr1 - yr0);
    }
    __device__ __forceinline__ double blend(double f00, double f01, double f10, double f11) const
    {
        return f00 * (1.0 - fy) * (1.0 - fx) + f01 * (1.0 - fy) * fx + f10 * fy * (1.0 - fx) + f11 * fy * fx;
    }
};

template <typename T>
__global__ __launch_bounds__(kBlock) void k_reproject_smooth_window(const ReprojectArgs a, const SmoothArgs sm, unsigned *redo)
{
    const int m = blockIdx.x * kBlock + threadIdx.x;
    const int nx = a.nx;
    const bool live = m < a.n_map;
    const SmoothCell c(a, sm, live ? a.x_map[m] : __builtin_nan(""), live ? a.y_map[m] : __builtin_nan(""));
    const int jx = (int)floor(c.xk0), iy = (int)floor(c.yr0);
    const bool window = c.inside && c.xk1 <= jx + 1.0 && c.yr1 <= iy + 1.0 && jx - 1 >= sm.x.first && jx + 2 <= sm.x.last &&
                        iy - 1 >= sm.y.first && iy + 2 <= sm.y.last;
    const double sx0 = c.xk0 - (double)jx, sx1 = c.xk1 - (double)jx, sy0 = c.yr0 - (double)iy, sy1 = c.yr1 - (double)iy;
    const bool x_edge = c.xk1 == jx + 1.0, y_edge = c.yr1 == iy + 1.0;
    const size_t corner = window ? (size_t)(iy - 1) * nx + (jx - 1) : 0;
    bool todo = false;
    for (int q = 0; q < sm.planes_per_lane; q++) {
        const int pl = blockIdx.y * sm.planes_per_lane + q;
        if (pl >= a.n_planes || !live) break;
        const T *img = (const T *)a.cube + (size_t)pl * a.ny * nx;
        double val = __builtin_nan("");
        if (c.inside) {
            bool fast = window;
            if (fast) {
                // (propagate_nan's test of the pixels around the sample is made with this one: floor(x) and ceil(x) are jx
                //  or jx + 1, those pixels are among the sixteen)
                double g0[4], g1[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const size_t at = corner + (size_t)i * nx;
                    double wa, wb, wc, wd;
                    load4_as_f64(img, at, wa, wb, wc, wd);
                    const PchipUnitPiece row(wa, wb, wc, wd);
                    g0[i] = row(sx0);
                    g1[i] = x_edge ? wc : row(sx1);
                    fast = fast && isfinite(wa) && isfinite(wb) && isfinite(wc) && isfinite(wd) && isfinite(g0[i]) && isfinite(g1[i]);
                }
                const PchipUnitPiece c0(g0[0], g0[1], g0[2], g0[3]), c1(g1[0], g1[1], g1[2], g1[3]);
                val = c.blend(c0(sy0), c1(sy0), y_edge ? g0[2] : c0(sy1), y_edge ? g1[2] : c1(sy1));
            }
            if (!fast) {
                val = __longlong_as_double((long long)kSmoothRedo);
                todo = true;
            }
        }
        a.out[(size_t)pl * a.n_map + m] = val;
    }
    // (the order of the list is that of arrival; the results do not depend on it)
    const int any = __syncthreads_or(todo);
    if (threadIdx.x == 0 && any) redo[1 + atomicAdd(redo, 1u)] = blockIdx.y * gridDim.x + blockIdx.x;
}

template <typename T>
__device__ __forceinline__ void smooth_gaps_block(const ReprojectArgs &a, const SmoothArgs &sm, bool flagged_only, unsigned bx, unsigned group)
{
    const int m = bx * kBlock + threadIdx.x;
    if (m >= a.n_map) return;
    const double nan = __builtin_nan("");
    const int nx = a.nx, ny = a.ny;
    const double x = a.x_map[m], y = a.y_map[m];
#pragma unroll 1
    for (int q = 0; q < sm.planes_per_lane; q++) {
        const int pl = group * sm.planes_per_lane + q;
        if (pl >= a.n_planes) break;
        double *out = a.out + (size_t)pl * a.n_map + m;
        if (flagged_only && (unsigned long long)__double_as_longlong(*out) != kSmoothRedo) continue;
        const T *img = (const T *)a.cube + (size_t)pl * ny * nx;
        const SmoothCell c(a, sm, x, y);
        bool skip = !c.inside;
        if (!skip && a.propagate_nan) {
            const long ia = (long)fmax(floor(x), 0.0), ib = (long)fmin(ceil(x), nx - 1.0);
            const long ja = (long)fmax(floor(y), 0.0), jb = (long)fmin(ceil(y), ny - 1.0);
            skip = isnan(load_as_f64(img, (size_t)ja * nx + ia)) || isnan(load_as_f64(img, (size_t)ja * nx + ib)) ||
                   isnan(load_as_f64(img, (size_t)jb * nx + ia)) || isnan(load_as_f64(img, (size_t)jb * nx + ib));
        }
        double val = nan;
        if (!skip) {
            auto node = [&](double xq, double yq) {
                auto column = [&](int i) {
                    auto row = [&](int j) { return load_as_f64(img, (size_t)i * nx + j); };
                    return pchip_gappy(row, sm.x.first, sm.x.last, xq);
                };
                return pchip_gappy(column, sm.y.first, sm.y.last, yq);
            };
            const double f00 = node(c.xk0, c.yr0), f01 = node(c.xk1, c.yr0), f10 = node(c.xk0, c.yr1), f11 = node(c.xk1, c.yr1);
            val = c.blend(f00, f01, f10, f11);
        }
        *out = val;
    }
}

// The workgroups of the window kernel's grid (grid_x by n_groups) listed in redo[1 .. redo[0]], a fixed number of blocks
// striding through the list; redo == nullptr: every (cell, plane) (PM_OPT_GENERAL_KERNEL: the cross-check of the window form).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_reproject_smooth_gaps(const ReprojectArgs a, const SmoothArgs sm, const unsigned *redo,
                                                                  unsigned grid_x, unsigned n_groups)
{
    const unsigned count = redo ? redo[0] : grid_x * n_groups;
    for (unsigned at = blockIdx.x; at < count; at += gridDim.x) {
        const unsigned id = redo ? redo[1 + at] : at;
        smooth_gaps_block<T>(a, sm, redo != nullptr, id % grid_x, id / grid_x);
    }
}

// nanmin / nanmax of the x and y maps: limits[0..3] = xmin, xmax, ymin, ymax; +inf / -inf when no cell is visible.
// Two stages - every block leaves its four values in partial[4 * block], a last block folds those. (Round 5: the single
// block this used to be took 3.5 ms over the 6.5 M cells of a 0.1 deg map - as long as the smooth interpolation of 32 planes.)
__device__ __forceinline__ void limits_fold(double (&sh)[4][kBlock], double xmin, double xmax, double ymin, double ymax, double *out)
{
    sh[0][threadIdx.x] = xmin;
    sh[1][threadIdx.x] = xmax;
    sh[2][threadIdx.x] = ymin;
    sh[3][threadIdx.x] = ymax;
    __syncthreads();
    for (int st = kBlock / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            sh[0][threadIdx.x] = fmin(sh[0][threadIdx.x], sh[0][threadIdx.x + st]);
            sh[1][threadIdx.x] = fmax(sh[1][threadIdx.x], sh[1][threadIdx.x + st]);
            sh[2][threadIdx.x] = fmin(sh[2][threadIdx.x], sh[2][threadIdx.x + st]);
            sh[3][threadIdx.x] = fmax(sh[3][threadIdx.x], sh[3][threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) out[threadIdx.x] = sh[threadIdx.x][0];
}
__global__ __launch_bounds__(kBlock) void k_map_limits(const double *x_map, const double *y_map, int n, double *partial)
{
    __shared__ double sh[4][kBlock];
    double xmin = __builtin_inf(), xmax = -__builtin_inf(), ymin = __builtin_inf(), ymax = -__builtin_inf();
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)n; i += (size_t)gridDim.x * kBlock) {
        const double x = x_map[i], y = y_map[i];
        if (!isnan(x)) {
            xmin = fmin(xmin, x);
            xmax = fmax(xmax, x);
        }
        if (!isnan(y)) {
            ymin = fmin(ymin, y);
            ymax = fmax(ymax, y);
        }
    }
    limits_fold(sh, xmin, xmax, ymin, ymax, partial + 4 * (size_t)blockIdx.x);
}
__global__ __launch_bounds__(kBlock) void k_map_limits_fold(const double *partial, int n_blocks, double *limits)
{
    __shared__ double sh[4][kBlock];
    double xmin = __builtin_inf(), xmax = -__builtin_inf(), ymin = __builtin_inf(), ymax = -__builtin_inf();
    for (int b = threadIdx.x; b < n_blocks; b += kBlock) {  // (fmin / fmax drop nothing here: partials are never NaN)
        xmin = fmin(xmin, partial[4 * b]);
        xmax = fmax(xmax, partial[4 * b + 1]);
        ymin = fmin(ymin, partial[4 * b + 2]);
        ymax = fmax(ymax, partial[4 * b + 3]);
    }
    limits_fold(sh, xmin, xmax, ymin, ymax, limits);
}

// mode 0: `stats` hold the plane medians. The lazy form (pm_launch_clean_lazy) - a plane's nanmedian, eight passes over
// it, is needed only where a non-finite pixel has no finite neighbour, which most data never has: mode 1 cleans with a
// provisional 0.0 and flags the planes in which some pixel took it (PlaneStats::needs_median); the median kernels then
// run for flagged planes only (the blocks of the others leave at once), and mode 2 cleans the flagged planes again.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_spline_clean(const T *cube, double *work, PlaneStats *stats, int ny, int nx, int mode)
{
    const size_t npx = (size_t)ny * nx;
    const int pl = blockIdx.y;
    if (mode == 2 && !stats[pl].needs_median) return;
    const double median = mode == 1 ? 0.0 : stats[pl].median;
    bool nm = false;
    // (grid-stride: the redo pass of the lazy form is launched with a few blocks per plane - nearly all of them leave at once)
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < npx; i += (size_t)gridDim.x * kBlock)
        work[(size_t)pl * npx + i] = cleaned_at(cube + (size_t)pl * npx, (long)(i / nx), (long)(i % nx), ny, nx, median, nm);
    if (mode == 1 && nm) atomicOr(&stats[pl].needs_median, 1);
}

// Solve B c = v along one axis of every plane in place, with the banded LU of the collocation matrix (unit lower
// triangle, no pivoting, the diagonal of U stored as its reciprocal: forward substitution, then back substitution with the
// last k results kept in registers).
// A line is 1024 dependent steps; what the kernels are built around is keeping a wave from WAITING at each of them.
//   * the LU row of a step is the same for every lane: fetched per step it is a scalar load that misses (the table of a
//     1024-sample axis is 57 KB, every wave is somewhere else in it) - one memory round trip per step. Here the rows of
//     16 steps come in with the data, 1-3 values per lane, and are read back from LDS;
//   * the next 16 samples of every line (and their LU rows) are in flight from HBM into registers while the current 16
//     are worked on;
//   * axis 1 (lines = image rows: a lane's line is contiguous, the lanes' lines 8 KB apart) goes through a 64 x 16 tile
//     in LDS: every global access is whole 128-byte lines - four row segments per instruction - instead of 64 different
//     lines per load instruction, the substitution runs lane-per-row on the tile's 16 values held in registers.
// Round 4, 128 planes of 1024^2: axis 0 1.18 -> 0.91 ms, axis 1 2.79 -> 0.91 ms (4.7 TB/s of the samples read and written
// twice). (A first tiled axis 1 WITHOUT the staged LU rows ran in the 2.8 ms of the plain lane-per-row kernel: the round
// trip per step was the cost, not the access pattern - profiles/EXPERIMENTS.md.)
constexpr int kSolveRows = 64, kSolveCols = 16, kSolveBand = 11;  // (2 k + 1 <= 11)
constexpr int kSolveRpi = 64 / kSolveCols;  // rows of a tile per load instruction of the axis-1 kernel
// A workgroup of these kernels is ONE wave: what its lanes exchange through LDS needs the order of the wave's own LDS
// instructions, which the hardware keeps, and no barrier. __syncthreads() also waits for every global store the wave has
// issued (vmcnt(0)) - here the 16 rows just written, a full memory round trip per tile that nothing else hides when a
// plane is alone: one plane of 1024^2 took 0.65 + 0.57 ms in the two solves.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// The 16 steps of a tile (cnt < 16: the last tile of a line), straight-line: the degree is a template argument, and the terms
// that would reach outside the matrix need no test - the band is stored with zeros there and `prev` starts as zeros, so they
// subtract an exact 0. (With the degree and those tests at run time every multiply-add sat in a basic block of its own
// behind an LDS read it waited for: ~60 cycles per term, 700 per step - the whole cost of a plane that is alone.)
// The terms are taken farthest first: the result of the step before enters the LAST multiply-add, so a step's place in
// the line's chain of dependent operations is one multiply-add (and one multiplication by the reciprocal diagonal on the
// way back), whatever the degree. Eight steps of a full tile at a time: their LU entries are read from LDS together, AHEAD
// of the chain (left to itself the compiler reads each step's entries right where it uses them, and every step waits for
// LDS). One definition for the line solves and the segmented ones: the same instructions, the same bits.
template <int K>
__device__ __forceinline__ void solve_fwd_tile(double (&v)[kSolveCols], double (&prev)[5], const double (*lu)[kSolveBand], int cnt)
{
    constexpr int kRun = 8;
    if (cnt == kSolveCols) {
#pragma unroll
        for (int j0 = 0; j0 < kSolveCols; j0 += kRun) {
            double l[kRun][K];
#pragma unroll
            for (int j = 0; j < kRun; j++)
#pragma unroll
                for (int q = 1; q <= K; q++) l[j][q - 1] = lu[j0 + j][K - q];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < kRun; j++) {
                double s = v[j0 + j];
#pragma unroll
                for (int q = K; q >= 1; q--) s -= l[j][q - 1] * prev[q - 1];
#pragma unroll
                for (int q = 4; q > 0; q--) prev[q] = prev[q - 1];
                prev[0] = s;
                v[j0 + j] = s;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < kSolveCols; j++)
            if (j < cnt) {
                double s = v[j];
#pragma unroll
                for (int q = K; q >= 1; q--) s -= lu[j][K - q] * prev[q - 1];
#pragma unroll
                for (int q = 4; q > 0; q--) prev[q] = prev[q - 1];
                prev[0] = s;
                v[j] = s;
            }
    }
}
template <int K>
__device__ __forceinline__ void solve_back_tile(double (&v)[kSolveCols], double (&prev)[5], const double (*lu)[kSolveBand], int cnt)
{
    constexpr int kRun = 8;
    if (cnt == kSolveCols) {
#pragma unroll
        for (int j0 = kSolveCols - kRun; j0 >= 0; j0 -= kRun) {
            double l[kRun][K + 1];
#pragma unroll
            for (int j = kRun - 1; j >= 0; j--)
#pragma unroll
                for (int q = 0; q <= K; q++) l[j][q] = lu[j0 + j][K + q];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = kRun - 1; j >= 0; j--) {
                double s = v[j0 + j];
#pragma unroll
                for (int q = K; q >= 1; q--) s -= l[j][q] * prev[q - 1];
                s *= l[j][0];
#pragma unroll
                for (int q = 4; q > 0; q--) prev[q] = prev[q - 1];
                prev[0] = s;
                v[j0 + j] = s;
            }
        }
    } else {
#pragma unroll
        for (int j = kSolveCols - 1; j >= 0; j--)
            if (j < cnt) {
                double s = v[j];
#pragma unroll
                for (int q = K; q >= 1; q--) s -= lu[j][K + q] * prev[q - 1];
                s *= lu[j][K];
#pragma unroll
                for (int q = 4; q > 0; q--) prev[q] = prev[q - 1];
                prev[0] = s;
                v[j] = s;
            }
    }
}
template <int K>
__global__ __launch_bounds__(kSolveRows) void k_spline_solve_rows(double *work, int n_planes, int ny, int nx, SplineAxis ax, const PlaneStats *only_flagged)
{
    __shared__ double tile[kSolveRows][kSolveCols + 1];
    __shared__ double lu[kSolveCols][kSolveBand];
    const int lane = threadIdx.x;
    const int groups = (ny + kSolveRows - 1) / kSolveRows;
    const int pl = blockIdx.x / groups, r0 = (blockIdx.x % groups) * kSolveRows;
    if (only_flagged && !only_flagged[pl].needs_median) return;  // (the second round: planes that were cleaned again with their nanmedian)
    const int rows = min(kSolveRows, ny - r0);
    double *base = work + ((size_t)pl * ny + r0) * nx;
    const int n = ax.n;
    constexpr int w = 2 * K + 1;
    const int lr = lane / kSolveCols, lc = lane % kSolveCols;  // this lane's (row within a group of four, column) in the copies
    constexpr int kLuRegs = (kSolveCols * kSolveBand + kSolveRows - 1) / kSolveRows;
    double prev[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    double v[kSolveCols], g[kSolveRows / kSolveRpi], glu[kLuRegs];
    // the next tile is on its way from HBM (into registers) while the current one is worked on
    auto fetch = [&](int c0) {
        const int cols = min(kSolveCols, n - c0);
#pragma unroll
        for (int it = 0; it < kSolveRows / kSolveRpi; it++) {
            const int r = it * kSolveRpi + lr;
            g[it] = (r < rows && lc < cols) ? base[(size_t)r * nx + c0 + lc] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < kLuRegs; e++) {
            const int x = e * kSolveRows + lane;
            glu[e] = x < cols * w ? ax.lu[(size_t)c0 * w + x] : 0.0;
        }
    };
    auto to_lds = [&](int cols) {
#pragma unroll
        for (int it = 0; it < kSolveRows / kSolveRpi; it++) tile[it * kSolveRpi + lr][lc] = g[it];
#pragma unroll
        for (int e = 0; e < kLuRegs; e++) {
            const int x = e * kSolveRows + lane;
            if (x < cols * w) lu[x / w][x % w] = glu[e];
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < kSolveCols; j++) v[j] = tile[lane][j];
    };
    auto tile_out = [&](int c0, int cols) {
#pragma unroll
        for (int j = 0; j < kSolveCols; j++) tile[lane][j] = v[j];
        wave_sync();
#pragma unroll
        for (int it = 0; it < kSolveRows / kSolveRpi; it++) {
            const int r = it * kSolveRpi + lr;
            if (r < rows && lc < cols) base[(size_t)r * nx + c0 + lc] = tile[r][lc];
        }
        wave_sync();
    };
    // forward substitution (unit lower triangle)
    fetch(0);
    for (int c0 = 0; c0 < n; c0 += kSolveCols) {
        const int cols = min(kSolveCols, n - c0);
        to_lds(cols);
        if (c0 + kSolveCols < n) fetch(c0 + kSolveCols);
        solve_fwd_tile<K>(v, prev, lu, cols);
        tile_out(c0, cols);
    }
    // back substitution (the forward pass's stores of the last tile are complete: same wave, program order)
#pragma unroll
    for (int q = 0; q < 5; q++) prev[q] = 0.0;
    const int c_last = ((n - 1) / kSolveCols) * kSolveCols;
    fetch(c_last);
    for (int c0 = c_last; c0 >= 0; c0 -= kSolveCols) {
        const int cols = min(kSolveCols, n - c0);
        to_lds(cols);
        if (c0 - kSolveCols >= 0) fetch(c0 - kSolveCols);
        solve_back_tile<K>(v, prev, lu, cols);
        tile_out(c0, cols);
    }
}

// Axis 0 (lines = image columns) in the same style: one wave takes 64 adjacent columns of a plane, 16 rows at a time - the
// lanes' loads are coalesced as they stand (no transposition), the LU rows of the 16 steps come through LDS, the next 16
// rows are in flight while the current ones are worked on.
// The NaN pre-clean rides on the forward pass: the samples come from the cube in its own dtype, a non-finite one is
// replaced on the way in (cleaned_at: its 3 x 3 neighbours straight from the cube - rare, and cached) - the cleaned plane is
// never written and read back (two of the ten passes a cubic reprojection makes over a plane). MODE 1 cleans with a
// provisional 0.0 for pixels without a finite neighbour and flags their planes (PlaneStats::needs_median, the lazy form of
// k_spline_clean); MODE 2 does the flagged planes again with their nanmedian (the others' blocks leave at once).
template <typename T, int MODE, int K>
__global__ __launch_bounds__(kSolveRows) void k_spline_solve_cols(const T *cube, double *work, PlaneStats *stats, int n_planes, int ny, int nx,
                                                                  SplineAxis ax)
{
    __shared__ double lu[2][kSolveCols][kSolveBand];
    const int lane = threadIdx.x;
    const int groups = (nx + kSolveRows - 1) / kSolveRows;
    const int pl = blockIdx.x / groups, x = (blockIdx.x % groups) * kSolveRows + lane;
    if (MODE == 2 && !stats[pl].needs_median) return;
    const double median = MODE == 2 ? stats[pl].median : 0.0;
    const T *img = cube + (size_t)pl * ny * nx;
    bool nm = false;
    const bool live = x < nx;
    double *col = work + (size_t)pl * ny * nx + (live ? x : 0);
    const int n = ax.n;
    constexpr int w = 2 * K + 1;
    constexpr int kLuRegs = (kSolveCols * kSolveBand + kSolveRows - 1) / kSolveRows;
    double prev[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    double v[kSolveCols], g[kSolveCols], glu[kLuRegs];
    bool forward = true;  // the forward pass takes its samples from the cube, cleaned; the backward pass from `work`
    auto fetch = [&](int i0) {
        const int cnt = min(kSolveCols, n - i0);
        if (forward) {
            // (the raw samples: what a non-finite one becomes is decided when the tile is taken up - a test here would wait
            //  for every load of the tile that is meant to be in flight while the previous one is worked on)
#pragma unroll
            for (int j = 0; j < kSolveCols; j++) g[j] = (live && j < cnt) ? load_as_f64(img, (size_t)(i0 + j) * nx + x) : 0.0;
        } else {
#pragma unroll
            for (int j = 0; j < kSolveCols; j++) g[j] = (live && j < cnt) ? col[(size_t)(i0 + j) * nx] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < kLuRegs; e++) {
            const int q = e * kSolveRows + lane;
            glu[e] = q < cnt * w ? ax.lu[(size_t)i0 * w + q] : 0.0;
        }
    };
    auto stage = [&](int buf, int cnt, int i0) {
#pragma unroll
        for (int e = 0; e < kLuRegs; e++) {
            const int q = e * kSolveRows + lane;
            if (q < cnt * w) lu[buf][q / w][q % w] = glu[e];
        }
        if (forward) {
#pragma unroll
            for (int j = 0; j < kSolveCols; j++) v[j] = (live && j < cnt) ? cleaned_value(img, g[j], (long)(i0 + j), (long)x, ny, nx, median, nm) : 0.0;
        } else {
#pragma unroll
            for (int j = 0; j < kSolveCols; j++) v[j] = g[j];
        }
        wave_sync();
    };
    auto put = [&](int i0, int cnt) {
#pragma unroll
        for (int j = 0; j < kSolveCols; j++)
            if (live && j < cnt) col[(size_t)(i0 + j) * nx] = v[j];
    };
    int buf = 0;
    // forward substitution (unit lower triangle)
    fetch(0);
    for (int i0 = 0; i0 < n; i0 += kSolveCols, buf ^= 1) {
        const int cnt = min(kSolveCols, n - i0);
        stage(buf, cnt, i0);
        if (i0 + kSolveCols < n) fetch(i0 + kSolveCols);
        solve_fwd_tile<K>(v, prev, lu[buf], cnt);
        put(i0, cnt);
    }
    if (MODE == 1 && nm) {
        atomicOr(&stats[pl].needs_median, 1);
        stats[n_planes].needs_median = 1;  // (the element behind the planes': some plane of the launch asked)
    }
    // back substitution
    forward = false;
#pragma unroll
    for (int q = 0; q < 5; q++) prev[q] = 0.0;
    const int i_last = ((n - 1) / kSolveCols) * kSolveCols;
    fetch(i_last);
    for (int i0 = i_last; i0 >= 0; i0 -= kSolveCols, buf ^= 1) {
        const int cnt = min(kSolveCols, n - i0);
        stage(buf, cnt, i0);
        if (i0 - kSolveCols >= 0) fetch(i0 - kSolveCols);
        solve_back_tile<K>(v, prev, lu[buf], cnt);
        put(i0, cnt);
    }
}

// ------------------------------------------------------------------ the same solves for planes that are few and large
// One lane per line makes a line's 2 n steps ONE chain: a 4096^2 plane alone is 64 waves (of the chip's 1024 SIMDs), each
// waiting through 8192 dependent steps per axis. The factors of a B-spline collocation matrix forget: away from the ends
// of a line L's multipliers and U's rows are the constants of a Toeplitz band whose recursion has the symbol's roots INSIDE
// the unit circle (degree 2: 0.172, 3: 0.268, 4: 0.361, 5: 0.431 per step), so a substitution started `warm` steps early
// with zeros for the values before it carries, when it reaches the segment it is meant for, an error of root^warm
// (<= 1e-35 of the values with spline_warm's lengths) - so far below half an ulp that it changes the rounding of no
// operation: the steps of the segment are the serial substitution's operations on the serial substitution's operands, and
// the results its bits (tests/test_gpu_splines_cube_scale.py holds the two forms to each other bit for bit). A line is cut into segments of `seg`
// samples (a multiple of the tile of 16), one wave takes 64 lines of one segment: forward from `warm` before the segment
// (from the line's start if that is nearer: exact), backward from `warm` beyond it (from the line's end if the start would
// fall into its last tile). The backward pass of a segment needs the forward results of the `warm` samples beyond it, which
// are another wave's: forward and backward are two launches, from one buffer into the other (pm_launch_spline, stages 4-7).
// where a segment's backward pass starts: the tile that ends `warm` beyond the segment, or the line's last tile
__device__ __forceinline__ int seg_back_top(int n, int e0, int warm)
{
    const int i_last = ((n - 1) / kSolveCols) * kSolveCols;
    return e0 + warm >= i_last ? i_last : e0 + warm - kSolveCols;
}

// Axis 0, one direction of one segment of 64 adjacent columns. Forward: samples from the cube, cleaned (MODE as in
// k_spline_solve_cols), results into `dst`; backward: from `src` into `dst`.
template <typename T, int MODE, int K, bool BACK>
__global__ __launch_bounds__(kSolveRows) void k_spline_seg_cols(const T *cube, const double *src, double *dst, PlaneStats *stats, int n_planes, int ny,
                                                                int nx, SplineAxis ax, int seg, int warm)
{
    __shared__ double lu[2][kSolveCols][kSolveBand];
    const int lane = threadIdx.x;
    const int n = ax.n;
    const int groups = (nx + kSolveRows - 1) / kSolveRows, nseg = (n + seg - 1) / seg;
    unsigned b = blockIdx.x;
    const int sg = b % nseg;
    b /= nseg;
    const int pl = b / groups, x = (b % groups) * kSolveRows + lane;
    if (MODE == 2 && !stats[pl].needs_median) return;
    const double median = MODE == 2 ? stats[pl].median : 0.0;
    const size_t plane = (size_t)pl * ny * nx;
    const T *img = cube + plane;
    bool nm = false;
    const bool live = x < nx;
    const double *in = src + plane + (live ? x : 0);
    double *out = dst + plane + (live ? x : 0);
    constexpr int w = 2 * K + 1;
    constexpr int kLuRegs = (kSolveCols * kSolveBand + kSolveRows - 1) / kSolveRows;
    double prev[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    double v[kSolveCols], g[kSolveCols], glu[kLuRegs];
    auto fetch = [&](int i0) {
        const int cnt = min(kSolveCols, n - i0);
        if (!BACK) {
#pragma unroll
            for (int j = 0; j < kSolveCols; j++) g[j] = (live && j < cnt) ? load_as_f64(img, (size_t)(i0 + j) * nx + x) : 0.0;
        } else {
#pragma unroll
            for (int j = 0; j < kSolveCols; j++) g[j] = (live && j < cnt) ? in[(size_t)(i0 + j) * nx] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < kLuRegs; e++) {
            const int q = e * kSolveRows + lane;
            glu[e] = q < cnt * w ? ax.lu[(size_t)i0 * w + q] : 0.0;
        }
    };
    auto stage = [&](int buf, int cnt, int i0) {
#pragma unroll
        for (int e = 0; e < kLuRegs; e++) {
            const int q = e * kSolveRows + lane;
            if (q < cnt * w) lu[buf][q / w][q % w] = glu[e];
        }
        if (!BACK) {
#pragma unroll
            for (int j = 0; j < kSolveCols; j++) v[j] = (live && j < cnt) ? cleaned_value(img, g[j], (long)(i0 + j), (long)x, ny, nx, median, nm) : 0.0;
        } else {
#pragma unroll
            for (int j = 0; j < kSolveCols; j++) v[j] = g[j];
        }
        wave_sync();
    };
    auto put = [&](int i0, int cnt) {
#pragma unroll
        for (int j = 0; j < kSolveCols; j++)
            if (live && j < cnt) out[(size_t)(i0 + j) * nx] = v[j];
    };
    const int s0 = sg * seg, e0 = min(n, s0 + seg);
    int buf = 0;
    if (!BACK) {
        const int begin = max(0, s0 - warm);
        fetch(begin);
        for (int i0 = begin; i0 < e0; i0 += kSolveCols, buf ^= 1) {
            const int cnt = min(kSolveCols, n - i0);
            stage(buf, cnt, i0);
            if (i0 + kSolveCols < e0) fetch(i0 + kSolveCols);
            solve_fwd_tile<K>(v, prev, lu[buf], cnt);
            if (i0 >= s0) put(i0, cnt);
        }
        if (MODE == 1 && nm) {
            atomicOr(&stats[pl].needs_median, 1);
            stats[n_planes].needs_median = 1;
        }
    } else {
        const int top = seg_back_top(n, e0, warm);
        fetch(top);
        for (int i0 = top; i0 >= s0; i0 -= kSolveCols, buf ^= 1) {
            const int cnt = min(kSolveCols, n - i0);
            stage(buf, cnt, i0);
            if (i0 - kSolveCols >= s0) fetch(i0 - kSolveCols);
            solve_back_tile<K>(v, prev, lu[buf], cnt);
            if (i0 < e0) put(i0, cnt);
        }
    }
}

// Axis 1, one direction of one segment of 64 image rows (the 64 x 16 tile of k_spline_solve_rows), from `src` into `dst`.
template <int K, bool BACK>
__global__ __launch_bounds__(kSolveRows) void k_spline_seg_rows(const double *src, double *dst, int n_planes, int ny, int nx, SplineAxis ax,
                                                                const PlaneStats *only_flagged, int seg, int warm)
{
    __shared__ double tile[kSolveRows][kSolveCols + 1];
    __shared__ double lu[kSolveCols][kSolveBand];
    const int lane = threadIdx.x;
    const int n = ax.n;
    const int groups = (ny + kSolveRows - 1) / kSolveRows, nseg = (n + seg - 1) / seg;
    unsigned b = blockIdx.x;
    const int sg = b % nseg;
    b /= nseg;
    const int pl = b / groups, r0 = (b % groups) * kSolveRows;
    if (only_flagged && !only_flagged[pl].needs_median) return;
    const int rows = min(kSolveRows, ny - r0);
    const size_t first = ((size_t)pl * ny + r0) * nx;
    const double *in = src + first;
    double *out = dst + first;
    constexpr int w = 2 * K + 1;
    const int lr = lane / kSolveCols, lc = lane % kSolveCols;
    constexpr int kLuRegs = (kSolveCols * kSolveBand + kSolveRows - 1) / kSolveRows;
    double prev[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    double v[kSolveCols], g[kSolveRows / kSolveRpi], glu[kLuRegs];
    auto fetch = [&](int c0) {
        const int cols = min(kSolveCols, n - c0);
#pragma unroll
        for (int it = 0; it < kSolveRows / kSolveRpi; it++) {
            const int r = it * kSolveRpi + lr;
            g[it] = (r < rows && lc < cols) ? in[(size_t)r * nx + c0 + lc] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < kLuRegs; e++) {
            const int x = e * kSolveRows + lane;
            glu[e] = x < cols * w ? ax.lu[(size_t)c0 * w + x] : 0.0;
        }
    };
    auto to_lds = [&](int cols) {
#pragma unroll
        for (int it = 0; it < kSolveRows / kSolveRpi; it++) tile[it * kSolveRpi + lr][lc] = g[it];
#pragma unroll
        for (int e = 0; e < kLuRegs; e++) {
            const int x = e * kSolveRows + lane;
            if (x < cols * w) lu[x / w][x % w] = glu[e];
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < kSolveCols; j++) v[j] = tile[lane][j];
    };
    auto tile_out = [&](int c0, int cols, bool keep) {
        if (keep) {
#pragma unroll
            for (int j = 0; j < kSolveCols; j++) tile[lane][j] = v[j];
            wave_sync();
#pragma unroll
            for (int it = 0; it < kSolveRows / kSolveRpi; it++) {
                const int r = it * kSolveRpi + lr;
                if (r < rows && lc < cols) out[(size_t)r * nx + c0 + lc] = tile[r][lc];
            }
        }
        wave_sync();
    };
    const int s0 = sg * seg, e0 = min(n, s0 + seg);
    if (!BACK) {
        const int begin = max(0, s0 - warm);
        fetch(begin);
        for (int c0 = begin; c0 < e0; c0 += kSolveCols) {
            const int cols = min(kSolveCols, n - c0);
            to_lds(cols);
            if (c0 + kSolveCols < e0) fetch(c0 + kSolveCols);
            solve_fwd_tile<K>(v, prev, lu, cols);
            tile_out(c0, cols, c0 >= s0);
        }
    } else {
        const int top = seg_back_top(n, e0, warm);
        fetch(top);
        for (int c0 = top; c0 >= s0; c0 -= kSolveCols) {
            const int cols = min(kSolveCols, n - c0);
            to_lds(cols);
            if (c0 - kSolveCols >= s0) fetch(c0 - kSolveCols);
            solve_back_tile<K>(v, prev, lu, cols);
            tile_out(c0, cols, c0 < e0);
        }
    }
}

__device__ __forceinline__ int spline_interval(const SplineAxis &ax, double x)
{
    // knots are samples (odd k) or sample midpoints (even k): the span follows from floor(x)
    int l = ax.k;
    const int hi = ax.n - 1;
    // t[k+1+j] = j + k/2 + 1 (odd k) or j + k/2 + 0.5 (even k); find the largest l with t[l] <= x
    const double off = (ax.k & 1) ? (double)(ax.k / 2 + 1) : (double)(ax.k / 2) + 0.5;
    int j = (int)floor(x - off) + 1;  // number of interior knots <= x
    if (j < 0) j = 0;
    l = ax.k + j;
    if (l > hi) l = hi;
    while (l < hi && x >= ax.t[l + 1]) l++;  // guard against rounding at knot values
    while (l > ax.k && x < ax.t[l]) l--;
    return l;
}
// The k + 1 B-splines that are non-zero on span l at x (de Boor's recurrence, fitpack fpbspl). Loops of fixed length with
// the degree as a guard - the degree is the same for every lane, the guards are scalar branches - so that h stays in
// registers (indexed by the degree it lived in scratch memory).
__device__ __forceinline__ void spline_basis(const SplineAxis &ax, double x, int l, double (&h)[6])
{
    double hh[6];
    h[0] = 1.0;
#pragma unroll
    for (int j = 1; j <= 5; j++) {
        if (j <= ax.k) {
#pragma unroll
            for (int i = 0; i < j; i++) hh[i] = h[i];
            h[0] = 0.0;
#pragma unroll
            for (int i = 1; i <= j; i++) {
                const int li = l + i, lj = li - j;
                const double tli = ax.t[li], tlj = ax.t[lj];
                const double f = hh[i - 1] / (tli - tlj);
                h[i - 1] += f * (tli - x);
                h[i] = f * (x - tlj);
            }
        }
    }
}
// A lane keeps its cell for `planes_per_lane` planes (blockIdx.y counts groups of planes, pm_smooth_grid): the span and the
// B-spline values of the cell - two searches, 2 x k (k + 1) / 2 divisions, the knots - are the same for every plane, which
// leaves (kr + 1) (kc + 1) coefficients and as many multiply-adds per plane. (Round 5, cubic, 64 planes on a 0.1 deg map:
// 137 us a plane with one plane per lane and the arrays in scratch.)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_spline_eval(const ReprojectArgs a, const SplineArgs sa, int planes_per_lane)
{
    const int m = blockIdx.x * kBlock + threadIdx.x;
    if (m >= a.n_map) return;
    const double nan = __builtin_nan("");
    const int nx = a.nx, ny = a.ny, kr = sa.rows.k, kc = sa.cols.k;
    const double x = a.x_map[m], y = a.y_map[m];
    const bool cell = !isnan(x) && !(a.propagate_nan && (x < 0.0 || y < 0.0 || x > nx - 1 || y > ny - 1));
    // propagate_nan: the (up to) four pixels around the sample
    const long ia = (long)fmax(floor(x), 0.0), ib = (long)fmin(ceil(x), nx - 1.0);
    const long ja = (long)fmax(floor(y), 0.0), jb = (long)fmin(ceil(y), ny - 1.0);
    const long ic = ia < nx - 2 ? ia : nx - 2;
    double hy[6], hx[6];
    size_t corner = 0;
    if (cell) {
        const double xc = fmin(fmax(x, 0.0), nx - 1.0), yc = fmin(fmax(y, 0.0), ny - 1.0);
        const int ly = spline_interval(sa.rows, yc), lx = spline_interval(sa.cols, xc);
        spline_basis(sa.rows, yc, ly, hy);
        spline_basis(sa.cols, xc, lx, hx);
        corner = (size_t)(ly - kr) * nx + (lx - kc);
    }
    for (int g = 0; g < planes_per_lane; g++) {
        const int pl = blockIdx.y * planes_per_lane + g;
        if (pl >= a.n_planes) break;
        const T *img = (const T *)a.cube + (size_t)pl * ny * nx;
        const double *c = sa.work + (size_t)pl * ny * nx + corner;
        double val = nan;
        if (cell) {
            // (the pixels of the NaN test and the coefficients are fetched together - one memory round trip a plane, not two;
            //  every address is inside the plane whatever the test says)
            bool skip = a.plane_stats[pl].all_nan;
            if (a.propagate_nan) {
                // pixels ia and ib of a row in one load: both are ic or ic + 1 (ib is ia or ia + 1)
                double v0, v1, w0, w1;
                load2_as_f64(img, (size_t)ja * nx + ic, v0, v1);
                load2_as_f64(img, (size_t)jb * nx + ic, w0, w1);
                skip = (int)skip | (int)isnan(ia == ic ? v0 : v1) | (int)isnan(ib == ic ? v0 : v1) | (int)isnan(ia == ic ? w0 : w1) |
                       (int)isnan(ib == ic ? w0 : w1);
            }
            double s = 0.0;
#pragma unroll
            for (int p = 0; p <= 5; p++) {
                if (p <= kr) {
                    // (pairs of coefficients per load where the degree has both)
                    double r = 0.0, c0, c1;
#pragma unroll
                    for (int q = 0; q <= 4; q += 2) {
                        if (q + 1 <= kc) {
                            load2_as_f64(c, (size_t)p * nx + q, c0, c1);
                            r += hx[q] * c0;
                            r += hx[q + 1] * c1;
                        } else if (q <= kc) {
                            r += hx[q] * c[(size_t)p * nx + q];
                        }
                    }
                    s += hy[p] * r;
                }
            }
            val = skip ? nan : s;
        }
        a.out[(size_t)pl * a.n_map + m] = val;
    }
}

// ------------------------------------------------------------------ per-plane nanmedian
// np.nanmedian of each plane (+-inf treated as NaN, body_xy.py:1882-1890) by an 8-pass
// radix select over the order-preserving 64-bit key of the doubles. Two ranks are tracked
// at once (the two middle elements of an even count). All planes are processed by the
// same launches; one pass = one streaming read of the cube.
__device__ __forceinline__ unsigned long long sortable_key(double v)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_to_double(unsigned long long k)
{
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_median_hist(const T *cube, size_t plane_elems, int shift, PlaneStats *stats,
                                                        unsigned int *hist /* [P][2][256] */, int lazy)
{
    __shared__ unsigned int h[2][256];
    const int pl = blockIdx.y;
    if (lazy && !stats[pl].needs_median) return;  // (the whole block: no plane of the lazy form that did not ask)
    h[0][threadIdx.x] = 0;
    h[1][threadIdx.x] = 0;
    __syncthreads();
    const T *img = cube + (size_t)pl * plane_elems;
    const unsigned long long mask = (shift == 56) ? 0ull : (~0ull << (shift + 8));
    const unsigned long long pa = stats[pl].prefix[0], pb = stats[pl].prefix[1];
    unsigned int n_nan = 0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < plane_elems; i += (size_t)gridDim.x * kBlock) {
        const double v = (double)img[i];
        if (!isfinite(v)) {
            n_nan += isnan(v) ? 1u : 0u;
            continue;
        }
        const unsigned long long key = sortable_key(v);
        const unsigned int bin = (unsigned int)(key >> shift) & 255u;
        if ((key & mask) == pa) atomicAdd(&h[0][bin], 1u);
        if ((key & mask) == pb) atomicAdd(&h[1][bin], 1u);
    }
    __syncthreads();
    unsigned int *g = hist + (size_t)pl * 512;
    if (h[0][threadIdx.x]) atomicAdd(&g[threadIdx.x], h[0][threadIdx.x]);
    if (h[1][threadIdx.x]) atomicAdd(&g[256 + threadIdx.x], h[1][threadIdx.x]);
    if (shift == 56 && n_nan) atomicAdd(&stats[pl].n_nan, (unsigned long long)n_nan);
}

// one 256-thread block per plane: pick the bin holding each tracked rank, extend the prefix
__global__ __launch_bounds__(kBlock) void k_median_pick(int shift, size_t plane_elems, PlaneStats *stats, unsigned int *hist, int lazy)
{
    const int pl = blockIdx.x;
    if (lazy && !stats[pl].needs_median) return;  // (its statistics stay zero: median 0.0, all_nan 0 - a plane that is not all NaN)
    unsigned int *g = hist + (size_t)pl * 512;
    __shared__ unsigned long long cum[2][256];
    cum[0][threadIdx.x] = g[threadIdx.x];
    cum[1][threadIdx.x] = g[256 + threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 2) {
        const int s = threadIdx.x;
        PlaneStats &st = stats[pl];
        if (shift == 56) {
            unsigned long long n = 0;
            for (int b = 0; b < 256; b++) n += cum[s][b];
            st.n_finite = n;
            st.rank[s] = (n == 0) ? 0 : (s == 0 ? (n - 1) / 2 : n / 2);
        }
        unsigned long long k = st.rank[s], acc = 0;
        int bin = 0;
        for (int b = 0; b < 256; b++) {
            if (acc + cum[s][b] > k) {
                bin = b;
                break;
            }
            acc += cum[s][b];
        }
        st.rank[s] = k - acc;
        st.prefix[s] |= ((unsigned long long)bin) << shift;
    }
    __syncthreads();
    g[threadIdx.x] = 0;
    g[256 + threadIdx.x] = 0;
    if (shift == 0 && threadIdx.x == 0) {
        PlaneStats &st = stats[pl];
        // np.nanmedian: mean of the two middle values; 0.0 if nothing is finite (:1887-1890)
        st.median = st.n_finite ? 0.5 * (key_to_double(st.prefix[0]) + key_to_double(st.prefix[1])) : 0.0;
        st.all_nan = (st.n_nan == (unsigned long long)plane_elems) ? 1 : 0;
    }
}

}  // namespace pm

extern "C++" {

// 'smooth' and the spline evaluation: planes per lane - as many as leave a few thousand workgroups, at most 8 (65 planes on a 0.1 deg map: 104 us a
// plane with 1, 79 with 2, 68 with 4, 61 with 8); `redo` holds 1 + grid.x * grid.y unsigned
dim3 pm_smooth_grid(int n_map, int n_planes)
{
    const long blocks_x = (n_map + pm::kBlock - 1) / pm::kBlock;
    long ppl = blocks_x * n_planes / 4096;
    ppl = ppl < 1 ? 1 : (ppl > 8 ? 8 : ppl);
    return dim3((unsigned)blocks_x, (unsigned)((n_planes + ppl - 1) / ppl));
}

template <typename T>
static void launch_reproject_t(const pm::ReprojectArgs &a, hipStream_t s)
{
    const int chunks = (a.n_map + pm::kBlock - 1) / pm::kBlock;
    const int whole = (a.n_planes / 8) * 8;  // (a group of fewer than 8 planes would leave XCDs idle: those go the plain way)
    pm::ReprojectArgs rest = a;
    if (whole && (unsigned long long)whole * (unsigned)chunks < (1ull << 31)) {
        pm::ReprojectArgs first = a;
        first.n_planes = whole;
        hipLaunchKernelGGL(pm::k_reproject_xcd<T>, dim3((unsigned)whole * (unsigned)chunks), dim3(pm::kBlock), 0, s, first, chunks);
        rest.n_planes = a.n_planes - whole;
        rest.cube = (const T *)a.cube + (size_t)whole * a.ny * a.nx;
        rest.out = a.out + (size_t)whole * a.n_map;
        rest.plane_flags = a.plane_flags ? a.plane_flags + whole : nullptr;
        rest.plane_stats = a.plane_stats ? a.plane_stats + whole : nullptr;
    }
    if (rest.n_planes > 0) {
        dim3 grid((unsigned)chunks, rest.n_planes);
        hipLaunchKernelGGL(pm::k_reproject<T>, grid, dim3(pm::kBlock), 0, s, rest);
    }
}

template <typename T>
static void launch_median_t(const void *cube, int n_planes, size_t plane_elems, pm::PlaneStats *stats,
                            unsigned int *hist, hipStream_t s, int lazy = 0)
{
    unsigned gx = (unsigned)((plane_elems + pm::kBlock * 16 - 1) / (pm::kBlock * 16));
    if (gx < 1) gx = 1;
    if (gx > 256) gx = 256;
    for (int shift = 56; shift >= 0; shift -= 8) {
        hipLaunchKernelGGL(pm::k_median_hist<T>, dim3(gx, n_planes), dim3(pm::kBlock), 0, s, (const T *)cube,
                           plane_elems, shift, stats, hist, lazy);
        hipLaunchKernelGGL(pm::k_median_pick, dim3(n_planes), dim3(pm::kBlock), 0, s, shift, plane_elems, stats, hist, lazy);
    }
}

template <typename T>
static void launch_smooth_t(const pm::ReprojectArgs &a, const pm::SmoothArgs &sm, unsigned *redo, hipStream_t s)
{
    dim3 grid = pm_smooth_grid(a.n_map, a.n_planes);
    pm::SmoothArgs smp = sm;
    smp.planes_per_lane = (a.n_planes + (int)grid.y - 1) / (int)grid.y;
    const size_t total = (size_t)grid.x * grid.y;
    if (!sm.general) {
        (void)hipMemsetAsync(redo, 0, sizeof(unsigned), s);
        hipLaunchKernelGGL(pm::k_reproject_smooth_window<T>, grid, dim3(pm::kBlock), 0, s, a, smp, redo);
    }
    hipLaunchKernelGGL(pm::k_reproject_smooth_gaps<T>, dim3((unsigned)(total < 2048 ? total : 2048)), dim3(pm::kBlock), 0, s, a, smp,
                       sm.general ? nullptr : redo, grid.x, grid.y);
}

void pm_launch_reproject_smooth(const pm::ReprojectArgs &a, const pm::SmoothArgs &sm, int dtype, unsigned *redo, hipStream_t s)
{
    switch (dtype) {
    case PM_F64: launch_smooth_t<double>(a, sm, redo, s); break;
    case PM_F32: launch_smooth_t<float>(a, sm, redo, s); break;
    case PM_I16: launch_smooth_t<int16_t>(a, sm, redo, s); break;
    case PM_I32: launch_smooth_t<int32_t>(a, sm, redo, s); break;
    case PM_U8: launch_smooth_t<uint8_t>(a, sm, redo, s); break;
    case PM_U16: launch_smooth_t<uint16_t>(a, sm, redo, s); break;
    }
}

// `limits`: 4 * (1 + pm::kMapLimitsBlocks) doubles - the result, then the blocks' partial values
void pm_launch_map_limits(const double *x_map, const double *y_map, int n, double *limits, hipStream_t s)
{
    int nb = (n + 8 * pm::kBlock - 1) / (8 * pm::kBlock);
    nb = nb < 1 ? 1 : (nb > pm::kMapLimitsBlocks ? pm::kMapLimitsBlocks : nb);
    hipLaunchKernelGGL(pm::k_map_limits, dim3(nb), dim3(pm::kBlock), 0, s, x_map, y_map, n, limits + 4);
    hipLaunchKernelGGL(pm::k_map_limits_fold, dim3(1), dim3(pm::kBlock), 0, s, limits + 4, nb, limits);
}

template <typename T>
static void launch_mapped_data_t(const pm::Params &p, const pm::ReprojectArgs &a, const double *lon, const double *lat,
                                 double *xo, double *yo, hipStream_t s)
{
    hipLaunchKernelGGL(pm::k_mapped_data<T>, dim3((a.n_map + pm::kSphBlock - 1) / pm::kSphBlock), dim3(pm::kSphBlock), 0, s, p, a,
                       lon, lat, xo, yo);
}

void pm_launch_mapped_data(const pm::Params &p, const pm::ReprojectArgs &a, const double *lon, const double *lat, double *xo,
                           double *yo, int dtype, hipStream_t s)
{
    switch (dtype) {
    case PM_F64: launch_mapped_data_t<double>(p, a, lon, lat, xo, yo, s); break;
    case PM_F32: launch_mapped_data_t<float>(p, a, lon, lat, xo, yo, s); break;
    case PM_I16: launch_mapped_data_t<int16_t>(p, a, lon, lat, xo, yo, s); break;
    case PM_I32: launch_mapped_data_t<int32_t>(p, a, lon, lat, xo, yo, s); break;
    case PM_U8: launch_mapped_data_t<uint8_t>(p, a, lon, lat, xo, yo, s); break;
    case PM_U16: launch_mapped_data_t<uint16_t>(p, a, lon, lat, xo, yo, s); break;
    }
}

void pm_launch_reproject(const pm::ReprojectArgs &a, int dtype, hipStream_t s)
{
    switch (dtype) {
    case PM_F64: launch_reproject_t<double>(a, s); break;
    case PM_F32: launch_reproject_t<float>(a, s); break;
    case PM_I16: launch_reproject_t<int16_t>(a, s); break;
    case PM_I32: launch_reproject_t<int32_t>(a, s); break;
    case PM_U8: launch_reproject_t<uint8_t>(a, s); break;
    case PM_U16: launch_reproject_t<uint16_t>(a, s); break;
    }
}

// the sparse host path: `a.cube` is the device's view of the pinned cube chunk, `t.table` has room
// for a.n_planes x t.n_list blocks
template <typename T>
static void launch_reproject_blocks_t(const pm::ReprojectArgs &a, const pm::BlockTable &t, hipStream_t s)
{
    hipLaunchKernelGGL(pm::k_reproject_blocks<T>, dim3((a.n_map + pm::kBlock - 1) / pm::kBlock, a.n_planes), dim3(pm::kBlock),
                       0, s, a, t);
}

template <typename T>
static void launch_mark_blocks_t(const pm::ReprojectArgs &a, unsigned char *flags, int shift, hipStream_t s)
{
    hipLaunchKernelGGL(pm::k_mark_blocks<T>, dim3((a.n_map + pm::kBlock - 1) / pm::kBlock), dim3(pm::kBlock), 0, s, a, flags, shift);
}

// flags[plane_bytes >> shift] (zeroed by the caller) <- 1 for every block the map samples
void pm_launch_mark_blocks(const pm::ReprojectArgs &a, unsigned char *flags, int shift, int dtype, hipStream_t s)
{
    switch (dtype) {
    case PM_F64: launch_mark_blocks_t<double>(a, flags, shift, s); break;
    case PM_F32: launch_mark_blocks_t<float>(a, flags, shift, s); break;
    case PM_I16: launch_mark_blocks_t<int16_t>(a, flags, shift, s); break;
    case PM_I32: launch_mark_blocks_t<int32_t>(a, flags, shift, s); break;
    case PM_U8: launch_mark_blocks_t<uint8_t>(a, flags, shift, s); break;
    case PM_U16: launch_mark_blocks_t<uint16_t>(a, flags, shift, s); break;
    }
}

// out[0..1] (zeroed) <- 128-bit fingerprint of the two maps
void pm_launch_hash_maps(const double *x_map, const double *y_map, int n, unsigned long long *out, hipStream_t s)
{
    const unsigned blocks = (unsigned)std::min<size_t>(((size_t)n + pm::kBlock - 1) / pm::kBlock, 256);
    hipLaunchKernelGGL(pm::k_hash_maps, dim3(blocks ? blocks : 1), dim3(pm::kBlock), 0, s, x_map, y_map, n, out);
}

// flags (n_pad = whole tiles of 4096, zero beyond n_blk) -> blkmap[n_blk], blklist[*total], *total
void pm_launch_number_blocks(const unsigned char *flags, size_t n_blk, int *tile_sums, int *blkmap, int *blklist, int *total,
                             hipStream_t s)
{
    const unsigned n_tiles = (unsigned)((n_blk + pm::kNumberTile - 1) / pm::kNumberTile);
    hipLaunchKernelGGL(pm::k_blocks_count, dim3(n_tiles), dim3(pm::kBlock), 0, s, flags, tile_sums);
    hipLaunchKernelGGL(pm::k_blocks_offsets, dim3(1), dim3(pm::kBlock), 0, s, tile_sums, (int)n_tiles, total);
    hipLaunchKernelGGL(pm::k_blocks_number, dim3(n_tiles), dim3(pm::kBlock), 0, s, flags, tile_sums, n_blk, blkmap, blklist);
}

void pm_launch_reproject_blocks(const pm::ReprojectArgs &a, const pm::BlockTable &t, int dtype, hipStream_t s, bool fetch)
{
    // (fetch == false: the host has filled the table, a.cube may be null)
    if (fetch)
        hipLaunchKernelGGL(pm::k_fetch_blocks, dim3((((size_t)t.n_list << (t.shift - 4)) + pm::kBlock - 1) / pm::kBlock, a.n_planes), dim3(pm::kBlock),
                           0, s, (const char *)a.cube, t);
    switch (dtype) {
    case PM_F64: launch_reproject_blocks_t<double>(a, t, s); break;
    case PM_F32: launch_reproject_blocks_t<float>(a, t, s); break;
    case PM_I16: launch_reproject_blocks_t<int16_t>(a, t, s); break;
    case PM_I32: launch_reproject_blocks_t<int32_t>(a, t, s); break;
    case PM_U8: launch_reproject_blocks_t<uint8_t>(a, t, s); break;
    case PM_U16: launch_reproject_blocks_t<uint16_t>(a, t, s); break;
    }
}

template <typename T>
static void launch_clean_lazy_t(const pm::ReprojectArgs &a, double *work, pm::PlaneStats *stats, unsigned int *hist, hipStream_t s)
{
    const size_t npx = (size_t)a.ny * a.nx;
    const dim3 grid((unsigned)((npx + pm::kBlock - 1) / pm::kBlock), a.n_planes);
    hipLaunchKernelGGL(pm::k_spline_clean<T>, grid, dim3(pm::kBlock), 0, s, (const T *)a.cube, work, stats, a.ny, a.nx, 1);
    launch_median_t<T>(a.cube, a.n_planes, npx, stats, hist, s, 1);
    const dim3 redo(std::min(grid.x, 256u), a.n_planes);
    hipLaunchKernelGGL(pm::k_spline_clean<T>, redo, dim3(pm::kBlock), 0, s, (const T *)a.cube, work, stats, a.ny, a.nx, 2);
}

template <typename T>
static void launch_spline_t(const pm::ReprojectArgs &a, const pm::SplineArgs &sa, pm::PlaneStats *stats, unsigned int *hist, hipStream_t s,
                            int stage)
{
    const size_t npx = (size_t)a.ny * a.nx;
    const unsigned cgroups = (unsigned)((a.nx + pm::kSolveRows - 1) / pm::kSolveRows) * (unsigned)a.n_planes;
    const unsigned rgroups = (unsigned)((a.ny + pm::kSolveRows - 1) / pm::kSolveRows) * (unsigned)a.n_planes;
    // first round: every plane, cleaned on the way into the axis-0 solve with a provisional 0.0 where a pixel has no finite
    // neighbour (their planes are flagged) ...
    auto cols_pass = [&](auto mode) {
        constexpr int M = decltype(mode)::value;
        auto go = [&](auto kk) {
            hipLaunchKernelGGL((pm::k_spline_solve_cols<T, M, decltype(kk)::value>), dim3(cgroups), dim3(pm::kSolveRows), 0, s, (const T *)a.cube,
                               sa.work, stats, a.n_planes, a.ny, a.nx, sa.rows);
        };
        switch (sa.rows.k) {
        case 1: go(std::integral_constant<int, 1>()); break;
        case 2: go(std::integral_constant<int, 2>()); break;
        case 3: go(std::integral_constant<int, 3>()); break;
        case 4: go(std::integral_constant<int, 4>()); break;
        default: go(std::integral_constant<int, 5>()); break;
        }
    };
    auto rows_pass = [&](const pm::PlaneStats *only_flagged) {
        auto go = [&](auto kk) {
            hipLaunchKernelGGL(pm::k_spline_solve_rows<decltype(kk)::value>, dim3(rgroups), dim3(pm::kSolveRows), 0, s, sa.work, a.n_planes, a.ny,
                               a.nx, sa.cols, only_flagged);
        };
        switch (sa.cols.k) {
        case 1: go(std::integral_constant<int, 1>()); break;
        case 2: go(std::integral_constant<int, 2>()); break;
        case 3: go(std::integral_constant<int, 3>()); break;
        case 4: go(std::integral_constant<int, 4>()); break;
        default: go(std::integral_constant<int, 5>()); break;
        }
    };
    // the segmented forms (sa.seg_rows / seg_cols > 0: few, large planes): forward into work2, backward into work, per axis
    auto with_degree = [&](int k, auto &&go) {
        switch (k) {
        case 1: go(std::integral_constant<int, 1>()); break;
        case 2: go(std::integral_constant<int, 2>()); break;
        case 3: go(std::integral_constant<int, 3>()); break;
        case 4: go(std::integral_constant<int, 4>()); break;
        default: go(std::integral_constant<int, 5>()); break;
        }
    };
    auto cols_seg = [&](auto mode) {
        constexpr int M = decltype(mode)::value;
        const unsigned blocks = cgroups * (unsigned)((a.ny + sa.seg_rows - 1) / sa.seg_rows);
        const int warm = pm::spline_warm(sa.rows.k);
        with_degree(sa.rows.k, [&](auto kk) {
            constexpr int K = decltype(kk)::value;
            hipLaunchKernelGGL((pm::k_spline_seg_cols<T, M, K, false>), dim3(blocks), dim3(pm::kSolveRows), 0, s, (const T *)a.cube,
                               (const double *)nullptr, sa.work2, stats, a.n_planes, a.ny, a.nx, sa.rows, sa.seg_rows, warm);
            hipLaunchKernelGGL((pm::k_spline_seg_cols<double, M, K, true>), dim3(blocks), dim3(pm::kSolveRows), 0, s, (const double *)nullptr,
                               (const double *)sa.work2, sa.work, stats, a.n_planes, a.ny, a.nx, sa.rows, sa.seg_rows, warm);
        });
    };
    auto rows_seg = [&](const pm::PlaneStats *only_flagged) {
        const unsigned blocks = rgroups * (unsigned)((a.nx + sa.seg_cols - 1) / sa.seg_cols);
        const int warm = pm::spline_warm(sa.cols.k);
        with_degree(sa.cols.k, [&](auto kk) {
            constexpr int K = decltype(kk)::value;
            hipLaunchKernelGGL((pm::k_spline_seg_rows<K, false>), dim3(blocks), dim3(pm::kSolveRows), 0, s, (const double *)sa.work, sa.work2,
                               a.n_planes, a.ny, a.nx, sa.cols, only_flagged, sa.seg_cols, warm);
            hipLaunchKernelGGL((pm::k_spline_seg_rows<K, true>), dim3(blocks), dim3(pm::kSolveRows), 0, s, (const double *)sa.work2, sa.work,
                               a.n_planes, a.ny, a.nx, sa.cols, only_flagged, sa.seg_cols, warm);
        });
    };
    const bool segmented = sa.seg_rows > 0;
    if (stage == 0) {
        if (segmented) cols_seg(std::integral_constant<int, 1>());
        else cols_pass(std::integral_constant<int, 1>());
    } else if (stage == 1) {
        if (segmented) rows_seg(nullptr);
        else rows_pass(nullptr);
    } else if (stage == 2) {
        // ... second round: the flagged planes alone, with their nanmedian (the blocks of the others leave at once)
        launch_median_t<T>(a.cube, a.n_planes, npx, stats, hist, s, 1);
        if (segmented) {
            cols_seg(std::integral_constant<int, 2>());
            rows_seg(stats);
        } else {
            cols_pass(std::integral_constant<int, 2>());
            rows_pass(stats);
        }
    } else {
        const dim3 egrid = pm_smooth_grid(a.n_map, a.n_planes);
        hipLaunchKernelGGL(pm::k_spline_eval<T>, egrid, dim3(pm::kBlock), 0, s, a, sa, (a.n_planes + (int)egrid.y - 1) / (int)egrid.y);
    }
}

template <typename T>
static void launch_clean_t(const pm::ReprojectArgs &a, double *work, hipStream_t s)
{
    const size_t npx = (size_t)a.ny * a.nx;
    hipLaunchKernelGGL(pm::k_spline_clean<T>, dim3((unsigned)((npx + pm::kBlock - 1) / pm::kBlock), a.n_planes),
                       dim3(pm::kBlock), 0, s, (const T *)a.cube, work, const_cast<pm::PlaneStats *>(a.plane_stats), a.ny, a.nx, 0);
}
// NaN-cleaned f64 copy of a.n_planes planes into `work`, medians computed only for the planes that need theirs.
// `stats` / `hist` zero-filled by the caller; afterwards `stats` hold all_nan (and the median of the planes that asked).
void pm_launch_clean_lazy(const pm::ReprojectArgs &a, double *work, int dtype, pm::PlaneStats *stats, unsigned int *hist, hipStream_t s)
{
    switch (dtype) {
    case PM_F64: launch_clean_lazy_t<double>(a, work, stats, hist, s); break;
    case PM_F32: launch_clean_lazy_t<float>(a, work, stats, hist, s); break;
    case PM_I16: launch_clean_lazy_t<int16_t>(a, work, stats, hist, s); break;
    case PM_I32: launch_clean_lazy_t<int32_t>(a, work, stats, hist, s); break;
    case PM_U8: launch_clean_lazy_t<uint8_t>(a, work, stats, hist, s); break;
    case PM_U16: launch_clean_lazy_t<uint16_t>(a, work, stats, hist, s); break;
    }
}
// NaN-cleaned f64 copy of a.n_planes planes into `work` (a.plane_stats must hold the medians)
void pm_launch_clean(const pm::ReprojectArgs &a, double *work, int dtype, hipStream_t s)
{
    switch (dtype) {
    case PM_F64: launch_clean_t<double>(a, work, s); break;
    case PM_F32: launch_clean_t<float>(a, work, s); break;
    case PM_I16: launch_clean_t<int16_t>(a, work, s); break;
    case PM_I32: launch_clean_t<int32_t>(a, work, s); break;
    case PM_U8: launch_clean_t<uint8_t>(a, work, s); break;
    case PM_U16: launch_clean_t<uint16_t>(a, work, s); break;
    }
}

// `stats` (= a.plane_stats, n_planes + 1 elements) and `hist` zero-filled by the caller: the clean pass fills what it needs of
// them (lazy form). Stages, in this order: 0 the axis-0 solve of every plane (flags the planes that need their nanmedian, and
// stats[n_planes].needs_median if any does), 1 the axis-1 solve, 2 the second round for the flagged planes (medians, both
// solves: the caller may leave it out when nothing was flagged - most data), 3 the evaluation at the map cells.
void pm_launch_spline(const pm::ReprojectArgs &a, const pm::SplineArgs &sa, int dtype, pm::PlaneStats *stats, unsigned int *hist,
                      hipStream_t s, int stage)
{
    switch (dtype) {
    case PM_F64: launch_spline_t<double>(a, sa, stats, hist, s, stage); break;
    case PM_F32: launch_spline_t<float>(a, sa, stats, hist, s, stage); break;
    case PM_I16: launch_spline_t<int16_t>(a, sa, stats, hist, s, stage); break;
    case PM_I32: launch_spline_t<int32_t>(a, sa, stats, hist, s, stage); break;
    case PM_U8: launch_spline_t<uint8_t>(a, sa, stats, hist, s, stage); break;
    case PM_U16: launch_spline_t<uint16_t>(a, sa, stats, hist, s, stage); break;
    }
}

// stats / hist must be zero-filled by the caller (hipMemsetAsync) before this call
void pm_launch_plane_medians(const void *cube, int dtype, int n_planes, size_t plane_elems, pm::PlaneStats *stats,
                             unsigned int *hist, hipStream_t s)
{
    switch (dtype) {
    case PM_F64: launch_median_t<double>(cube, n_planes, plane_elems, stats, hist, s); break;
    case PM_F32: launch_median_t<float>(cube, n_planes, plane_elems, stats, hist, s); break;
    case PM_I16: launch_median_t<int16_t>(cube, n_planes, plane_elems, stats, hist, s); break;
    case PM_I32: launch_median_t<int32_t>(cube, n_planes, plane_elems, stats, hist, s); break;
    case PM_U8: launch_median_t<uint8_t>(cube, n_planes, plane_elems, stats, hist, s); break;
    case PM_U16: launch_median_t<uint16_t>(cube, n_planes, plane_elems, stats, hist, s); break;
    }
}
}
