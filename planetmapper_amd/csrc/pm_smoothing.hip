// pm_smoothing.hip -- smoothing splines of BodyXY.map_img (`spline_smoothing > 0`, body_xy.py:1673-1680): FITPACK's
// `regrid` (Dierckx) for every plane of a cube AT ONCE, with the whole of its control flow on the device.
//
// regrid grows the knot sets of a plane from the least-squares polynomial until the least-squares spline has a residual
// sum fp <= s, then finds the smoothing parameter p with fp(p) = s by rational interpolation: a few dozen FITS, each of
// which decides what the next one is. A fit is two directional least-squares solves (all image columns, then all
// coefficient rows) and a residual sum over the pixels; the decisions between two fits are scalar. Fitted plane by plane
// from the host (rounds 2-4) that is a chain of small launches and a read-back per fit: 71 ms per 1024^2 plane on 16 waves
// of a chip that has 16 000 slots. Here
//   * the state of every plane's search (knots, residual sums per knot interval, the bracket of p) lives in HBM
//     (SmPlaneDev); k_smb_decide - one wave per plane - consumes the last fit and sets up the next: FITPACK's fpknot adds
//     knots wave-cooperatively, fprati picks p;
//   * k_smb_tables / k_smb_factor rebuild what the new knots / p need: B-spline values of the samples, the jump rows
//     B / p (fpdisc), and the triangular band factor R of [A; B / p] by Givens rotations (fpgivs / fprota) - one wave per
//     (plane, axis), the rows of both blocks merged by first column so that a row meets only k + 2 rows of R: the lanes
//     of the wave are the stages of a pipeline the rows stream through (R is stored once and never loaded);
//   * the fit kernels take blockIdx.y / .z = plane: every plane's fit of a round runs in the same launches, each plane
//     with its own knots. A directional solve is the corrected semi-normal equations R'R c = A'd + one refinement step
//     with the residual (error ~ cond(A) eps), fused per pass into ONE sweep per right-hand side: the lane walks the
//     samples once, a register window holds the k + 1 partial sums of A'd that are still open, each sum that closes is
//     fed straight into the forward substitution; the back substitution follows (the second pass takes its residuals from k_smb_res,
//     jump rows merged among the others). No atomics: a plane's result is the same bits whatever shares its launches (tests/test_gpu_splines_cube_scale.py);
//   * the host only sizes the grids: one 16-byte read-back per round (planes still searching, the largest coefficient
//     counts) - every wave's loops are bounded by that round's own tables, the round loop by FITPACK's own iteration caps.
// Planes whose search has ended stop taking part (their blocks leave at once); the cube is fitted in batches of as many
// planes as the workspace budget allows (6 work arrays of a plane each + ~0.4 MB of tables).
#include <type_traits>

#include "pm_host.hip.h"

namespace pm {

constexpr int kSmRow = 8;  // doubles per row of a band factor / jump matrix: entries 0..6 (degree <= 5: band = k + 2 <= 7), 1 / diagonal
constexpr int kSmTile = 64;

struct SmAxisDev {
    double *hb;      // m x 6: the k + 1 non-zero B-splines at sample i
    int *lb;         // m: coefficient index of the first of them (= knot interval - k)
    int *span;       // m: knot interval of the integer abscissa i
    double *R;       // nc x kSmRow: band factor of [A; B / p]
    double *Bp;      // nb x kSmRow: the jump rows B / p, row r covers coefficients r .. r + k + 1
    double *t;       // n knots (capacity m + k + 2)
    double *fpint;   // residual sum per knot interval (capacity m + 1)
    int *nrdata;     // data points strictly inside each interval (capacity m + 1)
    int *twin;       // != 0 and equal for the two halves of an interval fpknot split down the middle (their shares are the
                     // same product: equal in any arithmetic); 0 for sums that come from a fit (capacity m + 1)
    // the rows of [A; B / p] merged in the order of their first column (nb > 0): what the refinement pass walks
    int *mg_src;     // m + nb: sample i, or -(r + 1) for jump row r
    int *mg_l;       // m + nb: first column of the row
    double *mg_h;    // (m + nb) x kSmRow: its k + 2 entries
    int m, k, n, nplus, nb, knots_changed;
    __host__ __device__ int nc() const { return n - k - 1; }
    __host__ __device__ int nrint() const { return n - 2 * k - 1; }
};
struct SmPlaneDev {
    SmAxisDev y, x;   // FITPACK's "x" = the first array axis = image rows (y here), its "y" = image columns
    const double *z;  // the cleaned plane (ny x nx)
    double *U, *UT, *G, *CT;
    double *RB;       // residuals of the rows of a direction, (m + nb) x right-hand sides
    double *rowpart, *colpart;  // [tiles_x][ny], [tiles_y][nx]: residual sums per tile column / tile row
    double *rowsum, *colsum;
    double s, acc;
    double fp, fp0, fpold, reducy, reducx, fpms, p, p1, f1, p3, f3;
    int phase;  // 0 knot search, 1 smoothing parameter, 2 finished
    int iter, it2, lastdi, poly, ich1, ich3, fits;
    int active;   // a fit is wanted this round
    int all_nan;  // body_xy.py:1668-1670: the map of an all-NaN image is all NaN
    int plane;    // index in the chunk (cube, output, statistics)
    int knife_edge;  // the knot search took a decision between intervals whose residual shares tie to rounding (sm_add_knot)
    // the refinement pass of the latest fit, per direction: largest |correction| and largest |coefficient| (bit patterns of
    // non-negative doubles order like integers: atomicMax), and what k_smb_decide made of them
    unsigned long long ref_delta[2], ref_scale[2];
    double refine_ratio;   // max over the directions of correction / scale, latest fit
    int ill_conditioned;   // a fit of the search whose refinement moved its coefficients by more than kSmIllRatio of their scale
};
// A corrected-semi-normal-equations solve squares the condition number of [A; B / p] where FITPACK's Givens QR carries it once:
// its refinement step converges while cond^2 eps < 1 and then lands on the least-squares solution to ~cond eps. A fit whose
// refinement step still moves the coefficients by this much of their scale is beyond that - and the fits after it hang on
// its residual sum (knot placement, the bracket of p). Measured (round-6 soak, 191 000 plane fits, 3 beyond the 1e-7 bar, all
// on a 20-25 sample axis of degree 4-5): healthy fits 1e-15 .. 2e-12; smoothing parameters p ~ 1e7 .. 1e9 where the knot
// set leaves combinations of coefficients to the jump rows B / p alone 4.5e-10 and 1.3e-9 (4e-6 and 9e-5 of scale from
// scipy); a least-squares phase with a rank-deficient knot set 0.58 (its residual sums steer the search of p to another
// acceptable p: 5e-5). Reported: PM_OPT_LAST_SM_ILL_CONDITIONED.
constexpr double kSmIllRatio = 1e-10;

// dst[x] = value(x), x < n, by the NT threads of the workgroup: eight requests in flight per thread (a plain loop of
// load - store pairs waits out every trip to HBM on its own)
template <int NT, typename T, typename F>
__device__ __forceinline__ void sm_stage(T *dst, int n, F &&value)
{
    for (int x0 = threadIdx.x; x0 < n; x0 += NT * 8) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = x0 + u * NT < n ? value(x0 + u * NT) : T();
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (x0 + u * NT < n) dst[x0 + u * NT] = v[u];
    }
}

// ------------------------------------------------------------------ the search between two fits
__device__ __forceinline__ double wave_sum(const double *v, int n, int lane)
{
    double s = 0.0;
    for (int i = lane; i < n; i += 64) s += v[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);  // (commutative per step: every lane ends with the same bits)
    return s;
}

// a[lo .. hi] -> a[lo + 1 .. hi + 1], one wave, from the top in blocks of 64
template <typename T>
__device__ __forceinline__ void shift_up(T *a, int lo, int hi, int lane)
{
    for (int top = hi; top >= lo; top -= 64) {
        const int idx = top - lane;
        T v = T();
        if (idx >= lo) v = a[idx];
        __syncthreads();
        if (idx >= lo) a[idx + 1] = v;
    }
    __syncthreads();
}

// fpknot: a new knot at the middle data point of the interval with the largest residual sum (one wave)
// `remaining`: the knots this round may still add, this one included. *knife_edge is set when the choice made here is one
// the reference makes by rounding noise: fpknot's shares of a split interval (fpint = fpmax * an / am) are EQUAL in exact
// arithmetic for several intervals at once (F/2 * 4/10 and F * 9/20 * 4/9, say) and differ in their last bits by the
// rounding of those products - i.e. by the last bit of F, a sum of ~10^3 squared residuals that no implementation with
// other coefficient arithmetic than FITPACK's serial Givens sequence has to the bit (a one-ulp change of ONE input pixel
// flips scipy's own choice: tests/test_smoothing_knife_edge.py). Counted: intervals within 4 ulp of the largest sum that
// could take a knot, other than the twin of an interval split down the middle (SmAxisDev::twin - the same product on both
// sides: equal in FITPACK too, the first wins there as here); the edge is real when the round cannot give every one of
// them a knot.
__device__ bool sm_add_knot(SmAxisDev &a, int &n, int lane, int remaining, int *knife_edge)
{
    const int k = a.k, nri = n - 2 * k - 1;
    double best = 0.0;
    int number = 0x7fffffff;
    for (int j = lane; j < nri; j += 64) {
        const double v = a.fpint[j];
        if (a.nrdata[j] != 0 && v > best) { best = v; number = j; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(number, off, 64);
        if (ob > best || (ob == best && oi < number)) { best = ob; number = oi; }
    }
    if (number == 0x7fffffff) return false;
    {
        const double near = best * (1.0 - 0x1p-50);
        int others = 0;
        for (int j = lane; j < nri; j += 64) {
            const double v = a.fpint[j];
            const int tw = a.twin[j];
            const bool twin = tw != 0 && tw == a.twin[number];
            others += (j != number && a.nrdata[j] != 0 && v >= near && !twin) ? 1 : 0;
        }
        for (int off = 32; off > 0; off >>= 1) others += __shfl_xor(others, off, 64);
        if (others >= remaining) *knife_edge = 1;
    }
    const int maxpt = a.nrdata[number];
    // the new knot: ihalf data points beyond the interval's left knot (FITPACK counts the points before it, maxbeg - knots
    // are sample abscissae here, so that count is the left knot itself)
    const int ihalf = maxpt / 2 + 1, at = (int)a.t[number + k] + ihalf;
    __syncthreads();
    // one step up for everything behind the interval: the k + 1 topmost knots first (they have no interval of their own),
    // then fpint / nrdata [j] together with t[j + k], from the top in blocks of 64
    {
        const int idx = n - 1 - lane;
        double v = 0.0;
        if (lane <= k && idx >= number + k + 1) v = a.t[idx];
        __syncthreads();
        if (lane <= k && idx >= number + k + 1) a.t[idx + 1] = v;
    }
    for (int top = nri - 1; top >= number + 1; top -= 64) {
        const int j = top - lane;
        const bool mine = j >= number + 1;
        double vf = 0.0, vt = 0.0;
        int vn = 0, vw = 0;
        if (mine) {
            vf = a.fpint[j];
            vn = a.nrdata[j];
            vw = a.twin[j];
            vt = a.t[j + k];
        }
        __syncthreads();
        if (mine) {
            a.fpint[j + 1] = vf;
            a.nrdata[j + 1] = vn;
            a.twin[j + 1] = vw;
            a.t[j + k + 1] = vt;
        }
    }
    __syncthreads();
    if (lane == 0) {
        a.nrdata[number] = ihalf - 1;
        a.nrdata[number + 1] = maxpt - ihalf;
        a.fpint[number] = best * (double)(ihalf - 1) / (double)maxpt;
        a.fpint[number + 1] = best * (double)(maxpt - ihalf) / (double)maxpt;
        a.twin[number] = a.twin[number + 1] = (ihalf - 1 == maxpt - ihalf) ? n : 0;  // (n, the knot count before this one: unique per split)
        a.t[number + k + 1] = (double)at;
    }
    n += 1;
    __syncthreads();
    return true;
}

// per-interval residual sums from per-sample sums (a sample on a knot gives half to each side): one lane per interval,
// in the order of FITPACK's sequential loop
__device__ void sm_account(const SmAxisDev &a, int n, const double *sums, int lane)
{
    const int k = a.k, nri = n - 2 * k - 1;
    for (int num = lane; num < nri; num += 64) {
        const int lo = (int)a.t[num + k];
        const int hi = (num == nri - 1) ? a.m - 1 : (int)a.t[num + k + 1] - 1;
        double acc = (num > 0) ? 0.5 * sums[lo] : sums[lo];
        for (int i = lo + 1; i <= hi; i++) acc += sums[i];
        if (num < nri - 1) acc += 0.5 * sums[hi + 1];
        a.fpint[num] = acc;
        a.twin[num] = 0;
    }
}

// up to `count` new knots along one axis, its knot arrays worked on in LDS (`lds`: they fit): a new knot is a scan, two
// reductions and three shifts over them, each a round trip - to LDS a tenth of a microsecond, to HBM one or two
__device__ bool sm_refine(SmAxisDev &g, int &n, int count, int nmax, int lane, double *smem, int lds, int *knife_edge)
{
    SmAxisDev a = g;
    const int k = a.k, n0 = n;
    if (lds) {
        double *tS = smem, *fS = tS + (a.m + k + 2);
        int *nS = (int *)(fS + (a.m + 1)), *wS = nS + (a.m + 1);
        for (int i = lane; i < n0; i += 64) tS[i] = g.t[i];
        for (int i = lane; i < n0 - 2 * k - 1; i += 64) { fS[i] = g.fpint[i]; nS[i] = g.nrdata[i]; wS[i] = g.twin[i]; }
        a.t = tS; a.fpint = fS; a.nrdata = nS; a.twin = wS;
        __syncthreads();
    }
    bool added = false;
    for (int l = 0; l < count; l++) {  // (FITPACK tries `count` times whether or not an interval can still take a knot; once none can, none will)
        if (!sm_add_knot(a, n, lane, min(count - l, nmax - n), knife_edge)) break;
        added = true;
        if (n == nmax) break;
    }
    if (lds && added) {
        for (int i = lane; i < n; i += 64) g.t[i] = a.t[i];
        for (int i = lane; i < n - 2 * k - 1; i += 64) { g.fpint[i] = a.fpint[i]; g.nrdata[i] = a.nrdata[i]; g.twin[i] = a.twin[i]; }
    }
    __syncthreads();
    return added;
}

// One wave per plane: close the fit of the last round (residual sums -> fp, per-interval sums), take FITPACK's decision
// (fpregr: more knots / the next p / finished) and publish what the next fit is. status: [0] planes with a fit to run,
// [1] / [2] the largest coefficient counts along y / x among them, [3] whether any of them fits with p > 0.
__host__ __device__ inline size_t sm_decide_lds_bytes(int m, int k) { return ((size_t)(m + k + 2) + (size_t)(m + 1)) * sizeof(double) + 2 * (size_t)(m + 1) * sizeof(int); }
__global__ __launch_bounds__(64) void k_smb_decide(SmPlaneDev *planes, const PlaneStats *stats, int *status, int first, int tiles_x,
                                                   int tiles_y, int lds)
{
    extern __shared__ double sm_knots[];
    SmPlaneDev &P = planes[blockIdx.x];
    const int lane = threadIdx.x;
    const double con1 = 0.1, con9 = 0.9, con4 = 0.04;
    const int maxit = 20;
    int phase = P.phase;
    if (!first && phase == 2) return;
    const int my = P.y.m, mx = P.x.m, ky = P.y.k, kx = P.x.k;
    const int nminy = 2 * (ky + 1), nminx = 2 * (kx + 1), nmaxy = my + ky + 1, nmaxx = mx + kx + 1;
    int ny_n = P.y.n, nx_n = P.x.n;
    bool changed_y = false, changed_x = false;
    double fp = P.fp, fp0 = P.fp0, fpold = P.fpold, reducy = P.reducy, reducx = P.reducx, fpms = P.fpms;
    double p = P.p, p1 = P.p1, f1 = P.f1, p3 = P.p3, f3 = P.f3;
    int iter = P.iter, it2 = P.it2, lastdi = P.lastdi, poly = P.poly, ich1 = P.ich1, ich3 = P.ich3;
    int nplus_y = P.y.nplus, nplus_x = P.x.nplus;
    const double s = P.s, acc = P.acc;
    int knife_edge = 0;
    if (first) {
        const int all_nan = stats[P.plane].all_nan;
        if (lane == 0) P.all_nan = all_nan;
        phase = all_nan ? 2 : 0;
        ny_n = nminy;
        nx_n = nminx;
        for (int i = lane; i <= ky; i += 64) { P.y.t[i] = 0.0; P.y.t[ky + 1 + i] = (double)(my - 1); }
        for (int i = lane; i <= kx; i += 64) { P.x.t[i] = 0.0; P.x.t[kx + 1 + i] = (double)(mx - 1); }
        if (lane == 0) {
            P.y.nrdata[0] = my - 2;
            P.x.nrdata[0] = mx - 2;
            P.y.fpint[0] = 0.0;
            P.x.fpint[0] = 0.0;
            P.y.twin[0] = 0;
            P.x.twin[0] = 0;
        }
        changed_y = changed_x = true;
        fp = fp0 = fpold = reducy = reducx = fpms = 0.0;
        p = -1.0;
        p1 = f1 = f3 = 0.0;
        p3 = -1.0;
        iter = it2 = lastdi = ich1 = ich3 = 0;
        nplus_y = nplus_x = 0;
        poly = 1;
        if (lane == 0) {
            P.fits = 0;
            P.knife_edge = 0;
            P.ill_conditioned = 0;
            P.refine_ratio = 0.0;
            P.ref_delta[0] = P.ref_delta[1] = P.ref_scale[0] = P.ref_scale[1] = 0ull;
        }
    } else {
        // the residual sums of the fit, tile partials added in a fixed order
        for (int i = lane; i < my; i += 64) {
            double v = 0.0;
            for (int tx = 0; tx < tiles_x; tx++) v += P.rowpart[(size_t)tx * my + i];
            P.rowsum[i] = v;
        }
        for (int j = lane; j < mx; j += 64) {
            double v = 0.0;
            for (int ty = 0; ty < tiles_y; ty++) v += P.colpart[(size_t)ty * mx + j];
            P.colsum[j] = v;
        }
        __syncthreads();
        {
            // how far the refinement passes of the fit just made moved its coefficients, relative to their scale
            double ratio = 0.0;
            for (int dir = 0; dir < 2; dir++) {
                const double dl = __longlong_as_double((long long)P.ref_delta[dir]), sc = __longlong_as_double((long long)P.ref_scale[dir]);
                if (sc > 0.0) ratio = fmax(ratio, dl / sc);
            }
            __syncthreads();
            if (lane == 0) {
                P.refine_ratio = ratio;
                P.ref_delta[0] = P.ref_delta[1] = P.ref_scale[0] = P.ref_scale[1] = 0ull;
            }
        }
        fp = wave_sum(P.rowsum, my, lane);
        sm_account(P.y, ny_n, P.rowsum, lane);
        sm_account(P.x, nx_n, P.colsum, lane);
        __syncthreads();
        bool finished = false, to_phase1 = false;
        if (phase == 0) {
            if (poly) fp0 = fp;
            fpms = fp - s;
            if (fabs(fpms) < acc) {
                finished = true;
            } else if (fpms < 0.0) {
                if (poly) finished = true; else to_phase1 = true;
            } else if (ny_n == nmaxy && nx_n == nmaxx) {
                finished = true;  // the interpolating spline
            } else {
                if (lastdi < 0) reducy = fpold - fp;
                else if (lastdi > 0) reducx = fpold - fp;
                fpold = fp;
                auto nplus = [&](int n, int nmin, int npl_old, double reduc) {
                    if (n == nmin) return 1;
                    int npl1 = npl_old * 2;
                    if (reduc > acc) npl1 = (int)((double)npl_old * fpms / reduc);
                    return min(npl_old * 2, max(max(npl1, npl_old / 2), 1));
                };
                const int nply = nplus(ny_n, nminy, nplus_y, reducy), nplx = nplus(nx_n, nminx, nplus_x, reducx);
                bool first_axis = (nply < nplx) || (nply == nplx && lastdi >= 0);
                if (first_axis && ny_n == nmaxy) first_axis = false;
                if (!first_axis && nx_n == nmaxx) first_axis = true;
                lastdi = first_axis ? -1 : 1;
                // (FITPACK tries nplus times whether or not an interval can still take a knot; once none can, none will)
                bool added;
                if (first_axis) {
                    nplus_y = nply;
                    added = sm_refine(P.y, ny_n, nply, nmaxy, lane, sm_knots, lds, &knife_edge);
                    changed_y = added;
                } else {
                    nplus_x = nplx;
                    added = sm_refine(P.x, nx_n, nplx, nmaxx, lane, sm_knots, lds, &knife_edge);
                    changed_x = added;
                }
                iter++;
                // no interval with a positive residual sum and room for a knot (residual sums that are not numbers, in the
                // end): the next fit would be this one again, ny + nx times over - the search ends as if they had been run
                if (!added) iter = my + mx;
                if (iter >= my + mx) {
                    if (poly) finished = true; else to_phase1 = true;
                } else {
                    poly = (ny_n == nminy && nx_n == nminx) ? 1 : 0;
                }
            }
            if (to_phase1) {
                phase = 1;
                p1 = 0.0; f1 = fp0 - s; p3 = -1.0; f3 = fpms; p = 1.0;
                ich1 = ich3 = 0;
                it2 = 0;
            }
        } else {  // the smoothing parameter: F(p) = fp(p) - s = 0 by rational interpolation (fprati)
            fpms = fp - s;
            if (fabs(fpms) < acc || it2 == maxit - 1) {
                finished = true;
            } else {
                const double p2 = p, f2 = fpms;
                bool again = false;
                if (!ich3) {
                    if ((f2 - f3) <= acc) {  // initial p too large
                        p3 = p2; f3 = f2;
                        p *= con4;
                        if (p <= p1) p = p1 * con9 + p2 * con1;
                        again = true;
                    } else if (f2 < 0.0) {
                        ich3 = 1;
                    }
                }
                if (!again && !ich1) {
                    if ((f1 - f2) <= acc) {  // initial p too small
                        p1 = p2; f1 = f2;
                        p /= con4;
                        if (p3 >= 0.0 && p >= p3) p = p2 * con1 + p3 * con9;
                        again = true;
                    } else if (f2 > 0.0) {
                        ich1 = 1;
                    }
                }
                if (!again) {
                    if (f2 >= f1 || f2 <= f3) {
                        finished = true;
                    } else {
                        if (p3 > 0.0) {
                            const double h1 = f1 * (f2 - f3), h2 = f2 * (f3 - f1), h3 = f3 * (f1 - f2);
                            p = -(p1 * p2 * h3 + p2 * p3 * h1 + p3 * p1 * h2) / (p1 * h1 + p2 * h2 + p3 * h3);
                        } else {
                            p = (p1 * (f1 - f3) * f2 - p2 * (f2 - f3) * f1) / ((f1 - f2) * f3);
                        }
                        if (f2 < 0.0) { p3 = p2; f3 = f2; } else { p1 = p2; f1 = f2; }
                    }
                }
                if (!finished) it2++;
            }
        }
        if (finished) phase = 2;
    }
    const double p_fit = (phase == 1) ? p : -1.0;
    const int nri_y = ny_n - 2 * ky - 1, nri_x = nx_n - 2 * kx - 1;
    if (lane == 0) {
        if (knife_edge && !P.knife_edge) {
            P.knife_edge = 1;
            atomicAdd(&status[8], 1);  // (status[8]: zeroed once per call, not per round)
        }
        if (!first && !P.ill_conditioned && P.refine_ratio > kSmIllRatio) {
            P.ill_conditioned = 1;  // (sticky: every later decision of the search hangs on this fit's residual sum)
            atomicAdd(&status[9], 1);
        }
        P.phase = phase;
        P.active = phase != 2;
        P.y.n = ny_n; P.x.n = nx_n;
        P.y.nplus = nplus_y; P.x.nplus = nplus_x;
        P.y.knots_changed = changed_y; P.x.knots_changed = changed_x;
        P.y.nb = (p_fit > 0.0 && nri_y > 1) ? nri_y - 1 : 0;
        P.x.nb = (p_fit > 0.0 && nri_x > 1) ? nri_x - 1 : 0;
        P.fp = fp; P.fp0 = fp0; P.fpold = fpold; P.reducy = reducy; P.reducx = reducx; P.fpms = fpms;
        P.p = (phase == 1) ? p : -1.0; P.p1 = p1; P.f1 = f1; P.p3 = p3; P.f3 = f3;
        P.iter = iter; P.it2 = it2; P.lastdi = lastdi; P.poly = poly; P.ich1 = ich1; P.ich3 = ich3;
        if (phase != 2) {
            P.fits++;
            atomicAdd(&status[0], 1);
            atomicMax(&status[1], ny_n - ky - 1);
            atomicMax(&status[2], nx_n - kx - 1);
            if (p_fit > 0.0) atomicMax(&status[3], 1);
        }
    }
}

// ------------------------------------------------------------------ tables of a fit
__device__ __forceinline__ void bspl(const double *t, int k, double x, int l, double *h)
{
    double hh[6];
    h[0] = 1.0;
    for (int j = 1; j <= k; j++) {  // fpbspl
        for (int q = 0; q < j; q++) hh[q] = h[q];
        h[0] = 0.0;
        for (int q = 1; q <= j; q++) {
            const int li = l + q, lj = li - j;
            const double f = hh[q - 1] / (t[li] - t[lj]);
            h[q - 1] += f * (t[li] - x);
            h[q] = f * (x - t[lj]);
        }
    }
}

// grid (2 axes, planes): B-spline values + knot interval of every sample (when the knots changed), the jump rows B / p
// (fpdisc) when p > 0
__global__ __launch_bounds__(kBlock) void k_smb_tables(SmPlaneDev *planes)
{
    SmPlaneDev &P = planes[blockIdx.y];
    if (!P.active) return;
    SmAxisDev &a = blockIdx.x ? P.x : P.y;
    const int m = a.m, k = a.k, n = a.n;
    if (a.knots_changed)
        for (int i = threadIdx.x; i < m; i += kBlock) {
            int lo = k, hi = n - k - 2;  // the largest l <= n - k - 2 with t[l] <= i
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (a.t[mid] <= (double)i) lo = mid; else hi = mid - 1;
            }
            double h[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            bspl(a.t, k, (double)i, lo, h);
            for (int e = 0; e < 6; e++) a.hb[(size_t)i * 6 + e] = h[e];
            a.lb[i] = lo - k;
            a.span[i] = lo;
        }
    const int nb = a.nb;
    if (nb > 0) {
        const int band = k + 2, nri = n - 2 * k - 1;
        const double fac = (double)nri / (a.t[n - k - 1] - a.t[k]), p = P.p;
        for (int r = threadIdx.x; r < nb; r += kBlock) {
            const int l = r + k + 1;
            for (int j = 0; j < kSmRow; j++) {
                double v = 0.0;
                if (j < band) {
                    const int i = r + j;
                    double prod = 1.0;
                    bool first = true;
                    for (int q = 0; q < band; q++) {
                        if (i + q == l) continue;
                        const double h = a.t[l] - a.t[i + q];
                        prod = first ? h : prod * h * fac;
                        first = false;
                    }
                    v = (a.t[i + k + 1] - a.t[i]) / prod / p;
                }
                a.Bp[(size_t)r * kSmRow + j] = v;
            }
        }
        // the merged row stream: jump row r after the samples of the knot intervals <= r (this thread wrote the Bp rows
        // and, if the knots changed, the lb / hb entries it reads here)
        for (int i = threadIdx.x; i < m; i += kBlock) {
            const int l0 = a.lb[i], pos = i + min(l0, nb);
            a.mg_src[pos] = i;
            a.mg_l[pos] = l0;
            for (int e = 0; e < kSmRow; e++) a.mg_h[(size_t)pos * kSmRow + e] = e <= k ? a.hb[(size_t)i * 6 + e] : 0.0;
        }
        for (int r = threadIdx.x; r < nb; r += kBlock) {
            const int pos = (int)a.t[k + r + 1] + r;
            a.mg_src[pos] = -(r + 1);
            a.mg_l[pos] = r;
            for (int e = 0; e < kSmRow; e++) a.mg_h[(size_t)pos * kSmRow + e] = a.Bp[(size_t)r * kSmRow + e];
        }
    }
}

// Triangular band R of the QR factor of [A; B / p] by Givens rotations (fpgivs / fprota). FITPACK rotates the collocation
// rows in first and then each jump row from its first column TO THE END of the band (its fill-in travels: O(nb x nc)
// rotations). R is unique (positive diagonal), so here the rows of both blocks are taken in the order of their first column
// instead (the merged stream of k_smb_tables): a row that enters at coefficient j0 then meets only rows of R that hold
// nothing beyond column j0 + BAND - 1, and is used up after BAND rotations with the rows j0 .. j0 + BAND - 1 of R.
// That is a pipeline, and the wave runs it as one: LANE q holds row win0 + q of R in registers and performs the q-th
// rotation of every row of the stream; a row moves one lane on per tick (DPP), rotated rows follow one another by one tick -
// by two where the second enters one coefficient further on: then the rows of R move one lane down in its wake, lane 0's
// row is final and is stored (with the reciprocal of its diagonal, what the substitutions multiply by). Each row of R sees
// the stream's rows in their order and each stream row sees its BAND rows of R in theirs: the same operations on the same
// operands as one lane working through the stream alone - in (rows + advances) ticks instead of rows x BAND rotations.
// Nothing of R is ever loaded.
__device__ __forceinline__ void sm_givens(double piv, double ww, double &cs, double &sn, double &dd)
{
    // (cs, sn, dd) of the rotation that annihilates piv against the diagonal ww: one reciprocal square root for operands of
    // ordinary size, FITPACK's scaled form (fpgivs) otherwise
    const double x2 = fma(piv, piv, ww * ww);
    if (x2 > 1e-280 && x2 < 1e280) {
        const double r = rsqrt_fast(x2);
        dd = x2 * r;
        cs = ww * r;
        sn = piv * r;
    } else {
        const double store = fabs(piv);
        dd = (store >= ww) ? store * sqrt(1.0 + (ww / piv) * (ww / piv)) : ww * sqrt(1.0 + (piv / ww) * (piv / ww));
        cs = ww / dd;
        sn = piv / dd;
    }
}
// lane l <- lane l - 1 / lane l + 1 of the same row of 16 lanes (0 from beyond it)
__device__ __forceinline__ int dpp_prev(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true); }
__device__ __forceinline__ int dpp_next(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x101, 0xf, 0xf, true); }
__device__ __forceinline__ double dpp_prev(double v) { return __hiloint2double(dpp_prev(__double2hiint(v)), dpp_prev(__double2loint(v))); }
__device__ __forceinline__ double dpp_next(double v) { return __hiloint2double(dpp_next(__double2hiint(v)), dpp_next(__double2loint(v))); }

template <int BAND>
__device__ void sm_factor_stream(const double *rowh, int hs, int we, const int *rowl, int n_rows, double *__restrict__ R, int ncf)
{
    const int lane = threadIdx.x;
    const bool stage = lane < BAND;
    double w[BAND], h[BAND], ho[BAND];
#pragma unroll
    for (int b = 0; b < BAND; b++) w[b] = h[b] = ho[b] = 0.0;
    int tag = 0, to = 0;        // of the row a lane holds: bit 0 it is one, bit 1 it advanced the window, bits 2.. its first column
    int win0 = 0;               // (lane 0's is the one that counts: rows of R stored so far)
    int x = 0, prev_l = 0;      // the feeder: the same in every lane
    bool bubble = false;
    // the next row of the stream, requested a tick (at least) before lane 0 takes it
    int nl = 0;
    double nh[BAND];
    auto request = [&](int row) {
        const int rr = min(row, n_rows - 1);
        nl = rowl[rr];
#pragma unroll
        for (int b = 0; b < BAND; b++) nh[b] = b < we ? rowh[(size_t)rr * hs + b] : 0.0;
    };
    request(0);
    const int guard = 2 * n_rows + 2 * BAND + 8;
    for (int tick = 0; tick < guard; tick++) {
        // every row moves one lane on
#pragma unroll
        for (int b = 0; b < BAND; b++) h[b] = dpp_prev(ho[b]);
        tag = dpp_prev(to);
        // lane 0 takes the next row of the stream - one tick later if it enters one coefficient further on
        bool fed = false;
        if (x < n_rows) {
            if (nl > prev_l && !bubble) {
                bubble = true;
            } else {
                fed = true;
                if (lane == 0) {
                    tag = 1 | (nl > prev_l ? 2 : 0) | (nl << 2);
#pragma unroll
                    for (int b = 0; b < BAND; b++) h[b] = nh[b];
                }
                prev_l = nl;
                bubble = false;
                x++;
                request(x);
            }
        }
        if (!fed && lane == 0) tag = 0;
        const int valid = tag & 1, adv = tag & 2, j0 = tag >> 2;
        if (x >= n_rows && !fed && __builtin_amdgcn_ballot_w64(valid && stage) == 0) break;  // fed and drained
        // in the wake of a row that advanced the window the rows of R move one lane down
        const bool shift = valid && adv && stage;
        if (__builtin_amdgcn_ballot_w64(shift)) {
            double wn[BAND];
#pragma unroll
            for (int b = 0; b < BAND; b++) wn[b] = dpp_next(w[b]);
            if (shift) {
                if (lane == 0) {
#pragma unroll
                    for (int b = 0; b < BAND; b++) R[(size_t)win0 * kSmRow + b] = w[b];
                    for (int b = BAND; b < kSmRow - 1; b++) R[(size_t)win0 * kSmRow + b] = 0.0;
                    R[(size_t)win0 * kSmRow + kSmRow - 1] = 1.0 / w[0];
                    win0++;
                }
#pragma unroll
                for (int b = 0; b < BAND; b++) w[b] = lane == BAND - 1 ? 0.0 : wn[b];
            }
        }
        // this lane's rotation of the row it holds
        if (valid && stage && j0 + lane < ncf && h[0] != 0.0) {  // (a row that is used up meets zero pivots from there on)
            double cs, sn, dd;
            sm_givens(h[0], w[0], cs, sn, dd);
            w[0] = dd;
#pragma unroll
            for (int b = 1; b < BAND; b++) {
                const double s1 = h[b], s2 = w[b];
                w[b] = cs * s2 + sn * s1;
                h[b] = cs * s1 - sn * s2;
            }
        }
#pragma unroll
        for (int b = 0; b + 1 < BAND; b++) ho[b] = h[b + 1];
        ho[BAND - 1] = 0.0;
        to = tag;
    }
    // what is left in the lanes are the last rows of R (the last row enters at ncf - BAND or later: every row is stored)
    const int base = __builtin_amdgcn_readfirstlane(win0);
    if (stage && base + lane < ncf) {
        const int j = base + lane;
#pragma unroll
        for (int b = 0; b < BAND; b++) R[(size_t)j * kSmRow + b] = w[b];
        for (int b = BAND; b < kSmRow - 1; b++) R[(size_t)j * kSmRow + b] = 0.0;
        R[(size_t)j * kSmRow + kSmRow - 1] = 1.0 / w[0];
    }
}

// grid (2 axes, planes), one wave each: the band factor of this round's fit along the axes of degree K (the degree is a
// constant of the call: one instantiation per degree keeps each within its own registers). The wave first copies the row
// stream - entries and first columns, compact - into LDS: a row of 40-56 bytes per microsecond-long trip to HBM would leave
// the pipeline waiting most of the time.
__host__ __device__ inline size_t sm_factor_lds_bytes(int n_rows, int k) { return (size_t)n_rows * (k + 2) * sizeof(double) + (size_t)n_rows * sizeof(int); }
constexpr int kFactorBlock = 256;  // (four waves copy the stream into LDS, the first runs the pipeline)
template <int K>
__global__ __launch_bounds__(kFactorBlock) void k_smb_factor(SmPlaneDev *planes, int lds)
{
    extern __shared__ double sm_rows[];
    SmPlaneDev &P = planes[blockIdx.y];
    if (!P.active) return;
    const SmAxisDev &a = blockIdx.x ? P.x : P.y;
    if (a.k != K) return;
    if (!a.knots_changed && a.nb == 0) return;  // (the knot search: this axis kept its knots, and with them its factor)
    constexpr int BAND = K + 2;
    const bool merged = a.nb > 0;
    const int n_rows = a.m + a.nb, hs_g = merged ? kSmRow : 6, we = merged ? BAND : K + 1;
    const double *rowh_g = merged ? a.mg_h : a.hb;
    const int *rowl_g = merged ? a.mg_l : a.lb;
    if (lds) {
        double *hS = sm_rows;
        int *lS = (int *)(hS + (size_t)n_rows * BAND);
        sm_stage<kFactorBlock>(hS, n_rows * BAND, [&](int x) {
            const int row = x / BAND, e = x - row * BAND;
            return e < we ? rowh_g[(size_t)row * hs_g + e] : 0.0;
        });
        sm_stage<kFactorBlock>(lS, n_rows, [&](int x) { return rowl_g[x]; });
        __syncthreads();
        if (threadIdx.x < 64) sm_factor_stream<BAND>(hS, BAND, BAND, lS, n_rows, a.R, a.nc());
    } else if (threadIdx.x < 64) {
        sm_factor_stream<BAND>(rowh_g, hs_g, we, rowl_g, n_rows, a.R, a.nc());
    }
}

// ------------------------------------------------------------------ one directional least-squares solve
// dir 0: along the image rows' index for every image column (right-hand sides = the nx columns of z, solution U: nr x nx);
// dir 1: along the image columns' index for every row coefficient (right-hand sides = the nr columns of U', solution
// C': ncx x nr). pass 0: c = (R'R)^-1 A'd; pass 1: c += (R'R)^-1 [A; B/p]' ([d; 0] - [A; B/p] c).
// One lane per right-hand side q (coalesced over q): the samples are walked once, `s` holds the partial sums of the
// k + 1 coefficients the current knot interval touches; a coefficient whose last sample has passed is complete and takes
// its step of the forward substitution R'w = g at once (w -> G); then the back substitution R x = w.
// A sweep is ~3 m dependent steps: what it is built around is not WAITING at each of them. Everything that is the same
// for every lane - the entries and first column of each row, the rows of R - is staged in LDS once per workgroup (compact
// rows; when it does not fit, the sweep reads the tables where they are) and read four rows at a time; the right-hand
// side is the only stream from HBM and comes in tiles of 8 requested a tile ahead (so does the work vector on the way
// back); the column of R a closing needs is requested at the closing before. The second pass takes its residuals from a
// kernel of its own (k_smb_res: fully parallel) - the jump rows merged among the collocation rows as rows like any other.
constexpr int kSolveBlock = 256;
constexpr int kSmTileIn = 8, kSmGroup = 4;

// W: entries per row (k + 1 for the collocation rows alone, k + 2 with the jump rows among them)
template <int K, int W, bool LDS>
__device__ __forceinline__ void sm_sweep(const SmAxisDev &a, bool merged, const double *__restrict__ in, int si, int nrhs,
                                         double *__restrict__ c, double *__restrict__ g, int q, int pass, double *smem,
                                         unsigned long long *ref_delta = nullptr, unsigned long long *ref_scale = nullptr)
{
    constexpr int BAND = K + 2, TI = kSmTileIn, GR = kSmGroup;
    const int nc = a.nc(), n_rows = merged ? a.m + a.nb : a.m;
    const double *rowh_g = merged ? a.mg_h : a.hb;
    const int *rowl_g = merged ? a.mg_l : a.lb;
    const int hs_g = merged ? kSmRow : 6, we = merged ? W : K + 1;  // (a collocation row has k + 1 entries, the slot behind them is another row's)
    const double *rowh = rowh_g, *RT = a.R;
    const int *rowl = rowl_g;
    int hs = hs_g, rs = kSmRow, ri = kSmRow - 1;
    if (LDS) {
        double *hS = smem, *RS = hS + (size_t)n_rows * W;
        int *lS = (int *)(RS + (size_t)nc * (BAND + 1));
        sm_stage<kSolveBlock>(hS, n_rows * W, [&](int x) {
            const int row = x / W, e = x - row * W;
            return e < we ? rowh_g[(size_t)row * hs_g + e] : 0.0;
        });
        sm_stage<kSolveBlock>(RS, nc * (BAND + 1), [&](int x) {
            const int j = x / (BAND + 1), b = x - j * (BAND + 1);
            return a.R[(size_t)j * kSmRow + (b == BAND ? kSmRow - 1 : b)];
        });
        sm_stage<kSolveBlock>(lS, n_rows, [&](int x) { return rowl_g[x]; });
        __syncthreads();
        rowh = hS; hs = W;
        RT = RS; rs = BAND + 1; ri = BAND;
        rowl = lS;
    }
    if (q >= nrhs) return;
    double s[W], w[BAND - 1], rc[BAND - 1], rcinv;  // w[b]: w_{cur-1-b}; rc[b-1] = R(cur - b, cur), rcinv = 1 / R(cur, cur)
#pragma unroll
    for (int e = 0; e < W; e++) s[e] = 0.0;
#pragma unroll
    for (int b = 0; b < BAND - 1; b++) w[b] = 0.0;
    int cur = 0;
    auto fetch_col = [&](int j) {  // what the closing of coefficient j multiplies by, requested a closing ahead
#pragma unroll
        for (int b = 1; b < BAND; b++) rc[b - 1] = (j - b >= 0) ? RT[(size_t)(j - b) * rs + b] : 0.0;
        rcinv = RT[(size_t)j * rs + ri];
    };
    fetch_col(0);
    auto close = [&](double gj) {  // coefficient `cur` is complete: its step of R'w = g
        double sv = gj;
#pragma unroll
        for (int b = BAND - 1; b >= 1; b--) sv -= rc[b - 1] * w[b - 1];  // (the newest w last: the others do not wait for it)
        sv *= rcinv;
        g[(size_t)cur * nrhs + q] = sv;
#pragma unroll
        for (int b = BAND - 2; b > 0; b--) w[b] = w[b - 1];
        w[0] = sv;
        cur++;
        if (cur < nc) fetch_col(cur);
    };
    double tin[TI], tnext[TI];
    auto fetch = [&](int i0) {
#pragma unroll
        for (int u = 0; u < TI; u++) tnext[u] = (i0 + u < n_rows) ? in[(size_t)(i0 + u) * si + q] : 0.0;
    };
    fetch(0);
    for (int i0 = 0; i0 < n_rows; i0 += TI) {
#pragma unroll
        for (int u = 0; u < TI; u++) tin[u] = tnext[u];
        if (i0 + TI < n_rows) fetch(i0 + TI);
#pragma unroll
        for (int gq = 0; gq < TI / GR; gq++) {
            const int ib = i0 + gq * GR;
            if (ib < n_rows) {
                // the tables of four rows at once: one wait instead of eight
                int l4[GR];
                double h4[GR][W];
#pragma unroll
                for (int u = 0; u < GR; u++) {
                    const int ii = min(ib + u, n_rows - 1);
                    l4[u] = rowl[ii];
#pragma unroll
                    for (int e = 0; e < W; e++) h4[u][e] = (LDS || e < we) ? rowh[(size_t)ii * hs + e] : 0.0;
                }
                if (ib + GR <= n_rows && l4[GR - 1] == cur) {  // no coefficient closes inside the group
#pragma unroll
                    for (int u = 0; u < GR; u++)
#pragma unroll
                        for (int e = 0; e < W; e++) s[e] += h4[u][e] * tin[gq * GR + u];
                } else {
#pragma unroll
                    for (int u = 0; u < GR; u++)
                        if (ib + u < n_rows) {
                            while (cur < l4[u]) {
                                const double g0 = s[0];
#pragma unroll
                                for (int e = 0; e + 1 < W; e++) s[e] = s[e + 1];
                                s[W - 1] = 0.0;
                                close(g0);
                            }
#pragma unroll
                            for (int e = 0; e < W; e++) s[e] += h4[u][e] * tin[gq * GR + u];
                        }
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < W; e++)
        if (cur < nc) close(s[e]);
    // back substitution R x = w, then c = x (first pass) or c += x; w comes back in tiles requested a tile ahead
    double x[BAND - 1];
#pragma unroll
    for (int b = 0; b < BAND - 1; b++) x[b] = 0.0;
    double dmax = 0.0, cmax = 0.0;  // (second pass: how far the refinement moves this right-hand side's coefficients)
    auto fetch_g = [&](int jt) {
#pragma unroll
        for (int u = 0; u < TI; u++) tnext[u] = g[(size_t)max(jt - u, 0) * nrhs + q];
    };
    fetch_g(nc - 1);
    for (int jt = nc - 1; jt >= 0; jt -= TI) {
#pragma unroll
        for (int u = 0; u < TI; u++) tin[u] = tnext[u];
        if (jt - TI >= 0) fetch_g(jt - TI);
        double cold[TI];
        if (pass) {
#pragma unroll
            for (int u = 0; u < TI; u++) cold[u] = c[(size_t)max(jt - u, 0) * nrhs + q];
        }
#pragma unroll
        for (int u = 0; u < TI; u++) {  // (rows below 0 in the last tile come after the valid ones: computed on row 0's tables, never stored)
            const int j = max(jt - u, 0);
            double sv = tin[u];
#pragma unroll
            for (int b = BAND - 1; b >= 1; b--) sv -= RT[(size_t)j * rs + b] * x[b - 1];  // (entries beyond the matrix are zeros)
            sv *= RT[(size_t)j * rs + ri];
#pragma unroll
            for (int b = BAND - 2; b > 0; b--) x[b] = x[b - 1];
            x[0] = sv;
            if (jt - u >= 0) {
                const double cn = pass ? cold[u] + sv : sv;
                c[(size_t)j * nrhs + q] = cn;
                if (pass) {
                    dmax = fmax(dmax, fabs(sv));  // (fmax drops a NaN: a fit that went singular reports nothing here)
                    cmax = fmax(cmax, fabs(cn));
                }
            }
        }
    }
    if (pass && ref_delta) {
        for (int off = 32; off > 0; off >>= 1) {
            dmax = fmax(dmax, __shfl_xor(dmax, off, 64));
            cmax = fmax(cmax, __shfl_xor(cmax, off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMax(ref_delta, (unsigned long long)__double_as_longlong(dmax));
            atomicMax(ref_scale, (unsigned long long)__double_as_longlong(cmax));
        }
    }
}

// LDS the compact tables of a sweep take: rows x entries, the factor, the first columns
__host__ __device__ inline size_t sm_sweep_lds_bytes(int n_rows, int w, int k, int nc)
{
    return ((size_t)n_rows * w + (size_t)nc * (k + 3)) * sizeof(double) + (size_t)n_rows * sizeof(int);
}

// grid (right-hand sides / 256, planes); pass 0 walks the collocation rows with the data (z, or U' for dir 1), pass 1
// every row of [A; B / p] with the residuals k_smb_res left in RB
template <int K, int PASS>
__global__ __launch_bounds__(kSolveBlock) void k_smb_solve(SmPlaneDev *planes, int dir, int lds)
{
    extern __shared__ double sm_tables[];
    SmPlaneDev &P = planes[blockIdx.y];
    if (!P.active) return;
    const SmAxisDev &a = dir ? P.x : P.y;
    const int nr = P.y.nc();
    const int nrhs = dir ? nr : P.x.m;
    if ((int)(blockIdx.x * kSolveBlock) >= nrhs) return;
    const int q = blockIdx.x * kSolveBlock + threadIdx.x;
    double *c = dir ? P.CT : P.U;
    constexpr int W = PASS ? K + 2 : K + 1;
    const double *in = PASS ? P.RB : (dir ? P.UT : P.z);
    const int si = PASS ? nrhs : (dir ? nr : P.x.m);
    const bool merged = PASS && a.nb > 0;
    if (lds) sm_sweep<K, W, true>(a, merged, in, si, nrhs, c, P.G, q, PASS, sm_tables, &P.ref_delta[dir], &P.ref_scale[dir]);
    else sm_sweep<K, W, false>(a, merged, in, si, nrhs, c, P.G, q, PASS, nullptr, &P.ref_delta[dir], &P.ref_scale[dir]);
}

// residuals of every row of a direction's system after the first pass, in the order the second pass walks them:
// d - A c for the collocation rows, -(B / p) c for the jump rows. grid (right-hand sides / 256, rows / 16, planes): a
// workgroup takes 16 consecutive rows - their entries and first columns through LDS - and each lane carries the k + 2
// coefficients the current row touches in registers: rows come in the order of their first column, so the window moves
// on by one coefficient (one load) where the row stream does, and not at all while the rows share a knot interval.
constexpr int kResRows = 16;
__global__ __launch_bounds__(kBlock) void k_smb_res(SmPlaneDev *planes, int dir)
{
    __shared__ double hs[kResRows][kSmRow];
    __shared__ int l0s[kResRows], srcs[kResRows];
    const SmPlaneDev &P = planes[blockIdx.z];
    if (!P.active) return;
    const SmAxisDev &a = dir ? P.x : P.y;
    const int nr = P.y.nc(), nrhs = dir ? nr : P.x.m, nb = a.nb, n_rows = a.m + nb, k = a.k, nc = a.nc();
    const int x0 = blockIdx.y * kResRows;
    if ((int)(blockIdx.x * kBlock) >= nrhs || x0 >= n_rows) return;
    const int rows = min(kResRows, n_rows - x0);
    if (threadIdx.x < kResRows * kSmRow) {
        const int u = threadIdx.x / kSmRow, e = threadIdx.x % kSmRow;
        if (u < rows) {
            const int x = x0 + u, src = nb ? a.mg_src[x] : x;
            double h = 0.0;
            if (nb) h = a.mg_h[(size_t)x * kSmRow + e];
            else if (e <= k) h = a.hb[(size_t)src * 6 + e];
            hs[u][e] = h;
            if (e == 0) {
                l0s[u] = nb ? a.mg_l[x] : a.lb[src];
                srcs[u] = src;
            }
        }
    }
    __syncthreads();
    const int q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= nrhs) return;
    const double *__restrict__ in = dir ? P.UT : P.z;
    const double *__restrict__ c = dir ? P.CT : P.U;
    const int si = dir ? nr : P.x.m;
    constexpr int CW = 7;  // k + 2 <= 7 coefficients under a row
    const int width = k + 2;
    int cur = l0s[0];
    double cw[CW];
#pragma unroll
    for (int e = 0; e < CW; e++) cw[e] = (e < width && cur + e < nc) ? c[(size_t)(cur + e) * nrhs + q] : 0.0;
    for (int u = 0; u < rows; u++) {
        const int l0 = l0s[u];
        while (cur < l0) {
#pragma unroll
            for (int e = 0; e + 1 < CW; e++) cw[e] = cw[e + 1];
            cur++;
            // (the entry that comes into reach: position width - 1 of the window)
            const double fresh = (cur + width - 1 < nc) ? c[(size_t)(cur + width - 1) * nrhs + q] : 0.0;
#pragma unroll
            for (int e = 0; e < CW; e++)
                if (e == width - 1) cw[e] = fresh;
        }
        const int src = srcs[u];
        double v = src >= 0 ? in[(size_t)src * si + q] : 0.0;
#pragma unroll
        for (int e = 0; e < CW; e++) v -= hs[u][e] * cw[e];  // (entries beyond the row's own are zeros)
        P.RB[(size_t)(x0 + u) * nrhs + q] = v;
    }
}

// UT[j * nr + i] = U[i * nx + j] (LDS-tiled), grid (nx / 16, nr_max / 16, planes)
__global__ __launch_bounds__(kBlock) void k_smb_transpose(SmPlaneDev *planes)
{
    __shared__ double tile[16][17];
    const SmPlaneDev &P = planes[blockIdx.z];
    if (!P.active) return;
    const int rows = P.y.nc(), cols = P.x.m;
    if ((int)blockIdx.y * 16 >= rows) return;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
    if (i < rows && j < cols) tile[ty][tx] = P.U[(size_t)i * cols + j];
    __syncthreads();
    i = blockIdx.y * 16 + tx;
    j = blockIdx.x * 16 + ty;
    if (i < rows && j < cols) P.UT[(size_t)j * rows + i] = tile[tx][ty];
}

// Squared residuals of the fitted spline at the pixels of a 64 x 64 tile, summed per image row and per image column of
// the tile in a fixed order (k_smb_decide adds the tiles up): grid (tiles_x, tiles_y, planes), 4 waves, lane = column.
// The coefficients the tile's pixels touch - at most (64 + k)^2 - are staged in LDS (read from C' along its contiguous
// index): a lane's (k + 1)^2 reads per pixel would otherwise be as many gathers with a stride of nr doubles between lanes.
constexpr int kSmCt = kSmTile + 5;
__global__ __launch_bounds__(kBlock) void k_smb_resid(SmPlaneDev *planes)
{
    __shared__ double colacc[4][kSmTile];
    __shared__ double ct[kSmCt][kSmCt + 2];  // (an odd stride in doubles: lanes that differ in their row do not meet in a bank)
    __shared__ double hys[kSmTile][6];
    __shared__ int las[kSmTile];
    const SmPlaneDev &P = planes[blockIdx.z];
    if (!P.active) return;
    const SmAxisDev &ay = P.y, &ax = P.x;
    const int my = ay.m, mx = ax.m, nr = ay.nc(), ky = ay.k, kx = ax.k;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j_first = blockIdx.x * kSmTile, i_first = blockIdx.y * kSmTile;
    const int c_lo = ax.lb[j_first], c_n = ax.lb[min(j_first + kSmTile, mx) - 1] + kx + 1 - c_lo;
    const int a_lo = ay.lb[i_first], a_n = ay.lb[min(i_first + kSmTile, my) - 1] + ky + 1 - a_lo;
    for (int x = threadIdx.x; x < c_n * a_n; x += kBlock) {
        const int cc = x / a_n, aa = x - cc * a_n;
        ct[cc][aa] = P.CT[(size_t)(c_lo + cc) * nr + (a_lo + aa)];
    }
    for (int x = threadIdx.x; x < kSmTile * 6; x += kBlock) hys[x / 6][x % 6] = ay.hb[(size_t)min(i_first + x / 6, my - 1) * 6 + x % 6];
    if (threadIdx.x < kSmTile) las[threadIdx.x] = ay.lb[min(i_first + (int)threadIdx.x, my - 1)] - a_lo;
    __syncthreads();
    const int j = j_first + lane;
    const bool live = j < mx;
    double hx[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int lbx = 0;
    if (live) {
        lbx = ax.lb[j] - c_lo;
        for (int b = 0; b <= kx; b++) hx[b] = ax.hb[(size_t)j * 6 + b];
    }
    double csum = 0.0;
    for (int rr = 0; rr < kSmTile / 4; rr++) {
        const int i = i_first + wave * (kSmTile / 4) + rr;
        if (i >= my) break;
        double term = 0.0;
        if (live) {
            const int la = las[i - i_first];
            const double *hy = hys[i - i_first];
            double sv = 0.0;
            for (int b = 0; b <= kx; b++) {
                double r = 0.0;
                for (int e = 0; e <= ky; e++) r += hy[e] * ct[lbx + b][la + e];
                sv += hx[b] * r;
            }
            const double d = P.z[(size_t)i * mx + j] - sv;
            term = d * d;
        }
        csum += term;
        double rs = term;
        for (int off = 32; off > 0; off >>= 1) rs += __shfl_xor(rs, off, 64);
        if (lane == 0) P.rowpart[(size_t)blockIdx.x * my + i] = rs;
    }
    colacc[wave][lane] = csum;
    __syncthreads();
    if (wave == 0 && live)
        P.colpart[(size_t)blockIdx.y * mx + j] = ((colacc[0][lane] + colacc[1][lane]) + colacc[2][lane]) + colacc[3][lane];
}

// bispev of every plane's fitted spline at the map cells: grid (n_map / 256, planes)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_smb_eval(const ReprojectArgs a, const SmPlaneDev *planes)
{
    const int m = blockIdx.x * kBlock + threadIdx.x;
    if (m >= a.n_map) return;
    const SmPlaneDev &P = planes[blockIdx.y];
    const double nan = __builtin_nan("");
    const int nx = a.nx, ny = a.ny;
    const T *img = (const T *)a.cube + (size_t)P.plane * ny * nx;
    const double x = a.x_map[m], y = a.y_map[m];
    double val = nan;
    bool skip = isnan(x) || isnan(y) || P.all_nan;
    if (!skip && a.propagate_nan) {
        if (x < 0.0 || y < 0.0 || x > nx - 1 || y > ny - 1) {
            skip = true;
        } else {
            long ia = (long)fmax(floor(x), 0.0), ib = (long)fmin(ceil(x), nx - 1.0);
            long ja = (long)fmax(floor(y), 0.0), jb = (long)fmin(ceil(y), ny - 1.0);
            skip = isnan((double)img[(size_t)ja * nx + ia]) || isnan((double)img[(size_t)ja * nx + ib]) ||
                   isnan((double)img[(size_t)jb * nx + ia]) || isnan((double)img[(size_t)jb * nx + ib]);
        }
    }
    if (!skip) {
        const double xc = fmin(fmax(x, 0.0), nx - 1.0), yc = fmin(fmax(y, 0.0), ny - 1.0);
        // the knots are integer abscissae: the span of x is the span of floor(x)
        const int ly = P.y.span[(int)yc], lx = P.x.span[(int)xc];
        const int ky = P.y.k, kx = P.x.k, nr = P.y.nc();
        double hy[6], hx[6];
        bspl(P.y.t, ky, yc, ly, hy);
        bspl(P.x.t, kx, xc, lx, hx);
        double sv = 0.0;
        for (int q = 0; q <= kx; q++) {
            double r = 0.0;
            for (int p = 0; p <= ky; p++) r += hy[p] * P.CT[(size_t)(lx - kx + q) * nr + (ly - ky + p)];
            sv += hx[q] * r;
        }
        val = sv;
    }
    a.out[(size_t)P.plane * a.n_map + m] = val;
}

}  // namespace pm

namespace pmh {

int ensure_work(pm_ctx *ctx, size_t bytes);

// smoothing-spline reprojection of planes resident on the device
int reproject_smoothing_resident(pm_ctx *ctx, pm::ReprojectArgs a, int dtype, int k_rows, int k_cols, double s)
{
    using pm::kSmRow;
    using pm::kSmTile;
    const int ny = a.ny, nx = a.nx;
    const size_t npx = (size_t)ny * nx;
    const int tiles_x = (nx + kSmTile - 1) / kSmTile, tiles_y = (ny + kSmTile - 1) / kSmTile;
    // per-plane workspace: byte offsets inside a plane's slice of the arena
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    struct AxisOff { size_t hb, lb, span, R, Bp, t, fpint, nrdata, twin, mg_src, mg_l, mg_h; } ao[2];
    for (int ax = 0; ax < 2; ax++) {
        const size_t m = (size_t)(ax ? nx : ny), cap = m + 16;
        ao[ax].hb = take(cap * 6 * 8);
        ao[ax].lb = take(cap * 4);
        ao[ax].span = take(cap * 4);
        ao[ax].R = take(cap * kSmRow * 8);
        ao[ax].Bp = take(cap * kSmRow * 8);
        ao[ax].t = take(cap * 8);
        ao[ax].fpint = take(cap * 8);
        ao[ax].nrdata = take(cap * 4);
        ao[ax].twin = take(cap * 4);
        ao[ax].mg_src = take(2 * cap * 4);
        ao[ax].mg_l = take(2 * cap * 4);
        ao[ax].mg_h = take(2 * cap * kSmRow * 8);
    }
    const size_t o_rowpart = take((size_t)tiles_x * ny * 8), o_colpart = take((size_t)tiles_y * nx * 8);
    const size_t o_rowsum = take((size_t)ny * 8), o_colsum = take((size_t)nx * 8);
    const size_t o_U = take(npx * 8), o_UT = take(npx * 8), o_G = take(npx * 8), o_CT = take(npx * 8), o_RB = take(2 * npx * 8);
    const size_t per_plane = off;
    // planes whose searches advance together: what half of the free memory (at most 24 GiB) holds, PM_OPT_SM_BATCH_PLANES
    size_t free_b = 0, total_b = 0;
    PM_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
    const size_t have = ctx->sm_arena_bytes + ctx->work_bytes;  // (what a previous call left is ours to use again)
    size_t budget = std::min<size_t>((free_b + have) / 2, (size_t)24 << 30);
    size_t batch = std::max<size_t>(1, budget / (per_plane + npx * 8 + sizeof(pm::SmPlaneDev)));
    batch = std::min<size_t>(batch, (size_t)a.n_planes);
    batch = std::min<size_t>(batch, 4096);
    if (ctx->sm_batch_planes > 0) batch = std::min<size_t>(batch, (size_t)ctx->sm_batch_planes);
    const size_t desc_bytes = (batch * sizeof(pm::SmPlaneDev) + 255) & ~(size_t)255;
    const size_t arena_need = desc_bytes + 256 + batch * per_plane;
    if (arena_need > ctx->sm_arena_bytes) {
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->sm_arena) PM_HIP(ctx, hipFree(ctx->sm_arena));
        ctx->sm_arena = nullptr;
        ctx->sm_arena_bytes = 0;
        if (hipMalloc(&ctx->sm_arena, arena_need) != hipSuccess)
            return fail(ctx, PM_ERR_ALLOC, "allocation of the smoothing-spline workspace (%zu bytes) failed", arena_need);
        ctx->sm_arena_bytes = arena_need;
    }
    if (!ctx->sm_status_host) PM_HIP(ctx, hipHostMalloc((void **)&ctx->sm_status_host, 4 * sizeof(int)));
    // (more than 64 KB of dynamic LDS has to be enabled per kernel)
    // (HIP keeps function attributes per device: once per context, after pm's hipSetDevice - not once per process)
    if (!ctx->sm_lds_limit) ctx->sm_lds_limit = [] {
        const int want = 160 * 1024 - 256;
        bool ok = true;
        auto allow = [&](const void *f) { ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess; };
        allow((const void *)pm::k_smb_solve<1, 0>); allow((const void *)pm::k_smb_solve<1, 1>);
        allow((const void *)pm::k_smb_solve<2, 0>); allow((const void *)pm::k_smb_solve<2, 1>);
        allow((const void *)pm::k_smb_solve<3, 0>); allow((const void *)pm::k_smb_solve<3, 1>);
        allow((const void *)pm::k_smb_solve<4, 0>); allow((const void *)pm::k_smb_solve<4, 1>);
        allow((const void *)pm::k_smb_solve<5, 0>); allow((const void *)pm::k_smb_solve<5, 1>);
        allow((const void *)pm::k_smb_factor<1>); allow((const void *)pm::k_smb_factor<2>); allow((const void *)pm::k_smb_factor<3>);
        allow((const void *)pm::k_smb_factor<4>); allow((const void *)pm::k_smb_factor<5>);
        return ok ? (size_t)want : (size_t)(64 * 1024);
    }();
    const size_t lds_limit = ctx->sm_lds_limit;
    int rc = ensure_work(ctx, batch * npx * sizeof(double));
    if (rc != PM_OK) return rc;
    rc = ensure_stats(ctx, batch);
    if (rc != PM_OK) return rc;
    char *arena = (char *)ctx->sm_arena;
    pm::SmPlaneDev *planes = (pm::SmPlaneDev *)arena;
    int *status = (int *)(arena + desc_bytes);
    char *slices = arena + desc_bytes + 256;
    const hipStream_t st = ctx->stream;
    std::vector<pm::SmPlaneDev> desc(batch);
    const size_t plane_elems = npx;
    PM_HIP(ctx, hipMemsetAsync(status + 8, 0, 2 * sizeof(int), st));
    ctx->last_sm_knife_edges = 0;
    ctx->last_sm_ill_conditioned = 0;
    for (size_t p0 = 0; p0 < (size_t)a.n_planes; p0 += batch) {
        const int np = (int)std::min(batch, (size_t)a.n_planes - p0);
        pm::ReprojectArgs b = a;
        b.n_planes = np;
        b.cube = (const char *)a.cube + p0 * plane_elems * dtype_size(dtype);
        b.out = a.out + p0 * a.n_map;
        b.plane_stats = ctx->stats;
        PM_HIP(ctx, hipMemsetAsync(ctx->stats, 0, (size_t)np * sizeof(pm::PlaneStats), st));
        PM_HIP(ctx, hipMemsetAsync(ctx->hist, 0, (size_t)np * 512 * sizeof(unsigned int), st));
        pm_launch_clean_lazy(b, ctx->work, dtype, ctx->stats, ctx->hist, st);
        for (int pl = 0; pl < np; pl++) {
            pm::SmPlaneDev &d = desc[pl];
            std::memset(&d, 0, sizeof(d));
            char *sl = slices + (size_t)pl * per_plane;
            for (int ax = 0; ax < 2; ax++) {
                pm::SmAxisDev &A = ax ? d.x : d.y;
                A.hb = (double *)(sl + ao[ax].hb);
                A.lb = (int *)(sl + ao[ax].lb);
                A.span = (int *)(sl + ao[ax].span);
                A.R = (double *)(sl + ao[ax].R);
                A.Bp = (double *)(sl + ao[ax].Bp);
                A.t = (double *)(sl + ao[ax].t);
                A.fpint = (double *)(sl + ao[ax].fpint);
                A.nrdata = (int *)(sl + ao[ax].nrdata);
                A.twin = (int *)(sl + ao[ax].twin);
                A.mg_src = (int *)(sl + ao[ax].mg_src);
                A.mg_l = (int *)(sl + ao[ax].mg_l);
                A.mg_h = (double *)(sl + ao[ax].mg_h);
                A.m = ax ? nx : ny;
                A.k = ax ? k_cols : k_rows;
                A.n = 2 * (A.k + 1);
            }
            d.z = ctx->work + (size_t)pl * plane_elems;
            d.U = (double *)(sl + o_U); d.UT = (double *)(sl + o_UT); d.G = (double *)(sl + o_G); d.CT = (double *)(sl + o_CT);
            d.RB = (double *)(sl + o_RB);
            d.rowpart = (double *)(sl + o_rowpart); d.colpart = (double *)(sl + o_colpart);
            d.rowsum = (double *)(sl + o_rowsum); d.colsum = (double *)(sl + o_colsum);
            d.s = s;
            d.acc = 0.001 * s;
            d.plane = pl;
        }
        PM_HIP(ctx, hipMemcpyAsync(planes, desc.data(), (size_t)np * sizeof(pm::SmPlaneDev), hipMemcpyHostToDevice, st));
        // FITPACK's own caps bound the rounds: ny + nx knot iterations, then 20 steps of p (+ the closing decision)
        const int max_rounds = ny + nx + 20 + 2;
        const size_t decide_need = std::max(pm::sm_decide_lds_bytes(ny, k_rows), pm::sm_decide_lds_bytes(nx, k_cols));
        const size_t decide_lds = decide_need <= (size_t)(64 * 1024) ? decide_need : 0;
        int round = 0;
        for (; round < max_rounds; round++) {
            PM_HIP(ctx, hipMemsetAsync(status, 0, 4 * sizeof(int), st));
            hipLaunchKernelGGL(pm::k_smb_decide, dim3(np), dim3(64), decide_lds, st, planes, (const pm::PlaneStats *)ctx->stats, status,
                               round == 0 ? 1 : 0, tiles_x, tiles_y, decide_lds ? 1 : 0);
            PM_HIP(ctx, hipMemcpyAsync(ctx->sm_status_host, status, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
            PM_HIP(ctx, hipStreamSynchronize(st));
            const int n_active = ctx->sm_status_host[0], ncy = ctx->sm_status_host[1], ncx = ctx->sm_status_host[2];
            if (ctx->trace & 2) {  // PM_OPT_TRACE: the knot / smoothing-parameter search (the state of the batch's first planes)
                std::fprintf(stderr, "sm round %d: %d of %d planes fitting, coefficients up to (%d, %d)\n", round, n_active, np, ncy, ncx);
                for (int pl = 0; pl < std::min(np, 4); pl++) {
                    pm::SmPlaneDev d0;
                    PM_HIP(ctx, hipMemcpy(&d0, planes + pl, sizeof(d0), hipMemcpyDeviceToHost));
                    std::fprintf(stderr, "   plane %d: phase %d knots=(%d,%d) next p=%g last fp=%.17g refinement moved %.1e of scale\n", pl, d0.phase, d0.y.n, d0.x.n, d0.p, d0.fp, d0.refine_ratio);
                    if (ctx->trace & 1) {  // (with bit 0 as well: the interior knots themselves)
                        for (const pm::SmAxisDev *ax : {&d0.y, &d0.x}) {
                            std::vector<double> t((size_t)ax->n);
                            PM_HIP(ctx, hipMemcpy(t.data(), ax->t, t.size() * sizeof(double), hipMemcpyDeviceToHost));
                            std::fprintf(stderr, "      %c knots:", ax == &d0.y ? 'y' : 'x');
                            for (int i = ax->k + 1; i < ax->n - ax->k - 1; i++) std::fprintf(stderr, " %g", t[(size_t)i]);
                            std::fprintf(stderr, "\n");
                        }
                    }
                }
            }
            if (n_active == 0) break;
            if (ncy < 1 || ncy > ny || ncx < 1 || ncx > nx)
                return fail(ctx, PM_ERR_STATE, "smoothing-spline search: coefficient counts (%d, %d) outside the image (%d, %d)", ncy, ncx, ny, nx);
            hipLaunchKernelGGL(pm::k_smb_tables, dim3(2, np), dim3(pm::kBlock), 0, st, planes);
            const int any_p = ctx->sm_status_host[3];
            auto for_degree = [&](int k, auto &&launch) {  // the degree of an axis is a constant of the call
                switch (k) {
                case 1: launch(std::integral_constant<int, 1>()); break;
                case 2: launch(std::integral_constant<int, 2>()); break;
                case 3: launch(std::integral_constant<int, 3>()); break;
                case 4: launch(std::integral_constant<int, 4>()); break;
                default: launch(std::integral_constant<int, 5>()); break;
                }
            };
            {
                const size_t fy = pm::sm_factor_lds_bytes(ny + (any_p ? ncy : 0), k_rows), fx = pm::sm_factor_lds_bytes(nx + (any_p ? ncx : 0), k_cols);
                const size_t fb = std::max(fy, fx);
                auto factor = [&](auto K) {
                    hipLaunchKernelGGL(pm::k_smb_factor<decltype(K)::value>, dim3(2, np), dim3(pm::kFactorBlock), fb <= lds_limit ? fb : 0, st, planes, fb <= lds_limit ? 1 : 0);
                };
                for_degree(k_rows, factor);
                if (k_cols != k_rows) for_degree(k_cols, factor);
            }
            auto solve = [&](int dir) {
                // the compact tables of the LARGEST fit of the round decide between LDS and the tables where they are
                const int m = dir ? nx : ny, k = dir ? k_cols : k_rows, ncm = dir ? ncx : ncy, nrhs = dir ? ncy : nx;
                const int rows1 = m + (any_p ? ncm : 0);
                const size_t b0 = pm::sm_sweep_lds_bytes(m, k + 1, k, ncm), b1 = pm::sm_sweep_lds_bytes(rows1, k + 2, k, ncm);
                const dim3 grid((nrhs + pm::kSolveBlock - 1) / pm::kSolveBlock, np), block(pm::kSolveBlock);
                // (more than one refinement pass does not help where one does not: measured on the three ill-conditioned fits of
                //  the round-6 soak, 1 .. 4 passes - the iteration needs cond^2 eps < 1 to converge at all)
                for_degree(k, [&](auto K) {
                    constexpr int kk = decltype(K)::value;
                    hipLaunchKernelGGL((pm::k_smb_solve<kk, 0>), grid, block, b0 <= lds_limit ? b0 : 0, st, planes, dir, b0 <= lds_limit ? 1 : 0);
                    {
                        hipLaunchKernelGGL(pm::k_smb_res, dim3((nrhs + pm::kBlock - 1) / pm::kBlock, (rows1 + pm::kResRows - 1) / pm::kResRows, np), dim3(pm::kBlock), 0, st, planes, dir);
                        hipLaunchKernelGGL((pm::k_smb_solve<kk, 1>), grid, block, b1 <= lds_limit ? b1 : 0, st, planes, dir, b1 <= lds_limit ? 1 : 0);
                    }
                });
            };
            solve(0);
            hipLaunchKernelGGL(pm::k_smb_transpose, dim3((nx + 15) / 16, (ncy + 15) / 16, np), dim3(pm::kBlock), 0, st, planes);
            solve(1);
            hipLaunchKernelGGL(pm::k_smb_resid, dim3(tiles_x, tiles_y, np), dim3(pm::kBlock), 0, st, planes);
            PM_HIP(ctx, hipGetLastError());
        }
        if (round == max_rounds) return fail(ctx, PM_ERR_STATE, "smoothing-spline search did not end in %d rounds", max_rounds);
        const dim3 ge((a.n_map + pm::kBlock - 1) / pm::kBlock, np), bl(pm::kBlock);
        switch (dtype) {
        case PM_F64: hipLaunchKernelGGL(pm::k_smb_eval<double>, ge, bl, 0, st, b, (const pm::SmPlaneDev *)planes); break;
        case PM_F32: hipLaunchKernelGGL(pm::k_smb_eval<float>, ge, bl, 0, st, b, (const pm::SmPlaneDev *)planes); break;
        case PM_I16: hipLaunchKernelGGL(pm::k_smb_eval<int16_t>, ge, bl, 0, st, b, (const pm::SmPlaneDev *)planes); break;
        case PM_I32: hipLaunchKernelGGL(pm::k_smb_eval<int32_t>, ge, bl, 0, st, b, (const pm::SmPlaneDev *)planes); break;
        case PM_U8: hipLaunchKernelGGL(pm::k_smb_eval<uint8_t>, ge, bl, 0, st, b, (const pm::SmPlaneDev *)planes); break;
        case PM_U16: hipLaunchKernelGGL(pm::k_smb_eval<uint16_t>, ge, bl, 0, st, b, (const pm::SmPlaneDev *)planes); break;
        }
        PM_HIP(ctx, hipGetLastError());
        PM_HIP(ctx, hipMemcpyAsync(ctx->sm_status_host, status + 8, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
        PM_HIP(ctx, hipStreamSynchronize(st));
        ctx->last_sm_knife_edges = ctx->sm_status_host[0];  // (cumulative over the batches of the call)
        ctx->last_sm_ill_conditioned = ctx->sm_status_host[1];
        // (`desc` is rewritten for the next batch: the upload above must have been consumed - it has, every round synchronised)
    }
    return PM_OK;
}

}  // namespace pmh
