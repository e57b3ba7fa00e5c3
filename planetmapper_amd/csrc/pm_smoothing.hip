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
//     B / p (fpdisc), and the triangular band factor R of [A; B / p] by Givens rotations (fpgivs / fprota) - one lane per
//     (plane, axis), the rows of both blocks merged by first column so that the rows of R a row meets fit in registers (a
//     window that slides with the knot interval: R is stored once and never loaded);
//   * the fit kernels take blockIdx.y / .z = plane: every plane's fit of a round runs in the same launches, each plane
//     with its own knots. A directional solve is the corrected semi-normal equations R'R c = A'd + one refinement step
//     with the residual (error ~ cond(A) eps), fused per pass into ONE sweep per right-hand side: the lane walks the
//     samples once, a register window holds the k + 1 partial sums of A'd that are still open, each sum that closes is
//     fed straight into the forward substitution; the back substitution follows. No atomics anywhere: a plane's result is
//     the same bits whatever shares its launches (tests/test_gpu_splines_cube_scale.py);
//   * the host only sizes the grids: one 16-byte read-back per round (planes still searching, the largest coefficient
//     counts) - every wave's loops are bounded by that round's own tables, the round loop by FITPACK's own iteration caps.
// Planes whose search has ended stop taking part (their blocks leave at once); the cube is fitted in batches of as many
// planes as the workspace budget allows (4 work arrays of a plane each + ~100 KB of tables).
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
    int m, k, n, nplus, nb, knots_changed;
    __host__ __device__ int nc() const { return n - k - 1; }
    __host__ __device__ int nrint() const { return n - 2 * k - 1; }
};
struct SmPlaneDev {
    SmAxisDev y, x;   // FITPACK's "x" = the first array axis = image rows (y here), its "y" = image columns
    const double *z;  // the cleaned plane (ny x nx)
    double *U, *UT, *G, *CT;
    double *rowpart, *colpart;  // [tiles_x][ny], [tiles_y][nx]: residual sums per tile column / tile row
    double *rowsum, *colsum;
    double s, acc;
    double fp, fp0, fpold, reducy, reducx, fpms, p, p1, f1, p3, f3;
    int phase;  // 0 knot search, 1 smoothing parameter, 2 finished
    int iter, it2, lastdi, poly, ich1, ich3, fits;
    int active;   // a fit is wanted this round
    int all_nan;  // body_xy.py:1668-1670: the map of an all-NaN image is all NaN
    int plane;    // index in the chunk (cube, output, statistics)
};

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
__device__ bool sm_add_knot(SmAxisDev &a, int &n, int lane)
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
    const int maxpt = a.nrdata[number];
    int before = 0;
    for (int j = lane; j < number; j += 64) before += a.nrdata[j] + 1;
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
    const int ihalf = maxpt / 2 + 1, nrx = 1 + before + ihalf;  // 1-based data index: abscissa nrx - 1
    __syncthreads();
    shift_up(a.fpint, number + 1, nri - 1, lane);
    shift_up(a.nrdata, number + 1, nri - 1, lane);
    shift_up(a.t, number + k + 1, n - 1, lane);
    if (lane == 0) {
        a.nrdata[number] = ihalf - 1;
        a.nrdata[number + 1] = maxpt - ihalf;
        a.fpint[number] = best * (double)(ihalf - 1) / (double)maxpt;
        a.fpint[number + 1] = best * (double)(maxpt - ihalf) / (double)maxpt;
        a.t[number + k + 1] = (double)(nrx - 1);
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
    }
}

// One wave per plane: close the fit of the last round (residual sums -> fp, per-interval sums), take FITPACK's decision
// (fpregr: more knots / the next p / finished) and publish what the next fit is. status: [0] planes with a fit to run,
// [1] / [2] the largest coefficient counts along y / x among them.
__global__ __launch_bounds__(64) void k_smb_decide(SmPlaneDev *planes, const PlaneStats *stats, int *status, int first, int tiles_x,
                                                   int tiles_y)
{
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
        }
        changed_y = changed_x = true;
        fp = fp0 = fpold = reducy = reducx = fpms = 0.0;
        p = -1.0;
        p1 = f1 = f3 = 0.0;
        p3 = -1.0;
        iter = it2 = lastdi = ich1 = ich3 = 0;
        nplus_y = nplus_x = 0;
        poly = 1;
        if (lane == 0) P.fits = 0;
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
                bool added = false;
                if (first_axis) {
                    nplus_y = nply;
                    for (int l = 0; l < nply; l++) {
                        if (!sm_add_knot(P.y, ny_n, lane)) break;
                        added = true;
                        if (ny_n == nmaxy) break;
                    }
                    changed_y = added;
                } else {
                    nplus_x = nplx;
                    for (int l = 0; l < nplx; l++) {
                        if (!sm_add_knot(P.x, nx_n, lane)) break;
                        added = true;
                        if (nx_n == nmaxx) break;
                    }
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
    }
}

// Triangular band R of the QR factor of [A; B / p] by Givens rotations (fpgivs / fprota), one lane. FITPACK rotates the
// collocation rows in first and then each jump row from its first column TO THE END of the band (its fill-in travels:
// O(nb x nc) rotations). R is unique (positive diagonal), so here the rows of both blocks are merged in the order of their
// first column instead: a row that enters at coefficient j0 then meets only rows of R that hold nothing beyond column
// j0 + BAND - 1, and is used up after BAND rotations. Those BAND rows of R live in registers, a window that slides with
// j0 - a row that falls out of it is final and is stored with the reciprocal of its diagonal (what the substitutions
// multiply by); nothing of R is ever loaded. The next row of each block is in flight while the current one is rotated in.
template <int BAND>
__device__ void sm_factor_axis(const SmAxisDev &a)
{
    const int ncf = a.nc(), m = a.m, nb = a.nb, k = a.k;
    const double *__restrict__ hb = a.hb;
    const double *__restrict__ Bp = a.Bp;
    const int *__restrict__ lb = a.lb;
    double *__restrict__ R = a.R;
    double W[BAND][BAND], h[BAND], hd[BAND], hj[BAND];
#pragma unroll
    for (int r = 0; r < BAND; r++)
#pragma unroll
        for (int b = 0; b < BAND; b++) W[r][b] = 0.0;
    int win0 = 0, i = 0, r = 0, lbi = 0;
    auto fetch_data = [&](int ii) {
        lbi = lb[ii];
#pragma unroll
        for (int e = 0; e < BAND; e++) hd[e] = e <= k ? hb[(size_t)ii * 6 + e] : 0.0;
    };
    auto fetch_jump = [&](int rr) {
#pragma unroll
        for (int e = 0; e < BAND; e++) hj[e] = Bp[(size_t)rr * kSmRow + e];
    };
    auto retire = [&](int j, const double *w) {
#pragma unroll
        for (int b = 0; b < BAND; b++) R[(size_t)j * kSmRow + b] = w[b];
        for (int b = BAND; b < kSmRow - 1; b++) R[(size_t)j * kSmRow + b] = 0.0;
        R[(size_t)j * kSmRow + kSmRow - 1] = 1.0 / w[0];
    };
    fetch_data(0);
    if (nb > 0) fetch_jump(0);
    while (i < m || r < nb) {
        const bool data = i < m && (r >= nb || lbi <= r);
        int j0;
        if (data) {
            j0 = lbi;
#pragma unroll
            for (int e = 0; e < BAND; e++) h[e] = hd[e];
            if (++i < m) fetch_data(i);
        } else {
            j0 = r;
#pragma unroll
            for (int e = 0; e < BAND; e++) h[e] = hj[e];
            if (++r < nb) fetch_jump(r);
        }
        while (win0 < j0) {
            retire(win0, W[0]);
#pragma unroll
            for (int q = 0; q + 1 < BAND; q++)
#pragma unroll
                for (int b = 0; b < BAND; b++) W[q][b] = W[q + 1][b];
#pragma unroll
            for (int b = 0; b < BAND; b++) W[BAND - 1][b] = 0.0;
            win0++;
        }
        bool live = true;
#pragma unroll
        for (int q = 0; q < BAND; q++) {
            if (live && j0 + q < ncf) {
                const double piv = h[0];
                if (piv != 0.0) {
                    const double ww = W[q][0], store = fabs(piv);
                    const double dd = (store >= ww) ? store * sqrt(1.0 + (ww / piv) * (ww / piv)) : ww * sqrt(1.0 + (piv / ww) * (piv / ww));
                    const double cs = ww / dd, sn = piv / dd;
                    W[q][0] = dd;
#pragma unroll
                    for (int b = 1; b < BAND; b++) {
                        const double s1 = h[b], s2 = W[q][b];
                        W[q][b] = cs * s2 + sn * s1;
                        h[b] = cs * s1 - sn * s2;
                    }
                }
                bool any = false;
#pragma unroll
                for (int b = 0; b + 1 < BAND; b++) {
                    h[b] = h[b + 1];
                    any |= (h[b] != 0.0);
                }
                h[BAND - 1] = 0.0;
                live = any;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < BAND; q++)
        if (win0 + q < ncf) retire(win0 + q, W[q]);  // (the last row enters at ncf - BAND or later: every row of R is stored)
}

// grid (2 axes, planes), one wave each, lane 0 at work: the band factor(s) of this round's fit
__global__ __launch_bounds__(64) void k_smb_factor(SmPlaneDev *planes)
{
    SmPlaneDev &P = planes[blockIdx.y];
    if (!P.active || threadIdx.x != 0) return;
    const SmAxisDev &a = blockIdx.x ? P.x : P.y;
    if (!a.knots_changed && a.nb == 0) return;  // (the knot search: this axis kept its knots, and with them its factor)
    switch (a.k) {
    case 1: sm_factor_axis<3>(a); break;
    case 2: sm_factor_axis<4>(a); break;
    case 3: sm_factor_axis<5>(a); break;
    case 4: sm_factor_axis<6>(a); break;
    default: sm_factor_axis<7>(a); break;
    }
}

// ------------------------------------------------------------------ one directional least-squares solve
// dir 0: along the image rows' index for every image column (right-hand sides = the nx columns of z, solution U: nr x nx);
// dir 1: along the image columns' index for every row coefficient (right-hand sides = the nr columns of U', solution
// C': ncx x nr). pass 0: c = (R'R)^-1 A'd; pass 1: c += (R'R)^-1 (A'(d - A c) - (B/p)'(B/p) c).
// One lane per right-hand side q (coalesced over q): the samples are walked once, `s` holds the partial sums of the
// k + 1 coefficients the current knot interval touches; a coefficient whose last sample has passed is complete and takes
// its step of the forward substitution R'w = g at once (w -> G); then the back substitution R x = w.
template <int K>
__device__ void sm_solve(const SmPlaneDev &P, const SmAxisDev &a, const double *__restrict__ in, int si, int nrhs, double *__restrict__ c,
                         double *__restrict__ g, int q, int pass)
{
    constexpr int BAND = K + 2;
    const int m = a.m, nc = a.nc(), nb = pass ? a.nb : 0;
    const double *__restrict__ R = a.R;
    const double *__restrict__ hb = a.hb;
    const double *__restrict__ Bp = a.Bp;
    const int *__restrict__ lb = a.lb;
    double s[K + 1], w[BAND - 1], vw[BAND];  // w[b]: w_{j-1-b}; vw[b]: v_{j-b} of the jump rows
#pragma unroll
    for (int e = 0; e <= K; e++) s[e] = 0.0;
#pragma unroll
    for (int b = 0; b < BAND - 1; b++) w[b] = 0.0;
#pragma unroll
    for (int b = 0; b < BAND; b++) vw[b] = 0.0;
    auto close = [&](int j, double gj) {
        if (nb) {  // the jump rows have a zero right-hand side: their residual is -(B/p) c
#pragma unroll
            for (int b = BAND - 1; b > 0; b--) vw[b] = vw[b - 1];
            double v = 0.0;
            if (j < nb) {
#pragma unroll
                for (int e = 0; e < BAND; e++) v += Bp[(size_t)j * kSmRow + e] * c[(size_t)(j + e) * nrhs + q];
            }
            vw[0] = v;
#pragma unroll
            for (int b = 0; b < BAND; b++)
                if (j - b >= 0 && j - b < nb) gj -= Bp[(size_t)(j - b) * kSmRow + b] * vw[b];
        }
        double sv = gj;
#pragma unroll
        for (int b = 1; b < BAND; b++)
            if (j - b >= 0) sv -= R[(size_t)(j - b) * kSmRow + b] * w[b - 1];
        sv *= R[(size_t)j * kSmRow + kSmRow - 1];
        g[(size_t)j * nrhs + q] = sv;
#pragma unroll
        for (int b = BAND - 2; b > 0; b--) w[b] = w[b - 1];
        w[0] = sv;
    };
    int cur = 0;
    for (int i = 0; i < m; i++) {
        const int l0 = lb[i];
        while (cur < l0) {
            close(cur, s[0]);
#pragma unroll
            for (int e = 0; e < K; e++) s[e] = s[e + 1];
            s[K] = 0.0;
            cur++;
        }
        double r = in[(size_t)i * si + q];
        if (pass) {
#pragma unroll
            for (int e = 0; e <= K; e++) r -= hb[(size_t)i * 6 + e] * c[(size_t)(l0 + e) * nrhs + q];
        }
#pragma unroll
        for (int e = 0; e <= K; e++) s[e] += hb[(size_t)i * 6 + e] * r;
    }
#pragma unroll
    for (int e = 0; e <= K; e++)
        if (cur + e < nc) close(cur + e, s[e]);
    // back substitution R x = w, then c = x (first pass) or c += x
    double x[BAND - 1];
#pragma unroll
    for (int b = 0; b < BAND - 1; b++) x[b] = 0.0;
    for (int j = nc - 1; j >= 0; j--) {
        double sv = g[(size_t)j * nrhs + q];
#pragma unroll
        for (int b = 1; b < BAND; b++) sv -= R[(size_t)j * kSmRow + b] * x[b - 1];  // (entries beyond the matrix are zeros)
        sv *= R[(size_t)j * kSmRow + kSmRow - 1];
#pragma unroll
        for (int b = BAND - 2; b > 0; b--) x[b] = x[b - 1];
        x[0] = sv;
        const size_t ci = (size_t)j * nrhs + q;
        c[ci] = pass ? c[ci] + sv : sv;
    }
}

__global__ __launch_bounds__(64) void k_smb_solve(SmPlaneDev *planes, int dir, int pass)
{
    const SmPlaneDev &P = planes[blockIdx.y];
    if (!P.active) return;
    const SmAxisDev &a = dir ? P.x : P.y;
    const int nr = P.y.nc();
    const int nrhs = dir ? nr : P.x.m;
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= nrhs) return;
    const double *in = dir ? P.UT : P.z;
    const int si = dir ? nr : P.x.m;
    double *c = dir ? P.CT : P.U;
    switch (a.k) {
    case 1: sm_solve<1>(P, a, in, si, nrhs, c, P.G, q, pass); break;
    case 2: sm_solve<2>(P, a, in, si, nrhs, c, P.G, q, pass); break;
    case 3: sm_solve<3>(P, a, in, si, nrhs, c, P.G, q, pass); break;
    case 4: sm_solve<4>(P, a, in, si, nrhs, c, P.G, q, pass); break;
    default: sm_solve<5>(P, a, in, si, nrhs, c, P.G, q, pass); break;
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
__global__ __launch_bounds__(kBlock) void k_smb_resid(SmPlaneDev *planes)
{
    __shared__ double colacc[4][kSmTile];
    const SmPlaneDev &P = planes[blockIdx.z];
    if (!P.active) return;
    const SmAxisDev &ay = P.y, &ax = P.x;
    const int my = ay.m, mx = ax.m, nr = ay.nc(), ky = ay.k, kx = ax.k;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * kSmTile + lane;
    const bool live = j < mx;
    double hx[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    int lbx = 0;
    if (live) {
        lbx = ax.lb[j];
        for (int b = 0; b <= kx; b++) hx[b] = ax.hb[(size_t)j * 6 + b];
    }
    double csum = 0.0;
    for (int rr = 0; rr < kSmTile / 4; rr++) {
        const int i = blockIdx.y * kSmTile + wave * (kSmTile / 4) + rr;
        if (i >= my) break;
        double term = 0.0;
        if (live) {
            const int la = ay.lb[i];
            const double *hy = ay.hb + (size_t)i * 6;
            double sv = 0.0;
            for (int b = 0; b <= kx; b++) {
                double r = 0.0;
                for (int e = 0; e <= ky; e++) r += hy[e] * P.CT[(size_t)(lbx + b) * nr + (la + e)];
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
    struct AxisOff { size_t hb, lb, span, R, Bp, t, fpint, nrdata; } ao[2];
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
    }
    const size_t o_rowpart = take((size_t)tiles_x * ny * 8), o_colpart = take((size_t)tiles_y * nx * 8);
    const size_t o_rowsum = take((size_t)ny * 8), o_colsum = take((size_t)nx * 8);
    const size_t o_U = take(npx * 8), o_UT = take(npx * 8), o_G = take(npx * 8), o_CT = take(npx * 8);
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
                A.m = ax ? nx : ny;
                A.k = ax ? k_cols : k_rows;
                A.n = 2 * (A.k + 1);
            }
            d.z = ctx->work + (size_t)pl * plane_elems;
            d.U = (double *)(sl + o_U); d.UT = (double *)(sl + o_UT); d.G = (double *)(sl + o_G); d.CT = (double *)(sl + o_CT);
            d.rowpart = (double *)(sl + o_rowpart); d.colpart = (double *)(sl + o_colpart);
            d.rowsum = (double *)(sl + o_rowsum); d.colsum = (double *)(sl + o_colsum);
            d.s = s;
            d.acc = 0.001 * s;
            d.plane = pl;
        }
        PM_HIP(ctx, hipMemcpyAsync(planes, desc.data(), (size_t)np * sizeof(pm::SmPlaneDev), hipMemcpyHostToDevice, st));
        // FITPACK's own caps bound the rounds: ny + nx knot iterations, then 20 steps of p (+ the closing decision)
        const int max_rounds = ny + nx + 20 + 2;
        int round = 0;
        for (; round < max_rounds; round++) {
            PM_HIP(ctx, hipMemsetAsync(status, 0, 4 * sizeof(int), st));
            hipLaunchKernelGGL(pm::k_smb_decide, dim3(np), dim3(64), 0, st, planes, (const pm::PlaneStats *)ctx->stats, status,
                               round == 0 ? 1 : 0, tiles_x, tiles_y);
            PM_HIP(ctx, hipMemcpyAsync(ctx->sm_status_host, status, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
            PM_HIP(ctx, hipStreamSynchronize(st));
            const int n_active = ctx->sm_status_host[0], ncy = ctx->sm_status_host[1], ncx = ctx->sm_status_host[2];
            if (ctx->trace & 2) {  // PM_OPT_TRACE: the knot / smoothing-parameter search (the state of the batch's first planes)
                std::fprintf(stderr, "sm round %d: %d of %d planes fitting, coefficients up to (%d, %d)\n", round, n_active, np, ncy, ncx);
                for (int pl = 0; pl < std::min(np, 4); pl++) {
                    pm::SmPlaneDev d0;
                    PM_HIP(ctx, hipMemcpy(&d0, planes + pl, sizeof(d0), hipMemcpyDeviceToHost));
                    std::fprintf(stderr, "   plane %d: phase %d knots=(%d,%d) next p=%g last fp=%.17g\n", pl, d0.phase, d0.y.n, d0.x.n, d0.p, d0.fp);
                }
            }
            if (n_active == 0) break;
            if (ncy < 1 || ncy > ny || ncx < 1 || ncx > nx)
                return fail(ctx, PM_ERR_STATE, "smoothing-spline search: coefficient counts (%d, %d) outside the image (%d, %d)", ncy, ncx, ny, nx);
            hipLaunchKernelGGL(pm::k_smb_tables, dim3(2, np), dim3(pm::kBlock), 0, st, planes);
            hipLaunchKernelGGL(pm::k_smb_factor, dim3(2, np), dim3(64), 0, st, planes);
            hipLaunchKernelGGL(pm::k_smb_solve, dim3((nx + 63) / 64, np), dim3(64), 0, st, planes, 0, 0);
            hipLaunchKernelGGL(pm::k_smb_solve, dim3((nx + 63) / 64, np), dim3(64), 0, st, planes, 0, 1);
            hipLaunchKernelGGL(pm::k_smb_transpose, dim3((nx + 15) / 16, (ncy + 15) / 16, np), dim3(pm::kBlock), 0, st, planes);
            hipLaunchKernelGGL(pm::k_smb_solve, dim3((ncy + 63) / 64, np), dim3(64), 0, st, planes, 1, 0);
            hipLaunchKernelGGL(pm::k_smb_solve, dim3((ncy + 63) / 64, np), dim3(64), 0, st, planes, 1, 1);
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
        // (`desc` is rewritten for the next batch: the upload above must have been consumed - it has, every round synchronised)
    }
    return PM_OK;
}

}  // namespace pmh
