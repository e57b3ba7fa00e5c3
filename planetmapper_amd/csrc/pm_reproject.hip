// pm_reproject.hip -- host side of pm_map_cube: the reprojection of planes resident on the
// device for every interpolation mode of BodyXY.map_img (body_xy.py:1414-1904). Kernels:
// pm_kernels_reproject.hip. Host work here is set-up only: knots + banded factors of the spline
// axes, the footprint / oversampling grid of 'smooth', and the scalar knot / parameter search of
// the smoothing splines; all per-pixel arithmetic runs on the GPU.
#include <atomic>
#include <thread>

#include "pm_host.hip.h"

namespace pmh {

// Knots and banded LU of the B-spline collocation matrix of one axis (n unit-spaced samples,
// degree k, FITPACK's s = 0 knot placement), uploaded once per (n, k).
int ensure_axis(pm_ctx *ctx, int which, int n, int k, pm::SplineAxis &out)
{
    pm_ctx::AxisCache &ac = ctx->axis[which];
    if (ac.n != n || ac.k != k) {
        const int w = 2 * k + 1;
        std::vector<double> t((size_t)n + k + 1, 0.0), a((size_t)n * w, 0.0);
        for (int i = 0; i <= k; i++) t[n + i] = (double)(n - 1);
        const int k3 = k / 2;
        for (int l = 0; l < n - k - 1; l++) {
            const int j = k3 + 1 + l;
            t[k + 1 + l] = (k3 * 2 == k) ? 0.5 * ((double)j + (double)(j - 1)) : (double)j;
        }
        for (int i = 0; i < n; i++) {
            int l = k;
            while (l < n - 1 && (double)i >= t[l + 1]) l++;
            double h[6], hh[6];
            h[0] = 1.0;
            for (int j = 1; j <= k; j++) {  // fpbspl
                for (int q = 0; q < j; q++) hh[q] = h[q];
                h[0] = 0.0;
                for (int q = 1; q <= j; q++) {
                    const int li = l + q, lj = li - j;
                    const double f = hh[q - 1] / (t[li] - t[lj]);
                    h[q - 1] += f * (t[li] - (double)i);
                    h[q] = f * ((double)i - t[lj]);
                }
            }
            for (int q = 0; q <= k; q++) {
                const int j = l - k + q;
                a[(size_t)i * w + (j - i + k)] = h[q];
            }
        }
        for (int p = 0; p < n; p++) {  // banded LU, no pivoting (totally positive matrix)
            const double piv = a[(size_t)p * w + k];
            for (int i = p + 1; i <= p + k && i < n; i++) {
                const double f = a[(size_t)i * w + (p - i + k)] / piv;
                if (f == 0.0) continue;
                a[(size_t)i * w + (p - i + k)] = f;
                for (int j = p + 1; j <= p + k && j < n; j++) a[(size_t)i * w + (j - i + k)] -= f * a[(size_t)p * w + (j - p + k)];
            }
        }
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ac.t) PM_HIP(ctx, hipFree(ac.t));
        if (ac.lu) PM_HIP(ctx, hipFree(ac.lu));
        ac = pm_ctx::AxisCache();
        if (hipMalloc((void **)&ac.t, t.size() * sizeof(double)) != hipSuccess ||
            hipMalloc((void **)&ac.lu, a.size() * sizeof(double)) != hipSuccess)
            return fail(ctx, PM_ERR_ALLOC, "hipMalloc of spline factors failed");
        PM_HIP(ctx, hipMemcpy(ac.t, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice));
        PM_HIP(ctx, hipMemcpy(ac.lu, a.data(), a.size() * sizeof(double), hipMemcpyHostToDevice));
        ac.n = n;
        ac.k = k;
    }
    out.t = ac.t;
    out.lu = ac.lu;
    out.n = n;
    out.k = k;
    return PM_OK;
}

int ensure_work(pm_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->work_bytes) return PM_OK;
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->work) PM_HIP(ctx, hipFree(ctx->work));
    ctx->work = nullptr;
    ctx->work_bytes = 0;
    hipError_t e = hipMalloc((void **)&ctx->work, bytes);
    if (e != hipSuccess) return fail(ctx, PM_ERR_ALLOC, "hipMalloc(%zu) of the spline workspace failed", bytes);
    ctx->work_bytes = bytes;
    return PM_OK;
}

// get_xy_pchip body_xy.py:1724-1741: original pixel coordinates within `pad` of the map's
// footprint [lo, hi], refined by the largest factor <= oversample_by whose grid fits
// max_size. Returns false if no pixel is in range or the grid has a single point.
bool smooth_axis(int n, double lo, double hi, int oversample_by, int max_size, pm::SmoothAxis &ax)
{
    const double pad = 5.0;  // limit_padding
    int first = -1, last = -1;
    for (int j = 0; j < n; j++)
        if ((double)j >= lo - pad && (double)j <= hi + pad) {
            if (first < 0) first = j;
            last = j;
        }
    if (first < 0) return false;
    const long old_size = last - first + 1;
    ax.first = first;
    ax.last = last;
    ax.num = (int)old_size;
    ax.oversampled = 0;
    ax.step = 1.0;
    for (long o = oversample_by; o > 1; o--) {
        const long num = old_size * o - (o - 1);
        if (num <= max_size) {
            if (num > 1) {
                ax.num = (int)num;
                ax.oversampled = 1;
                ax.step = ((double)last - (double)first) / (double)(num - 1);  // numpy.linspace
            }
            break;
        }
    }
    return ax.num >= 2;
}

// 'smooth' reprojection of planes resident on the device. `limits`: nanmin / nanmax of the
// x and y maps (host values).
int reproject_smooth_resident(pm_ctx *ctx, const pm::ReprojectArgs &a, int dtype, const double *limits)
{
    if (!(limits[0] <= limits[1]) || !(limits[2] <= limits[3])) {
        // no visible map cell: every output is NaN (the reference only gets this far for
        // all-NaN planes; for others its own axis trimming fails with an IndexError)
        std::vector<double> nanrow((size_t)a.n_map, std::nan(""));
        for (int p = 0; p < a.n_planes; p++)
            PM_HIP(ctx, hipMemcpyAsync(a.out + (size_t)p * a.n_map, nanrow.data(), (size_t)a.n_map * sizeof(double),
                                       hipMemcpyHostToDevice, ctx->stream));
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return PM_OK;
    }
    pm::SmoothArgs sm;
    if (!smooth_axis(a.nx, limits[0], limits[1], ctx->smooth_oversample_by, ctx->smooth_max_size, sm.x) ||
        !smooth_axis(a.ny, limits[2], limits[3], ctx->smooth_oversample_by, ctx->smooth_max_size, sm.y))
        return fail(ctx, PM_ERR_INVALID_ARGUMENT,
                    "smooth interpolation needs at least two image pixels per axis near the mapped region");
    pm_launch_reproject_smooth(a, sm, dtype, ctx->stream);
    PM_HIP(ctx, hipGetLastError());
    return PM_OK;
}


// ------------------------------------------------------------------ smoothing splines (spline_smoothing > 0)
// BodyXY._do_spline_interpolation with s > 0 (body_xy.py:1673-1680) is FITPACK's `regrid`
// (Dierckx): grow the knot sets from the least-squares polynomial until the least-squares
// spline has a residual sum fp <= s, then find the smoothing parameter p with fp(p) = s by
// rational interpolation. The control flow (a few dozen scalar decisions per plane) and the
// QR factors of the two small banded design matrices stay on the host; every fit - two
// directional least-squares solves over all image columns / coefficient rows and the residual
// sums over all pixels - runs on the GPU.
struct SmAxis {
    int m = 0, k = 0, n = 0, nplus = 0;
    std::vector<double> t, fpint;
    std::vector<int> nrdata;
    std::vector<double> hb, R, Bp;  // tables for the current knots / p
    std::vector<int> lb, span, first, last;
    bool knots_changed = true;
    int nc() const { return n - k - 1; }
    int nrint() const { return n - 2 * k - 1; }

    void init(int m_, int k_)
    {
        m = m_; k = k_; n = 2 * (k + 1); nplus = 0;
        t.assign((size_t)m + k + 2, 0.0);
        for (int i = 0; i <= k; i++) t[k + 1 + i] = (double)(m - 1);
        fpint.assign((size_t)m + 1, 0.0);
        nrdata.assign((size_t)m + 1, 0);
        nrdata[0] = m - 2;
        knots_changed = true;
    }
    // fpknot: new knot at the middle data point of the interval with the largest residual sum
    void add_knot()
    {
        const int nri = nrint();
        double fpmax = 0.0;
        int number = -1, maxpt = 0, maxbeg = 0, jbegin = 1;
        for (int j = 0; j < nri; j++) {
            const int jp = nrdata[j];
            if (!(fpmax >= fpint[j] || jp == 0)) { fpmax = fpint[j]; number = j; maxpt = jp; maxbeg = jbegin; }
            jbegin += jp + 1;
        }
        if (number < 0) return;
        const int ihalf = maxpt / 2 + 1, nrx = maxbeg + ihalf;  // 1-based data index: abscissa nrx - 1
        for (int j = nri - 1; j > number; j--) { fpint[j + 1] = fpint[j]; nrdata[j + 1] = nrdata[j]; }
        for (int j = n - 1; j >= number + k + 1; j--) t[j + 1] = t[j];
        nrdata[number] = ihalf - 1;
        nrdata[number + 1] = maxpt - ihalf;
        fpint[number] = fpmax * (double)nrdata[number] / (double)maxpt;
        fpint[number + 1] = fpmax * (double)nrdata[number + 1] / (double)maxpt;
        t[number + k + 1] = (double)(nrx - 1);
        n += 1;
        knots_changed = true;
    }
    static void bspl(const double *t, int k, double x, int l, double *h)
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
    // B-spline values of every sample + knot interval of every integer abscissa
    void build_tables()
    {
        hb.assign((size_t)m * 6, 0.0);
        lb.assign((size_t)m, 0);
        span.assign((size_t)m, 0);
        int l = k;
        for (int i = 0; i < m; i++) {
            while (l < n - k - 2 && (double)i >= t[l + 1]) l++;
            bspl(t.data(), k, (double)i, l, &hb[(size_t)i * 6]);
            lb[i] = l - k;
            span[i] = l;
        }
        // samples whose k + 1 B-splines include coefficient j: lb[i] in [j - k, j]
        first.assign((size_t)nc(), m);
        last.assign((size_t)nc(), -1);
        for (int i = 0; i < m; i++)
            for (int e = 0; e <= k; e++) {
                first[lb[i] + e] = std::min(first[lb[i] + e], i);
                last[lb[i] + e] = std::max(last[lb[i] + e], i);
            }
    }
    // triangular band of the QR factor of [A; B / p] (Givens rotations, fpgivs / fprota), and
    // the scaled jump rows B / p themselves (fpdisc) for the refinement step on the device
    void factor(double p)
    {
        const int ncf = nc(), band = k + 2, nri = nrint();
        const int nb = (p > 0.0 && nri > 1) ? nri - 1 : 0;
        R.assign((size_t)ncf * pm::kSmBand, 0.0);
        Bp.assign((size_t)(nb > 0 ? nb : 1) * pm::kSmBand, 0.0);
        if (nb) {
            const double fac = (double)nri / (t[n - k - 1] - t[k]);
            for (int r = 0; r < nb; r++) {
                const int l = r + k + 1;
                for (int j = 0; j < band; j++) {
                    const int i = r + j;
                    double prod = 1.0;
                    bool first = true;
                    for (int q = 0; q < band; q++) {
                        if (i + q == l) continue;
                        const double h = t[l] - t[i + q];
                        prod = first ? h : prod * h * fac;
                        first = false;
                    }
                    Bp[(size_t)r * pm::kSmBand + j] = (t[i + k + 1] - t[i]) / prod / p;
                }
            }
        }
        for (int i = 0; i < m + nb; i++) {
            double h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int j0;
            if (i < m) {
                for (int e = 0; e <= k; e++) h[e] = hb[(size_t)i * 6 + e];
                j0 = lb[i];
            } else {
                const int r = i - m;
                for (int e = 0; e < band; e++) h[e] = Bp[(size_t)r * pm::kSmBand + e];
                j0 = r;
            }
            for (int j = j0; j < ncf; j++) {
                const double piv = h[0];
                if (piv != 0.0) {
                    double *Rj = &R[(size_t)j * pm::kSmBand];
                    const double ww = Rj[0], store = std::fabs(piv);
                    const double dd = (store >= ww) ? store * std::sqrt(1.0 + (ww / piv) * (ww / piv))
                                                    : ww * std::sqrt(1.0 + (piv / ww) * (piv / ww));
                    const double cs = ww / dd, sn = piv / dd;
                    Rj[0] = dd;
                    for (int b = 1; b < band; b++) {
                        const double s1 = h[b], s2 = Rj[b];
                        Rj[b] = cs * s2 + sn * s1;
                        h[b] = cs * s1 - sn * s2;
                    }
                }
                bool any = false;
                for (int b = 0; b < band - 1; b++) { h[b] = h[b + 1]; any |= (h[b] != 0.0); }
                h[band - 1] = 0.0;
                if (!any) break;
            }
        }
    }
    // per-interval residual sums from per-sample sums: a sample on a knot gives half to each side
    void account(const double *sums)
    {
        const int nri = nrint();
        for (int j = 0; j < nri; j++) fpint[j] = 0.0;
        int old = 0;
        for (int i = 0; i < m; i++) {
            const int num = lb[i];
            fpint[num] += sums[i];
            if (num != old) { fpint[num] -= 0.5 * sums[i]; fpint[num - 1] += 0.5 * sums[i]; }
            old = num;
        }
    }
};

// Device workspace of the fits of one (ny, nx) plane (one per worker, pm_ctx::SmWorker). The small
// per-fit tables form ONE contiguous block with a pinned host mirror of the same layout:
// a fit uploads them with a single asynchronous copy and reads its residual sums back with another.
struct SmTables {  // byte offsets inside the table block
    size_t hb_y, hb_x, R_y, R_x, Bp_y, Bp_x, t_y, t_x, lb_y, lb_x, fl_y, fl_x, span_y, span_x, sums, bytes;
};
struct SmDevice {
    double *U, *UT, *G, *CT, *RB;
    char *tables;       // device table block
    char *tables_host;  // pinned mirror
    SmTables o;
    template <typename T> T *dev(size_t off) const { return (T *)(tables + off); }
    template <typename T> T *host(size_t off) const { return (T *)(tables_host + off); }
};

int ensure_sm_arena(pm_ctx *ctx, pm_ctx::SmWorker &w, int ny, int nx, SmDevice &d)
{
    if (!w.stream) PM_HIP(ctx, hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
    const size_t npx = (size_t)ny * nx, mx = (size_t)std::max(ny, nx) + 8;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    SmTables &t = d.o;
    t.hb_y = take(mx * 6 * 8); t.hb_x = take(mx * 6 * 8);
    t.R_y = take(mx * pm::kSmBand * 8); t.R_x = take(mx * pm::kSmBand * 8);
    t.Bp_y = take(mx * pm::kSmBand * 8); t.Bp_x = take(mx * pm::kSmBand * 8);
    t.t_y = take(mx * 8); t.t_x = take(mx * 8);
    t.lb_y = take(mx * 4); t.lb_x = take(mx * 4);
    t.fl_y = take(mx * 8); t.fl_x = take(mx * 8);  // first[nc] then last[nc]
    t.span_y = take(mx * 4); t.span_x = take(mx * 4);
    t.bytes = off;               // ... everything above is uploaded per fit
    t.sums = take(2 * mx * 8);   // row sums then column sums, read back per fit
    const size_t table_bytes = off;
    const size_t oU = take(npx * 8), oUT = take(npx * 8), oG = take(npx * 8), oCT = take(npx * 8), oRB = take(npx * 8);
    if (off > w.arena_bytes) {
        PM_HIP(ctx, hipStreamSynchronize(w.stream));
        if (w.arena) PM_HIP(ctx, hipFree(w.arena));
        if (w.tables_host) PM_HIP(ctx, hipHostFree(w.tables_host));
        w.arena = nullptr;
        w.tables_host = nullptr;
        w.arena_bytes = 0;
        if (hipMalloc(&w.arena, off) != hipSuccess || hipHostMalloc(&w.tables_host, table_bytes) != hipSuccess)
            return fail(ctx, PM_ERR_ALLOC, "allocation of the smoothing-spline workspace (%zu bytes) failed", off);
        w.arena_bytes = off;
    }
    char *b = (char *)w.arena;
    d.tables = b;
    d.tables_host = (char *)w.tables_host;
    d.U = (double *)(b + oU); d.UT = (double *)(b + oUT); d.G = (double *)(b + oG); d.CT = (double *)(b + oCT);
    d.RB = (double *)(b + oRB);
    return PM_OK;
}

// one fit for the current knots and p (p <= 0: least-squares spline); returns fp and updates
// the per-interval residual sums of both axes. z: cleaned plane on the device.
int sm_fit(pm_ctx *ctx, hipStream_t s, const double *z, SmAxis &ay, SmAxis &ax, double p, const SmDevice &d, double &fp)
{
    const SmTables &t = d.o;
    auto put = [&](size_t off, const void *src, size_t bytes) { std::memcpy(d.tables_host + off, src, bytes); };
    // (the mirror is rewritten only after the previous fit's read-back has synchronised the stream)
    if (ay.knots_changed) {
        ay.build_tables();
        put(t.hb_y, ay.hb.data(), ay.hb.size() * 8);
        put(t.lb_y, ay.lb.data(), ay.lb.size() * 4);
        put(t.fl_y, ay.first.data(), ay.first.size() * 4);
        put(t.fl_y + (size_t)ay.nc() * 4, ay.last.data(), ay.last.size() * 4);
        put(t.t_y, ay.t.data(), (size_t)ay.n * 8);
        put(t.span_y, ay.span.data(), ay.span.size() * 4);
        ay.knots_changed = false;
    }
    if (ax.knots_changed) {
        ax.build_tables();
        put(t.hb_x, ax.hb.data(), ax.hb.size() * 8);
        put(t.lb_x, ax.lb.data(), ax.lb.size() * 4);
        put(t.fl_x, ax.first.data(), ax.first.size() * 4);
        put(t.fl_x + (size_t)ax.nc() * 4, ax.last.data(), ax.last.size() * 4);
        put(t.t_x, ax.t.data(), (size_t)ax.n * 8);
        put(t.span_x, ax.span.data(), ax.span.size() * 4);
        ax.knots_changed = false;
    }
    ay.factor(p);
    ax.factor(p);
    put(t.R_y, ay.R.data(), ay.R.size() * 8);
    put(t.R_x, ax.R.data(), ax.R.size() * 8);
    put(t.Bp_y, ay.Bp.data(), ay.Bp.size() * 8);
    put(t.Bp_x, ax.Bp.data(), ax.Bp.size() * 8);
    PM_HIP(ctx, hipMemcpyAsync(d.tables, d.tables_host, t.bytes, hipMemcpyHostToDevice, s));
    const int nby = (p > 0.0 && ay.nrint() > 1) ? ay.nrint() - 1 : 0, nbx = (p > 0.0 && ax.nrint() > 1) ? ax.nrint() - 1 : 0;
    pm::SmoothFitAxis fy = {d.dev<double>(t.hb_y), d.dev<int>(t.lb_y), d.dev<int>(t.fl_y), d.dev<int>(t.fl_y) + ay.nc(),
                            d.dev<double>(t.R_y), d.dev<double>(t.Bp_y), ay.m, ay.k, ay.nc(), nby};
    pm::SmoothFitAxis fx = {d.dev<double>(t.hb_x), d.dev<int>(t.lb_x), d.dev<int>(t.fl_x), d.dev<int>(t.fl_x) + ax.nc(),
                            d.dev<double>(t.R_x), d.dev<double>(t.Bp_x), ax.m, ax.k, ax.nc(), nbx};
    const int ny = ay.m, nx = ax.m, nr = ay.nc();
    double *rowsum = d.dev<double>(t.sums), *colsum = rowsum + ny;
    // along image rows for every image column: U (nr x nx)
    pm_launch_sm_solve(fy, z, (size_t)nx, 1, nx, d.G, d.U, d.RB, s);
    // U' (nx x nr), then along image columns for every row coefficient: CT (ncx x nr)
    pm_launch_transpose(d.U, d.UT, nr, nx, s);
    pm_launch_sm_solve(fx, d.UT, (size_t)nr, 1, nr, d.G, d.CT, d.RB, s);
    PM_HIP(ctx, hipMemsetAsync(rowsum, 0, (size_t)(ny + nx) * 8, s));
    pm_launch_sm_resid(fy, fx, z, d.CT, rowsum, colsum, s);
    PM_HIP(ctx, hipGetLastError());
    double *sums = d.host<double>(t.sums);
    PM_HIP(ctx, hipMemcpyAsync(sums, rowsum, (size_t)(ny + nx) * 8, hipMemcpyDeviceToHost, s));
    PM_HIP(ctx, hipStreamSynchronize(s));
    fp = 0.0;
    for (int i = 0; i < ny; i++) fp += sums[i];
    if (ctx->trace & 2)  // PM_OPT_TRACE: the knot / smoothing-parameter search
        std::fprintf(stderr, "sm_fit ny=%d nx=%d knots=(%d,%d) p=%g fp=%.17g\n", ny, nx, ay.n, ax.n, p, fp);
    ay.account(sums);
    ax.account(sums + ny);
    return PM_OK;
}

// FITPACK fpregr for one cleaned plane: on return ay / ax hold the knots and d.CT the coefficients
int sm_regrid(pm_ctx *ctx, hipStream_t stream, const double *z, int ny, int nx, int k_rows, int k_cols, double s,
              SmAxis &ay, SmAxis &ax, const SmDevice &d)
{
    const double tol = 0.001, con1 = 0.1, con9 = 0.9, con4 = 0.04;
    const int maxit = 20;
    const double acc = tol * s;
    ay.init(ny, k_rows);
    ax.init(nx, k_cols);
    const int nminy = 2 * (k_rows + 1), nminx = 2 * (k_cols + 1), nmaxy = ny + k_rows + 1, nmaxx = nx + k_cols + 1;
    int lastdi = 0, rc;
    bool poly = false, done = false;
    double fp = 0.0, fp0 = 0.0, fpold = 0.0, reducy = 0.0, reducx = 0.0, fpms = 0.0;
    // (FITPACK's "x" is the first array axis = image rows, "y" the image columns)
    for (int iter = 0; iter < ny + nx; iter++) {
        poly = (ay.n == nminy && ax.n == nminx);
        rc = sm_fit(ctx, stream, z, ay, ax, -1.0, d, fp);
        if (rc != PM_OK) return rc;
        if (poly) fp0 = fp;
        fpms = fp - s;
        if (std::fabs(fpms) < acc) { done = true; break; }
        if (fpms < 0.0) break;
        if (ay.n == nmaxy && ax.n == nmaxx) { done = true; break; }  // interpolating spline
        if (lastdi < 0) reducy = fpold - fp;
        else if (lastdi > 0) reducx = fpold - fp;
        fpold = fp;
        auto nplus = [&](const SmAxis &a, int nmin, double reduc) {
            if (a.n == nmin) return 1;
            int npl1 = a.nplus * 2;
            if (reduc > acc) npl1 = (int)((double)a.nplus * fpms / reduc);
            return std::min(a.nplus * 2, std::max(std::max(npl1, a.nplus / 2), 1));
        };
        const int nply = nplus(ay, nminy, reducy), nplx = nplus(ax, nminx, reducx);
        bool first_axis = (nply < nplx) || (nply == nplx && lastdi >= 0);
        if (first_axis && ay.n == nmaxy) first_axis = false;
        if (!first_axis && ax.n == nmaxx) first_axis = true;
        SmAxis &a = first_axis ? ay : ax;
        lastdi = first_axis ? -1 : 1;
        a.nplus = first_axis ? nply : nplx;
        const int nmax = first_axis ? nmaxy : nmaxx;
        for (int l = 0; l < a.nplus; l++) {
            a.add_knot();
            if (a.n == nmax) break;
        }
    }
    if (!done && !poly) {
        double p1 = 0.0, f1 = fp0 - s, p3 = -1.0, f3 = fpms, p = 1.0;
        bool ich1 = false, ich3 = false;
        for (int iter = 0; iter < maxit; iter++) {
            rc = sm_fit(ctx, stream, z, ay, ax, p, d, fp);
            if (rc != PM_OK) return rc;
            fpms = fp - s;
            if (std::fabs(fpms) < acc || iter == maxit - 1) break;
            const double p2 = p, f2 = fpms;
            if (!ich3) {
                if ((f2 - f3) <= acc) {  // initial p too large
                    p3 = p2; f3 = f2;
                    p *= con4;
                    if (p <= p1) p = p1 * con9 + p2 * con1;
                    continue;
                }
                if (f2 < 0.0) ich3 = true;
            }
            if (!ich1) {
                if ((f1 - f2) <= acc) {  // initial p too small
                    p1 = p2; f1 = f2;
                    p /= con4;
                    if (p3 >= 0.0 && p >= p3) p = p2 * con1 + p3 * con9;
                    continue;
                }
                if (f2 > 0.0) ich1 = true;
            }
            if (f2 >= f1 || f2 <= f3) break;
            if (p3 > 0.0) {  // fprati
                const double h1 = f1 * (f2 - f3), h2 = f2 * (f3 - f1), h3 = f3 * (f1 - f2);
                p = -(p1 * p2 * h3 + p2 * p3 * h1 + p3 * p1 * h2) / (p1 * h1 + p2 * h2 + p3 * h3);
            } else {
                p = (p1 * (f1 - f3) * f2 - p2 * (f2 - f3) * f1) / ((f1 - f2) * f3);
            }
            if (f2 < 0.0) { p3 = p2; f3 = f2; } else { p1 = p2; f1 = f2; }
        }
    }
    return PM_OK;
}

// smoothing-spline reprojection of planes resident on the device
int reproject_smoothing_resident(pm_ctx *ctx, pm::ReprojectArgs a, int dtype, int k_rows, int k_cols, double s)
{
    const size_t plane_elems = (size_t)a.ny * a.nx;
    size_t chunk = (size_t)(1ull << 30) / (plane_elems * sizeof(double));
    chunk = std::max<size_t>(1, std::min<size_t>(chunk, (size_t)a.n_planes));
    if (chunk > 32768) chunk = 32768;
    int rc = ensure_work(ctx, chunk * plane_elems * sizeof(double));
    if (rc != PM_OK) return rc;
    rc = ensure_stats(ctx, chunk);
    if (rc != PM_OK) return rc;
    // The fit of a plane is a host-driven search (a few dozen fits, each a chain of small launches
    // and a read-back the next decision waits for): planes are dealt to worker threads, each with
    // its own stream and workspace, so that several such chains are in flight.
    const int want = std::max(1, std::min(ctx->sm_worker_count, (int)pm_ctx::kSmWorkers));  // PM_OPT_SM_WORKERS
    const int n_workers = (int)std::min<size_t>((size_t)want, std::min<size_t>(chunk, (size_t)a.n_planes));
    std::vector<SmDevice> devs(n_workers);
    for (int w = 0; w < n_workers; w++) {
        rc = ensure_sm_arena(ctx, ctx->sm_workers[w], a.ny, a.nx, devs[w]);
        if (rc != PM_OK) return rc;
    }
    std::vector<pm::PlaneStats> stats(chunk);
    const std::vector<double> nan_row((size_t)a.n_map, std::nan(""));
    for (size_t p0 = 0; p0 < (size_t)a.n_planes; p0 += chunk) {
        const int np = (int)std::min(chunk, (size_t)a.n_planes - p0);
        pm::ReprojectArgs b = a;
        b.n_planes = np;
        b.cube = (const char *)a.cube + p0 * plane_elems * dtype_size(dtype);
        b.out = a.out + p0 * a.n_map;
        b.plane_stats = ctx->stats;
        PM_HIP(ctx, hipMemsetAsync(ctx->stats, 0, (size_t)np * sizeof(pm::PlaneStats), ctx->stream));
        PM_HIP(ctx, hipMemsetAsync(ctx->hist, 0, (size_t)np * 512 * sizeof(unsigned int), ctx->stream));
        pm_launch_clean_lazy(b, ctx->work, dtype, ctx->stats, ctx->hist, ctx->stream);
        PM_HIP(ctx, hipMemcpyAsync(stats.data(), ctx->stats, (size_t)np * sizeof(pm::PlaneStats), hipMemcpyDeviceToHost,
                                   ctx->stream));
        PM_HIP(ctx, hipStreamSynchronize(ctx->stream));  // cleaned planes + statistics are complete
        std::atomic<int> next{0};
        std::vector<int> rcs(n_workers, PM_OK);
        auto worker = [&](int w) -> int {
            PM_HIP(ctx, hipSetDevice(ctx->device));
            const hipStream_t st = ctx->sm_workers[w].stream;
            const SmDevice &d = devs[w];
            SmAxis ay, ax;
            for (int pl = next.fetch_add(1); pl < np; pl = next.fetch_add(1)) {
                if (stats[pl].all_nan) {  // body_xy.py:1668-1670: the map of an all-NaN image is all NaN
                    PM_HIP(ctx, hipMemcpyAsync(b.out + (size_t)pl * a.n_map, nan_row.data(), (size_t)a.n_map * 8,
                                               hipMemcpyHostToDevice, st));
                    PM_HIP(ctx, hipStreamSynchronize(st));
                    continue;
                }
                const double *z = ctx->work + (size_t)pl * plane_elems;
                const int r = sm_regrid(ctx, st, z, a.ny, a.nx, k_rows, k_cols, s, ay, ax, d);
                if (r != PM_OK) return r;
                // (knots and spans of the final fit are already in the device table block)
                pm::SmoothEvalArgs e = {d.CT, d.dev<double>(d.o.t_y), d.dev<double>(d.o.t_x), d.dev<int>(d.o.span_y),
                                        d.dev<int>(d.o.span_x), ay.nc(), ax.nc(), k_rows, k_cols, pl};
                pm_launch_sm_eval(b, e, dtype, st);
                PM_HIP(ctx, hipGetLastError());
                PM_HIP(ctx, hipStreamSynchronize(st));  // knots / spans are reused by the next plane
            }
            return PM_OK;
        };
        const int active = std::min(n_workers, np);
        if (active <= 1) {
            rcs[0] = worker(0);
        } else {
            std::vector<std::thread> threads;
            for (int w = 0; w < active; w++) threads.emplace_back([&, w] { rcs[w] = worker(w); });
            for (auto &t : threads) t.join();
        }
        for (int w = 0; w < active; w++)
            if (rcs[w] != PM_OK) return rcs[w];
    }
    return PM_OK;
}

// Spline reprojection of planes resident on the device (plane chunks bound the workspace).
int reproject_spline_resident(pm_ctx *ctx, pm::ReprojectArgs a, int dtype, int k_rows, int k_cols)
{
    const size_t plane_elems = (size_t)a.ny * a.nx;
    pm::SplineArgs sa;
    int rc = ensure_axis(ctx, 0, a.ny, k_rows, sa.rows);
    if (rc != PM_OK) return rc;
    rc = ensure_axis(ctx, 1, a.nx, k_cols, sa.cols);
    if (rc != PM_OK) return rc;
    size_t chunk = (size_t)(2ull << 30) / (plane_elems * sizeof(double));
    if (chunk < 1) chunk = 1;
    if (chunk > (size_t)a.n_planes) chunk = (size_t)a.n_planes;
    if (chunk > 32768) chunk = 32768;
    rc = ensure_work(ctx, chunk * plane_elems * sizeof(double));
    if (rc != PM_OK) return rc;
    rc = ensure_stats(ctx, chunk);
    if (rc != PM_OK) return rc;
    sa.work = ctx->work;
    for (size_t p0 = 0; p0 < (size_t)a.n_planes; p0 += chunk) {
        const int np = (int)std::min(chunk, (size_t)a.n_planes - p0);
        pm::ReprojectArgs b = a;
        b.n_planes = np;
        b.cube = (const char *)a.cube + p0 * plane_elems * dtype_size(dtype);
        b.out = a.out + p0 * a.n_map;
        b.plane_stats = ctx->stats;
        PM_HIP(ctx, hipMemsetAsync(ctx->stats, 0, (size_t)np * sizeof(pm::PlaneStats), ctx->stream));
        PM_HIP(ctx, hipMemsetAsync(ctx->hist, 0, (size_t)np * 512 * sizeof(unsigned int), ctx->stream));
        pm_launch_spline(b, sa, dtype, ctx->stats, ctx->hist, ctx->stream);
    }
    PM_HIP(ctx, hipGetLastError());
    return PM_OK;
}

// Reproject `a.n_planes` planes that are resident on the device. First pass without plane
// statistics; if a plane reports that it needs its nanmedian (flags bit 1), the medians are
// computed and the planes are mapped again. `sync_now`: examine the flags immediately
// (host-buffer callers) or leave that to pm_synchronize().
int reproject_resident(pm_ctx *ctx, pm::ReprojectArgs a, int dtype, bool sync_now);

int finish_reproject(pm_ctx *ctx, const pm::ReprojectArgs &a, int dtype)
{
    std::vector<int> hflags(a.n_planes);
    PM_HIP(ctx, hipMemcpyAsync(hflags.data(), a.plane_flags, (size_t)a.n_planes * sizeof(int), hipMemcpyDeviceToHost,
                               ctx->stream));
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    bool any = false, stale = false;
    int n_redo = 0;
    for (int f : hflags) {
        any = any || (f == a.seq);
        n_redo += (f == a.seq);
        stale = stale || (f > ctx->checked_seq && f < a.seq);
    }
    ctx->checked_seq = a.seq;
    ctx->last_redo_planes = n_redo;
    if (stale)
        return fail(ctx, PM_ERR_STATE,
                    "an earlier asynchronous pm_map_cube call sampled pixels that need the plane nanmedian (+-inf "
                    "or all-NaN neighbourhoods) and was superseded before pm_synchronize(); synchronize after each "
                    "call for such data");
    if (!any) return PM_OK;
    int rc = ensure_stats(ctx, (size_t)a.n_planes);
    if (rc != PM_OK) return rc;
    size_t plane_elems = (size_t)a.ny * a.nx;
    PM_HIP(ctx, hipMemsetAsync(ctx->stats, 0, (size_t)a.n_planes * sizeof(pm::PlaneStats), ctx->stream));
    PM_HIP(ctx, hipMemsetAsync(ctx->hist, 0, (size_t)a.n_planes * 512 * sizeof(unsigned int), ctx->stream));
    for (int p0 = 0; p0 < a.n_planes; p0 += 32768) {
        int np = std::min(32768, a.n_planes - p0);
        pm_launch_plane_medians((const char *)a.cube + (size_t)p0 * plane_elems * dtype_size(dtype), dtype, np, plane_elems,
                                ctx->stats + p0, ctx->hist + (size_t)p0 * 512, ctx->stream);
        pm::ReprojectArgs b = a;
        b.n_planes = np;
        b.cube = (const char *)a.cube + (size_t)p0 * plane_elems * dtype_size(dtype);
        b.out = a.out + (size_t)p0 * a.n_map;
        b.plane_flags = a.plane_flags + p0;
        b.plane_stats = ctx->stats + p0;
        pm_launch_reproject(b, dtype, ctx->stream);
    }
    PM_HIP(ctx, hipGetLastError());
    PM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PM_OK;
}

int reproject_resident(pm_ctx *ctx, pm::ReprojectArgs a, int dtype, bool sync_now)
{
    size_t plane_elems = (size_t)a.ny * a.nx;
    a.plane_stats = nullptr;
    a.seq = ++ctx->map_seq;
    // blockIdx.y is limited to 65535 planes per launch
    for (int p0 = 0; p0 < a.n_planes; p0 += 32768) {
        pm::ReprojectArgs b = a;
        b.n_planes = std::min(32768, a.n_planes - p0);
        b.cube = (const char *)a.cube + (size_t)p0 * plane_elems * dtype_size(dtype);
        b.out = a.out + (size_t)p0 * a.n_map;
        b.plane_flags = a.plane_flags + p0;
        pm_launch_reproject(b, dtype, ctx->stream);
    }
    PM_HIP(ctx, hipGetLastError());
    if (ctx->chunk_cb) ctx->chunk_cb(ctx->chunk_user, 0, a.n_planes);
    if (sync_now) return finish_reproject(ctx, a, dtype);
    ctx->pending = true;
    ctx->pending_args = a;
    ctx->pending_dtype = dtype;
    return PM_OK;
}

}  // namespace pmh
