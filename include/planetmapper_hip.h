/*
 * planetmapper_hip.h -- C ABI of libplanetmapper_hip.so
 *
 * MI355X (gfx950) engine for the per-pixel hot path of ortk95/planetmapper:
 * backplane image generation and map reprojection. The reference has no FFI of
 * its own for this path (it is pure Python calling CSPICE through spiceypy once
 * per pixel); each entry point below therefore names the reference *Python*
 * interface it replaces (file:line relative to the reference checkout). The
 * Python package `planetmapper_amd` binds these symbols with ctypes and presents
 * the reference's BodyXY / Observation method names on top.
 *
 * Conventions
 *   - plain C types only; no C++ exceptions cross the boundary;
 *   - every function returns PM_OK (0) or a negative pm_status; the message of the
 *     most recent failure on a context is available from pm_last_error();
 *   - images are row-major double[ny][nx], pixel (0,0) bottom-left, integer
 *     coordinates are pixel centres (reference: planetmapper/__init__.py:19-23);
 *     "not on the disc / not visible" is a quiet NaN;
 *   - buffers are caller-owned. A pointer is a HOST pointer unless the call takes
 *     a `mem` argument equal to PM_MEM_DEVICE, in which case it is a device (HBM)
 *     pointer valid on the context's GPU;
 *   - a context is bound to one GPU and one HIP stream; calls on one context must
 *     not be made concurrently from several threads. Calls are synchronous for
 *     host buffers; for device buffers work is enqueued on the context stream and
 *     pm_synchronize() waits for it.
 */
#ifndef PLANETMAPPER_HIP_H
#define PLANETMAPPER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_ABI_VERSION 3 /* 2: pm_geometry.DVT, DAT; 3: pm_geometry.WP, pm_dlpack_*, pm_host_free(NULL, ...) */

typedef enum pm_status {
    PM_OK = 0,
    PM_ERR_INVALID_ARGUMENT = -1, /* ValueError in the reference */
    PM_ERR_NO_DEVICE = -2,        /* no usable gfx950 device / HIP runtime failure */
    PM_ERR_HIP = -3,              /* a HIP call failed; see pm_last_error */
    PM_ERR_STATE = -4,            /* geometry or disc not set yet */
    PM_ERR_ALLOC = -5,
    PM_ERR_UNSUPPORTED = -6,      /* valid in the reference, not implemented here yet */
    PM_ERR_PEER = -7              /* pm_map_cube_sharded: another rank failed; the gathered result is not valid */
} pm_status;

/* PM_MEM_HOST_CUBE (pm_map_cube only): `cube` is a HOST pointer, x_map / y_map / out are DEVICE
 * pointers - the planes are fed from host memory (pinned: gathered in place, asynchronous like a
 * PM_MEM_DEVICE call; pageable: pipelined copy, synchronous) and the mapped planes stay in HBM,
 * e.g. as this rank's slot of an RCCL all-gather. nearest / linear with NaN propagation. */
typedef enum pm_mem { PM_MEM_HOST = 0, PM_MEM_DEVICE = 1, PM_MEM_HOST_CUBE = 2 } pm_mem;

/*
 * Backplane identifiers: bit positions of `plane_mask`, in the order of the
 * reference's default registry (planetmapper/body_xy.py:4198-4356).
 */
typedef enum pm_plane {
    PM_LON_GRAPHIC = 0,      /* body_xy.py:3302 get_lon_img            [deg] */
    PM_LAT_GRAPHIC = 1,      /* body_xy.py:3324 get_lat_img            [deg] */
    PM_LON_CENTRIC = 2,      /* body_xy.py:3346 _get_lonlat_centric_img [deg] */
    PM_LAT_CENTRIC = 3,
    PM_RA = 4,               /* body_xy.py:3409 _get_radec_img         [deg] */
    PM_DEC = 5,
    PM_PIXEL_X = 6,          /* body_xy.py:3494 get_x_img              [px]  */
    PM_PIXEL_Y = 7,
    PM_KM_X = 8,             /* body_xy.py:3545 _get_km_xy_img         [km]  */
    PM_KM_Y = 9,
    PM_ANGULAR_X = 10,       /* body_xy.py:3610 get_angular_x_img   [arcsec] */
    PM_ANGULAR_Y = 11,
    PM_PHASE = 12,           /* body_xy.py:3658 _get_illumination_gie_img [deg] */
    PM_INCIDENCE = 13,
    PM_EMISSION = 14,
    PM_AZIMUTH = 15,         /* body_xy.py:3742 get_azimuth_angle_img  [deg] */
    PM_LOCAL_SOLAR_TIME = 16,/* body_xy.py:3787 get_local_solar_time_img [h] */
    PM_DISTANCE = 17,        /* body_xy.py:3869 get_distance_img       [km]  */
    PM_RADIAL_VELOCITY = 18, /* body_xy.py:3895 get_radial_velocity_img [km/s] */
    PM_DOPPLER = 19,         /* body_xy.py:3938 get_doppler_img              */
    PM_LIMB_DISTANCE = 20,   /* body_xy.py:3964 _get_limb_coordinate_imgs [km] */
    PM_LIMB_LON_GRAPHIC = 21,
    PM_LIMB_LAT_GRAPHIC = 22,
    PM_RING_RADIUS = 23,     /* body_xy.py:4059 _get_ring_plane_coordinate_imgs [km] */
    PM_RING_LON_GRAPHIC = 24,
    PM_RING_DISTANCE = 25,
    PM_NUM_PLANES = 26
} pm_plane;

#define PM_PLANE_BIT(p) (((uint64_t)1) << (p))

/*
 * Geometry block: everything the per-pixel kernels need about one Body at one
 * epoch, computed ONCE on the host (reference: SPICE calls in
 * planetmapper/base.py:795-839 and planetmapper/body.py:501-606). All vectors
 * are km (km/s, km/s^2) in J2000 unless stated; matrices are row-major.
 *
 * Light-time corrected quantities are reconstructed on the device as
 *   T(t)  = T0 + VT (t - t0) + AT (t - t0)^2 / 2              t0 = et - lt_c
 *   R(t)  = Rz_frame(wdot (t - t0)) R0                        (J2000 -> body-fixed)
 *   S(t)  = S0 + VS (t - ts0) + AS (t - ts0)^2 / 2            (Sun wrt P_T(t0))
 * which is exact to < 1e-9 km / 1e-13 rad over the <= 1 s (disc) to 1e3 s (ring
 * plane) spans the path needs.
 */
typedef struct pm_geometry {
    double et;          /* observation epoch, TDB seconds past J2000 (BodyBase.et) */
    double lt_c;        /* one-way light time to target centre (BodyBase.target_light_time) */
    double clight;      /* km/s, spice.clight() */
    double radii[3];    /* Body.radii (a, b, c) WITHOUT altitude adjustment */

    double T0[3];       /* Body._target_obsvec: target centre at t0 wrt observer at et */
    double VT[3];       /* d/dt of the SSB position of the target centre at t0 (see DVT) */
    double AT[3];       /* d2/dt2 of it */
    double VO[3];       /* SSB velocity of observer at et (radial velocity only) */
    double DVT[3];      /* what a STATE of the target centre at t0 (spkezr: Body._state_from_targvec,
                           body.py:2830-2845) adds to VT. VT / AT are the derivatives of the target's
                           POSITION with respect to time - what re-evaluating the ephemeris at et - lt, as
                           CSPICE does at every light-time pass, amounts to; a segment that carries its
                           own velocity polynomial (SPK type 3: Jupiter wrt its barycentre in jup120)
                           gives a state velocity 2e-6 km/s away from that. Radial velocity only; 0 for
                           type 2 chains. */
    double DAT[3];      /* the same for the acceleration: d/dt of the state velocity minus AT (7e-9 km/s^2) */

    double ts0;         /* epoch at which S0/VS/AS are evaluated (~ t0 - Sun light time) */
    double S0[3];       /* P_sun(ts0) - P_T(t0) */
    double VS[3];       /* d/dt of the SSB position of the Sun at ts0 */
    double AS[3];       /* d2/dt2 of it */

    double R0[9];       /* pxform(J2000 -> target frame) at t0 */
    double wdot;        /* spin rate about body +z, rad/s (dW/dt of the IAU model) */

    double sub_sp[3];     /* Body._subpoint_targvec (body-fixed)          body.py:538 */
    double sub_ray[3];    /* Body._subpoint_rayvec  (body-fixed, obs->sp) body.py:538 */
    double sub_obsvec[3]; /* Body._subpoint_obsvec  (J2000)               body.py:551 */
    double sub_et;        /* Body._subpoint_et */
    double sub_dist;      /* Body.subpoint_distance */

    double ring_n[3];   /* unit normal of Body._ring_plane (J2000)        body.py:583 */
    double ring_k;      /* plane constant (>= 0), n . X = k */

    double M[9];        /* Body._get_obsvec2angular_matrix()              body.py:1317 */
    double diameter_arcsec; /* Body.target_diameter_arcsec                body.py:574 */
    double km_per_arcsec;   /* Body.km_per_arcsec                         body.py:577 */
    double np_angle_rad;    /* Body.north_pole_angle() in radians         body.py:2985 */
    double lst_sun_lon;     /* planetocentric east longitude (rad) of the Sun in the
                               target frame at t0, LT+S (et2lst)           body.py:2364 */
    double WP[3];       /* the part of the target frame's angular velocity that is NOT about its own z axis (J2000
                           components, rad/s): the drift of the pole - what the derivative block of sxform holds beyond
                           wdot. It enters the velocity of a surface point, i.e. the STATE planes only (spkcpt:
                           Body._state_from_targvec, body.py:2830-2845): Jupiter 7e-14 rad/s = 1e-9 km/s across the
                           disc in RADIAL-VELOCITY - the one figure the golden plane shows once the observer's
                           velocity is the ephemeris's own and not a fit. Positions ignore it (1e-17 rad over a disc's
                           light-time span). */

    int32_t west_positive;  /* 1 if Body.positive_longitude_direction == 'W' body.py:528 */
    int32_t reserved;
} pm_geometry;

/* Disc parameters + image size (reference: BodyXY.set_disc_params body_xy.py:700,
 * set_img_size body_xy.py:941, `optimize_speed` base.py:229). */
typedef struct pm_disc {
    double x0, y0, r0;      /* pixels */
    double rotation_rad;    /* BodyXY._rotation_radians, already reduced mod 2 pi */
    int32_t nx, ny;
    int32_t optimize_speed; /* enables the radius pre-mask of body_xy.py:3201-3218. (Either way the library also skips
                             * pixels whose line of sight passes the body further out than its limb can reach - a
                             * conservative bound from the geometry block; they are NaN in the reference with or without
                             * the pre-mask, so the planes do not change.) */
    int32_t reserved;
} pm_disc;

typedef enum pm_interpolation {
    PM_INTERP_NEAREST = 0, /* body_xy.py:1633 _do_nearest_interpolation */
    PM_INTERP_LINEAR = 1,  /* body_xy.py:1651 _do_spline_interpolation, kx=ky=1, s=0 */
    /* 'smooth': PCHIP resampling onto an oversampled grid (rows, then columns; non-finite
     * pixels are bridged) followed by bilinear sampling, body_xy.py:1704-1853. Options:
     * pm_set_smooth_options(). */
    PM_INTERP_SMOOTH = 2,
    /* RectBivariateSpline(kx=k_rows, ky=k_cols, s=0) interpolating splines ('quadratic' =
     * (2, 2), 'cubic' = (3, 3), mixed degrees), body_xy.py:1651-1702:
     * PM_INTERP_SPLINE(k_rows, k_cols); k_rows is the degree along image rows (axis 0),
     * which the reference calls kx. Degrees 1..5. */
    PM_INTERP_SPLINE_FLAG = 0x100
} pm_interpolation;
#define PM_INTERP_SPLINE(k_rows, k_cols) (PM_INTERP_SPLINE_FLAG | ((k_rows) << 4) | (k_cols))

typedef enum pm_dtype {
    PM_F64 = 0, PM_F32 = 1, PM_I16 = 2, PM_I32 = 3, PM_U8 = 4, PM_U16 = 5
} pm_dtype;

/* Coordinate systems of pm_transform (reference: "coordinate systems" of
 * planetmapper/__init__.py:12-60). */
typedef enum pm_coord {
    PM_COORD_XY = 0,      /* image pixels                                      */
    PM_COORD_RADEC = 1,   /* observer RA/Dec [deg]                             */
    PM_COORD_ANGULAR = 2, /* arcsec from the target centre, East / North        */
    PM_COORD_KM = 3,      /* km in the target plane, North pole up             */
    PM_COORD_LONLAT = 4   /* planetographic lon/lat [deg]                      */
} pm_coord;
#define PM_TF_NOT_VISIBLE_NAN 1 /* from lon/lat: NaN when hidden from the observer */
#define PM_TF_PLANETOCENTRIC 2  /* lon/lat are planetocentric                      */

typedef struct pm_ctx pm_ctx; /* opaque */

/* Library / device ---------------------------------------------------------- */
int pm_abi_version(void);
/* Number of usable gfx950 devices (0 if none); never fails. */
int pm_device_count(void);
/* Create a context on GPU `device`. Fails (NULL + *status) if there is no GPU:
 * this library has no CPU fallback.
 * Environment: the library reads PM_RCCL_LIBRARY (pm_comm_*: the collective library to bind) and the launcher's
 * LOCAL_WORLD_SIZE (copy threads per rank). The debug / A-B knobs of tools/ - PM_FORCE_GENERAL, PM_FUSE_PLANES,
 * PM_LT_MODE, PM_HOSTPIPE_TRACE, PM_SM_DEBUG, PM_SM_BATCH_PLANES - set the DEFAULTS of the matching pm_set_option()
 * options at pm_create, and ONLY when PM_DEBUG_ENV=1 is set beside them: without it they are ignored, so a stray
 * variable in a user's shell cannot select another algorithm. */
pm_ctx *pm_create(int device, int *status);
void pm_destroy(pm_ctx *ctx);
const char *pm_last_error(const pm_ctx *ctx);
int pm_synchronize(pm_ctx *ctx);
/* For PM_MEM_DEVICE calls, errors detected on the device (pm_map_cube) are reported
 * by the next pm_synchronize(). */
/* HIP stream the context launches on (as a void*), for event timing by callers. */
void *pm_stream(pm_ctx *ctx);
/* Launch on a caller-owned hipStream_t instead (e.g. torch.cuda.current_stream()). */
int pm_set_stream(pm_ctx *ctx, void *hip_stream);

/*
 * Per-context options (no reference counterpart except where noted).
 *   PM_OPT_GENERAL_KERNEL   1: the image planes always go through the general kernel (arbitrary
 *                           rotations, full quadratic motion model, observer anywhere outside the
 *                           body) instead of the spheroid fast path the library selects when its
 *                           guards hold; the MAP planes (pm_backplanes_map, pm_xy_map) go through
 *                           the J2000 kernel with the general helpers instead of the B0 kernels;
 *                           'smooth' reprojection takes the gap-aware PCHIP form at every cell.
 *                           Results agree within the parity bars; used by the tests to cover both
 *                           sets of kernels. Default 0 (with PM_DEBUG_ENV=1 the environment variable
 *                           PM_FORCE_GENERAL=1 sets the default to 1 at pm_create).
 *   PM_OPT_HOST_CHUNK_BYTES bytes per stage of the pipelined host path (PM_MEM_HOST calls move
 *                           data through pinned staging buffers in chunks of this size so that
 *                           H2D copies, kernels and D2H copies of consecutive chunks overlap).
 *                           Default 32 MiB; clamped to [1 MiB, 1 GiB].
 *   PM_OPT_HOST_COPY_THREADS CPU threads that move pageable caller memory to / from the pinned
 *                           staging buffers. Default min(16, cores available to the process,
 *                           divided by LOCAL_WORLD_SIZE when a launcher set it); 1..64, 0 = default.
 *   PM_OPT_HOST_CUBE_ROUTE  (= PM_OPT_ZERO_COPY, its name when there were two routes)
 *                           how the host cube of pm_map_cube (nearest / linear) crosses PCIe:
 *                           0 = whole planes by DMA;
 *                           1 = a PINNED cube (pm_host_alloc / pm_host_register) is gathered in
 *                               place by the reprojection kernel;
 *                           2 = the 128-byte blocks (PM_OPT_FETCH_BLOCK_BYTES) of a plane that the map samples (the same in
 *                               every plane; found once per call by the sampling code itself) are
 *                               fetched by the GPU from a PINNED cube, once each, into a table in
 *                               HBM and sampled from there;
 *                           3 = the same table in 16-byte blocks, collected by the copy threads
 *                               into pinned staging and sent by DMA - any host memory; the link
 *                               carries little more than the sampled pixels;
 *                           4 = hybrid of 2 and 3 for a PINNED cube and a rank short of CPU threads:
 *                               short chunks of planes, alternately fetched by the GPU and collected by
 *                               the threads, so that the threads collect chunk k + 1 while the GPU's
 *                               reads of chunk k cross the link; the share of fetched planes is the one
 *                               at which both legs take the same time by the measured rates (one half
 *                               when forced before anything was measured);
 *                          -1 (default) = the library chooses, per context (i.e. per rank of a sharded
 *                               cube), the route that MEASURED fastest on the problem at hand: the first
 *                               call with enough planes feeds three short chunks through each candidate
 *                               (3, 0, and 2 for a pinned cube), times them, and maps the rest - and
 *                               every later call on the same plane size / map / memory kind / thread
 *                               count - by the fastest, or by 4 where the measured rates of 2 and 3
 *                               predict a hybrid 7 % or more faster than either (kept only while whole
 *                               calls confirm it; results are bit-identical whatever the route).
 *                               Until then: 3 when the table is under 40 % of the size of the planes,
 *                               else 0.
 *                           (1 and 2 on pageable memory, 2 and 3 on planes that are not a whole
 *                           number of blocks: as 0; 4 on pageable memory: as 3.)
 *   PM_OPT_SPARSE_FRAME     image planes into host memory (pm_backplanes_img, PM_MEM_HOST): the planes of
 *                           the disc are NaN outside the radius pre-mask of optimize_speed; 1 = only
 *                           bands of rows around that circle cross PCIe (as rectangles) and the copy
 *                           threads write the NaN around them; 0 = whole planes; -1 (default) = 1 for
 *                           planes of 64 MiB and more of which the circle covers under 85 % (smaller
 *                           planes gain nothing: measured).
 *   PM_OPT_BLOCK_TABLE_CACHE 1 (default): the block table of routes 2 / 3 (which blocks of a plane the map
 *                           samples, their numbering, the list the copy threads walk) is kept in the
 *                           context and reused by later calls with the same x/y map (compared by a 128-bit
 *                           fingerprint computed on the GPU), plane shape, element size and sampling mode -
 *                           as the reference keeps its x/y map between planes (body_xy.py:3478). 0: rebuilt
 *                           by every call.
 *   PM_OPT_BLOCK_TABLE_HITS read-only: calls that reused the cached table since the context was created.
 *   PM_OPT_ROUTE_EXPLORE    1 (default): measure the candidate routes as described under
 *                           PM_OPT_HOST_CUBE_ROUTE -1; 0: choose by table size only. Setting it (to either
 *                           value) forgets what has been measured.
 *   PM_OPT_LAST_CUBE_ROUTE  read-only: the route (0..4) the latest host-cube call ended on, -1 none yet.
 *   PM_OPT_FETCH_BLOCK_BYTES 128 (default), 64 or 256: the size of the blocks of routes 2 and 4 (the GPU's own reads
 *                           of a pinned cube). 128 bytes is the granularity of PCIe reads: at BASELINE config 5 it
 *                           moves 4.2 MB per plane instead of the 4.7 MB of 256-byte blocks and measured 8 % faster
 *                           (64 planes: 5.46 against 5.95 ms; 64-byte blocks 5.60 ms, profiles/r04_route_ab.jsonl).
 *   PM_OPT_LT_MODE          light time of the spheroid image kernel (A/B and the light-time test; none changes a result
 *                           beyond the parity bars): 0 (default) closed form where the limb is clear, 1 the
 *                           reference's own sequence of epochs for every pixel, 2 that sequence shortened by a Newton
 *                           step on its seed.
 *   PM_OPT_TRACE            mask, on stderr: 1 stage times of every host-path call, 2 the knot / smoothing-parameter
 *                           search of the smoothing splines (with 1 as well: the knots of the first planes). Default 0.
 *   PM_OPT_TRACE            ... 4: the device-side sums of PM_OPT_LAST_STAGE_NS are collected (a pair of timing events per chunk; a
 *                           sharded call waits for its exchanges before the agreement, so that the two are timed apart).
 *   PM_OPT_SM_BATCH_PLANES  the most planes of a cube whose smoothing-spline fits (spline_smoothing > 0) advance together
 *                           in the same launches: 0 (default) = as many as half of the free device memory holds (a plane
 *                           takes 7 float64 arrays of its own size), 1 .. 4096 = a cap. A plane's result does not depend on it.
 *   PM_OPT_SPLINE_SEGMENT   the interpolating-spline solves of pm_map_cube (degrees 1-5, spline_smoothing = 0) run one
 *                           lane per image line - a chain of 2 n dependent steps per axis, which leaves most of the
 *                           chip idle when the planes of a call are few and large. 0 (default): such calls (fewer
 *                           waves than SIMDs) cut every line into segments whose substitutions start a few dozen samples
 *                           early (the factors' recursion forgets geometrically: the same steps on the same operands, a
 *                           second workspace of the planes' size); -1: never; n >= 64: segments of n samples (rounded up
 *                           to 16) for every call. A/B and the test that holds the two forms to each other.
 *   PM_OPT_LAST_SPLINE_SEGMENT read-only: the axis-0 segment length of the latest spline call, 0 = one lane per line.
 *   PM_OPT_LAST_SM_KNIFE_EDGES read-only: planes of the latest smoothing-spline call (spline_smoothing > 0) whose knot
 *                           search chose between knot intervals whose residual shares are equal in exact arithmetic and
 *                           ordered only by the rounding of FITPACK's `fpmax * an / am` products - a choice scipy itself
 *                           makes by the last bit of a sum of ~10^3 squared residuals (a one-ulp change of one input pixel
 *                           flips it; the two outcomes are both smoothing splines FITPACK accepts, percent-level apart).
 *                           0 = every decision of the search was the reference's own beyond rounding.
 *   PM_OPT_LAST_SM_ILL_CONDITIONED read-only: planes of the latest smoothing-spline call whose search went through a fit beyond
 *                           what the library's least-squares solve (semi-normal equations + one refinement step: the
 *                           condition number squared, where FITPACK's Givens QR carries it once) resolves: the refinement
 *                           still moved the coefficients by more than 1e-10 of their scale (healthy fits: 1e-15 .. 2e-12).
 *                           Seen on 20-25 sample axes of degree 4-5 - smoothing parameters p ~ 1e7 .. 1e9, or a rank-deficient
 *                           knot set in the least-squares phase (3 fits in 191 000 of a fuzz soak): the map then differs from
 *                           scipy's by up to 1e-4 of the data scale (scipy itself moved by 1e-1 between versions on such
 *                           fits: tests/test_observation.py:1163-1170 of the reference). 0 = every fit converged.
 *   PM_OPT_LAST_STAGE_NS + k read-only, ns: where the latest host-fed pm_map_cube (PM_MEM_HOST / PM_MEM_HOST_CUBE, nearest /
 *                           linear) or pm_map_cube_sharded of this context spent its time. Host clock, always recorded:
 *                           0 the whole pm_map_cube call = 1 + 2 + 5 + 6 + 7; 1 fingerprint of the x / y maps + block-table
 *                           look-up or build; 2 route plan, layout, scratch; 5 from the first chunk to the last one issued
 *                           (inside it: 3 until the first byte of the segment is handed to the link - the first chunk's
 *                           collection, the link idle - and 4 the calling thread inside the copy threads' collections);
 *                           6 from the last chunk issued until the streams are idle (last DMA, last kernels); 7 flag
 *                           read-back and nanmedian replay. Device clock, with PM_OPT_TRACE bit 4 only (else 0): 8 the sum of
 *                           the chunks' H2D copies, 9 the sum of their kernels. pm_map_cube_sharded adds 10 what was left of
 *                           the exchanges once the mapping had finished (bit 4 only), 11 the agreement (with bit 4 off: the
 *                           rest of the exchanges + the agreement), 12 the whole sharded call.
 *   PM_OPT_HYBRID_FETCH_PERMILLE read-only: the share of planes (in 1/1000) a hybrid segment (route 4) has the GPU
 *                           fetch on the current problem; 0 while no hybrid has been planned.
 *   PM_OPT_HOST_COPY_THREADS_IN_USE read-only: the copy threads the host pipe of this context runs with (0 before
 *                           its first host-buffer call).
 *   PM_OPT_FUSE_PLANES      1: an image-plane request that holds planes of the intercept AND planes every pixel
 *                           has (RA / Dec, pixel x / y, km, angular, limb) runs as ONE launch on the spheroid
 *                           fast path (k_disc_sph<FLAGS, TRI, SKY>); 0 (default): one launch per group. The same
 *                           device code either way - bit-identical planes. Default 0 because it measured faster:
 *                           all 26 planes of a 4096^2 frame 0.654 ms in two launches against 0.698 ms in one (the
 *                           sky planes stream at 7 TB/s from k_sky's 256-thread workgroups, at 5 TB/s from the
 *                           one-wave workgroups the intercept kernel is built around).
 *   PM_OPT_LAST_REDO_PLANES read-only: how many planes of the latest FINISHED nearest / linear pm_map_cube were
 *                           mapped a second time with their nanmedian (the values the first pass stored for
 *                           them were provisional).
 *   PM_OPT_ROUTE_NS_PER_PLANE + r  read-only: measured cost of route r on the current problem, ns per plane of
 *                           the whole pipeline (0: not measured).
 *   PM_OPT_LAST_LT_PATH     read-only: how the latest image-plane call solved the light time of the intercept, a mask:
 *                           1 closed form (waves clear of the limb), 2 ... stepping through the reference's iterates to
 *                           its final epoch (one epoch quantum is visible on the body), 4 the reference's own sequence
 *                           without the Newton step wherever the closed form does not apply, 8 one quantum TURNS the body
 *                           visibly (illumination and state at the epoch of illumf_c's own light-time solution).
 *   PM_OPT_LAST_DISC_KERNEL read-only (pm_get_option): which kernel the latest image-plane call
 *                           dispatched for the planes that need the intercept: 0 none yet,
 *                           1 spheroid fast path, 2 its triaxial variant, 3 general kernel.
 */
typedef enum pm_option {
    PM_OPT_GENERAL_KERNEL = 1,
    PM_OPT_HOST_CHUNK_BYTES = 2,
    PM_OPT_HOST_COPY_THREADS = 3,
    PM_OPT_ZERO_COPY = 4,
    PM_OPT_HOST_CUBE_ROUTE = 4,
    PM_OPT_LAST_DISC_KERNEL = 5,
    PM_OPT_SPARSE_FRAME = 6,
    PM_OPT_BLOCK_TABLE_CACHE = 7,
    PM_OPT_BLOCK_TABLE_HITS = 8,
    PM_OPT_ROUTE_EXPLORE = 9,
    PM_OPT_LAST_CUBE_ROUTE = 10,
    PM_OPT_LAST_REDO_PLANES = 11,
    PM_OPT_FUSE_PLANES = 12,
    PM_OPT_HOST_COPY_THREADS_IN_USE = 13,
    PM_OPT_HYBRID_FETCH_PERMILLE = 14,
    PM_OPT_FETCH_BLOCK_BYTES = 15,
    PM_OPT_LT_MODE = 24,
    PM_OPT_TRACE = 25,
    PM_OPT_SM_BATCH_PLANES = 26,
    PM_OPT_LAST_LT_PATH = 27,
    PM_OPT_SPLINE_SEGMENT = 28,
    PM_OPT_LAST_SPLINE_SEGMENT = 29,
    PM_OPT_LAST_SM_KNIFE_EDGES = 30,
    PM_OPT_LAST_SM_ILL_CONDITIONED = 31,
    PM_OPT_ROUTE_NS_PER_PLANE = 16, /* + route 0..4 */
    PM_OPT_LAST_STAGE_NS = 32       /* + stage 0..12, read-only: see below */
} pm_option;
int pm_set_option(pm_ctx *ctx, int option, int64_t value);
int pm_get_option(pm_ctx *ctx, int option, int64_t *value);

/*
 * Progress of a nearest / linear pm_map_cube (no reference counterpart: the reference's plane loop,
 * observation.py:892-904, reports through its progress hook, base.py:773-783). `cb(user, first, n)` is
 * called ON THE CALLING THREAD, inside pm_map_cube, each time the kernels of planes [first, first + n)
 * have been enqueued on the context stream - planes arrive in order, once each - so that a caller can
 * queue work behind them (an event on pm_stream(), then a collective on another stream) while later
 * planes are still being collected and copied in. What those kernels store is provisional for planes that
 * turn out to need their nanmedian: after the call has finished (pm_synchronize() for device buffers)
 * PM_OPT_LAST_REDO_PLANES says whether any were redone. cb = NULL removes the callback. The callback must
 * not call back into the same context.
 */
typedef void (*pm_chunk_callback)(void *user, int first_plane, int n_planes);
int pm_set_chunk_callback(pm_ctx *ctx, pm_chunk_callback cb, void *user);

/*
 * Pinned (page-locked) host memory. PM_MEM_HOST calls accept any host pointer; buffers that
 * are pinned - allocated here or registered in place - are moved by DMA at the full PCIe rate
 * without the staging copy, and are what routes 1 and 2 of PM_OPT_HOST_CUBE_ROUTE need. The reference
 * hands numpy arrays around (observation.py:240-318 loads the cube, body_xy.py:3166 makes the
 * planes); planetmapper_amd allocates those arrays from this pool.
 */
int pm_host_alloc(pm_ctx *ctx, uint64_t bytes, void **hptr);
int pm_host_free(pm_ctx *ctx, void *hptr); /* ctx may be NULL: a block that outlived the context that allocated it */
int pm_host_register(pm_ctx *ctx, void *hptr, uint64_t bytes);
int pm_host_unregister(pm_ctx *ctx, void *hptr);

/* Device memory helpers so non-HIP callers can keep data resident in HBM. */
int pm_device_malloc(pm_ctx *ctx, uint64_t bytes, void **dptr);
int pm_device_free(pm_ctx *ctx, void *dptr);
int pm_memcpy_h2d(pm_ctx *ctx, void *dst_dev, const void *src_host, uint64_t bytes);
int pm_memcpy_d2h(pm_ctx *ctx, void *dst_host, const void *src_dev, uint64_t bytes);

/*
 * DLPack export of device memory (the results the Python layer leaves in HBM: planetmapper_amd/device_array.py; no
 * counterpart in the reference, whose getters return numpy arrays - body_xy.py:2586-2630). A `hold` counts the live
 * exports of one block. pm_dlpack_export returns a DLManagedTensor* (dlpack.h, legacy layout: what a 'dltensor' capsule
 * carries; device type kDLROCM) whose deleter is C code of this library: a consumer that lets go - from any thread, at
 * interpreter shutdown - calls into no host language. pm_dlpack_release ends the owner's interest: with no export left
 * the hold is deleted at once (and the block freed with hipFree if `free_memory`), returns 1; otherwise the LAST
 * export's deleter does both, returns 0. pm_dlpack_delete runs a managed tensor's deleter (a capsule nobody consumed).
 * dtype_code: 0 int, 1 unsigned, 2 float (DLDataTypeCode).
 */
typedef struct pm_dlpack_hold pm_dlpack_hold;
pm_dlpack_hold *pm_dlpack_hold_create(void *dptr, int device);
void *pm_dlpack_export(pm_dlpack_hold *hold, int dtype_code, int bits, int ndim, const int64_t *shape);
int64_t pm_dlpack_exports(const pm_dlpack_hold *hold);
int pm_dlpack_release(pm_dlpack_hold *hold, int free_memory);
void pm_dlpack_delete(void *managed_tensor);

/* State ----------------------------------------------------------------------- */
/* replaces Body.__init__ state (body.py:323-606) as seen by the pixel loops */
int pm_set_geometry(pm_ctx *ctx, const pm_geometry *geometry);
/* replaces BodyXY.set_disc_params / set_img_size (body_xy.py:700-961) */
int pm_set_disc(pm_ctx *ctx, const pm_disc *disc);

/* Image-space backplanes ------------------------------------------------------- */
/*
 * replaces the pixel loops BodyXY._get_targvec_img (body_xy.py:3195),
 * _get_lonlat_img (:3281), _get_lonlat_centric_img (:3346), _get_radec_img (:3409),
 * _get_km_xy_img (:3545), _get_illumination_gie_img (:3658), get_azimuth_angle_img
 * (:3742), get_local_solar_time_img (:3787), _get_state_imgs (:3830),
 * get_radial_velocity_img (:3895), get_doppler_img (:3938),
 * _get_limb_coordinate_imgs (:3964) and _get_ring_plane_coordinate_imgs (:4059)
 * with ONE fused kernel launch.
 *
 * plane_mask : OR of PM_PLANE_BIT(p) for the planes wanted.
 * out        : array of PM_NUM_PLANES pointers; out[p] must point to ny*nx doubles
 *              for every requested p (others are ignored and may be NULL).
 * alt        : altitude adjustment in km (reference `alt` kwarg of
 *              get_backplane_img, body_xy.py:2586; semantics of
 *              _AdjustedSurfaceAltitude body.py:172-229).
 */
int pm_backplanes_img(pm_ctx *ctx, uint64_t plane_mask, double alt,
                      double *const *out, int mem);

/*
 * Row block of pm_backplanes_img: image rows [row_begin, row_begin + n_rows) only, out[p] ->
 * n_rows*nx doubles (output row r = image row row_begin + r). Pixels are independent
 * (body_xy.py:3155-3164 iterates them one by one), so a frame shards over GPUs by row blocks
 * with no halo; planetmapper_amd.distributed.backplanes_img_sharded all-gathers the blocks.
 */
int pm_backplanes_img_rows(pm_ctx *ctx, uint64_t plane_mask, double alt, int row_begin,
                           int n_rows, double *const *out, int mem);

/* Point transforms --------------------------------------------------------------- */
/*
 * replaces the array-valued coordinate transforms built on
 * SpiceBase._maybe_transform_as_arrays (base.py:719-757):
 *   BodyXY.xy2radec / radec2xy / xy2lonlat / lonlat2xy / xy2km / km2xy / xy2angular /
 *   angular2xy (body_xy.py:385-561); Body.lonlat2radec / radec2lonlat / radec2angular /
 *   angular2radec / angular2lonlat / lonlat2angular / km2radec / radec2km / km2lonlat /
 *   lonlat2km / km2angular / angular2km (body.py:1083-1217, 1375-1800).
 * n points (a[i], b[i]) in system `from` -> (out_a[i], out_b[i]) in system `to`.
 * alt: TO lon/lat = altitude adjustment of the surface (radii + alt); FROM lon/lat =
 * altitude of the point above the surface. flags: PM_TF_*. Points that miss the body /
 * are not visible / are non-finite give NaN (not_found_nan=True semantics).
 */
int pm_transform(pm_ctx *ctx, int from, int to, uint64_t n, const double *a,
                 const double *b, double alt, int flags, double *out_a, double *out_b,
                 int mem);

/*
 * replaces the point forms of the ring / limb backplane loops and of radec2lonlat for n sky
 * points: Body.radec2lonlat (body.py:1083-1112), Body.ring_plane_coordinates
 * (body.py:2617-2658; ring_only_visible = its only_visible argument, rule :2598-2611) and
 * Body.limb_coordinates_from_radec (body.py:2040-2110).
 * out: 8 arrays of n doubles, array k at out + k * n: planetographic lon, lat [deg] of the
 * intercept (NaN off the disc), ring-plane radius [km], longitude [deg], distance [km],
 * limb longitude, latitude [deg], distance above the limb [km]. alt: altitude adjustment of the
 * surface. Non-finite RA/Dec give NaN rows.
 */
int pm_radec_query(pm_ctx *ctx, uint64_t n, const double *ra_deg, const double *dec_deg,
                   double alt, int ring_only_visible, double *out, int mem);

/* Map-space ---------------------------------------------------------------------- */
/*
 * replaces BodyXY._get_targvec_map (body_xy.py:3227), _get_illumf_map (:3667),
 * _get_obsvec_map (:3273), _get_radec_map (:3419) and _get_xy_map (:3478):
 * planetographic (lon, lat) grid [deg] -> image pixel coordinates of each map
 * location, NaN where not visible or outside the image frame.
 * lon_deg/lat_deg/x_map/y_map: n0*n1 doubles each.
 */
int pm_xy_map(pm_ctx *ctx, const double *lon_deg, const double *lat_deg, int n0,
              int n1, double alt, double *x_map, double *y_map, int mem);

/*
 * Map-space backplanes (reference get_*_map family, e.g. body_xy.py:3290-3300,
 * 3667-3675, 3843-3867): same plane ids as pm_backplanes_img, evaluated on a
 * lon/lat grid. out[p] -> n0*n1 doubles.
 */
int pm_backplanes_map(pm_ctx *ctx, uint64_t plane_mask, const double *lon_deg,
                      const double *lat_deg, int n0, int n1, double alt,
                      double *const *out, int mem);

/*
 * replaces BodyXY.map_img (body_xy.py:1414-1631) applied to every plane of a cube,
 * i.e. Observation._get_mapped_data (observation.py:876-905).
 * cube: P planes of ny*nx elements of `dtype`, plane-major; out: P*n0*n1 doubles.
 * interpolation: pm_interpolation; propagate_nan as in map_img.
 * The NaN pre-clean of _replace_nans_with_interpolated_values (body_xy.py:1871-1904)
 * is honoured for propagate_nan == 0 and for +-inf pixels (window means on the fly,
 * plane nanmedian by a GPU radix select when a sampled pixel needs it).
 * PM_MEM_DEVICE: asynchronous (no host round trip per call); planes that turn out to
 * need the nanmedian are completed by the next pm_synchronize(), until which cube / maps /
 * out must stay valid; if another pm_map_cube was enqueued in between, pm_synchronize()
 * reports PM_ERR_STATE instead. Calls with propagate_nan == 0 complete synchronously. The spline
 * interpolations wait on the host for their first solve (whether a plane needs its nanmedian decides
 * what is launched next) and 'smooth' for the footprint of the map; the rest of their work is
 * enqueued like the other modes'.
 */
int pm_map_cube(pm_ctx *ctx, const void *cube, int dtype, int n_planes,
                const double *x_map, const double *y_map, int n0, int n1,
                int interpolation, int propagate_nan, double *out, int mem);

/*
 * replaces Observation.get_mapped_data (observation.py:826-905) as a whole: the x/y map of a
 * lon/lat grid (pm_xy_map) AND the reprojection of the cube's planes onto it (pm_map_cube), with the
 * same results as those two calls. With device buffers, nearest / linear interpolation and up to 8
 * planes it is ONE kernel launch (one lane per map cell: pixel coordinates, then the planes), which
 * is what a frame-by-frame caller wants; any other request runs the two calls. x_map / y_map
 * (n0*n1 doubles each) receive the map - they are outputs here and must stay valid until
 * pm_synchronize() like the other buffers of an asynchronous pm_map_cube.
 */
int pm_mapped_data(pm_ctx *ctx, const void *cube, int dtype, int n_planes, const double *lon_deg,
                   const double *lat_deg, int n0, int n1, double alt, int interpolation,
                   int propagate_nan, double *x_map, double *y_map, double *out, int mem);

/*
 * Plane sharding over the GPUs of one node -------------------------------------------------
 * The planes of a cube are independent (observation.py:892-904): rank r of `world` maps the
 * contiguous block [start, stop) = pm_shard_bounds(P, world, r), per_rank = ceil(P / world) planes
 * (trailing ranks may own fewer, or none), on its own GPU; ONE all-gather of the mapped planes
 * (RCCL over xGMI) assembles the result on every rank. One process (and one pm_ctx) per GPU.
 *
 * pm_comm: an RCCL communicator bound at run time (dlopen of librccl.so.1; PM_ERR_UNSUPPORTED if
 * RCCL is not installed). Rank 0 calls pm_comm_unique_id() and hands the 128 bytes to the other
 * ranks by any means (file, socket, MPI, torch.distributed); every rank then calls
 * pm_comm_create() - a collective call. A process that drives the sharding through
 * torch.distributed instead (planetmapper_amd.distributed) needs none of this.
 *
 * pm_map_cube_sharded(): `local_cube` holds ONLY this rank's planes [start, stop) (host memory
 * with mem = PM_MEM_HOST_CUBE - each rank feeds its block over its own PCIe link - or HBM with
 * PM_MEM_DEVICE); x_map / y_map / out_all are device pointers; out_all has room for
 * world * per_rank * n0 * n1 doubles, block r at out_all + r * per_rank * n0 * n1 (planes beyond P
 * are NaN padding). With gather != 0 the block is cut into exchanges of
 * pm_exchange_planes(per_rank, n0, n1) planes (a function of shapes only: every rank issues the same
 * sequence). ONE pm_map_cube maps the block - its own pipeline stays whole - and each time the kernels of
 * further planes are on the context stream (pm_set_chunk_callback) the exchanges they complete are started
 * on the communicator's own stream (ncclSend / ncclRecv with every peer in one group: an all-gather whose
 * pieces land rank-major), so exchange k crosses xGMI while later planes are still being collected and
 * copied in. After the mapping has finished (flag check / nanmedian replay) whatever was not started - a
 * rank without planes, a rank whose mapping failed - is started all the same, and one 8-byte all-reduce
 * closes the call: (ranks that failed, ranks that redid planes after sending them). Any failure: every
 * rank returns an error (its own code, PM_ERR_PEER for a failure elsewhere) - no rank is left waiting in a
 * collective, the gathered cube is valid everywhere or nowhere. Any redo (rare: +-inf pixels): every rank
 * sends its block once more. A failing RCCL call aborts the communicator. The call returns when out_all is
 * complete on this rank.
 * With gather == 0, or comm == NULL (single process), only this rank's block is written - no
 * collective (each rank keeps / writes its slice). mem = PM_MEM_HOST (gather == 0 only):
 * everything is host memory and out_all is the caller's whole (P, n0, n1) array, e.g. ONE array in
 * shared memory for all ranks - this rank writes planes [start, stop) of it and nothing else.
 */
typedef struct pm_comm pm_comm; /* opaque */
int pm_shard_bounds(int n_planes, int world, int rank, int *start, int *stop, int *per_rank);
/* planes per exchange of the pipelined all-gather (>= 1; at most 8 exchanges of at least 4 MiB each) */
int pm_exchange_planes(int per_rank, int n0, int n1);
int pm_comm_unique_id(void *id128);
int pm_comm_create(pm_ctx *ctx, int world, int rank, const void *id128, pm_comm **comm);
int pm_comm_destroy(pm_comm *comm);
int pm_map_cube_sharded(pm_ctx *ctx, pm_comm *comm, const void *local_cube, int dtype,
                        int n_planes_total, const double *x_map, const double *y_map, int n0,
                        int n1, int interpolation, int propagate_nan, double *out_all, int mem,
                        int gather);

/*
 * Options of PM_INTERP_SMOOTH: the `smooth_oversample_by` and
 * `smooth_max_oversampled_img_size` arguments of BodyXY.map_img (body_xy.py:1427-1428,
 * used at :1724-1741). Defaults 5 and 10000, as in the reference; oversample_by <= 1
 * samples the PCHIP-cleaned image on the original pixel grid.
 */
int pm_set_smooth_options(pm_ctx *ctx, int oversample_by, int max_oversampled_img_size);

/*
 * `spline_smoothing` of BodyXY.map_img (body_xy.py:1420, used at :1673-1680 as the `s` of
 * scipy's RectBivariateSpline = FITPACK regrid). 0 (default): interpolating splines. s > 0:
 * smoothing splines for PM_INTERP_LINEAR and PM_INTERP_SPLINE(kr, kc) - knots are added where
 * the residuals are largest until the least-squares spline reaches a residual sum <= s, then
 * the smoothing parameter is found by FITPACK's rational interpolation; fits run per plane on
 * the GPU (calls with s > 0 complete synchronously).
 */
int pm_set_spline_smoothing(pm_ctx *ctx, double s);

#ifdef __cplusplus
}
#endif
#endif /* PLANETMAPPER_HIP_H */
