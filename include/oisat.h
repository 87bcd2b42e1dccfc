/*
 * oisat.h -- C-ABI of liboisat_hip.so: the MI355X (gfx950) implementation of the
 * optimal-interpolation hot path of ahsouri/OI-SAT-GMI.
 *
 * The reference has NO native/FFI layer: its boundary for this path is the Python call surface
 * (SURVEY.md section 8(b)).  This header is therefore the C-ABI *underneath* that surface -- what
 * the Python drop-in (oi-sat-gmi_amd/oisatgmi/) binds with ctypes, and what a maintainer of the
 * reference would bind from its own modules (stub shown in INTEGRATION.md).  Each entry point
 * names the reference code it replaces (file:line into the reference tree).
 *
 * Conventions
 *   - plain C: opaque handle, raw pointers, explicit int64 sizes; no torch / numpy types.
 *   - every function returns 0 on success, a negative OISAT_E* code on failure;
 *     oisat_last_error() gives the thread-local message.
 *   - "dev" pointers are HIP device pointers owned by the caller (e.g. torch tensors'
 *     data_ptr(), or memory from oisat_dmalloc).  Kernels are enqueued on the handle's stream
 *     (oisat_set_stream) and are asynchronous unless stated.  Internal workspaces are grow-only:
 *     a compute call allocates only when it meets a size larger than any before it on this handle
 *     (oisat_dense_reserve pre-sizes the dense path so that this never happens inside a timed or
 *     repeated region); nothing synchronises the device except where a host result is returned.
 *   - dtype: OISAT_F32 (0) or OISAT_F64 (1) selects the field element type; arithmetic is done
 *     in that type with the reference's operation order (no fast-math, no FMA contraction), and
 *     reductions always accumulate in double.
 */
#ifndef OISAT_H
#define OISAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OISAT_F32 0
#define OISAT_F64 1

#define OISAT_OK 0
#define OISAT_EINVAL (-1)   /* bad argument */
#define OISAT_EHIP (-2)     /* HIP runtime error */
#define OISAT_ENOMEM (-3)   /* device allocation failed */
#define OISAT_ENOTPD (-4)   /* Cholesky met a non-positive pivot */
#define OISAT_ENODEV (-5)   /* no usable gfx950 device */

#define OISAT_MAX_SCALES 128

typedef struct oisat_ctx oisat_ctx;

/* ---- lifetime, errors, memory, stream -------------------------------------------------------- */
int oisat_init(int device_id, oisat_ctx** out);
void oisat_shutdown(oisat_ctx* h);
const char* oisat_last_error(void);
const char* oisat_version(void);
int oisat_device_info(oisat_ctx* h, char* name_out, int name_cap, int* cu_count, int64_t* hbm_bytes);

int oisat_set_stream(oisat_ctx* h, void* hip_stream);       /* NULL = the default stream */
int oisat_stream_create(oisat_ctx* h);                      /* give this handle its own non-blocking stream: several
                                                               handles on one device then run concurrently (tiles) */
int oisat_stream_create_masked(oisat_ctx* h, int reserve_per_xcd); /* like oisat_stream_create, but the stream's kernels keep
                                                               off `reserve_per_xcd` CUs of every XCD (hipExtStreamCreateWithCUMask):
                                                               used for the bulk group of a BatchedFactor so that the dependent
                                                               chain of the critical group finds CUs free of GEMM waves.  Such a
                                                               stream synchronises with the NULL stream (HIP: it is a blocking one) */
int oisat_bind_thread(oisat_ctx* h);                        /* hipSetDevice(handle's device) for the CALLING host thread:
                                                               HIP's current device is per thread, so a worker thread that
                                                               drives a handle calls this once before anything else.  One
                                                               handle is driven by one host thread at a time. */
int oisat_wait_for(oisat_ctx* waiter, oisat_ctx* signaler); /* work enqueued on waiter's stream after this call starts only
                                                               when everything enqueued on signaler's stream so far is
                                                               done (event record + stream wait; nothing blocks the host) */
int oisat_sync(oisat_ctx* h);                               /* hipStreamSynchronize(stream) */
int oisat_query(oisat_ctx* h, int* busy);                   /* hipStreamQuery(stream): *busy = 1 while work is in flight;
                                                               lets a host that drives several handles serve whichever
                                                               finishes first (BatchedFactor: a group's solves are released
                                                               when THAT group is factored, whatever the enqueue order) */
int oisat_dmalloc(oisat_ctx* h, size_t bytes, void** dev_out);
int oisat_dfree(oisat_ctx* h, void* dev);
int oisat_h2d(oisat_ctx* h, void* dev_dst, const void* host_src, size_t bytes);   /* async on stream */
int oisat_d2h(oisat_ctx* h, void* host_dst, const void* dev_src, size_t bytes);   /* synchronises */
int oisat_memset(oisat_ctx* h, void* dev, int byte_value, size_t bytes);

/* per-kernel HIP-event timing (bench.py roofline leg).  Off by default. */
int oisat_prof_enable(oisat_ctx* h, int on);
int oisat_prof_reset(oisat_ctx* h);
/* synchronises; writes up to cap records; returns the number of distinct kernels (>=0) */
int oisat_prof_collect(oisat_ctx* h, int cap, char (*names)[64], double* total_ms, int64_t* launches);

/* ---- element-wise OI: optimal_interpolation.py:6-52 ------------------------------------------ */
/* Regularisation sweep, optimal_interpolation.py:26-33: for each scale s,
 *   t = Sa*s; K = t*(t+So)^-1; Sb = (1-K)*t; AK = 1 - Sb/t; mean_s = nanmean(AK).
 * Sa, So: dev, n elements.  scales: HOST doubles (nscales <= OISAT_MAX_SCALES).
 * mean_out / count_out: HOST arrays of nscales (sum of non-NaN AK / count; count of non-NaN).
 * Synchronises (returns host values).  Deterministic: fixed launch shape, fixed reduction order. */
int oisat_oi_curve(oisat_ctx* h, int dtype, const void* Sa, const void* So, int64_t n,
                   const double* scales, int nscales, double* mean_out, int64_t* count_out);

/* Analysis for ONE scale, optimal_interpolation.py:14 and :46-52:
 *   Y[Y<0] = 0 (written back in place), K, AK, Sb as above, inc = K*(Y-Xa), Xb = Xa+inc,
 *   err = sqrt(Sb).  All dev, n elements; any of Xb/AK/inc/err may be NULL to skip it. */
int oisat_oi_apply(oisat_ctx* h, int dtype, const void* Xa, void* Y_inout, const void* Sa,
                   const void* So, int64_t n, double scale, void* Xb, void* AK, void* inc, void* err);

/* Fully device-side OI: sweep + knee pick (Kneedle, the algorithm of kneed.KneeLocator used at
 * optimal_interpolation.py:37-41, restated) + analysis, no host round trip; graph-capturable.
 * index_dev: dev int32[1] receives the chosen index; curve_dev: dev double[nscales] (may be NULL). */
int oisat_oi_fused(oisat_ctx* h, int dtype, const void* Xa, void* Y_inout, const void* Sa,
                   const void* So, int64_t n, const double* scales, int nscales, int forced_index,
                   void* Xb, void* AK, void* inc, void* err, int32_t* index_dev, double* curve_dev);

/* ---- monthly averaging: averaging.py:11-24 and :97-108 ---------------------------------------- */
/* out[c] = nanmean_k stack[k][c]  (np.nanmean(axis=0), sequential-k summation like numpy).
 * stack: dev, k*n contiguous.  inf_to_nan != 0 first maps +/-inf to NaN (averaging.py:92). */
int oisat_nanmean_stack(oisat_ctx* h, int dtype, const void* stack, int k, int64_t n, int inf_to_nan, void* out);

/* error_averager, averaging.py:11-24: per cell drop NaN/inf, out = sqrt(sum / count^2);
 * count 0 -> NaN.  square_input != 0: stack holds sigma and is squared first (averaging.py:101). */
int oisat_error_average(oisat_ctx* h, int dtype, const void* stack, int k, int64_t n, int square_input, void* out);

/* out = (x - offset) / slope  (bias_correct, driver.py:65-106) or x / divisor with offset 0
 * (O3 unit conversion, driver.py:62-63). */
int oisat_affine(oisat_ctx* h, int dtype, const void* x, int64_t n, double offset, double slope, void* out);

/* Sa = (Xa*error_ctm/100)^2 and So = err^2, the argument wiring of oisatgmi.oi, driver.py:110-114 */
int oisat_oi_variances(oisat_ctx* h, int dtype, const void* Xa, const void* sat_err, int64_t n,
                       double error_ctm, void* Sa_out, void* So_out);

/* Emission scaling factor written to the output file: posterior/prior with NaN, +/-inf and 0 mapped to 1.0
 * (write_to_nc, driver.py:204-206). */
int oisat_scaling_factor(oisat_ctx* h, int dtype, const void* posterior, const void* prior, int64_t n, void* out);

/* ---- AMF recalculation (upstream of the averaging): amf_recal.py ---------------------------------------- */
/* Model partial column deltap*profile/g/Mair*N_A*1e-4*1e-15*100*1e-9, left to right in `dtype` (:51-56).
 * profile == NULL: the air column deltap/g/Mair*N_A*1e-4*1e-15*100 (ak_conv_mopitt.py:66). */
int oisat_partial_column(oisat_ctx* h, int dtype, const void* deltap, const void* profile, int64_t n, void* out);

/* Per-pixel vertical interpolation and AMF (the Python double loop :93-119 plus the record update :176-182):
 * scattering weights interpolated in log-pressure onto the model levels (scipy interp1d, linear,
 * fill_value="extrapolate": stable sort, searchsorted-left, clipped end segments), inf -> 0, tropopause
 * mask, model SCD / VCD by nansum in NumPy's pairwise order, new AMF = SCD/VCD (NaN if VCD == 0),
 * vcd_out = amf*vcd/new_amf, ctm_vcd = model VCD (NaN where vcd_out is NaN/inf).
 * Cubes are level-major [nz][n]; satellite cubes double, model cubes of ctm_dtype (np.log and the VCD
 * nansum are evaluated in that dtype, as NumPy does); nzs <= 64, nzc <= 128; tropopause may be NULL. */
int oisat_amf_recal(oisat_ctx* h, const double* sat_pmid, const double* sat_sw, int nzs, int ctm_dtype,
                    const void* ctm_pmid, const void* ctm_partial, int nzc, const double* tropopause, const double* vcd,
                    const double* amf, int64_t n, double* new_amf, double* vcd_out, double* ctm_vcd);

/* No scattering weights (:160-171): ctm_vcd = nansum over levels of the partial columns (levels with
 * p < tropopause dropped when tropopause != NULL), NaN where vcd is NaN; cube and output in `dtype`. */
int oisat_column_sum(oisat_ctx* h, int dtype, const void* ctm_pmid, const void* ctm_partial, int nzc,
                     const double* tropopause, const double* vcd, int64_t n, void* ctm_vcd);

/* ---- averaging-kernel convolution: ak_conv_mopitt.py / ak_conv_gosat.py (driver.py:46-51 conv_ak) ------------
 * The satellite_opt counterpart of oisat_amf_recal: per pixel the model column (cubes of dtype ctm_dtype,
 * level-major [nzc][n]) is interpolated in log-pressure onto the satellite levels (double, [nzs][n]) with
 * scipy interp1d's arithmetic, then the retrieval's averaging kernels are applied.
 * MOPITT (ak_conv_mopitt.py:118-146): fill NaN outside the model's pressure range; averaging_kernels is
 *   [nzs+1][n] (row 0 = surface); model_vcd = aprior_column + nansum(AK[1:]*(log10 x - log10 apriori_profile))
 *   + AK[0]*(log10 x_model[0] - log10 apriori_surface); model_xcol = 1e6*model_vcd/nansum(air partial column);
 *   pixels with NaN vcd are skipped (both NaN), +/-inf vcd -> model_vcd NaN.
 * GOSAT (ak_conv_gosat.py:118-143): linear extrapolation; model_xcol = nansum over levels of
 *   pressure_weight*(apriori + AK*(x - apriori)) with non-positive terms dropped; NaN/inf x_col -> NaN. */
int oisat_ak_conv_mopitt(oisat_ctx* h, int ctm_dtype, const void* ctm_pmid, const void* ctm_profile,
                         const void* ctm_air_partial, int nzc, const double* sat_pmid,
                         const double* averaging_kernels, const double* apriori_profile, int nzs,
                         const double* aprior_column, const double* apriori_surface, const double* vcd,
                         int64_t n, double* model_vcd, double* model_xcol);
int oisat_ak_conv_gosat(oisat_ctx* h, int ctm_dtype, const void* ctm_pmid, const void* ctm_profile, int nzc,
                        const double* sat_pmid, const double* averaging_kernels,
                        const double* apriori_profile, const double* pressure_weight, int nzs,
                        const double* x_col, int64_t n, double* model_xcol);

/* ---- model precipitable water for SSMIS: pwv_cal.py (driver.py:42-44 cal_pwv) ---------------------------------
 * oisat_water_column: deltap*profile/g/10000 per level, left to right in `dtype` (:63,:70).
 * oisat_pwv_sum: out = nansum_k(partial[k]/1000) level after level in `dtype`, NaN where the observation
 * (double vcd[n]) is NaN or +/-inf (:96-98).  partial: dev [nz][n] (after the optional model upscaling). */
int oisat_water_column(oisat_ctx* h, int dtype, const void* deltap, const void* profile, int64_t n, void* out);
int oisat_pwv_sum(oisat_ctx* h, int dtype, const void* partial, int nz, const double* vcd, int64_t n, void* out);

/* ---- regridding: interpolator.py:10-97 --------------------------------------------------------- */
/* signal.convolve2d(Z, ones(ky,kx)/(kx*ky)^(1|2), boundary='symm', mode='same'),
 * interpolator.py:40-46,:72-76.  Z, out: dev Ny*Nx row-major.  variance != 0 -> /(kx*ky)^2. */
int oisat_boxfilter_symm(oisat_ctx* h, int dtype, const void* Z, int64_t Ny, int64_t Nx, int ky, int kx,
                         int variance, void* out);

/* Bounded-radius exact nearest neighbour (what cKDTree.query + the `dists > 2*threshold` mask
 * need, interpolator.py:145-150,:78-91,:28-33).  Points/targets: dev double lon/lat arrays.
 * idx_out: dev int32[T], -1 where no point lies within max_dist (those cells are NaN-masked by
 * the caller's gather).  Exact ties resolve to the lowest point index (see oisat_nn_query_ties).  Distances are Euclidean in
 * degree space in double, like the reference.  Synchronises internally (sizes a workspace). */
int oisat_nn_query(oisat_ctx* h, const double* plon, const double* plat, int64_t P,
                   const double* tlon, const double* tlat, int64_t T, double max_dist,
                   int32_t* idx_out, double* dist_out /* may be NULL */);

/* Same search, and additionally reports the targets whose nearest point is NOT unique: tie_list (dev int32[T])
 * receives, in no particular order, the ids of the kept targets for which a second point lies at the same distance
 * (to within 8 ulp of the squared distance); *n_ties (host) = how many.  Where the minimum is unique it is what
 * cKDTree(points).query(xi) returns (interpolator.py:82-88); where it is not, the reference's answer is whichever
 * of the equidistant nodes scipy's tree traversal meets first, so the host side re-queries exactly the listed
 * targets against that tree and patches idx_out (oisatgmi/interpolator.py: NNIndex.query_device).  This happens on
 * the reference's own MOPITT / GOSAT settings: grid_size 1.0 (reader.py:1209,:1271) against a model longitude
 * spacing of 1.25 or 2.5 degrees puts every other model centre midway between two fine-grid nodes. */
int oisat_nn_query_ties(oisat_ctx* h, const double* plon, const double* plat, int64_t P,
                        const double* tlon, const double* tlat, int64_t T, double max_dist,
                        int32_t* idx_out, double* dist_out /* may be NULL */,
                        int32_t* tie_list, int64_t* n_ties);

/* out[f][t] = idx[t] >= 0 ? values[f][idx[t]] : NaN  for nfields stacked fields
 * (the `Z.ravel()[idx]` + mask of _interpolosis type 2/4, interpolator.py:17-20,:28-33). */
int oisat_gather_mask(oisat_ctx* h, int dtype, const void* values, int64_t P, int nfields,
                      const int32_t* idx, int64_t T, void* out);

/* LinearNDInterpolator(tri, values, fill_value=nan) evaluated at T targets for nfields stacked fields
 * (_interpolosis type 1, interpolator.py:12-16).  The Delaunay triangulation is built by qhull on the
 * host as in the reference (:153) and handed over as its arrays: simplices int32[ns][3], neighbors
 * int32[ns][3] (-1 = hull), transform double[ns][3][2] (scipy layout: Tinv rows, then the offset
 * vertex), vertex_to_simplex int32[P].  nn_idx: nearest swath pixel per target from oisat_nn_query
 * (-1: masked -> NaN); the point-location walk starts there.  Outside the hull -> NaN.  When the walk meets a
 * degenerate simplex or does not converge it falls back, like scipy's _find_simplex_directed, to the brute-force scan
 * (_find_simplex_bruteforce: bounding box, every simplex, eps_broad towards NaN-transform simplices).
 * bounds_host: HOST double[4] = {min x, max x, min y, max y} of the triangulated points (Delaunay.min_bound /
 * max_bound) for that scan's bounding-box test; NULL = no box test. */
int oisat_linear_interp(oisat_ctx* h, int dtype, const double* tlon, const double* tlat, int64_t T,
                        const int32_t* nn_idx, const int32_t* vertex_to_simplex, const int32_t* simplices,
                        const int32_t* neighbors, const double* transform, int64_t nsimplex,
                        const void* values, int64_t P, int nfields, void* out, const double* bounds_host);

/* scipy's Delaunay.transform on the device (the host computes it with three LAPACK calls per simplex: 0.55-1.0 s for the
 * 197,000 simplices of an OMI granule, more than qhull).  points: dev double[P][2] as Delaunay.points; simplices: dev
 * int32[nsimplex][3]; transform_out: dev double[nsimplex][3][2] = (T^-1 rows, r_2), NaN for a simplex whose 2 x 2 matrix
 * is singular or has a 1-norm condition number above 1 / (1000 eps), as scipy marks it.  Same elimination order as LAPACK's
 * (the values agree with scipy's to the last bit on the build host).  suspects (dev int32[nsimplex]) receives, in no
 * particular order, the simplices within four orders of magnitude of that limit or singular, *n_suspect (host) how many:
 * scipy's own decision and values for exactly those are a call of its routine on that subset (interpolator.py host side). */
int oisat_tri_transform(oisat_ctx* h, const double* points, int64_t P, const int32_t* simplices, int64_t nsimplex,
                        double* transform_out, int32_t* suspects, int64_t* n_suspect);

/* Targets whose location in the triangulation is NOT unique: amb_list (dev int32[T], no particular order) receives the
 * ids of the targets for which a second simplex accepts the point as well (it lies on a shared facet or vertex to within
 * scipy's eps) or whose walk had to fall back to the brute-force scan; *n_amb (host) = how many.  scipy evaluates targets
 * sequentially and starts every walk where the previous one ended (LinearNDInterpolator -> qhull._find_simplex with a
 * carried `start`), so for such targets the simplex -- and, next to a NaN vertex, the NaN pattern of interpolator.py:12-16 --
 * depends on the order of evaluation.  The normal case for level-3 lattice products (MOPITT, reader.py:1150-1211: every
 * fine node on the diagonal of a lattice square).  The host locates the listed targets with the same sequential search
 * (Delaunay.find_simplex over the whole target list) and passes the result to oisat_linear_interp_forced. */
int oisat_linear_locate(oisat_ctx* h, const double* tlon, const double* tlat, int64_t T, const int32_t* nn_idx,
                        const int32_t* vertex_to_simplex, const int32_t* simplices, const int32_t* neighbors,
                        const double* transform, int64_t nsimplex, int64_t P, const double* bounds_host,
                        int32_t* amb_list, int64_t* n_amb);

/* oisat_linear_interp with per-target simplices from the host: forced dev int32[T]; -2 = locate on the device (walk),
 * -1 = outside the triangulation (NaN), >= 0 = evaluate in that simplex. */
int oisat_linear_interp_forced(oisat_ctx* h, int dtype, const double* tlon, const double* tlat, int64_t T,
                               const int32_t* nn_idx, const int32_t* vertex_to_simplex, const int32_t* simplices,
                               const int32_t* neighbors, const double* transform, int64_t nsimplex,
                               const void* values, int64_t P, int nfields, void* out, const double* bounds_host,
                               const int32_t* forced);

/* RBFInterpolator(points, values, neighbors=5)(targets) for nfields stacked fields (_interpolosis type 3,
 * interpolator.py:21-27): thin-plate-spline kernel, degree-1 polynomial tail, no smoothing; per target the
 * `neighbors` (3..5, = min(5, P) in scipy) nearest points, ids sorted ascending, one (neighbors+3)^2 system
 * solved in double.  nn_idx: nearest point per target from oisat_nn_query with max_dist = cell (the mask
 * radius 2*threshold); targets with nn_idx < 0 are NaN after the reference's mask and get no value here
 * (oisat_rbf_check_masked looks at their neighbourhoods).  A NaN value makes every target whose
 * neighbourhood holds it NaN (as dgesv does).  *n_singular (host, may be NULL) = number of evaluated
 * targets whose system had a zero pivot (scipy raises LinAlgError("Singular matrix")); those targets are
 * written NaN.  Synchronises internally. */
int oisat_rbf_interp(oisat_ctx* h, int dtype, const double* plon, const double* plat, int64_t P,
                     const double* tlon, const double* tlat, int64_t T, const int32_t* nn_idx, double cell,
                     int neighbors, const void* values, int nfields, void* out, int64_t* n_singular);

/* oisat_rbf_interp that also reports the targets whose K-th neighbour is not unique (the runner-up is exactly as near):
 * which of the candidates RBFInterpolator's own KDTree(y).query(x, K) returns is a property of scipy's tree, and a regular
 * lattice of points (an L3 product) produces such targets wholesale.  tie_list: dev int32[T], *n_ties (host) entries filled,
 * in no particular order; their values in `out` come from the lowest-index choice and their zero pivots are NOT in
 * *n_singular -- send them through oisat_rbf_interp_forced with the neighbours the host's tree names. */
int oisat_rbf_interp_ties(oisat_ctx* h, int dtype, const double* plon, const double* plat, int64_t P,
                          const double* tlon, const double* tlat, int64_t T, const int32_t* nn_idx, double cell,
                          int neighbors, const void* values, int nfields, void* out, int64_t* n_singular,
                          int32_t* tie_list, int64_t* n_ties);

/* Type-3 evaluation of n listed targets on neighbourhoods given by the caller: targets dev int32[n] (indices into the T
 * targets), ids dev int32[n * neighbors] (point indices, any order; an entry outside [0, P) leaves that target as it is).
 * Overwrites out[f * T + target] for every field; *n_singular (host, may be NULL) = zero pivots among them. */
int oisat_rbf_interp_forced(oisat_ctx* h, int dtype, const double* plon, const double* plat, int64_t P,
                            const double* tlon, const double* tlat, int64_t T, const int32_t* targets,
                            const int32_t* ids, int64_t n, int neighbors, const void* values, int nfields, void* out,
                            int64_t* n_singular);

/* The other half of interpolator.py:21-27: scipy evaluates RBFInterpolator at EVERY target and masks afterwards,
 * so a singular neighbourhood raises LinAlgError for the whole call even when its target is masked (a regular
 * lattice of points and a target beyond its edge: five collinear neighbours).  For the targets with nn_idx < 0:
 * the `neighbors` nearest points are found (two-level search on a hash of at most 128 x 128 cells, so that a
 * target a whole domain away from every point costs no more than a near one), the system is factored, and
 * *n_singular (host) = number of zero pivots met.  Independent of the values: once per (points, targets) pair.
 * Synchronises internally. */
int oisat_rbf_check_masked(oisat_ctx* h, const double* plon, const double* plat, int64_t P, const double* tlon,
                           const double* tlat, int64_t T, const int32_t* nn_idx, double cell, int neighbors,
                           int64_t* n_singular);

/* _upscaler fused (interpolator.py:72-91): box-average of the ky*kx window around fine node
 * idx[t] (symmetric boundary, NaN-poisoning, optional variance kernel), evaluated only at the
 * fine nodes the model cells pick; idx < 0 -> NaN.  Z: dev nfields*Ny*Nx; out: dev nfields*T. */
int oisat_boxfilter_pick(oisat_ctx* h, int dtype, const void* Z, int64_t Ny, int64_t Nx, int nfields,
                         int ky, int kx, int variance, const int32_t* idx, int64_t T, void* out);

/* out = mask_dev ? x*1 : NaN  -- the `field*mask` of interpolator.py:126-128,:163; square != 0
 * squares first (uncertainty**2*mask, :186).  flag: dev array of dtype, kept if flag > thresh. */
int oisat_flag_mask(oisat_ctx* h, int dtype, const void* x, const void* flag, int64_t n, double thresh,
                    int square, void* out);

int oisat_sqrt(oisat_ctx* h, int dtype, const void* x, int64_t n, void* out);      /* interpolator.py:188 */

/* ---- dense Gaussian-B analysis (north-star extension; no reference counterpart) ---------------- */
/* x_a = x_b + B H^T (H B H^T + R)^-1 (y - H x_b),  B = D^1/2 C D^1/2,
 * C_ij = exp(-|p_i - p_j|^2 * g),  g = R_earth^2/(2 L^2), p = unit vectors (chord distance).
 * Reduces to optimal_interpolation.py:27,:49-50 when L -> 0 and H selects grid cells.
 * All coordinates are double SoA [3][count]; S is float, row-major, leading dimension ld. */

/* S = sig_a sig_b C(a,b) + delta_ab var_a, written for the lower triangle by 64x64 tiles (diagonal
 * tiles complete) over mp = roundup(m,128) rows; rows/cols m..mp are identity padding.
 * S must hold mp rows of ld >= mp floats.  oxyz: dev double[3*m]. */
int oisat_cov_build(oisat_ctx* h, const double* oxyz, const double* osig, const double* ovar, int64_t m,
                    double g, float* S, int64_t ld);

/* d = y - xb[cell]  (innovation; dev double out). */
int oisat_innovation(oisat_ctx* h, int dtype, const void* xb, const int64_t* cell, const double* y,
                     int64_t m, double* d_out);

/* The one O(m^3) kernel of the factorization, exposed for tests and microbenchmarks:
 * mode 0: C -= A B^T, mode 1: C = A B^T.  C: M x N (ldc), A: M x K (lda), B: N x K (ldb), row-major,
 * K-contiguous operands; M, N multiples of 128, K of 32; lower != 0: C is anchored on the diagonal and only its
 * lower triangle is defined afterwards -- tiles strictly above the diagonal are skipped, at 128- or 64-row/column
 * granularity depending on the launch size, so entries above the diagonal may or may not have been updated.
 * fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32. */
int oisat_gemm_nt(oisat_ctx* h, float* C, int64_t ldc, const float* A, int64_t lda, const float* B, int64_t ldb,
                  int64_t M, int64_t N, int64_t K, int mode, int lower);

/* In-place blocked Cholesky S = L L^T: L in the lower triangle; the strictly-upper part is NOT part of the result
 * (inside the 128x128 diagonal tiles the trailing updates write it; elsewhere it is left untouched), fp32 MFMA
 * trailing updates.  info_host: 0 ok, j>0 = first non-positive pivot column (1-based). */
int oisat_potrf(oisat_ctx* h, float* S, int64_t m, int64_t ld, int* info_host);

/* z <- L^-T L^-1 z  (dev double[m], fp32 factor, double accumulation). */
int oisat_potrs(oisat_ctx* h, const float* L, int64_t m, int64_t ld, double* z_inout);

/* r = d - (C.*sig sig^T + diag(var)) z  evaluated in double on the fly (iterative refinement).
 * olat_sorted (may be NULL): dev double[m], latitudes of the observations in degrees, valid only if the
 * observations are stored in ASCENDING latitude order; pairs whose latitudes differ by more than the angle at
 * which exp(-g chord^2) < 2^-64 are then skipped (one contiguous column range per block of rows). */
int oisat_cov_residual(oisat_ctx* h, const double* oxyz, const double* osig, const double* ovar, int64_t m,
                       double g, const double* d, const double* z, double* r_out, const double* olat_sorted);

/* Solve (H B H^T + R) z = d with the fp32 factor as a preconditioner: z = M^-1 d, then at most `refine` rounds of
 * { r = d - S z in float64; stop if |r| <= tol |d|; z += M^-1 r }.  The stopping test runs on the device (the launches of
 * the rounds after convergence return at once; nothing waits for the host).  tol: oisat_set_refine_tol, default 1e-6
 * (the increment is K r away from the exact one and |K| <= 1: fields within 1e-6 |d|); 0 = always run every round.
 * resid_host (may be NULL): relative residual norms before each round and after the last, refine+1 entries; rounds that
 * were not run repeat the last computed value.  Synchronises only when resid_host is given. */
int oisat_gain_solve(oisat_ctx* h, const float* L, const double* oxyz, const double* osig, const double* ovar,
                     int64_t m, int64_t ld, double g, const double* d, int refine, double* z_out,
                     double* resid_host, const double* olat_sorted /* as for oisat_cov_residual; may be NULL */);

/* inc_i = sig_i * sum_a C(i,a) osig_a z_a  (= row i of B H^T times z);  xa = xb + inc.
 * gxyz: dev double[3*n]; gsig: dev double[n]; z: dev double[m].  xb/xa/inc of `dtype`
 * (either of xa, inc may be NULL).  glat (dev double[n], degrees) and olat_sorted (as for oisat_cov_residual)
 * enable the latitude window: a block of cells only visits the observations within its latitude span +/- the
 * cut-off angle.  Either may be NULL: every pair is evaluated. */
int oisat_apply_increment(oisat_ctx* h, int dtype, const double* gxyz, const double* gsig, int64_t n,
                          const double* oxyz, const double* osig, const double* z, int64_t m, double g,
                          const void* xb, void* xa, void* inc, const double* glat, const double* olat_sorted);

/* The same for cells that form a regular ny x nx grid (row-major, cell = y * nx + x): the kernel then takes compact 32-wide
 * PATCHES of cells per workgroup instead of runs of consecutive cells, gives each patch a bounding sphere and skips the
 * observations of the latitude window that lie beyond the covariance's reach (2^-64) of that sphere -- on a 0.25 deg grid at
 * L = 300 km most of what the latitude window keeps: a polar cap's cells no longer visit the observations on the other
 * side of the pole.  Same sums over the same observations in the same order (terms below 2^-64 of a term left out). */
int oisat_apply_increment_grid(oisat_ctx* h, int dtype, const double* gxyz, const double* gsig, int64_t ny, int64_t nx,
                               const double* oxyz, const double* osig, const double* z, int64_t m, double g,
                               const void* xb, void* xa, void* inc, const double* glat, const double* olat_sorted);

/* perm: dev int32[m], a permutation of 0 .. m-1 that lists the observations (indices into the ascending-latitude order
 * they are stored in) along a space-filling curve, so that 64 consecutive entries are neighbours in space.  The float64
 * residual of the gain solves that follow on this handle for systems of exactly m observations (oisat_gain_solve,
 * oisat_cov_residual) then takes its blocks of 64 rows from this list: a block has a small bounding sphere, and of the
 * latitude window's observations only those within the covariance's reach (2^-64) of it are visited -- the residual's rows
 * are otherwise 64 consecutive LATITUDES, all around the globe.  Same terms per row in the same order.  NULL / m = 0 clears.
 * ONE-SHOT: the next oisat_gain_solve / oisat_cov_residual call on the handle takes the list (if its m matches) and the handle
 * forgets it, whatever that call returns -- perm must stay valid until that call's work has run, and is never read after. */
int oisat_set_obs_blocks(oisat_ctx* h, const int32_t* perm, int64_t m);

/* X <- X L^-T for nrows (multiple of 128) extra rows, X: dev float[nrows][ldx], ldx >= roundup(m,128).
 * Same MFMA GEMMs as the factorization (block forward substitution with the inverted diagonal blocks).
 * Must follow oisat_potrf of this L. */
int oisat_trsm_rows(oisat_ctx* h, const float* L, int64_t m, int64_t ld, float* X, int64_t nrows, int64_t ldx);

/* Posterior error sqrt(diag(B - B H^T S^-1 H B)) for grid cells [i0, i1):
 * err_i^2 = sig_i^2 - || L^-1 (H B)_{:,i} ||^2 -- rows of (H B)^T are generated on the device in chunks of
 * chunk_rows cells (<= 0: 4096), pushed through oisat_trsm_rows and reduced.  n*m^2 flop: meant for
 * regional / tiled analyses.  Generalises sqrt(Sb) of optimal_interpolation.py:29,:52.  err: dev float[i1-i0]. */
int oisat_posterior_error(oisat_ctx* h, const float* L, int64_t m, int64_t ld, const double* gxyz,
                          const double* gsig, int64_t n, int64_t i0, int64_t i1, const double* oxyz,
                          const double* osig, double g, int64_t chunk_rows, float* err);

/* diag(K H) at the observations: ak_a = 1 - R_aa (S^-1)_aa, (S^-1)_aa = || row a of L^-T ||^2
 * (the dense counterpart of AK = 1 - Sb/(Sa*reg), optimal_interpolation.py:31).  ak_out: dev double[m]. */
int oisat_gain_diag(oisat_ctx* h, const float* L, int64_t m, int64_t ld, const double* ovar, int64_t chunk_rows,
                    double* ak_out);

/* The ticket list of a task-graph launch over nsys systems of block_rows[s] block rows (largest first), as the library would
 * build it -- host only, no device: int32 quadruples (kind, system, i, j) with kind 0 = chain of `system` (i = first row of its
 * trace stamps), 1 = tile task T(i, j), 2 = SUB(j) (tile (j+1, j)), 3 = PRE(j) (tile (j, j)).  (The solve tasks of
 * oisat_batch_analyse are not tickets: they enter the launch's ready queue when their input is complete.)  wave <= 0: the
 * default (eight systems per wave behind wave 0).  capacity = 0 just counts.  max_wave_chains_out (may be NULL): the chain
 * tickets that can be resident at one time (the largest wave's and the next one's) -- a launch is only made when four times
 * that many workgroups are resident (each chain holds one and waits for tasks that the others must draw); otherwise the
 * factorization keeps the lock-step recursion.  For tests of the scheduling rule (every input of a task carries a lower
 * ticket, or is its system's chain) and of that bound. */
int oisat_dag_task_order(int nsys, const int32_t* block_rows, int wave, int32_t* tasks_out, int64_t capacity,
                         int64_t* ntasks_out, int32_t* reserve_out, int32_t* max_wave_chains_out);

/* Schedule of the factorizations this handle runs from now on (oisat_potrf, batches made by oisat_batch_create): 1 = the
 * task graph (ONE persistent launch of left-looking tile tasks, csrc/dense_dag.inc) wherever it applies, 0 = the recursion
 * (one launch per node, lock-step over a batch), -1 (default) = by size: the task graph from three block rows up, unless the
 * environment says otherwise (OISAT_DAG=0 | 1).  A caller that overlaps several batches on one GPU -- twelve months as two
 * lock-step groups -- switches the task graph off for them: a persistent launch holds every workgroup slot until it is done. */
int oisat_set_task_graph(oisat_ctx* h, int mode);

/* How a handle's batched factorization shares the GPU with other handles' work that runs at the same time (several
 * groups of systems factored side by side, oisatgmi/dense.py BatchedFactor): wave_prio 0..3 = s_setprio of its kernels'
 * waves -- where waves of two groups sit on one SIMD the arbiter serves the higher one first, so the group with the
 * longest dependent chain (the polar caps of a localised month) is not slowed by the other group's bulk GEMMs;
 * gemm_wg_per_cu 1 | 2 (0 = default 2) = workgroups per CU of its persistent GEMM launches, 1 leaves a slot on every CU
 * to the other group.  Results do not depend on either. */
int oisat_set_share(oisat_ctx* h, int wave_prio, int gemm_wg_per_cu);

/* Relative residual |d - S z| / |d| at which oisat_gain_solve stops refining on this handle (default 1e-6; 0: never). */
int oisat_set_refine_tol(oisat_ctx* h, double tol);

/* ---- batched factorization: many independent systems in lock-step ---------------------------------------------------
 * The tiles of a localised analysis (and the months of a batch) are small systems -- 4,000-18,000 observations -- whose
 * factorization is a chain of ~3 dependent launches per 128 columns, most of them far too small to fill 256 CUs.
 * oisat_batch_potrf advances all matrices of a batch through the SAME recursion at once: every launch covers every
 * matrix the step applies to (blockIdx.y = matrix), so the chain is paid once per batch and the launches are large.
 * The recursion tree is that of the LARGEST matrix of the batch: its factor is bit-identical to oisat_potrf's, the
 * smaller ones see their trailing updates associated differently and agree with oisat_potrf to fp32 rounding.
 *
 * oisat_batch_create: S[i] (dev, m[i] rows padded to roundup(m[i],128), leading dimension ld[i]) and tinv[i] (dev,
 *   roundup(m[i],128)*128 floats: receives the inverted diagonal blocks) for nmat matrices; HOST arrays of device pointers
 *   / sizes, copied.  The batch belongs to handle h and is factored on h's stream.
 * oisat_batch_potrf: pad + factor every matrix in place.  info_host (may be NULL; then failures are left to
 *   oisat_solve_status): int[2] = {first non-positive pivot column (1-based, 0 = none), index of that matrix}; synchronises.
 * oisat_factor_adopt: tell handle h that L (as factored by a batch, or by oisat_potrf on another handle) with inverted
 *   diagonal blocks tinv is "its" factor, so that oisat_gain_solve / oisat_potrs / oisat_trsm_rows / oisat_posterior_error
 *   on h accept it.  The caller orders the streams (oisat_wait_for). */
int oisat_batch_create(oisat_ctx* h, int nmat, float* const* S, const int64_t* m, const int64_t* ld, float* const* tinv,
                       int* batch_id_out);
int oisat_batch_potrf(oisat_ctx* h, int batch_id, int* info_host);

/* The solve phase of a batch in lock-step as well (round 3): after oisat_batch_set_solve has told the batch where every
 * member keeps its observations, innovation, solution, work vectors and grid, oisat_batch_solve enqueues -- on the batch's
 * handle, behind oisat_batch_potrf, no host hand-over -- the gain solve and the increment of ALL members with one launch
 * per step: pad, forward / backward sweep (tickets over (member, block row), rows ascending, so the factors of all systems
 * are streamed level by level), float64 residual, convergence test (per member, oisat_set_refine_tol), at most `refine`
 * corrections, increment.  Per member the arithmetic is that of oisat_gain_solve + oisat_apply_increment.
 * Arrays are indexed like those of oisat_batch_create.  work[i]: dev double[2 * roundup(m_i, 128)]; state[i]: dev, 256
 * zeroed bytes; observations in ascending latitude (olat[i]); xa[i] | inc[i]: where the analysis and the increment of
 * member i go (dtype of oisat_batch_solve). */
int oisat_batch_set_solve(oisat_ctx* h, int batch_id, int nmat, const double* const* oxyz, const double* const* osig,
                          const double* const* ovar, const double* const* d, const double* const* olat, double* const* z,
                          double* const* work, void* const* state, const double* const* gxyz, const double* const* gsig,
                          const double* const* glat, const int64_t* n, const void* const* xb, void* const* xa,
                          void* const* inc);
/* nx[i]: width of member i's cell grid (its n[i] cells are (n[i] / nx[i]) x nx[i], row-major; 0 = no such shape): the
 * increment of oisat_batch_solve then works like oisat_apply_increment_grid.  perm (may be NULL, entries may be NULL):
 * perm[i] = member i's observations along a space-filling curve, dev int32[m_i] (see oisat_set_obs_blocks): its float64
 * residuals then run on compact blocks of rows.  After oisat_batch_set_solve. */
int oisat_batch_set_grid(oisat_ctx* h, int batch_id, int nmat, const int64_t* nx, const int32_t* const* perm);
int oisat_batch_solve(oisat_ctx* h, int batch_id, int dtype, double g, int refine);
/* oisat_batch_potrf + oisat_batch_solve as ONE task-graph launch: every member's factorization, gain solve (sweeps, float64
 * residuals, convergence test, at most `refine` corrections) and increment are tasks of one persistent launch -- put into its
 * ready queue by the task that completes their input -- so the solves of the systems that are factored first run underneath
 * the factorization of the others instead of behind the whole batch
 * (a localised 720x1440 month: the solve phase was 12 of 62 ms with idle MFMA pipes).  Per member the same arithmetic as the
 * two calls.  Needs oisat_batch_set_solve (and oisat_batch_set_grid) and a batch whose factorization runs as a task graph
 * (oisat_set_task_graph; OISAT_EINVAL otherwise -- use the two calls).  info_host as in oisat_batch_potrf (NULL: unchecked,
 * asynchronous; failures go to oisat_solve_status_ex). */
int oisat_batch_analyse(oisat_ctx* h, int batch_id, int dtype, double g, int refine, int* info_host);
/* *yes_out = 1 if the batch's factorization runs as a task graph (decided at oisat_batch_create: the handle's schedule, the
 * sizes, and whether the chains of its largest wave leave room on the handle's CUs), i.e. whether oisat_batch_analyse applies. */
int oisat_batch_is_task_graph(oisat_ctx* h, int batch_id, int* yes_out);
int oisat_batch_destroy(oisat_ctx* h, int batch_id);
int oisat_factor_adopt(oisat_ctx* h, const float* L, int64_t m, int64_t ld, float* tinv);

/* ---- RCCL over xGMI: the two collectives of the sharded path ---------------------------------
 * The reference runs one scheduler job per month and exchanges nothing (run/job_submitter_sbatch.py:45-68).  One
 * process per GPU, (month x tile) units sharded statically: what is shared is broadcast once (the model grid), finished
 * fields are gathered to the root -- there is no collective on the data path.  librccl is dlopen'ed at the first call
 * (a copy already in the process, e.g. PyTorch's, is reused).  Enqueued on the handle's stream, asynchronous.
 *   oisat_comm_unique_id  on ONE rank: 128 opaque bytes, to be handed to every rank by the launcher (file, env, MPI, ...)
 *   oisat_comm_init       collective over all ranks: joins rank `rank` of `nranks`
 *   oisat_comm_bcast      dev_buf (bytes) of `root` -> every rank's dev_buf, in place
 *   oisat_comm_gather     every rank's `bytes` at send_dev -> recv_dev + r*bytes on `root` (recv_dev ignored elsewhere):
 *                         a gather, not an all-gather -- only the root needs the fields */
int oisat_comm_unique_id(char* id_out, int cap /* >= 128 */);
int oisat_comm_init(oisat_ctx* h, int rank, int nranks, const char* unique_id);
int oisat_comm_bcast(oisat_ctx* h, void* dev_buf, size_t bytes, int root);
int oisat_comm_gather(oisat_ctx* h, const void* send_dev, size_t bytes, void* recv_dev, int root);
int oisat_comm_destroy(oisat_ctx* h);

/* Status of the dense solves enqueued on this handle since the last call with clear != 0 (synchronises the stream; one
 * 36-byte read-back).  oisat_potrf / oisat_gain_solve only report failures when given info_host / resid_host; an
 * unchecked (fully asynchronous) run records them here instead.  out[0 .. nwords-1], nwords <= OISAT_STATUS_WORDS:
 *   OISAT_STATUS_NOTPD_COL          1-based column of the first non-positive pivot of any factorization (0 = none)
 *   OISAT_STATUS_NOTPD_BLOCKS       number of diagonal blocks that met one
 *   OISAT_STATUS_TRSV_TIMEOUTS      triangular-solve workgroups that gave up waiting for a predecessor (their part of z is a
 *                                   NaN fill pattern)
 *   OISAT_STATUS_UNCONVERGED        gain solves (oisat_gain_solve, members of oisat_batch_solve / oisat_batch_analyse) that took
 *                                   every one of their `refine` corrections and whose float64 residual |d - S z| was still above
 *                                   tol |d| behind the last one (oisat_set_refine_tol; tol = 0 and refine = 0 -- the plain
 *                                   solve, no residual is formed -- never count): z is the best
 *                                   iterate, NOT within the tolerance -- the exact gain of optimal_interpolation.py:27 is what
 *                                   the 1e-5 bar is measured against, so a caller must not take such fields for converged ones
 *   OISAT_STATUS_UNCONVERGED_MEMBER the caller's index (oisat_batch_create order) of the first such batch member; -1 = none,
 *                                   or a single-system solve
 *   OISAT_STATUS_DAG_TIMEOUTS       task-graph factorizations that ended on a time-out (bounded spins): incomplete factors
 * A caller must see zeros in words 0-3 and 5 before it trusts z / the analysis fields. */
enum { OISAT_STATUS_NOTPD_COL = 0, OISAT_STATUS_NOTPD_BLOCKS = 1, OISAT_STATUS_TRSV_TIMEOUTS = 2, OISAT_STATUS_UNCONVERGED = 3,
       OISAT_STATUS_UNCONVERGED_MEMBER = 4, OISAT_STATUS_DAG_TIMEOUTS = 5, OISAT_STATUS_WORDS = 6 };
int oisat_solve_status_ex(oisat_ctx* h, int32_t* out, int nwords, int clear);
/* The round-2 form: three words.  (It clears all six, so trsv_timeouts here also counts unconverged solves and task-graph
 * time-outs: three zeros still mean "trust the fields".)  Any of the three may be NULL. */
int oisat_solve_status(oisat_ctx* h, int* first_notpd_col, int* n_notpd_blocks, int* trsv_timeouts, int clear);

/* Pre-size every internal workspace of the dense path for analyses of up to max_obs observations (and, if
 * diag_chunk_rows > 0, for oisat_posterior_error / oisat_gain_diag with that chunk size), so that no later
 * oisat_potrf / oisat_gain_solve / ... call on this handle allocates or frees device memory. */
int oisat_dense_reserve(oisat_ctx* h, int64_t max_obs, int64_t diag_chunk_rows);

#ifdef __cplusplus
}
#endif
#endif /* OISAT_H */
