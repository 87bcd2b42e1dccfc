// Regridding primitives: interpolator.py:10-97 of the reference
//   * box filter  = signal.convolve2d(Z, ones/(kx*ky)^(1|2), boundary='symm', mode='same') (:40-46,:72-76)
//   * nearest neighbour = cKDTree(points).query(targets) + `dists > 2*threshold -> NaN` (:28-33,:78-91,:145-150)
//
// The reference builds a fresh k-d tree over ~1e6 nodes for every field it regrids.  Here the
// neighbour search is done once per (points, targets) pair on the device with a uniform-cell hash
// whose cell edge equals the mask radius: only neighbours within `max_dist` can survive the mask,
// so the 3x3 cell block around a target contains every candidate and the search is exact.  The
// resulting index vector is reused by every field (one gather kernel for a whole stack of fields).
// Distances are Euclidean in degree space, in double, exactly like the reference.  Exactly equidistant
// points DO occur on the reference's own settings (grid_size 1.0 against a 1.25 / 2.5 deg model longitude
// spacing puts every other model centre midway between two fine nodes); there the reference returns
// whichever node scipy's k-d tree meets first.  The kernel reports such targets (oisat_nn_query_ties) and
// the host resolves just those with the same tree the reference builds; everywhere else the minimum is
// unique and the lowest-index rule below never decides anything.
#include "oisat_common.h"

namespace {

__device__ __forceinline__ int64_t reflect(int64_t i, int64_t n) {
    // scipy 'symm': edge-repeating reflection, possibly more than once for windows wider than n
    while (i < 0 || i >= n) {
        if (i < 0) i = -i - 1;
        if (i >= n) i = 2 * n - 1 - i;
    }
    return i;
}

template <typename T>
__global__ __launch_bounds__(256) void boxfilter_kernel(const T* __restrict__ Z, int64_t Ny, int64_t Nx, int ky, int kx,
                                                         T w, T* __restrict__ out) {
    const int64_t total = Ny * Nx;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride) {
        const int64_t i = p / Nx, j = p % Nx;
        T acc = T(0);
        for (int a = 0; a < ky; ++a) {
            const int64_t ii = reflect(i - ky / 2 + a, Ny);
            for (int b = 0; b < kx; ++b) {
                const int64_t jj = reflect(j - kx / 2 + b, Nx);
                acc += Z[ii * Nx + jj] * w;
            }
        }
        out[p] = acc;
    }
}

// G lanes cooperate on one picked node: lanes stride the ky*kx window, then a fixed xor-tree adds
// the G partials (deterministic).  G in {1,4,16,64}.
template <typename T, int G>
__global__ __launch_bounds__(256) void boxfilter_pick_kernel(const T* __restrict__ Z, int64_t Ny, int64_t Nx, int nfields,
                                                              int ky, int kx, T w, const int32_t* __restrict__ idx, int64_t Tn,
                                                              T* __restrict__ out) {
    const int64_t gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int sub = threadIdx.x % G;
    const int64_t total = Tn * nfields;
    const int64_t gstride = ((int64_t)gridDim.x * blockDim.x) / G;
    const int win = ky * kx;
    for (int64_t q0 = 0; q0 < total; q0 += gstride) {        // uniform trip count: shuffles need all lanes
        const int64_t q = q0 + gid;
        const bool live = q < total;
        const int64_t f = live ? q / Tn : 0, t = live ? q % Tn : 0;
        const int32_t node = live ? idx[t] : -1;
        T acc = T(0);
        if (node >= 0) {
            const int64_t i = node / Nx, j = node % Nx;
            const T* Zf = Z + f * Ny * Nx;
            for (int e = sub; e < win; e += G) {
                const int a = e / kx, b = e % kx;
                acc += Zf[reflect(i - ky / 2 + a, Ny) * Nx + reflect(j - kx / 2 + b, Nx)] * w;
            }
        }
#pragma unroll
        for (int m = G / 2; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, kWave);
        if (live && sub == 0) out[q] = node >= 0 ? acc : nan_of<T>();
    }
}

// Row-wise form (kx <= 64): a wave takes 64 / kx consecutive targets, lane = (target slot, window column b).  For the
// regular grids of _upscaler consecutive model cells pick consecutive windows, so for each of the ky window rows the
// wave's ONE load instruction reads up to 64 consecutive floats of a fine-grid row -- whole 128-byte lines, each fetched
// once -- where the lane-strided form above reads ten 40-byte pieces per instruction (1.8x the algorithmic bytes left L2
// for the fabric: the pieces of a line went to different XCDs).  Blocks are renumbered so that every XCD (private L2)
// owns one contiguous range of targets.  Per window: column sums over the rows (row order), then the columns left to
// right on the slot's first lane -- a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void boxfilter_pick_rows_kernel(const T* __restrict__ Z, int64_t Ny, int64_t Nx, int nfields,
                                                                   int ky, int kx, T w, const int32_t* __restrict__ idx, int64_t Tn,
                                                                   T* __restrict__ out, int tpw, int64_t waves_per_field) {
    const unsigned nb = gridDim.x;                                     // a multiple of 8 (host)
    const unsigned blk = (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3);
    const int lane = threadIdx.x & 63;
    const int slot = lane / kx, b = lane - slot * kx;
    const int64_t total = waves_per_field * nfields;
    for (int64_t wq = (int64_t)blk * 4 + (threadIdx.x >> 6); wq < total; wq += (int64_t)nb * 4) {     // wave-uniform
        const int64_t f = wq / waves_per_field, tw = wq - f * waves_per_field;
        const int64_t t = tw * tpw + slot;
        const bool live = slot < tpw && t < Tn;
        const int32_t node = live ? idx[t] : -1;
        T acc = T(0);
        if (node >= 0) {
            const int64_t i = node / Nx, j = node % Nx;
            const int64_t col = reflect(j - kx / 2 + b, Nx);
            const T* Zf = Z + f * Ny * Nx;
            // eight window rows in flight at a time (independent loads), added in row order
            for (int a0 = 0; a0 < ky; a0 += 8) {
                T v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (a0 + q < ky) ? Zf[reflect(i - ky / 2 + a0 + q, Ny) * Nx + col] : T(0);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (a0 + q < ky) acc += v[q] * w;
            }
        }
        T tot = acc;
        for (int bb = 1; bb < kx; ++bb) {
            const T o = __shfl(acc, (lane + bb) & 63, kWave);
            if (b == 0) tot += o;
        }
        if (live && b == 0) out[f * Tn + t] = node >= 0 ? tot : nan_of<T>();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_mask_kernel(const T* __restrict__ values, int64_t P, int nfields,
                                                           const int32_t* __restrict__ idx, int64_t Tn, T* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < Tn; t += stride) {
        const int32_t s = idx[t];
        for (int f = 0; f < nfields; ++f) out[(int64_t)f * Tn + t] = s >= 0 ? values[(int64_t)f * P + s] : nan_of<T>();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void flag_mask_kernel(const T* __restrict__ x, const T* __restrict__ flag, int64_t n, T thresh,
                                                         bool square, T* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        T v = x[i];
        if (square) v = v * v;
        out[i] = flag[i] > thresh ? v * T(1) : v * nan_of<T>();     // x*1.0 or x*NaN, interpolator.py:126-128
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sqrt_kernel(const T* __restrict__ x, int64_t n, T* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = sqrt(x[i]);
}

// ---- uniform-cell hash nearest neighbour -------------------------------------------------------
struct HashGrid {
    double x0, y0, inv_h;
    int nbx, nby;
};

__device__ __forceinline__ int cell_of(const HashGrid& g, double x, double y, int& cx, int& cy) {
    cx = (int)floor((x - g.x0) * g.inv_h);
    cy = (int)floor((y - g.y0) * g.inv_h);
    return cy * g.nbx + cx;
}

__global__ __launch_bounds__(256) void nn_count_kernel(const double* __restrict__ px, const double* __restrict__ py, int64_t P,
                                                        HashGrid g, unsigned* __restrict__ counts, int32_t* __restrict__ pcell) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
        const double x = px[i], y = py[i];
        int c = -1;
        if (x == x && y == y) {
            int cx, cy;
            cell_of(g, x, y, cx, cy);
            cx = cx < 0 ? 0 : (cx >= g.nbx ? g.nbx - 1 : cx);
            cy = cy < 0 ? 0 : (cy >= g.nby ? g.nby - 1 : cy);
            c = cy * g.nbx + cx;
            atomicAdd(&counts[c], 1u);
        }
        pcell[i] = c;
    }
}

// Exclusive scan of the cell counters.  One tile = 4096 counters: 1024 threads x 4 consecutive values, wave scans by
// shuffle, the 16 wave totals scanned by wave 0.  (Round 2's version was one block that gave every thread one long chunk
// and walked it with dependent loads: 31 us for 17,000 cells.)
__device__ __forceinline__ unsigned tile_scan4(const unsigned v[4], unsigned ex[4], unsigned* wsum /*[17] shared*/) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned tsum = v[0] + v[1] + v[2] + v[3];
    unsigned inc = tsum;                                          // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(inc, o, kWave);
        if (lane >= o) inc += up;
    }
    __syncthreads();                                              // wsum of the previous tile has been consumed
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    if (wv == 0) {
        unsigned x = lane < 16 ? wsum[lane] : 0u, xi = x;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            const unsigned up = __shfl_up(xi, o, kWave);
            if (lane >= o) xi += up;
        }
        if (lane < 16) wsum[lane] = xi - x;                       // exclusive wave offsets
        if (lane == 15) wsum[16] = xi;                            // tile total
    }
    __syncthreads();
    unsigned run = wsum[wv] + inc - tsum;
    ex[0] = run; run += v[0];
    ex[1] = run; run += v[1];
    ex[2] = run; run += v[2];
    ex[3] = run;
    return wsum[16];
}

// Single-pass scan with decoupled look-back: block = one tile of 4096 counters (taken by ticket, so every predecessor has
// started), publishes its tile total, sums the totals / prefixes of the tiles before it, publishes its inclusive prefix.
// status[t]: bits 62-63 = 1 (tile total) | 2 (inclusive prefix), low 32 bits the value; status and ticket arrive zeroed.
// ticket[1]: raised if a look-back ever gives up on a predecessor (bounded spin) -- the prefix is then wrong, and the host
// fails the query instead of handing out indices built on it (nn_query_impl reads the word back).
__global__ __launch_bounds__(1024) void nn_scan_lookback_kernel(const unsigned* __restrict__ counts, int64_t n, unsigned* __restrict__ start,
                                                                 unsigned* __restrict__ cursor, unsigned long long* __restrict__ status,
                                                                 unsigned* __restrict__ ticket) {
    __shared__ unsigned wsum[17];
    __shared__ unsigned s_tile, s_prev;
    const int tid = threadIdx.x;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const int64_t tile = s_tile, first = tile * 4096, last = first + 4096 < n ? first + 4096 : n;
    const int64_t i = first + (int64_t)tid * 4;
    unsigned v[4], ex[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (i + q < last) ? counts[i + q] : 0u;
    const unsigned tot = tile_scan4(v, ex, wsum);
    if (tid == 0) {
        unsigned prev = 0;
        if (tile > 0) {
            __hip_atomic_store(&status[tile], (1ull << 62) | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int64_t j = tile - 1; j >= 0; --j) {
                unsigned long long st;
                unsigned spins = 0;
                while (((st = __hip_atomic_load(&status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 62) == 0ull) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 24)) {                         // bounded: a predecessor always publishes (it holds a lower ticket);
                        __hip_atomic_store(&ticket[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // if it ever does not: say so
                        break;
                    }
                }
                prev += (unsigned)(st & 0xffffffffull);
                if ((st >> 62) == 2ull) break;
            }
        }
        __hip_atomic_store(&status[tile], (2ull << 62) | (unsigned long long)(prev + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_prev = prev;
        if (last == n) start[n] = prev + tot;
    }
    __syncthreads();
    const unsigned carry = s_prev;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (i + q < last) {
            start[i + q] = carry + ex[q];
            cursor[i + q] = carry + ex[q];
        }
}

__global__ __launch_bounds__(256) void nn_scatter_kernel(const int32_t* __restrict__ pcell, int64_t P, unsigned* __restrict__ cursor,
                                                          int32_t* __restrict__ sorted, const double* __restrict__ px,
                                                          const double* __restrict__ py, double2* __restrict__ sxy) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += stride) {
        const int c = pcell[i];
        if (c >= 0) {
            const unsigned pos = atomicAdd(&cursor[c], 1u);
            sorted[pos] = (int32_t)i;
            sxy[pos] = make_double2(px[i], py[i]);          // the query walks a cell's points as one contiguous run
        }
    }
}

// Exact ties: when a second point lies at the same distance (to within a few ulp) the target's id is appended to
// `tie_list`; the caller resolves those targets with the reference's own tree (see oisat_nn_query_ties).
__global__ __launch_bounds__(256) void nn_query_kernel(const double* __restrict__ px, const double* __restrict__ py,
                                                        const double* __restrict__ tx, const double* __restrict__ ty, int64_t Tn,
                                                        HashGrid g, const unsigned* __restrict__ start,
                                                        const int32_t* __restrict__ sorted, const double2* __restrict__ sxy,
                                                        double max_dist, int32_t* __restrict__ idx_out, double* __restrict__ dist_out,
                                                        int32_t* __restrict__ tie_list, unsigned* __restrict__ tie_count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < Tn; t += stride) {
        const double x = tx[t], y = ty[t];
        double best = __builtin_inf(), second = __builtin_inf();
        int32_t bi = -1;
        if (x == x && y == y) {
            int cx, cy;
            cell_of(g, x, y, cx, cy);
            for (int yy = cy - 1; yy <= cy + 1; ++yy) {
                if (yy < 0 || yy >= g.nby) continue;
                for (int xx = cx - 1; xx <= cx + 1; ++xx) {
                    if (xx < 0 || xx >= g.nbx) continue;
                    const int c = yy * g.nbx + xx;
                    for (unsigned s = start[c]; s < start[c + 1]; ++s) {
                        const double2 p = sxy[s];
                        const double dx = p.x - x, dy = p.y - y;
                        const double d2 = dx * dx + dy * dy;
                        if (d2 < best) { second = best; best = d2; bi = sorted[s]; }
                        else if (d2 == best) {                  // exact tie: lowest point index (the order inside a cell is arbitrary)
                            const int32_t i = sorted[s];
                            second = best;
                            if (i < bi) bi = i;
                        } else if (d2 < second) second = d2;
                    }
                }
            }
        }
        const double d = sqrt(best);
        const bool keep = bi >= 0 && !(d > max_dist);          // mask is `dists > 2*threshold`
        idx_out[t] = keep ? bi : -1;
        if (dist_out) dist_out[t] = keep ? d : __builtin_inf();
        if (tie_list && keep && second - best <= best * 1.8e-15) tie_list[atomicAdd(tie_count, 1u)] = (int32_t)t;
    }
}

// ---- Delaunay linear interpolation (interpolator type 1) -------------------------------------------
// LinearNDInterpolator(tri, values, fill_value=nan)((X, Y)), interpolator.py:12-16.  The triangulation
// itself comes from qhull on the host, exactly as in the reference (:153); what the reference then
// repeats for every field -- point location and barycentric evaluation over ~1e6 targets -- runs here,
// once per target for all stacked fields.  Point location is scipy's directed walk
// (_find_simplex_directed): hop across the facet opposite the first barycentric coordinate below -eps,
// leave the hull -> NaN; the walk starts at a simplex incident to the target's nearest swath pixel
// (known from the neighbour search), so it takes a handful of hops.
// scipy's _find_simplex_bruteforce (qhull.pyx), what _find_simplex_directed falls back to when the walk meets a
// degenerate simplex (NaN transform / a coordinate above 1 + eps) or runs out of hops: bounding-box test, then every
// simplex in index order -- a valid transform is tested with eps; a degenerate one (NaN transform) is replaced by its
// valid neighbours, tested with the wider eps_broad = sqrt(DBL_EPSILON) on the side that faces the degenerate simplex.
// Rare (regular swath grids can make qhull emit zero-area simplices), so a plain per-thread scan is fine.
struct TriBounds { double xmin, xmax, ymin, ymax; };

__device__ inline int32_t find_simplex_bruteforce(double x, double y, const int32_t* __restrict__ neighbors,
                                                  const double* __restrict__ transform, int32_t ns, TriBounds bb, double eps,
                                                  double& c0, double& c1, double& c2) {
    const double eps_broad = 1.4901161193847656e-08;
    if (x < bb.xmin - eps || x > bb.xmax + eps || y < bb.ymin - eps || y > bb.ymax + eps) return -1;
    for (int32_t is = 0; is < ns; ++is) {
        const double* tr = transform + (int64_t)is * 6;
        if (tr[0] == tr[0]) {                                          // _barycentric_inside
            const double dx = x - tr[4], dy = y - tr[5];
            c0 = tr[0] * dx + tr[1] * dy;
            if (!(-eps <= c0 && c0 <= 1.0 + eps)) continue;
            c1 = tr[2] * dx + tr[3] * dy;
            if (!(-eps <= c1 && c1 <= 1.0 + eps)) continue;
            c2 = 1.0 - c0 - c1;
            if (!(-eps <= c2 && c2 <= 1.0 + eps)) continue;
            return is;
        }
        for (int k = 0; k < 3; ++k) {
            const int32_t nb = neighbors[(int64_t)is * 3 + k];
            if (nb == -1) continue;
            const double* tn = transform + (int64_t)nb * 6;
            if (tn[0] != tn[0]) continue;
            const double dx = x - tn[4], dy = y - tn[5];
            const double b0 = tn[0] * dx + tn[1] * dy, b1 = tn[2] * dx + tn[3] * dy;
            const double b[3] = {b0, b1, 1.0 - b0 - b1};
            bool inside = true;
            for (int m = 0; m < 3; ++m) {
                const double lo = (neighbors[(int64_t)nb * 3 + m] == is) ? -eps_broad : -eps;
                if (!(lo <= b[m] && b[m] <= 1.0 + eps)) { inside = false; break; }
            }
            if (inside) { c0 = b[0]; c1 = b[1]; c2 = b[2]; return nb; }
        }
    }
    return -1;
}

// Ambiguous locations.  scipy evaluates the targets one after the other and starts each walk at the simplex the previous
// target ended in, so a target that lies ON a shared facet or vertex (within eps) gets whichever of the simplices that
// accept it the walk from the previous target meets first.  The value is the same from either side -- unless the third
// vertex carries NaN, and then the NaN pattern of the output depends on it.  This is the normal case for level-3 lattice
// products (MOPITT MOP03: every fine node sits on the diagonal of a lattice square).  The kernel therefore reports the
// targets for which a second simplex also accepts the point (`amb_list`); the host locates exactly those with scipy's own
// sequential search over the same triangulation and hands the simplices back through `forced` (-2: not forced, walk;
// -1: outside; >= 0: that simplex).
__device__ __forceinline__ bool simplex_accepts(const double* __restrict__ tr, double x, double y, double eps) {
    if (tr[0] != tr[0]) return true;                                   // degenerate neighbour: let scipy decide
    const double dx = x - tr[4], dy = y - tr[5];
    const double b0 = tr[0] * dx + tr[1] * dy, b1 = tr[2] * dx + tr[3] * dy, b2 = 1.0 - b0 - b1;
    return b0 >= -eps && b0 <= 1.0 + eps && b1 >= -eps && b1 <= 1.0 + eps && b2 >= -eps && b2 <= 1.0 + eps;
}

template <typename T>
__global__ __launch_bounds__(256) void linear_interp_kernel(const double* __restrict__ tx, const double* __restrict__ ty, int64_t Tn,
                                                             const int32_t* __restrict__ nn_idx, const int32_t* __restrict__ v2s,
                                                             const int32_t* __restrict__ simplices, const int32_t* __restrict__ neighbors,
                                                             const double* __restrict__ transform, int32_t ns, const T* __restrict__ values,
                                                             int64_t P, int nfields, T* __restrict__ out, TriBounds bb,
                                                             const int32_t* __restrict__ forced, int32_t* __restrict__ amb_list,
                                                             unsigned* __restrict__ amb_count) {
    const double eps = 100.0 * 2.220446049250313e-16;              // scipy: eps = 100 * DBL_EPSILON
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < Tn; t += stride) {
        const int32_t nn = nn_idx[t];
        int32_t is = -1;
        double c0 = 0.0, c1 = 0.0, c2 = 0.0;
        const int32_t f = forced ? forced[t] : -2;
        if (nn >= 0 && f >= -1) {                                   // located by scipy's sequential search
            is = f < ns ? f : -1;
            if (is >= 0) {
                const double* tr = transform + (int64_t)is * 6;
                const double dx = tx[t] - tr[4], dy = ty[t] - tr[5];
                c0 = tr[0] * dx + tr[1] * dy;
                c1 = tr[2] * dx + tr[3] * dy;
                c2 = 1.0 - c0 - c1;
            }
        } else if (nn >= 0) {
            const double x = tx[t], y = ty[t];
            bool amb = false;
            is = v2s[nn];
            if (is < 0 || is >= ns) is = 0;
            const int max_hops = 1 + ns / 4;
            int hop = 0;
            for (; hop < max_hops; ++hop) {
                const double* tr = transform + (int64_t)is * 6;      // [Tinv(2x2) | r(2)]
                const double dx = x - tr[4], dy = y - tr[5];
                int go = -2;                                           // -2: inside; >= -1: hop target; -3: broken simplex
                c0 = tr[0] * dx + tr[1] * dy;
                if (c0 < -eps) go = neighbors[(int64_t)is * 3 + 0];
                else {
                    if (!(c0 <= 1.0 + eps)) go = -3;
                    c1 = tr[2] * dx + tr[3] * dy;
                    if (c1 < -eps) go = neighbors[(int64_t)is * 3 + 1];
                    else {
                        if (!(c1 <= 1.0 + eps)) go = -3;
                        c2 = 1.0 - c0 - c1;
                        if (c2 < -eps) go = neighbors[(int64_t)is * 3 + 2];
                        else if (!(c2 <= 1.0 + eps)) go = -3;
                    }
                }
                if (go == -2) break;                                   // found
                if (go == -1) { is = -1; break; }                      // left the hull: NaN, no second look (scipy: return -1)
                if (go == -3) {                                        // degenerate simplex in the way: brute force
                    is = find_simplex_bruteforce(x, y, neighbors, transform, ns, bb, eps, c0, c1, c2);
                    amb = true;
                    break;
                }
                is = go;
            }
            if (hop >= max_hops) {                                     // no convergence
                is = find_simplex_bruteforce(x, y, neighbors, transform, ns, bb, eps, c0, c1, c2);
                amb = true;
            }
            if (amb_list) {
                if (is >= 0 && !amb) {                                 // does the simplex across a near facet accept it too?
                    const double c[3] = {c0, c1, c2};
                    for (int k = 0; k < 3; ++k) {
                        if (c[k] > 1e-6) continue;
                        const int32_t nb = neighbors[(int64_t)is * 3 + k];
                        if (nb >= 0 && simplex_accepts(transform + (int64_t)nb * 6, x, y, eps)) amb = true;
                    }
                }
                if (amb) amb_list[atomicAdd(amb_count, 1u)] = (int32_t)t;
            }
        }
        if (is < 0) {
            for (int f2 = 0; f2 < nfields; ++f2) out[(int64_t)f2 * Tn + t] = nan_of<T>();
        } else {
            const int32_t v0 = simplices[(int64_t)is * 3], v1 = simplices[(int64_t)is * 3 + 1], v2 = simplices[(int64_t)is * 3 + 2];
            for (int f2 = 0; f2 < nfields; ++f2) {
                const T* vf = values + (int64_t)f2 * P;
                double o = 0.0;
                o += c0 * (double)vf[v0];
                o += c1 * (double)vf[v1];
                o += c2 * (double)vf[v2];
                out[(int64_t)f2 * Tn + t] = (T)o;
            }
        }
    }
}

// ---- local thin-plate-spline RBF (interpolator type 3) ----------------------------------------------
// RBFInterpolator(points, values, neighbors=5)(targets), interpolator.py:21-27: for every target the K
// nearest points (ids sorted ascending, as scipy sorts them), the (K+3)x(K+3) system
//     [ tps(|y_i - y_j|)   1  xhat_i  yhat_i ] [c]   [d]
//     [ 1 xhat yhat (transposed)    0        ] [ ] = [0]        tps(r) = r^2 log r,  tps(0) = 0
// on neighbourhood coordinates shifted/scaled to [-1, 1], and out = [tps(|x - y_i|), 1, xhat, yhat] . c.
// scipy factors each distinct neighbourhood with LAPACK dgesv in a Python loop.  Here one thread owns a
// target: it finds its K neighbours by expanding rings of the point hash until the K-th distance is
// inside the searched square, factors the symmetric system once with partially pivoted LU held in
// registers, solves for the evaluation weights w = A^-1 vec (A is symmetric, so out = w[:K] . d) and
// applies them to every stacked field.  A NaN value poisons its neighbourhood exactly as a NaN
// right-hand side poisons dgesv's coefficients.  A zero pivot is LAPACK's info > 0: counted, NaN written.
// Targets whose nearest point is beyond the mask radius (nn_idx < 0) come back NaN after the mask anyway, but scipy evaluates
// them too and one singular neighbourhood among them raises LinAlgError for the whole call (a regular lattice of points and
// a target beyond its edge: five collinear neighbours).  They get a second launch (FAR) on a coarse hash (at most 128 cells
// a side, so an isolated target's ring search is bounded) that finds the neighbours, factors and counts zero pivots, and
// writes nothing.
__device__ __forceinline__ double tps(double r2) {      // r^2 log r with r = sqrt(r2), evaluated as scipy does
    const double r = sqrt(r2);
    return r == 0.0 ? 0.0 : (r * r) * log(r);
}

// One hash cell into the K best (distance, then index) of a target.  A cell wholly farther than the K-th candidate holds
// nothing new (bounds widened by 1e-9 of a cell against the rounding of the cell assignment; the box's edge cells also hold
// what was clamped into them).
template <int K>
__device__ __forceinline__ void knn_scan_cell(const HashGrid& g, double hcell, int xx, int yy, double x, double y,
                                              const double* __restrict__ px, const double* __restrict__ py,
                                              const unsigned* __restrict__ start, const int32_t* __restrict__ sorted,
                                              double (&bd)[K], int32_t (&id)[K]) {
    const int c = yy * g.nbx + xx;
    const unsigned s0 = start[c], s1 = start[c + 1];
    if (s0 == s1) return;
    const double xlo = xx == 0 ? -__builtin_inf() : g.x0 + (xx - 1e-9) * hcell;
    const double xhi = xx == g.nbx - 1 ? __builtin_inf() : g.x0 + (xx + 1 + 1e-9) * hcell;
    const double ylo = yy == 0 ? -__builtin_inf() : g.y0 + (yy - 1e-9) * hcell;
    const double yhi = yy == g.nby - 1 ? __builtin_inf() : g.y0 + (yy + 1 + 1e-9) * hcell;
    const double gx = fmax(0.0, fmax(xlo - x, x - xhi)), gy = fmax(0.0, fmax(ylo - y, y - yhi));
    if ((gx * gx + gy * gy) * (1.0 - 1e-12) > bd[K - 1]) return;
    for (unsigned s = s0; s < s1; ++s) {
        const int32_t i = sorted[s];
        const double dx = px[i] - x, dy = py[i] - y;
        const double d2 = dx * dx + dy * dy;
        if (d2 < bd[K - 1] || (d2 == bd[K - 1] && i < id[K - 1])) {
            bd[K - 1] = d2;
            id[K - 1] = i;
#pragma unroll
            for (int q = K - 1; q > 0; --q) {
                const bool sw = bd[q] < bd[q - 1] || (bd[q] == bd[q - 1] && id[q] < id[q - 1]);
                const double td = bd[q];
                const int32_t ti = id[q];
                bd[q] = sw ? bd[q - 1] : td;
                id[q] = sw ? id[q - 1] : ti;
                bd[q - 1] = sw ? td : bd[q - 1];
                id[q - 1] = sw ? ti : id[q - 1];
            }
        }
    }
}

// The (K+3)x(K+3) system of one target on its K neighbours: ids sorted, built, factored, evaluation weights solved for.
// Returns true when a pivot was exactly zero (LAPACK's info > 0).
template <int K>
__device__ __forceinline__ bool rbf_factor(double x, double y, int32_t (&id)[K], const double* __restrict__ px,
                                           const double* __restrict__ py, double (&w)[K + 3]) {
    constexpr int N = K + 3;
    bool singular = false;
    {
#pragma unroll
        for (int a = 0; a < K - 1; ++a)                       // ids ascending (np.sort(yindices, axis=1))
#pragma unroll
            for (int q = 0; q < K - 1 - a; ++q) {
                const int32_t lo = id[q] < id[q + 1] ? id[q] : id[q + 1];
                const int32_t hi = id[q] < id[q + 1] ? id[q + 1] : id[q];
                id[q] = lo;
                id[q + 1] = hi;
            }
        double yx[K], yy[K];
        double xmin = __builtin_inf(), xmax = -__builtin_inf(), ymin = __builtin_inf(), ymax = -__builtin_inf();
#pragma unroll
        for (int q = 0; q < K; ++q) {
            yx[q] = px[id[q]];
            yy[q] = py[id[q]];
            xmin = fmin(xmin, yx[q]); xmax = fmax(xmax, yx[q]);
            ymin = fmin(ymin, yy[q]); ymax = fmax(ymax, yy[q]);
        }
        const double shx = (xmax + xmin) / 2, shy = (ymax + ymin) / 2;
        double scx = (xmax - xmin) / 2, scy = (ymax - ymin) / 2;
        if (scx == 0.0) scx = 1.0;
        if (scy == 0.0) scy = 1.0;
        double a[N][N + 1];                                    // [A | vec]
#pragma unroll
        for (int i = 0; i < K; ++i) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const double dx = yx[i] - yx[j], dy = yy[i] - yy[j];
                a[i][j] = (i == j) ? 0.0 : tps(dx * dx + dy * dy);
            }
            a[i][K] = 1.0;
            a[i][K + 1] = (yx[i] - shx) / scx;
            a[i][K + 2] = (yy[i] - shy) / scy;
            a[K][i] = 1.0;
            a[K + 1][i] = a[i][K + 1];
            a[K + 2][i] = a[i][K + 2];
            const double dx = x - yx[i], dy = y - yy[i];
            a[i][N] = tps(dx * dx + dy * dy);
        }
#pragma unroll
        for (int i = K; i < N; ++i)
#pragma unroll
            for (int j = K; j < N; ++j) a[i][j] = 0.0;
        a[K][N] = 1.0;
        a[K + 1][N] = (x - shx) / scx;
        a[K + 2][N] = (y - shy) / scy;
        // LU with partial pivoting (first largest |.|, as idamax), augmented column carried along
#pragma unroll
        for (int k = 0; k < N; ++k) {
            int p = k;
            double pm = fabs(a[k][k]);
#pragma unroll
            for (int i = k + 1; i < N; ++i) {
                const double v = fabs(a[i][k]);
                if (v > pm) { pm = v; p = i; }
            }
            if (!(pm > 0.0)) singular = true;
#pragma unroll
            for (int i = k + 1; i < N; ++i) {
                const bool sw = (i == p);
#pragma unroll
                for (int j = k; j <= N; ++j) {
                    const double u = a[i][j], v = a[k][j];
                    a[i][j] = sw ? v : u;
                    a[k][j] = sw ? u : v;
                }
            }
            const double inv = 1.0 / a[k][k];
#pragma unroll
            for (int i = k + 1; i < N; ++i) {
                const double l = a[i][k] * inv;
#pragma unroll
                for (int j = k + 1; j <= N; ++j) a[i][j] -= l * a[k][j];
            }
        }
#pragma unroll
        for (int i = N - 1; i >= 0; --i) {
            double sacc = a[i][N];
#pragma unroll
            for (int j = i + 1; j < N; ++j) sacc -= a[i][j] * w[j];
            w[i] = sacc / a[i][i];
        }
    }
    return singular;
}

template <typename T, int K>
__global__ __launch_bounds__(64) void rbf_interp_kernel(const double* __restrict__ px, const double* __restrict__ py,
                                                         const double* __restrict__ tx, const double* __restrict__ ty, int64_t Tn,
                                                         const int32_t* __restrict__ nn_idx, HashGrid g,
                                                         const unsigned* __restrict__ start, const int32_t* __restrict__ sorted,
                                                         const T* __restrict__ values, int64_t P, int nfields, T* __restrict__ out,
                                                         int* __restrict__ n_singular, int32_t* __restrict__ tie_list,
                                                         unsigned* __restrict__ n_ties) {
    constexpr int N = K + 3;
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= Tn) return;
    bool ok = nn_idx[t] >= 0;
    const double x = tx[t], y = ty[t];
    ok = ok && x == x && y == y;
    double bd[K + 1];                                        // the K neighbours and the runner-up
    int32_t idk[K + 1];
#pragma unroll
    for (int q = 0; q <= K; ++q) { bd[q] = __builtin_inf(); idk[q] = -1; }
    if (ok) {
        int cx, cy;
        cell_of(g, x, y, cx, cy);
        // a target outside the points' box starts from the nearest cell of the box: a point in an unvisited cell is still
        // more than r cells away along one axis, as seen from the target too
        cx = cx < 0 ? 0 : (cx >= g.nbx ? g.nbx - 1 : cx);
        cy = cy < 0 ? 0 : (cy >= g.nby ? g.nby - 1 : cy);
        const double hcell = 1.0 / g.inv_h;
        int rlast = cx > cy ? cx : cy;                       // ring beyond which every cell has been visited
        if (g.nbx - 1 - cx > rlast) rlast = g.nbx - 1 - cx;
        if (g.nby - 1 - cy > rlast) rlast = g.nby - 1 - cy;
        for (int r = 0; r <= rlast; ++r) {
            for (int yy = cy - r; yy <= cy + r; ++yy) {
                if (yy < 0 || yy >= g.nby) continue;
                const bool edge_row = (yy == cy - r) || (yy == cy + r);
                const int step = edge_row ? 1 : (2 * r > 0 ? 2 * r : 1);      // interior rows: only the two end cells
                for (int xx = cx - r; xx <= cx + r; xx += step) {
                    if (xx < 0 || xx >= g.nbx) continue;
                    knn_scan_cell<K + 1>(g, hcell, xx, yy, x, y, px, py, start, sorted, bd, idk);
                }
            }
            // every unvisited point lies at least r cells away along one axis
            const double reach = r * hcell * (1.0 - 1e-12);
            if (idk[K] >= 0 && bd[K] < reach * reach) break;
        }
        ok = idk[K - 1] >= 0;
    }
    // the runner-up as near as the K-th neighbour: which of them scipy's tree returns is its traversal's business; the host
    // asks that tree and sends the target through rbf_forced_kernel
    const bool tie = ok && tie_list && idk[K] >= 0 && bd[K] == bd[K - 1];
    if (tie) tie_list[atomicAdd(n_ties, 1u)] = (int32_t)t;
    int32_t id[K];
#pragma unroll
    for (int q = 0; q < K; ++q) id[q] = idk[q];
    bool singular = false;
    double w[N];
    if (ok) {
        singular = rbf_factor<K>(x, y, id, px, py, w);
        if (singular && !tie) atomicAdd(n_singular, 1);
    }
    for (int f = 0; f < nfields; ++f) {
        double o = __builtin_nan("");
        if (ok && !singular) {
            const T* vf = values + (int64_t)f * P;
            o = 0.0;
#pragma unroll
            for (int q = 0; q < K; ++q) o += w[q] * (double)vf[id[q]];
        }
        out[(int64_t)f * Tn + t] = (T)o;
    }
}

// Targets whose neighbourhood the host dictates (ids[i * K ..], from scipy's own tree: exact ties for the K-th neighbour).
template <typename T, int K>
__global__ __launch_bounds__(64) void rbf_forced_kernel(const double* __restrict__ px, const double* __restrict__ py,
                                                         const double* __restrict__ tx, const double* __restrict__ ty, int64_t Tn,
                                                         const int32_t* __restrict__ targets, const int32_t* __restrict__ ids, int64_t n,
                                                         const T* __restrict__ values, int64_t P, int nfields, T* __restrict__ out,
                                                         int* __restrict__ n_singular) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const int64_t t = targets[i];
    if (t < 0 || t >= Tn) return;
    int32_t id[K];
    bool ok = true;
#pragma unroll
    for (int q = 0; q < K; ++q) {
        id[q] = ids[i * K + q];
        ok = ok && id[q] >= 0 && id[q] < P;
    }
    if (!ok) return;
    double w[K + 3];
    const bool singular = rbf_factor<K>(tx[t], ty[t], id, px, py, w);
    if (singular) atomicAdd(n_singular, 1);
    for (int f = 0; f < nfields; ++f) {
        double o = __builtin_nan("");
        if (!singular) {
            const T* vf = values + (int64_t)f * P;
            o = 0.0;
#pragma unroll
            for (int q = 0; q < K; ++q) o += w[q] * (double)vf[id[q]];
        }
        out[(int64_t)f * Tn + t] = (T)o;
    }
}

// Points per block of kSuper x kSuper hash cells (one thread per block; the far check's first level).
constexpr int kSuper = 8;
__global__ __launch_bounds__(256) void rbf_super_count_kernel(HashGrid g, const unsigned* __restrict__ start, int nsx, int nsy,
                                                               unsigned* __restrict__ super_count) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nsx * nsy) return;
    const int sx = b % nsx, sy = b / nsx;
    const int x0 = sx * kSuper, x1 = min(x0 + kSuper, g.nbx);
    unsigned n = 0;
    for (int yy = sy * kSuper; yy < min((sy + 1) * kSuper, g.nby); ++yy) n += start[yy * g.nbx + x1] - start[yy * g.nbx + x0];
    super_count[b] = n;
}

// The neighbourhoods of the MASKED targets (nn_idx < 0: the nearest point is beyond the mask radius, possibly a whole domain
// away): found, factored, zero pivots counted, nothing written.  Rings of hash cells around such a target are mostly empty and
// their number grows with the square of its distance, so the search is two-level instead: the occupied block of 8 x 8 cells
// nearest to the target first (that bounds the K-th distance well), then every other occupied block that still reaches
// inside that bound, cell by cell under the same bound.
template <int K>
__global__ __launch_bounds__(64) void rbf_far_check_kernel(const double* __restrict__ px, const double* __restrict__ py,
                                                            const double* __restrict__ tx, const double* __restrict__ ty, int64_t Tn,
                                                            const int32_t* __restrict__ nn_idx, HashGrid g,
                                                            const unsigned* __restrict__ start, const int32_t* __restrict__ sorted,
                                                            const unsigned* __restrict__ super_count, int nsx, int nsy,
                                                            int* __restrict__ n_singular) {
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= Tn || nn_idx[t] >= 0) return;
    const double x = tx[t], y = ty[t];
    if (!(fabs(x) < __builtin_inf() && fabs(y) < __builtin_inf())) return;
    double bd[K];
    int32_t id[K];
#pragma unroll
    for (int q = 0; q < K; ++q) { bd[q] = __builtin_inf(); id[q] = -1; }
    const double hcell = 1.0 / g.inv_h, hs = hcell * kSuper;
    auto block_gap2 = [&](int sx, int sy) {            // lower bound of the squared distance to anything in the block
        const double xlo = sx == 0 ? -__builtin_inf() : g.x0 + (sx - 1e-9) * hs;
        const double xhi = sx == nsx - 1 ? __builtin_inf() : g.x0 + (sx + 1 + 1e-9) * hs;
        const double ylo = sy == 0 ? -__builtin_inf() : g.y0 + (sy - 1e-9) * hs;
        const double yhi = sy == nsy - 1 ? __builtin_inf() : g.y0 + (sy + 1 + 1e-9) * hs;
        const double gx = fmax(0.0, fmax(xlo - x, x - xhi)), gy = fmax(0.0, fmax(ylo - y, y - yhi));
        return (gx * gx + gy * gy) * (1.0 - 1e-12);
    };
    auto scan_block = [&](int sx, int sy) {
        for (int yy = sy * kSuper; yy < min((sy + 1) * kSuper, g.nby); ++yy)
            for (int xx = sx * kSuper; xx < min((sx + 1) * kSuper, g.nbx); ++xx)
                knn_scan_cell<K>(g, hcell, xx, yy, x, y, px, py, start, sorted, bd, id);
    };
    int first = -1;
    double first_gap = __builtin_inf();
    for (int b = 0; b < nsx * nsy; ++b) {
        if (super_count[b] == 0) continue;
        const double gap = block_gap2(b % nsx, b / nsx);
        if (gap < first_gap) { first_gap = gap; first = b; }
    }
    if (first < 0) return;                              // no finite point at all
    scan_block(first % nsx, first / nsx);
    for (int b = 0; b < nsx * nsy; ++b) {
        if (b == first || super_count[b] == 0) continue;
        if (block_gap2(b % nsx, b / nsx) > bd[K - 1]) continue;
        scan_block(b % nsx, b / nsx);
    }
    if (id[K - 1] < 0) return;                          // fewer than K finite points: the caller has refused that already
    double w[K + 3];
    if (rbf_factor<K>(x, y, id, px, py, w)) atomicAdd(n_singular, 1);
}

// Barycentric transforms of a triangulation's simplices: scipy's ``Delaunay.transform`` (qhull._get_barycentric_transforms) --
// per simplex the 2 x 2 matrix T_ij = (r_j - r_2)_i, LAPACK dgetrf + dgecon + dgetrs against the identity, NaN where the
// 1-norm condition estimate falls below 1000 eps -- costs 0.55-1.0 s of host time per 98,640-pixel granule (three LAPACK
// calls per simplex under the GIL), more than qhull itself.  Here one thread per simplex walks the same elimination in the
// same order of operations: partial pivoting on the first column, the multiplier as a product with the pivot's reciprocal,
// u22 = p22 - l p12 in two roundings, the back substitution's update fused, divisions as products with reciprocals -- which
// on the build host reproduces scipy's values bit for bit (197,254 of 197,254 simplices of a granule; OpenBLAS picks its
// kernels by CPU, so scipy itself is only reproducible to the last bit per machine).  The condition number of a 2 x 2 matrix
// is computed exactly; a simplex within four orders of magnitude of the limit (or singular) is reported as `suspect` and the
// host asks scipy about exactly those.
__global__ __launch_bounds__(256) void tri_transform_kernel(const double* __restrict__ pts, const int32_t* __restrict__ simplices,
                                                             int64_t ns, double* __restrict__ out, int32_t* __restrict__ suspects,
                                                             unsigned* __restrict__ n_suspect) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= ns) return;
    const int32_t v0 = simplices[3 * i], v1 = simplices[3 * i + 1], v2 = simplices[3 * i + 2];
    const double rx = pts[2 * (int64_t)v2], ry = pts[2 * (int64_t)v2 + 1];
    const double ax = pts[2 * (int64_t)v0] - rx, ay = pts[2 * (int64_t)v0 + 1] - ry;
    const double bx = pts[2 * (int64_t)v1] - rx, by = pts[2 * (int64_t)v1 + 1] - ry;
    // what LAPACK sees (column-major reading of scipy's row-major T): M = [[ax, ay], [bx, by]]
    const double anorm = fmax(fabs(ax) + fabs(bx), fabs(ay) + fabs(by));
    const bool piv = fabs(bx) > fabs(ax);
    const double p11 = piv ? bx : ax, p12 = piv ? by : ay, p21 = piv ? ax : bx, p22 = piv ? ay : by;
    const double ip11 = 1.0 / p11;
    const double l = __dmul_rn(p21, ip11);
    const double u22 = __dsub_rn(p22, __dmul_rn(l, p12));
    const double iu22 = 1.0 / u22;
    double x[2][2];                                            // x[c] = M^-1 e_c
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const double b1 = (c == 0) != piv ? 1.0 : 0.0, b2 = (c == 0) != piv ? 0.0 : 1.0;      // the permuted unit vector
        const double y2 = __dsub_rn(b2, __dmul_rn(l, b1));
        const double x2 = __dmul_rn(y2, iu22);
        const double t = __fma_rn(-p12, x2, b1);
        x[c][0] = __dmul_rn(t, ip11);
        x[c][1] = x2;
    }
    const double inorm = fmax(fabs(x[0][0]) + fabs(x[0][1]), fabs(x[1][0]) + fabs(x[1][1]));
    const double rcond = 1.0 / (anorm * inorm);
    const bool singular = p11 == 0.0 || u22 == 0.0 || !(rcond == rcond) || !(fabs(rcond) < __builtin_inf());
    const bool degenerate = singular || rcond < 1000.0 * 2.220446049250313e-16;
    if (singular || rcond < 1e-9) suspects[atomicAdd(n_suspect, 1u)] = (int32_t)i;
    double* o = out + 6 * i;
    const double nan = __builtin_nan("");
    o[0] = degenerate ? nan : x[0][0];
    o[1] = degenerate ? nan : x[0][1];
    o[2] = degenerate ? nan : x[1][0];
    o[3] = degenerate ? nan : x[1][1];
    o[4] = degenerate ? nan : rx;
    o[5] = degenerate ? nan : ry;
}

}  // namespace

extern "C" int oisat_tri_transform(oisat_ctx* h, const double* points, int64_t P, const int32_t* simplices, int64_t nsimplex,
                                   double* transform_out, int32_t* suspects, int64_t* n_suspect) {
    ARG_CHECK(h && points && simplices && transform_out && suspects && n_suspect);
    ARG_CHECK(P > 0 && nsimplex > 0 && nsimplex < (int64_t)INT32_MAX);
    unsigned* count = (unsigned*)oisat_ws(h, 1, 64);
    unsigned* count_host = (unsigned*)oisat_pinned(h, 64);
    if (!count || !count_host) return OISAT_ENOMEM;
    HIP_TRY(hipMemsetAsync(count, 0, 64, h->stream));
    OISAT_LAUNCH(h, "tri_transform", tri_transform_kernel, dim3((unsigned)cdiv(nsimplex, 256)), dim3(256), 0, points, simplices, nsimplex,
                 transform_out, suspects, count);
    HIP_TRY(hipMemcpyAsync(count_host, count, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *n_suspect = (int64_t)count_host[0];
    return OISAT_OK;
}

extern "C" int oisat_boxfilter_symm(oisat_ctx* h, int dtype, const void* Z, int64_t Ny, int64_t Nx, int ky, int kx, int variance,
                                    void* out) {
    ARG_CHECK(h && Z && out && Ny > 0 && Nx > 0 && ky > 0 && kx > 0);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    const double kk = (double)kx * (double)ky;
    const double w = variance ? 1.0 / (kk * kk) : 1.0 / kk;
    const int grid = stream_grid(Ny * Nx, 256);
    if (dtype == OISAT_F32) {
        OISAT_LAUNCH(h, "boxfilter_symm", (boxfilter_kernel<float>), dim3(grid), dim3(256), 0, (const float*)Z, Ny, Nx, ky, kx,
                     (float)w, (float*)out);
    } else {
        OISAT_LAUNCH(h, "boxfilter_symm", (boxfilter_kernel<double>), dim3(grid), dim3(256), 0, (const double*)Z, Ny, Nx, ky,
                     kx, w, (double*)out);
    }
    return OISAT_OK;
}

template <typename T>
static int pick_impl(oisat_ctx* h, const void* Z, int64_t Ny, int64_t Nx, int nfields, int ky, int kx, double w,
                     const int32_t* idx, int64_t Tn, void* out) {
    const int win = ky * kx;
    const int64_t groups = Tn * nfields;
    if (kx <= 64 && win >= 3) {                              // the row-wise form (12.5 vs 32.6 us: profiles/EXPERIMENTS.md); wider windows: lane-strided
        const int tpw = 64 / kx;
        const int64_t wpf = cdiv(Tn, tpw);
        int64_t blocks = cdiv(wpf * nfields, 4);
        if (blocks > 8192) blocks = 8192;
        blocks = cdiv(blocks, 8) * 8;
        OISAT_LAUNCH(h, "boxfilter_pick", (boxfilter_pick_rows_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, (const T*)Z, Ny, Nx,
                     nfields, ky, kx, (T)w, idx, Tn, (T*)out, tpw, wpf);
        return OISAT_OK;
    }
    if (win >= 48) {
        const int grid = stream_grid(groups * 64, 256);
        OISAT_LAUNCH(h, "boxfilter_pick", (boxfilter_pick_kernel<T, 64>), dim3(grid), dim3(256), 0, (const T*)Z, Ny, Nx, nfields,
                     ky, kx, (T)w, idx, Tn, (T*)out);
    } else if (win >= 12) {
        const int grid = stream_grid(groups * 16, 256);
        OISAT_LAUNCH(h, "boxfilter_pick", (boxfilter_pick_kernel<T, 16>), dim3(grid), dim3(256), 0, (const T*)Z, Ny, Nx, nfields,
                     ky, kx, (T)w, idx, Tn, (T*)out);
    } else if (win >= 3) {
        const int grid = stream_grid(groups * 4, 256);
        OISAT_LAUNCH(h, "boxfilter_pick", (boxfilter_pick_kernel<T, 4>), dim3(grid), dim3(256), 0, (const T*)Z, Ny, Nx, nfields,
                     ky, kx, (T)w, idx, Tn, (T*)out);
    } else {
        const int grid = stream_grid(groups, 256);
        OISAT_LAUNCH(h, "boxfilter_pick", (boxfilter_pick_kernel<T, 1>), dim3(grid), dim3(256), 0, (const T*)Z, Ny, Nx, nfields,
                     ky, kx, (T)w, idx, Tn, (T*)out);
    }
    return OISAT_OK;
}

extern "C" int oisat_boxfilter_pick(oisat_ctx* h, int dtype, const void* Z, int64_t Ny, int64_t Nx, int nfields, int ky, int kx,
                                    int variance, const int32_t* idx, int64_t Tn, void* out) {
    ARG_CHECK(h && Z && out && idx && Ny > 0 && Nx > 0 && ky > 0 && kx > 0 && nfields > 0 && Tn > 0);
    ARG_CHECK(Ny * Nx < (int64_t)INT32_MAX);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    const double kk = (double)kx * (double)ky;
    const double w = variance ? 1.0 / (kk * kk) : 1.0 / kk;
    if (dtype == OISAT_F32) return pick_impl<float>(h, Z, Ny, Nx, nfields, ky, kx, w, idx, Tn, out);
    return pick_impl<double>(h, Z, Ny, Nx, nfields, ky, kx, w, idx, Tn, out);
}

extern "C" int oisat_gather_mask(oisat_ctx* h, int dtype, const void* values, int64_t P, int nfields, const int32_t* idx,
                                 int64_t Tn, void* out) {
    ARG_CHECK(h && values && idx && out && P > 0 && nfields > 0 && Tn > 0);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    const int grid = stream_grid(Tn, 256);
    if (dtype == OISAT_F32) {
        OISAT_LAUNCH(h, "gather_mask", (gather_mask_kernel<float>), dim3(grid), dim3(256), 0, (const float*)values, P, nfields,
                     idx, Tn, (float*)out);
    } else {
        OISAT_LAUNCH(h, "gather_mask", (gather_mask_kernel<double>), dim3(grid), dim3(256), 0, (const double*)values, P,
                     nfields, idx, Tn, (double*)out);
    }
    return OISAT_OK;
}

extern "C" int oisat_flag_mask(oisat_ctx* h, int dtype, const void* x, const void* flag, int64_t n, double thresh, int square,
                               void* out) {
    ARG_CHECK(h && x && flag && out && n > 0);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    const int grid = stream_grid(n, 256);
    if (dtype == OISAT_F32) {
        OISAT_LAUNCH(h, "flag_mask", (flag_mask_kernel<float>), dim3(grid), dim3(256), 0, (const float*)x, (const float*)flag, n,
                     (float)thresh, square != 0, (float*)out);
    } else {
        OISAT_LAUNCH(h, "flag_mask", (flag_mask_kernel<double>), dim3(grid), dim3(256), 0, (const double*)x, (const double*)flag,
                     n, thresh, square != 0, (double*)out);
    }
    return OISAT_OK;
}

extern "C" int oisat_sqrt(oisat_ctx* h, int dtype, const void* x, int64_t n, void* out) {
    ARG_CHECK(h && x && out && n > 0);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    const int grid = stream_grid(n, 256);
    if (dtype == OISAT_F32) {
        OISAT_LAUNCH(h, "sqrt", (sqrt_kernel<float>), dim3(grid), dim3(256), 0, (const float*)x, n, (float*)out);
    } else {
        OISAT_LAUNCH(h, "sqrt", (sqrt_kernel<double>), dim3(grid), dim3(256), 0, (const double*)x, n, (double*)out);
    }
    return OISAT_OK;
}

// host-side min/max of a device coordinate array (sizes the hash grid)
__global__ __launch_bounds__(256) void minmax_kernel(const double* __restrict__ x, const double* __restrict__ y, int64_t n,
                                                      double* __restrict__ out /* [4*gridDim] */) {
    __shared__ double sm[4][256];
    double xmin = __builtin_inf(), xmax = -__builtin_inf(), ymin = __builtin_inf(), ymax = -__builtin_inf();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double a = x[i], b = y[i];
        if (a == a && b == b) {
            xmin = fmin(xmin, a); xmax = fmax(xmax, a);
            ymin = fmin(ymin, b); ymax = fmax(ymax, b);
        }
    }
    sm[0][threadIdx.x] = xmin; sm[1][threadIdx.x] = xmax; sm[2][threadIdx.x] = ymin; sm[3][threadIdx.x] = ymax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sm[0][threadIdx.x] = fmin(sm[0][threadIdx.x], sm[0][threadIdx.x + s]);
            sm[1][threadIdx.x] = fmax(sm[1][threadIdx.x], sm[1][threadIdx.x + s]);
            sm[2][threadIdx.x] = fmin(sm[2][threadIdx.x], sm[2][threadIdx.x + s]);
            sm[3][threadIdx.x] = fmax(sm[3][threadIdx.x], sm[3][threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int q = 0; q < 4; ++q) out[blockIdx.x * 4 + q] = sm[q][0];
}

// Hash the points into uniform cells of edge `cell` (coarsened if that would need too many cells).
// On return start[c]..start[c+1] index `sorted` (point ids of cell c); both live in workspace slot 2.
static int build_hash(oisat_ctx* h, const double* plon, const double* plat, int64_t P, double cell, HashGrid* g_out,
                      const unsigned** start_out, const int32_t** sorted_out, const double2** sxy_out = nullptr,
                      const unsigned** scan_error_out = nullptr, int max_side = 0) {
    // 1. bounding box of the points
    const int mm_blocks = 64;
    double* mm_dev = (double*)oisat_ws(h, 1, sizeof(double) * 4 * mm_blocks);
    double* mm_host = (double*)oisat_pinned(h, sizeof(double) * 4 * mm_blocks);
    if (!mm_dev || !mm_host) return OISAT_ENOMEM;
    OISAT_LAUNCH(h, "nn_minmax", minmax_kernel, dim3(mm_blocks), dim3(256), 0, plon, plat, P, mm_dev);
    HIP_TRY(hipMemcpyAsync(mm_host, mm_dev, sizeof(double) * 4 * mm_blocks, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
    for (int b = 0; b < mm_blocks; ++b) {
        xmin = fmin(xmin, mm_host[b * 4 + 0]); xmax = fmax(xmax, mm_host[b * 4 + 1]);
        ymin = fmin(ymin, mm_host[b * 4 + 2]); ymax = fmax(ymax, mm_host[b * 4 + 3]);
    }
    // (no finite point at all -> a 1x1 empty grid below: every target comes back -1 / +inf)
    // 2. hash grid
    const double spanx = (xmin <= xmax) ? xmax - xmin : 0.0, spany = (ymin <= ymax) ? ymax - ymin : 0.0;
    const int64_t max_cells = 4 * 1024 * 1024;
    while ((floor(spanx / cell) + 1.0) * (floor(spany / cell) + 1.0) > (double)max_cells) cell *= 2.0;
    if (max_side > 0) cell = fmax(cell, fmax(spanx, spany) / (max_side - 1));           // floor(span / cell) + 1 <= max_side
    HashGrid g;
    g.x0 = (xmin <= xmax) ? xmin : 0.0;
    g.y0 = (ymin <= ymax) ? ymin : 0.0;
    g.inv_h = 1.0 / cell;
    g.nbx = (int)floor(spanx / cell) + 1;
    g.nby = (int)floor(spany / cell) + 1;
    const int64_t ncell = (int64_t)g.nbx * g.nby;
    // workspace slot 2: counts | scan status (one word per tile of 4096) + ticket | start (ncell+1) | cursor | pcell (P) | sorted (P)
    const int64_t ntile = cdiv(ncell, 4096);
    const size_t o_counts = 0;
    const size_t o_status = (o_counts + sizeof(unsigned) * (ncell + 4) + 7) / 8 * 8;
    const size_t o_ticket = o_status + sizeof(unsigned long long) * (ntile + 1);
    const size_t o_start = o_ticket + 16;
    const size_t o_cursor = o_start + sizeof(unsigned) * (ncell + 4);
    const size_t o_pcell = o_cursor + sizeof(unsigned) * (ncell + 4);
    const size_t o_sorted = o_pcell + sizeof(int32_t) * (P + 4);
    const size_t o_sxy = (o_sorted + sizeof(int32_t) * (P + 4) + 15) / 16 * 16;          // coordinates in cell order
    const size_t total = o_sxy + sizeof(double2) * (P + 4);
    char* ws = (char*)oisat_ws(h, 2, total);
    if (!ws) return OISAT_ENOMEM;
    unsigned* counts = (unsigned*)(ws + o_counts);
    unsigned long long* status = (unsigned long long*)(ws + o_status);
    unsigned* ticket = (unsigned*)(ws + o_ticket);
    unsigned* start = (unsigned*)(ws + o_start);
    unsigned* cursor = (unsigned*)(ws + o_cursor);
    int32_t* pcell = (int32_t*)(ws + o_pcell);
    int32_t* sorted = (int32_t*)(ws + o_sorted);
    double2* sxy = (double2*)(ws + o_sxy);
    HIP_TRY(hipMemsetAsync(counts, 0, o_start, h->stream));             // counters, scan status words and ticket in one fill
    OISAT_LAUNCH(h, "nn_count", nn_count_kernel, dim3(stream_grid(P, 256)), dim3(256), 0, plon, plat, P, g, counts, pcell);
    OISAT_LAUNCH(h, "nn_scan", nn_scan_lookback_kernel, dim3((unsigned)ntile), dim3(1024), 0, (const unsigned*)counts, ncell, start,
                 cursor, status, ticket);
    OISAT_LAUNCH(h, "nn_scatter", nn_scatter_kernel, dim3(stream_grid(P, 256)), dim3(256), 0, (const int32_t*)pcell, P, cursor,
                 sorted, plon, plat, sxy);
    *g_out = g;
    *start_out = start;
    *sorted_out = sorted;
    if (sxy_out) *sxy_out = sxy;
    if (scan_error_out) *scan_error_out = ticket + 1;
    return OISAT_OK;
}

// after the caller's synchronisation: did the hash's counter scan give up on a predecessor (nn_scan_lookback_kernel)?
static int scan_error_check(const unsigned* host_word) {
    if (*host_word == 0u) return OISAT_OK;
    oisat_set_error("nearest-neighbour hash: the counter scan gave up waiting for a predecessor tile (bounded spin): no indices handed out");
    return OISAT_EHIP;
}

static int nn_query_impl(oisat_ctx* h, const double* plon, const double* plat, int64_t P, const double* tlon, const double* tlat,
                         int64_t Tn, double max_dist, int32_t* idx_out, double* dist_out, int32_t* tie_list, int64_t* n_ties) {
    ARG_CHECK(h && plon && plat && tlon && tlat && idx_out);
    ARG_CHECK(P > 0 && P < (int64_t)INT32_MAX && Tn > 0 && Tn < (int64_t)INT32_MAX && max_dist > 0.0 && std::isfinite(max_dist));
    // cell edge = mask radius: the 3x3 block around a target holds every candidate that can survive the mask
    HashGrid g;
    const unsigned* start;
    const int32_t* sorted;
    const double2* sxy;
    const unsigned* scan_error;
    const int rc = build_hash(h, plon, plat, P, max_dist, &g, &start, &sorted, &sxy, &scan_error);
    if (rc != OISAT_OK) return rc;
    unsigned* count = nullptr;
    unsigned* count_host = (unsigned*)oisat_pinned(h, 64);
    if (!count_host) return OISAT_ENOMEM;
    if (tie_list) {
        count = (unsigned*)oisat_ws(h, 1, 64);
        if (!count) return OISAT_ENOMEM;
        HIP_TRY(hipMemsetAsync(count, 0, 64, h->stream));
    }
    // latency-bound (dependent loads through the hash): one target per thread up to 16,384 blocks
    const int64_t qgrid = cdiv(Tn, 256) < 16384 ? cdiv(Tn, 256) : 16384;
    OISAT_LAUNCH(h, "nn_query", nn_query_kernel, dim3((unsigned)qgrid), dim3(256), 0, plon, plat, tlon, tlat, Tn, g, start,
                 sorted, sxy, max_dist, idx_out, dist_out, tie_list, count);
    // one small read-back per query (the callers download the indices next anyway): the tie count and the scan's error word
    if (tie_list) HIP_TRY(hipMemcpyAsync(count_host, count, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(count_host + 1, scan_error, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (tie_list) *n_ties = (int64_t)count_host[0];
    return scan_error_check(count_host + 1);
}

extern "C" int oisat_nn_query(oisat_ctx* h, const double* plon, const double* plat, int64_t P, const double* tlon,
                              const double* tlat, int64_t Tn, double max_dist, int32_t* idx_out, double* dist_out) {
    return nn_query_impl(h, plon, plat, P, tlon, tlat, Tn, max_dist, idx_out, dist_out, nullptr, nullptr);
}

extern "C" int oisat_nn_query_ties(oisat_ctx* h, const double* plon, const double* plat, int64_t P, const double* tlon,
                                   const double* tlat, int64_t Tn, double max_dist, int32_t* idx_out, double* dist_out,
                                   int32_t* tie_list, int64_t* n_ties) {
    ARG_CHECK(tie_list && n_ties);
    return nn_query_impl(h, plon, plat, P, tlon, tlat, Tn, max_dist, idx_out, dist_out, tie_list, n_ties);
}

template <typename T>
static int rbf_launch(oisat_ctx* h, int K, const double* plon, const double* plat, const double* tlon, const double* tlat,
                      int64_t Tn, const int32_t* nn_idx, HashGrid g, const unsigned* start, const int32_t* sorted, const void* values,
                      int64_t P, int nfields, void* out, int* flag, int32_t* tie_list, unsigned* n_ties) {
    const dim3 grid((unsigned)cdiv(Tn, 64)), block(64);
    if (K == 5) {
        OISAT_LAUNCH(h, "rbf_interp", (rbf_interp_kernel<T, 5>), grid, block, 0, plon, plat, tlon, tlat, Tn, nn_idx, g, start, sorted,
                     (const T*)values, P, nfields, (T*)out, flag, tie_list, n_ties);
    } else if (K == 4) {
        OISAT_LAUNCH(h, "rbf_interp", (rbf_interp_kernel<T, 4>), grid, block, 0, plon, plat, tlon, tlat, Tn, nn_idx, g, start, sorted,
                     (const T*)values, P, nfields, (T*)out, flag, tie_list, n_ties);
    } else {
        OISAT_LAUNCH(h, "rbf_interp", (rbf_interp_kernel<T, 3>), grid, block, 0, plon, plat, tlon, tlat, Tn, nn_idx, g, start, sorted,
                     (const T*)values, P, nfields, (T*)out, flag, tie_list, n_ties);
    }
    return OISAT_OK;
}

template <typename T>
static int rbf_forced_launch(oisat_ctx* h, int K, const double* plon, const double* plat, const double* tlon, const double* tlat,
                             int64_t Tn, const int32_t* targets, const int32_t* ids, int64_t n, const void* values, int64_t P,
                             int nfields, void* out, int* flag) {
    const dim3 grid((unsigned)cdiv(n, 64)), block(64);
    if (K == 5) {
        OISAT_LAUNCH(h, "rbf_forced", (rbf_forced_kernel<T, 5>), grid, block, 0, plon, plat, tlon, tlat, Tn, targets, ids, n,
                     (const T*)values, P, nfields, (T*)out, flag);
    } else if (K == 4) {
        OISAT_LAUNCH(h, "rbf_forced", (rbf_forced_kernel<T, 4>), grid, block, 0, plon, plat, tlon, tlat, Tn, targets, ids, n,
                     (const T*)values, P, nfields, (T*)out, flag);
    } else {
        OISAT_LAUNCH(h, "rbf_forced", (rbf_forced_kernel<T, 3>), grid, block, 0, plon, plat, tlon, tlat, Tn, targets, ids, n,
                     (const T*)values, P, nfields, (T*)out, flag);
    }
    return OISAT_OK;
}

// The singular-neighbourhood counter (+ the tie counter behind it) and their host copy live behind the 2 KB build_hash
// uses of the same two buffers.
static int rbf_counter(oisat_ctx* h, int** flag, int** flag_host, unsigned** table) {
    char* ws1 = (char*)oisat_ws(h, 1, 8192);
    char* pin = (char*)oisat_pinned(h, 4096);
    if (!ws1 || !pin) return OISAT_ENOMEM;
    *flag = (int*)(ws1 + 2048);
    *flag_host = (int*)(pin + 2048);
    *table = (unsigned*)(ws1 + 4096);                    // 1024 block counters
    HIP_TRY(hipMemsetAsync(*flag, 0, 64, h->stream));
    return OISAT_OK;
}

static int rbf_interp_impl(oisat_ctx* h, int dtype, const double* plon, const double* plat, int64_t P, const double* tlon,
                           const double* tlat, int64_t Tn, const int32_t* nn_idx, double cell, int neighbors, const void* values,
                           int nfields, void* out, int64_t* n_singular, int32_t* tie_list, int64_t* n_ties) {
    ARG_CHECK(h && plon && plat && tlon && tlat && nn_idx && values && out);
    ARG_CHECK(P >= 3 && P < (int64_t)INT32_MAX && Tn > 0 && nfields > 0 && cell > 0.0 && std::isfinite(cell));
    ARG_CHECK(neighbors >= 3 && neighbors <= 5 && neighbors <= P);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    HashGrid g;
    const unsigned* start;
    const int32_t* sorted;
    const unsigned* scan_error;
    int rc = build_hash(h, plon, plat, P, cell, &g, &start, &sorted, nullptr, &scan_error);
    if (rc != OISAT_OK) return rc;
    int *flag, *flag_host;
    unsigned* table;
    rc = rbf_counter(h, &flag, &flag_host, &table);
    if (rc != OISAT_OK) return rc;
    unsigned* ties_dev = (unsigned*)(flag + 1);
    if (dtype == OISAT_F32)
        rc = rbf_launch<float>(h, neighbors, plon, plat, tlon, tlat, Tn, nn_idx, g, start, sorted, values, P, nfields, out, flag,
                               tie_list, ties_dev);
    else
        rc = rbf_launch<double>(h, neighbors, plon, plat, tlon, tlat, Tn, nn_idx, g, start, sorted, values, P, nfields, out, flag,
                                tie_list, ties_dev);
    if (rc != OISAT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(flag_host, flag, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(flag_host + 2, scan_error, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (n_singular) *n_singular = flag_host[0];
    if (n_ties) *n_ties = (int64_t)(unsigned)flag_host[1];
    return scan_error_check((const unsigned*)(flag_host + 2));
}

extern "C" int oisat_rbf_interp(oisat_ctx* h, int dtype, const double* plon, const double* plat, int64_t P, const double* tlon,
                                const double* tlat, int64_t Tn, const int32_t* nn_idx, double cell, int neighbors,
                                const void* values, int nfields, void* out, int64_t* n_singular) {
    return rbf_interp_impl(h, dtype, plon, plat, P, tlon, tlat, Tn, nn_idx, cell, neighbors, values, nfields, out, n_singular,
                           nullptr, nullptr);
}

extern "C" int oisat_rbf_interp_ties(oisat_ctx* h, int dtype, const double* plon, const double* plat, int64_t P, const double* tlon,
                                     const double* tlat, int64_t Tn, const int32_t* nn_idx, double cell, int neighbors,
                                     const void* values, int nfields, void* out, int64_t* n_singular, int32_t* tie_list,
                                     int64_t* n_ties) {
    ARG_CHECK(tie_list && n_ties);
    return rbf_interp_impl(h, dtype, plon, plat, P, tlon, tlat, Tn, nn_idx, cell, neighbors, values, nfields, out, n_singular,
                           tie_list, n_ties);
}

extern "C" int oisat_rbf_interp_forced(oisat_ctx* h, int dtype, const double* plon, const double* plat, int64_t P,
                                       const double* tlon, const double* tlat, int64_t Tn, const int32_t* targets,
                                       const int32_t* ids, int64_t n, int neighbors, const void* values, int nfields, void* out,
                                       int64_t* n_singular) {
    ARG_CHECK(h && plon && plat && tlon && tlat && targets && ids && values && out);
    ARG_CHECK(P >= 3 && P < (int64_t)INT32_MAX && Tn > 0 && n > 0 && n <= Tn && nfields > 0);
    ARG_CHECK(neighbors >= 3 && neighbors <= 5 && neighbors <= P);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    int *flag, *flag_host;
    unsigned* table;
    int rc = rbf_counter(h, &flag, &flag_host, &table);
    if (rc != OISAT_OK) return rc;
    if (dtype == OISAT_F32)
        rc = rbf_forced_launch<float>(h, neighbors, plon, plat, tlon, tlat, Tn, targets, ids, n, values, P, nfields, out, flag);
    else
        rc = rbf_forced_launch<double>(h, neighbors, plon, plat, tlon, tlat, Tn, targets, ids, n, values, P, nfields, out, flag);
    if (rc != OISAT_OK) return rc;
    HIP_TRY(hipMemcpyAsync(flag_host, flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (n_singular) *n_singular = flag_host[0];
    return OISAT_OK;
}

extern "C" int oisat_rbf_check_masked(oisat_ctx* h, const double* plon, const double* plat, int64_t P, const double* tlon,
                                      const double* tlat, int64_t Tn, const int32_t* nn_idx, double cell, int neighbors,
                                      int64_t* n_singular) {
    ARG_CHECK(h && plon && plat && tlon && tlat && nn_idx && n_singular);
    ARG_CHECK(P >= 3 && P < (int64_t)INT32_MAX && Tn > 0 && cell > 0.0 && std::isfinite(cell));
    ARG_CHECK(neighbors >= 3 && neighbors <= 5 && neighbors <= P);
    HashGrid g;
    const unsigned* start;
    const int32_t* sorted;
    const unsigned* scan_error;
    // a hash of at most 128 cells a side: 16 x 16 blocks of 8 x 8 cells whatever the extent of the points
    int rc = build_hash(h, plon, plat, P, cell, &g, &start, &sorted, nullptr, &scan_error, 128);
    if (rc != OISAT_OK) return rc;
    int *flag, *flag_host;
    unsigned* table;
    rc = rbf_counter(h, &flag, &flag_host, &table);
    if (rc != OISAT_OK) return rc;
    const int nsx = (int)cdiv(g.nbx, kSuper), nsy = (int)cdiv(g.nby, kSuper);
    if (nsx * nsy > 1024) {
        oisat_set_error("rbf_check_masked: %d x %d blocks", nsx, nsy);
        return OISAT_EINVAL;
    }
    OISAT_LAUNCH(h, "rbf_super_count", rbf_super_count_kernel, dim3((unsigned)cdiv(nsx * nsy, 256)), dim3(256), 0, g, start, nsx, nsy,
                 table);
    const dim3 grid((unsigned)cdiv(Tn, 64)), block(64);
    if (neighbors == 5) {
        OISAT_LAUNCH(h, "rbf_far_check", rbf_far_check_kernel<5>, grid, block, 0, plon, plat, tlon, tlat, Tn, nn_idx, g, start, sorted,
                     (const unsigned*)table, nsx, nsy, flag);
    } else if (neighbors == 4) {
        OISAT_LAUNCH(h, "rbf_far_check", rbf_far_check_kernel<4>, grid, block, 0, plon, plat, tlon, tlat, Tn, nn_idx, g, start, sorted,
                     (const unsigned*)table, nsx, nsy, flag);
    } else {
        OISAT_LAUNCH(h, "rbf_far_check", rbf_far_check_kernel<3>, grid, block, 0, plon, plat, tlon, tlat, Tn, nn_idx, g, start, sorted,
                     (const unsigned*)table, nsx, nsy, flag);
    }
    HIP_TRY(hipMemcpyAsync(flag_host, flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(flag_host + 1, scan_error, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *n_singular = flag_host[0];
    return scan_error_check((const unsigned*)(flag_host + 1));
}

static int linear_impl(oisat_ctx* h, int dtype, const double* tlon, const double* tlat, int64_t Tn, const int32_t* nn_idx,
                       const int32_t* vertex_to_simplex, const int32_t* simplices, const int32_t* neighbors, const double* transform,
                       int64_t nsimplex, const void* values, int64_t P, int nfields, void* out, const double* bounds_host,
                       const int32_t* forced, int32_t* amb_list, int64_t* n_amb) {
    ARG_CHECK(h && tlon && tlat && nn_idx && vertex_to_simplex && simplices && neighbors && transform);
    ARG_CHECK(nfields == 0 || (values && out));
    TriBounds bb = {-1e300, 1e300, -1e300, 1e300};                    // no bounds given: never "fully outside"
    if (bounds_host) bb = TriBounds{bounds_host[0], bounds_host[1], bounds_host[2], bounds_host[3]};
    ARG_CHECK(Tn > 0 && Tn < (int64_t)INT32_MAX && nsimplex > 0 && nsimplex < (int64_t)INT32_MAX && P > 0 && nfields >= 0);
    ARG_CHECK(dtype == OISAT_F32 || dtype == OISAT_F64);
    unsigned* count = nullptr;
    unsigned* count_host = nullptr;
    if (amb_list) {
        count = (unsigned*)oisat_ws(h, 1, 64);
        count_host = (unsigned*)oisat_pinned(h, 64);
        if (!count || !count_host) return OISAT_ENOMEM;
        HIP_TRY(hipMemsetAsync(count, 0, 64, h->stream));
    }
    const int grid = stream_grid(Tn, 256);
    if (dtype == OISAT_F32) {
        OISAT_LAUNCH(h, "linear_interp", (linear_interp_kernel<float>), dim3(grid), dim3(256), 0, tlon, tlat, Tn, nn_idx,
                     vertex_to_simplex, simplices, neighbors, transform, (int32_t)nsimplex, (const float*)values, P, nfields, (float*)out, bb,
                     forced, amb_list, count);
    } else {
        OISAT_LAUNCH(h, "linear_interp", (linear_interp_kernel<double>), dim3(grid), dim3(256), 0, tlon, tlat, Tn, nn_idx,
                     vertex_to_simplex, simplices, neighbors, transform, (int32_t)nsimplex, (const double*)values, P, nfields,
                     (double*)out, bb, forced, amb_list, count);
    }
    if (amb_list) {
        HIP_TRY(hipMemcpyAsync(count_host, count, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        *n_amb = (int64_t)count_host[0];
    }
    return OISAT_OK;
}

extern "C" int oisat_linear_interp(oisat_ctx* h, int dtype, const double* tlon, const double* tlat, int64_t Tn, const int32_t* nn_idx,
                                   const int32_t* vertex_to_simplex, const int32_t* simplices, const int32_t* neighbors,
                                   const double* transform, int64_t nsimplex, const void* values, int64_t P, int nfields, void* out,
                                   const double* bounds_host) {
    ARG_CHECK(nfields > 0);
    return linear_impl(h, dtype, tlon, tlat, Tn, nn_idx, vertex_to_simplex, simplices, neighbors, transform, nsimplex, values, P, nfields,
                       out, bounds_host, nullptr, nullptr, nullptr);
}

extern "C" int oisat_linear_locate(oisat_ctx* h, const double* tlon, const double* tlat, int64_t Tn, const int32_t* nn_idx,
                                   const int32_t* vertex_to_simplex, const int32_t* simplices, const int32_t* neighbors,
                                   const double* transform, int64_t nsimplex, int64_t P, const double* bounds_host, int32_t* amb_list,
                                   int64_t* n_amb) {
    ARG_CHECK(amb_list && n_amb);
    return linear_impl(h, OISAT_F64, tlon, tlat, Tn, nn_idx, vertex_to_simplex, simplices, neighbors, transform, nsimplex, nullptr, P, 0,
                       nullptr, bounds_host, nullptr, amb_list, n_amb);
}

extern "C" int oisat_linear_interp_forced(oisat_ctx* h, int dtype, const double* tlon, const double* tlat, int64_t Tn,
                                          const int32_t* nn_idx, const int32_t* vertex_to_simplex, const int32_t* simplices,
                                          const int32_t* neighbors, const double* transform, int64_t nsimplex, const void* values,
                                          int64_t P, int nfields, void* out, const double* bounds_host, const int32_t* forced) {
    ARG_CHECK(nfields > 0 && forced);
    return linear_impl(h, dtype, tlon, tlat, Tn, nn_idx, vertex_to_simplex, simplices, neighbors, transform, nsimplex, values, P, nfields,
                       out, bounds_host, forced, nullptr, nullptr);
}
