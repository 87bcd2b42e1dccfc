/* Plain C99 client of the C-ABI (include/oisat.h): proves the boundary is usable without C++ or Python.
 * Element-wise OI for one scaling (optimal_interpolation.py:27-31,:49-52) on a handful of cells. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "oisat.h"

#define CHECK(x)                                                            \
    do {                                                                    \
        int rc_ = (x);                                                      \
        if (rc_ != OISAT_OK) {                                              \
            fprintf(stderr, "%s -> %d: %s\n", #x, rc_, oisat_last_error()); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

int main(void) {
    enum { N = 8 };
    double Xa[N] = {1, 2, 3, 4, 5, 6, 7, 8}, Y[N] = {1.5, -1, 2.5, NAN, 6, 5, 9, 7};
    double Sa[N], So[N] = {0.1, 0.2, 0.3, NAN, 0.5, 0.6, 0.7, 0.8}, out[4][N];
    for (int i = 0; i < N; ++i) Sa[i] = (Xa[i] * 50.0 / 100.0) * (Xa[i] * 50.0 / 100.0);
    oisat_ctx* h = NULL;
    CHECK(oisat_init(0, &h));
    void *d[8];
    for (int i = 0; i < 8; ++i) CHECK(oisat_dmalloc(h, sizeof(double) * N, &d[i]));
    CHECK(oisat_h2d(h, d[0], Xa, sizeof(Xa)));
    CHECK(oisat_h2d(h, d[1], Y, sizeof(Y)));
    CHECK(oisat_h2d(h, d[2], Sa, sizeof(Sa)));
    CHECK(oisat_h2d(h, d[3], So, sizeof(So)));
    CHECK(oisat_sync(h));
    CHECK(oisat_oi_apply(h, OISAT_F64, d[0], d[1], d[2], d[3], N, 1.0, d[4], d[5], d[6], d[7]));
    for (int i = 0; i < 4; ++i) CHECK(oisat_d2h(h, out[i], d[4 + i], sizeof(double) * N));
    CHECK(oisat_d2h(h, Y, d[1], sizeof(Y)));
    int bad = 0;
    for (int i = 0; i < N; ++i) {
        double y = Y[i], t = Sa[i] * 1.0, k = t * (1.0 / (t + So[i])), sb = (1.0 - k) * t;
        double want[4] = {Xa[i] + k * (y - Xa[i]), 1.0 - sb / t, k * (y - Xa[i]), sqrt(sb)};
        for (int f = 0; f < 4; ++f) {
            int both_nan = isnan(want[f]) && isnan(out[f][i]);
            if (!both_nan && fabs(out[f][i] - want[f]) > 1e-14 * fabs(want[f])) { ++bad; printf("mismatch cell %d field %d: %g vs %g\n", i, f, out[f][i], want[f]); }
        }
    }
    if (Y[1] != 0.0) { ++bad; printf("Y was not clamped in place\n"); }
    /* error behaviour: negative status + message */
    if (oisat_oi_apply(h, 9, d[0], d[1], d[2], d[3], N, 1.0, d[4], NULL, NULL, NULL) != OISAT_EINVAL) { ++bad; printf("bad dtype accepted\n"); }
    for (int i = 0; i < 8; ++i) CHECK(oisat_dfree(h, d[i]));
    oisat_shutdown(h);
    if (bad) printf("C ABI smoke FAILED (%d)\n", bad);
    else printf("C ABI smoke ok %s\n", oisat_version());
    return bad ? 1 : 0;
}
