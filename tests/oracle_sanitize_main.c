/* Test harness (tests/test_oracle_sanitizers.py): every entry point of oracle/gp_oracle.c on ragged shapes, padded leading
 * dimensions and tight heap buffers, built with -fsanitize=address,undefined.  A read or write one element past a buffer,
 * a signed overflow or a misaligned access ends the program non-zero.  Results are checked only for finiteness and for the
 * factor / solve identity: the values themselves are pinned elsewhere (tests/test_oracle.py, golden vectors). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

int oracle_gaussian(int which, double *out, long ld, const double *x1, long n, const double *x2, long m, int d, double h, double w);
int oracle_periodic(int which, double *out, long ld, const double *x1, long n, const double *x2, long m, int d, double h, double w, double p);
void oracle_add_diag(double *K, long n, long ld, double s);
int oracle_potrf_lower(double *A, long n, long ld);
void oracle_potrs_lower(const double *L, long n, long ld, double *b);

static double rnd(unsigned *s) { *s = *s * 1664525u + 1013904223u; return ((double)(*s >> 8) / 16777216.0) * 6.0 - 3.0; }

int main(void)
{
    unsigned seed = 12345u;
    const long shapes[][3] = {{1, 1, 1}, {3, 5, 2}, {17, 9, 1}, {33, 40, 7}, {64, 1, 3}};
    for (unsigned t = 0; t < sizeof(shapes) / sizeof(shapes[0]); ++t) {
        const long n = shapes[t][0], m = shapes[t][1];
        const int d = (int)shapes[t][2];
        const long ld = m + (t % 2);                                  /* a padded leading dimension every other case */
        double *x1 = malloc(sizeof(double) * n * d), *x2 = malloc(sizeof(double) * m * d), *out = malloc(sizeof(double) * n * ld);
        if (!x1 || !x2 || !out) return 2;
        for (long i = 0; i < n * d; ++i) x1[i] = rnd(&seed);
        for (long i = 0; i < m * d; ++i) x2[i] = rnd(&seed);
        for (int which = 0; which <= 5; ++which) {
            if (oracle_gaussian(which, out, ld, x1, n, x2, m, d, 1.1, 0.7) != 0) return 3;
            for (long i = 0; i < n; ++i) for (long j = 0; j < m; ++j) if (!isfinite(out[i * ld + j])) return 4;
        }
        for (int which = 0; which <= 9; ++which) {
            const int rc = oracle_periodic(which, out, ld, x1, n, x2, m, d, 1.1, 0.8, 2.3);
            if (d == 1 || which == 0) {
                if (rc != 0) return 5;
                for (long i = 0; i < n; ++i) for (long j = 0; j < m; ++j) if (!isfinite(out[i * ld + j])) return 6;
            } else if (rc == 0) return 7;                             /* derivative members are 1-D only: must refuse */
        }
        free(x1); free(x2); free(out);
    }
    /* K + s^2 I -> factor -> solve, tight buffers, ld > n */
    for (long n = 1; n <= 37; n += 9) {
        const long ld = n + 3;
        double *x = malloc(sizeof(double) * n), *K = malloc(sizeof(double) * n * ld), *K0 = malloc(sizeof(double) * n * ld);
        double *b = malloc(sizeof(double) * n), *b0 = malloc(sizeof(double) * n);
        if (!x || !K || !K0 || !b || !b0) return 2;
        for (long i = 0; i < n; ++i) { x[i] = rnd(&seed); b[i] = b0[i] = rnd(&seed); }
        if (oracle_gaussian(0, K, ld, x, n, x, n, 1, 1.0, 0.5) != 0) return 8;
        oracle_add_diag(K, n, ld, 0.7);
        for (long i = 0; i < n; ++i) for (long j = 0; j < n; ++j) K0[i * ld + j] = K[i * ld + j];
        if (oracle_potrf_lower(K, n, ld) != 0) return 9;
        oracle_potrs_lower(K, n, ld, b);
        for (long i = 0; i < n; ++i) {                                /* K0 b = b0 */
            double v = 0.0;
            for (long j = 0; j < n; ++j) v += K0[i * ld + j] * b[j];
            if (fabs(v - b0[i]) > 1e-9 * (1.0 + fabs(b0[i]))) return 10;
        }
        K0[0] = -1.0;                                                 /* not positive definite: info = 1, nothing written out of bounds */
        if (oracle_potrf_lower(K0, n, ld) != 1) return 11;
        free(x); free(K); free(K0); free(b); free(b0);
    }
    puts("oracle sanitizer run ok");
    return 0;
}
