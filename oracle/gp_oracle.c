/*
 * gp_oracle.c -- CPU restatement of the reference's native kernel-matrix loops.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (gaussian_processes_amd/)
 * may link, load or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every function below
 * against golden vectors generated from the real reference (built and imported
 * in the authoring container by oracle/make_golden.py) -- bit-for-bit at d = 1.
 *
 * Each function cites the reference lines it restates (paths relative to
 * /root/reference).  The reference is strictly 1-D (x: shape (n,)); here inputs
 * are (n, d) row-major and d = 1 reproduces the reference exactly.  For d > 1
 * the squared distance is r2 = sum_k (x1[i,k] - x2[j,k])^2, accumulated in
 * k order, used wherever the reference uses (x1[i] - x2[j])**2.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC gp_oracle.c -lm -o libgp_oracle.so
 * (single thread, like the reference: no nogil/prange anywhere in gp/ext).
 */
#include <math.h>
#include <stddef.h>

/* gp/ext/gaussian_c.pyx:14-15 ; gp/ext/gp_c.pyx:14
 * MIN = log(2^(minexp+4)) = log(2^-1018) */
#define ORACLE_MIN (-705.6238298100243)

static double sqrt_2_div_pi(void) { return sqrt(2.0 / M_PI); }

static double sqdist(const double *a, const double *b, int d)
{
    double r2 = 0.0;
    for (int k = 0; k < d; ++k) {
        double t = a[k] - b[k];
        r2 = r2 + t * t;
    }
    return r2;
}

/* which: 0 K            gaussian_c.pyx:18-36
 *        1 dK_dh        gaussian_c.pyx:51-69
 *        2 dK_dw        gaussian_c.pyx:72-92
 *        3 d2K_dhdh     gaussian_c.pyx:95-113
 *        4 d2K_dhdw     gaussian_c.pyx:116-136  (== d2K_dwdh, :139-140)
 *        5 d2K_dwdw     gaussian_c.pyx:143-164
 * out is (n, m) row-major with leading dimension ld. */
int oracle_gaussian(int which, double *out, long ld, const double *x1, long n,
                    const double *x2, long m, int d, double h, double w)
{
    const double S = sqrt_2_div_pi();
    const double h2 = h * h;
    const double w2 = w * w;
    const double c1 = -0.5 / w2;
    double c2 = 0, c3 = 0, c4 = 0;
    switch (which) {
    case 0: c2 = 0.5 * S * h2 / w; break;
    case 1: c2 = S * h / w; break;
    case 2: c2 = 0.5 * S * h2 / w2; c3 = 0.5 * S * h2 / pow(w, 4); break;
    case 3: c2 = S / w; break;
    case 4: c2 = S * h / w2; c3 = S * h / pow(w, 4); break;
    case 5: c2 = S * h2 / pow(w, 3); c3 = 2.5 * S * h2 / pow(w, 5);
            c4 = 0.5 * S * h2 / pow(w, 7); break;
    default: return -1;
    }
    for (long i = 0; i < n; ++i) {
        for (long j = 0; j < m; ++j) {
            double d2 = sqdist(x1 + i * d, x2 + j * d, d);
            double e = c1 * d2;
            double v;
            if (e < ORACLE_MIN) {
                v = 0.0;
            } else {
                switch (which) {
                case 0: case 1: case 3: v = c2 * exp(e); break;
                case 2: case 4: v = exp(e) * (c3 * d2 - c2); break;
                default: v = exp(e) * (c4 * (d2 * d2) - c3 * d2 + c2); break;
                }
            }
            out[i * ld + j] = v;
        }
    }
    return 0;
}

/* Periodic kernel, gp/ext/periodic_c.pyx.  No underflow clamp in the reference.
 * which: 0 K (:18-30)  1 dK_dh (:53-65)  2 dK_dw (:68-80)  3 dK_dp (:83-96)
 *        4 d2K_dhdh (:99-111) 5 d2K_dhdw (:114-126) 6 d2K_dhdp (:129-142)
 *        7 d2K_dwdw (:160-172) 8 d2K_dwdp (:175-188) 9 d2K_dpdp (:223-235)
 * (d2K_dwdh :145-157 == 5, d2K_dpdh :191-204 == 6, d2K_dpdw :207-220 == 8.)
 * For d > 1 only which == 0 is defined: the exponent uses
 * sum_k sin(0.5*(x1[i,k]-x2[j,k])/p)^2. */
/* sin-only members (reference expressions contain no cos) */
static double periodic_sin_only(int which, double dd, double h, double w, double p)
{
    const double h2 = h * h;
    const double w2 = w * w;
    double sn = sin(0.5 * dd / p);
    double ex = exp(-2.0 * (sn * sn) / w2);
    switch (which) {
    case 0: return h2 * ex;
    case 1: return 2.0 * h * ex;
    case 2: return 4.0 * h2 * ex * (sn * sn) / pow(w, 3);
    case 4: return 2.0 * ex;
    case 5: return 8.0 * h * ex * (sn * sn) / pow(w, 3);
    default: /* 7 */
        return -12.0 * h2 * ex * (sn * sn) / pow(w, 4)
               + 16.0 * h2 * ex * pow(sn, 4) / pow(w, 6);
    }
}

/* members whose reference expression has both sin and cos of the same argument
 * (gcc fuses the pair into sincos() in the reference build and here alike) */
static double periodic_sin_cos(int which, double dd, double h, double w, double p)
{
    const double h2 = h * h;
    const double w2 = w * w;
    const double p2 = p * p;
    double sn = sin(0.5 * dd / p);
    double cs = cos(0.5 * dd / p);
    double ex = exp(-2.0 * (sn * sn) / w2);
    switch (which) {
    case 3: return 2.0 * dd * h2 * ex * sn * cs / (p2 * w2);
    case 6: return 4.0 * dd * h * ex * sn * cs / (p2 * w2);
    case 8: return -4.0 * dd * h2 * ex * sn * cs / (p2 * pow(w, 3))
                   + 8.0 * dd * h2 * ex * pow(sn, 3) * cs / (p2 * pow(w, 5));
    default: /* 9 */
        return (dd * dd) * h2 * ex * (sn * sn) / (pow(p, 4) * w2)
               - 1.0 * (dd * dd) * h2 * ex * (cs * cs) / (pow(p, 4) * w2)
               + 4.0 * (dd * dd) * h2 * ex * (sn * sn) * (cs * cs) / (pow(p, 4) * pow(w, 4))
               - 4.0 * dd * h2 * ex * sn * cs / (pow(p, 3) * w2);
    }
}

int oracle_periodic(int which, double *out, long ld, const double *x1, long n,
                    const double *x2, long m, int d, double h, double w, double p)
{
    const double h2 = h * h;
    const double w2 = w * w;
    if (which < 0 || which > 9) return -1;
    if (d != 1 && which != 0) return -2;
    const int has_cos = (which == 3 || which == 6 || which >= 8);
    for (long i = 0; i < n; ++i) {
        for (long j = 0; j < m; ++j) {
            double v;
            if (d == 1) {
                double dd = x1[i] - x2[j];
                v = has_cos ? periodic_sin_cos(which, dd, h, w, p)
                            : periodic_sin_only(which, dd, h, w, p);
            } else {
                double s2 = 0.0;
                for (int k = 0; k < d; ++k) {
                    double sn = sin(0.5 * (x1[i * d + k] - x2[j * d + k]) / p);
                    s2 = s2 + sn * sn;
                }
                v = h2 * exp(-2.0 * s2 / w2);
            }
            out[i * ld + j] = v;
        }
    }
    return 0;
}

/* gp/gp.py:265  K += eye(n) * s**2  (in place on the diagonal) */
void oracle_add_diag(double *K, long n, long ld, double s)
{
    const double s2 = s * s;
    for (long i = 0; i < n; ++i) K[i * ld + i] += s2;
}

/* Plain unblocked lower Cholesky (row-major, in place), strict upper zeroed.
 * Stand-in for LAPACK dpotrf reached through scipy.linalg.cholesky
 * (gp/gp.py:294) for hosts without SciPy; returns LAPACK-style info:
 * 0 ok, j+1 when the (j+1)-th leading minor is not positive definite. */
int oracle_potrf_lower(double *A, long n, long ld)
{
    for (long j = 0; j < n; ++j) {
        double ajj = A[j * ld + j];
        for (long k = 0; k < j; ++k) ajj -= A[j * ld + k] * A[j * ld + k];
        if (!(ajj > 0.0)) return (int)(j + 1);
        ajj = sqrt(ajj);
        A[j * ld + j] = ajj;
        for (long i = j + 1; i < n; ++i) {
            double v = A[i * ld + j];
            for (long k = 0; k < j; ++k) v -= A[i * ld + k] * A[j * ld + k];
            A[i * ld + j] = v / ajj;
        }
        for (long k = j + 1; k < n; ++k) A[j * ld + k] = 0.0;
    }
    return 0;
}

/* cho_solve((L, True), b)  gp/gp.py:332-334 : forward then back substitution. */
void oracle_potrs_lower(const double *L, long n, long ld, double *b)
{
    for (long i = 0; i < n; ++i) {
        double v = b[i];
        for (long k = 0; k < i; ++k) v -= L[i * ld + k] * b[k];
        b[i] = v / L[i * ld + i];
    }
    for (long i = n - 1; i >= 0; --i) {
        double v = b[i];
        for (long k = i + 1; k < n; ++k) v -= L[k * ld + i] * b[k];
        b[i] = v / L[i * ld + i];
    }
}
