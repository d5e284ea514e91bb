/*
 * oracle/oracle_kernels.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, strict IEEE, no fast-math) of the likelihood
 * kernels on mc3's chain-update hot path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product path
 * (mc3_amd/) never does.
 *
 * Every function cites the reference lines it follows (paths relative to the
 * mc3 v3.2.1 tree).  Parity status: PINNED -- tests/test_oracle_golden.py checks
 * these functions against the reference's own known-answer tests
 * (tests/test_stats.py:115-171, 280-302) and against vectors generated from the
 * reference itself (tests/golden/make_golden.py).
 *
 * Arrays are contiguous float64; the reference honours arbitrary strides
 * through INDd (src_c/include/ind.h:7), the host wrappers here require
 * contiguity instead (SURVEY.md section 8b "Input contract").
 */
#include <math.h>
#include <stddef.h>

/* ---- priors(): src_c/include/stats.h:90-109 -------------------------------
 * off[i] = param - prior for the Gaussian-prior parameters only (the mask
 * priorlow>0 & priorup>0 is applied by the caller, mc3/stats/stats.py:211-216).
 * The Jeffreys branch (priorlow == -1) is kept for completeness although that
 * mask makes it unreachable from ms.chisq. */
double oracle_priors(const double *off, const double *plow, const double *pup,
                     int np)
{
    double total = 0.0;
    for (int i = 0; i < np; i++) {
        if (plow[i] == -1.0) {
            total += 2.0 * log(off[i]);
        } else {
            double s = (off[i] > 0.0) ? off[i] / pup[i] : off[i] / plow[i];
            total += s * s;
        }
    }
    return total;
}

/* ---- chisq(): src_c/_chisq.c:111-140 ---------------------------------------
 * sum_i ((model-data)/uncert)^2 in index order, then the prior term. */
double oracle_chisq(const double *model, const double *data,
                    const double *uncert, int n,
                    const double *off, const double *plow, const double *pup,
                    int np)
{
    double total = 0.0;
    for (int i = 0; i < n; i++) {
        double r = (model[i] - data[i]) / uncert[i];
        total += r * r;
    }
    if (off != NULL && np > 0)
        total += oracle_priors(off, plow, pup, np);
    return total;
}

/* ---- residuals(): src_c/_chisq.c:37-79 -------------------------------------
 * out has n + np entries: weighted data residuals, then prior residuals. */
void oracle_residuals(const double *model, const double *data,
                      const double *uncert, int n,
                      const double *off, const double *plow, const double *pup,
                      int np, double *out)
{
    for (int i = 0; i < n; i++)
        out[i] = (model[i] - data[i]) / uncert[i];
    if (off == NULL)
        return;
    for (int i = 0; i < np; i++)
        out[n + i] = (off[i] > 0.0) ? off[i] / pup[i] : off[i] / plow[i];
}

/* ---- one Daubechies-4 filter step: src_c/include/wavelet.h:16-51 ----------
 * Forward (isign >= 0): smooth coefficients to a[0..nh), detail to a[nh..n),
 * last pair wraps around.  Inverse: the transpose filter.  tmp: n doubles. */
static const double D4_C0 = 0.4829629131445341;
static const double D4_C1 = 0.83651630373780772;
static const double D4_C2 = 0.22414386804201339;
static const double D4_C3 = -0.12940952255126034;

static void d4_step(double *a, int n, int isign, double *tmp)
{
    if (n < 4)
        return;
    const int nh = n >> 1;
    if (isign >= 0) {
        for (int i = 0; i < nh; i++) {
            const double a0 = a[2*i], a1 = a[2*i + 1];
            const double a2 = a[(2*i + 2) % n], a3 = a[(2*i + 3) % n];
            tmp[i]      = D4_C0*a0 + D4_C1*a1 + D4_C2*a2 + D4_C3*a3;
            tmp[i + nh] = D4_C3*a0 - D4_C2*a1 + D4_C1*a2 - D4_C0*a3;
        }
    } else {
        /* output pair (2i, 2i+1) combines smooth/detail i-1 and i (wrapping) */
        for (int i = 0; i < nh; i++) {
            const int im = (i + nh - 1) % nh;
            const double sp = a[im], dp = a[im + nh];
            const double sc = a[i],  dc = a[i + nh];
            tmp[2*i]     = D4_C2*sp + D4_C1*dp + D4_C0*sc + D4_C3*dc;
            tmp[2*i + 1] = D4_C3*sp - D4_C0*dp + D4_C1*sc - D4_C2*dc;
        }
    }
    for (int i = 0; i < n; i++)
        a[i] = tmp[i];
}

/* ---- pyramid driver: src_c/include/wavelet.h:109-128 ---------------------- */
void oracle_dwt(double *a, int n, int isign, double *tmp)
{
    if (n < 4)
        return;
    if (isign >= 0) {
        for (int nn = n; nn >= 4; nn >>= 1)
            d4_step(a, nn, isign, tmp);
    } else {
        for (int nn = 4; nn <= n; nn <<= 1)
            d4_step(a, nn, isign, tmp);
    }
}

static int ceil_log2(int n)
{
    int m = 0;
    while ((1 << m) < n)
        m++;
    return m;
}

/* ---- daub4(): src_c/_dwt.c:154-186 ------------------------------------------
 * out has 2^ceil(log2 n) entries (zero padded); isign>=0 forward, else inverse.
 * Returns the padded length. */
int oracle_daub4(const double *in, int n, int isign, double *out, double *tmp)
{
    const int m = ceil_log2(n);
    const int nw = 1 << m;
    for (int j = 0; j < n; j++)
        out[j] = in[j];
    for (int j = n; j < nw; j++)
        out[j] = 0.0;
    oracle_dwt(out, nw, isign, tmp);
    return nw;
}

/* ---- wavelet pseudo-chi-squared: src_c/_dwt.c:56-119 -----------------------
 * Carter & Winn (2009) eqs. 32-34.  params[npars-3..npars-1] = gamma, sigma_r,
 * sigma_w.  n must be a power of two: for other sizes the reference transforms
 * the unpadded length and reads an unwritten scratch slot, so
 * there is no defined answer to restate.  Residual sign is data - model
 * (_dwt.c:89).  wres/tmp: n doubles each.  Returns NaN for a bad size. */
double oracle_dwt_chisq(const double *params, int npars,
                        const double *model, const double *data, int n,
                        const double *off, const double *plow,
                        const double *pup, int np,
                        double *wres, double *tmp)
{
    const double g = 0.72134752;
    const double gamma = params[npars - 3];
    const double sigmar = params[npars - 2];
    const double sigmaw = params[npars - 1];
    const int M = ceil_log2(n);
    if ((1 << M) != n || n < 2)
        return NAN;

    for (int j = 0; j < n; j++)
        wres[j] = data[j] - model[j];
    oracle_dwt(wres, n, 1, tmp);

    const double sS2 = sigmar*sigmar*pow(2.0, -gamma)*g + sigmaw*sigmaw;
    double total = wres[0]*wres[0]/sS2 + wres[1]*wres[1]/sS2
                 + 2.0*log(2*M_PI*sS2);
    for (int m = 1; m < M; m++) {
        const double sW2 = sigmar*sigmar*pow(2.0, -gamma*m) + sigmaw*sigmaw;
        const int nm = 1 << m;
        double ss = 0.0;
        for (int j = 0; j < nm; j++)
            ss += wres[nm + j]*wres[nm + j];
        total += ss/sW2 + nm*log(2*M_PI*sW2);
    }
    if (off != NULL && np > 0)
        total += oracle_priors(off, plow, pup, np);
    return total;
}
