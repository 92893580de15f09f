/*
 * oracle/detmath.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the three transcendental steps on the MCTS path, written only with
 * IEEE-754 +, -, *, /, fma and round-to-nearest-even so that the HIP kernels can reproduce
 * them bit for bit (the product's own copy lives in go-ai_amd/csrc/detmath.h; the two are
 * written independently and compared by tests/test_detmath.py and the GPU parity tests):
 *
 *   softmax32  -- prior = scipy.special.softmax(pi_logits) in fp32
 *                 (/root/reference/go_ai/search/mct.py:41)
 *   softmax64  -- prior[valid] = softmax(-child.val[valid]) in fp64
 *                 (/root/reference/go_ai/search/tree.py:152-157)
 *   tanh64     -- np.tanh(child.post_vals) in fp64
 *                 (/root/reference/go_ai/search/tree.py:180)
 *
 * The reference gets these from numpy/scipy SIMD routines whose last-bit behaviour is
 * CPU-dispatch dependent; they are NOT reproducible on a GPU.  These functions agree with
 * numpy to <= 2 ulp (fp32 exp), <= 3 ulp (fp64 tanh) -- pinned by tests/test_detmath.py --
 * and the visit-count parity tests show that this never flips a PUCT argmax on the
 * reference-generated golden searches.
 *
 * Summation order of the softmax denominators (shared with the wavefront kernels):
 * 64 strided partial sums  part[l] = e[l] + e[l+64] + e[l+128] + ...  (ascending), then a
 * halving tree  for off = 32,16,..,1: part[l] += part[l+off]  (l < off).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static float f32_from_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static double f64_from_bits(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

float goai_oracle_exp32(float x)
{
    if (!(x >= -104.0f)) return 0.0f;      /* also catches NaN -> 0, never fed */
    if (x > 88.0f) x = 88.0f;
    float n = rintf(x * 0x1.715476p+0f);
    float r = fmaf(n, -0x1.62e4p-1f, x);
    r = fmaf(n, -0x1.7f7d1cp-20f, r);
    float p = 0x1.a01a02p-13f;
    p = fmaf(p, r, 0x1.6c16c2p-10f);
    p = fmaf(p, r, 0x1.111112p-7f);
    p = fmaf(p, r, 0x1.555556p-5f);
    p = fmaf(p, r, 0x1.555556p-3f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int ni = (int)n;
    if (ni >= -126) return p * f32_from_bits((uint32_t)(ni + 127) << 23);
    return (p * f32_from_bits((uint32_t)(ni + 100 + 127) << 23)) * 0x1p-100f;
}

double goai_oracle_exp64(double x)
{
    if (!(x >= -746.0)) return 0.0;
    if (x > 709.0) x = 709.0;
    double n = rint(x * 0x1.71547652b82fep+0);
    double r = fma(n, -0x1.62e42fee00000p-1, x);
    r = fma(n, -0x1.a39ef35793c76p-33, r);
    double p = 0x1.6124613a86d09p-33;
    p = fma(p, r, 0x1.1eed8eff8d898p-29);
    p = fma(p, r, 0x1.ae64567f544e4p-26);
    p = fma(p, r, 0x1.27e4fb7789f5cp-22);
    p = fma(p, r, 0x1.71de3a556c734p-19);
    p = fma(p, r, 0x1.a01a01a01a01ap-16);
    p = fma(p, r, 0x1.a01a01a01a01ap-13);
    p = fma(p, r, 0x1.6c16c16c16c17p-10);
    p = fma(p, r, 0x1.1111111111111p-7);
    p = fma(p, r, 0x1.5555555555555p-5);
    p = fma(p, r, 0x1.5555555555555p-3);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    int ni = (int)n;
    if (ni >= -1022) return p * f64_from_bits((uint64_t)(ni + 1023) << 52);
    return (p * f64_from_bits((uint64_t)(ni + 1000 + 1023) << 52)) * 0x1p-1000;
}

/* expm1(y) = y * P(y) for 0 <= y < 1, Taylor to y^21. */
static double expm1_small(double y)
{
    static const double c[21] = {
        1.0, 0.5, 0x1.5555555555555p-3, 0x1.5555555555555p-5, 0x1.1111111111111p-7,
        0x1.6c16c16c16c17p-10, 0x1.a01a01a01a01ap-13, 0x1.a01a01a01a01ap-16,
        0x1.71de3a556c734p-19, 0x1.27e4fb7789f5cp-22, 0x1.ae64567f544e4p-26,
        0x1.1eed8eff8d898p-29, 0x1.6124613a86d09p-33, 0x1.93974a8c07c9dp-37,
        0x1.ae7f3e733b81fp-41, 0x1.ae7f3e733b81fp-45, 0x1.952c77030ad4ap-49,
        0x1.6827863b97d97p-53, 0x1.2f49b46814157p-57, 0x1.e542ba4020225p-62,
        0x1.71b8ef6dcf572p-66 };
    double p = c[20];
    for (int k = 19; k >= 0; k--) p = fma(p, y, c[k]);
    return y * p;
}

double goai_oracle_tanh64(double x)
{
    double a = fabs(x), r;
    if (a != a) return x;
    if (a < 0x1p-28) return x;
    if (a >= 20.0) r = 1.0;
    else if (a < 0.5) { double t = expm1_small(2.0 * a); r = t / (t + 2.0); }
    else { double e = goai_oracle_exp64(2.0 * a); r = 1.0 - 2.0 / (e + 1.0); }
    return x < 0 ? -r : r;
}

/* fp32 softmax over n logits (n <= 64*16). */
void goai_oracle_softmax32(const float *x, int n, float *out)
{
    float m = x[0], part[64];
    for (int i = 1; i < n; i++) if (x[i] > m) m = x[i];
    for (int l = 0; l < 64; l++) part[l] = 0.0f;
    for (int i = 0; i < n; i++) { out[i] = goai_oracle_exp32(x[i] - m); }
    for (int l = 0; l < 64; l++) { float s = 0.0f; for (int i = l; i < n; i += 64) s += out[i]; part[l] = s; }
    for (int off = 32; off >= 1; off >>= 1) for (int l = 0; l < off; l++) part[l] += part[l + off];
    for (int i = 0; i < n; i++) out[i] = out[i] / part[0];
}

/* fp64 masked softmax: out[i] = softmax(x restricted to valid)[i], 0 where !valid. */
void goai_oracle_softmax64_masked(const double *x, const uint8_t *valid, int n, double *out)
{
    double m = 0.0, part[64];
    int have = 0;
    for (int i = 0; i < n; i++) if (valid[i] && (!have || x[i] > m)) { m = x[i]; have = 1; }
    for (int i = 0; i < n; i++) out[i] = valid[i] ? goai_oracle_exp64(x[i] - m) : 0.0;
    for (int l = 0; l < 64; l++) { double s = 0.0; for (int i = l; i < n; i += 64) s += out[i]; part[l] = s; }
    for (int off = 32; off >= 1; off >>= 1) for (int l = 0; l < off; l++) part[l] += part[l + off];
    for (int i = 0; i < n; i++) out[i] = valid[i] ? out[i] / part[0] : 0.0;
}

/* numpy's pairwise float64 sum (numpy/_core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum)
 * as np.mean(list) uses it at /root/reference/go_ai/search/tree.py:180. */
double goai_oracle_pairwise_sum64(const double *a, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8], res;
        int i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return goai_oracle_pairwise_sum64(a, n2) + goai_oracle_pairwise_sum64(a + n2, n - n2);
    }
}
