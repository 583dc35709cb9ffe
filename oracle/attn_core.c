/* Plain-C (fp64) restatement of the cosine window-attention core -- TEST INFRASTRUCTURE ONLY.
 *
 * Independent of both the PyTorch oracle (swin_oracle.py) and the HIP kernels: scalar loops written from the spec
 * (SURVEY.md appendix C), following the reference line by line:
 *   swinv2_global.py:300-301  qkv feature index = s*C + head*d + j
 *   swinv2_global.py:304      q/max(|q|,1e-12) . k/max(|k|,1e-12)
 *   swinv2_global.py:305-306  * exp(min(tau_h, ln 100))
 *   swinv2_global.py:307      + bias[h][q][k]              (optional)
 *   swinv2_global.py:309-314  + mask[window % nW][q][k]    (optional; closed form: token >= thr in the last window row)
 *   swinv2_global.py:315-318  softmax over k, P.V, heads concatenated as head*d + j
 * Only tests/ may load the resulting oracle/_c/libattn_oracle.so (via ctypes).  Parity: pinned against the golden
 * vectors generated from the real reference (tests/test_oracle_golden.py::test_c_attention_core).
 */
#include <math.h>
#include <stdlib.h>

/* qkv [Bw][L][3C], logit_scale [h], bias [h][L][L] or NULL, out [Bw][L][C].
 * mask: windows per sample nW = nwh*nww; if thr > 0, windows with (w % nW) / nww == nwh-1 add -100 to pairs whose
 * tokens lie on different sides of `thr`. */
int swv2_oracle_attention_core(const double* qkv, const double* logit_scale, const double* bias, double* out, int Bw, int L,
                               int C, int h, int nwh, int nww, int thr) {
    const int d = C / h, nW = nwh * nww;
    double* s = (double*)malloc(sizeof(double) * L);
    if (!s) return -1;
    for (int w = 0; w < Bw; ++w) {
        const int masked = thr > 0 && ((w % nW) / nww) == nwh - 1;
        for (int hd = 0; hd < h; ++hd) {
            double tau = logit_scale[hd];
            if (tau > log(100.0)) tau = log(100.0);
            const double sigma = exp(tau);
            for (int q = 0; q < L; ++q) {
                const double* qv = qkv + ((size_t)w * L + q) * 3 * C + hd * d;
                double qn = 0;
                for (int j = 0; j < d; ++j) qn += qv[j] * qv[j];
                qn = sqrt(qn);
                if (qn < 1e-12) qn = 1e-12;
                double mx = -1e300;
                for (int k = 0; k < L; ++k) {
                    const double* kv = qkv + ((size_t)w * L + k) * 3 * C + C + hd * d;
                    double kn = 0, dot = 0;
                    for (int j = 0; j < d; ++j) { kn += kv[j] * kv[j]; dot += qv[j] * kv[j]; }
                    kn = sqrt(kn);
                    if (kn < 1e-12) kn = 1e-12;
                    double v = sigma * dot / (qn * kn);
                    if (bias) v += bias[((size_t)hd * L + q) * L + k];
                    if (masked && ((q >= thr) != (k >= thr))) v += -100.0;
                    s[k] = v;
                    if (v > mx) mx = v;
                }
                double sum = 0;
                for (int k = 0; k < L; ++k) { s[k] = exp(s[k] - mx); sum += s[k]; }
                for (int j = 0; j < d; ++j) {
                    double acc = 0;
                    for (int k = 0; k < L; ++k) acc += s[k] * qkv[((size_t)w * L + k) * 3 * C + 2 * C + hd * d + j];
                    out[((size_t)w * L + q) * C + hd * d + j] = acc / sum;
                }
            }
        }
    }
    free(s);
    return 0;
}
