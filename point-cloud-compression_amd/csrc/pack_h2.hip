// pack_h2.hip -- host-side packing of the AE weights for the f16x2 mode (mfma_chain.h: two fp16 pieces per operand, three
// products per fp32 product).  Pure host code, runs once at model load; all pointers are HOST pointers.
//
// fp16 has five exponent bits, so every MFMA operand is scaled by an exact power of two into the format's range:
//   * activations entering layer l carry sigma_l, chosen so that a RIGOROUS upper bound of the layer's input maps to <= 2^15
//     (half of fp16's largest number).  The bounds are interval bounds of the reference's layers (pn_kit.py:98-144,146-211;
//     AE.py:16-27): post-ReLU inputs lie in [0, B_j], so output i is at most sum_j max(W_ij, 0) B_j + max(b_i, 0); signed
//     inputs (coordinates, latents) count with |W_ij|.  They start from inputs of magnitude <= 1: the kernels normalise every
//     patch by a power of two s <= 1 of their own (encoder: the patch's largest |coordinate|; decoder: the patch's largest head
//     activation or latent) and scale the biases with it -- Conv/ReLU stacks are positively homogeneous in (input, biases) --
//     so no input can overflow, and none is amplified (s <= 1) beyond what the bounds assume;
//   * weights of layer l carry tau_l = the largest power of two with max|W| tau <= 2^14, so that the lo piece of all but
//     negligible weights is a normal fp16 number;
//   * a layer's accumulator is sigma_l tau_l (W y + b s); its bias is stored here as b sigma_l tau_l (the kernel multiplies by s),
//     and the split in front of the next layer multiplies by rho = sigma_{l+1} / (sigma_l tau_l).
// Every scale is a power of two, so scaling commutes with fp32 rounding: the scaled chain computes exactly what the unscaled one
// would, and the only difference from the fp32 product is the 22-23-bit operand representation.
#include <math.h>
#include <string.h>

#include <vector>

#include "blobs.h"
#include "common.h"

extern "C" size_t pccx_ae_encoder_blob_floats(void);
extern "C" size_t pccx_ae_decoder_blob_floats(int k);
extern "C" int pccx_pack_ae_encoder(const float *, const float *, const float *, const float *, const float *, const float *, const float *,
                                    const float *, const float *, const float *, const float *, const float *, const float *, const float *,
                                    int, float *);
extern "C" int pccx_pack_ae_decoder(const float *, const float *, const float *, const float *, const float *, const float *, const float *,
                                    const float *, const float *, const float *, const float *, const float *, const float *, const float *,
                                    int, int, float *);

namespace {

// IEEE binary16 conversions (round to nearest even; written out so that the host build needs no fp16 runtime support)
uint16_t f2h(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7F800000u ? 0x7E00u : 0x7C00u));       // >= 65536, inf, nan
    if (x < 0x38800000u) {                                                                          // below 2^-14: subnormal half
        float v;
        memcpy(&v, &x, 4);
        return (uint16_t)(sign | (uint32_t)nearbyintf(v * 0x1p24f));                               // 0x400 = the smallest normal
    }
    uint32_t h = ((((x >> 23) - 112u) << 10) | ((x & 0x7FFFFFu) >> 13));
    const uint32_t rem = x & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;                                        // a carry into the exponent is right
    return (uint16_t)(sign | h);
}

float h2f(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    float v;
    if (e == 0) {
        v = ldexpf((float)m, -24);
        return sign ? -v : v;
    }
    const uint32_t bits = e == 31 ? (sign | 0x7F800000u | (m << 13)) : (sign | ((e + 112u) << 23) | (m << 13));
    memcpy(&v, &bits, 4);
    return v;
}

double pow2_floor(double x)
{
    int e;
    frexp(x, &e);                                     // x = m 2^e, m in [0.5, 1)
    return ldexp(1.0, e - 1);
}

double act_scale(double bound)                       // bound * scale <= 2^15
{
    if (!(bound > 0x1p-40)) bound = 0x1p-40;
    return pow2_floor(32768.0 / bound);
}

double w_scale(const float *W, size_t n)            // max|W| * scale <= 2^14
{
    double m = 0;
    for (size_t i = 0; i < n; ++i) m = fmax(m, fabs((double)W[i]));
    return m > 0 ? pow2_floor(16384.0 / m) : 1.0;
}

double vmax(const std::vector<double> &v)
{
    double m = 0;
    for (double x : v) m = fmax(m, x);
    return m;
}

// upper bounds of relu(W x + b s), s in (0, 1]: x_j in [0, bx_j] for the non-negative inputs, |x_j| <= bx_j where sgn[j]
std::vector<double> relu_bounds(const float *W, const float *b, int out, int in, const std::vector<double> &bx, const std::vector<char> &sgn)
{
    std::vector<double> r((size_t)out);
    for (int i = 0; i < out; ++i) {
        double a = b ? fmax((double)b[i], 0.0) : 0.0;
        for (int j = 0; j < in; ++j) {
            const double w = W[(size_t)i * in + j];
            a += (sgn[j] ? fabs(w) : fmax(w, 0.0)) * bx[j];
        }
        r[i] = a;
    }
    return r;
}

// fp32 fragments [KT16][src_w] (kt-major, 256 floats each; `src` may point at a column of a wider layer) -> f16x2 planes
// [T][W][2]: k-tiles 2t and 2t+1 concatenate into one K = 32 operand (lane (m, kg) holds k-slots 8 kg + j), an odd last k-tile
// pairs with zeros.  The same lane-local rearrangement as decoder.hip's b3_split_kernel.
void h2_planes(const float *src, int KT16, int W, int src_w, double tau, float *dst)
{
    const int T = (KT16 + 1) / 2;
    const float tf = (float)tau;
    for (int t = 0; t < T; ++t)
        for (int j = 0; j < W; ++j)
            for (int lane = 0; lane < 64; ++lane) {
                uint16_t h[8], l[8];
                for (int i = 0; i < 8; ++i) {
                    const int kt = 2 * t + (i >> 2);
                    const float x = kt < KT16 ? src[((size_t)kt * src_w + j) * 256 + lane * 4 + (i & 3)] * tf : 0.f;
                    h[i] = f2h(x);
                    l[i] = f2h(x - h2f(h[i]));
                }
                uint32_t *d0 = (uint32_t *)(dst + ((size_t)(t * W + j) * 2) * 256) + lane * 4, *d1 = d0 + 256;
                for (int q = 0; q < 4; ++q) {
                    d0[q] = (uint32_t)h[2 * q] | ((uint32_t)h[2 * q + 1] << 16);
                    d1[q] = (uint32_t)l[2 * q] | ((uint32_t)l[2 * q + 1] << 16);
                }
            }
}

bool all_finite(const float *p, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (!isfinite(p[i])) return false;
    return true;
}

void scaled_copy(float *dst, const float *src, int n, int padded, double s)
{
    for (int i = 0; i < padded; ++i) dst[i] = i < n ? (float)((double)src[i] * s) : 0.f;
}

}  // namespace

extern "C" size_t pccx_ae_encoder_h2_blob_floats(void) { return ENC_H2_BLOB_FLOATS; }
extern "C" size_t pccx_ae_decoder_h2_blob_floats(int k) { return k >= 1 ? (size_t)DEC_H2_BLOB_FLOATS(k) : 0; }

extern "C" int pccx_pack_ae_encoder_h2(const float *sa_w0, const float *sa_b0, const float *sa_w1, const float *sa_b1, const float *sa_w2,
                                       const float *sa_b2, const float *pn_w0, const float *pn_b0, const float *pn_w1, const float *pn_b1,
                                       const float *pn_w2, const float *pn_b2, const float *pn_w3, const float *pn_b3, int d, float *blob)
{
    std::vector<float> enc(pccx_ae_encoder_blob_floats());
    const int rc = pccx_pack_ae_encoder(sa_w0, sa_b0, sa_w1, sa_b1, sa_w2, sa_b2, pn_w0, pn_b0, pn_w1, pn_b1, pn_w2, pn_b2, pn_w3, pn_b3, d, enc.data());
    if (rc != PCCX_OK) return rc;
    PCCX_CHECK_ARG(blob, "pccx_pack_ae_encoder_h2: null pointer");
    PCCX_CHECK_ARG(all_finite(sa_w0, 96) && all_finite(sa_b0, 32) && all_finite(sa_w1, 64 * 32) && all_finite(sa_b1, 64) &&
                       all_finite(sa_w2, 128 * 64) && all_finite(sa_b2, 128) && all_finite(pn_w0, 128 * 131) && all_finite(pn_b0, 128) &&
                       all_finite(pn_w1, 256 * 128) && all_finite(pn_b1, 256) && all_finite(pn_w2, 512 * 256) && all_finite(pn_b2, 512) &&
                       all_finite(pn_w3, (size_t)d * 512) && all_finite(pn_b3, d),
                   "pccx_pack_ae_encoder_h2: non-finite weight or bias (the f16x2 scales need finite layer bounds)");
    memset(blob, 0, sizeof(float) * ENC_H2_BLOB_FLOATS);
    float *meta = blob + ENC_H2_META;

    // ---- SetAbstraction (pn_kit.py:146-211).  conv0 stays an fp32 MFMA; its input is the difference of two normalised coordinates
    std::vector<double> B0(32);
    for (int c = 0; c < 32; ++c)
        B0[c] = 2.0 * (fabs((double)sa_w0[3 * c]) + fabs((double)sa_w0[3 * c + 1]) + fabs((double)sa_w0[3 * c + 2])) + fmax((double)sa_b0[c], 0.0);
    const double sig0 = act_scale(vmax(B0)), tau1 = w_scale(sa_w1, 64 * 32);
    const std::vector<double> B1 = relu_bounds(sa_w1, sa_b1, 64, 32, B0, std::vector<char>(32, 0));
    const double sig1 = act_scale(vmax(B1)), tau2 = w_scale(sa_w2, 128 * 64);
    const std::vector<double> B2 = relu_bounds(sa_w2, sa_b2, 128, 64, B1, std::vector<char>(64, 0));
    meta[H2E_RHO0] = (float)sig0;
    meta[H2E_RHO1] = (float)(sig1 / (sig0 * tau1));
    meta[H2E_INV2] = (float)(1.0 / (sig1 * tau2));
    scaled_copy(blob + ENC_H2_SA_B1, sa_b1, 64, 64, sig0 * tau1);
    scaled_copy(blob + ENC_H2_SA_B2, sa_b2, 128, 128, 1.0);
    h2_planes(enc.data() + ENC_SA_W1, 2, 4, 4, tau1, blob + ENC_H2_SA_W);
    h2_planes(enc.data() + ENC_SA_W2, 4, 8, 8, tau2, blob + ENC_H2_SA_W + (size_t)ENC_H2_SA_W1_FRAGS * 256);

    // ---- PointNet (pn_kit.py:98-144) on cat(xyz, feature) (AE.py:39): columns 0..2 signed coordinates (|x| <= 1), 3..130 features
    std::vector<double> bin(131);
    std::vector<char> sgn(131, 0);
    for (int j = 0; j < 3; ++j) { bin[j] = 1.0; sgn[j] = 1; }
    for (int j = 0; j < 128; ++j) bin[3 + j] = B2[j];
    const double sig_in = act_scale(fmax(vmax(B2), 1.0)), tp0 = w_scale(pn_w0, 128 * 131);
    const std::vector<double> P0 = relu_bounds(pn_w0, pn_b0, 128, 131, bin, sgn);
    const double sp0 = act_scale(vmax(P0)), tp1 = w_scale(pn_w1, 256 * 128);
    const std::vector<double> P1 = relu_bounds(pn_w1, pn_b1, 256, 128, P0, std::vector<char>(128, 0));
    const double sp1 = act_scale(vmax(P1)), tp2 = w_scale(pn_w2, 512 * 256);
    const std::vector<double> P2 = relu_bounds(pn_w2, pn_b2, 512, 256, P1, std::vector<char>(256, 0));
    const double sp2 = act_scale(vmax(P2)), tp3 = w_scale(pn_w3, (size_t)d * 512);
    meta[H2E_RHO_IN] = (float)sig_in;
    meta[H2E_RHO_P1] = (float)(sp0 / (sig_in * tp0));
    meta[H2E_RHO_P2] = (float)(sp1 / (sp0 * tp1));
    meta[H2E_RHO_P3] = (float)(sp2 / (sp1 * tp2));
    meta[H2E_INV_OUT] = (float)(1.0 / (sp2 * tp3));
    scaled_copy(blob + ENC_H2_PN_B0, pn_b0, 128, 128, sig_in * tp0);
    scaled_copy(blob + ENC_H2_PN_B1, pn_b1, 256, 256, sp0 * tp1);
    scaled_copy(blob + ENC_H2_PN_B2, pn_b2, 512, 512, sp1 * tp2);
    scaled_copy(blob + ENC_H2_PN_B3, pn_b3, d, 16, sp2 * tp3);
    // stream order: L0 [5][8], L1 [4][16], then for each half h of layer 2's output tiles: L2 [8][16 tiles 16h..] and the eight
    // k-steps 8h.. of layer 3 that consume them (as pccx_pack_pn_b3)
    struct Seg { size_t src; int kt16, w, src_w; double tau; };
    const Seg segs[6] = {{ENC_PN_W0, 9, 8, 8, tp0},
                         {ENC_PN_W1, 8, 16, 16, tp1},
                         {ENC_PN_W2, 16, 16, 32, tp2},
                         {ENC_PN_W3, 16, 1, 1, tp3},
                         {ENC_PN_W2 + (size_t)16 * 256, 16, 16, 32, tp2},
                         {ENC_PN_W3 + (size_t)16 * 256, 16, 1, 1, tp3}};
    size_t f = 0;
    for (int l = 0; l < 6; ++l) {
        h2_planes(enc.data() + segs[l].src, segs[l].kt16, segs[l].w, segs[l].src_w, segs[l].tau, blob + ENC_H2_PN_STREAM + f * 256);
        f += (size_t)((segs[l].kt16 + 1) / 2) * segs[l].w * 2;
    }
    if (f != PN_H2_STREAM_FRAGS) { pccx_set_error("pccx_pack_ae_encoder_h2: stream has %zu fragments", f); return PCCX_ERR_ARG; }
    {   // the order of the two-tiles-per-wave form (blobs.h: ENC_H2_PN_STREAM2)
        float *st = blob + ENC_H2_PN_STREAM2;
        h2_planes(enc.data() + ENC_PN_W0, 9, 8, 8, tp0, st);
        size_t g = PN_H2_S2_L0_FRAGS;
        h2_planes(enc.data() + ENC_PN_W1, 8, 16, 16, tp1, st + g * 256);
        g += 128;
        for (int e = 0; e < 8; ++e) {
            h2_planes(enc.data() + ENC_PN_W2 + (size_t)4 * e * 256, 16, 4, 32, tp2, st + g * 256);
            g += 64;
            h2_planes(enc.data() + ENC_PN_W3 + (size_t)4 * e * 256, 4, 1, 1, tp3, st + g * 256);
            g += 4;
        }
        if (g != PN_H2_S2_L0_FRAGS + PN_H2_S2_MAIN_FRAGS) { pccx_set_error("pccx_pack_ae_encoder_h2: second stream has %zu fragments", g); return PCCX_ERR_ARG; }
    }
    for (int i = 0; i < 8; ++i)
        if (!(meta[i] > 0.f) || !isfinite(meta[i])) { pccx_set_error("pccx_pack_ae_encoder_h2: scale %d is not a positive finite number (non-finite weights?)", i); return PCCX_ERR_ARG; }
    return PCCX_OK;
}

extern "C" int pccx_pack_ae_decoder_h2(const float *ip_w0, const float *ip_b0, const float *ip_w1, const float *ip_b1, const float *ip_w2,
                                       const float *ip_b2, const float *m_w0, const float *m_b0, const float *m_w1, const float *m_b1,
                                       const float *m_w2, const float *m_b2, const float *m_w3, const float *m_b3, int k, int d, float *blob)
{
    std::vector<float> dec(pccx_ae_decoder_blob_floats(k));
    const int rc = pccx_pack_ae_decoder(ip_w0, ip_b0, ip_w1, ip_b1, ip_w2, ip_b2, m_w0, m_b0, m_w1, m_b1, m_w2, m_b2, m_w3, m_b3, k, d, dec.data());
    if (rc != PCCX_OK) return rc;
    PCCX_CHECK_ARG(blob, "pccx_pack_ae_decoder_h2: null pointer");
    PCCX_CHECK_ARG(all_finite(ip_w2, (size_t)k * 128 * 1024) && all_finite(ip_b2, (size_t)k * 128) && all_finite(m_w0, (size_t)128 * (128 + d)) &&
                       all_finite(m_b0, 128) && all_finite(m_w1, 64 * 128) && all_finite(m_b1, 64) && all_finite(m_w2, 32 * 64) &&
                       all_finite(m_b2, 32) && all_finite(m_w3, 3 * 32) && all_finite(m_b3, 3),
                   "pccx_pack_ae_decoder_h2: non-finite weight or bias (the f16x2 scales need finite layer bounds)");
    memset(blob, 0, sizeof(float) * (size_t)DEC_H2_BLOB_FLOATS(k));
    float *meta = blob + DEC_H2_META;
    // the head activation (relu(inv_pool.2), 1024 channels) and the latent arrive normalised per patch to at most 1
    const double sig_h = 32768.0, tau_g = w_scale(ip_w2, (size_t)k * 128 * 1024);
    // inv_pool.4 (AE.py:24-26): row o = c*k + p; channel c of inv_mlp's input is bounded by the largest of its k rows
    std::vector<double> Bc(128, 0.0);
    for (int o = 0; o < k * 128; ++o) {
        double a = fmax((double)ip_b2[o], 0.0);
        for (int j = 0; j < 1024; ++j) a += fmax((double)ip_w2[(size_t)o * 1024 + j], 0.0);
        Bc[o / k] = fmax(Bc[o / k], a);
    }
    const int in0 = 128 + d;
    std::vector<double> bin((size_t)in0);
    std::vector<char> sgn((size_t)in0, 0);
    for (int j = 0; j < 128; ++j) bin[j] = Bc[j];
    for (int j = 128; j < in0; ++j) { bin[j] = 1.0; sgn[j] = 1; }
    const double sig0 = act_scale(fmax(vmax(Bc), 1.0)), tau0 = w_scale(m_w0, (size_t)128 * in0);
    const std::vector<double> M0 = relu_bounds(m_w0, m_b0, 128, in0, bin, sgn);
    const double sig1 = act_scale(vmax(M0)), tau1 = w_scale(m_w1, 64 * 128);
    const std::vector<double> M1 = relu_bounds(m_w1, m_b1, 64, 128, M0, std::vector<char>(128, 0));
    const double sig2 = act_scale(vmax(M1)), tau2 = w_scale(m_w2, 32 * 64);
    const std::vector<double> M2 = relu_bounds(m_w2, m_b2, 32, 64, M1, std::vector<char>(64, 0));
    const double sig3 = act_scale(vmax(M2)), tau3 = w_scale(m_w3, 3 * 32);
    meta[H2D_SIG_H] = (float)sig_h;
    meta[H2D_RHO0] = (float)(sig0 / (sig_h * tau_g));
    meta[H2D_SIG_Q] = (float)sig0;
    meta[H2D_RHO1] = (float)(sig1 / (sig0 * tau0));
    meta[H2D_RHO2] = (float)(sig2 / (sig1 * tau1));
    meta[H2D_RHO3] = (float)(sig3 / (sig2 * tau2));
    meta[H2D_INV_OUT] = (float)(1.0 / (sig3 * tau3));
    scaled_copy(blob + DEC_H2_M_B0, m_b0, 128, 128, sig0 * tau0);
    scaled_copy(blob + DEC_H2_M_B1, m_b1, 64, 64, sig1 * tau1);
    scaled_copy(blob + DEC_H2_M_B2, m_b2, 32, 32, sig2 * tau2);
    scaled_copy(blob + DEC_H2_M_B3, m_b3, 3, 16, sig3 * tau3);
    for (int p = 0; p < k; ++p)
        for (int c = 0; c < 128; ++c) blob[DEC_H2_G_B + (size_t)p * 128 + c] = (float)((double)ip_b2[(size_t)c * k + p] * sig_h * tau_g);
    // per-point streams
    const size_t src_stride = (size_t)DEC_STREAM_CHUNKS * DEC_WS_CHUNK * 256, dst_stride = (size_t)DEC_H2_STREAM_CHUNKS * DEC_H2_CHUNK * 256;
    std::vector<float> tail((size_t)DEC_H2_TAIL_FRAGS * 256);
    {
        const int KT16[4] = {9, 8, 4, 2}, MTL[4] = {8, 4, 2, 1};
        const size_t src[4] = {DEC_M_W0, DEC_M_W1, DEC_M_W2, DEC_M_W3};
        const double tau[4] = {tau0, tau1, tau2, tau3};
        size_t f = 0;
        for (int l = 0; l < 4; ++l) {
            h2_planes(dec.data() + src[l], KT16[l], MTL[l], MTL[l], tau[l], tail.data() + f * 256);
            f += (size_t)((KT16[l] + 1) / 2) * MTL[l] * 2;
        }
        if (f != DEC_H2_TAIL_FRAGS) { pccx_set_error("pccx_pack_ae_decoder_h2: tail has %zu fragments", f); return PCCX_ERR_ARG; }
    }
    for (int p = 0; p < k; ++p) {
        float *st = blob + DEC_H2_G_W(k) + (size_t)p * dst_stride;
        h2_planes(dec.data() + DEC_G_W(k) + (size_t)p * src_stride, 64, 8, 8, tau_g, st);
        memcpy(st + (size_t)DEC_H2_GEMM_FRAGS * 256, tail.data(), tail.size() * sizeof(float));
        memcpy(st + (size_t)(DEC_H2_GEMM_FRAGS + DEC_H2_TAIL_FRAGS) * 256, tail.data(), tail.size() * sizeof(float));       // second copy (blobs.h)
    }
    for (int i = 0; i < 7; ++i)
        if (!(meta[i] > 0.f) || !isfinite(meta[i])) { pccx_set_error("pccx_pack_ae_decoder_h2: scale %d is not a positive finite number (non-finite weights?)", i); return PCCX_ERR_ARG; }
    return PCCX_OK;
}
