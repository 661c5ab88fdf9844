// pack.hip -- host-side packing of the reference's state_dict tensors into the blobs of blobs.h.
// Pure host code (no kernels): runs once at model load.  All pointers here are HOST pointers.
#include <string.h>

#include "blobs.h"
#include "common.h"

namespace {

// frag(kt,mt)[lane][r] = W[row_of(16mt + (lane&15))][col_of(16kt + 4(lane>>4) + r)], zero where either is < 0.
struct KtMajor {
    int MT;
    size_t operator()(int kt, int mt) const { return (size_t)kt * MT + mt; }
};

template <class RowOf, class ColOf, class FragAt>
void pack_dense(const float *W, int ldw, int KT, int MT, float *dst, RowOf row_of, ColOf col_of, FragAt frag_at)
{
    for (int kt = 0; kt < KT; ++kt)
        for (int mt = 0; mt < MT; ++mt)
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const long orow = row_of(16 * mt + (lane & 15));
                    const long ocol = col_of(16 * kt + 4 * (lane >> 4) + r);
                    dst[(frag_at(kt, mt) * 64 + lane) * 4 + r] = (orow >= 0 && ocol >= 0) ? W[(size_t)orow * ldw + ocol] : 0.f;
                }
}

template <class RowOf, class ColOf>
void pack_dense(const float *W, int ldw, int KT, int MT, float *dst, RowOf row_of, ColOf col_of)
{
    pack_dense(W, ldw, KT, MT, dst, row_of, col_of, KtMajor{MT});
}

struct Ident {
    int n;
    long operator()(int i) const { return i < n ? i : -1; }
};

void copy_pad(float *dst, const float *src, int n, int padded)
{
    memset(dst, 0, sizeof(float) * (size_t)padded);
    if (src) memcpy(dst, src, sizeof(float) * (size_t)n);
}

}  // namespace

extern "C" size_t pccx_ae_encoder_blob_floats(void) { return ENC_BLOB_FLOATS; }
extern "C" size_t pccx_ae_decoder_blob_floats(int k) { return k >= 1 ? (size_t)DEC_BLOB_FLOATS(k) : 0; }
extern "C" size_t pccx_prob_blob_floats(void) { return PRB_BLOB_FLOATS; }

extern "C" int pccx_pack_ae_encoder(const float *sa_w0, const float *sa_b0, const float *sa_w1, const float *sa_b1,
                                    const float *sa_w2, const float *sa_b2, const float *pn_w0, const float *pn_b0,
                                    const float *pn_w1, const float *pn_b1, const float *pn_w2, const float *pn_b2,
                                    const float *pn_w3, const float *pn_b3, int d, float *blob)
{
    PCCX_CHECK_ARG(sa_w0 && sa_b0 && sa_w1 && sa_b1 && sa_w2 && sa_b2 && pn_w0 && pn_b0 && pn_w1 && pn_b1 && pn_w2 &&
                       pn_b2 && pn_w3 && pn_b3 && blob,
                   "pccx_pack_ae_encoder: null pointer");
    PCCX_CHECK_ARG(d >= 1 && d <= 16, "pccx_pack_ae_encoder: bottleneck d=%d unsupported (1..16)", d);
    memset(blob, 0, sizeof(float) * ENC_BLOB_FLOATS);
    for (int c = 0; c < 32; ++c) {
        blob[ENC_SA_W0B0 + 4 * c + 0] = sa_w0[3 * c + 0];
        blob[ENC_SA_W0B0 + 4 * c + 1] = sa_w0[3 * c + 1];
        blob[ENC_SA_W0B0 + 4 * c + 2] = sa_w0[3 * c + 2];
        blob[ENC_SA_W0B0 + 4 * c + 3] = sa_b0[c];
    }
    copy_pad(blob + ENC_SA_B1, sa_b1, 64, 64);
    copy_pad(blob + ENC_SA_B2, sa_b2, 128, 128);
    pack_dense(sa_w1, 32, 2, 4, blob + ENC_SA_W1, Ident{64}, Ident{32});
    pack_dense(sa_w2, 64, 4, 8, blob + ENC_SA_W2, Ident{128}, Ident{64});
    copy_pad(blob + ENC_PN_B0, pn_b0, 128, 128);
    copy_pad(blob + ENC_PN_B1, pn_b1, 256, 256);
    copy_pad(blob + ENC_PN_B2, pn_b2, 512, 512);
    copy_pad(blob + ENC_PN_B3, pn_b3, d, 16);
    // PointNet input is cat((xyz, sa_feature), dim=1) (AE.py:39): original column 0..2 = xyz,
    // 3..130 = feature.  Kernel channel order: 128 features first, then xyz.
    pack_dense(pn_w0, 131, 9, 8, blob + ENC_PN_W0, Ident{128},
               [](int ch) -> long { return ch < 128 ? 3 + ch : (ch < 131 ? ch - 128 : -1); });
    pack_dense(pn_w1, 128, 8, 16, blob + ENC_PN_W1, Ident{256}, Ident{128});
    pack_dense(pn_w2, 256, 16, 32, blob + ENC_PN_W2, Ident{512}, Ident{256});
    pack_dense(pn_w3, 512, 32, 1, blob + ENC_PN_W3, Ident{d}, Ident{512});
    // consumption-order stream for the LDS-staged PointNet kernel
    {
        float *st = blob + ENC_PN_STREAM;
        int f = 0;
        auto put = [&](int base, int MT, int kt, int mt) {
            memcpy(st + (size_t)f * 256, blob + base + ((size_t)kt * MT + mt) * 256, 256 * sizeof(float));
            ++f;
        };
        for (int kt = 0; kt < 9; ++kt) for (int mt = 0; mt < 8; ++mt) put(ENC_PN_W0, 8, kt, mt);
        for (int kt = 0; kt < 8; ++kt) for (int mt = 0; mt < 16; ++mt) put(ENC_PN_W1, 16, kt, mt);
        for (int mp = 0; mp < 16; ++mp) {
            for (int kt = 0; kt < 16; ++kt) for (int m = 0; m < 2; ++m) put(ENC_PN_W2, 32, kt, 2 * mp + m);
            for (int k3 = 0; k3 < 2; ++k3) put(ENC_PN_W3, 1, 2 * mp + k3, 0);
        }
        if (f != ENC_PN_STREAM_FRAGS) { pccx_set_error("pccx_pack_ae_encoder: stream has %d fragments", f); return PCCX_ERR_ARG; }
    }
    return PCCX_OK;
}

extern "C" int pccx_pack_ae_decoder(const float *ip_w0, const float *ip_b0, const float *ip_w1, const float *ip_b1,
                                    const float *ip_w2, const float *ip_b2, const float *m_w0, const float *m_b0,
                                    const float *m_w1, const float *m_b1, const float *m_w2, const float *m_b2,
                                    const float *m_w3, const float *m_b3, int k, int d, float *blob)
{
    PCCX_CHECK_ARG(ip_w0 && ip_b0 && ip_w1 && ip_b1 && ip_w2 && ip_b2 && m_w0 && m_b0 && m_w1 && m_b1 && m_w2 && m_b2 &&
                       m_w3 && m_b3 && blob,
                   "pccx_pack_ae_decoder: null pointer");
    PCCX_CHECK_ARG(d >= 1 && d <= 16 && k >= 1, "pccx_pack_ae_decoder: unsupported d=%d k=%d", d, k);
    memset(blob, 0, sizeof(float) * (size_t)DEC_BLOB_FLOATS(k));
    copy_pad(blob + DEC_H_B1, ip_b0, 256, 256);
    copy_pad(blob + DEC_H_B2, ip_b1, 1024, 1024);
    pack_dense(ip_w0, d, 1, 16, blob + DEC_H_W1, Ident{256}, Ident{d});
    pack_dense(ip_w1, 256, 16, 64, blob + DEC_H_W2, Ident{1024}, Ident{256});
    copy_pad(blob + DEC_M_B0, m_b0, 128, 128);
    copy_pad(blob + DEC_M_B1, m_b1, 64, 64);
    copy_pad(blob + DEC_M_B2, m_b2, 32, 32);
    copy_pad(blob + DEC_M_B3, m_b3, 3, 16);
    // inv_mlp input is cat((linear_output, latent), dim=1) (AE.py:51): columns 0..127 then 128..128+d-1
    pack_dense(m_w0, 128 + d, 9, 8, blob + DEC_M_W0, Ident{128}, Ident{128 + d});
    pack_dense(m_w1, 128, 8, 4, blob + DEC_M_W1, Ident{64}, Ident{128});
    pack_dense(m_w2, 64, 4, 2, blob + DEC_M_W2, Ident{32}, Ident{64});
    pack_dense(m_w3, 32, 2, 1, blob + DEC_M_W3, Ident{3}, Ident{32});
    // inv_pool.4: output o = c*k + p after .view(BS,-1,k) (AE.py:49).  Rows are permuted to
    // o' = p*128 + c so that an m-tile of the GEMM is 16 channels of ONE point p: its accumulators
    // are then directly the B operand of inv_mlp's first layer.
    for (int p = 0; p < k; ++p)
        for (int c = 0; c < 128; ++c) blob[DEC_G_B + p * 128 + c] = ip_b2[(size_t)c * k + p];
    // Per-point weight stream (blobs.h): [p][ 64 kt x 8 m-tiles | inv_mlp fragments | pad ].
    const size_t stride = (size_t)DEC_STREAM_CHUNKS * DEC_WS_CHUNK; // fragments per point
    pack_dense(ip_w2, 1024, 64, k * 8, blob + DEC_G_W(k),
               [k](int row) -> long { return (long)(row & 127) * k + (row >> 7); }, Ident{1024},
               [=](int kt, int mt) -> size_t { return (size_t)(mt >> 3) * stride + (size_t)kt * 8 + (mt & 7); });
    for (int p = 0; p < k; ++p) {
        float *st = blob + DEC_G_W(k) + ((size_t)p * stride + DEC_STREAM_GEMM_FRAGS) * 256;
        memcpy(st, blob + DEC_M_W0, sizeof(float) * 114 * 256);   // DEC_M_W0..W3 are contiguous, kt-major each
    }
    return PCCX_OK;
}

extern "C" int pccx_pack_prob(const float *p_w0, const float *p_b0, const float *p_w1, const float *p_b1, const float *p_w2,
                              const float *p_b2, const float *m_w0, const float *m_b0, const float *m_w1, const float *m_b1,
                              const float *m_w2, const float *m_b2, int d, int L, float *blob)
{
    PCCX_CHECK_ARG(p_w0 && p_b0 && p_w1 && p_b1 && p_w2 && p_b2 && m_w0 && m_b0 && m_w1 && m_b1 && m_w2 && m_b2 && blob,
                   "pccx_pack_prob: null pointer");
    PCCX_CHECK_ARG(d >= 1 && L >= 1 && d * L <= 128, "pccx_pack_prob: d*L=%d > 128 unsupported", d * L);
    memset(blob, 0, sizeof(float) * PRB_BLOB_FLOATS);
    copy_pad(blob + PRB_P_B0, p_b0, 64, 64);
    copy_pad(blob + PRB_P_B1, p_b1, 128, 128);
    copy_pad(blob + PRB_P_B2, p_b2, 256, 256);
    copy_pad(blob + PRB_M_B0, m_b0, 512, 512);
    copy_pad(blob + PRB_M_B1, m_b1, 512, 512);
    copy_pad(blob + PRB_M_B2, m_b2, d * L, 128);
    pack_dense(p_w0, 3, 1, 4, blob + PRB_P_W0, Ident{64}, Ident{3});
    pack_dense(p_w1, 64, 4, 8, blob + PRB_P_W1, Ident{128}, Ident{64});
    pack_dense(p_w2, 128, 8, 16, blob + PRB_P_W2, Ident{256}, Ident{128});
    // mlp input is cat((sampled_xyz, feature), dim=2) (AE.py:115): columns 0..2 xyz, 3..258 feature.
    pack_dense(m_w0, 259, 17, 32, blob + PRB_M_W0, Ident{512},
               [](int ch) -> long { return ch < 256 ? 3 + ch : (ch < 259 ? ch - 256 : -1); });
    pack_dense(m_w1, 512, 32, 32, blob + PRB_M_W1, Ident{512}, Ident{512});
    pack_dense(m_w2, 512, 32, 8, blob + PRB_M_W2, Ident{d * L}, Ident{512});
    {   // consumption-order stream of model_mlp for the LDS ring
        float *st = blob + PRB_STREAM;
        int f = 0;
        auto put = [&](size_t base, int MT, int kt, int mt) {
            memcpy(st + (size_t)f * 256, blob + base + ((size_t)kt * MT + mt) * 256, 256 * sizeof(float));
            ++f;
        };
        for (int mt = 0; mt < 32; ++mt) put(PRB_M_W0, 32, 16, mt);          // layer 0: only the xyz k-tile is per centre (prob.hip)
        for (int mp = 0; mp < 16; ++mp) {
            for (int kt = 0; kt < 32; ++kt) for (int m = 0; m < 2; ++m) put(PRB_M_W1, 32, kt, 2 * mp + m);
            for (int k2 = 0; k2 < 2; ++k2) for (int mt = 0; mt < 8; ++mt) put(PRB_M_W2, 8, 2 * mp + k2, mt);
        }
        if (f != PRB_STREAM_FRAGS) { pccx_set_error("pccx_pack_prob: stream has %d fragments", f); return PCCX_ERR_ARG; }
    }
    return PCCX_OK;
}
