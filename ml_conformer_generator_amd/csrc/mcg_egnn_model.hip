// EGNN weights: reference-layout tensors (SURVEY.md section 8b key list) repacked into MFMA fragment order and
// uploaded once per checkpoint; precision mode and measurement options of the model handle.
#include "mcg_egnn_internal.h"
#include "mcg_gemm.h"

#include <atomic>

namespace {

constexpr int H = MCG_H, HP = MCG_HP, NT = MCG_NT, IN_NF = MCG_IN_NF;
constexpr int GROUP_LDS_FLOATS = MCG_GROUP_LDS_FLOATS;

template <class F>
void pack_B(std::vector<float>& dst, int K, int n_tiles, F value /* (n, k) -> W */, int pad_floats = 0) {
    const int steps = K / 4;
    dst.assign((size_t)steps * n_tiles * 64 + pad_floats, 0.f);
    for (int st = 0; st < steps; ++st)
        for (int nt = 0; nt < n_tiles; ++nt)
            for (int l = 0; l < 64; ++l) {
                const int k = mcg_kperm(st, l >> 4, K);
                const int n = nt * 16 + (l & 15);
                dst[((size_t)st * n_tiles + nt) * 64 + l] = value(n, k);
            }
}

}  // namespace

int mcg_upload_f(const std::vector<float>& v, float** d) {
    MCG_HIP(hipMalloc((void**)d, v.size() * sizeof(float)));
    MCG_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return MCG_OK;
}
int mcg_upload_u16(const std::vector<uint16_t>& v, uint16_t** d) {
    MCG_HIP(hipMalloc((void**)d, v.size() * sizeof(uint16_t) + 64));
    MCG_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    return MCG_OK;
}
int mcg_upload_i(const std::vector<int>& v, int** d) {
    MCG_HIP(hipMalloc((void**)d, (v.size() ? v.size() : 1) * sizeof(int)));
    if (v.size()) MCG_HIP(hipMemcpy(*d, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
    return MCG_OK;
}

namespace {

int build_edge_layer(mcg_egnn* m, EdgeLayer& L, const float* w1 /*[420][842]*/, const float* b1, const float* w2,
                     const float* b2, const float* wv, float bv) {
    std::vector<float> buf;
    // (Wa | Wb): 54 column tiles over K = 420   (B-pack4: consumed by the row-block GEMM)
    buf.clear();
    mcg_pack_b4(buf, H, 2 * NT, [&](int n, int k) -> float {
        if (n < HP) return n < H ? w1[(size_t)n * (2 * H + 2) + k] : 0.f;
        const int nn = n - HP;
        return nn < H ? w1[(size_t)nn * (2 * H + 2) + H + k] : 0.f;
    });
    if (int e = mcg_upload_f(buf, &L.pab_Bp)) return e;
    m->allocs.push_back(L.pab_Bp);
    std::vector<float> v(2 * HP, 0.f);
    for (int n = 0; n < H; ++n) v[n] = b1[n];
    if (int e = mcg_upload_f(v, &L.pab_bias)) return e;
    m->allocs.push_back(L.pab_bias);
    v.assign(HP + 32, 0.f);     // (+32: the bf16 kernel reads k up to 447)
    for (int n = 0; n < H; ++n) v[n] = w1[(size_t)n * (2 * H + 2) + 2 * H];       // current d2 column (egnn.py:199)
    if (int e = mcg_upload_f(v, &L.wd)) return e;
    m->allocs.push_back(L.wd);
    for (int n = 0; n < H; ++n) v[n] = w1[(size_t)n * (2 * H + 2) + 2 * H + 1];   // initial d2 column
    if (int e = mcg_upload_f(v, &L.wd0)) return e;
    m->allocs.push_back(L.wd0);
    // + one LDS group of padding: the staged tail group over-reads up to 28 KiB (k_edge_lds)
    pack_B(buf, H, NT, [&](int n, int k) -> float { return n < H ? w2[(size_t)n * H + k] : 0.f; }, GROUP_LDS_FLOATS);
    if (int e = mcg_upload_f(buf, &L.w2_Bp)) return e;
    m->allocs.push_back(L.w2_Bp);
    buf.clear();            // the same weights as B-pack4 (quarter-tile body of the edge kernel)
    mcg_pack_b4(buf, H, NT, [&](int n, int k) -> float { return n < H ? w2[(size_t)n * H + k] : 0.f; });
    if (int e = mcg_upload_f(buf, &L.w2_Bp4)) return e;
    m->allocs.push_back(L.w2_Bp4);
    {   // bf16 operand packs of the same weights
        std::vector<uint16_t> b16;
        mcg_pack_b16(b16, H, 2 * NT, [&](int n, int k) -> float {
            if (n < HP) return n < H ? w1[(size_t)n * (2 * H + 2) + k] : 0.f;
            const int nn = n - HP;
            return nn < H ? w1[(size_t)nn * (2 * H + 2) + H + k] : 0.f;
        });
        if (int e = mcg_upload_u16(b16, &L.pab_Bp16)) return e;
        m->allocs.push_back(L.pab_Bp16);
        b16.clear();
        mcg_pack_b16x3(b16, H, 2 * NT, [&](int n, int k) -> float {
            if (n < HP) return n < H ? w1[(size_t)n * (2 * H + 2) + k] : 0.f;
            const int nn = n - HP;
            return nn < H ? w1[(size_t)nn * (2 * H + 2) + H + k] : 0.f;
        });
        if (int e = mcg_upload_u16(b16, &L.pab_Bp16x3)) return e;
        m->allocs.push_back(L.pab_Bp16x3);
        b16.clear();
        mcg_pack_b16(b16, H, NT, [&](int n, int k) -> float { return n < H ? w2[(size_t)n * H + k] : 0.f; });
        if (int e = mcg_upload_u16(b16, &L.w2_Bp16)) return e;
        m->allocs.push_back(L.w2_Bp16);
        // f32x6 mode: w = w1 + w2 + w3 with bf16 parts (each the RNE rounding of what the previous ones left),
        // packed part-major inside every 32-k block
        const int kb_n = mcg_kblocks16(H);
        std::vector<uint16_t> x3((size_t)kb_n * 3 * NT * 64 * 8, 0);
        auto bf_to_f = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; __builtin_memcpy(&f, &u, 4); return f; };
        for (int kb = 0; kb < kb_n; ++kb)
            for (int nt = 0; nt < NT; ++nt)
                for (int l = 0; l < 64; ++l)
                    for (int j = 0; j < 8; ++j) {
                        const int k = 32 * kb + 8 * (l >> 4) + j, n = nt * 16 + (l & 15);
                        float r = (k < H && n < H) ? w2[(size_t)n * H + k] : 0.f;
                        for (int part = 0; part < 3; ++part) {
                            const uint16_t hbits = mcg_f32_to_bf16_bits(r);
                            x3[((((size_t)kb * 3 + part) * NT + nt) * 64 + l) * 8 + j] = hbits;
                            r -= bf_to_f(hbits);
                        }
                    }
        if (int e = mcg_upload_u16(x3, &L.w2_Bp16x3)) return e;
        m->allocs.push_back(L.w2_Bp16x3);
    }
    v.assign(HP, 0.f);
    for (int n = 0; n < H; ++n) v[n] = b2[n];
    if (int e = mcg_upload_f(v, &L.b2)) return e;
    m->allocs.push_back(L.b2);
    v.assign(HP, 0.f);
    for (int n = 0; n < H; ++n) v[n] = wv[n];
    if (int e = mcg_upload_f(v, &L.wv)) return e;
    m->allocs.push_back(L.wv);
    L.bv = bv;
    return MCG_OK;
}

int build_node_layer(mcg_egnn* m, NodeLayer& L, const float* w3 /*[420][840]*/, const float* b3, const float* w4,
                     const float* b4) {
    std::vector<float> buf;
    // two K segments: [h | agg]  (egnn.py:66), each a B-pack4
    mcg_pack_b4(buf, H, NT, [&](int n, int k) -> float { return n < H ? w3[(size_t)n * (2 * H) + k] : 0.f; });
    mcg_pack_b4(buf, H, NT, [&](int n, int k) -> float { return n < H ? w3[(size_t)n * (2 * H) + H + k] : 0.f; });
    if (int e = mcg_upload_f(buf, &L.w3_Bp)) return e;
    m->allocs.push_back(L.w3_Bp);
    std::vector<float> v(HP, 0.f);
    for (int n = 0; n < H; ++n) v[n] = b3[n];
    if (int e = mcg_upload_f(v, &L.b3)) return e;
    m->allocs.push_back(L.b3);
    buf.clear();
    mcg_pack_b4(buf, H, NT, [&](int n, int k) -> float { return n < H ? w4[(size_t)n * H + k] : 0.f; });
    if (int e = mcg_upload_f(buf, &L.w4_Bp)) return e;
    m->allocs.push_back(L.w4_Bp);
    {
        std::vector<uint16_t> b16;
        mcg_pack_b16(b16, H, NT, [&](int n, int k) -> float { return n < H ? w3[(size_t)n * (2 * H) + k] : 0.f; });
        mcg_pack_b16(b16, H, NT, [&](int n, int k) -> float { return n < H ? w3[(size_t)n * (2 * H) + H + k] : 0.f; });
        if (int e = mcg_upload_u16(b16, &L.w3_Bp16)) return e;
        m->allocs.push_back(L.w3_Bp16);
        b16.clear();
        mcg_pack_b16(b16, H, NT, [&](int n, int k) -> float { return n < H ? w4[(size_t)n * H + k] : 0.f; });
        if (int e = mcg_upload_u16(b16, &L.w4_Bp16)) return e;
        m->allocs.push_back(L.w4_Bp16);
        b16.clear();
        mcg_pack_b16x3(b16, H, NT, [&](int n, int k) -> float { return n < H ? w3[(size_t)n * (2 * H) + k] : 0.f; });
        mcg_pack_b16x3(b16, H, NT, [&](int n, int k) -> float { return n < H ? w3[(size_t)n * (2 * H) + H + k] : 0.f; });
        if (int e = mcg_upload_u16(b16, &L.w3_Bp16x3)) return e;
        m->allocs.push_back(L.w3_Bp16x3);
        b16.clear();
        mcg_pack_b16x3(b16, H, NT, [&](int n, int k) -> float { return n < H ? w4[(size_t)n * H + k] : 0.f; });
        if (int e = mcg_upload_u16(b16, &L.w4_Bp16x3)) return e;
        m->allocs.push_back(L.w4_Bp16x3);
    }
    for (int n = 0; n < H; ++n) v[n] = b4[n];
    if (int e = mcg_upload_f(v, &L.b4)) return e;
    m->allocs.push_back(L.b4);
    return MCG_OK;
}

}  // namespace

static int egnn_build(mcg_egnn* m, const float* const* tensors, int n_blocks) {
    m->n_blocks = n_blocks;
    const float* const* t = tensors;
    // embedding (420x12) / bias, embedding_out (12x420) / bias
    std::vector<float> v((size_t)IN_NF * HP, 0.f);
    for (int n = 0; n < H; ++n)
        for (int k = 0; k < IN_NF; ++k) v[(size_t)k * HP + n] = t[0][(size_t)n * IN_NF + k];
    if (int e = mcg_upload_f(v, &m->emb_wT)) return e;
    m->allocs.push_back(m->emb_wT);
    v.assign(HP, 0.f);
    for (int n = 0; n < H; ++n) v[n] = t[1][n];
    if (int e = mcg_upload_f(v, &m->emb_b)) return e;
    m->allocs.push_back(m->emb_b);
    v.assign((size_t)IN_NF * HP, 0.f);
    for (int o = 0; o < IN_NF; ++o)
        for (int k = 0; k < H; ++k) v[(size_t)o * HP + k] = t[2][(size_t)o * H + k];
    if (int e = mcg_upload_f(v, &m->out_w)) return e;
    m->allocs.push_back(m->out_w);
    v.assign(16, 0.f);
    for (int o = 0; o < IN_NF; ++o) v[o] = t[3][o];
    if (int e = mcg_upload_f(v, &m->out_b)) return e;
    m->allocs.push_back(m->out_b);
    m->gcl_edge.resize(2 * n_blocks);
    m->gcl_node.resize(2 * n_blocks);
    m->equiv.resize(n_blocks);
    int idx = 4;
    for (int b = 0; b < n_blocks; ++b) {
        for (int gi = 0; gi < 2; ++gi) {
            const float* const* q = t + idx;   // edge0.w,b edge2.w,b node0.w,b node2.w,b att.w,b
            if (int e = build_edge_layer(m, m->gcl_edge[2 * b + gi], q[0], q[1], q[2], q[3], q[8], q[9][0])) return e;
            if (int e = build_node_layer(m, m->gcl_node[2 * b + gi], q[4], q[5], q[6], q[7])) return e;
            idx += 10;
        }
        const float* const* q = t + idx;       // coord0.w,b coord2.w,b coord4.w
        if (int e = build_edge_layer(m, m->equiv[b], q[0], q[1], q[2], q[3], q[4], 0.f)) return e;
        idx += 5;
    }
    return MCG_OK;
}

extern "C" {

int mcg_egnn_create(const float* const* tensors, int n_tensors, int hidden, int n_blocks, mcg_egnn** out) {
    if (!tensors || !out || hidden != H || n_blocks < 1 || n_tensors != 4 + n_blocks * 25) {
        mcg_set_error("mcg_egnn_create: bad arguments (hidden must be %d, n_tensors = 4 + 25*n_blocks)", H);
        return MCG_ERR_ARG;
    }
    mcg_egnn* m = new mcg_egnn();
    static std::atomic<uint64_t> next_uid{1};
    m->uid = next_uid.fetch_add(1);
    if (int e = egnn_build(m, tensors, n_blocks)) {
        mcg_egnn_destroy(m);          // frees whatever was uploaded before the failure
        return e;
    }
    *out = m;
    return MCG_OK;
}

// mode 0 (default): exact fp32 MFMA.  1: MFMA operands (activations and weights) rounded to bf16, fp32 accumulate and
// epilogue (BASELINE.json configs[4]).  2: "f32x6" split-operand edge contraction (fp32-accurate).
int mcg_egnn_set_precision(mcg_egnn* m, int mode) {
    if (!m) return MCG_ERR_ARG;
    if (mode < 0 || mode > 2) { mcg_set_error("mcg_egnn_set_precision: mode must be 0 (fp32), 1 (bf16) or 2 (f32x6)"); return MCG_ERR_ARG; }
    m->bf16 = mode == 1;
    m->x6 = mode == 2;
    ++m->opt_epoch;
    return MCG_OK;
}

int mcg_egnn_set_option(mcg_egnn* m, int option, int value) {
    if (!m) return MCG_ERR_ARG;
    switch (option) {
        case MCG_OPT_X6_GEMM: m->x6_gemm = value != 0; ++m->opt_epoch; return MCG_OK;
        case MCG_OPT_GEMM_RN: if (value >= 0 && value <= 3) { m->gemm_rn = value; ++m->opt_epoch; return MCG_OK; } break;
        case MCG_OPT_GEMM_X6_RN: if (value >= 0 && value <= 3) { m->gemm_x6_rn = value; ++m->opt_epoch; return MCG_OK; } break;
        case MCG_OPT_GEMM_BF16_LDS: if (value >= 0 && value <= 2) { m->gemm_bf16_lds = value; ++m->opt_epoch; return MCG_OK; } break;
        case MCG_OPT_NODE_FUSED: if (value >= 0 && value <= 2) { m->node_fused = value; ++m->opt_epoch; return MCG_OK; } break;
        default: break;
    }
    mcg_set_error("mcg_egnn_set_option: unknown option %d or value %d out of range", option, value);
    return MCG_ERR_ARG;
}

void mcg_egnn_destroy(mcg_egnn* m) {
    if (!m) return;
    for (void* p : m->allocs) (void)hipFree(p);
    delete m;
}

}  // extern "C"
