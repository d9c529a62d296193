// Device half of a batch plan (mcg_plan): uploads the host tables of mcg_plan_host.cpp, allocates the workspace that all
// 101 denoiser calls of a sampling run reuse, and optionally cuts the batch into molecule ranges on separate HIP streams.
#include "mcg_egnn_internal.h"

#include <algorithm>
#include <cstring>

namespace { constexpr int HP = MCG_HP; }

void mcg_plan_mark_done(const mcg_plan* p, void* stream) { mcg_plan_mark(p, (hipStream_t)stream); }
int mcg_plan_B(const mcg_plan* p) { return p->B; }
int mcg_plan_N(const mcg_plan* p) { return p->N; }
const int* mcg_plan_n_nodes(const mcg_plan* p) { return p->n_nodes; }

// `ss`: the non-blocking stream the uploads and the workspace memset go through (the top-level plan's capture stream; idle at this point)
// (`*ss_io` null: the stream is created HERE, behind the host-side checks - argument and size errors are reported without touching the
//  device - and handed back; the caller owns it from then on, also when this function fails later)
static int plan_create_single(int B, int N, const int32_t* n_nodes_host, const mcg_plan_opts* opts, mcg_plan** out, hipStream_t* ss_io) {
    if (B < 1 || N < 1 || !n_nodes_host || !out) {
        mcg_set_error("mcg_plan_create: bad arguments");
        return MCG_ERR_ARG;
    }
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipGetLastError();
    McgPlanHost H;
    if (int e = mcg_plan_build_host(B, N, n_nodes_host, opts, cus, H)) return e;
    if (!*ss_io) MCG_HIP(hipStreamCreateWithFlags(ss_io, hipStreamNonBlocking));
    const hipStream_t ss = *ss_io;
    mcg_plan* p = new mcg_plan();
    p->B = H.B; p->N = H.N; p->M = H.M; p->n_rows = H.n_rows; p->n_mtiles = H.n_mtiles; p->MT = H.MT; p->n_waves = H.n_waves;
    p->n_pslots = H.n_pslots; p->wgc = H.wgc;
    std::vector<int>&nn = H.nn, &node_off = H.node_off, &node_mol = H.node_mol, &wave_poff = H.wave_poff, &ij = H.ij,
                    &node_slots = H.node_slots;
    McgPlanHost::Set (&ht)[2] = H.ht;
    const int n_sets = H.n_sets, best = H.MT, R = 16 * H.MT;
    const bool slots_ok = H.slots_ok, segs_ok = H.segs_ok;
    if (!segs_ok) {
        mcg_set_error("mcg_plan_create: edge_mt = %d puts more than 16 atoms' rows into one %d-row unit (molecules this "
                      "small need edge_mt = 1)", best, R);
        delete p;
        return MCG_ERR_ARG;
    }
    if (!slots_ok) {
        mcg_set_error("mcg_plan_create: a molecule's edge rows span more than 8 tiles per atom (N > 114 is not supported)");
        delete p;
        return MCG_ERR_ARG;
    }
    int max_uslots = 0;
    for (int k = 0; k < n_sets; ++k) max_uslots = std::max(max_uslots, ht[k].n_uslots);
    for (int k = 0; k < n_sets; ++k)
        for (int& v : ht[k].node_slots) if (v < 0) v = max_uslots;       // the zero row (never written) is common to both sets
    // ---- ONE block of tables (one upload) and ONE block of workspace (one memset), both from the plan pool
    // (mcg_devmem.hip): a caller of a ragged workload builds a plan per call, and 20 hipMalloc + 20 synchronous copies per
    // plan were most of what its first call cost.
    struct Tab { const std::vector<int>* v; int** dst; };
    int *d_ij = nullptr, *d_wi[2] = {nullptr, nullptr}, *d_ns[2] = {nullptr, nullptr}, *d_ps4 = nullptr;
    // an atom's first four per-unit slots as one int4 (unused = the zero row behind the last slot)
    std::vector<int> ps4((size_t)std::max(p->M, 1) * 4, p->n_pslots);
    p->pspan = 0;
    for (int v = 0; v < p->M; ++v) {
        int cnt = 0;
        for (int k = 0; k < 8; ++k) {
            const int sl = node_slots[(size_t)v * 8 + k];
            if (sl < 0) continue;
            if (cnt < 4) ps4[(size_t)v * 4 + cnt] = sl;
            ++cnt;
        }
        p->pspan = std::max(p->pspan, cnt);
    }
    std::vector<Tab> tabs = {{&ij, &d_ij}, {&node_slots, &p->node_slots}, {&nn, &p->n_nodes}, {&node_off, &p->node_off},
                             {&wave_poff, &p->wave_poff}, {&node_mol, &p->node_mol}, {&ps4, &d_ps4}};
    for (int k = 0; k < n_sets; ++k) { tabs.push_back({&ht[k].wg_info, &d_wi[k]}); tabs.push_back({&ht[k].node_slots, &d_ns[k]}); }
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t tab_bytes = 0;
    std::vector<size_t> tab_off;
    for (const Tab& t : tabs) { tab_off.push_back(tab_bytes); tab_bytes += al(std::max<size_t>(t.v->size(), 1) * sizeof(int)); }
    std::vector<char> stage(tab_bytes, 0);
    for (size_t k = 0; k < tabs.size(); ++k)
        if (!tabs[k].v->empty()) memcpy(stage.data() + tab_off[k], tabs[k].v->data(), tabs[k].v->size() * sizeof(int));
    void* tab_block = nullptr;
    if (int e = mcg_dev_alloc(tab_bytes, &tab_block)) { delete p; return e; }
    p->allocs.push_back(tab_block);
    // (NOT through the legacy stream - hipMemcpy / hipMemset: on ROCm 7.2 those FAIL while any other thread of the process is inside a
    //  stream capture, thread-local or not (round 6, test_plans_destroyed_on_another_thread_while_this_one_captures_graphs: 23 of 25
    //  runs) - but through the plan's own non-blocking capture stream.  Not a process-wide setup stream either: a persistent extra
    //  stream shifts the process's streams over the hardware queues and cost two rank processes sharing one GPU 15 %.)
    if (hipMemcpyAsync(tab_block, stage.data(), tab_bytes, hipMemcpyHostToDevice, ss) != hipSuccess || hipStreamSynchronize(ss) != hipSuccess) {
        (void)hipGetLastError();
        mcg_set_error("mcg_plan_create: table upload failed");
        mcg_plan_destroy(p);
        return MCG_ERR_HIP;
    }
    for (size_t k = 0; k < tabs.size(); ++k) *tabs[k].dst = reinterpret_cast<int*>((char*)tab_block + tab_off[k]);
    p->row_ij = reinterpret_cast<int2*>(d_ij);
    p->pslots4 = reinterpret_cast<int4*>(d_ps4);
    for (int k = 0; k < n_sets; ++k) {
        mcg_plan::UnitTables& T = p->ut[k];
        T.wg_info = reinterpret_cast<int4*>(d_wi[k]);
        T.node_slots = reinterpret_cast<int4*>(d_ns[k]);
        T.n_units = ht[k].n_units; T.n_full_wg = ht[k].n_full; T.n_uslots = ht[k].n_uslots; T.max_span = ht[k].span;
    }
    p->have_alt = n_sets == 2;
    const size_t M1 = (size_t)(p->M > 0 ? p->M : 1);
    struct { float** ptr; size_t n; } bufs[] = {
        // (+64 floats: the bf16 kernels read activation rows up to k = 447, i.e. 16 floats past the last row)
        {&p->x, M1 * 4}, {&p->x0, M1 * 4}, {&p->h, M1 * HP + 64}, {&p->h2, M1 * HP + 64}, {&p->pab, M1 * MCG_PAB_BLOCKED_FLOATS + 64},
        {&p->agg, M1 * HP + 64}, {&p->t1, M1 * HP + 64}, {&p->P, (size_t)(p->n_pslots + 1) * HP + 64}, {&p->Px, (size_t)(p->n_pslots + 1) * 4},
        {&p->U, (size_t)(max_uslots + 1) * HP + 64}, {&p->Ux, (size_t)(max_uslots + 1) * 4}};
    size_t ws_bytes = 0;
    for (auto& b : bufs) ws_bytes += al(b.n * sizeof(float));
    void* ws_block = nullptr;
    if (int e = mcg_dev_alloc(ws_bytes, &ws_block)) { mcg_plan_destroy(p); return e; }
    p->allocs.push_back(ws_block);
    // (the plan's own streams are non-blocking, i.e. they do not order with stream 0: wait for the memset here)
    if (hipMemsetAsync(ws_block, 0, ws_bytes, ss) != hipSuccess || hipStreamSynchronize(ss) != hipSuccess) {
        (void)hipGetLastError();
        mcg_set_error("mcg_plan_create: workspace memset failed");
        mcg_plan_destroy(p);
        return MCG_ERR_HIP;
    }
    size_t off = 0;
    for (auto& b : bufs) { *b.ptr = reinterpret_cast<float*>((char*)ws_block + off); off += al(b.n * sizeof(float)); }
    *out = p;
    return MCG_OK;
}

extern "C" {

void mcg_plan_destroy(mcg_plan* p) {
    if (!p) return;
    // the blocks go back to the plan pool, not to the driver: nothing may still be running on them when another plan can
    // take them (hipFree used to wait implicitly).  Once, up front, before the molecule ranges hand back theirs: on the
    // plan's OWN device (the caller's current device may be another one - a generator on cuda:1, a GC thread), and only for
    // the plan's own last launch (ev_done, recorded behind every entry point), not for the whole device.
    int prev_dev = -1;
    if (!p->is_sub) {
        if (hipGetDevice(&prev_dev) != hipSuccess) { (void)hipGetLastError(); prev_dev = -1; }
        if (p->dev >= 0 && prev_dev != p->dev && hipSetDevice(p->dev) != hipSuccess) (void)hipGetLastError();
        if (p->ev_done) {
            if (p->ev_pending && hipEventSynchronize(p->ev_done) != hipSuccess) { (void)hipGetLastError(); (void)hipDeviceSynchronize(); }
        } else if (!p->allocs.empty()) {
            (void)hipDeviceSynchronize();           // a plan that never got its event (failed half-way through creation)
        }
        for (hipStream_t st : p->streams) (void)hipStreamSynchronize(st);       // joined behind ev_done already: returns at once
    }
    if (p->graph_exec) (void)hipGraphExecDestroy(p->graph_exec);
    for (hipGraphExec_t ge : p->retired_graphs) (void)hipGraphExecDestroy(ge);
    if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
    for (mcg_plan* q : p->subs) mcg_plan_destroy(q);
    for (hipStream_t st : p->streams) (void)hipStreamDestroy(st);
    for (hipEvent_t e : p->ev_join) (void)hipEventDestroy(e);
    if (p->ev_fork) (void)hipEventDestroy(p->ev_fork);
    for (void* q : p->allocs) mcg_dev_free(q);
    if (p->ev_done) (void)hipEventDestroy(p->ev_done);
    const bool restore = !p->is_sub && prev_dev >= 0 && prev_dev != p->dev;
    delete p;
    if (restore && hipSetDevice(prev_dev) != hipSuccess) (void)hipGetLastError();
}

static int plan_finish(mcg_plan* p, int B, int N, const int32_t* n_nodes_host, const mcg_plan_opts* opts);

int mcg_plan_create(int B, int N, const int32_t* n_nodes_host, int edge_mt, mcg_plan** out) {
    mcg_plan_opts o{};
    o.edge_mt = edge_mt;
    return mcg_plan_create_ex(B, N, n_nodes_host, &o, out);
}

int mcg_plan_create_ex(int B, int N, const int32_t* n_nodes_host, const mcg_plan_opts* opts, mcg_plan** out) {
    const int n_ranges = opts ? opts->n_ranges : 0;
    if (n_ranges < 0 || n_ranges > 4) { mcg_set_error("mcg_plan_create_ex: n_ranges must be 0 (auto) .. 4"); return MCG_ERR_ARG; }
    if (opts)
        for (int k = 0; k < 5; ++k)
            if (opts->reserved[k] != 0) { mcg_set_error("mcg_plan_create_ex: mcg_plan_opts.reserved must be zero"); return MCG_ERR_ARG; }
    mcg_plan* p = nullptr;
    hipStream_t cs = nullptr;             // the plan's capture stream: created inside, plan creation uploads through it
    if (int e = plan_create_single(B, N, n_nodes_host, opts, &p, &cs)) { if (cs) (void)hipStreamDestroy(cs); return e; }
    p->cap_stream = cs;
    if (hipGetDevice(&p->dev) != hipSuccess) { (void)hipGetLastError(); p->dev = -1; }
    if (hipEventCreateWithFlags(&p->ev_done, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); p->ev_done = nullptr; }
    if (int e = plan_finish(p, B, N, n_nodes_host, opts)) {
        mcg_plan_destroy(p);          // streams, events, sub-plans and buffers created so far
        return e;
    }
    *out = p;
    return MCG_OK;
}

// graph staging buffer, capture stream and the optional split into molecule ranges
static int plan_finish(mcg_plan* p, int B, int N, const int32_t* n_nodes_host, const mcg_plan_opts* opts) {
    const int n_ranges = opts ? opts->n_ranges : 0;
    if (int e = mcg_dev_alloc((size_t)B * sizeof(float), (void**)&p->t_buf)) return e;
    p->allocs.push_back(p->t_buf);
    // Split the batch into `parts` molecule ranges of ~equal edge count, one HIP stream each: the launch-bound node GEMMs
    // of one range run under another range's edge kernel, and the ramps / tails of the edge kernels overlap.
    // Exact-fp32 plans (16-row tiles), measured per denoiser call (tools/bench_kernels.py --ranges; W = workgroup-equivalents =
    // tiles / 4, 27-atom molecules unless noted):  W = 702 (config 2): 4.54 / 4.96 / 4.93 ms with 1 / 2 / 3 ranges;
    // 1 053: 6.85 / 6.42 / 6.83;  1 229: 7.93 / 7.55 / 7.60;  1 404: 8.79 / 8.84 / 8.49;  2 106: 13.01 / 12.94 / 12.66;
    // 2 808: 16.92 / 17.12 / 16.05 (4: 16.72);  3 005 (config 3 shape, ragged): - / 17.37 / 17.11 (4: 18.0);
    // 4 212: 3 ranges 24.99, 4: 23.94, 5: 26.7.  Hence the table below.  Other plans (64-row units of the bf16 / split-operand
    // kernels) keep round 1's rule; the split-operand modes' host mirror asks for two ranges explicitly (f32x6 at
    // config 2: 3.53 -> 3.18 ms per call - their edge kernel is short against the node phase).  Round 5, 64-row bf16 plans with
    // the LDS-staged node GEMM, ms per call with 1 / 2 / 3 ranges at the 256-ragged shape (12 020 tiles): 4.60 / 4.63 / 5.45
    // (round 4, 32-row GEMM: 5.55 / 5.25 / 5.74).  With the final round-5 kernels (blocked layer-1 inputs, single-product
    // aggregation) the same shape runs 4.36 / 4.08 / 5.02, and 27-atom batches of 96 / 128 / 160 / 192 molecules 2.25 / 2.26,
    // 2.54 / 2.73, 2.98 / 3.16, 3.51 / 3.30 with 1 / 2 ranges (tools/ab_bf16_ranges.sh): two ranges pay exactly when each
    // half still has the 80 row blocks of 32 atoms from which the bf16 node GEMMs take the LDS-staged kernel
    // (MCG_LDSG_MIN_ROWBLOCKS, mcg_gemm.h) - 5 120 atoms in all.  Round 6, with the node phase of a layer as one fused launch from
    // 32 row blocks on (mcg_node_fused.h): 27-atom batches of 48 / 64 / 80 / 96 / 112 / 128 / 256 / 320 molecules 1.62 / 1.60, 1.64 / 1.80,
    // 1.88 / 1.70, 2.13 / 1.94, 2.15 / 2.11, 2.38 / 2.20, 3.78 / 3.66 (3: 3.82), 5.05 / 4.43 (3: 4.47) ms with 1 / 2 ranges, the 256-ragged
    // shape 4.16 / 3.93 / 4.12: the same law one level down - two ranges pay when each half keeps the 32 row blocks from which the
    // fused launch runs (2 048 atoms in all); three never do up to 8 640 atoms.
    constexpr int BF16_TWO_RANGES_FROM_ATOMS = 2 * 32 * 32;
    int parts = 1;
    if (n_ranges > 0) parts = n_ranges;
    else if (p->MT == 1) parts = p->n_mtiles < 3600 ? 1 : p->n_mtiles < 5200 ? 2 : p->n_mtiles < 14000 ? 3 : 4;
    else parts = p->M >= BF16_TWO_RANGES_FROM_ATOMS ? 2 : 1;
    if (parts < 2 || B < 2 * parts || p->n_rows < 4096) return MCG_OK;
    const std::vector<int> cuts = mcg_plan_range_cuts(B, n_nodes_host, parts);     // host-only, sanitizer-covered
    for (size_t k = 0; k + 1 < cuts.size(); ++k) {
        const int b0 = cuts[k], b1 = cuts[k + 1];
        mcg_plan* sub = nullptr;
        if (int e = plan_create_single(b1 - b0, N, n_nodes_host + b0, opts, &sub, &p->cap_stream)) return e;
        sub->is_sub = true;            // destroyed with its parent, which synchronises the device once for all of them
        p->subs.push_back(sub);
        p->sub_b0.push_back(b0);
        hipStream_t st; hipEvent_t ev;
        MCG_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        MCG_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        p->streams.push_back(st);
        p->ev_join.push_back(ev);
    }
    MCG_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
    return MCG_OK;
}

// mode: -1 auto, 0 four-tile units only, 1 the stand-alone column-split kernel
int mcg_plan_set_latency_mode(mcg_plan* p, int mode) {
    if (!p || mode < -1 || mode > 1) return MCG_ERR_ARG;
    p->latency_mode = mode;
    for (mcg_plan* q : p->subs) q->latency_mode = mode;
    if (p->graph_exec) { (void)hipGraphExecDestroy(p->graph_exec); p->graph_exec = nullptr; }   // re-capture
    return MCG_OK;
}

int mcg_plan_ranges(const mcg_plan* p) { return p ? (p->subs.empty() ? 1 : (int)p->subs.size()) : 0; }

int mcg_plan_info(const mcg_plan* p, int32_t* info /*[8]*/) {
    if (!p || !info) return MCG_ERR_ARG;
    info[0] = p->M; info[1] = p->n_rows; info[2] = p->MT; info[3] = p->n_waves; info[4] = p->n_pslots;
    info[5] = p->B; info[6] = p->N; info[7] = p->n_mtiles;
    return MCG_OK;
}

// Debug/test hook: copy one of the plan's internal buffers out.  which: 0 h[M][432], 1 pab[M][864], 2 agg[M][432],
// 3 t1[M][432], 4 x[M][4]
int mcg_plan_peek(const mcg_plan* pl, int which, float* dst, void* stream) {
    if (!pl || !dst) return MCG_ERR_ARG;
    mcg_plan_mark_guard done{pl, (hipStream_t)stream};
    const float* src = nullptr; size_t n = 0;
    switch (which) {
        case 0: src = pl->h; n = (size_t)pl->M * HP; break;
        case 1: src = pl->pab; n = (size_t)pl->M * 2 * HP; break;
        case 2: src = pl->agg; n = (size_t)pl->M * HP; break;
        case 3: src = pl->t1; n = (size_t)pl->M * HP; break;
        case 4: src = pl->x; n = (size_t)pl->M * 4; break;
        default: return MCG_ERR_ARG;
    }
    MCG_HIP(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MCG_OK;
}

}  // extern "C"
