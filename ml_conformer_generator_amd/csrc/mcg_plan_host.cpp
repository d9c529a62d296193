// HOST half of mcg_plan_create (see mcg_plan_host.h) and its self-check.  Plain C++: no HIP call, no device type.
#include "mcg_plan_host.h"

#include <algorithm>

int mcg_plan_build_host(int B, int N, const int32_t* n_nodes_host, const mcg_plan_opts* opts, int cus, McgPlanHost& H) {
    if (B < 1 || N < 1 || !n_nodes_host) { mcg_set_error("mcg_plan_create: bad arguments"); return MCG_ERR_ARG; }
    const int edge_mt = opts ? opts->edge_mt : 0;
    const int four_tile = opts ? opts->four_tile_units : 0;
    if (edge_mt != 0 && edge_mt != 1 && edge_mt != 4) { mcg_set_error("mcg_plan_create: edge_mt must be 0 (auto), 1 (16-row tiles) or 4 (64-row units)"); return MCG_ERR_ARG; }
    McgPlanHost* const p = &H;
    p->B = B; p->N = N;
    std::vector<int>&nn = H.nn, &node_off = H.node_off, &row_off = H.row_off;
    nn.assign(B, 0); node_off.assign(B + 1, 0); row_off.assign(B + 1, 0);
    for (int b = 0; b < B; ++b) {
        if (n_nodes_host[b] < 0 || n_nodes_host[b] > N) {
            mcg_set_error("mcg_plan_create: n_nodes[%d]=%d outside [0,%d]", b, n_nodes_host[b], N);
            return MCG_ERR_ARG;
        }
        nn[b] = n_nodes_host[b];
        node_off[b + 1] = node_off[b] + nn[b];
        row_off[b + 1] = row_off[b] + nn[b] * (nn[b] > 0 ? nn[b] - 1 : 0);
        // 32-bit row / byte offsets (buffer-descriptor addressing): <= 1e6 atoms and 2^30 edge rows per plan,
        // i.e. ~37 000 molecules of 27 atoms - shard larger batches over plans / ranks
        if (node_off[b + 1] > 1000000 || row_off[b + 1] > (1 << 30)) {
            mcg_set_error("mcg_plan_create: batch too large for one plan (%d atoms after molecule %d; limit 1e6 atoms, 2^30 edge rows)",
                          node_off[b + 1], b);
            return MCG_ERR_ARG;
        }
    }
    p->M = node_off[B];
    p->n_rows = row_off[B];
    p->n_mtiles = (p->n_rows + 15) / 16;
    // rows per unit: 16 (MT = 1: the exact-fp32 kernels; 2 workgroups per CU, whose waves cover each other's barrier /
    // epilogue bubbles - 32 rows per wave measured slower at configs 2 and 3 and was removed) or 64 (MT = 4: the
    // 64-row units of the bf16 / f32x6 kernels)
    const int best = edge_mt == 4 ? 4 : 1;
    p->MT = best;
    p->n_waves = (p->n_mtiles + best - 1) / best;
    const int R = 16 * best;

    std::vector<int>&node_mol = H.node_mol, &wave_poff = H.wave_poff;
    node_mol.assign(p->M, 0); wave_poff.assign(p->n_waves + 1, 0);
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < nn[b]; ++i) node_mol[node_off[b] + i] = b;
    // (i, j) of every edge row, padded to whole tiles, + the row's SEGMENT inside its unit (R rows): the rank of
    // node i among the nodes that own rows of the unit.  Counting only row-owning nodes keeps a 16-row tile at
    // <= 16 segments whatever the molecule sizes (1-atom molecules own node indices but no rows); every epilogue
    // handles segment ids 0..15, so wider units (edge_mt 2 / 4) are refused when a unit would need more.
    // Per-node partial-slot table node_slots[v][k] (ascending unit order = the order the sums are taken in).
    std::vector<int>&ij = H.ij, &node_slots = H.node_slots;
    ij.assign((size_t)(p->n_mtiles > 0 ? p->n_mtiles : 1) * 32, -1);
    node_slots.assign((size_t)p->M * 8, -1);
    bool &slots_ok = H.slots_ok, &segs_ok = H.segs_ok;
    slots_ok = segs_ok = true;
    {
        std::vector<int> unit_nseg(p->n_waves + 1, 0);
        std::vector<int> row_seg((size_t)p->n_rows, 0);
        int cur_unit = -1, cur_node = -1, seg = -1;
        for (int b = 0; b < B; ++b) {
            const int n = nn[b];
            for (int i = 0; i < n && n > 1; ++i)
                for (int jj = 0; jj < n - 1; ++jj) {
                    const int r = row_off[b] + i * (n - 1) + jj;
                    const int v = node_off[b] + i;
                    const int u = r / R;
                    if (u != cur_unit) { cur_unit = u; cur_node = v; seg = 0; }
                    else if (v != cur_node) { cur_node = v; ++seg; }
                    unit_nseg[u] = seg + 1;
                    row_seg[r] = seg;
                }
        }
        for (int w = 0; w < p->n_waves; ++w) {
            if (unit_nseg[w] > 16) segs_ok = false;
            wave_poff[w + 1] = wave_poff[w] + unit_nseg[w];
        }
        p->n_pslots = wave_poff[p->n_waves];
        for (int b = 0; b < B && slots_ok; ++b) {
            const int n = nn[b];
            for (int i = 0; i < n && n > 1; ++i) {
                const int v = node_off[b] + i;
                const int first = row_off[b] + i * (n - 1);
                const int w_lo = first / R, w_hi = (first + n - 2) / R;
                if (w_hi - w_lo + 1 > 8) { slots_ok = false; break; }
                for (int w = w_lo; w <= w_hi; ++w) {
                    const int r = std::max(first, w * R);         // the node's first row inside unit w
                    node_slots[(size_t)v * 8 + (w - w_lo)] = wave_poff[w] + row_seg[r];
                }
                for (int jj = 0; jj < n - 1; ++jj) {
                    const size_t r = (size_t)first + jj;
                    ij[2 * r] = v;
                    ij[2 * r + 1] = (node_off[b] + jj + (jj >= i ? 1 : 0)) | (row_seg[r] << 24);
                }
            }
        }
    }
    // workgroup-level tables (MT = 1).  A "unit" is what one workgroup of the throughput kernel processes:
    //   unit w <  n_full : tiles 4w .. 4w+3 = rows [64w, 64w + 64), one tile per wave (the LDS-staged body)
    //   unit w >= n_full : ONE tile, 4 * n_full + (w - n_full), its columns split over the 4 waves (the quarter-tile
    //                      body: the last, partly filled round of the chip runs 4x more workgroups with 4x shorter chains)
    std::vector<int> wave_ws(p->n_waves + 1, 0), unit_sbase, wg_info;
    std::vector<int> node_slots2((size_t)(p->M > 0 ? p->M : 1) * 4, 0);
    std::vector<int> first_node(p->n_waves, -1), last_node(p->n_waves, -1);
    bool wgc_ok = best == 1 && p->n_waves > 0;
    if (wgc_ok)
        for (int b = 0; b < B; ++b) {
            const int n = nn[b];
            for (int i = 0; i < n && n > 1; ++i) {
                const int v = node_off[b] + i;
                const int first = row_off[b] + i * (n - 1), last = first + n - 2;
                for (int u = first / 16; u <= last / 16; ++u) {
                    if (first_node[u] < 0) first_node[u] = v;
                    last_node[u] = v;
                }
            }
        }
    const int n_wg_all = (p->n_waves + 3) / 4;
    int span = 2;
    int n_units = 0;
    // tables for `n_full` four-tile units followed by one-tile units; false when an atom's rows would span more than
    // `max_span` units or a four-tile unit touches more than 16 atoms
    auto build_units = [&](int n_full, int max_span) -> bool {
        const int n_tail = n_full < n_wg_all ? p->n_waves - 4 * n_full : 0;
        n_units = n_full + n_tail;
        unit_sbase.assign(n_units + 1, 0);
        wg_info.assign((size_t)n_units * 4, 0);
        span = 2;
        for (int w = 0; w < n_units; ++w) {
            const int u0 = w < n_full ? 4 * w : 4 * n_full + (w - n_full);
            const int u1 = w < n_full ? std::min(4 * w + 4, p->n_waves) : u0 + 1;
            int slots = 0, rows = 0;                // rows: LDS rows the waves park their segment sums in (<= 16 fit)
            int ws_pack = 0, ns_pack = 0;
            for (int u = u0; u < u1; ++u) {
                const int nseg = wave_poff[u + 1] - wave_poff[u];
                rows += nseg;
                const bool cont = u > u0 && nseg > 0 && first_node[u] == last_node[u - 1];
                const int ws0 = cont ? slots - 1 : slots;
                wave_ws[u] = ws0;
                slots = ws0 + nseg;
                ws_pack |= (ws0 & 0xff) << (8 * (u - u0));
                ns_pack |= (nseg & 0xff) << (8 * (u - u0));
            }
            if (rows > 16) return false;
            wg_info[4 * (size_t)w] = unit_sbase[w]; wg_info[4 * (size_t)w + 1] = slots;
            wg_info[4 * (size_t)w + 2] = ws_pack; wg_info[4 * (size_t)w + 3] = ns_pack;
            unit_sbase[w + 1] = unit_sbase[w] + slots;
        }
        for (size_t k = 0; k < node_slots2.size(); ++k) node_slots2[k] = -1;     // unused: patched to the zero row below
        auto unit_of = [&](int u) { return u < 4 * n_full ? u / 4 : n_full + (u - 4 * n_full); };
        for (int b = 0; b < B; ++b) {
            const int n = nn[b];
            for (int i = 0; i < n && n > 1; ++i) {
                const int v = node_off[b] + i;
                const int first = row_off[b] + i * (n - 1), last = first + n - 2;
                int cnt = 0, prev = -1;
                for (int u = first / 16; u <= last / 16; ++u) {
                    const int w = unit_of(u);
                    if (w == prev) continue;
                    prev = w;
                    if (cnt == max_span) return false;
                    // slot of atom v inside unit w = slot base of the tile holding its first row there + its segment there
                    const int r = std::max(first, 16 * u);
                    node_slots2[4 * (size_t)v + cnt++] = unit_sbase[w] + wave_ws[u] + (ij[2 * (size_t)r + 1] >> 24);
                }
                span = std::max(span, cnt);
            }
        }
        return true;
    };
    // table sets: [0] the automatic split, [1] four-tile units only (when different)
    McgPlanHost::Set (&ht)[2] = H.ht;
    int& n_sets = H.n_sets;
    n_sets = 0;
    auto keep = [&](int n_full) {
        McgPlanHost::Set& t = ht[n_sets++];
        t.wg_info = wg_info; t.node_slots = node_slots2; t.n_units = n_units; t.n_full = n_full;
        t.n_uslots = unit_sbase[n_units]; t.span = span;
    };
    if (wgc_ok) {
        // Only COMPLETE rounds of the chip (2 resident workgroups per CU) take the four-tile body: in a partly filled last
        // round every SIMD would walk a whole tile's 93 k-cycle MFMA chain with part of the chip idle.  The quarter-tile
        // body costs ~18 % more SIMD time per tile (4x the row decode / prologue / epilogue per tile), so a last round that
        // is more than ~80 % full (r > 400 of 512 workgroups) stays with four-tile units.  Measured per edge launch
        // (tools/tail_sweep.sh, 27-atom molecules): 702 workgroups 156.9 -> 148.3 us, 1053: 242.6 -> 209.4 us,
        // 351: 101.0 -> 88.7 us, 44: 55.8 -> 17.5 us.
        // mcg_plan_opts::four_tile_units overrides the rule: -1 none, n > 0 the first n (rounded up to a multiple of 8,
        // capped at all of them) - measurement and tests.
        const int round = 2 * (cus > 0 ? cus : 256);
        const int r = n_wg_all % round;
        int n_full = r * 512 <= 400 * round ? n_wg_all - r : n_wg_all;
        if (four_tile < 0) n_full = 0;
        else if (four_tile > 0) n_full = (int)std::min<long>((((long)four_tile + 7) / 8) * 8, (long)n_wg_all);
        if (n_full < n_wg_all && build_units(n_full, 4)) keep(n_full);
        wgc_ok = build_units(n_wg_all, 2);
        if (wgc_ok) keep(n_wg_all);
        else if (n_sets > 0) wgc_ok = true;         // (four-tile units alone would touch > 16 atoms: tiny molecules)
        else if (build_units(0, 4)) { keep(0); wgc_ok = true; }     // ... then quarter-tile units only (16 rows: <= 16 atoms)
    }
    p->wgc = wgc_ok;
    return MCG_OK;
}

// Host-only self-check of a plan's tables (no GPU call: it runs on a CPU-only box and is what the CPU tests drive):
// builds them as mcg_plan_create_ex would for a device with `cus` compute units and verifies, independently of how they
// were built, that every edge row's (unit, tile, segment) lands in a slot that its atom lists, that no slot is shared
// by two atoms, that a four-tile unit parks at most 16 rows, and that the row table names the right (i, j).
extern "C" int mcg_plan_check_tables(int B, int N, const int32_t* n_nodes_host, const mcg_plan_opts* opts, int cus, int32_t* info /*[8]*/) {
    if (B < 1 || N < 1 || !n_nodes_host || !info) { mcg_set_error("mcg_plan_check_tables: bad arguments"); return MCG_ERR_ARG; }
    McgPlanHost H;
    if (int e = mcg_plan_build_host(B, N, n_nodes_host, opts, cus, H)) return e;
    const int n_waves = H.n_waves, M = H.M, MT = H.MT;
    const bool wgc = H.wgc;
    for (int k = 0; k < 8; ++k) info[k] = 0;
    info[0] = H.n_sets;
    if (!H.segs_ok || !H.slots_ok) { mcg_set_error("mcg_plan_check_tables: the batch does not fit the requested edge_mt"); return MCG_ERR_ARG; }
    auto fail = [&](const char* what, int a, int b2) { mcg_set_error("mcg_plan_check_tables: %s (%d, %d)", what, a, b2); return MCG_ERR_STATE; };
    // the row table
    for (int b = 0; b < B; ++b) {
        const int n = H.nn[b];
        for (int i = 0; i < n && n > 1; ++i)
            for (int jj = 0; jj < n - 1; ++jj) {
                const size_t r = (size_t)H.row_off[b] + (size_t)i * (n - 1) + jj;
                if (H.ij[2 * r] != H.node_off[b] + i) return fail("row table: wrong i", b, i);
                if ((H.ij[2 * r + 1] & 0xffffff) != H.node_off[b] + jj + (jj >= i ? 1 : 0)) return fail("row table: wrong j", b, i);
            }
    }
    if (!wgc) return MCG_OK;
    if (MT != 1) return fail("workgroup-level tables on a plan that is not 16-row", MT, 0);
    const int n_wg_all = (n_waves + 3) / 4;
    for (int k = 0; k < H.n_sets; ++k) {
        const McgPlanHost::Set& T = H.ht[k];
        const int n_full = T.n_full;
        const int n_tail = n_full < n_wg_all ? n_waves - 4 * n_full : 0;
        if (T.n_units != n_full + n_tail) return fail("unit count", T.n_units, n_full + n_tail);
        if ((int)T.wg_info.size() != 4 * T.n_units) return fail("wg_info size", (int)T.wg_info.size(), T.n_units);
        int run = 0;
        for (int w = 0; w < T.n_units; ++w) {
            if (T.wg_info[4 * (size_t)w] != run) return fail("slot base not a running sum", w, run);
            run += T.wg_info[4 * (size_t)w + 1];
            if (w < n_full) {
                int rows = 0;
                for (int lt = 0; lt < 4; ++lt) rows += (T.wg_info[4 * (size_t)w + 3] >> (8 * lt)) & 0xff;
                if (rows > 16) return fail("a four-tile unit parks more than 16 rows", w, rows);
            }
        }
        if (run != T.n_uslots) return fail("slot total", run, T.n_uslots);
        std::vector<int> owner((size_t)T.n_uslots, -1), hits((size_t)M * 4, 0);
        int span_seen = 0;
        for (int b = 0; b < B; ++b) {
            const int n = H.nn[b];
            for (int i = 0; i < n && n > 1; ++i) {
                const int v = H.node_off[b] + i;
                for (int jj = 0; jj < n - 1; ++jj) {
                    const size_t r = (size_t)H.row_off[b] + (size_t)i * (n - 1) + jj;
                    const int u = (int)(r / 16);
                    const int w = u < 4 * n_full ? u / 4 : n_full + (u - 4 * n_full);
                    const int lt = w < n_full ? u - 4 * w : 0;
                    const int seg = H.ij[2 * r + 1] >> 24;
                    const int ws0 = (T.wg_info[4 * (size_t)w + 2] >> (8 * lt)) & 0xff, ns = (T.wg_info[4 * (size_t)w + 3] >> (8 * lt)) & 0xff;
                    if (seg < 0 || seg >= ns) return fail("segment id outside its tile's count", (int)r, seg);
                    if (ws0 + seg >= T.wg_info[4 * (size_t)w + 1]) return fail("slot beyond the unit's count", (int)r, ws0 + seg);
                    const int slot = T.wg_info[4 * (size_t)w] + ws0 + seg;
                    if (owner[slot] >= 0 && owner[slot] != v) return fail("slot shared by two atoms", slot, v);
                    owner[slot] = v;
                    int found = -1;
                    for (int q = 0; q < 4; ++q) if (T.node_slots[4 * (size_t)v + q] == slot) found = q;
                    if (found < 0) return fail("row's slot missing from its atom's list", (int)r, slot);
                    hits[4 * (size_t)v + found] = 1;
                }
                int used = 0;
                for (int q = 0; q < 4; ++q) {
                    const int sl = T.node_slots[4 * (size_t)v + q];
                    if (sl >= 0) { if (!hits[4 * (size_t)v + q]) return fail("atom lists a slot none of its rows writes", v, sl); ++used; }
                    for (int q2 = 0; q2 < q; ++q2) if (sl >= 0 && T.node_slots[4 * (size_t)v + q2] == sl) return fail("atom lists a slot twice", v, sl);
                }
                span_seen = std::max(span_seen, used);
            }
        }
        for (int sl = 0; sl < T.n_uslots; ++sl) if (owner[sl] < 0) return fail("slot that no row writes", sl, k);
        if (span_seen > T.span) return fail("an atom owns more rows of U than the set says", span_seen, T.span);
        if (k == 0) { info[1] = T.n_units; info[2] = T.n_full; info[3] = T.n_uslots; info[4] = T.span; }
        else { info[5] = T.n_units; info[6] = T.n_uslots; info[7] = T.span; }
    }
    return MCG_OK;
}

std::vector<int> mcg_plan_range_cuts(int B, const int32_t* n_nodes_host, int parts) {
    std::vector<int> cuts{0};
    if (B < 1 || !n_nodes_host) return cuts;
    if (parts > B) parts = B;
    if (parts < 1) parts = 1;
    std::vector<long> cum((size_t)B + 1, 0);
    for (int b = 0; b < B; ++b) cum[b + 1] = cum[b] + (long)n_nodes_host[b] * (n_nodes_host[b] > 0 ? n_nodes_host[b] - 1 : 0);
    int b0 = 0;
    for (int k = 0; k < parts; ++k) {
        int b1 = B;
        if (k + 1 < parts) {
            const long target = cum[B] * (k + 1) / parts;
            b1 = b0 + 1;
            while (b1 < B && cum[b1] < target) ++b1;
            const int last = B - (parts - 1 - k);        // leave one molecule for each of the remaining ranges
            if (b1 > last) b1 = last;
        }
        cuts.push_back(b1);
        b0 = b1;
    }
    return cuts;
}
