// Host-only driver of the plan builder's self-check (mcg_plan_host.cpp), built with AddressSanitizer + UBSan by
// `make -C ml_conformer_generator_amd/csrc asan` (CPU only - GPU sanitizers are not available on this pool).
//   plan_host_check <file>      one batch per line:  N cus edge_mt four_tile_units expect_ok n_1 n_2 ... n_B
// Every batch goes through mcg_plan_build_host + mcg_plan_check_tables and mcg_plan_range_cuts (1..6 parts); the exit code is the number of batches whose
// outcome differs from `expect_ok` (sanitizer reports abort the process on their own).
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "../../ml_conformer_generator_amd/csrc/mcg_plan_host.h"

static char g_err[512];
extern "C" void mcg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <batches.txt>\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "r");
    if (!f) { perror(argv[1]); return 2; }
    char* line = nullptr;
    size_t cap = 0;
    int n_batches = 0, n_bad = 0;
    long rows_total = 0;
    while (getline(&line, &cap, f) > 0) {
        std::istringstream in(line);
        int N, cus, edge_mt, four_tile, expect_ok;
        if (!(in >> N >> cus >> edge_mt >> four_tile >> expect_ok)) continue;
        std::vector<int32_t> nn;
        for (int v; in >> v;) nn.push_back(v);
        if (nn.empty()) continue;
        mcg_plan_opts o;
        memset(&o, 0, sizeof(o));
        o.edge_mt = edge_mt;
        o.four_tile_units = four_tile;
        int32_t info[8];
        g_err[0] = 0;
        const int rc = mcg_plan_check_tables((int)nn.size(), N, nn.data(), &o, cus, info);
        ++n_batches;
        // the molecule-range cuts of the stream split: contiguous, non-empty, covering, for every number of parts
        for (int parts = 1; parts <= 6; ++parts) {
            const std::vector<int> cuts = mcg_plan_range_cuts((int)nn.size(), nn.data(), parts);
            const int want = parts < (int)nn.size() ? parts : (int)nn.size();
            bool ok = (int)cuts.size() == want + 1 && cuts.front() == 0 && cuts.back() == (int)nn.size();
            for (size_t k = 0; ok && k + 1 < cuts.size(); ++k) ok = cuts[k + 1] > cuts[k];
            if (!ok) { ++n_bad; fprintf(stderr, "batch %d: bad range cuts for %d parts\n", n_batches, parts); }
        }
        for (int32_t v : nn) rows_total += (long)v * (v > 0 ? v - 1 : 0);
        if ((rc == 0) != (expect_ok != 0)) {
            ++n_bad;
            fprintf(stderr, "batch %d: rc %d (expected %s): %s\n", n_batches, rc, expect_ok ? "ok" : "an error", g_err);
        }
    }
    free(line);
    fclose(f);
    printf("checked %d batches (%ld edge rows), %d unexpected outcomes\n", n_batches, rows_total, n_bad);
    return n_bad > 125 ? 125 : n_bad;
}
