// HOST half of a batch plan: everything of mcg_plan_create that is plain host data - offsets, the edge-row table, the
// unit tables of the throughput edge kernel - built without touching the GPU.  No HIP types: mcg_plan_host.cpp is also
// compiled by the host compiler with AddressSanitizer / UBSan (`make asan`, tests/test_host_logic.py).
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/mlconfgen_hip.h"
#include "mcg_error.h"

struct McgPlanHost {
    // geometry
    int B = 0, N = 0, M = 0, n_rows = 0, n_mtiles = 0, MT = 1, n_waves = 0, n_pslots = 0;
    bool wgc = false;               // workgroup-level tables (`ht`) are valid
    // tables (uploaded as they are by mcg_egnn_plan.hip)
    std::vector<int> nn, node_off, row_off, node_mol, wave_poff, ij, node_slots;
    struct Set { std::vector<int> wg_info, node_slots; int n_units = 0, n_full = 0, n_uslots = 0, span = 2; };
    Set ht[2];                      // [0]: the automatic split into four-tile and quarter-tile units, [1]: four-tile units only
    int n_sets = 0;
    bool slots_ok = true, segs_ok = true;
};

// `cus`: compute units of the device the plan is for (a round of the chip = 2 resident workgroups per CU).
// `opts` may be null (defaults).  Returns MCG_OK or MCG_ERR_ARG with mcg_set_error text.
int mcg_plan_build_host(int B, int N, const int32_t* n_nodes_host, const mcg_plan_opts* opts, int cus, McgPlanHost& H);

// Cut a batch into `parts` contiguous molecule ranges of ~equal edge-row count (mcg_egnn_plan.hip runs each on its own HIP
// stream).  Returns the range starts b0[0..parts'] with b0[0] = 0 and b0[parts'] = B; every range holds at least one
// molecule (parts' = min(parts, B) >= 1), also when one molecule owns most of the rows.
std::vector<int> mcg_plan_range_cuts(int B, const int32_t* n_nodes_host, int parts);
