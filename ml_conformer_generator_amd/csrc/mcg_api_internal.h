// Internal glue between the translation units of libmlconfgen_hip.so.
#pragma once
#include "../../include/mlconfgen_hip.h"

// accessors of the opaque plan (defined in mcg_egnn_plan.hip)
int mcg_plan_B(const mcg_plan* p);
int mcg_plan_N(const mcg_plan* p);
const int* mcg_plan_n_nodes(const mcg_plan* p);   // device pointer
// "everything enqueued on this plan so far" -> the plan's completion event, recorded on the caller's stream
// (mcg_plan_destroy waits for it before the plan's blocks return to the pool)
void mcg_plan_mark_done(const mcg_plan* p, void* stream);
