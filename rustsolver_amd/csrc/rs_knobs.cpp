// rs_knobs.cpp -- the ONE place where the library reads switches from the environment.
//
// A caller chooses kernel forms through the API (rs_kernel_forms in rs_solver_params, rs_table_params, rs_deal_trainer_params.prefetch).  The sixteen variables below
// exist for the test-suite and the profiling tools: each forces a form that the engine otherwise picks by batch or table size (so that small test inputs meet the forms big
// inputs get), or switches a facility off that has a fallback (staged rows, helper processes, launch overlap); they are resolved once per solver / table / trainer, at its
// creation.  Round 4 removed twenty-four more: forms measured as losers went with their code (the whole-deal-loop lane kernels, dense walks on delta rows, wide opponent
// records, the grouped compiles), tuning constants became constants, optimisation layers that have no regime left where they are off lost their switch.  DESIGN.md section 9.
#include <cstdlib>

#include "rs_internal.hpp"

namespace rs {

namespace {
enum Kind { FLAG /* present = on, whatever it holds */, INT /* atoi */, BOOL /* atoi != 0 */, LONG };
struct Entry {
    const char *name;
    Kind kind;
    int Knobs::*field;
    long Knobs::*lfield;
};
const Entry kEntries[] = {
    {"RS_JIT_LANES", INT, &Knobs::lanes, nullptr},
    {"RS_JIT_ORDERED", INT, &Knobs::ordered, nullptr},
    {"RS_JIT_ROWS", INT, &Knobs::rows, nullptr},
    {"RS_JIT_ROWS_CHUNK", INT, &Knobs::rows_chunk, nullptr},
    {"RS_JIT_DUMP", FLAG, &Knobs::dump, nullptr},
    {"RS_JIT_SCAN_ALL", INT, &Knobs::scan_all, nullptr},
    {"RS_JIT_NO_SIBLINGS", INT, &Knobs::no_siblings, nullptr},
    {"RS_JIT_NO_STAGE", FLAG, &Knobs::no_stage, nullptr},
    {"RS_JIT_DIRECT_ROWS", INT, &Knobs::direct_rows, nullptr},
    {"RS_BR_DEPTH_FIRST", FLAG, &Knobs::br_depth_first, nullptr},
    {"RS_JIT_NO_PROCS", FLAG, &Knobs::jit_no_procs, nullptr},
    {"RS_JIT_LDS_MAX", INT, &Knobs::lds_max, nullptr},
    {"RS_JIT_MAX_BLOCKS", INT, &Knobs::max_blocks, nullptr},
    {"RS_JIT_NO_OVERLAP", FLAG, &Knobs::no_overlap, nullptr},
    {"RS_JIT_NO_MERGE", FLAG, &Knobs::no_merge, nullptr},
    {"RS_TABLE_TILE_LANES", LONG, nullptr, &Knobs::tile_lanes},
};
}  // namespace

Knobs knobs_resolve(const rs_kernel_forms *forms) {
    Knobs k;
    if (forms) {   // the caller's API first
        if (forms->lane_fan == RS_FAN_NONE || forms->lane_fan == RS_FAN_EXPAND) k.fan = forms->lane_fan - 1;
        if (forms->deals_per_thread == 1 || forms->deals_per_thread == 2 || forms->deals_per_thread == 4) k.lanes = forms->deals_per_thread;
        if (forms->shadow == RS_SHADOW_ALL) k.shadow_all = 1;
        if (forms->deal_order == RS_FORM_ON) k.ordered = 1;
        else if (forms->deal_order == RS_FORM_OFF) k.ordered = 0;
        if (forms->delta_rows == RS_FORM_ON) k.rows = 1;
        else if (forms->delta_rows == RS_FORM_OFF) k.rows = 0;
        if (forms->direct_rows == RS_FORM_ON) k.direct_rows = 1;
        else if (forms->direct_rows == RS_FORM_OFF) k.direct_rows = 0;
        if (forms->kept_records == RS_FORM_OFF) k.no_kept = 1;
    }
    for (const Entry &e : kEntries) {   // then the test-only overrides
        const char *v = getenv(e.name);
        if (!v) continue;
        switch (e.kind) {
        case FLAG: k.*(e.field) = 1; break;
        case INT: k.*(e.field) = atoi(v); break;
        case BOOL: k.*(e.field) = atoi(v) != 0 ? 1 : 0; break;
        case LONG: k.*(e.lfield) = atol(v); break;
        }
    }
    return k;
}

// Test-only: the collective library rs_comm.cpp loads instead of librccl.so.  tests/stub_rccl.c implements the same five entry points over shared memory for several
// PROCESSES ON ONE GPU (RCCL refuses two ranks on one device), so that the multi-rank paths run with more than one rank on a one-card box.
const char *rccl_library_override() {
    const char *e = getenv("RS_RCCL_LIB");
    return (e && *e) ? e : nullptr;
}

std::string jit_cache_dir() {
    if (const char *e = getenv("RS_JIT_CACHE")) return *e ? std::string(e) : std::string();
    const char *home = getenv("HOME");
    if (!home || !*home) return std::string();
    return std::string(home) + "/.cache/rustsolver_amd";
}

}  // namespace rs
