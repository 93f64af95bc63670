// Library-level entry points of libprotstruc_hip.so (see include/protstruc_hip.h).
// The library keeps no mutable state: there is nothing to initialise, configure or tear down.
#include <hip/hip_runtime.h>

#include "../../include/protstruc_hip.h"

// 2: per-call K1 configuration (ps_k1_config, ps_pairwise_distance_cfg_f32) replaced the process-global
//    ps_set_tuning / ps_get_tuning of version 1.
// 3: ps_k1_plan_f32 (which kernel a K1 launch takes; host-only query); ps_inter_residue_geometry_f32 takes exact_sqrt.
// 4: ps_pairwise_angles_f32 and ps_inter_residue_geometry_f32 take exact_angles (0: fast arithmetic, 1: the reference's
//    order of operations).
extern "C" int ps_abi_version(void) { return PS_ABI_VERSION; }

extern "C" const char* ps_error_string(int code) { return hipGetErrorString(static_cast<hipError_t>(code)); }

// 1 when the library was built with -DPS_EXPERIMENTS (timing experiments that can write wrong values); the product
// build returns 0 and refuses any ps_k1_config with experiment != 0.
extern "C" int ps_has_experiments(void) {
#ifdef PS_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}
