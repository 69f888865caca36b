// daco_scan16_cvrp64.hip -- the CVRP instantiations of scan16_kernel (daco_scan16_kernel.h) with float64 load bookkeeping
// (cvrp_nls/aco.py:254-272)
#include "daco_scan16_kernel.h"

namespace daco {

hipError_t launch_cvrp_scan16_f64(const SampleParams &sp, bool logp, hipStream_t s) { return launch_by_lanes<true, true>(sp, logp, s); }

}  // namespace daco
