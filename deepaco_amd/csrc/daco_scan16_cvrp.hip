// daco_scan16_cvrp.hip -- the CVRP instantiations of scan16_kernel (daco_scan16_kernel.h), float32 load bookkeeping
#include "daco_scan16_kernel.h"

namespace daco {

hipError_t launch_cvrp_scan16_f64(const SampleParams &sp, bool logp, hipStream_t s);    // daco_scan16_cvrp64.hip

// entry used by daco_cvrp_sample in DACO_SCAN mode for n <= DACO_SCAN32_MAX_N; float64 demands (cvrp_nls/): the F64 variants
hipError_t launch_cvrp_scan16(const SampleParams &sp, bool logp, hipStream_t s) {
  if (sp.demand64) return launch_cvrp_scan16_f64(sp, logp, s);
  return launch_by_lanes<true, false>(sp, logp, s);
}

}  // namespace daco
