// daco_scan16.hip -- the TSP instantiations of scan16_kernel (daco_scan16_kernel.h has the kernel and its description; the CVRP
// ones are daco_scan16_cvrp.hip and, with float64 load bookkeeping, daco_scan16_cvrp64.hip: three translation units build in parallel)
#include "daco_scan16_kernel.h"

namespace daco {

// entry used by daco_tsp_sample in DACO_SCAN mode for n <= DACO_SCAN16_MAX_N (under DACO_SCAN_LAYOUT=16: 512)
hipError_t launch_tsp_scan16(const SampleParams &sp, bool logp, hipStream_t s) { return launch_by_lanes<false, false>(sp, logp, s); }

}  // namespace daco
