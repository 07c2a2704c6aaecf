// The stationary kernels as functions of the scaled squared distance -- ONE definition shared by the assembly
// (kmat.hip) and the refinement's residual pass (refine.hip), which must reproduce the assembled entries bit for bit.
#pragma once
#include "gpn_common.h"

namespace gpn {

template <int KIND>
__device__ __forceinline__ double kernel_of_r2(double r2, double var) {
  if constexpr (KIND == GPN_SQDIST) {
    return r2;                                         // util.py:73-88 (lengthscale-scaled)
  } else if constexpr (KIND == GPN_RBF) {
    return var * exp(-0.5 * r2);                       // kernels.py:220-222
  } else {
    const double r = sqrt(fmax(r2, 1e-40));            // kernels.py:172
    if constexpr (KIND == GPN_MATERN52) {
      const double s5 = 2.23606797749978969641;        // sqrt(5), kernels.py:204-212
      return var * (1.0 + s5 * r + 5.0 / 3.0 * r * r) * exp(-s5 * r);
    } else if constexpr (KIND == GPN_MATERN32) {
      const double r3 = 1.73205080756887729353 * r;    // kernels.py:196-201
      return var * (1.0 + r3) * exp(-r3);
    } else if constexpr (KIND == GPN_PERIODIC) {
      return var * cos(r);                             // kernels.py:228-235
    } else {
      return var * exp(-r);                            // kernels.py:189-190
    }
  }
}

}  // namespace gpn
