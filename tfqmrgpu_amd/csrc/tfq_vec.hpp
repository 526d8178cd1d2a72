// internal launch interface of the vector / column kernels (tfq_vec.hip) and the multiply (tfq_spmm.hip)
#pragma once
#include "tfq_device.hpp"

namespace tfq {

enum {
    VEC_SETUP = 0,     // clear vectors, v5 := B, tau := |b|^2, rho := 1, pz <- v3.v5
    VEC_DEC35,         // beta, rho from pz
    VEC_XPAY_V6,       // v6 := v5 + beta v6
    VEC_DEC34,         // alfa, c67 from pz
    VEC_V5_NRM,        // v5 += alfa v9 ; pd <- |v5|^2
    VEC_DECT_C67,      // tau, var, eta, c67 from pd
    VEC_X_V6_V7,       // [x += eta2 v7] ; v7 := v6 + c67a v7 ; x += eta v7 ; v6 += alfa v4 ; v7 := v6 + c67 v7
    VEC_DECT_FIN,      // tau, var, eta from pd + per-column stopping record
    VEC_X_FLUSH,       // x += eta2 v7, only in front of a residual probe
    VEC_PROBE_COL      // per-column residual record from pd
};

hipError_t vec_launch(int op, DevPlan const& d, double tol, int maxIt, hipStream_t s);   // the runtime call of VEC_SETUP may fail

// Y = A*X over the plan's pair list with a fused epilogue (EPI_* in tfq_device.hpp)
void spmm_launch(int epi, DevPlan const& d, hipStream_t s);

// the kernel family spmm_launch runs for this plan (tfqmrgpuExt_getMultiplyKernel)
char const* spmm_kernel_family(DevPlan const& d);

// Y = A*X on two X-shaped vectors of the plan, no epilogue (tfqmrgpuExt_applyOperator)
void spmm_apply(DevPlan const& d, void const* X, void* Y, hipStream_t s);

// the epilogue of spmm_launch alone, for a product Y that a user-defined operator has already written:
// Yext holds the blocks in the caller's order, i2u[internal block] = caller's block
void epilogue_launch(int epi, DevPlan const& d, void const* Yext, uint32_t const* i2u, hipStream_t s);

} // namespace tfq
