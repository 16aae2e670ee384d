// Host-visible launch interface of kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "problem.hpp"

namespace mrcal_amd {


// device pointers of one operating point
struct EvalBuffers
{
    double*  b;      // packed state             [Nstate]
    double*  joint;  // prologue scratch         [Nobs_board][JOINT_STRIDE]
    double*  x;      // residuals                [Nmeas]
    double*  Jv;     // CSR values               [Nnz]
    int32_t* Jp;     // CSR rowptr               [Nmeas+1]
    int32_t* Ji;     // CSR colidx               [Nnz]
    double*  gram;   // per-observation Gram     [Nobs_board][gram_stride(Ndist)]; NULL: don't form it
};

bool lens_supported(int lens_type);

// x (and J values if with_jacobian) at B.b. ev_j0/ev_j1, if given, bracket the
// board Jacobian kernel on the stream
hipError_t launch_evaluate(const DeviceProblem& P, const EvalBuffers& B, bool with_jacobian,
                           int lds_bytes, hipStream_t stream,
                           hipEvent_t ev_j0, hipEvent_t ev_j1);

// rowptr/colidx; iteration-invariant
hipError_t launch_structure(const DeviceProblem& P, const EvalBuffers& B, hipStream_t stream);

} // namespace mrcal_amd
