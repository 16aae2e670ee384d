// What the evaluation's translation units share (kernels.hip, splined_kernels.hip): the state's accessors and the two
// launches the splined models' evaluation borrows from the parametric ones' unit (round 6: the splined kernels were cut
// out of kernels.hip, which had grown past 2500 lines)
#pragma once
#include <hip/hip_runtime.h>
#include "problem.hpp"
#include "kernels.hpp"

namespace mrcal_amd {

////////////////////////////////////////////////////////////////////////////////
// state access: packed state b[] (if the block is being optimized) or seeds
////////////////////////////////////////////////////////////////////////////////
// (the state: a pointer to the packed vector, or anything indexable like one - dogleg_choose.hpp's TrialState)
template<class BV>
__device__ __forceinline__
double get_intrinsic(const DeviceProblem& P, const BV& b, int icam, int i)
{
    if(i < P.Ncore)
    {
        if(P.Ncore_state)
            return b[P.i_state_intrinsics + icam*P.Nintr_state + i] *
                ((i < 2) ? SCALE_INTRINSICS_FOCAL_LENGTH : SCALE_INTRINSICS_CENTER_PIXEL);
        return P.seed_intrinsics[icam*P.Nintrinsics + i];
    }
    if(P.Ndist_state)
        return b[P.i_state_intrinsics + icam*P.Nintr_state + P.Ncore_state + (i - P.Ncore)] * SCALE_DISTORTION;
    return P.seed_intrinsics[icam*P.Nintrinsics + i];
}
template<class BV>
__device__ __forceinline__
void get_rt_cam_ref(double* rt, const DeviceProblem& P, const BV& b, int icam_extrinsics)
{
    if(P.do_optimize_extrinsics)
    {
        const int s = P.i_state_extrinsics + 6*icam_extrinsics;
        for(int i=0;i<3;i++) rt[i]   = b[s + i]   * SCALE_ROTATION_CAMERA;
        for(int i=0;i<3;i++) rt[3+i] = b[s + 3+i] * SCALE_TRANSLATION_CAMERA;
    }
    else
        for(int i=0;i<6;i++) rt[i] = P.seed_rt_cam_ref[6*icam_extrinsics + i];
}
template<class BV>
__device__ __forceinline__
void get_rt_ref_frame(double* rt, const DeviceProblem& P, const BV& b, int iframe)
{
    if(P.do_optimize_frames)
    {
        const int s = P.i_state_frames + 6*iframe;
        for(int i=0;i<3;i++) rt[i]   = b[s + i]   * SCALE_ROTATION_FRAME;
        for(int i=0;i<3;i++) rt[3+i] = b[s + 3+i] * SCALE_TRANSLATION_FRAME;
    }
    else
        for(int i=0;i<6;i++) rt[i] = P.seed_rt_ref_frame[6*iframe + i];
}
template<class BV>
__device__ __forceinline__
void get_warp(double* w, const DeviceProblem& P, const BV& b)
{
    if(P.has_warp_state)
    {
        w[0] = b[P.i_state_warp+0] * SCALE_CALOBJECT_WARP;
        w[1] = b[P.i_state_warp+1] * SCALE_CALOBJECT_WARP;
    }
    else
    {
        w[0] = P.seed_warp[0];
        w[1] = P.seed_warp[1];
    }
}

// (1024 whatever there is to clear: every workgroup of the launch derives the dog-leg step's scalars first, and with 4096
//  of them BASELINE configuration 2's prologue took 33 us instead of 25)
#define PROLOGUE_ZERO_BLOCKS(total) (1024*64/PRO_T)
// kernels.hip
void launch_triangulated(const DeviceProblem& P, const EvalBuffers& B, bool with_jacobian, hipStream_t stream);
void launch_prologue(const DeviceProblem& P, const EvalBuffers& B, int nblocks_obs, int nblocks_unpack, int nblocks_zero,
                     int nblocks_reg, bool with_jacobian, hipStream_t stream);
// splined_kernels.hip
void launch_eval_splined(const DeviceProblem& P, const EvalBuffers& B, bool with_jacobian,
                         hipStream_t stream, hipEvent_t ev_j0, hipEvent_t ev_j1, int parts);
void launch_structure_splined_regularization(const DeviceProblem& P, const EvalBuffers& B, int Nreg, hipStream_t stream);

} // namespace mrcal_amd
