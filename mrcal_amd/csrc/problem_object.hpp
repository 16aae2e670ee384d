// Definition of the opaque mrcal_amd_problem_t
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "layout.hpp"
#include "problem.hpp"
#include "kernels.hpp"

struct mrcal_amd_problem
{
    mrcal_amd::Layout        L;          // state layout global, measurement layout local to the shard
    mrcal_amd::DeviceProblem D;
    mrcal_amd::EvalBuffers   B;
    int64_t                  Nnz        = 0;
    int                      lds_bytes  = 0;
    std::vector<int>         board_sel;  // global index of each local board observation
    std::vector<double>      b_host;     // the seed, packed

    hipStream_t stream = NULL;
    hipEvent_t  ev_j0  = NULL, ev_j1 = NULL;
    bool        have_jacobian_timing = false;

    double* d_seed_intrinsics   = NULL;
    double* d_seed_rt_cam_ref   = NULL;
    double* d_seed_rt_ref_frame = NULL;
    double* d_seed_points       = NULL;
    mrcal_amd::BoardObsMeta* d_board_meta = NULL;
    double*                  d_board_pool = NULL;
    mrcal_amd::PointObsMeta* d_point_meta = NULL;
    double*                  d_point_pool = NULL;
    int*                     d_imagersizes = NULL;

    mrcal_amd_problem() { memset(&B, 0, sizeof(B)); memset(&D, 0, sizeof(D)); memset(&L, 0, sizeof(L)); }
    ~mrcal_amd_problem();
};
