// Definition of the opaque mrcal_amd_problem_t
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "layout.hpp"
#include "problem.hpp"
#include "kernels.hpp"
#include "solver_kernels.hpp"

// One operating point of the dog-leg iteration: the state, the cost function
// there (x, J) and its normal equations. The solver flips between two of these
// (libdogleg's beforeStep / afterStep)
struct mrcal_amd_oppoint
{
    double* b  = NULL;          // packed state [Nstate]
    double* x  = NULL;          // residuals    [Nmeas]
    double* Jv = NULL;          // CSR values   [Nnz]
    mrcal_amd::NormalBuffers N = {};
    double* step_cauchy = NULL; // [Nstate]
    double* step_gn     = NULL; // [Nstate]

    // host mirrors
    double norm2_x = 0, cauchy_lensq = 0, gn_lensq = 0;
    bool   have_normal = false, cauchy_valid = false, gn_valid = false, did_step_to_edge = false;
};

struct mrcal_amd_solver_stats
{
    int    Niterations = 0, Nevaluations = 0, Nfactorizations = 0, Noutlier_passes = 0;
    double norm2_x = -1.0, lambda = 0.0;
    double seconds_solve = 0.0;
};

struct mrcal_amd_problem
{
    mrcal_amd::Layout        L;          // state layout global, measurement layout local to the shard
    mrcal_amd::DeviceProblem D;
    mrcal_amd::NormalDims    nd;
    mrcal_amd::BlockRanges   br = {0,0,0,0};   // the E blocks this shard owns
    bool                     is_leader = true;
    int64_t                  Nnz        = 0;
    int64_t                  board_alg_bytes = 0; // algorithmic HBM bytes of one board-kernel launch
    int                      lds_bytes  = 0;
    std::vector<int>         board_sel;  // global index of each local board observation
    std::vector<double>      b_host;     // the seed, packed

    hipStream_t stream = NULL;
    hipEvent_t  ev_j0  = NULL, ev_j1 = NULL;
    bool        have_jacobian_timing = false;
    // optional: an event pair per Jacobian-kernel launch, to average over a timed region
    std::vector<hipEvent_t> ev_pool;
    int         ev_pool_used = 0;
    bool        ev_pool_enabled = false;

    // inputs
    double* d_seed_intrinsics   = NULL;
    double* d_seed_rt_cam_ref   = NULL;
    double* d_seed_rt_ref_frame = NULL;
    double* d_seed_points       = NULL;
    mrcal_amd::BoardObsMeta* d_board_meta = NULL;
    double*                  d_board_pool = NULL;
    mrcal_amd::PointObsMeta* d_point_meta = NULL;
    double*                  d_point_pool = NULL;
    int*                     d_imagersizes = NULL;

    // shared between the operating points
    double*  d_joint = NULL;
    double*  d_gram  = NULL;
    int32_t* d_Jp    = NULL;
    int32_t* d_Ji    = NULL;

    mrcal_amd_oppoint op[2];
    int icur = 0;                        // which operating point the accessors/evaluate() address

    // solver
    bool                       solver_ready = false;
    mrcal_amd::AssemblyPlan    plan = {};
    mrcal_amd::FactorBuffers   F    = {};
    double*                    d_step   = NULL;   // [Nstate]
    int*                       d_counts = NULL;   // [4]
    double*                    h_scalars = NULL;  // pinned [64]
    mrcal_amd_solver_stats     stats;

    mrcal_amd::EvalBuffers eval_buffers(int i, bool with_gram) const
    {
        mrcal_amd::EvalBuffers B;
        B.b = op[i].b; B.joint = d_joint; B.x = op[i].x; B.Jv = op[i].Jv;
        B.Jp = d_Jp; B.Ji = d_Ji; B.gram = with_gram ? d_gram : NULL;
        return B;
    }

    mrcal_amd_problem() { memset(&D, 0, sizeof(D)); memset(&L, 0, sizeof(L)); memset(&nd, 0, sizeof(nd)); }
    ~mrcal_amd_problem();
};

namespace mrcal_amd {
// allocates the second operating point and all solver scratch; idempotent
bool problem_prepare_solver(mrcal_amd_problem* P);
// x, J (and the normal equations if with_normal) at op[i].b
bool problem_evaluate_op(mrcal_amd_problem* P, int i, bool with_jacobian, bool with_normal);
}
