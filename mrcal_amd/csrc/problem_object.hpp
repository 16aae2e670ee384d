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
// (libdogleg's beforeStep / afterStep). Host copy of the device pointers; the
// device has the same in mrcal_amd_problem::d_ops
struct mrcal_amd_oppoint : public mrcal_amd::OpDev
{
    mrcal_amd_oppoint() { memset(static_cast<mrcal_amd::OpDev*>(this), 0, sizeof(mrcal_amd::OpDev)); }
    bool have_normal = false;
};

struct mrcal_amd_solver_stats
{
    int    Niterations = 0, Nevaluations = 0, Nfactorizations = 0, Noutlier_passes = 0;
    int    Noutliers_triangulated = 0;
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
    // a second stream for work of a step that nothing on the first one waits for at once (the gather of the
    // splined assembly runs beside the block elimination and the SYRK): fork / join events
    hipStream_t side_stream = NULL;
    hipEvent_t  ev_fork = NULL, ev_join = NULL;
    bool        have_jacobian_timing = false;
    // (round 6) the solve without the Jacobian stream (include/mrcal_amd.h, mrcal_amd_problem_set_jacobian_stream):
    // solve_stores_jacobian = what the solver's own evaluations do; jfree_now = a solver entry point is queueing
    // evaluations in that mode right now; jacobian_stale = op[icur].Jv is NOT the Jacobian at op[icur].b: whoever
    // hands J out (get_J, dev_J_values) evaluates first (problem_ensure_jacobian)
    int         last_ctl_error = 0;         // SolverCtl::error of the last run_dogleg()
    double      lchol_diag_ratio = 1.0;     // min / max of the diagonal of the big Cholesky's factors over the last pass
    bool        sweep_fallback_wanted = false;
    bool        solve_stores_jacobian = true;
    bool        jfree_now             = false;
    bool        jacobian_stale        = false;
    bool        capturing = false;      // a hipGraph capture is in progress on the stream
    // optional: an event pair per Jacobian-kernel launch, to average over a timed region
    std::vector<hipEvent_t> ev_pool;
    int         ev_pool_used = 0;
    int         ev_pool_seen = 0, ev_pool_stride = 1;      // launches since _begin(); every stride-th one is timed
    bool        ev_pool_enabled = false;
    int*                cperm_cur_alloc = NULL;    // what F.cperm_cur points at while the compaction is on (a communicator turns it off)

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
    mrcal_amd::TriPairMeta*  d_tri_meta    = NULL;
    double*                  d_tri_px      = NULL;
    int*                     d_tri_outlier = NULL;
    // host copies, for the outlier logic (sequential, mrcal.c:3978-4402)
    std::vector<mrcal_amd::TriPairMeta> tri_meta_host;
    std::vector<double>      tri_px_host;
    std::vector<int>         tri_outlier_host;
    int                      tri_obs0 = 0;      // the shard's first triangulated observation in the caller's array

    // shared between the operating points
    double*  d_joint = NULL;
    double*  d_gram  = NULL;
    int32_t* d_Jp    = NULL;
    int32_t* d_Ji    = NULL;

    mrcal_amd_oppoint op[2];
    mrcal_amd::OpDev* d_ops = NULL;      // device copy of op[0..1]
    int icur = 0;                        // which operating point the accessors/evaluate() address

    // solver
    bool                       solver_ready = false;
    mrcal_amd::AssemblyPlan    plan = {};
    mrcal_amd::FactorBuffers   F    = {};
    double*                    d_step   = NULL;   // [Nstate]
    double*                    d_comm   = NULL;   // comm2 of the sharded step (4 doubles) + scratch for host-side sums
    struct mrcal_amd_comm*     comm     = NULL;   // attached communicator (not owned): solve/run_steps run sharded
    bool                       sharded_external = false;   // sharded step, the CALLER does the collectives
    bool                       verbose = false;            // mrcal_optimize(verbose): the iteration trace on stderr
    int*                       d_counts = NULL;   // [4]
    double*                    d_outlier_part = NULL;   // [outlier_partial_doubles()]
    double*                    h_scalars = NULL;  // pinned [64]
    // device-side dog-leg control (solver_kernels.hpp): the block, and a ring
    // of pinned host copies the host polls without stalling the queue
    mrcal_amd::SolverCtl*      d_ctl = NULL;
    mrcal_amd::SolverCtl*      h_ctl_ring = NULL;
    mrcal_amd::SolverCtl*      snap_target = NULL;      // where the step being queued leaves its control-block snapshot
    std::vector<hipEvent_t>    ctl_events;
    bool                       ctl_initialized = false;
    // one trial step captured as a graph: whole [0], or split around the board
    // kernel ([1] before, [2] after) when that kernel is being timed with events
    hipGraphExec_t             step_graph[3] = {NULL, NULL, NULL};
    mrcal_amd_solver_stats     stats;

    // op i, resolved by the host
    mrcal_amd::OpRef opref(int i) const
    {
        mrcal_amd::OpRef R = { d_ops + i, NULL, NULL };
        return R;
    }
    mrcal_amd::EvalBuffers eval_buffers(const mrcal_amd::OpRef& R, bool with_gram) const
    {
        mrcal_amd::EvalBuffers B;
        B.R = R; B.joint = d_joint; B.Jp = d_Jp; B.Ji = d_Ji; B.gram = with_gram ? d_gram : NULL;
        for(int i=0;i<5;i++) B.zero_n[i] = 0;
        B.zero_total = 0;
        B.choose = NULL;
        B.store_jacobian = true;
        return B;
    }
    mrcal_amd::EvalBuffers eval_buffers(int i, bool with_gram) const { return eval_buffers(opref(i), with_gram); }

    mrcal_amd_problem() { memset(&D, 0, sizeof(D)); memset(&L, 0, sizeof(L)); memset(&nd, 0, sizeof(nd)); }
    ~mrcal_amd_problem();
};

namespace mrcal_amd {
// allocates the second operating point and all solver scratch; idempotent
bool problem_prepare_solver(mrcal_amd_problem* P);
// x, J (and the normal equations if with_normal) at op[i].b
bool problem_evaluate_op(mrcal_amd_problem* P, int i, bool with_jacobian, bool with_normal);
// tears a problem nobody can reach any more down on a thread of its own (problem.cpp: ProblemReaper)
void problem_destroy_later(mrcal_amd_problem* P);
// makes op[icur].Jv the Jacobian at op[icur].b if a solve without the Jacobian stream left it otherwise
bool problem_ensure_jacobian(mrcal_amd_problem* P);
// the same, with the operating point possibly resolved on the device
bool problem_evaluate_ref(mrcal_amd_problem* P, const mrcal_amd::OpRef& R, bool with_jacobian, bool with_normal,
                          int parts = mrcal_amd::EVAL_PART_ALL, hipStream_t stream = NULL /* default: the problem's */,
                          const mrcal_amd::ChooseArgs* choose = NULL /* the prologue launch also chooses the trial point */);
// uploads op[0..1] to d_ops
bool problem_sync_ops(mrcal_amd_problem* P);
}
