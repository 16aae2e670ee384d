// Kernels that are launched from another translation unit than the one that defines them, and the host entry points
// between the units (round 6: the cut of solver_kernels.hip). A launch needs the kernel's declaration only: the code
// object it lives in is its defining unit's
#pragma once
#include "solver_device.hpp"

namespace mrcal_amd {

__global__ void assemble_splined_kernel(DeviceProblem P, NormalDims nd, OpRef R, AssemblyPlan plan,
                             const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji,
                             int npairs_extra, int pairs_row0, int compact_extra);
__global__ void assemble_splined_gather_kernel(DeviceProblem P, NormalDims nd, OpRef R, AssemblyPlan plan, int nwaves, int block0, int window);
__global__ void assemble_splined_gather_knots_kernel(DeviceProblem P, NormalDims nd, OpRef R, AssemblyPlan plan);
__global__ void rows_pairs_kernel(NormalDims nd, OpRef R, int row0, int row1,
                       const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji, double* __restrict__ row_part);
__global__ void spl_compact_kernel(DeviceProblem P, NormalDims nd, OpRef R, const int* __restrict__ nd_lim);
__global__ void assemble_splined_combine_kernel(DeviceProblem P, NormalDims nd, OpRef R, AssemblyPlan plan, int nrow_parts);
__global__ void assemble_finalize_kernel(int npos, NormalDims nd, const OpDev* __restrict__ ops, const int* __restrict__ sel,
                              const int* __restrict__ skip, AssemblyPlan plan);
__global__ void assemble_factor_kernel(DeviceProblem P, NormalDims nd, BlockRanges br, const OpDev* __restrict__ ops,
                            const int* __restrict__ sel_eval, const int* __restrict__ sel_cur,
                            const SolverCtl* __restrict__ ctl, const int* __restrict__ skip,
                            const int* __restrict__ mode_ptr, int mode_host,
                            int do_factor, double lambda_host,
                            AssemblyPlan plan, const double* __restrict__ gram, FactorBuffers F,
                            int nframe_blocks, int row0, int row1,
                            const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji, int ngen /* the launch's last workgroups: the planned rows */);
__global__ void eblock_factor_kernel(NormalDims nd, BlockRanges br, int first, OpRef R, double lambda_host, const SolverCtl* ctl,
                          double* __restrict__ Wt, double* __restrict__ LD, double* __restrict__ y,
                          int* __restrict__ status, unsigned* __restrict__ occ, int nocc, double* __restrict__ Wtile);
__global__ void step2_reduce_kernel(NormalDims nd, const OpDev* __restrict__ ops, const SolverCtl* __restrict__ ctl,
                         const SolverCtlFlags* __restrict__ fl, int is_leader, int nred,
                         int nslots, const double* __restrict__ Spart,
                         double* __restrict__ S, double* __restrict__ r, const int* __restrict__ status,
                         const unsigned char* __restrict__ live, int* __restrict__ cperm_cur, double* __restrict__ iso,
                         int* __restrict__ err /* SolverCtl::error */,
                         double* __restrict__ ndMA, double* __restrict__ ndMB, int* __restrict__ ndp_cur, int nfill /* workgroups behind the last */,
                         int ride_finish, Step2Dev sd, double* __restrict__ Spk /* a packed copy of S's lower triangle, or NULL */);
__global__ void backsub_kernel(NormalDims nd, BlockRanges br, OpRef R, const int* __restrict__ skip_also,
                    const double* __restrict__ Wt, const double* __restrict__ LD,
                    const double* __restrict__ y, const double* __restrict__ ds);
__global__ void rows_generic_kernel(NormalDims nd, OpRef R, int row0, int row1,
                         const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji);
__global__ void zero_normal_kernel(NormalDims nd, OpRef R);

// Round 5, the splined models: the matrix that was factored may be the camera block WITHOUT its isolated variables (the
// control points no board covers: all they have is their regularization, a 2 x 2 block a control point, coupled to
// nothing: LcholCompact below). Then column c of the factored matrix is camera-block variable perm[c], the solution goes
// to dout[perm[c]], and the workgroups past the columns' solve the 2 x 2 blocks: d = -S2^-1 r2 by the closed form of a
// 2 x 2 Cholesky (not positive definite: status, like a pivot of the big matrix)
struct LcholCompact
{
    const int*    cperm;     // [Nc] position -> camera-block variable | [Nc] variable -> position | [1] n = the coupled ones (they come first); NULL: none of this
    const double* iso;       // [Nc/2][4]: per isolated pair (positions n + 2 q, n + 2 q + 1) s00, s10, s11, then [Nc] their rhs behind all blocks
    double*       dout;      // [Nc] the solution in the camera block's own order
    int           Nc;
    // the dissection (lchol_nd_*; NULL: none): the plan of the point that was reduced (FactorBuffers::ndp_cur). Where it is
    // active the matrix factored here is the SEPARATOR's: its column c is position nA + nB + c of the plan's map
    const int*    ndh;
    // ... and then every workgroup, with its 16 entries of d_S at hand, leaves the sides' share of them behind:
    // ndpart[column block][position i of A | B] = sum over its 16 columns s of L_SX[s][i] d_S[s]  (lchol_nd_apply_kernel adds the
    // blocks' shares in block order: w = z + L_SX^T d_S)
    const double* ndMA; const double* ndMB;
    double*       ndpart;    // [ceil(Nc/16)][2 LCH_ND_WMAX]
};
// (round 6) [2]: the smallest and the largest diagonal entry of L over the factorizations of a solve, as the bit patterns
// of positive doubles (which order like integers): FactorBuffers::diag_minmax. NULL: not tracked
typedef unsigned long long* LcholDiagSpread;

struct LcholChain
{
    double*    M;        // [(nx + ns + 1)][nx + ns]
    double*    Linv;     // [own][64][64] | Yb [64 own][64 own] | zc [64 own],  own = nx/64
    const int* nx_dev;   // the chain's own columns (a multiple of 64)
    const int* ns_dev;   // its border
};

struct LcholNdLaunch { LcholChain A, B; const int* ndh; NdLimits lim; };

hipError_t launch_cholesky_large(int n, const int* skip, double* M, double* Linv, int* status, hipStream_t stream,
                                 const Step2Dev* sd = NULL, bool* fused = NULL, const int* n_dev = NULL, const LcholCompact* compact = NULL,
                                 int likely_panels = 0 /* with n_dev: launches 0 .. likely_panels one by one, the rest in lchol_tail_kernel; 0: all one by one */,
                                 unsigned* tail_counter = NULL, const LcholNdLaunch* nds = NULL,
                                 bool finish_done = false /* with sd: the end-of-trial logic has run already (the first launch goes by `skip`); the verdict still rides in the last */,
                                 bool sweep = false /* the solve by the backward sweep in groups of panels (rounds 2-3; backward stable) instead of through
                                                       L^-1 built on the side (lchol_inverse_block): FactorBuffers::use_sweep */,
                                 LcholDiagSpread diag_minmax = NULL);
// ---- between assembly.hip / schur.hip and the trial step's launchers (step.hip)
// a pair chunk is reduced by one workgroup per 256 Gram positions
__host__ __device__ __forceinline__ int assemble_chunk_slices(const DeviceProblem& P) { return (gram_stride(P.Ndist) + 255) >> 8; }
inline size_t assemble_lds_bytes(const NormalDims& nd) { return (size_t)(6*nd.Nc + 42 + 48)*sizeof(double); }
// the first of the rows the assembly takes one lane each: with a plan for the rows that share destinations
// (GenPlan) only the regularization rows are left, whose destinations are their own
inline int assemble_row0(const DeviceProblem& P, const AssemblyPlan& plan)
{
    return (plan.gen.Nrows > 0) ? P.i_meas_regularization : 2*P.W*P.H*P.Nobs_board;
}
inline int assemble_row_blocks(const DeviceProblem& P, const AssemblyPlan& plan)
{
    const int row0 = assemble_row0(P, plan);
    return (P.Nmeas > row0) ? (P.Nmeas - row0 + 255)/256 : 0;
}
#ifndef SYRK_STRIP_FROM
#define SYRK_STRIP_FROM 256      // camera blocks wider than this: the strip SYRK kernels (one A operand for four B operands), the tile occupancy of Wt
#endif
#define SRED_SPLIT 4      // threads sharing one output element of the reduction (adjacent lanes): schur_reduce_body
hipError_t launch_gen_rows(const NormalDims& nd, const AssemblyPlan& plan, const OpRef& R, const int32_t* Jp, hipStream_t stream);
// (round 6) the planned rows as the last workgroups of assemble_factor_kernel's launch (assembly.hip)
int    gen_ride_blocks(const AssemblyPlan& plan);
size_t assemble_lds_bytes_with_gen(const NormalDims& nd, const AssemblyPlan& plan);
hipError_t launch_gen_finalize(const NormalDims& nd, const AssemblyPlan& plan, const OpRef& R, hipStream_t stream);
int launch_syrk(const NormalDims& nd, const BlockRanges& br, const int* skip, const FactorBuffers& F,
                const FinalizeRide* ride, hipStream_t stream, const unsigned char** live /* out: the slots' flags, or NULL */);
hipError_t launch_nd_plans_off(const OpDev* ops, int Nc, hipStream_t stream);
// the one-workgroup LDS Cholesky (cholesky_lds.hip), finish = the kernel's FINISH (0: a factorization and solve and nothing
// else; 1: the end of the trial in front, the verdict behind; 2: the verdict behind alone); and its fallback in global memory
hipError_t launch_cholesky_lds(int finish, int n, const int* skip, int keep_factor, double* S, double* r, int* status,
                               const Step2Dev& sd, hipStream_t stream);
hipError_t launch_cholesky_global(int n, const int* skip, double* S, double* r, int* status, hipStream_t stream);
// (round 6) finish = 2 with the quadratic form's workgroups in the launch (cholesky_lds.hip step2_chol_quadform_kernel)
hipError_t launch_cholesky_lds_quadform(int n, const NormalDims& nd, const FactorBuffers& F, const Step2Dev& sd,
                                        double* qf_part, int nqf, hipStream_t stream);
// where the reduction leaves the packed copy of S's lower triangle for it (behind [S | r | g_S | 2 | 64] in F.S's allocation)
inline double* factor_S_packed(const FactorBuffers& F, int Nc) { return F.S + (size_t)Nc*Nc + 2*(size_t)Nc + 2 + 64; }
// (problem.cpp allocates Nc (Nc + 1)/2 + Nc + 2 doubles there: the triangle and, as its row Nc, the right-hand side)

} // namespace mrcal_amd
