// Host-visible interface of solver_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "problem.hpp"
#include "kernels.hpp"

namespace mrcal_amd {

// partition of the state into the dense shared block S and the block-diagonal
// eliminated set E (see solver_kernels.hip)
struct NormalDims
{
    int Nstate;
    int Nie;          // intrinsics + extrinsics state variables: S indices [0,Nie) == state indices
    int Nwarp;        // 0 or 2: S indices [Nie, Nie+Nwarp)
    int i_state_warp;
    int Nc;           // Nie + Nwarp
    int NE;           // frame + point state variables; E index e == state index Nie + e
    int Nfb;          // frame blocks (6x6)
    int Npb;          // point blocks (3x3)
    int NEb;          // Nfb + Npb
};

// The E blocks a shard owns: a contiguous range of frame blocks plus (on the
// shard leader) all the point blocks
struct BlockRanges
{
    int frame_lo, frame_hi;   // frame blocks [lo,hi)
    int point_lo, point_hi;   // point blocks, as block indices (>= Nfb)
    __host__ __device__ int count() const { return (frame_hi - frame_lo) + (point_hi - point_lo); }
    __host__ __device__ int block(int i) const
    {
        const int nf = frame_hi - frame_lo;
        return (i < nf) ? frame_lo + i : point_lo + (i - nf);
    }
    // E-row range of part 0 (frames) / 1 (points)
    __host__ __device__ void e_range(const NormalDims& nd, int part, int* lo, int* hi) const
    {
        if(part == 0) { *lo = 6*frame_lo; *hi = 6*frame_hi; }
        else          { *lo = 6*nd.Nfb + 3*(point_lo - nd.Nfb); *hi = 6*nd.Nfb + 3*(point_hi - nd.Nfb); }
    }
};

enum { SC_NORM2_X = 0, SC_NORM2_G, SC_GNG, SC_TMP0, SC_TMP1, SC_TMP2, SC_TMP3, NSCALARS = 8 };

// the normal equations of one operating point, unfactored
struct NormalBuffers
{
    double* A;        // [Nc][Nc]
    double* Bt;       // [NE][Nc]
    double* D;        // [NEb][6][6]
    double* g;        // [Nstate]   Jt x, state order
    double* scalars;  // [NSCALARS]
};

// factorization scratch, one set
struct FactorBuffers
{
    double* Wt;       // [NE][Nc]   L_e^-1 Bt_e
    double* LD;       // [NEb][6][6] Cholesky factors of the D blocks
    double* y;        // [NE]       L_e^-1 g_e
    double* S;        // [Nc][Nc]   Schur complement -> its Cholesky factor (lower)
    double* r;        // [Nc]       reduced rhs -> d_s
    int*    status;   // [1] nonzero: not positive definite
};

// iteration-invariant work lists for the assembly
struct AssemblyPlan
{
    int* frame_obs_begin;  // [Nframes+1]
    int* chunk_begin;      // [Nchunks+1]
    int* pair_obs;         // [Nobs_board]
    int  Nchunks;
};

hipError_t launch_assemble(const DeviceProblem& P, const NormalDims& nd, const AssemblyPlan& plan,
                           const EvalBuffers& B, const NormalBuffers& N, hipStream_t stream);
hipError_t launch_factor_local(const NormalDims& nd, const BlockRanges& br,
                               const NormalBuffers& N, const FactorBuffers& F,
                               double lambda, bool is_leader, hipStream_t stream);
hipError_t launch_solve_backsub(const NormalDims& nd, const BlockRanges& br,
                                const FactorBuffers& F, double* step_gn, hipStream_t stream);
hipError_t launch_factor_and_solve(const NormalDims& nd, const BlockRanges& br,
                                   const NormalBuffers& N, const FactorBuffers& F,
                                   double lambda, double* step_gn, hipStream_t stream);
hipError_t launch_quadform(const NormalDims& nd, const NormalBuffers& N, const double* v, double* out,
                           hipStream_t stream);
hipError_t launch_dot(int n, const double* a, const double* b, double* out, hipStream_t stream);
hipError_t launch_axpby(int n, double alpha, const double* a, double beta, const double* b, double* y,
                        hipStream_t stream);
hipError_t launch_outlier_stats(int Npoints_board, double thresh_sq, const double* x, const double* pool,
                                int* counts, double* sums, hipStream_t stream);
hipError_t launch_mark_outliers(int Npoints_board, double thresh_sq, const double* x, double* pool,
                                int* counts, hipStream_t stream);

} // namespace mrcal_amd
