// The CHOLMOD_factorization equivalent: factor JtJ of a CSR Jacobian on the
// GPU, solve against it.
//
// Reference behaviour: the mrcal.CHOLMOD_factorization Python type
// (mrcal-pywrap.c:111-214: cholmod_analyze + cholmod_factorize of Jt;
// :425-569 solve_xt_JtJ_bt(); :580-592 rcond()). CHOLMOD is a general sparse
// direct solver; this is the structured solver of assembly.hip / schur.hip / cholesky_*.hip given the
// partition of the state: a dense leading block S (intrinsics, extrinsics, and
// the trailing warp pair) and block-diagonal 6x6 / 3x3 blocks E (frames,
// points) that no row couples to each other. A matrix without such a structure
// is handled as all-S (dense), which is fine for small problems only.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include "layout.hpp"
#include "host_state.hpp"
#include "problem.hpp"
#include "kernels.hpp"
#include "solver_kernels.hpp"
#include "problem_object.hpp"
#include "../../include/mrcal_amd.h"

using namespace mrcal_amd;

#define HIP_TRY(expr, onfail)                                           \
    do {                                                                \
        hipError_t _e = (expr);                                         \
        if(_e != hipSuccess)                                            \
        {                                                               \
            set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            onfail;                                                     \
        }                                                               \
    } while(0)

struct mrcal_amd_factorization
{
    NormalDims    nd;
    BlockRanges   br;
    OpDev         op;         // host copy of the pointers
    OpDev*        d_op  = NULL;
    FactorBuffers F     = {};
    int32_t*      d_Jp  = NULL;
    int32_t*      d_Ji  = NULL;
    double*       d_rhs = NULL;
    double*       d_sol = NULL;
    double*       d_mm  = NULL;   // [2] min, max of the factor's diagonal
    double*       d_rhs_batch = NULL, *d_sol_batch = NULL;   // [batch_capacity][Nstate]: right-hand sides / solutions of a solve call
    double*       d_batch_scratch = NULL;                    // y, r, partial sums of the batch (launch_fsolve_sys_batch)
    int           batch_capacity = 0;
    int           Nmeas = 0;
    hipStream_t   stream = NULL;
    std::vector<void*> allocs;

    template<class T> bool alloc(T** p, size_t n)
    {
        *p = NULL;
        if(n == 0) n = 1;
        if(hipMalloc((void**)p, n*sizeof(T)) != hipSuccess)
        {
            set_error("out of device memory allocating %zu bytes for a factorization", n*sizeof(T));
            return false;
        }
        allocs.push_back((void*)*p);
        return true;
    }
    void release(void* p)
    {
        if(!p) return;
        for(size_t i = 0; i < allocs.size(); i++)
            if(allocs[i] == p) { hipStreamSynchronize(stream); hipFree(p); allocs.erase(allocs.begin() + i); return; }
    }
    ~mrcal_amd_factorization()
    {
        for(void* p : allocs) hipFree(p);
        if(stream) hipStreamDestroy(stream);
    }
};

extern "C" {

// the buffers of a factorization of this shape; nd and br are the caller's
static mrcal_amd_factorization* factorization_alloc(const NormalDims& nd_in, const BlockRanges& br, int Nmeas, int64_t Nnz)
{
    mrcal_amd_factorization* f = new mrcal_amd_factorization();
    f->Nmeas = Nmeas;
    f->nd = nd_in; f->br = br;
    const NormalDims& nd = f->nd;
    if((double)nd.Nc*nd.Nc*8.0 > 64e9)
    {
        set_error("the dense block of this factorization would be %d x %d: too large", nd.Nc, nd.Nc);
        delete f; return NULL;
    }
    bool ok = true;
    memset(&f->op, 0, sizeof(f->op));
    HIP_TRY(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking), ok = false);
    ok = ok && f->alloc(&f->d_Jp, (size_t)Nmeas+1);
    ok = ok && f->alloc(&f->d_Ji, (size_t)Nnz);
    ok = ok && f->alloc(&f->op.Jv, (size_t)Nnz);
    ok = ok && f->alloc(&f->op.x,  (size_t)Nmeas);
    ok = ok && f->alloc(&f->op.A,  (size_t)nd.Nc*nd.Nc);
    ok = ok && f->alloc(&f->op.Bt, (size_t)nd.NE*nd.Nc);
    ok = ok && f->alloc(&f->op.D,  (size_t)nd.NEb*36);
    ok = ok && f->alloc(&f->op.g,  (size_t)nd.Nstate);
    ok = ok && f->alloc(&f->op.scalars, (size_t)NSCALARS);
    ok = ok && f->alloc(&f->op.step_gn, (size_t)nd.Nstate);
    ok = ok && f->alloc(&f->d_op, 1);
    ok = ok && f->alloc(&f->F.Wt, (size_t)nd.NE*nd.Nc);
    ok = ok && f->alloc(&f->F.LD, (size_t)nd.NEb*36);
    ok = ok && f->alloc(&f->F.y,  (size_t)nd.NE);
    ok = ok && f->alloc(&f->F.S,  (size_t)nd.Nc*nd.Nc + nd.Nc);
    ok = ok && f->alloc(&f->F.Spart, schur_partial_doubles(nd));
    ok = ok && f->alloc(&f->F.Linv,  cholesky_large_workspace_doubles(nd.Nc));
    ok = ok && f->alloc(&f->F.status, 1);
    ok = ok && f->alloc(&f->d_rhs, (size_t)nd.Nstate);
    ok = ok && f->alloc(&f->d_sol, (size_t)nd.Nstate);
    ok = ok && f->alloc(&f->d_mm, 2);
    if(!ok) { delete f; return NULL; }
    f->F.r = f->F.S + (size_t)nd.Nc*nd.Nc;
    return f;
}
// the blocks of f->op hold the normal equations: eliminate, factor, check. Deletes f on failure
static int& last_create_status() { static thread_local int status = 0; return status; }
static mrcal_amd_factorization* factorization_finish(mrcal_amd_factorization* f)
{
    const NormalDims& nd = f->nd;
    bool ok = true;
    const OpRef R = { f->d_op, NULL, NULL };
    HIP_TRY(launch_factor_local(nd, f->br, R, f->F, 0.0, NULL, true, f->stream), ok = false);
    if(ok) HIP_TRY(launch_solve_backsub(nd, f->br, R, f->F, NULL, true, f->stream), ok = false);
    int status = 0;
    double bad_structure = 0.0;
    if(ok) HIP_TRY(hipMemcpyAsync(&status, f->F.status, sizeof(int), hipMemcpyDeviceToHost, f->stream), ok = false);
    if(ok) HIP_TRY(hipMemcpyAsync(&bad_structure, f->op.scalars + SC_BAD_STRUCTURE, sizeof(double), hipMemcpyDeviceToHost, f->stream), ok = false);
    if(ok) HIP_TRY(hipStreamSynchronize(f->stream), ok = false);
    if(!ok) { delete f; return NULL; }
    if(bad_structure == 2.0)
    {
        // (rows_repro_row / rows_generic_wave<true>: the pre-rounded sums are made of constants 2^(c_i + c_j + N ...), which
        //  must be doubles: columns whose largest |value| is beyond ~1e+-100, or not finite, are not served)
        set_error("the factorization failed: the matrix holds values that are not finite, or columns beyond 1e+-100 in magnitude");
        delete f;
        return NULL;
    }
    if(bad_structure != 0.0)
    {
        set_error("a row couples two eliminated blocks, or a column index is out of range: this matrix does not have the declared structure");
        delete f;
        return NULL;
    }
    if(status != 0)
    {
        // like the reference: "CHOLMOD factorization failed (singular JtJ)"
        set_error("the factorization failed: JtJ is not positive definite");
        last_create_status() = 1;
        delete f;
        return NULL;
    }
    last_create_status() = 0;
    return f;
}

// why the last _create() / _create_from_problem() of this thread returned what it did: 0 a factorization; 1 NULL because
// JtJ is not positive definite (the reference's None); 2 NULL for any other reason (no memory, a shard, a malformed matrix)
int mrcal_amd_factorization_last_status(void) { return last_create_status(); }

mrcal_amd_factorization_t*
mrcal_amd_factorization_create(int Nmeas, int Nstate,
                               const int32_t* rowptr, const int32_t* colidx, const double* values,
                               int Nstate_shared_leading, int Nframe_blocks, int Npoint_blocks, int Nwarp)
{
    last_create_status() = 2;
    last_error_string().clear();
    if(mrcal_amd_device_count() <= 0)
    {
        set_error("no HIP device is visible: libmrcal_amd has no CPU fallback");
        return NULL;
    }
    const int NE = 6*Nframe_blocks + 3*Npoint_blocks;
    if(Nstate_shared_leading < 0 || Nstate_shared_leading + NE + Nwarp != Nstate ||
       (Nwarp != 0 && Nwarp != 2) || Nmeas < 0)
    {
        set_error("inconsistent state partition: %d + 6*%d + 3*%d + %d != %d",
                  Nstate_shared_leading, Nframe_blocks, Npoint_blocks, Nwarp, Nstate);
        return NULL;
    }
    const int64_t Nnz = rowptr[Nmeas];
    NormalDims nd;
    memset(&nd, 0, sizeof(nd));
    nd.Nstate = Nstate; nd.Nwarp = Nwarp;
    nd.i_state_warp = Nstate - Nwarp; nd.Nc = Nstate_shared_leading + nd.Nwarp;
    normal_dims_set_partition(nd, Nstate_shared_leading);
    nd.NE = NE; nd.Nfb = Nframe_blocks; nd.Npb = Npoint_blocks; nd.NEb = nd.Nfb + nd.Npb;
    BlockRanges br;
    memset(&br, 0, sizeof(br));
    br.frame_lo = 0; br.frame_hi = nd.Nfb; br.point_lo = nd.Nfb; br.point_hi = nd.NEb;
    mrcal_amd_factorization* f = factorization_alloc(nd, br, Nmeas, Nnz);
    if(f == NULL) return NULL;

    bool ok = true;
    HIP_TRY(hipMemcpy(f->d_Jp, rowptr, ((size_t)Nmeas+1)*sizeof(int32_t), hipMemcpyHostToDevice), ok = false);
    HIP_TRY(hipMemcpy(f->d_Ji, colidx, (size_t)Nnz*sizeof(int32_t),        hipMemcpyHostToDevice), ok = false);
    HIP_TRY(hipMemcpy(f->op.Jv, values, (size_t)Nnz*sizeof(double),        hipMemcpyHostToDevice), ok = false);
    // (on f->stream: it is a non-blocking stream, which does not order itself behind the null stream's memsets)
    HIP_TRY(hipMemsetAsync(f->op.x, 0, (size_t)(Nmeas > 0 ? Nmeas : 1)*sizeof(double), f->stream), ok = false);
    HIP_TRY(hipMemcpy(f->d_op, &f->op, sizeof(OpDev), hipMemcpyHostToDevice), ok = false);
    HIP_TRY(hipMemsetAsync(f->F.status, 0, sizeof(int), f->stream), ok = false);
    if(!ok) { delete f; return NULL; }

    // (the partition is validated by the assembly itself, on the device: a row that touches two eliminated
    //  blocks or has a column out of range raises SC_BAD_STRUCTURE. A loop over all entries on the host was
    //  25 ms at 37 M entries)
    // This assembly goes row by row with atomics, on sums made so that no addition rounds (rows_repro_kernel, round 4): the same
    // bits whatever order the atomics land in.
    // The factorization optimizer_callback() returns does not come through here: mrcal_amd_factorization_create_from_problem()
    const OpRef R = { f->d_op, NULL, NULL };
    double* scratch = NULL;
    HIP_TRY(hipMalloc((void**)&scratch, assemble_rows_scratch_doubles(f->nd)*sizeof(double)), { delete f; return NULL; });
    const hipError_t ea = launch_assemble_rows(f->nd, R, Nmeas, f->d_Jp, f->d_Ji, f->stream, scratch, Nnz);
    mrcal_amd_factorization* out = (ea == hipSuccess) ? factorization_finish(f) : NULL;      // (synchronizes the stream)
    if(ea != hipSuccess) { set_error("launch_assemble_rows: %s", hipGetErrorString(ea)); delete f; }
    if(scratch != NULL) hipFree(scratch);
    return out;
}

// The factorization of JtJ at the problem's current state, from the problem itself (round 4): x, J and the block
// normal equations are evaluated by the problem's own kernels - the per-observation Grams, the fixed-order sums:
// no atomics, the same bits every time - and copied device to device; then the same elimination and Cholesky as
// above. What optimizer_callback() returns is built this way: the CSR it hands to the caller does not come back up
// across PCIe (31 ms at the metric's size), and nothing in it depends on the order in which atomics landed
mrcal_amd_factorization_t* mrcal_amd_factorization_create_from_problem(mrcal_amd_problem_t* P)
{
    last_error_string().clear();
    last_create_status() = 2;
    if(P == NULL) { set_error("no problem"); return NULL; }
    if((int)P->board_sel.size() != P->L.dims.Nobservations_board || P->comm != NULL)
    {
        set_error("factorization: this problem is a shard (it holds a part of the rows)");
        return NULL;
    }
    if(!problem_prepare_solver(P)) return NULL;
    if(!problem_evaluate_op(P, P->icur, true, true)) return NULL;
    const mrcal_amd_oppoint& N = P->op[P->icur];
    mrcal_amd_factorization* f = factorization_alloc(P->nd, P->br, P->L.Nmeas, P->Nnz);
    if(f == NULL) return NULL;
    const NormalDims& nd = f->nd;
    bool ok = true;
    hipStream_t st = P->stream;
    auto copy = [&](void* dst, const void* src, size_t bytes)
    {
        if(ok && bytes > 0) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st), ok = false);
    };
    copy(f->d_Jp,  P->d_Jp, ((size_t)P->L.Nmeas + 1)*sizeof(int32_t));
    copy(f->d_Ji,  P->d_Ji, (size_t)P->Nnz*sizeof(int32_t));
    copy(f->op.Jv, N.Jv,    (size_t)P->Nnz*sizeof(double));
    copy(f->op.A,  N.A,     (size_t)nd.Nc*nd.Nc*sizeof(double));
    copy(f->op.Bt, N.Bt,    (size_t)nd.NE*nd.Nc*sizeof(double));
    copy(f->op.D,  N.D,     (size_t)nd.NEb*36*sizeof(double));
    copy(f->op.g,  N.g,     (size_t)nd.Nstate*sizeof(double));
    copy(f->op.scalars, N.scalars, (size_t)NSCALARS*sizeof(double));
    if(ok) HIP_TRY(hipStreamSynchronize(st), ok = false);
    if(ok) HIP_TRY(hipMemcpy(f->d_op, &f->op, sizeof(OpDev), hipMemcpyHostToDevice), ok = false);
    if(ok) HIP_TRY(hipMemsetAsync(f->F.status, 0, sizeof(int), f->stream), ok = false);
    if(ok) HIP_TRY(hipMemsetAsync(f->op.x, 0, (size_t)(f->Nmeas > 0 ? f->Nmeas : 1)*sizeof(double), f->stream), ok = false);
    if(!ok) { delete f; return NULL; }
    return factorization_finish(f);
}
int mrcal_amd_factorization_Nmeasurements(const mrcal_amd_factorization_t* f) { return f->Nmeas; }

void mrcal_amd_factorization_destroy(mrcal_amd_factorization_t* f) { delete f; }
int  mrcal_amd_factorization_Nstate(const mrcal_amd_factorization_t* f) { return f->nd.Nstate; }

// xt[i,:] = (JtJ)^-1 bt[i,:], i in [0,Nrhs). Host pointers, C-contiguous (Nrhs,Nstate)
// The right-hand sides go up in batches and the solutions come down in batches: a copy to or from pageable
// host memory per right-hand side is a synchronization each, more than the six kernels of a solve take
static bool solve_batched(mrcal_amd_factorization_t* f, int sys, const double* bt, int Nrhs, double* xt)
{
    const size_t n = (size_t)f->nd.Nstate;
    if(Nrhs <= 0) return true;
    // <= 32 MB of right-hand sides each way, <= 256 MB of scratch (y, r and the partial sums of every right-hand side)
    const size_t per_rhs = fsolve_batch_scratch_doubles(f->nd, 1);
    const size_t part_per_rhs = per_rhs - (size_t)f->nd.NE - (size_t)f->nd.Nc;
    size_t cap = std::min<size_t>((size_t)Nrhs, ((size_t)32 << 20)/(n*sizeof(double) + 1));
    cap = std::min<size_t>(cap, ((size_t)256 << 20)/(per_rhs*sizeof(double) + 1));
    const int BATCH = (int)std::max<size_t>(1, std::min<size_t>(cap, 16384));
    if(f->batch_capacity < BATCH)
    {
        double *rb = NULL, *sb = NULL, *sc = NULL;
        if(!f->alloc(&rb, (size_t)BATCH*n) || !f->alloc(&sb, (size_t)BATCH*n) || !f->alloc(&sc, (size_t)BATCH*per_rhs)) return false;
        f->release(f->d_rhs_batch); f->release(f->d_sol_batch); f->release(f->d_batch_scratch);
        f->d_rhs_batch = rb; f->d_sol_batch = sb; f->d_batch_scratch = sc; f->batch_capacity = BATCH;
    }
    for(int i0 = 0; i0 < Nrhs; i0 += BATCH)
    {
        const int nb = std::min(BATCH, Nrhs - i0);
        double* y    = f->d_batch_scratch;
        double* r    = y + (size_t)BATCH*f->nd.NE;
        double* part = r + (size_t)BATCH*f->nd.Nc;
        HIP_TRY(hipMemcpyAsync(f->d_rhs_batch, bt + (size_t)i0*n, (size_t)nb*n*sizeof(double), hipMemcpyHostToDevice, f->stream), return false);
        HIP_TRY(launch_fsolve_sys_batch(f->nd, f->F, sys, f->d_rhs_batch, f->d_sol_batch, nb, y, r, part, part_per_rhs, f->stream), return false);
        HIP_TRY(hipMemcpyAsync(xt + (size_t)i0*n, f->d_sol_batch, (size_t)nb*n*sizeof(double), hipMemcpyDeviceToHost, f->stream), return false);
    }
    HIP_TRY(hipStreamSynchronize(f->stream), return false);
    return true;
}
bool mrcal_amd_factorization_solve(mrcal_amd_factorization_t* f, const double* bt, int Nrhs, double* xt)
{
    last_error_string().clear();
    return solve_batched(f, FSOLVE_A, bt, Nrhs, xt);
}

// The other systems of cholmod_solve2() (mrcal-pywrap.c:467-493; sys = CHOLMOD's
// codes: 0 A, 1 LDLt, 2 LD, 3 DLt, 4 L, 5 Lt, 6 D, 7 P, 8 Pt) against this
// factorization: L L^T = P (JtJ) P^T with the frame/point blocks first and D = I
// (factorization_solve.hip, "factor order"). Vectors of the L/D systems live in that
// order, as CHOLMOD's live in its own
bool mrcal_amd_factorization_solve_sys(mrcal_amd_factorization_t* f, int sys, const double* bt, int Nrhs, double* xt)
{
    last_error_string().clear();
    if(sys < 0 || sys > FSOLVE_Pt) { set_error("unknown system %d", sys); return false; }
    return solve_batched(f, sys, bt, Nrhs, xt);
}

// y = Jt x with the J this factorization was made from (it is resident): mrcal-genpywrap.py:658-731 _Jt_x
bool mrcal_amd_factorization_Jt_x(mrcal_amd_factorization_t* f, const double* x, double* y)
{
    last_error_string().clear();
    const size_t n = (size_t)f->nd.Nstate;
    HIP_TRY(hipMemcpyAsync(f->op.x, x, (size_t)f->Nmeas*sizeof(double), hipMemcpyHostToDevice, f->stream), return false);
    double* scratch = NULL;
    HIP_TRY(hipMalloc((void**)&scratch, csr_Jt_x_scratch_doubles(f->Nmeas, (int)n)*sizeof(double)), return false);
    bool ok = true;
    HIP_TRY(launch_csr_Jt_x(f->Nmeas, (int)n, f->d_Jp, f->d_Ji, f->op.Jv, f->op.x, f->d_sol, scratch, f->stream), ok = false);
    if(ok) HIP_TRY(hipMemcpyAsync(y, f->d_sol, n*sizeof(double), hipMemcpyDeviceToHost, f->stream), ok = false);
    if(ok) HIP_TRY(hipStreamSynchronize(f->stream), ok = false);
    hipFree(scratch);
    return ok;
}

// out (Nx x Nx) = A Jt J At over the Nleading_rows_J leading rows of J, A (Nx x Nstate) row-major, host:
// mrcal-genpywrap.py:477-657 _A_Jt_J_At. Nx <= 8
bool mrcal_amd_factorization_A_Jt_J_At(mrcal_amd_factorization_t* f, const double* A, int Nx, int Nleading_rows_J, double* out)
{
    last_error_string().clear();
    if(Nx < 1 || Nx > 8) { set_error("_A_Jt_J_At: A must have 1..8 rows, not %d", Nx); return false; }
    if(Nleading_rows_J <= 0 || Nleading_rows_J > f->Nmeas)
    {
        set_error("Nleading_rows_J must be passed, and must be > 0 (and at most the %d rows of J)", f->Nmeas);
        return false;
    }
    const size_t n = (size_t)f->nd.Nstate;
    double *dA = NULL, *dout = NULL;
    bool ok = true;
    const size_t nscratch = (size_t)64*((Nleading_rows_J + 255)/256);
    HIP_TRY(hipMalloc((void**)&dA, (size_t)Nx*n*sizeof(double)), return false);
    HIP_TRY(hipMalloc((void**)&dout, (64 + nscratch)*sizeof(double)), ok = false);
    if(ok) HIP_TRY(hipMemcpyAsync(dA, A, (size_t)Nx*n*sizeof(double), hipMemcpyHostToDevice, f->stream), ok = false);
    if(ok) HIP_TRY(launch_csr_A_Jt_J_At(Nx, Nleading_rows_J, (int)n, f->d_Jp, f->d_Ji, f->op.Jv, dA, dout, dout + 64, f->stream), ok = false);
    if(ok) HIP_TRY(hipMemcpyAsync(out, dout, (size_t)Nx*Nx*sizeof(double), hipMemcpyDeviceToHost, f->stream), ok = false);
    if(ok) HIP_TRY(hipStreamSynchronize(f->stream), ok = false);
    hipFree(dA); hipFree(dout);
    return ok;
}

// The same two for a CSR matrix that is on the host (the reference's signatures take the
// p, i, x arrays: mrcal._mrcal_npsp._Jt_x, _A_Jt_J_At): upload, compute, free
namespace {
struct CsrOnDevice
{
    int32_t *Jp = NULL, *Ji = NULL; double *Jx = NULL; bool ok = false;
    CsrOnDevice(int Nrows, const int32_t* p, const int32_t* i, const double* x)
    {
        const size_t nnz = (size_t)p[Nrows];
        if(hipMalloc((void**)&Jp, ((size_t)Nrows+1)*sizeof(int32_t)) != hipSuccess) return;
        if(hipMalloc((void**)&Ji, (nnz ? nnz : 1)*sizeof(int32_t)) != hipSuccess) return;
        if(hipMalloc((void**)&Jx, (nnz ? nnz : 1)*sizeof(double)) != hipSuccess) return;
        if(hipMemcpy(Jp, p, ((size_t)Nrows+1)*sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) return;
        if(nnz && hipMemcpy(Ji, i, nnz*sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) return;
        if(nnz && hipMemcpy(Jx, x, nnz*sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return;
        ok = true;
    }
    ~CsrOnDevice() { hipFree(Jp); hipFree(Ji); hipFree(Jx); }
};
}
// the caller's CSR is what the kernels index with: rowptr non-decreasing from 0, every column inside the matrix
static bool csr_is_valid(int Nrows, int Ncols, const int32_t* Jp, const int32_t* Ji)
{
    if(Nrows < 0 || Ncols < 0 || Jp == NULL || Jp[0] != 0) { set_error("malformed CSR matrix: rowptr must start at 0"); return false; }
    for(int r = 0; r < Nrows; r++)
        if(Jp[r+1] < Jp[r]) { set_error("malformed CSR matrix: rowptr decreases at row %d", r); return false; }
    const int64_t nnz = Jp[Nrows];
    unsigned bad = 0;
    for(int64_t p = 0; p < nnz; p++) bad |= (unsigned)((unsigned)Ji[p] >= (unsigned)Ncols);
    if(bad) { set_error("malformed CSR matrix: a column index is outside [0,%d)", Ncols); return false; }
    return true;
}
bool mrcal_amd_csr_Jt_x(int Nrows, int Ncols, const int32_t* Jp, const int32_t* Ji, const double* Jx,
                        const double* x, double* y)
{
    last_error_string().clear();
    if(mrcal_amd_device_count() <= 0) { set_error("no HIP device is visible: libmrcal_amd has no CPU fallback"); return false; }
    if(!csr_is_valid(Nrows, Ncols, Jp, Ji)) return false;
    CsrOnDevice J(Nrows, Jp, Ji, Jx);
    if(!J.ok) { set_error("could not put J on the device"); return false; }
    double *dx = NULL, *dy = NULL;
    bool ok = true;
    HIP_TRY(hipMalloc((void**)&dx, (size_t)(Nrows > 0 ? Nrows : 1)*sizeof(double)), return false);
    HIP_TRY(hipMalloc((void**)&dy, ((size_t)(Ncols > 0 ? Ncols : 1) + csr_Jt_x_scratch_doubles(Nrows, Ncols))*sizeof(double)), ok = false);
    if(ok) HIP_TRY(hipMemcpy(dx, x, (size_t)Nrows*sizeof(double), hipMemcpyHostToDevice), ok = false);
    if(ok) HIP_TRY(hipDeviceSynchronize(), ok = false);
    if(ok) HIP_TRY(launch_csr_Jt_x(Nrows, Ncols, J.Jp, J.Ji, J.Jx, dx, dy, dy + (Ncols > 0 ? Ncols : 1), NULL), ok = false);
    if(ok) HIP_TRY(hipMemcpy(y, dy, (size_t)Ncols*sizeof(double), hipMemcpyDeviceToHost), ok = false);
    hipFree(dx); hipFree(dy);
    return ok;
}
bool mrcal_amd_csr_A_Jt_J_At(int Nrows, int Ncols, const int32_t* Jp, const int32_t* Ji, const double* Jx,
                             const double* A, int Nx, int Nleading_rows_J, double* out)
{
    last_error_string().clear();
    if(mrcal_amd_device_count() <= 0) { set_error("no HIP device is visible: libmrcal_amd has no CPU fallback"); return false; }
    if(Nx < 1 || Nx > 8) { set_error("_A_Jt_J_At: A must have 1..8 rows, not %d", Nx); return false; }
    if(Nleading_rows_J <= 0 || Nleading_rows_J > Nrows)
    {
        set_error("Nleading_rows_J must be passed, and must be > 0 (and at most the %d rows of J)", Nrows);
        return false;
    }
    if(!csr_is_valid(Nrows, Ncols, Jp, Ji)) return false;
    CsrOnDevice J(Nrows, Jp, Ji, Jx);
    if(!J.ok) { set_error("could not put J on the device"); return false; }
    double *dA = NULL, *dout = NULL;
    bool ok = true;
    const size_t nscratch = (size_t)64*((Nleading_rows_J + 255)/256);
    HIP_TRY(hipMalloc((void**)&dA, (size_t)Nx*Ncols*sizeof(double)), return false);
    HIP_TRY(hipMalloc((void**)&dout, (64 + nscratch)*sizeof(double)), ok = false);
    if(ok) HIP_TRY(hipMemcpy(dA, A, (size_t)Nx*Ncols*sizeof(double), hipMemcpyHostToDevice), ok = false);
    if(ok) HIP_TRY(hipDeviceSynchronize(), ok = false);
    if(ok) HIP_TRY(launch_csr_A_Jt_J_At(Nx, Nleading_rows_J, Ncols, J.Jp, J.Ji, J.Jx, dA, dout, dout + 64, NULL), ok = false);
    if(ok) HIP_TRY(hipMemcpy(out, dout, (size_t)Nx*Nx*sizeof(double), hipMemcpyDeviceToHost), ok = false);
    hipFree(dA); hipFree(dout);
    return ok;
}

// like cholmod_rcond() for an LL' factorization: (min diag / max diag)^2
double mrcal_amd_factorization_rcond(mrcal_amd_factorization_t* f)
{
    last_error_string().clear();
    double mm[2] = { 1e300, 0.0 };
    HIP_TRY(hipMemcpyAsync(f->d_mm, mm, sizeof(mm), hipMemcpyHostToDevice, f->stream), return -1.0);
    HIP_TRY(launch_fsolve_diag_minmax(f->nd, f->F, f->d_mm, f->stream), return -1.0);
    HIP_TRY(hipMemcpyAsync(mm, f->d_mm, sizeof(mm), hipMemcpyDeviceToHost, f->stream), return -1.0);
    HIP_TRY(hipStreamSynchronize(f->stream), return -1.0);
    if(!(mm[1] > 0.0)) return 0.0;
    const double q = mm[0]/mm[1];
    return q*q;
}

} // extern "C"
