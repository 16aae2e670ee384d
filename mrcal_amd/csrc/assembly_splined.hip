// The splined models' assembly: local Grams on the matrix cores, staged triangles, the gathers, the compaction's plan
// (round 6: one of the translation units solver_kernels.hip was cut into; solver_device.hpp has what they share)
#include "solver_device.hpp"
#include "assembly_splined.hpp"
#include "solver_kernel_decls.hpp"

namespace mrcal_amd {

// Board rows of the SPLINED models. There is no per-observation Gram for them
// from the Jacobian kernel (the columns of a row depend on where the corner
// lands in the knot grid), and one lane per row with global atomics for every
// pair of its ~26 entries is 108 M atomics at 160k rows: 13 ms. But:
//   - the rows of ONE observation only touch a small set of camera-block
//     variables: the core, the extrinsics, the warp and the knots under the
//     board, K = (order+1 + span)^2 of them per surface;
//   - an x row touches the x surface only, a y row the y surface only
//     (board_splined_kernel: column col0 + .. + xy), and all rows are equally long
// Three kernels, no atomics whose order matters, the same bits every time:
//   assemble_splined_kernel: one workgroup per frame and surface (x rows, y rows: a pass each). A pass writes its
//     rows DENSELY over the local columns
//         [ K knots | 4 core | 6 extrinsics | 2 warp | 6 frame | x ]
//     into an LDS tile and forms the lower triangle of that matrix's Gram on the FP64 matrix cores: one product
//     yields the A, Bt, D_f, g and |x|^2 contributions at once. What belongs to the frame (Bt, D_f, g_f) leaves
//     through LDS sums - one addition per entry and workgroup -; the camera-block rows and the x row are
//     STAGED, a packed lower triangle per pass, with the knot box in a header. An observation whose box does not fit
//     the tile (more than SPL_TW-19 = 109 knots: a close-up) is cut into overlapping SUB-BOXES, each a pass of its own
//     over the corners it owns (solver_kernels.hpp SPL_MAXSUB)
//   assemble_splined_gather_knots_kernel: a workgroup per control point's row of the camera block, a LANE per place
//     of the row that can hold anything (25 + the core); a pass that holds the row is one load for half a wave
//   assemble_splined_gather_kernel: one workgroup per row that every pass holds (core, extrinsics, warp; the x row:
//     g and |x|^2), split over SPLG_E workgroups each and summed in order by assemble_splined_combine_kernel. It walks
//     the passes in order and adds the row's staged entries into an LDS copy of the row. (Until the end of round 4
//     the control points' rows came this way too: MRCAL_AMD_SPL_ROW_GATHER)
// (The one-kernel version flushed each pass's tile sums with global atomics, 8000 of them
//  per observation; two thirds of a workgroup's time was that flush: 340 us at 30 x 20 knots,
//  800 frames.) An observation of more than SPL_MAXSUB sub-boxes goes the generic way,
// row by row, with atomics; its header says so. Only the lower triangle of A is written
#define SPL_TW      128         // local columns
#define SPL_NDENSE  12
#define SPL_NEXTRA  (SPL_NDENSE + 6 + 1)
// The Gram of a pass on the matrix cores (round 4; it was 36 multiply-adds per row and thread fed by 16 LDS reads:
// 57k of a workgroup's 142k cycles). NS = ceil(NC/16) tile columns in use; the lower triangle of the NS x NS grid
// of 16 x 16 tiles is dealt to the four waves BY TILE ROW, so that a wave's tiles share their operands: wave w has
// row Ia = NS-1-w with its Ia+1 tiles and, if it exists, row Ib = w-(8-NS) (NS = 8: 9 tiles each; 7: 7 each;
// 6: 6,5,5,5; 5: 5,4,3,3). v_mfma_f64_16x16x4: lane (r16 = lane % 16, kq = lane / 16) feeds Jd[4 s + kq][16 I + r16]
// and Jd[4 s + kq][16 J + r16], register v of the result is G[16 I + kq + 4 v][16 J + r16]. A step s of a wave is
// Ia+1 (+2) LDS reads for up to 9 matrix instructions; the reads of step s+1 are issued before the instructions of
// step s. Row stride LD = 16 NS, + 16 if that is a multiple of 32 doubles: the four kq groups of a read then start 32
// banks apart. The rows of a pass that fit the tile (64 KB: two workgroups a CU with room to spare) are taken in one go (100 corners x 80 columns do).
typedef double spl_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int spl_tri(int r) { return (r*(r + 1)) >> 1; }
// where pass (observation o, surface xy, sub-box isub) stages its triangle, and its header: the first sub-box of every
// observation where the only one always was (the usual case reads what it always read), the others behind all of those
// (in allocations of their own: with one allocation four times the size, the launches that run side by side - the
//  gather, the SYRK - were 40% slower at BASELINE configuration 2, which has no second sub-box anywhere)
__device__ __forceinline__ double* spl_slot(const AssemblyPlan& plan, int o, int xy, int isub)
{
    return (isub == 0) ? plan.chunk_part  + ((size_t)2*o + xy)*SPL_TRI
                       : plan.chunk_extra + (((size_t)2*o + xy)*(SPL_MAXSUB - 1) + (isub - 1))*SPL_TRI;
}
__device__ __forceinline__ SplHdr* spl_hdr_at(const AssemblyPlan& plan, int o, int isub)
{
    return (isub == 0) ? plan.spl_hdr + o : plan.spl_hdr_extra + (size_t)o*(SPL_MAXSUB - 1) + (isub - 1);
}
// a workgroup barrier that orders the LDS traffic only: the global stores in flight (the staged triangle) are not waited for
__device__ __forceinline__ void spl_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// (+)= the Gram of nsteps x 4 rows (rows past the data are zero): the NA = Ia + 1 tiles of row Ia in acc[0 .. NA-1], the
// nb <= min(4, 9 - NA) tiles of row Ib behind them. FIRST: the accumulators start here (the first rows of a pass)
template<int NA, bool FIRST> __device__ __forceinline__
void spl_gram_mfma(const double* __restrict__ Jd, int LD, int nsteps, int r16, int kq, int offIa, int offIb, int nb, spl_d4 (&acc)[9])
{
    if(FIRST)
    {
#pragma unroll
        for(int u = 0; u < 9; u++) acc[u] = spl_d4{0.0, 0.0, 0.0, 0.0};
    }
    const double* __restrict__ rowp = Jd + kq*LD + r16;
    const int step = 4*LD;
    double b0[NA], b1[NA], aA0, aA1, aB0 = 0.0, aB1 = 0.0;
    auto load = [&](double (&bb)[NA], double& aA, double& aB, const double* __restrict__ rp)
    {
#pragma unroll
        for(int J = 0; J < NA; J++) bb[J] = rp[16*J];
        aA = rp[offIa];
        if(nb > 0) aB = rp[offIb];
    };
    auto mma = [&](const double (&bb)[NA], double aA, double aB)
    {
#pragma unroll
        for(int J = 0; J < NA; J++) acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(aA, bb[J], acc[J], 0, 0, 0);
#define SPL_ROWB(J) if constexpr((J) < NA && NA + (J) < 9) { if((J) < nb) acc[NA + (J)] = __builtin_amdgcn_mfma_f64_16x16x4f64(aB, bb[J], acc[NA + (J)], 0, 0, 0); }
        SPL_ROWB(0) SPL_ROWB(1) SPL_ROWB(2) SPL_ROWB(3)
#undef SPL_ROWB
    };
    load(b0, aA0, aB0, rowp);
    int s = 0;
#pragma unroll 1
    for(; s + 2 <= nsteps; s += 2)
    {
        load(b1, aA1, aB1, rowp + step);
        mma(b0, aA0, aB0);
        rowp += 2*step;
        // (past the end: the last rows once more, unused)
        load(b0, aA0, aB0, (s + 2 < nsteps) ? rowp : rowp - step);
        mma(b1, aA1, aB1);
    }
    if(s < nsteps) mma(b0, aA0, aB0);
}
// c / wx = (c spl_magic(wx)) >> 16 for 0 <= c < 2^16/wx: the local columns (< 128) of a box up to 128 wide. (An integer
// division is ~40 vector instructions; the gather made four for every pass that held its row, and its waves share a SIMD
// four at a time: those, and nine ds_bpermute, were most of the 18k cycles a batch of three passes took)
__host__ __device__ __forceinline__ int spl_magic(int wx) { return (65536 + wx - 1)/(wx > 0 ? wx : 1); }
// state index of local column c of a pass, -1: not a camera-block variable of this pass (a frame column, x,
// or a variable that is not being optimized)
__device__ __forceinline__
int spl_col_state(const DeviceProblem& P, const NormalDims& nd, int c, int K, int ix0, int iy0, int wx, int xy,
                  int i_state_intrinsics, int i_state_extrinsics, int wx_magic /* spl_magic(wx): c / wx without the division */)
{
    if(c < K)
    {
        const int cy = (c*wx_magic) >> 16, cx = c - cy*wx;
        return (i_state_intrinsics >= 0)
            ? i_state_intrinsics + P.Ncore_state + 2*((iy0 + cy)*P.cfg.spline_Nx + ix0 + cx) + xy : -1;
    }
    const int d = c - K;
    if(d < 4)           return (i_state_intrinsics >= 0 && d < P.Ncore_state) ? i_state_intrinsics + d : -1;
    if(d < 10)          return (i_state_extrinsics >= 0) ? i_state_extrinsics + (d - 4) : -1;
    if(d < SPL_NDENSE)  return nd.Nwarp ? nd.i_state_warp + (d - 10) : -1;
    return -1;
}
// (not inlined: what it needs in registers should not count against the kernel's usual path)
__device__ __noinline__
void spl_rows_fallback(const NormalDims nd /* by value: by reference the kernel's copy lives in scratch, and the usual path reads it from there */, const OpDev& O, int r_first, int r1,
                       const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji)
{
    for(int r = r_first; r < r1; r += blockDim.x) rows_generic_row(nd, O, r, r1, Jp, Ji);
}
bool splined_needs_repro_rows(const DeviceProblem& P)
{
    if(P.lens_type != MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC) return false;
    // discrete points whose rows' patch columns move with every evaluation, or a board that can cover more than the
    // sub-boxes hold
    return (P.Nobs_point > 0 && P.Ndist_state > 0) || (P.Nobs_board > 0 && P.Nframes > 0 && spl_fallback_possible(P));
}
#ifndef SPL_WAVES_PER_EU
#define SPL_WAVES_PER_EU 2
#endif
__device__ __forceinline__ void rows_pairs_body(const NormalDims& nd, const OpDev& O, int row0, int row1, const int32_t* __restrict__ Jp,
                                                const int32_t* __restrict__ Ji, double* __restrict__ row_part, int block);
__device__ __forceinline__ void spl_compact_body(const DeviceProblem& P, const NormalDims& nd, const OpDev& O, int* __restrict__ lds_c, const int* __restrict__ nd_lim);
// Riding along behind the frames' workgroups (round 5; they were launches of their own on the side stream, behind a fork
// that cost the main stream 8 us): `npairs_extra` workgroups of rows_pairs_body() - the regularization rows from
// pairs_row0 on, which write the camera block's A and g: nothing this kernel's own workgroups touch (never where they can
// fall back to row-by-row atomics: the launcher sees to it) - and, if compact_extra, one of spl_compact_body()
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SPL_WAVES_PER_EU)))
void assemble_splined_kernel(DeviceProblem P, NormalDims nd, OpRef R, AssemblyPlan plan,
                             const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji,
                             int npairs_extra, int pairs_row0, int compact_extra)
{
    if(opref_skip(R)) return;
    __shared__ __attribute__((aligned(16))) double Jd[SPL_LDS_DOUBLES];     // [rows][LD] of a pass
    if((int)blockIdx.x >= 2*P.Nframes)
    {
        const int e = (int)blockIdx.x - 2*P.Nframes;
        if(e < npairs_extra)   rows_pairs_body(nd, opref_get(R), pairs_row0, P.Nmeas, Jp, Ji, plan.row_part, e);
        else if(compact_extra) spl_compact_body(P, nd, opref_get(R), (int*)Jd, plan.nd_lim);
        return;
    }
    __shared__ double F[7*SPL_TW];          // the frame rows and the x row of a pass's Gram
    __shared__ double FD[7*6];              // the frame's own block and its part of the gradient, summed over the passes
    __shared__ unsigned char own[1024];     // sub-boxes: which of them a corner belongs to
    __shared__ unsigned short crow[1024];   // ... and its row among the corners of the sub-box being assembled (0xffff: another's)
    __shared__ int s_nown;
    __shared__ double FB[6*SPL_NDENSE];     // the frame rows against the core, the extrinsics and the warp: over an observation's two passes (the warp: over the frame)
    const OpDev& O = opref_get(R);
    const double* __restrict__ Jv = O.Jv;
    const double* __restrict__ x  = O.x;
    // A workgroup per frame and SURFACE (round 4; it was per frame, the two passes one after the other): the x rows touch
    // the x surface's control points only, the y rows the y surface's, so the two passes share nothing they write but
    // the frame's block, its part of the gradient and the core / extrinsics / warp columns of Bt - to each of which
    // each of the two adds ONE number, atomically, onto zero: a + b is b + a. Twice the workgroups of half the length:
    // nothing at BASELINE configuration 2 (800 frames on 512 places), and a calibration of 186 frames of close-ups,
    // eight passes each, no longer leaves a quarter of the CUs without a workgroup
    const int f = blockIdx.x >> 1, xy0 = blockIdx.x & 1, t = threadIdx.x;
    const int lane = t & 63, r16 = lane & 15, kq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int o0 = __builtin_amdgcn_readfirstlane(plan.frame_obs_begin[f]), o1 = __builtin_amdgcn_readfirstlane(plan.frame_obs_begin[f+1]);
    const int NPTS = P.W*P.H, Nx = P.cfg.spline_Nx;
    const int Ncs = P.Ncore_state;
    if(t < 42) FD[t] = 0.0;
    if(t < 6*SPL_NDENSE) FB[t] = 0.0;
    // (-DSPL_TS: cycles per phase, printed by three of the workgroups)
#ifdef SPL_TS
    long long ts_bbox = 0, ts_zero = 0, ts_scatter = 0, ts_gram = 0, ts_out = 0, ts0 = clock64(), ts1;
    const long long tw0 = wall_clock64(), tc0 = ts0;
#define SPL_TICK(what) { ts1 = clock64(); what += ts1 - ts0; ts0 = ts1; }
#else
#define SPL_TICK(what)
#endif

    for(int o = o0; o < o1; o++)
    {
        // (the observation's record and its box are the same for every lane, and said so: the tile's geometry, the
        //  waves' shares and the loops' bounds stay in scalar registers)
        const BoardObsMeta mv = P.board_meta[o];
        const int4 boxv = ((const int4*)O.spl_box)[o];
        const int m_isi = __builtin_amdgcn_readfirstlane(mv.i_state_intrinsics);
        const int m_ise = __builtin_amdgcn_readfirstlane(mv.i_state_extrinsics);
        const int r0 = __builtin_amdgcn_readfirstlane(mv.i_meas0), r1 = r0 + 2*NPTS;
        // (rowptr[i_meas0 + r] = i_nnz0 + r*nnz_per_row: board_splined_kernel. From the record: two loads fewer in line)
        const int p00 = __builtin_amdgcn_readfirstlane((int)mv.i_nnz0);
        const int L   = __builtin_amdgcn_readfirstlane(mv.nnz_per_row);     // entries per row, the same for all rows of the observation
        // the box of control points under this observation's inliers: board_splined_kernel left it with the rows
        const int4 box  = make_int4(__builtin_amdgcn_readfirstlane(boxv.x), __builtin_amdgcn_readfirstlane(boxv.y),
                                    __builtin_amdgcn_readfirstlane(boxv.z), __builtin_amdgcn_readfirstlane(boxv.w));
        const bool any = (P.Ndist_row > 0) && box.y >= 0;
        const int ox0 = any ? box.x : 0, oy0 = any ? box.z : 0;
        const int owx = any ? box.y - box.x + 1 : 0, owy = any ? box.w - box.z + 1 : 0;
        // sub-boxes (solver_kernels.hpp): one if the box fits the tile, else a grid of them, SPL_SUB_MAX wide and
        // high at most, each owning the corners whose patch starts in its first SPL_SUB_MAX - order columns and rows
        const int order = P.cfg.spline_order, T = SPL_SUB_MAX - order;
        int nsx = 1, nsy = 1;
        if(owx*owy + SPL_NEXTRA > SPL_TW)
        {
            nsx = max(1, (owx - order + T - 1)/T);
            nsy = max(1, (owy - order + T - 1)/T);
        }
        const int nsub = nsx*nsy;
        SPL_TICK(ts_bbox)
        if(nsub > SPL_MAXSUB || (nsub > 1 && NPTS > (int)sizeof(own)))
        {
            if(xy0 == 0)
            {
                if(t == 0) plan.spl_hdr[o] = SplHdr{ 0, 0, -1, -1 };
                // (round 5: its rows go through the pre-rounded sums behind this launch, repro_step_*: the same bits
                //  every time. Without their buffers - never, for a problem that can come here - row by row with atomics)
                if(plan.repro.lvl[0] == NULL) spl_rows_fallback(nd, O, r0 + t, r1, Jp, Ji);
            }
            continue;
        }
        if(nsub > 1)
        {
            // whose corner: from the first control point of its x row (an outlier's, outside the box: nobody's)
            for(int c = t; c < NPTS; c += blockDim.x)
            {
                const int rel  = Ji[p00 + 2*c*L + (Ncs ? 2 : 0)] - (m_isi + Ncs);
                const int knot = rel >> 1, px = knot % Nx - ox0, py = knot / Nx - oy0;
                const int gx = px / T, gy = py / T;
                own[c] = (px >= 0 && py >= 0 && gx < nsx && gy < nsy) ? (unsigned char)(gy*nsx + gx) : (unsigned char)255;
            }
            __syncthreads();
        }
#pragma unroll 1
        for(int isub = 0; isub < nsub; isub++)
        {
        const int sgx = isub % nsx, sgy = isub / nsx;
        const int ix0 = ox0 + ((nsub > 1) ? sgx*T : 0), iy0 = oy0 + ((nsub > 1) ? sgy*T : 0);
        const int wx  = (nsub > 1) ? min(T + order, ox0 + owx - ix0) : owx;
        const int wy  = (nsub > 1) ? min(T + order, oy0 + owy - iy0) : owy;
        const int K   = wx*wy;
        // (the first header says how many there are: wy | nsub << 16)
        // (wx with its reciprocal for the gather: wx | spl_magic(wx) << 8)
        const int wx_magic = spl_magic(wx);
        if(t == 0 && xy0 == 0) *spl_hdr_at(plan, o, isub) = SplHdr{ ix0, iy0, wx | (wx_magic << 8), (isub == 0) ? (wy | (nsub << 16)) : wy };
        // a sub-box's pass is over ITS corners only, packed: their rows in the tile, in corner order (a pass over all
        // the corners with the others' rows left zero is as long as the whole observation's: 0.29 ms more a step with
        // 2 x 2 sub-boxes under every board)
        if(nsub > 1)
        {
            __syncthreads();        // (the previous sub-box's passes are through with crow)
            if(wave == 0)
            {
                int nown = 0;
                for(int cb = 0; cb < NPTS; cb += 64)
                {
                    const int  c    = cb + lane;
                    const bool mine = c < NPTS && own[c] == isub;
                    const unsigned long long mm = __ballot(mine);
                    if(c < NPTS) crow[c] = mine ? (unsigned short)(nown + __popcll(mm & ((1ull << lane) - 1ull))) : (unsigned short)0xffff;
                    nown += __popcll(mm);
                }
                if(lane == 0) s_nown = nown;
            }
            __syncthreads();
        }
        const int nrows = (nsub > 1) ? s_nown : NPTS;   // the pass's rows
        const int NC = K + SPL_NEXTRA;                  // local columns in use
        const int NS = (NC + 15) >> 4;                  // 16-column tiles in use
        const int LD = 16*(NS + 1 - (NS & 1));          // row stride: an odd number of tiles
        const int rows_cap = min((NPTS + 3) & ~3, (SPL_LDS_DOUBLES / LD) & ~3);
        const int lx = K + SPL_NDENSE + 6;              // the x column
        const int fr0 = K + SPL_NDENSE;                 // the first frame column
        // this wave's tile rows
        // (every second workgroup deals the rows the other way round: two workgroups share a CU, and their waves w a SIMD)
        const int wv = ((((int)blockIdx.x >> 8) ^ (int)blockIdx.x) & 1) ? 3 - wave : wave;
        const int Ia = NS - 1 - wv, Ib = wv - (8 - NS);
        const int na = (Ia >= 0) ? Ia + 1 : 0, nb = (Ib >= 0 && Ia >= 0) ? Ib + 1 : 0;
        const unsigned L_magic = (unsigned)((0x100000000ull + (unsigned)L - 1)/(unsigned)L);      // e / L = e L_magic >> 32, e (L-1) < 2^32

        {
            const int xy = xy0;
            // state index -> local column of this pass
            auto local_of = [&](int col) -> int
            {
                if(P.do_optimize_frames && col >= nd.E_state0 && col < nd.E_state0 + nd.NE) return fr0 + (col - (nd.E_state0 + 6*f));
                if(m_isi >= 0 && col >= m_isi && col < m_isi + P.Nintr_state)
                {
                    const int rel = col - m_isi;
                    if(rel < Ncs) return K + rel;
                    const int knot = (rel - Ncs) >> 1;
                    return (knot / Nx - iy0)*wx + (knot % Nx - ix0);
                }
                if(m_ise >= 0 && col >= m_ise && col < m_ise + 6)
                    return K + 4 + (col - m_ise);
                return K + 10 + (col - P.i_state_warp);
            };
            // A board's rows mostly fit the tile at once. When they do not, every chunk of rows makes its own Gram and
            // ADDS it to what is staged (and to F): no accumulator is live while rows are fetched - held across the
            // loop they were spilled on every path, and a reload from scratch waits for every store in flight
            double* __restrict__ G = spl_slot(plan, o, xy, isub);
#pragma unroll 1
            for(int c0 = 0; c0 < max(nrows, 1); c0 += rows_cap)
            {
                constexpr bool first = true;
                spl_d4 acc[9];
                // (the lane's indices made opaque at the head of each phase: what is computed from them is computed in
                //  the phase, not in front of the loop over the observations and carried - spilled - through everything)
                int tq = t;
                asm volatile("" : "+v"(tq));
                const int nr = min(rows_cap, nrows - c0), nr4 = (nr + 3) & ~3;
                // (one box: the entries of the corners c0 .. c0 + nr; sub-boxes: of all the corners, kept if the corner's
                //  packed row is one of this chunk's)
                const int pbase = p00 + ((nsub == 1 ? 2*c0 : 0) + xy)*L;
                const int ne = (nsub == 1 ? nr : NPTS)*L;
                // Six entries per thread asked for together (one entry at a time, its column only when the value is
                // not zero, is two memory round trips per entry), the first six BEFORE the tile is cleared: their
                // trip to memory and the clearing overlap
                // (nothing is done with what a load returns before the batch is put away: a select on the spot is a wait on the spot)
                constexpr int EB = 4;
                double v0[EB], v1[EB]; int ci0[EB], ci1[EB], ii0[EB], ii1[EB];
                auto ask = [&](int e0, double (&v)[EB], int (&ci)[EB], int (&ii)[EB])
                {
#pragma unroll
                    for(int u = 0; u < EB; u++)
                    {
                        const int e = e0 + 256*u + tq;
                        const bool ok = e < ne;
                        const int i = ok ? (int)__umulhi((unsigned)e, L_magic) : 0, k = ok ? e - i*L : 0;
                        const int p = pbase + 2*i*L + k;        // (a valid entry either way)
                        ii[u] = ok ? i : -1;
                        v[u]  = Jv[p];
                        ci[u] = Ji[p];
                    }
                };
                auto put = [&](const double (&v)[EB], const int (&ci)[EB], const int (&ii)[EB])
                {
#pragma unroll
                    for(int u = 0; u < EB; u++)
                    {
                        if(ii[u] < 0 || v[u] == 0.0) continue;
                        int row = ii[u];
                        if(nsub > 1) { row = (int)crow[ii[u]] - c0; if(row < 0 || row >= nr) continue; }
                        Jd[row*LD + local_of(ci[u])] = v[u];
                    }
                };
                ask(0, v0, ci0, ii0);
                const double xv = x[r0 + 2*(((nsub == 1) ? c0 : 0) + max(0, min(tq, ((nsub == 1) ? nr : NPTS) - 1))) + xy];
                for(int i = tq; i < nr4*(LD/2); i += blockDim.x) ((double2*)Jd)[i] = make_double2(0.0, 0.0);
                spl_lds_barrier();
                SPL_TICK(ts_zero)
                for(int e0 = 0; e0 < ne; e0 += 2*EB*256)
                {
                    if(e0 + EB*256 < ne)   ask(e0 + EB*256, v1, ci1, ii1);
                    put(v0, ci0, ii0);
                    if(e0 + EB*256 >= ne)  break;
                    if(e0 + 2*EB*256 < ne) ask(e0 + 2*EB*256, v0, ci0, ii0);
                    put(v1, ci1, ii1);
                }
                if(nsub == 1)
                {
                    if(tq < nr) Jd[tq*LD + lx] = xv;
                    for(int i = tq + blockDim.x; i < nr; i += blockDim.x) Jd[i*LD + lx] = x[r0 + 2*(c0 + i) + xy];
                }
                else
                    for(int i = tq; i < NPTS; i += blockDim.x)
                    {
                        const int row = (int)crow[i] - c0;
                        if(row >= 0 && row < nr) Jd[row*LD + lx] = (i == tq) ? xv : x[r0 + 2*i + xy];
                    }
                spl_lds_barrier();
                SPL_TICK(ts_scatter)
                int r16g = r16, kqg = kq;
                asm volatile("" : "+v"(r16g), "+v"(kqg));
                switch(na)
                {
                case 0: if(first) { for(int u = 0; u < 9; u++) acc[u] = spl_d4{0.0, 0.0, 0.0, 0.0}; } break;
                case 1: spl_gram_mfma<1, first>(Jd, LD, nr4 >> 2, r16g, kqg, 16*Ia, 16*Ib, nb, acc); break;
                case 2: spl_gram_mfma<2, first>(Jd, LD, nr4 >> 2, r16g, kqg, 16*Ia, 16*Ib, nb, acc); break;
                case 3: spl_gram_mfma<3, first>(Jd, LD, nr4 >> 2, r16g, kqg, 16*Ia, 16*Ib, nb, acc); break;
                case 4: spl_gram_mfma<4, first>(Jd, LD, nr4 >> 2, r16g, kqg, 16*Ia, 16*Ib, nb, acc); break;
                case 5: spl_gram_mfma<5, first>(Jd, LD, nr4 >> 2, r16g, kqg, 16*Ia, 16*Ib, nb, acc); break;
                case 6: spl_gram_mfma<6, first>(Jd, LD, nr4 >> 2, r16g, kqg, 16*Ia, 16*Ib, nb, acc); break;
                case 7: spl_gram_mfma<7, first>(Jd, LD, nr4 >> 2, r16g, kqg, 16*Ia, 16*Ib, nb, acc); break;
                default:spl_gram_mfma<8, first>(Jd, LD, nr4 >> 2, r16g, kqg, 16*Ia, 16*Ib, nb, acc); break;
                }
                SPL_TICK(ts_gram)
                // out: the whole lower triangle is staged - the camera-block rows and the x row for the gather - with
                // stores nobody here waits for; the six frame rows and the x row also go to F, for the second half of this
                // (the lane's coordinates made opaque here: left alone, the compiler computes the 36 store addresses once,
                //  in front of the loop over the observations, and keeps them - the kernel spills)
                int kq_o = kq, r16_o = r16;
                asm volatile("" : "+v"(kq_o), "+v"(r16_o));
                const bool add = c0 > 0;
                auto out_tile = [&](const spl_d4& a4, int I, int J)
                {
                    const int col = 16*J + r16_o;
#pragma unroll
                    for(int vv = 0; vv < 4; vv++)
                    {
                        const int row = 16*I + kq_o + 4*vv;
                        if(row < NC && col <= row)
                        {
                            double* __restrict__ g = &G[spl_tri(row) + col];
                            *g = add ? *g + a4[vv] : a4[vv];
                            if(row >= fr0)
                            {
                                double* __restrict__ ff = &F[(row - fr0)*SPL_TW + col];
                                *ff = add ? *ff + a4[vv] : a4[vv];
                            }
                        }
                    }
                };
                if(!add)
                {
#pragma unroll
                    for(int u = 0; u < 9; u++)
                    {
                        if(u < na)           out_tile(acc[u], Ia, u);
                        else if(u - na < nb) out_tile(acc[u], Ib, u - na);
                    }
                }
                else
                {
#pragma unroll
                    for(int u = 0; u < 9; u++)
                    {
                        if(u < na)           out_tile(acc[u], Ia, u);
                        else if(u - na < nb) out_tile(acc[u], Ib, u - na);
                    }
                }
                // (the tile is cleared again only after everybody is through with it)
                if(c0 + rows_cap < nrows) spl_lds_barrier();
            }
            spl_lds_barrier();
            // what belongs to the frame: rows K+12 .. K+17 against the camera-block columns (Bt) and against each
            // other (D_f); the x row against the frame columns (g_f). Element (ia, col) is the same thread's in every
            // pass and observation: its read-modify-writes of one address follow each other in program order
            int tf = t;
            asm volatile("" : "+v"(tf));
            if(P.do_optimize_frames)
                for(int e = tf; e < 7*SPL_TW; e += blockDim.x)
                {
                    const int ia = e >> 7, col = e & (SPL_TW - 1);      // ia 6: the x row
                    const int row = fr0 + ia;
                    const int ib = col - fr0;
                    if(col > row || (ia == 6 && (ib < 0 || ib >= 6))) continue;
                    const double vv = F[ia*SPL_TW + col];
                    if(vv == 0.0) continue;
                    if(ib >= 0)        FD[ia*6 + ib] += vv;                     // (ia 6: g_f)
                    else if(col >= K)  FB[ia*SPL_NDENSE + (col - K)] += vv;
                    else
                    {
                        const int cs = spl_col_state(P, nd, col, K, ix0, iy0, wx > 0 ? wx : 1, xy, m_isi, m_ise, wx_magic);
                        if(cs < 0) continue;
                        // a knot's column is written by this pass of this observation and by nobody else - or, cut
                        // into sub-boxes, by the passes of those that hold it, one after the other (the barrier below)
                        double* __restrict__ dst = &O.Bt[(size_t)(6*f + ia)*nd.Nc + state_to_SE(nd, cs)];
                        if(nsub == 1) *dst = vv; else *dst += vv;
                    }
                }
            if(nsub > 1) __syncthreads();
            // (F is written again after the next pass's barriers)
            SPL_TICK(ts_out)
        }
        }   // isub
        // the observation's core and extrinsics columns of Bt: one addition each, nothing read back (an atomic one, as
        // below). The warp's columns wait for the frame's last observation
        spl_lds_barrier();
        if(P.do_optimize_frames && t < 6*SPL_NDENSE)
        {
            const int ia = t / SPL_NDENSE, d = t - ia*SPL_NDENSE;
            if(d < 10 && FB[t] != 0.0)
            {
                const int cs = spl_col_state(P, nd, d, 0, 0, 0, 1, 0, m_isi, m_ise, 0);   // (a column past the control points: their box does not matter)
                if(cs >= 0) atomicAdd(&O.Bt[(size_t)(6*f + ia)*nd.Nc + state_to_SE(nd, cs)], FB[t]);
                FB[t] = 0.0;
            }
        }
    }
    // the frame's block and gradient: one addition each (an atomic one: an observation that went row by row - too many
    // knots - adds to the same entries with atomics, possibly still in flight)
    __syncthreads();
    if(P.do_optimize_frames && t < 6*SPL_NDENSE)
    {
        const int ia = t / SPL_NDENSE, d = t - ia*SPL_NDENSE;
        if(d >= 10 && nd.Nwarp && FB[t] != 0.0) atomicAdd(&O.Bt[(size_t)(6*f + ia)*nd.Nc + state_to_SE(nd, nd.i_state_warp + (d - 10))], FB[t]);
    }
    if(P.do_optimize_frames && t < 42)
    {
        const int ia = t / 6, ib = t - 6*ia;
        if(ia == 6) { if(FD[t] != 0.0) atomicAdd(&O.g[nd.E_state0 + 6*f + ib], FD[t]); }
        else
        {
            const double vv = (ib <= ia) ? FD[ia*6 + ib] : FD[ib*6 + ia];
            if(vv != 0.0) atomicAdd(&O.D[(size_t)f*36 + ia*6 + ib], vv);
        }
    }
#ifdef SPL_TS
    if((f == 0 || f == 400 || f == 799) && (t == 0 || t == 255))
        printf("splined assembly f %d t %d: bbox %lld zero %lld scatter %lld gram %lld out %lld cycles\n", f, t, ts_bbox, ts_zero, ts_scatter, ts_gram, ts_out);
    // (the workgroup's place in time: the constant 100 MHz clock at its start and end, and its own cycles)
    if((f % 100 == 0 || f == 255 || f == 256 || f == 511 || f == 512 || f == 799) && t == 0)
        printf("splined assembly f %d: wall %lld .. %lld (x10 ns), %lld cycles\n", f, tw0, wall_clock64(), clock64() - tc0);
#endif
}

#define SPLG_BATCH 3      // passes whose loads are in flight together (4: 104 registers with the sub-boxes' loop, and two of these
                          // workgroups and a SYRK workgroup no longer share a CU's registers: the launch beside the SYRK 122 us instead of 86)
__global__ __launch_bounds__(64*SPLG_WAVES)
// window > 0 (a launch of the knots' rows alone): a wave's copy of the row is its last window+1 columns and the
// camera's core - a knot's row of A holds nothing else: two control points meet in a Gram only if some corner's
// (order+1)^2 patch holds both, so only within `order` knots of each other either way, and the lower triangle is the
// part at or before the row: window = 2 (order Nx + order) columns. (The whole row, 1207 doubles a wave, was 77 KB of
// LDS a workgroup - two workgroups a CU, 12 of its 20 us clearing and adding up zeros.) block0: the first row's number
__global__ __launch_bounds__(64*SPLG_WAVES)
void assemble_splined_gather_kernel(DeviceProblem P, NormalDims nd, OpRef R, AssemblyPlan plan, int nwaves, int block0, int window)
{
    if(opref_skip(R)) return;
    extern __shared__ double accs[];        // [nwaves][stride]
    const OpDev& O = opref_get(R);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int stride = (window > 0) ? window + 1 + 4 : nd.Nc + 1;
    const int Nx = P.cfg.spline_Nx, Ncs = P.Ncore_state;
    const int nknot = splg_nknotrows(P), nintr = P.Nintr_state > 0 ? P.Ncameras_intrinsics*P.Nintr_state : 0;
    const int blk = block0 + (int)blockIdx.x;
    // the row, and the observations this workgroup walks
    int r, obs0, obs1, dense = -1, chunk = 0;
    if(blk < nknot)
    {
        const int per = P.Nintr_state - Ncs, ic = blk / per;
        r = ic*P.Nintr_state + Ncs + (blk - ic*per);
        obs0 = 0; obs1 = P.Nobs_board;
    }
    else
    {
        dense = (blk - nknot) / SPLG_E; chunk = (blk - nknot) % SPLG_E;
        const int ncore = P.Nintr_state > 0 ? P.Ncameras_intrinsics*Ncs : 0;
        r = (dense < ncore) ? (dense / Ncs)*P.Nintr_state + dense % Ncs : nintr + (dense - ncore);
        const int per = (P.Nobs_board + SPLG_E - 1)/SPLG_E;
        obs0 = min(P.Nobs_board, chunk*per); obs1 = min(P.Nobs_board, obs0 + per);
    }
    for(int i = t; i < nwaves*stride; i += blockDim.x) accs[i] = 0.0;
    __syncthreads();
    const bool xrow = (r == nd.Nc);
    const int  sr   = xrow ? -1 : (S_to_state(nd, r));     // state index of the row
#ifdef SPLG_TS
    long long gts0 = clock64(), gts_match = 0, gtq; int gn_match = 0, gn_batch = 0;
#endif
    if(wave < nwaves)
    {
        double* __restrict__ acc = accs + wave*stride;
        const int per = (obs1 - obs0 + nwaves - 1)/nwaves;
        const int w0 = min(obs1, obs0 + wave*per), w1 = min(obs1, w0 + per);
        for(int ob = w0; ob < w1; ob += 64)
        {
            // lane l: does observation ob + l hold the row, and where (local row of the x pass, of the y pass)
            // (an observation cut into sub-boxes has a pass, a header and a staged triangle for each: first the first
            //  sub-box of the 64 observations, then the second of those that have one, ...)
            const int o = ob + lane;
            int isi = -1, ise = -1, nsub_l = 0;
            SplHdr h0 = { 0, 0, -1, -1 };
            if(o < w1)
            {
                h0 = plan.spl_hdr[o];
                isi = P.board_meta[o].i_state_intrinsics; ise = P.board_meta[o].i_state_extrinsics;
                if(h0.wx >= 0) { nsub_l = h0.wy >> 16; h0.wy &= 0xffff; }
            }
            auto passes_of = [&](const int isub, const SplHdr hraw) __attribute__((always_inline))
            {
            int lr0 = -1, lr1 = -1, kind = 2;       // kind 0: a knot's row, 1: a core row, 2: the others
            SplHdr h = hraw;
            int hmagic = 0;
            if(hraw.wx >= 0) { h.wx = hraw.wx & 0xff; hmagic = hraw.wx >> 8; }
            if(isub < nsub_l)
            {
                if(hraw.wx >= 0)
                {
                    const int K = h.wx*h.wy;
                    if(xrow) lr0 = lr1 = K + SPL_NDENSE + 6;
                    else if(r >= nd.S_split) lr0 = lr1 = K + 10 + (r - nd.S_split);       // (a warp term: the splined models keep the stationary partition)
                    else if(ise >= 0 && sr >= ise && sr < ise + 6) lr0 = lr1 = K + 4 + (sr - ise);
                    else if(isi >= 0 && sr >= isi && sr < isi + P.Nintr_state)
                    {
                        const int rel = sr - isi;
                        if(rel < Ncs) { lr0 = lr1 = K + rel; kind = 1; }
                        else
                        {
                            const int knot = (rel - Ncs) >> 1;
                            const int ax = knot % Nx - h.ix0, ay = knot / Nx - h.iy0;
                            kind = 0;
                            if(ax >= 0 && ax < h.wx && ay >= 0 && ay < h.wy)
                            {
                                if((rel - Ncs) & 1) lr1 = ay*h.wx + ax; else lr0 = ay*h.wx + ax;
                            }
                        }
                    }
                }
            }
            unsigned long long mm = __ballot(lr0 >= 0 || lr1 >= 0);
#ifdef SPLG_TS
            gn_match += __popcll(mm); gtq = clock64();
#endif
            while(mm)
            {
#ifdef SPLG_TS
                gn_batch++;
#endif
                // up to SPLG_BATCH observations: every load first, then the sums in order
                double v[SPLG_BATCH][2][2], vc[SPLG_BATCH][2];
                int    cs[SPLG_BATCH][2], csc[SPLG_BATCH], Kk[SPLG_BATCH];
                bool   on[SPLG_BATCH][2];
                // (nothing is done with what a load returns before all of a batch's loads are out - a select on the spot is a
                //  wait on the spot, and the 18 loads of three matches were most of 18 trips to memory: 20k cycles a batch.
                //  Which of the values count: a bit each)
                unsigned wanted = 0u;         // (six bits a match)
#pragma unroll
                for(int k = 0; k < SPLG_BATCH; k++)
                {
                    const bool have = mm != 0ull;
                    const int src = __builtin_amdgcn_readfirstlane(have ? __ffsll((long long)mm) - 1 : 0);
                    if(have) mm &= mm - 1;
                    const int oo  = ob + src;
                    // (one lane's values for the wave: v_readlane, not nine trips through the LDS crossbar)
                    const int ix0 = __builtin_amdgcn_readlane(h.ix0, src), iy0 = __builtin_amdgcn_readlane(h.iy0, src);
                    const int wx  = __builtin_amdgcn_readlane(h.wx, src),  wy  = __builtin_amdgcn_readlane(h.wy, src);
                    const int wxm = __builtin_amdgcn_readlane(hmagic, src);
                    const int si  = __builtin_amdgcn_readlane(isi, src), se = __builtin_amdgcn_readlane(ise, src), kd = __builtin_amdgcn_readlane(kind, src);
                    const int l0  = __builtin_amdgcn_readlane(lr0, src), l1 = __builtin_amdgcn_readlane(lr1, src);
                    const int K   = wx*wy;
                    Kk[k]  = K;
                    csc[k] = (have && kd == 0 && lane < Ncs && si >= 0) ? si + lane : -1;
#pragma unroll
                    for(int it = 0; it < 2; it++)
                    {
                        const int lc = lane + 64*it;
                        // the column's variable: the same in both passes but for the surface of a knot
                        int c = spl_col_state(P, nd, lc, K, ix0, iy0, wx > 0 ? wx : 1, 0, si, se, wxm);
                        if(kd == 1 && lc < K) c = -1;                      // a core row: the knots are above the diagonal
                        if(xrow && lc == K + SPL_NDENSE + 6) c = -2;        // |x|^2
                        cs[k][it] = have ? c : -1;
                    }
#pragma unroll
                    for(int xy = 0; xy < 2; xy++)
                    {
                        const int lr = xy ? l1 : l0;
                        on[k][xy] = have && lr >= 0;
                        const double* __restrict__ Gp = spl_slot(plan, oo, xy, isub);
#pragma unroll
                        for(int it = 0; it < 2; it++)
                        {
                            const int lc = lane + 64*it;
                            // (always a load, from an address that is always valid: a load under a condition is a branch and a wait)
                            const bool want = on[k][xy] && lc <= lr && cs[k][it] != -1;
                            v[k][xy][it] = Gp[want ? spl_tri(lr) + lc : 0];
                            if(want) wanted |= 1u << (6*k + 2*xy + it);
                        }
                        // a knot's row against the core: below it in local order
                        {
                            const bool want = on[k][xy] && csc[k] >= 0;
                            vc[k][xy] = Gp[want ? spl_tri(K + lane) + lr : 0];
                            if(want) wanted |= 1u << (6*k + 4 + xy);
                        }
                    }
                }
#pragma unroll
                for(int k = 0; k < SPLG_BATCH; k++)
#pragma unroll
                    for(int xy = 0; xy < 2; xy++)
                    {
                        if(!on[k][xy]) continue;
#pragma unroll
                        for(int it = 0; it < 2; it++)
                        {
                            const int c = cs[k][it];
                            const double vv = ((wanted >> (6*k + 2*xy + it)) & 1u) ? v[k][xy][it] : 0.0;
                            if(c == -1 || vv == 0.0) continue;
                            if(c == -2) { acc[nd.Nc] += vv; continue; }
                            const int se = state_to_SE(nd, c + ((lane + 64*it < Kk[k]) ? xy : 0));
                            if(window <= 0) acc[se] += vv;
                            else
                            {
                                // (a control point further away than a patch reaches: a structural zero that is not one)
                                const int pw = se - (r - window);
                                if(pw >= 0) acc[pw] += vv; else O.scalars[SC_BAD_STRUCTURE] = 1.0;
                            }
                        }
                        const double vcc = ((wanted >> (6*k + 4 + xy)) & 1u) ? vc[k][xy] : 0.0;
                        if(csc[k] >= 0 && vcc != 0.0)
                        {
                            if(window <= 0) acc[state_to_SE(nd, csc[k])] += vcc;
                            else            acc[window + 1 + lane] += vcc;        // (csc = the camera's core + lane)
                        }
                    }
            }
#ifdef SPLG_TS
            gts_match += clock64() - gtq;
#endif
            };
            // (the first sub-box - almost always the only one - exactly as before there were any)
            passes_of(0, h0);
#ifndef SPLG_NO_SUB
            if(__any(nsub_l > 1))
                for(int isub = 1; __any(isub < nsub_l); isub++)
                {
                    SplHdr h = { 0, 0, -1, -1 };
                    if(isub < nsub_l) h = *spl_hdr_at(plan, o, isub);
                    passes_of(isub, h);
                }
#endif
        }
    }
#ifdef SPLG_TS
    if(lane == 0 && (wave == 0 || wave == 5) && window > 0 && (blk % 149 == 0))
        printf("gather row %d wave %d: %lld cycles to the barrier, %lld of them in %d matches (%d batches)\n", blk, wave, clock64() - gts0, gts_match, gn_match, gn_batch);
#endif
    __syncthreads();
    // the waves' copies, in wave order
    for(int c = t; c < stride; c += blockDim.x)
    {
        double s = 0.0;
        for(int w = 0; w < nwaves; w++) s += accs[w*stride + c];
        if(dense >= 0) { plan.spl_part[((size_t)dense*SPLG_E + chunk)*stride + c] = s; continue; }
        int col = c;
        if(window > 0)
        {
            const int per = P.Nintr_state - Ncs, ic = blk / per;
            col = (c <= window) ? r - window + c : ic*P.Nintr_state + (c - window - 1);
        }
        if(s != 0.0 && col >= 0 && col <= r) O.A[(size_t)r*nd.Nc + col] += s;
    }
}
// The rows of the camera block that are control points, the other way round (round 4; the kernel above still takes the
// rows every pass holds). A control point's row of A's lower triangle has (order)(2 order + 1) + order + 1 places that
// can hold anything - the control points of its surface up to `order` back in either direction, 25 for order 3 - and the
// camera's core: one LANE per place, and a pass that holds the row is ONE load for the wave (half a wave: lanes 32..63
// take the next pass): G[tri(lr) + lr + dy wx + dx]. No LDS copy of the row, no search for what a column is, nothing
// read that is a structural zero; each lane adds its place's entries in a fixed order (its half's passes in order, the
// two halves, then the waves in order), the same bits every time. A wave scans 64 observations' headers at a time and
// leaves those that hold the row in an LDS list; SPLK_INFLIGHT entries a half are asked for together.
// (The row-per-workgroup gather above spent 18-20k cycles on a batch of three passes - 128 local columns looked up,
//  loaded and added through LDS for the 25 that count: 79 us at BASELINE configuration 2, 108 on a real calibration)
#define SPLK_INFLIGHT 4
struct SplkMatch { int slot; int lr; int wx; int ax_wy; };  // the pass's staged triangle (index of its slot), the row, the box (wx | which allocation << 16; ax | wy << 16)
__global__ __launch_bounds__(64*SPLK_WAVES)
void assemble_splined_gather_knots_kernel(DeviceProblem P, NormalDims nd, OpRef R, AssemblyPlan plan)
{
    if(opref_skip(R)) return;
    __shared__ SplkMatch list[SPLK_WAVES][64];
    __shared__ double    part[SPLK_WAVES][32];
    const OpDev& O = opref_get(R);
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = lane >> 5, j = lane & 31;
    const int Nx = P.cfg.spline_Nx, Ncs = P.Ncore_state, order = P.cfg.spline_order;
    // the row: control point (kx, ky) of surface s of camera ic
    const int blk = blockIdx.x;
    const int per = P.Nintr_state - Ncs, ic = blk / per, rel = blk - ic*per;
    const int r   = ic*P.Nintr_state + Ncs + rel;                   // its index in the camera block
    const int sr  = S_to_state(nd, r);
    const int knot = rel >> 1, s = rel & 1, kx = knot % Nx, ky = knot / Nx;
    // this lane's place: (dx, dy), dy < 0 any dx, dy == 0 dx <= 0; or a core variable; or none
    const int ndx = 2*order + 1, nback = order*ndx, nplaces = nback + order + 1;
    int dx = 0, dy = 0, core = -1;
    bool place = false;
    if(j < nback)          { dy = -order + j / ndx; dx = -order + j % ndx; place = true; }
    else if(j < nplaces)   { dy = 0; dx = -order + (j - nback); place = true; }
    else if(j < nplaces + Ncs) core = j - nplaces;
    place = place && kx + dx >= 0 && kx + dx < Nx && ky + dy >= 0;
    double acc = 0.0;

    const int Nobs = P.Nobs_board;
    const int per_wave = (Nobs + SPLK_WAVES - 1)/SPLK_WAVES;
    const int w0 = min(Nobs, wave*per_wave), w1 = min(Nobs, w0 + per_wave);
    SplkMatch* __restrict__ mine = list[wave];
    for(int ob = w0; ob < w1; ob += 64)
    {
        const int o = ob + lane;
        int nsub_l = 0;
        SplHdr h0 = { 0, 0, -1, -1 };
        bool cam = false;
        if(o < w1)
        {
            h0 = plan.spl_hdr[o];
            const int isi = P.board_meta[o].i_state_intrinsics;
            cam = isi >= 0 && sr >= isi && sr < isi + P.Nintr_state;
            if(h0.wx >= 0) { nsub_l = h0.wy >> 16; h0.wy &= 0xffff; }
            if(!cam) nsub_l = 0;
        }
        for(int isub = 0; __any(isub < nsub_l); isub++)
        {
            SplHdr h = h0;
            if(isub > 0 && isub < nsub_l) h = *spl_hdr_at(plan, o, isub);
            bool hit = false;
            int  lr = 0, wx = 1, wy = 1;
            if(isub < nsub_l && h.wx >= 0)
            {
                wx = h.wx & 0xff; wy = h.wy;
                const int ax = kx - h.ix0, ay = ky - h.iy0;
                hit = ax >= 0 && ax < wx && ay >= 0 && ay < wy;
                lr  = ay*wx + ax;
            }
            // the passes that hold the row, in observation order, into the list
            const unsigned long long mm = __ballot(hit);
            const int n = __popcll(mm);
            if(hit)
            {
                const int at = __popcll(mm & ((1ull << lane) - 1ull));
                const double* __restrict__ G = spl_slot(plan, o, s, isub);
                mine[at] = SplkMatch{ (int)((G - ((isub > 0) ? plan.chunk_extra : plan.chunk_part))/SPL_TRI), lr, wx | ((isub > 0) ? 0x10000 : 0), (kx - h.ix0) | (wy << 16) };
            }
            // (the list is this wave's own: no barrier, the LDS keeps a wave's accesses in order)
            for(int i0 = 0; i0 < n; i0 += 2*SPLK_INFLIGHT)
            {
                double v[SPLK_INFLIGHT];
                bool   on[SPLK_INFLIGHT];
#pragma unroll
                for(int b = 0; b < SPLK_INFLIGHT; b++)
                {
                    const int i = i0 + 2*b + half;
                    on[b] = false; v[b] = 0.0;
                    if(i < n)
                    {
                        const SplkMatch m = mine[i];
                        const int mwx = m.wx & 0xffff;
                        // (a second, third.. sub-box's triangle is in the other allocation: spl_slot())
                        const double* __restrict__ G = ((m.wx & 0x10000) ? plan.chunk_extra : plan.chunk_part) + (size_t)m.slot*SPL_TRI;
                        const int ax = m.ax_wy & 0xffff;
                        if(place && ax + dx >= 0 && ax + dx < mwx && m.lr + dy*mwx >= 0)
                        {
                            on[b] = true;
                            v[b]  = G[spl_tri(m.lr) + m.lr + dy*mwx + dx];
                        }
                        else if(core >= 0)
                        {
                            on[b] = true;
                            v[b]  = G[spl_tri(mwx*(m.ax_wy >> 16) + core) + m.lr];
                        }
                    }
                }
#pragma unroll
                for(int b = 0; b < SPLK_INFLIGHT; b++) if(on[b]) acc += v[b];
            }
        }
    }
    // the halves, then the waves, in order
    acc += __shfl(acc, (lane + 32) & 63);
    if(lane < 32) part[wave][lane] = acc;
    __syncthreads();
    if(t < 32)
    {
        double total = 0.0;
        for(int w = 0; w < SPLK_WAVES; w++) total += part[w][t];
        if(total != 0.0)
        {
            const int col = (core >= 0) ? ic*P.Nintr_state + core : r + 2*(dy*Nx + dx);
            if((place || core >= 0) && col >= 0 && col <= r) O.A[(size_t)r*nd.Nc + col] += total;
        }
    }
}
// The regularization rows of a splined model (regularization_splined_kernel in kernels.hip): per knot a radial
// and a tangential row on the knot's two variables, then one row per centre-pixel variable, then unity_cam01.
// Rows 2 i and 2 i + 1 share their columns; no two PAIRS do. One lane per pair, the pair's rows one after the
// other, plain adds: the same bits every time. (Row by row with atomics, the two rows of a knot race.) |x|^2 of a
// workgroup's rows goes to row_part[blockIdx.x]; the combine kernel adds those in order. Lower triangle of A only
__device__ __forceinline__
void rows_pairs_body(const NormalDims& nd, const OpDev& O, int row0, int row1,
                     const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji, double* __restrict__ row_part, int block)
{
    const double* __restrict__ Jv = O.Jv;
    const int i = block*blockDim.x + threadIdx.x;
    double n2 = 0.0;
    for(int k = 0; k < 2; k++)
    {
        const int r = row0 + 2*i + k;
        if(r >= row1) break;
        const double xr = O.x[r];
        n2 += xr*xr;
        const int p0 = Jp[r], p1 = Jp[r+1];
        for(int p = p0; p < p1; p++)
        {
            const int ci = Ji[p];
            const double vi = Jv[p];
            if((unsigned)ci >= (unsigned)nd.Nstate) { O.scalars[SC_BAD_STRUCTURE] = 1.0; continue; }
            const int si = state_to_SE(nd, ci);
            if(si < 0) { O.scalars[SC_BAD_STRUCTURE] = 1.0; continue; }       // a regularization row has camera-block variables only
            O.g[ci] += vi*xr;
            for(int q = p0; q < p1; q++)
            {
                const int cj = Ji[q];
                if((unsigned)cj >= (unsigned)nd.Nstate) continue;
                const int sj = state_to_SE(nd, cj);
                if(sj >= 0 && sj <= si) O.A[(size_t)si*nd.Nc + sj] += vi*Jv[q];
            }
        }
    }
    for(int off = 32; off > 0; off >>= 1) n2 += __shfl_down(n2, off);
    __shared__ double part[4];
    if((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = n2;
    __syncthreads();
    if(threadIdx.x == 0) row_part[block] = (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ __launch_bounds__(256)
void rows_pairs_kernel(NormalDims nd, OpRef R, int row0, int row1,
                       const int32_t* __restrict__ Jp, const int32_t* __restrict__ Ji, double* __restrict__ row_part)
{
    if(opref_skip(R)) return;
    rows_pairs_body(nd, opref_get(R), row0, row1, Jp, Ji, row_part, blockIdx.x);
}
// Which control points does a board cover at this point? (Round 5.) The others have their regularization rows and
// nothing else: a 2 x 2 block of the camera block each, coupled to nothing - and in the Cholesky of the camera block
// every one of their columns is a pivot of the sequential chain all the same: at BASELINE configuration 2 (30 x 20
// control points over 150 degrees, boards 4 m away) 277 of the 600 control points, 554 of 1206 pivots. So the
// camera block is put in the order [coupled variables | isolated pairs] (each part in its own order), the reduction
// writes the coupled part as a dense matrix of its own and the pairs' blocks beside it (schur_reduce_body), and the
// factorization's launches past the coupled part's last panel find nothing to do (lchol_plan()).
// One workgroup: the observations' boxes (OpDev::spl_box, left by board_splined_kernel) marked in LDS, then a scan over
// the camera block's variables. cperm: [Nc] position -> variable | [Nc] variable -> position | [1] the coupled ones
__device__ __forceinline__
void spl_compact_body(const DeviceProblem& P, const NormalDims& nd, const OpDev& O, int* __restrict__ lds_c /* [Nknots_all] used | [SPLC_T/64] wave totals | spl_compact_lds_ints(): the dissection's scratch */,
                      const int* __restrict__ nd_lim /* NdLimits on the device; NULL: no dissection */)
{
    if(O.cperm == NULL) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int Nx = P.cfg.spline_Nx, Ny = P.cfg.spline_Ny, NK = Nx*Ny;
    const int nknots = P.Ncameras_intrinsics*NK;
    int* __restrict__ used = lds_c;
    int* __restrict__ wtot = lds_c + nknots;
    // (the dissection's scratch behind the wave totals: [0] the widest box | [1..4] the plan | [8 ..] covered control points per grid column | 64-bit scan words)
    int* __restrict__ ndw = wtot + SPLC_T/64;
    for(int i = t; i < nknots; i += SPLC_T) used[i] = 0;
    if(t < 8 + Nx && t < SPLC_T) ndw[t] = 0;
    __syncthreads();
    if(P.Ndist_state > 0 && O.spl_box != NULL)
        for(int o = t; o < P.Nobs_board; o += SPLC_T)
        {
            const int4 box = ((const int4*)O.spl_box)[o];
            const int isi = P.board_meta[o].i_state_intrinsics;
            if(box.y < 0 || isi < 0) continue;                       // no inlier under this observation
            atomicMax(&ndw[0], box.y - box.x + 1);
            const int icam = (isi - P.i_state_intrinsics)/P.Nintr_state;
            for(int iy = box.z; iy <= box.w; iy++)
                for(int ix = box.x; ix <= box.y; ix++)
                    used[icam*NK + iy*Nx + ix] = 1;                  // (everybody writes the same 1)
        }
    __syncthreads();
    // the variables, a run of consecutive ones per thread: coupled unless it is a control point's that no box holds
    const int per = (nd.Nc + SPLC_T - 1)/SPLC_T;
    const int c0 = min(nd.Nc, t*per), c1 = min(nd.Nc, c0 + per);
    auto coupled = [&](int c) -> bool
    {
        const int st = S_to_state(nd, c);
        const int rel = st - P.i_state_intrinsics;
        if(P.Ndist_state <= 0 || rel < 0 || rel >= P.Ncameras_intrinsics*P.Nintr_state) return true;
        const int icam = rel / P.Nintr_state, k = rel - icam*P.Nintr_state - P.Ncore_state;
        if(k < 0) return true;                                       // the core
        return used[icam*NK + (k >> 1)] != 0;
    };
    int mine = 0;
    for(int c = c0; c < c1; c++) mine += coupled(c) ? 1 : 0;
    // exclusive scan of `mine` over the threads: within the wave, then the waves' totals
    int incl = mine;
    for(int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if(lane >= off) incl += v; }
    if(lane == 63) wtot[wave] = incl;
    __syncthreads();
    int before = incl - mine, total = 0;
    for(int w = 0; w < SPLC_T/64; w++) { const int v = wtot[w]; if(w < wave) before += v; total += v; }
    // (a camera block with nothing coupled in it cannot be: boards make rows. If it were, nothing is compacted)
    const int n1 = (total > 0) ? total : nd.Nc;
    int* __restrict__ perm = O.cperm;
    int* __restrict__ iperm = O.cperm + nd.Nc;
    int at_c = before, at_i = n1 + (c0 - before);
    for(int c = c0; c < c1; c++)
    {
        const bool cpl = (total > 0) ? coupled(c) : true;
        const int pos = cpl ? at_c++ : at_i++;
        perm[pos] = c; iperm[c] = pos;
    }
    if(t == 0) O.cperm[2*nd.Nc] = n1;

    // ---- the dissection (lchol_nd_*): a strip of grid columns as wide as the widest box less one; A the covered control
    // points left of it, B those right of it, S the strip's and every coupled variable that is no control point
    if(O.ndp == NULL) return;
    int* __restrict__ ndh   = O.ndp;
    int* __restrict__ npos  = ndh + NDH_WORDS;
    int* __restrict__ nperm = npos + nd.Nc;
    int* __restrict__ colcnt = ndw + 8;
    const bool eligible = nd_lim != NULL && P.Ncameras_intrinsics == 1 && total > 0 && P.Ndist_state > 0 && Nx + 8 <= SPLC_T;
    if(eligible && t < Nx)
    {
        int k = 0;
        for(int iy = 0; iy < Ny; iy++) k += used[iy*Nx + t];
        colcnt[t] = k;
    }
    __syncthreads();
    if(t == 0)
    {
        // the best strip there is (what learn_likely_size() provides launches for), and the best one that fits what was provided
        const int std_cost = (n1 + ND_PANEL - 1)/ND_PANEL;
        int ideal = 0, idA = 0, idB = 0, idS = n1, idcost = std_cost;
        int act = 0, bestA = 0, bestB = 0, bestS = n1, bestc0 = 0, bestcost = std_cost, ws = 0;
        if(eligible && ndw[0] >= 2)
        {
            ws = ndw[0] - 1;
            int left = 0, all = 0;
            for(int x = 0; x < Nx; x++) all += colcnt[x];
            int strip = 0;
            for(int x = 0; x < ws && x < Nx; x++) strip += colcnt[x];       // the strip [s0, s0 + ws), s0 = 0 to begin with
            for(int s0 = 0; s0 + ws < Nx; s0++)
            {
                if(s0 > 0) { left += colcnt[s0 - 1]; strip += colcnt[s0 + ws - 1] - colcnt[s0 - 1]; }
                const int ar = 2*left, br = 2*(all - left - strip), sr = n1 - ar - br;
                const int a = (ar + ND_PANEL - 1)/ND_PANEL, b = (br + ND_PANEL - 1)/ND_PANEL, s = (sr + ND_PANEL - 1)/ND_PANEL;
                if(a < 1 || b < 1 || sr < 1) continue;
                // launches on the chain: the rounds of the longer side, the junction, the separator's panels
                const int cost = max(a, b) + 1 + s;
                if(cost < idcost) { idcost = cost; idA = ar; idB = br; idS = sr; ideal = 1; }
                const bool fits = nd_lim[0] > 0 && a <= nd_lim[0] && b <= nd_lim[0] && sr <= nd_lim[1] && ND_PANEL*max(a, b) <= LCH_ND_WMAX;
                if(fits && cost < bestcost) { bestcost = cost; bestA = ar; bestB = br; bestS = sr; bestc0 = s0; act = 1; }
            }
        }
        ndh[NDH_IDEAL_A] = ideal ? idA : 0; ndh[NDH_IDEAL_B] = ideal ? idB : 0; ndh[NDH_IDEAL_NS] = ideal ? idS : 0;
        const int a = (bestA + ND_PANEL - 1)/ND_PANEL, b = (bestB + ND_PANEL - 1)/ND_PANEL;
        ndh[NDH_ACTIVE] = act;
        ndh[NDH_NA] = act ? ND_PANEL*a : 0; ndh[NDH_NB] = act ? ND_PANEL*b : 0; ndh[NDH_NS] = act ? bestS : n1;
        ndh[NDH_NSEFF] = act ? bestS : n1;
        ndh[NDH_ARAW] = act ? bestA : 0; ndh[NDH_BRAW] = act ? bestB : 0;
        ndw[1] = act; ndw[2] = bestc0; ndw[3] = ws; ndw[4] = bestA; ndw[5] = bestB;
    }
    __syncthreads();
    if(!ndw[1]) return;
    {
        const int sc0 = ndw[2], sws = ndw[3], arw = ndw[4], brw = ndw[5];
        const int nA = ndh[NDH_NA], nB = ndh[NDH_NB];
        // class of a coupled variable: 1 A, 2 B, 0 S
        auto cls_of = [&](int c) -> int
        {
            const int st = S_to_state(nd, c);
            const int rel = st - P.i_state_intrinsics;
            if(rel < 0 || rel >= P.Nintr_state) return 0;
            const int k = rel - P.Ncore_state;
            if(k < 0) return 0;
            const int x = (k >> 1) % Nx;
            return (x < sc0) ? 1 : ((x >= sc0 + sws) ? 2 : 0);
        };
        // exclusive scans of the three classes' counts over the threads' runs, in one 64-bit word (21 bits a count)
        unsigned long long mine3 = 0ull;
        for(int c = c0; c < c1; c++)
            if(coupled(c)) mine3 += 1ull << (21*cls_of(c));
        unsigned long long incl3 = mine3;
        for(int off = 1; off < 64; off <<= 1) { const unsigned long long v = __shfl_up(incl3, off); if(lane >= off) incl3 += v; }
        unsigned long long* __restrict__ wt3 = (unsigned long long*)(((size_t)(colcnt + Nx) + 7) & ~(size_t)7);
        __syncthreads();
        if(lane == 63) wt3[wave] = incl3;
        __syncthreads();
        unsigned long long before3 = incl3 - mine3;
        for(int w = 0; w < wave; w++) before3 += wt3[w];
        int at[3] = { (int)(before3 & 0x1fffff), (int)((before3 >> 21) & 0x1fffff), (int)((before3 >> 42) & 0x1fffff) };
        const int base[3] = { nA + nB, 0, nA };
        for(int c = c0; c < c1; c++)
        {
            if(!coupled(c)) { npos[c] = 3 << 28; continue; }
            const int k = cls_of(c), idx = at[k]++;
            npos[c] = (k << 28) | idx;
            nperm[base[k] + idx] = c;
        }
        // the pads: positions without a variable
        for(int p = arw + t; p < nA; p += SPLC_T) nperm[p] = -1;
        for(int p = nA + brw + t; p < nA + nB; p += SPLC_T) nperm[p] = -1;
    }
}
// the plans of both operating points put out of use (the host has changed what it provides launches for: plans made
// against the old limits may not fit the new grids; the points' next reductions go the ordinary way)
__global__ void nd_plans_off_kernel(const OpDev* __restrict__ ops, int Nc)
{
    const OpDev& O = ops[threadIdx.x];
    if(threadIdx.x >= 2 || O.ndp == NULL || O.cperm == NULL) return;
    O.ndp[NDH_ACTIVE] = 0; O.ndp[NDH_NA] = 0; O.ndp[NDH_NB] = 0; O.ndp[NDH_ARAW] = 0; O.ndp[NDH_BRAW] = 0;
    O.ndp[NDH_NS] = O.cperm[2*Nc]; O.ndp[NDH_NSEFF] = O.cperm[2*Nc];
}
hipError_t launch_nd_plans_off(const OpDev* ops, int Nc, hipStream_t stream)
{
    hipLaunchKernelGGL(nd_plans_off_kernel, dim3(1), dim3(64), 0, stream, ops, Nc);
    return hipGetLastError();
}
__global__ __launch_bounds__(SPLC_T)
void spl_compact_kernel(DeviceProblem P, NormalDims nd, OpRef R, const int* __restrict__ nd_lim)
{
    if(opref_skip(R)) return;
    extern __shared__ int lds_cc[];
    spl_compact_body(P, nd, opref_get(R), lds_cc, nd_lim);
}

// the SPLG_E parts of a row every pass holds, in order; and |x|^2 of the regularization rows
__global__ __launch_bounds__(256)
void assemble_splined_combine_kernel(DeviceProblem P, NormalDims nd, OpRef R, AssemblyPlan plan, int nrow_parts)
{
    if(opref_skip(R)) return;
    const OpDev& O = opref_get(R);
    const int dense = blockIdx.x, stride = nd.Nc + 1;
    const int nintr = P.Nintr_state > 0 ? P.Ncameras_intrinsics*P.Nintr_state : 0;
    const int ncore = P.Nintr_state > 0 ? P.Ncameras_intrinsics*P.Ncore_state : 0;
    const int r = (dense < ncore) ? (dense / P.Ncore_state)*P.Nintr_state + dense % P.Ncore_state : nintr + (dense - ncore);
    for(int c = threadIdx.x; c < stride; c += blockDim.x)
    {
        double s = 0.0;
        for(int e = 0; e < SPLG_E; e++) s += plan.spl_part[((size_t)dense*SPLG_E + e)*stride + c];
        if(r == nd.Nc && c == nd.Nc)
            for(int b = 0; b < nrow_parts; b++) s += plan.row_part[b];
        if(s == 0.0) continue;
        if(r == nd.Nc)
        {
            if(c == nd.Nc) O.scalars[SC_NORM2_X] += s;
            else           O.g[S_to_state(nd, c)] += s;
        }
        else if(c <= r) O.A[(size_t)r*nd.Nc + c] += s;
    }
}

} // namespace mrcal_amd
