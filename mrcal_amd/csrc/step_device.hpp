// Device functions of the trial step's last launch (step.hip step2_backsub_quadform_kernel). In a header of their own
// since round 6, when cholesky_lds.hip ran them as workgroups of the factorization's launch (LEDGER R6.16: slower; gone)
#pragma once
#include "solver_device.hpp"

namespace mrcal_amd {

////////////////////////////////////////////////////////////////////////////////
// v^T N v = |J v|^2 from the blocks;  dot products
////////////////////////////////////////////////////////////////////////////////
// out[0] += v^T N v, and if nout == 3: out[1] += g . v,  out[2] += v . v   with N = [A B; Bt D]
// of the operating point: v^T N v = v_S^T A v_S + 2 v_E^T (Bt v_S) + v_E^T D v_E.
// One wave per group of rows of [A ; Bt]; one atomic triple per workgroup
// (QF_ROWS_PER_WAVE: solver_kernels.hpp. With 8 rows a wave, a 1206-variable camera block had 188 workgroups
//  walking 46 MB of Bt: 1.4 TB/s)
// this workgroup's (256 threads) part of (v^T N v, g.v, v.v): returned in threads 0, 1, 2
// (or this QUARTER's of a workgroup of 1024, `block` counting the quarters: returned in the quarter's threads 0, 1, 2.
//  Every thread of the workgroup must call)
__device__ __forceinline__
double quadform_body(const NormalDims& nd, const OpDev& O, const double* __restrict__ v, int block, bool vv_E_only = false,
                     const unsigned* __restrict__ occ = NULL /* eblock_factor_kernel's bit per (block, 16-column tile) of Wt - and of Bt: the same columns */,
                     int nocc = 0)
{
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, quarter = threadIdx.x >> 8;
    const int Nrows = nd.Nc + nd.NE;
    const int row0  = (block*4 + wave)*QF_ROWS_PER_WAVE;

    // the 8 rows of this wave against v_S, all loads in flight together
    const double* __restrict__ M[QF_ROWS_PER_WAVE];
#pragma unroll
    for(int rr = 0; rr < QF_ROWS_PER_WAVE; rr++)
    {
        int row = row0 + rr;
        if(row >= Nrows) row = Nrows - 1;   // duplicate work, discarded below
        M[rr] = (row < nd.Nc) ? O.A + (size_t)row*nd.Nc : O.Bt + (size_t)(row - nd.Nc)*nd.Nc;
    }
    double acc[QF_ROWS_PER_WAVE];
#pragma unroll
    for(int rr = 0; rr < QF_ROWS_PER_WAVE; rr++) acc[rr] = 0.0;
    // (of A only the lower triangle: the splined assembly writes no other. An entry below the diagonal counts twice)
    int rowc[QF_ROWS_PER_WAVE];
    // (the splined models: a row of Bt holds something under the frame's board only - a sixth of its 76 tiles, 46 MB of
    //  zeros a step otherwise: a lane whose tile is empty asks for nothing)
    const unsigned* __restrict__ ob[QF_ROWS_PER_WAVE];
#pragma unroll
    for(int rr = 0; rr < QF_ROWS_PER_WAVE; rr++)
    {
        rowc[rr] = min(row0 + rr, Nrows - 1);
        ob[rr] = NULL;
        if(occ != NULL && rowc[rr] >= nd.Nc)
        {
            int blk, a, de, e0;
            E_to_block(nd, rowc[rr] - nd.Nc, &blk, &a, &de, &e0);
            ob[rr] = occ + (size_t)blk*nocc;
        }
    }
#pragma unroll 4
    for(int c = lane; c < nd.Nc; c += 64)
    {
        const double vs = v[S_to_state(nd, c)];
        const int tile = c >> 4;
#pragma unroll
        for(int rr = 0; rr < QF_ROWS_PER_WAVE; rr++)
        {
            const double wgt = (rowc[rr] >= nd.Nc) ? 1.0 : (c < rowc[rr]) ? 2.0 : (c == rowc[rr]) ? 1.0 : 0.0;
            // (a branch around the load: the lanes without one reading the row's first entry instead - no branch, the loads
            //  of four steps in flight - measured slower, 24 us against 20)
            const bool there = (ob[rr] == NULL) || ((ob[rr][tile >> 5] >> (tile & 31)) & 1u);
            if(wgt != 0.0 && there) acc[rr] += wgt*(M[rr][c]*vs);
        }
    }
#pragma unroll
    for(int rr = 0; rr < QF_ROWS_PER_WAVE; rr++)
        for(int off=32; off>0; off>>=1) acc[rr] += __shfl_down(acc[rr], off);
    // lane rr finishes row rr
    double mine = 0.0;
#pragma unroll
    for(int rr = 0; rr < QF_ROWS_PER_WAVE; rr++)
    {
        const double a0 = __shfl(acc[rr], 0);
        if(lane == rr) mine = a0;
    }
    double t_vNv = 0.0, t_gv = 0.0, t_vv = 0.0;
    const int row = row0 + lane;
    if(lane < QF_ROWS_PER_WAVE && row < Nrows)
    {
        int    is;      // state index of this row's variable
        double wgt;
        if(row < nd.Nc) { is = S_to_state(nd, row); wgt = 1.0; }
        else            { is = E_to_state(nd, row - nd.Nc);                              wgt = 2.0; }
        const double vr = v[is];
        double total = wgt*vr*mine;
        if(row >= nd.Nc)
        {
            int blk, a, de, e0;
            E_to_block(nd, row - nd.Nc, &blk, &a, &de, &e0);
            double s = 0.0;
            for(int c=0;c<de;c++) s += O.D[(size_t)blk*36 + a*6 + c]*v[nd.E_state0 + e0 + c];
            total += vr*s;
        }
        t_vNv = total;
        t_gv  = O.g[is]*vr;
        t_vv  = (vv_E_only && row < nd.Nc) ? 0.0 : vr*vr;
    }
    for(int off=4; off>0; off>>=1)
    {
        t_vNv += __shfl_down(t_vNv, off);
        t_gv  += __shfl_down(t_gv,  off);
        t_vv  += __shfl_down(t_vv,  off);
    }
    __shared__ double part_q[4][4][3];
    double (*part)[3] = part_q[quarter];
    if(lane == 0) { part[wave][0] = t_vNv; part[wave][1] = t_gv; part[wave][2] = t_vv; }
    __syncthreads();
    const int tq = threadIdx.x & 255;
    if(tq < 3)
        return (part[0][tq] + part[1][tq]) + (part[2][tq] + part[3][tq]);
    return 0.0;
}

// The back-substitution of ONE eliminated block by one wave: d_e = -L^-T (y_e + Wt_e d_s); (|d_e|^2, d_e . g_e) into
// dots_part[ibk]. ready(): called once everything that does not depend on d_s has been asked for (a place to wait for
// d_s's maker, for a caller that shares a launch with it); false: there is no d_s, nothing is done
template<class Ready>
__device__ __forceinline__
void backsub_eblock(const NormalDims& nd, const BlockRanges& br, const OpDev& O,
                    const double* __restrict__ Wt, const double* __restrict__ LD,
                    const double* __restrict__ y, const double* __restrict__ ds,
                    double* __restrict__ dots_part, const int ibk, const unsigned* __restrict__ occ, int nocc, Ready ready)
{
    double* __restrict__ step = O.step_gn;
    const int lane = threadIdx.x & 63;
    if(ibk >= br.count()) return;
    const int blk = br.block(ibk);
    const int de  = (blk < nd.Nfb) ? 6 : 3;
    const int e0  = (blk < nd.Nfb) ? 6*blk : 6*nd.Nfb + 3*(blk - nd.Nfb);
    // all the loads first: L, y, g_e and this lane's columns of Wt_e against d_s
    // (unconditional, clamped: three loads under conditions were three branches with a wait each)
    const double Lv = LD[(size_t)blk*36 + min(lane, 35)];
    const double yv = y[e0 + min(lane, de - 1)];
    const double gv = O.g[nd.E_state0 + e0 + min(lane, de - 1)];
    double part[6] = {0,0,0,0,0,0};
    // (the tiles of Wt that hold nothing - five in six under the splined models - are not asked for)
    const unsigned* __restrict__ ob = (occ != NULL) ? occ + (size_t)blk*nocc : (const unsigned*)NULL;
    // camera blocks to 256 variables (every one whose factorization is the one-workgroup kernel): this lane's (up to)
    // four columns of Wt_e in registers before d_s is asked for. The same products in the same order as the loop below
    constexpr int MAXC = 4;
    const bool pre = (nd.Nc <= 64*MAXC) && ob == NULL;
    double wt[MAXC][6];
    if(pre)
    {
#pragma unroll
        for(int cc = 0; cc < MAXC; cc++)
        {
            const int c = lane + 64*cc;
#pragma unroll
            for(int i=0;i<6;i++) wt[cc][i] = (i < de && c < nd.Nc) ? Wt[(size_t)(e0+i)*nd.Nc + c] : 0.0;
        }
    }
    if(!ready()) return;
    if(pre)
    {
#pragma unroll
        for(int cc = 0; cc < MAXC; cc++)
        {
            const int c = lane + 64*cc;
            if(c >= nd.Nc) break;
            const double d = ds[c];
#pragma unroll
            for(int i=0;i<6;i++) if(i < de) part[i] += wt[cc][i]*d;
        }
    }
    else
    {
    // (four column groups asked for together: with a 1206-variable camera block the loop is 19 round trips otherwise)
#pragma unroll 4
    for(int c = lane; c < nd.Nc; c += 64)
    {
        const double d = ds[c];
        const int tile = c >> 4;
        if(ob != NULL && !((ob[tile >> 5] >> (tile & 31)) & 1u)) continue;
#pragma unroll
        for(int i=0;i<6;i++) if(i < de) part[i] += Wt[(size_t)(e0+i)*nd.Nc + c]*d;
    }
    }
#pragma unroll
    for(int i=0;i<6;i++)
        for(int off=32; off>0; off>>=1) part[i] += __shfl_down(part[i], off);
    // lane 0 holds the sums; L, y, g come from the lanes that loaded them (no LDS, no barrier)
    double v[6], Lr[6][6], ge[6];
#pragma unroll
    for(int i=0;i<6;i++)
    {
        v[i]  = __shfl(yv, i) + __shfl(part[i], 0);
        ge[i] = __shfl(gv, i);
#pragma unroll
        for(int k=0;k<6;k++) Lr[i][k] = __shfl(Lv, i*6 + k);
    }
    if(lane == 0)
    {
        double d2 = 0.0, dg = 0.0;
#pragma unroll
        for(int i=5;i>=0;i--)
        {
            if(i >= de) continue;
            double sacc = v[i];
#pragma unroll
            for(int k=i+1;k<6;k++) if(k < de) sacc -= Lr[k][i]*v[k];
            v[i] = sacc/Lr[i][i];
        }
#pragma unroll
        for(int i=0;i<6;i++)
            if(i < de)
            {
                const double d = -v[i];
                step[nd.E_state0 + e0 + i] = d;
                d2 += d*d; dg += d*ge[i];
            }
        dots_part[2*ibk] = d2; dots_part[2*ibk + 1] = dg;
    }
}

} // namespace mrcal_amd
