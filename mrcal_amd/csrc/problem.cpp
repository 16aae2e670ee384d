// The resident problem object and the compute half of the C ABI.
//
// mrcal_amd_problem_t owns every HBM buffer of one calibration problem (or of
// one frame-shard of it) and the HIP stream its kernels run on. The drop-in
// mrcal_optimizer_callback() is a thin shell over it: create, evaluate, copy
// out, destroy.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <map>
#include <algorithm>
#include "layout.hpp"
#include "host_state.hpp"
#include "problem.hpp"
#include "kernels.hpp"
#include "problem_object.hpp"
#include "host_copy.hpp"
#include <thread>
#include <unordered_map>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <chrono>
#include <unistd.h>

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

namespace mrcal_amd {

template<class T>
static bool dev_alloc(T** p, size_t n)
{
    *p = NULL;
    if(n == 0) n = 1;
    HIP_TRY(hipMalloc((void**)p, n*sizeof(T)), return false);
    return true;
}
template<class T>
static bool dev_upload(T** p, const T* host, size_t n)
{
    if(!dev_alloc(p, n)) return false;
    if(n > 0 && host != NULL)
        HIP_TRY(hipMemcpy(*p, host, n*sizeof(T), hipMemcpyHostToDevice), return false);
    return true;
}

} // namespace

mrcal_amd_problem::~mrcal_amd_problem()
{
    hipFree(d_seed_intrinsics); hipFree(d_seed_rt_cam_ref); hipFree(d_seed_rt_ref_frame);
    hipFree(d_seed_points); hipFree(d_board_meta); hipFree(d_board_pool);
    hipFree(d_point_meta); hipFree(d_point_pool); hipFree(d_imagersizes);
    hipFree(d_tri_meta); hipFree(d_tri_px); hipFree(d_tri_outlier);
    hipFree(d_joint); hipFree(d_gram); hipFree(d_Jp); hipFree(d_Ji);
    for(int i=0;i<2;i++)
    {
        hipFree(op[i].b); hipFree(op[i].x); hipFree(op[i].Jv); hipFree(op[i].spl_box);
        hipFree(op[i].A); hipFree(op[i].Bt); hipFree(op[i].D); hipFree(op[i].g); hipFree(op[i].scalars);
        hipFree(op[i].step_cauchy); hipFree(op[i].step_gn);
    }
    hipFree(d_ops);
    hipFree(plan.frame_obs_begin); hipFree(plan.frame_obs); hipFree(plan.chunk_begin); hipFree(plan.pair_obs); hipFree(plan.pos_table);
    hipFree(plan.chunk_pair); hipFree(plan.obs_pair); hipFree(plan.pair_table); hipFree(plan.frame_pos); hipFree(plan.obs_cols);
    hipFree(plan.chunk_part); hipFree(plan.dest_id); hipFree(plan.dest_begin); hipFree(plan.dest_src);
    hipFree(plan.pair_chunk_begin); hipFree(plan.row_part); hipFree(plan.qf_part); hipFree(plan.dots_part);
    hipFree(plan.spl_hdr); hipFree(plan.spl_part); hipFree(plan.spl_hdr_extra); hipFree(plan.chunk_extra);
    {
        mrcal_amd::GenPlan& G = plan.gen;
        hipFree(G.rows); hipFree(G.chunk_begin); hipFree(G.chunk_group); hipFree(G.group_k); hipFree(G.group_off); hipFree(G.spos);
        hipFree(G.scol); hipFree(G.part); hipFree(G.dest_id); hipFree(G.dest_begin); hipFree(G.dest_src); hipFree(G.group_chunk_begin);
        hipFree(G.eb_block); hipFree(G.eb_begin); hipFree(G.eb_rows); hipFree(G.eb_group); hipFree(G.eb_epos);
    }
    hipFree(F.Wt); hipFree(F.LD); hipFree(F.y); hipFree(F.S); hipFree(F.Spart); hipFree(F.Linv); hipFree(F.diag_minmax); hipFree(F.status); hipFree(F.occ); hipFree(F.Wtile);
    hipFree(cperm_cur_alloc); hipFree(F.iso); hipFree(op[0].cperm); hipFree(op[1].cperm);
    hipFree(op[0].ndp); hipFree(op[1].ndp); hipFree(F.ndp_cur); hipFree(F.ndMA); hipFree(F.ndMB); hipFree(F.ndLinvA); hipFree(F.ndLinvB); hipFree(F.ndPart); hipFree(F.nd_lim_dev);
    hipFree(plan.repro.lvl[0]); hipFree(plan.repro.lvl[1]); hipFree(plan.repro.lvl[2]); hipFree(plan.repro.cmax); hipFree(plan.repro.any);
    hipFree(d_step); hipFree(d_comm); hipFree(d_counts); hipFree(d_outlier_part); hipFree(d_ctl);
    if(h_scalars)  hipHostFree(h_scalars);
    if(h_ctl_ring) hipHostFree(h_ctl_ring);
    for(hipEvent_t e : ctl_events) hipEventDestroy(e);
    for(int i=0;i<3;i++) if(step_graph[i]) hipGraphExecDestroy(step_graph[i]);
    for(hipEvent_t e : ev_pool) hipEventDestroy(e);
    if(ev_j0)  hipEventDestroy(ev_j0);
    if(ev_j1)  hipEventDestroy(ev_j1);
    if(ev_fork) hipEventDestroy(ev_fork);
    if(ev_join) hipEventDestroy(ev_join);
    if(side_stream) hipStreamDestroy(side_stream);
    if(stream) hipStreamDestroy(stream);
}

namespace mrcal_amd {

// The fixed-order plan for the rows outside the Grams that share destinations: discrete points, triangulated
// pairs (GenPlan, solver_kernels.hpp). From the CSR structure itself, which does not change between evaluations
// (except the splined models' patch columns: no plan then, those rows keep the atomics)
static bool build_gen_plan(mrcal_amd_problem* P)
{
    GenPlan& G = P->plan.gen;
    memset(&G, 0, sizeof(G));
    const Layout& L = P->L;
    const NormalDims& nd = P->nd;
    const int r0 = L.i_meas_points, r1 = L.i_meas_regularization;
    G.row_first = r0; G.row_end = r1;
    if(r1 <= r0) return true;
    if(L.lensmodel.type == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC && L.Nmeas_points > 0 && L.Ndist_state > 0) return true;
    std::vector<int32_t> Jp((size_t)(r1 - r0) + 1);
    HIP_TRY(hipMemcpy(Jp.data(), P->d_Jp + r0, Jp.size()*sizeof(int32_t), hipMemcpyDeviceToHost), return false);
    const int32_t p0 = Jp[0], p1 = Jp[r1 - r0];
    std::vector<int32_t> Ji((size_t)(p1 - p0 > 0 ? p1 - p0 : 1));
    if(p1 > p0) HIP_TRY(hipMemcpy(Ji.data(), P->d_Ji + p0, (size_t)(p1 - p0)*sizeof(int32_t), hipMemcpyDeviceToHost), return false);

    struct RowInfo { int group, eblk, epos; };
    std::vector<RowInfo> info((size_t)(r1 - r0));
    // a row's signature [k | spos.. | scol..] -> its group, the groups numbered as they first appear. (Round 6: the
    // signature on the stack and a hash in front of the comparison; three vectors and an ordered map of vectors a row
    // were 10 ms of BASELINE configuration 4's 67 000 rows - as long as its five dog-leg steps and their launches together)
    std::unordered_map<uint64_t, std::vector<int>> groups;
    std::vector<std::vector<int>> group_sig;
    int kmax = 0;
    for(int r = r0; r < r1; r++)
    {
        const int a = Jp[r - r0] - p0, b = Jp[r - r0 + 1] - p0;
        int sig[2*GEN_KMAX + 1], scol[GEN_KMAX];
        int k = 0;
        int eblk = -1, epos = -1, ecount = 0;
        for(int p = a; p < b; p++)
        {
            const int c = Ji[p];
            if(c < 0 || c >= nd.Nstate) return true;                 // (flagged at run time by the row-by-row path)
            const int se = state_to_SE(nd, c);
            if(se >= 0)
            {
                if(k >= GEN_KMAX) return true;
                sig[1 + k] = p - a; scol[k] = se; k++;
            }
            else
            {
                const int e = -se - 1;
                const int blk = (e < 6*nd.Nfb) ? e/6 : nd.Nfb + (e - 6*nd.Nfb)/3;
                const int e0  = (blk < nd.Nfb) ? 6*blk : 6*nd.Nfb + 3*(blk - nd.Nfb);
                const int de  = (blk < nd.Nfb) ? 6 : 3;
                // the block's columns must be all there, side by side, in order
                if(ecount == 0) { if(e != e0) return true; eblk = blk; epos = p - a; }
                else if(blk != eblk || e != e0 + ecount) return true;
                ecount++;
                if(ecount > de) return true;
            }
        }
        if(eblk >= 0 && ecount != ((eblk < nd.Nfb) ? 6 : 3)) return true;
        if(k > kmax) kmax = k;
        sig[0] = k;
        for(int i = 0; i < k; i++) sig[1 + k + i] = scol[i];
        const int nsig = 2*k + 1;
        uint64_t h = 1469598103934665603ull;
        for(int i = 0; i < nsig; i++) { h ^= (uint64_t)(uint32_t)sig[i]; h *= 1099511628211ull; }
        std::vector<int>& cand = groups[h];
        int g = -1;
        for(int gc : cand)
            if((int)group_sig[gc].size() == nsig && !memcmp(group_sig[gc].data(), sig, nsig*sizeof(int))) { g = gc; break; }
        if(g < 0) { g = (int)group_sig.size(); cand.push_back(g); group_sig.emplace_back(sig, sig + nsig); }
        info[r - r0] = RowInfo{ g, eblk, epos };
    }
    const int Ngroups = (int)group_sig.size();
    const int stride  = (kmax*(kmax+1))/2 + kmax + 1;
    if(stride > 1023 || Ngroups >= (1 << 20)) return true;
    // (gen_eblock keeps a block's rows of Bt in LDS: 6 Nc doubles, within the 64 KB a launch gets without asking)
    {
        bool any_eblock = false;
        for(const RowInfo& ri : info) any_eblock = any_eblock || ri.eblk >= 0;
        if(any_eblock && ((size_t)6*nd.Nc + 42)*sizeof(double) > 64*1024) return true;
    }

    // rows by (group, row); chunks
    // (by counting: the groups are few)
    std::vector<int> rows((size_t)(r1 - r0));
    {
        std::vector<int> at(Ngroups + 1, 0);
        for(const RowInfo& ri : info) at[ri.group + 1]++;
        for(int g = 0; g < Ngroups; g++) at[g + 1] += at[g];
        for(int i = 0; i < r1 - r0; i++) rows[at[info[i].group]++] = r0 + i;
    }
    std::vector<int> chunk_begin, chunk_group, group_chunk_begin(Ngroups + 1, 0);
    for(size_t i = 0; i < rows.size();)
    {
        const int g = info[rows[i] - r0].group;
        size_t j = i;
        while(j < rows.size() && j - i < GEN_CHUNK && info[rows[j] - r0].group == g) j++;
        chunk_begin.push_back((int)i); chunk_group.push_back(g);
        group_chunk_begin[g + 1]++;
        i = j;
    }
    chunk_begin.push_back((int)rows.size());
    for(int g = 0; g < Ngroups; g++) group_chunk_begin[g+1] += group_chunk_begin[g];
    std::vector<int> group_k(Ngroups), group_off(Ngroups), spos_all, scol_all;
    for(int g = 0; g < Ngroups; g++)
    {
        const std::vector<int>& sig = group_sig[g];
        const int k = sig[0];
        group_k[g] = k; group_off[g] = (int)spos_all.size();
        spos_all.insert(spos_all.end(), sig.begin() + 1, sig.begin() + 1 + k);
        scol_all.insert(scol_all.end(), sig.begin() + 1 + k, sig.end());
    }
    // destinations: entries of A (both orientations, as the row-by-row path adds them), of g (S part), |x|^2
    const int nA = nd.Nc*nd.Nc;
    std::map<int, std::vector<int>> src;
    for(int g = 0; g < Ngroups; g++)
    {
        const int k = group_k[g];
        const int* sc = scol_all.data() + group_off[g];
        int pos = 0;
        for(int p = 0; p < k; p++)
            for(int q = p; q < k; q++, pos++)
            {
                const int code = (g << 10) | pos;
                src[sc[p]*nd.Nc + sc[q]].push_back(code);
                if(q != p) src[sc[q]*nd.Nc + sc[p]].push_back(code);
            }
        for(int p = 0; p < k; p++, pos++) src[nA + sc[p]].push_back((g << 10) | pos);
        src[nA + nd.Nc].push_back((g << 10) | pos);
    }
    std::vector<int> dest_id, dest_begin(1, 0), dest_src;
    for(auto& kv : src)
    {
        dest_id.push_back(kv.first);
        dest_src.insert(dest_src.end(), kv.second.begin(), kv.second.end());
        dest_begin.push_back((int)dest_src.size());
    }
    // the eliminated blocks: their rows, in row order
    std::map<int, std::vector<int>> by_block;
    for(int r = r0; r < r1; r++)
        if(info[r - r0].eblk >= 0) by_block[info[r - r0].eblk].push_back(r);
    std::vector<int> eb_block, eb_begin(1, 0), eb_rows, eb_group, eb_epos;
    for(auto& kv : by_block)
    {
        eb_block.push_back(kv.first);
        for(int r : kv.second) { eb_rows.push_back(r); eb_group.push_back(info[r - r0].group); eb_epos.push_back(info[r - r0].epos); }
        eb_begin.push_back((int)eb_rows.size());
    }
    auto nonempty = [](std::vector<int>& v) { if(v.empty()) v.push_back(0); };
    const int Nchunks = (int)chunk_group.size(), Neb = (int)eb_block.size(), Ndest = (int)dest_id.size();
    nonempty(chunk_group); nonempty(spos_all); nonempty(scol_all); nonempty(dest_id); nonempty(dest_src);
    nonempty(eb_block); nonempty(eb_rows); nonempty(eb_group); nonempty(eb_epos);
    bool ok = true;
    ok = ok && dev_upload(&G.rows,        rows.data(),        rows.size());
    ok = ok && dev_upload(&G.chunk_begin, chunk_begin.data(), chunk_begin.size());
    ok = ok && dev_upload(&G.chunk_group, chunk_group.data(), chunk_group.size());
    ok = ok && dev_upload(&G.group_k,     group_k.data(),     group_k.size());
    ok = ok && dev_upload(&G.group_off,   group_off.data(),   group_off.size());
    ok = ok && dev_upload(&G.spos,        spos_all.data(),    spos_all.size());
    ok = ok && dev_upload(&G.scol,        scol_all.data(),    scol_all.size());
    ok = ok && dev_upload(&G.dest_id,     dest_id.data(),     dest_id.size());
    ok = ok && dev_upload(&G.dest_begin,  dest_begin.data(),  dest_begin.size());
    ok = ok && dev_upload(&G.dest_src,    dest_src.data(),    dest_src.size());
    ok = ok && dev_upload(&G.group_chunk_begin, group_chunk_begin.data(), group_chunk_begin.size());
    ok = ok && dev_upload(&G.eb_block,    eb_block.data(),    eb_block.size());
    ok = ok && dev_upload(&G.eb_begin,    eb_begin.data(),    eb_begin.size());
    ok = ok && dev_upload(&G.eb_rows,     eb_rows.data(),     eb_rows.size());
    ok = ok && dev_upload(&G.eb_group,    eb_group.data(),    eb_group.size());
    ok = ok && dev_upload(&G.eb_epos,     eb_epos.data(),     eb_epos.size());
    ok = ok && dev_alloc (&G.part, (size_t)(Nchunks > 0 ? Nchunks : 1)*stride);
    if(!ok) return false;
    G.Nrows = r1 - r0; G.Nchunks = Nchunks; G.Ngroups = Ngroups; G.stride = stride; G.kmax = kmax;
    G.Ndest = Ndest; G.Neblocks = Neb;
    return true;
}

bool problem_prepare_solver(mrcal_amd_problem* P)
{
    if(P->solver_ready) return true;
    const Layout& L = P->L;
    NormalDims& nd = P->nd;

    // second operating point
    bool ok = true;
    ok = ok && dev_alloc(&P->op[1].b,  (size_t)L.Nstate);
    ok = ok && dev_alloc(&P->op[1].x,  (size_t)L.Nmeas);
    ok = ok && dev_alloc(&P->op[1].Jv, (size_t)P->Nnz);
    if(P->op[0].spl_box != NULL) ok = ok && dev_alloc(&P->op[1].spl_box, (size_t)4*P->D.Nobs_board);
    if(L.lensmodel.type != MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC && (size_t)P->D.Nobs_board*gram_stride(L.Ndist) >= ((size_t)1 << 32))
    {
        // (reduce_pair_chunk() addresses the Grams with 32-bit element offsets; this is 34 GB of Grams)
        set_error("too many board observations: %d Grams of %d doubles", P->D.Nobs_board, gram_stride(L.Ndist));
        return false;
    }
    ok = ok && dev_alloc(&P->d_gram,   (L.lensmodel.type == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC) ? (size_t)1 : (size_t)P->D.Nobs_board*gram_stride(L.Ndist));
    for(int i=0;i<2 && ok;i++)
    {
        ok = ok && dev_alloc(&P->op[i].A,       (size_t)nd.Nc*nd.Nc);
        ok = ok && dev_alloc(&P->op[i].Bt,      (size_t)nd.NE*nd.Nc);
        ok = ok && dev_alloc(&P->op[i].D,       (size_t)nd.NEb*36);
        ok = ok && dev_alloc(&P->op[i].g,       (size_t)nd.Nstate);
        ok = ok && dev_alloc(&P->op[i].scalars, (size_t)NSCALARS);
        ok = ok && dev_alloc(&P->op[i].step_cauchy, (size_t)nd.Nstate);
        ok = ok && dev_alloc(&P->op[i].step_gn,     (size_t)nd.Nstate);
    }
    ok = ok && dev_alloc(&P->F.Spart, schur_partial_doubles(nd));
    ok = ok && dev_alloc(&P->F.Linv,  cholesky_large_workspace_doubles(nd.Nc));
    if(cholesky_large_workspace_doubles(nd.Nc) > 1) ok = ok && dev_alloc(&P->F.diag_minmax, 2);
    // the tile occupancy of Wt: only where the couplings are sparse (the splined models) and the strip SYRK runs
    if(L.lensmodel.type == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC && nd.Nc > 256 && nd.Nc <= 4096)
    {
        ok = ok && dev_alloc(&P->F.occ, (size_t)(nd.NEb > 0 ? nd.NEb : 1)*occ_words(nd));
        ok = ok && dev_alloc(&P->F.Wtile, (size_t)((nd.Nc + 15)/16)*16*(size_t)(nd.NE > 0 ? nd.NE : 1));
    }
    // The splined models' camera block without the control points no board covers (round 5; assembly_splined.hip,
    // spl_compact_kernel / LcholCompact): where the big camera block's launch-per-panel Cholesky runs, every row that
    // touches a control point is a board's (no discrete points: they have no boxes) and all rows are here (not a shard:
    // the ranks of a sharded solve sum their camera blocks entry by entry). MRCAL_AMD_NO_SPL_COMPACT=1: off
    {
        // (nor with the backward sweep, which knows nothing of a size the device decides)
        P->F.use_sweep = test_hooks().lchol_sweep ? 1 : 0;
        static const bool env_off = (getenv("MRCAL_AMD_NO_SPL_COMPACT") != NULL);
        const bool off = env_off || P->F.use_sweep;
        const bool whole = (int)P->board_sel.size() == L.dims.Nobservations_board && P->comm == NULL;
        if(!off && whole && L.lensmodel.type == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC && cholesky_large_workspace_doubles(nd.Nc) > 1 && nd.Nc <= 4096 &&
           P->D.Nobs_board > 0 && P->D.Nobs_point == 0 && P->D.Ndist_state > 0 && !nd.elim_extrinsics)
        {
            for(int i=0;i<2 && ok;i++)
            {
                ok = ok && dev_alloc(&P->op[i].cperm, (size_t)2*nd.Nc + 1);
                // (until the first evaluation: the identity)
                std::vector<int> id((size_t)2*nd.Nc + 1);
                for(int c = 0; c < nd.Nc; c++) { id[c] = c; id[nd.Nc + c] = c; }
                id[2*nd.Nc] = nd.Nc;
                if(ok) HIP_TRY(hipMemcpy(P->op[i].cperm, id.data(), id.size()*sizeof(int), hipMemcpyHostToDevice), ok = false);
            }
            ok = ok && dev_alloc(&P->cperm_cur_alloc, (size_t)2*nd.Nc + 2);
            ok = ok && dev_alloc(&P->F.iso, (size_t)4*(nd.Nc/2 + 1) + nd.Nc + 2);
            P->F.cperm_cur = P->cperm_cur_alloc;
            P->plan.spl_compact = 1;
            // ... and in a nested-dissection order where the boards leave a strip worth having (cholesky_large.hip,
            // lchol_nd_*): one camera's grid. MRCAL_AMD_NO_ND=1: off
            static const bool nd_off = (getenv("MRCAL_AMD_NO_ND") != NULL);
            if(!nd_off && P->D.Ncameras_intrinsics == 1)
            {
                const size_t Npos = (size_t)nd.Nc + 2*ND_PANEL;
                const size_t W = LCH_ND_WMAX, wsz = (W/ND_PANEL)*ND_PANEL*ND_PANEL + W*W + W;
                std::vector<int> h0(nd_plan_ints(nd.Nc), 0);
                h0[NDH_NS] = nd.Nc; h0[NDH_NSEFF] = nd.Nc;
                for(int i=0;i<2 && ok;i++)
                {
                    ok = ok && dev_alloc(&P->op[i].ndp, h0.size());
                    if(ok) HIP_TRY(hipMemcpy(P->op[i].ndp, h0.data(), h0.size()*sizeof(int), hipMemcpyHostToDevice), ok = false);
                }
                ok = ok && dev_alloc(&P->F.ndp_cur, h0.size());
                if(ok) HIP_TRY(hipMemcpy(P->F.ndp_cur, h0.data(), h0.size()*sizeof(int), hipMemcpyHostToDevice), ok = false);
                ok = ok && dev_alloc(&P->F.ndMA, (Npos + 1)*Npos) && dev_alloc(&P->F.ndMB, (Npos + 1)*Npos);
                ok = ok && dev_alloc(&P->F.ndLinvA, wsz) && dev_alloc(&P->F.ndLinvB, wsz);
                ok = ok && dev_alloc(&P->F.ndPart, (size_t)((nd.Nc + 15)/16)*2*LCH_ND_WMAX);
                ok = ok && dev_alloc(&P->F.nd_lim_dev, 2);
                if(ok) HIP_TRY(hipMemset(P->F.nd_lim_dev, 0, 2*sizeof(int)), ok = false);
                P->F.nd_lim = NdLimits{0, 0}; P->F.nd_likely_panels = 0;
                P->plan.nd_lim = P->F.nd_lim_dev;
            }
        }
    }
    // the rows of a splined problem that no plan covers: their three levels of pre-rounded sums (ReproStep), zero at rest
    if(ok && splined_needs_repro_rows(P->D))
    {
        ReproStep& rs = P->plan.repro;
        rs.one = (size_t)nd.Nc*nd.Nc + (size_t)nd.NE*nd.Nc + (size_t)nd.NEb*36 + (size_t)nd.Nstate + 1;
        for(int l = 0; l < 3 && ok; l++)
        {
            ok = ok && dev_alloc(&rs.lvl[l], rs.one);
            if(ok) HIP_TRY(hipMemset(rs.lvl[l], 0, rs.one*sizeof(double)), ok = false);
        }
        ok = ok && dev_alloc(&rs.cmax, (size_t)nd.Nstate + 1);
        ok = ok && dev_alloc(&rs.any, 1);
        if(ok) HIP_TRY(hipMemset(rs.cmax, 0, ((size_t)nd.Nstate + 1)*sizeof(unsigned long long)), ok = false);
        if(ok) HIP_TRY(hipMemset(rs.any, 0, sizeof(int)), ok = false);
        if(!ok) rs.lvl[0] = NULL;
    }
    {
        char* ctl = NULL;
        ok = ok && dev_alloc(&ctl, solver_ctl_bytes());
        P->d_ctl = (SolverCtl*)ctl;
    }
    ok = ok && dev_alloc(&P->F.Wt, (size_t)nd.NE*nd.Nc);
    ok = ok && dev_alloc(&P->F.LD, (size_t)nd.NEb*36);
    ok = ok && dev_alloc(&P->F.y,  (size_t)nd.NE);
    // S and r contiguous: one all-reduce moves both
    // [S | r | g_S | |x|^2 | status]: comm1 of the sharded step (step2_comm1_doubles())
    // (round 6: + a packed copy of the lower triangle behind them, for the one-workgroup Cholesky: factor_S_packed())
    ok = ok && dev_alloc(&P->F.S,  (size_t)nd.Nc*nd.Nc + 2*nd.Nc + 2 + 64 + ((((size_t)nd.Nc*(nd.Nc + 1)) >> 1) + nd.Nc + 2));
    P->F.r = ok ? P->F.S + (size_t)nd.Nc*nd.Nc : NULL;
    ok = ok && dev_alloc(&P->F.status, 1);
    ok = ok && dev_alloc(&P->d_step,   (size_t)nd.Nstate);
    ok = ok && dev_alloc(&P->d_comm,   (size_t)nd.Nstate + 64);
    ok = ok && dev_alloc(&P->d_counts, 4);
    ok = ok && dev_alloc(&P->d_outlier_part, outlier_partial_doubles());
    if(!ok) return false;
    // only the lower triangle of S is ever written; the rest rides along in the
    // all-reduce of [S | r] and should be numbers
    HIP_TRY(hipMemset(P->F.S, 0, ((size_t)nd.Nc*nd.Nc + 2*nd.Nc + 2)*sizeof(double)), return false);
    // rows of blocks this shard does not own are never written: they must read as 0
    HIP_TRY(hipMemset(P->F.Wt, 0, (size_t)(nd.NE*nd.Nc > 0 ? nd.NE*nd.Nc : 1)*sizeof(double)), return false);
    HIP_TRY(hipMemset(P->F.y,  0, (size_t)(nd.NE > 0 ? nd.NE : 1)*sizeof(double)), return false);
    for(int i=0;i<2;i++)
    {
        HIP_TRY(hipMemset(P->op[i].step_gn,     0, (size_t)nd.Nstate*sizeof(double)), return false);
        HIP_TRY(hipMemset(P->op[i].step_cauchy, 0, (size_t)nd.Nstate*sizeof(double)), return false);
    }
    HIP_TRY(hipHostMalloc((void**)&P->h_scalars, 64*sizeof(double)), return false);
    if(!problem_sync_ops(P)) return false;

    // assembly work lists. Observations of one frame are contiguous; the
    // observations of one (intrinsics,extrinsics) pair are gathered in chunks
    const int Nobs = P->D.Nobs_board;
    std::vector<BoardObsMeta> meta(Nobs);
    if(Nobs > 0)
        HIP_TRY(hipMemcpy(meta.data(), P->d_board_meta, (size_t)Nobs*sizeof(BoardObsMeta), hipMemcpyDeviceToHost), return false);
    // the eliminated pose of an observation (its frame; with elim_extrinsics its camera, which may be the
    // reference: none then), and the one that stays in the camera block
    const bool elimx = nd.elim_extrinsics != 0;
    auto eblock_of = [&](const BoardObsMeta& m) -> int { return elimx ? m.icam_extrinsics : m.iframe; };
    // (what a Gram position means is common to the observations with the same camera-block columns AND the same
    //  columns present: the key of a "pair" carries whether there is an eliminated pose - a camera at the
    //  reference has none)
    auto spose_of  = [&](const BoardObsMeta& m) -> int { return 2*(elimx ? m.iframe : m.icam_extrinsics) + ((eblock_of(m) >= 0) ? 1 : 0); };
    const int Neblocks_board = elimx ? P->D.Ncameras_extrinsics : L.dims.Nframes;
    std::vector<int> frame_begin(Neblocks_board+1, 0);
    for(int o=0;o<Nobs;o++) if(eblock_of(meta[o]) >= 0) frame_begin[eblock_of(meta[o])+1]++;
    for(int f=0;f<Neblocks_board;f++) frame_begin[f+1] += frame_begin[f];
    // sanity: contiguity
    for(int o=1;o<Nobs;o++)
        if(meta[o].iframe < meta[o-1].iframe)
        {
            set_error("board observations must be sorted by frame");
            return false;
        }
    // (the observations of a frame are contiguous, those of a camera need not be: a list then)
    std::vector<int> eblock_obs;
    if(elimx)
    {
        std::vector<int> fill(frame_begin.begin(), frame_begin.end() - 1);
        eblock_obs.assign(Nobs > 0 ? Nobs : 1, 0);
        for(int o=0;o<Nobs;o++) if(eblock_of(meta[o]) >= 0) eblock_obs[fill[eblock_of(meta[o])]++] = o;
    }

    std::vector<int> order(Nobs);
    for(int o=0;o<Nobs;o++) order[o] = o;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b)
                     {
                         if(meta[a].icam_intrinsics != meta[b].icam_intrinsics) return meta[a].icam_intrinsics < meta[b].icam_intrinsics;
                         return spose_of(meta[a]) < spose_of(meta[b]);
                     });
    const int CHUNK = REDUCE_CHUNK;
    std::vector<int> chunk_begin;
    for(int i=0;i<Nobs;)
    {
        int j = i;
        while(j < Nobs && j - i < CHUNK &&
              meta[order[j]].icam_intrinsics == meta[order[i]].icam_intrinsics &&
              spose_of(meta[order[j]]) == spose_of(meta[order[i]])) j++;
        chunk_begin.push_back(i);
        i = j;
    }
    chunk_begin.push_back(Nobs);
    P->plan.Nchunks = (int)chunk_begin.size() - 1;
    ok = ok && dev_upload(&P->plan.frame_obs_begin, frame_begin.data(), frame_begin.size());
    if(elimx) ok = ok && dev_upload(&P->plan.frame_obs, eblock_obs.data(), eblock_obs.size());
    ok = ok && dev_upload(&P->plan.chunk_begin,     chunk_begin.data(), chunk_begin.size());
    ok = ok && dev_upload(&P->plan.pair_obs,        order.data(),       order.size());
    {
        // the (intrinsics, extrinsics) pairs, in the order of `order`
        std::vector<int> obs_pair(Nobs > 0 ? Nobs : 1, 0), pair_rep;
        for(int k=0;k<Nobs;k++)
        {
            const int o = order[k];
            if(k == 0 || meta[o].icam_intrinsics != meta[order[k-1]].icam_intrinsics ||
                         spose_of(meta[o]) != spose_of(meta[order[k-1]]))
                pair_rep.push_back(o);
            obs_pair[o] = (int)pair_rep.size() - 1;
        }
        std::vector<int> chunk_pair(P->plan.Nchunks > 0 ? P->plan.Nchunks : 1, 0);
        for(int c=0;c<P->plan.Nchunks;c++) chunk_pair[c] = obs_pair[order[chunk_begin[c]]];
        P->plan.Npairs = (int)pair_rep.size();

        const int nblk = tile_nblk(L.Ndist), npos = gram_stride(L.Ndist);
        std::vector<int>    tab(npos, 0);
        std::vector<PairOp> ptab((size_t)(pair_rep.empty() ? 1 : pair_rep.size())*npos, PairOp{PAIROP_NONE, 0});
        for(int pos = 0; pos < npos; pos++)
        {
            int i, j; bool diag;
            if(!gram_pos_to_entry(nblk, pos, &i, &j, &diag)) continue;
            tab[pos] = (int)(0x80000000u | (diag ? 0x10000u : 0u) | ((unsigned)i << 8) | (unsigned)j);
            for(size_t ip = 0; ip < pair_rep.size(); ip++)
            {
                const BoardObsMeta& m = meta[pair_rep[ip]];
                const TileColInfo ci = board_tile_col_info(P->D, m, i), cj = board_tile_col_info(P->D, m, j);
                PairOp op = { PAIROP_NONE, 0 };
                const bool fi = ci.kind == COL_FRAME, fj = cj.kind == COL_FRAME;
                const bool si = ci.kind == COL_S,     sj = cj.kind == COL_S;
                const bool xi = ci.kind == COL_X,     xj = cj.kind == COL_X;
                if(fi && fj)
                    op = PairOp{ PAIROP_D | (diag ? 0 : PAIROP_MIRROR), ci.idx | (cj.idx << 16) };
                else if(fi)
                {
                    // (frame, S) or (frame, x). In a diagonal block the mirrored
                    // position carries the same product: it is taken there only
                    if(!diag && sj)      op = PairOp{ PAIROP_BT, ci.idx | (state_to_SE(nd, cj.idx) << 16) };
                    else if(!diag && xj) op = PairOp{ PAIROP_GF, ci.idx };
                }
                else if(fj)
                {
                    if(si)      op = PairOp{ PAIROP_BT, cj.idx | (state_to_SE(nd, ci.idx) << 16) };
                    else if(xi) op = PairOp{ PAIROP_GF, cj.idx };
                }
                else if((si || xi) && (sj || xj) && !(diag && xi && sj))
                {
                    if(xi && xj)   op = PairOp{ PAIROP_NORM, 0 };
                    else if(xj)    op = PairOp{ PAIROP_G, ci.idx };
                    else if(xi)    op = PairOp{ PAIROP_G, cj.idx };
                    else           op = PairOp{ PAIROP_A | (diag ? 0 : PAIROP_MIRROR),
                                                state_to_SE(nd, ci.idx) | (state_to_SE(nd, cj.idx) << 16) };
                }
                ptab[ip*npos + pos] = op;
                const int k = op.op & 0xff;
                if(k == PAIROP_D || k == PAIROP_BT || k == PAIROP_GF) tab[pos] |= 0x20000;
            }
        }
        ok = ok && dev_upload(&P->plan.pos_table,  tab.data(),        tab.size());
        ok = ok && dev_upload(&P->plan.pair_table, ptab.data(),       ptab.size());
        {
            // the frame part, per position and per observation (AssemblyPlan::frame_pos). Derived from the table
            // above and checked against it: every pair's operation at every position must come back out
            const int nintr = (P->D.Nintr_state > 0) ? P->D.Ncameras_intrinsics*P->D.Nintr_state : 0;
            std::vector<int> fpos(npos, FRAMEPOS_NONE);
            std::vector<int> pair_cols(2*(pair_rep.empty() ? 1 : pair_rep.size()), -1);
            auto classify = [&](const PairOp& op, int* kind, int* a, int* k, int* base) -> void
            {
                *kind = FRAMEPOS_NONE; *a = 0; *k = 0; *base = -1;
                const int b = op.aux >> 16;
                switch(op.op & 0xff)
                {
                case PAIROP_D:  *kind = (op.op & PAIROP_MIRROR) ? FRAMEPOS_D_MIRROR : FRAMEPOS_D; *a = op.aux & 0xffff; *k = b; break;
                case PAIROP_GF: *kind = FRAMEPOS_GF; *a = op.aux & 0xffff; break;
                case PAIROP_BT:
                    *a = op.aux & 0xffff;
                    if(b < nintr)        { *kind = FRAMEPOS_BT_INTRINSICS; *base = (b/P->D.Nintr_state)*P->D.Nintr_state; *k = b - *base; }
                    else if(b < nd.Nc - nd.Nwarp) { *kind = FRAMEPOS_BT_EXTRINSICS; *base = nintr + ((b - nintr)/6)*6;     *k = b - *base; }
                    else                 { *kind = FRAMEPOS_BT_WARP; *k = b; }
                    break;
                default: break;
                }
            };
            bool consistent = true;
            for(size_t ip = 0; ip < pair_rep.size(); ip++)
                for(int pos = 0; pos < npos; pos++)
                {
                    int kind, a, k, base;
                    classify(ptab[ip*npos + pos], &kind, &a, &k, &base);
                    if(kind == FRAMEPOS_NONE) continue;
                    const int code = kind | (a << 3) | (k << 6);
                    if(fpos[pos] == FRAMEPOS_NONE) fpos[pos] = code;
                    else if(fpos[pos] != code) consistent = false;
                    if(kind == FRAMEPOS_BT_INTRINSICS || kind == FRAMEPOS_BT_EXTRINSICS)
                    {
                        int& c = pair_cols[2*ip + (kind == FRAMEPOS_BT_EXTRINSICS ? 1 : 0)];
                        if(c < 0) c = base; else if(c != base) consistent = false;
                    }
                }
            // ... and back: a pair without a block (the camera at the reference has no extrinsics) has nothing
            // at that block's positions, and every other position reads the same through both tables
            for(size_t ip = 0; ip < pair_rep.size() && consistent; ip++)
                for(int pos = 0; pos < npos; pos++)
                {
                    // (observations without an eliminated pose - a camera at the reference, with elim_extrinsics -
                    //  are in no block's list: what the frame part's tables say about them is never looked at)
                    if(eblock_of(meta[pair_rep[ip]]) < 0) break;
                    int kind, a, k, base;
                    classify(ptab[ip*npos + pos], &kind, &a, &k, &base);
                    const int fk = fpos[pos] & 7;
                    const bool absent = (fk == FRAMEPOS_BT_INTRINSICS && pair_cols[2*ip] < 0) || (fk == FRAMEPOS_BT_EXTRINSICS && pair_cols[2*ip+1] < 0);
                    if(kind == FRAMEPOS_NONE ? !(fk == FRAMEPOS_NONE || absent) : absent) consistent = false;
                }
            // (the splined models assemble from staged rows, not from Grams: no use for these tables)
            const bool uses_grams = P->D.lens_type != MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC;
            if(!consistent && uses_grams) { set_error("internal: the Gram positions of the frame part depend on the camera pair"); return false; }
            std::vector<int> obs_cols(2*(Nobs > 0 ? Nobs : 1), -1);
            for(int o = 0; o < Nobs; o++) { obs_cols[2*o] = pair_cols[2*obs_pair[o]]; obs_cols[2*o+1] = pair_cols[2*obs_pair[o]+1]; }
            ok = ok && dev_upload(&P->plan.frame_pos, fpos.data(),     fpos.size());
            ok = ok && dev_upload(&P->plan.obs_cols,  obs_cols.data(), obs_cols.size());
        }
        ok = ok && dev_upload(&P->plan.obs_pair,   obs_pair.data(),   obs_pair.size());
        ok = ok && dev_upload(&P->plan.chunk_pair, chunk_pair.data(), chunk_pair.size());

        // The fixed-order reduction of the camera-block part (solver_kernels.hpp):
        // for every destination - entry of A, of g (S part), |x|^2 - the (pair,
        // position) sources that add to it, in (pair, position) order
        // (the splined models have no Grams: nothing is reduced this way)
        const bool with_grams = (L.lensmodel.type != MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC);
        const int Npairs = with_grams ? (int)pair_rep.size() : 0;
        if(with_grams && npos > 1024)
        {
            set_error("internal error: %d Gram positions per observation", npos);
            return false;
        }
        std::vector<int> pair_chunk_begin(Npairs + 1, 0);
        for(int c=0;c<P->plan.Nchunks && Npairs > 0;c++) pair_chunk_begin[chunk_pair[c] + 1]++;
        for(int ip=0;ip<Npairs;ip++) pair_chunk_begin[ip+1] += pair_chunk_begin[ip];
        const int nA = nd.Nc*nd.Nc, Ndest_all = nA + nd.Nc + 1;
        std::vector<std::vector<int>> src(Npairs > 0 ? Ndest_all : 0);
        for(int ip = 0; ip < Npairs; ip++)
            for(int pos = 0; pos < npos; pos++)
            {
                const PairOp op = ptab[(size_t)ip*npos + pos];
                const int k = op.op & 0xff, code = (ip << 10) | pos;
                if(k == PAIROP_NORM) src[nA + nd.Nc].push_back(code);
                else if(k == PAIROP_G)
                {
                    const int sc = state_to_SE(nd, op.aux);     // a camera-block variable: S index >= 0
                    if(sc >= 0) src[nA + sc].push_back(code);
                }
                else if(k == PAIROP_A)
                {
                    const int a = op.aux & 0xffff, b = op.aux >> 16;
                    src[a*nd.Nc + b].push_back(code);
                    if(op.op & PAIROP_MIRROR) src[b*nd.Nc + a].push_back(code);
                }
            }
        std::vector<int> dest_id, dest_begin(1, 0), dest_src;
        for(int d = 0; d < (int)src.size(); d++)
        {
            // |x|^2 is always a destination when there are rows outside the Grams (their partials are added there)
            if(src[d].empty() && d != nA + nd.Nc) continue;
            dest_id.push_back(d);
            dest_src.insert(dest_src.end(), src[d].begin(), src[d].end());
            dest_begin.push_back((int)dest_src.size());
        }
        P->plan.Ndest = (int)dest_id.size();
        if(dest_id.empty())  dest_id.push_back(0);
        if(dest_src.empty()) dest_src.push_back(0);
        ok = ok && dev_upload(&P->plan.dest_id,          dest_id.data(),          dest_id.size());
        ok = ok && dev_upload(&P->plan.dest_begin,       dest_begin.data(),       dest_begin.size());
        ok = ok && dev_upload(&P->plan.dest_src,         dest_src.data(),         dest_src.size());
        ok = ok && dev_upload(&P->plan.pair_chunk_begin, pair_chunk_begin.data(), pair_chunk_begin.size());
        if(with_grams || Nobs == 0)
            ok = ok && dev_alloc(&P->plan.chunk_part, with_grams ? (size_t)(P->plan.Nchunks > 0 ? P->plan.Nchunks : 1)*npos : (size_t)1);
        else
        {
            // splined models: the staged Grams of assemble_splined_kernel (two passes per observation), the knot
            // boxes, and the parts of the rows of the camera block that are not knots (+ the x row)
            const int nknotrows = P->D.Nintr_state > 0 ? P->D.Ncameras_intrinsics*(P->D.Nintr_state - P->D.Ncore_state) : 0;
            ok = ok && dev_alloc(&P->plan.chunk_part,    (size_t)2*Nobs*SPL_TRI);
            ok = ok && dev_alloc(&P->plan.spl_hdr,       (size_t)Nobs);
            ok = ok && dev_alloc(&P->plan.spl_part,   (size_t)(nd.Nc + 1 - nknotrows)*SPLG_E*(nd.Nc + 1));
        }
        ok = ok && build_gen_plan(P);
        {
            // (with the plan above, the row-by-row workgroups of the assembly take the regularization rows only)
            const int row0 = (P->plan.gen.Nrows > 0) ? L.i_meas_regularization : 2*P->D.W*P->D.H*Nobs;
            P->plan.row_part_n = (Nobs > 0 && L.Nmeas > row0) ? (L.Nmeas - row0 + 255)/256 : 0;
            ok = ok && dev_alloc(&P->plan.row_part, (size_t)(P->plan.row_part_n > 0 ? P->plan.row_part_n : 1));
            P->plan.qf_part_n = (nd.Nc + nd.NE + 4*QF_ROWS_PER_WAVE - 1)/(4*QF_ROWS_PER_WAVE);
            ok = ok && dev_alloc(&P->plan.qf_part, (size_t)4*(P->plan.qf_part_n > 0 ? P->plan.qf_part_n : 1));
            ok = ok && dev_alloc(&P->plan.dots_part, (size_t)2*(nd.NEb > 0 ? nd.NEb : 1));
        }
    }
    // (the second, third and fourth sub-boxes of the splined models' close-ups, which mostly nobody touches: last)
    if(P->plan.spl_hdr != NULL)
    {
        const size_t Nobs = (size_t)(P->D.Nobs_board > 0 ? P->D.Nobs_board : 1);
        ok = ok && dev_alloc(&P->plan.chunk_extra,   (size_t)2*Nobs*(SPL_MAXSUB - 1)*SPL_TRI);
        ok = ok && dev_alloc(&P->plan.spl_hdr_extra, Nobs*(SPL_MAXSUB - 1));
    }
    if(!ok) return false;
    // rows of blocks nobody writes (frames without observations in this shard) must read as zero
    for(int i=0;i<2;i++)
    {
        HIP_TRY(hipMemset(P->op[i].Bt, 0, (size_t)(nd.NE*nd.Nc > 0 ? nd.NE*nd.Nc : 1)*sizeof(double)), return false);
        HIP_TRY(hipMemset(P->op[i].D,  0, (size_t)(nd.NEb > 0 ? nd.NEb*36 : 1)*sizeof(double)), return false);
        HIP_TRY(hipMemset(P->op[i].g,  0, (size_t)(nd.Nstate > 0 ? nd.Nstate : 1)*sizeof(double)), return false);
    }

    P->solver_ready = true;
    return true;
}

bool problem_sync_ops(mrcal_amd_problem* P)
{
    if(P->d_ops == NULL && !dev_alloc(&P->d_ops, 2)) return false;
    OpDev h[2] = { P->op[0], P->op[1] };
    HIP_TRY(hipMemcpy(P->d_ops, h, sizeof(h), hipMemcpyHostToDevice), return false);
    return true;
}

bool problem_evaluate_ref(mrcal_amd_problem* P, const OpRef& R, bool with_jacobian, bool with_normal, int parts,
                          hipStream_t stream, const ChooseArgs* choose)
{
    if(with_normal && !P->solver_ready) { set_error("solver buffers are not allocated"); return false; }
    if(stream == NULL) stream = P->stream;
    EvalBuffers B = P->eval_buffers(R, with_normal);
    if(choose != NULL && !((parts & EVAL_PART_PROLOGUE) && prologue_takes_choose(P->D)))
    {
        set_error("internal error: this evaluation has no prologue launch to choose the trial point in");
        return false;
    }
    B.choose = choose;
    // (round 6) a solve that was told to leave the Jacobian stream out: only the solver's own evaluations, and only
    // where the board kernel is the rows' one reader (never the splined models, whose assembly reads them back)
    if(P->jfree_now && with_normal && with_jacobian && P->D.Nobs_board > 0 &&
       P->D.lens_type != MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC)
    {
        B.store_jacobian = false;
        if(parts & EVAL_PART_BOARD) P->jacobian_stale = true;
    }
    if(with_normal && (parts & EVAL_PART_ZERO))
    {
        if((parts & EVAL_PART_PROLOGUE) && P->D.Nobs_board > 0)
        {
            // the prologue kernel clears the normal equations on the side
            const NormalDims& nd = P->nd;
            B.zero_n[0] = (long long)nd.Nc*nd.Nc; B.zero_n[1] = (long long)nd.NE*nd.Nc; B.zero_n[2] = (long long)nd.NEb*36;
            B.zero_n[3] = nd.Nstate;               B.zero_n[4] = NSCALARS;
            B.zero_total = B.zero_n[0] + B.zero_n[1] + B.zero_n[2] + B.zero_n[3] + B.zero_n[4];
        }
        else
            HIP_TRY(launch_zero_normal(P->nd, R, stream), return false);
    }
    // An event pair around the Jacobian kernel costs ~5.6 us on EACH side of it on the stream (measured: the
    // gaps prologue -> kernel -> assembly in a kernel trace; every other boundary of the step is back to
    // back). So: none inside the solver's steps unless the benchmark asked for timings, and then only around
    // every ev_pool_stride-th launch; a host-driven evaluate() keeps its pair (last_jacobian_kernel_ms())
    hipEvent_t e0 = NULL, e1 = NULL;
    if(with_jacobian && (parts & EVAL_PART_BOARD) && !P->capturing)
    {
        if(P->ev_pool_enabled)
        {
            if((P->ev_pool_seen++ % P->ev_pool_stride) == 0 && P->ev_pool_used + 2 <= (int)P->ev_pool.size())
            {
                e0 = P->ev_pool[P->ev_pool_used++];
                e1 = P->ev_pool[P->ev_pool_used++];
            }
        }
        else if(parts == EVAL_PART_ALL) { e0 = P->ev_j0; e1 = P->ev_j1; }
    }
    HIP_TRY(launch_evaluate(P->D, B, with_jacobian, P->lds_bytes, stream, e0, e1, parts),
            return false);
    if(parts & EVAL_PART_BOARD)
        P->have_jacobian_timing = with_jacobian && P->D.Nobs_board > 0 && e0 != NULL;
    if(with_normal && (parts & EVAL_PART_ASSEMBLE))
        HIP_TRY(launch_assemble(P->D, P->nd, P->br, P->plan, B, stream), return false);
    return true;
}

bool problem_ensure_jacobian(mrcal_amd_problem* P)
{
    if(!P->jacobian_stale) return true;
    if(!problem_evaluate_op(P, P->icur, true, false)) return false;
    HIP_TRY(hipStreamSynchronize(P->stream), return false);
    return true;
}

bool problem_evaluate_op(mrcal_amd_problem* P, int i, bool with_jacobian, bool with_normal)
{
    if(!problem_evaluate_ref(P, P->opref(i), with_jacobian, with_normal, EVAL_PART_ALL)) return false;
    // (a host-driven evaluation always streams J: jfree_now is up only inside the solver's entry points)
    if(with_jacobian && i == P->icur) P->jacobian_stale = false;
    P->stats.Nevaluations++;
    if(with_normal) P->op[i].have_normal = true;
    return true;
}

} // namespace mrcal_amd

namespace { int& elimination_policy() { static int policy = 0; return policy; } }

// (round 6) The drop-in entry points make a problem, use it once and tear it down: some forty hipFree() calls, each of
// which waits for the device - 4 ms at the metric's size, a tenth of an mrcal_optimize() call. A problem that nobody can
// reach any more is torn down by a thread of its own instead, while the caller already has its results; at most two
// are in line at a time (a caller in a loop does not pile up gigabytes: the third it tears down itself, as before)
namespace mrcal_amd {
class ProblemReaper
{
    std::mutex m; std::condition_variable cv, cv_idle;
    std::deque<mrcal_amd_problem*> q;
    bool busy = false;
    pid_t owner = 0;        // the process the thread was started in (a fork()ed child has the object and no thread)
    int  device = 0;
    void run()
    {
        (void)hipSetDevice(device);
        for(;;)
        {
            mrcal_amd_problem* P = NULL;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return !q.empty(); });
                P = q.front(); q.pop_front(); busy = true;
            }
            delete P;
            { std::lock_guard<std::mutex> lk(m); busy = false; }
            cv_idle.notify_all();
        }
    }
public:
    // (never destroyed: its thread waits on it for as long as the process lives. What IS done when the process ends:
    //  what is in line is torn down before the runtime's own destructors run - Drain)
    static ProblemReaper& get()
    {
        static ProblemReaper* r = new ProblemReaper;
        static struct Drain { ProblemReaper* r; ~Drain() { r->drain(); } } d{r};
        return *r;
    }
    void later(mrcal_amd_problem* P)
    {
        if(P == NULL) return;
        {
            std::unique_lock<std::mutex> lk(m);
            if(owner != getpid())
            {
                q.clear(); busy = false; owner = getpid();
                (void)hipGetDevice(&device);
                std::thread([this] { run(); }).detach();
            }
            // (never a wait for the thread here: with two in line already the caller tears this one down itself)
            if(q.size() + (busy ? 1 : 0) < 2) { q.push_back(P); P = NULL; }
        }
        if(P == NULL) cv.notify_one();
        else          delete P;
    }
    void drain()
    {
        std::unique_lock<std::mutex> lk(m);
        if(owner == getpid()) cv_idle.wait_for(lk, std::chrono::seconds(5), [&] { return q.empty() && !busy; });
    }
};
void problem_destroy_later(mrcal_amd_problem* P) { ProblemReaper::get().later(P); }
} // namespace mrcal_amd

extern "C" {

// 0: the library chooses (default); 1: the frames and points are eliminated; 2: the extrinsics, where the problem
// allows it. For the problems created AFTER the call. Returns the previous setting
int mrcal_amd_set_elimination(int policy)
{
    const int old = elimination_policy();
    if(policy >= 0 && policy <= 2) elimination_policy() = policy;
    return old;
}

const char* mrcal_amd_last_error(void)
{
    return last_error_string().c_str();
}

int mrcal_amd_device_count(void)
{
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

mrcal_amd_problem_t*
mrcal_amd_problem_create_sharded(const double*                 intrinsics,
                         const mrcal_pose_t*           rt_cam_ref,
                         const mrcal_pose_t*           rt_ref_frame,
                         const mrcal_point3_t*         points,
                         const mrcal_calobject_warp_t* calobject_warp,
                         int Ncameras_intrinsics, int Ncameras_extrinsics, int Nframes,
                         int Npoints, int Npoints_fixed,
                         const mrcal_observation_board_t* observations_board,
                         const mrcal_observation_point_t* observations_point,
                         int Nobservations_board,
                         int Nobservations_point,
                         const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                         int Nobservations_point_triangulated,
                         const mrcal_point3_t* observations_board_pool,
                         const mrcal_point3_t* observations_point_pool,
                         const mrcal_lensmodel_t* lensmodel,
                         const int* imagersizes,
                         mrcal_problem_selections_t problem_selections,
                         double calibration_object_spacing,
                         int calibration_object_width_n,
                         int calibration_object_height_n,
                         int shard_begin_frame, int shard_end_frame,
                         int shard_begin_point, int shard_end_point,
                         int shard_begin_tripoint, int shard_end_tripoint,
                         bool is_shard_leader)
{
    last_error_string().clear();

    if(mrcal_amd_device_count() <= 0)
    {
        set_error("no HIP device is visible: libmrcal_amd has no CPU fallback");
        return NULL;
    }
    if(!lens_supported(lensmodel->type))
    {
        char name[128] = "?";
        mrcal_lensmodel_name(name, sizeof(name), lensmodel);
        set_error("lens model %s (%d) is not implemented on the GPU yet", name, (int)lensmodel->type);
        return NULL;
    }
    if(observations_point_triangulated == NULL || Nobservations_point_triangulated <= 0)
    {
        observations_point_triangulated = NULL;
        Nobservations_point_triangulated = 0;
    }
    if(Nobservations_board > 0 &&
       (calibration_object_width_n <= 0 || calibration_object_height_n <= 0))
    {
        set_error("board observations given, but the board has no corners");
        return NULL;
    }

    if(Nobservations_board <= 0) { Nobservations_board = 0; calibration_object_width_n = calibration_object_height_n = 0; }
    if(Nobservations_point <= 0)   Nobservations_point = 0;

    const mrcal_problem_selections_t sel =
        effective_selections(problem_selections, *lensmodel, Nobservations_board);

    mrcal_amd_problem* P = new mrcal_amd_problem();

    // The STATE layout is global: every shard sees the whole state vector
    Dims dg;
    dg.Ncameras_intrinsics = Ncameras_intrinsics;
    dg.Ncameras_extrinsics = Ncameras_extrinsics;
    dg.Nframes             = Nframes;
    dg.Npoints             = Npoints;
    dg.Npoints_fixed       = Npoints_fixed;
    dg.Nobservations_board = Nobservations_board;
    dg.Nobservations_point = Nobservations_point;
    dg.object_width_n      = calibration_object_width_n;
    dg.object_height_n     = calibration_object_height_n;
    const Layout Lg = make_layout(dg, sel, *lensmodel, observations_point_triangulated, Nobservations_point_triangulated);

    // The MEASUREMENT layout is local to the shard
    // shard_end_frame < 0: the whole problem. Anything else is a shard, even an
    // empty frame range (a points-only problem under the multi-GPU driver gives
    // every rank the range (0,0)): only the leader then owns the points, the
    // triangulated pairs and the regularization rows
    const bool sharded = shard_end_frame >= 0;
    if(!sharded) is_shard_leader = true;
    std::vector<int> board_sel;
    board_sel.reserve(Nobservations_board);
    for(int i=0; i<Nobservations_board; i++)
    {
        const int f = observations_board[i].iframe;
        if(!sharded || (f >= shard_begin_frame && f < shard_end_frame))
            board_sel.push_back(i);
    }
    const int Nboard_local = (int)board_sel.size();
    // discrete points: the shard owns the points [shard_begin_point, shard_end_point) (their 3x3 blocks of JtJ,
    // their rows of x and J) wherever their observations sit in the caller's array (the API does
    // not promise point-sorted observations). shard_end_point < 0: all of them on the leader, none elsewhere
    if(!sharded || shard_end_point < 0) { shard_begin_point = 0; shard_end_point = (!sharded || is_shard_leader) ? Npoints : 0; }
    std::vector<int> point_sel;
    for(int i=0; i<Nobservations_point; i++)
    {
        const int ip = observations_point[i].i_point;
        if(ip >= shard_begin_point && ip < shard_end_point) point_sel.push_back(i);
    }
    const int Npoint_local = (int)point_sel.size();
    // triangulated points: a point's observations are consecutive (last_in_set ends the set) and its pairs are its
    // own, so the shard takes the point SETS [shard_begin_tripoint, shard_end_tripoint): one contiguous range of
    // observations [tri_o0, tri_o1). < 0: all with the leader
    int tri_o0 = 0, tri_o1 = 0;
    {
        if(!sharded || shard_end_tripoint < 0) { shard_begin_tripoint = 0; shard_end_tripoint = (!sharded || is_shard_leader) ? 0x7fffffff : 0; }
        int iset = 0;
        bool in_range = false;
        for(int i=0; i<Nobservations_point_triangulated; i++)
        {
            const bool mine = iset >= shard_begin_tripoint && iset < shard_end_tripoint;
            if(mine && !in_range) { tri_o0 = i; in_range = true; }
            if(mine) tri_o1 = i + 1;
            if(observations_point_triangulated[i].last_in_set) iset++;
        }
        if(!in_range) tri_o0 = tri_o1 = 0;
    }
    const int Ntri_local = tri_o1 - tri_o0;
    const mrcal_observation_point_triangulated_t* tri_local = (Ntri_local > 0) ? observations_point_triangulated + tri_o0 : NULL;

    Layout L = Lg;
    L.dims.Nobservations_board = Nboard_local;   // NOTE: has_warp etc. stay global
    L.dims.Nobservations_point = Npoint_local;
    L.Nmeas_boards         = Nboard_local * calibration_object_width_n*calibration_object_height_n * 2;
    L.Nmeas_points         = Npoint_local * 2;
    L.Nmeas_triangulated   = num_measurements_triangulated_initial(tri_local, Ntri_local, -1);
    if(!is_shard_leader) { L.Nmeas_regularization = 0; L.has_unity_cam01 = false; L.Nreg_percamera = 0; }
    L.i_meas_boards         = 0;
    L.i_meas_points         = L.Nmeas_boards;
    L.i_meas_triangulated   = L.i_meas_points + L.Nmeas_points;
    L.i_meas_regularization = L.i_meas_triangulated + L.Nmeas_triangulated;
    L.Nmeas                 = L.i_meas_regularization + L.Nmeas_regularization;
    P->L = L;

    const int NPTS = calibration_object_width_n*calibration_object_height_n;

    // per-observation metadata + CSR offsets
    std::vector<BoardObsMeta> bmeta(Nboard_local);
    int64_t innz = 0;
    int     imeas = 0;
    int     kmax  = 0;
    for(int j=0; j<Nboard_local; j++)
    {
        const mrcal_observation_board_t& o = observations_board[board_sel[j]];
        BoardObsMeta& m = bmeta[j];
        memset(&m, 0, sizeof(m));
        m.icam_intrinsics    = o.icam.intrinsics;
        m.icam_extrinsics    = o.icam.extrinsics;
        m.iframe             = o.iframe;
        m.nnz_per_row        = nnz_per_board_row(L, o.icam.extrinsics);
        m.i_state_intrinsics = (L.Nintr_state > 0) ? L.i_state_intrinsics + o.icam.intrinsics*L.Nintr_state : -1;
        m.i_state_extrinsics = (L.Nstate_extrinsics > 0 && o.icam.extrinsics >= 0) ? L.i_state_extrinsics + 6*o.icam.extrinsics : -1;
        m.i_state_frame      = (L.Nstate_frames > 0) ? L.i_state_frames + 6*o.iframe : -1;
        m.i_meas0            = imeas;
        m.i_nnz0             = innz;
        imeas += 2*NPTS;
        innz  += (int64_t)2*NPTS*m.nnz_per_row;
        if(m.nnz_per_row > kmax) kmax = m.nnz_per_row;
    }
    // per board observation 24 P (read qx,qy,w) + 16 P (write x) + 16 P k (write J values)
    P->board_alg_bytes = (int64_t)Nboard_local*NPTS*(24 + 16) + innz*8;
    std::vector<PointObsMeta> pmeta(Npoint_local);
    for(int j=0; j<Npoint_local; j++)
    {
        const mrcal_observation_point_t& o = observations_point[point_sel[j]];
        PointObsMeta& m = pmeta[j];
        memset(&m, 0, sizeof(m));
        const bool variable = sel.do_optimize_frames && o.i_point < Npoints - Npoints_fixed;
        m.icam_intrinsics    = o.icam.intrinsics;
        m.icam_extrinsics    = o.icam.extrinsics;
        m.i_point            = o.i_point;
        m.nnz_per_row        = nnz_per_point_row(L, o.icam.extrinsics, o.i_point);
        m.i_state_intrinsics = (L.Nintr_state > 0) ? L.i_state_intrinsics + o.icam.intrinsics*L.Nintr_state : -1;
        m.i_state_extrinsics = (L.Nstate_extrinsics > 0 && o.icam.extrinsics >= 0) ? L.i_state_extrinsics + 6*o.icam.extrinsics : -1;
        m.i_state_point      = variable ? L.i_state_points + 3*o.i_point : -1;
        m.i_meas0            = imeas;
        m.i_nnz0             = innz;
        imeas += 2;
        innz  += 2*m.nnz_per_row;
    }
    // triangulated points: one row per pair (i0 < i1) of observations of a point
    std::vector<TriPairMeta> tmeta;
    if(L.Nmeas_triangulated > 0)
    {
        const mrcal_observation_point_triangulated_t* ot = tri_local;      // (indices local to the shard's range)
        for(int i0 = 0; i0 < Ntri_local; i0++)
        {
            if(ot[i0].last_in_set) continue;
            for(int i1 = i0+1; i1 < Ntri_local; i1++)
            {
                TriPairMeta m;
                memset(&m, 0, sizeof(m));
                m.i0 = i0; m.i1 = i1;
                m.icam_extrinsics0 = ot[i0].icam.extrinsics;
                m.icam_extrinsics1 = ot[i1].icam.extrinsics;
                m.i_state_extrinsics0 = (L.Nstate_extrinsics > 0 && m.icam_extrinsics0 >= 0) ? L.i_state_extrinsics + 6*m.icam_extrinsics0 : -1;
                m.i_state_extrinsics1 = (L.Nstate_extrinsics > 0 && m.icam_extrinsics1 >= 0) ? L.i_state_extrinsics + 6*m.icam_extrinsics1 : -1;
                m.i_meas = imeas;
                m.i_nnz0 = innz;
                imeas += 1;
                innz  += (m.i_state_extrinsics0 >= 0 ? 6 : 0) + (m.i_state_extrinsics1 >= 0 ? 6 : 0);
                tmeta.push_back(m);
                if(ot[i1].last_in_set) break;
            }
        }
        if((int)tmeta.size() != L.Nmeas_triangulated)
        {
            set_error("internal error: %d triangulated pairs, the layout says %d", (int)tmeta.size(), L.Nmeas_triangulated);
            delete P;
            return NULL;
        }
    }
    const int64_t innz_reg = innz;
    if(L.Nmeas_regularization > 0)
    {
        // one nonzero per row, except the splined models' knot rows, which have 2 (mrcal.c:847-869)
        if(lensmodel->type == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC)
            innz += (int64_t)Ncameras_intrinsics*(L.Nreg_percamera + (sel.do_apply_regularization ? L.Ndist_state : 0));
        else
            innz += (int64_t)Ncameras_intrinsics*L.Nreg_percamera;
        innz += (L.has_unity_cam01 ? 3 : 0);
    }
    P->Nnz = innz;
    if(innz > 0x7fffffffLL)
    {
        // the reference's CSR uses int32 offsets (cholmod itype int); so do we
        set_error("Jacobian has %lld nonzeros: more than int32 CSR offsets can address. Shard the problem", (long long)innz);
        delete P;
        return NULL;
    }
    // tile columns: k, +2 for the full core, +1 for the residual column (see board_kernel)
    // LDS of the board kernel: the 64-row tile + the observation's pixels and
    // weights (the splined models' kernels use none)
    const bool splined = (lensmodel->type == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC);
    // the tile + the staged observation (in whole 64-element chunks) + the joint pose record
    P->lds_bytes = splined ? 0 : (64*tile_stride(L.Ndist) + ((3*NPTS + 63) & ~63) + JOINT_REC + 4) * (int)sizeof(double);
    if(P->lds_bytes > 160*1024)
    {
        set_error("the board has %d corners and the lens model %d distortion parameters: the LDS tile would not fit", NPTS, L.Ndist);
        delete P;
        return NULL;
    }
    (void)kmax;

    // board pool of the local observations
    std::vector<mrcal_point3_t> pool_local;
    const mrcal_point3_t* pool_src = observations_board_pool;
    if(sharded)
    {
        pool_local.resize((size_t)Nboard_local*NPTS);
        for(int j=0; j<Nboard_local; j++)
            memcpy(&pool_local[(size_t)j*NPTS], &observations_board_pool[(size_t)board_sel[j]*NPTS],
                   NPTS*sizeof(mrcal_point3_t));
        pool_src = pool_local.data();
    }
    P->board_sel = board_sel;
    std::vector<mrcal_point3_t> point_pool_local((size_t)(Npoint_local > 0 ? Npoint_local : 1));
    for(int j=0; j<Npoint_local; j++) point_pool_local[j] = observations_point_pool[point_sel[j]];

    bool ok = true;
    HIP_TRY(hipStreamCreateWithFlags(&P->stream, hipStreamNonBlocking), ok = false);
    HIP_TRY(hipStreamCreateWithFlags(&P->side_stream, hipStreamNonBlocking), ok = false);
    HIP_TRY(hipEventCreateWithFlags(&P->ev_fork, hipEventDisableTiming), ok = false);
    HIP_TRY(hipEventCreateWithFlags(&P->ev_join, hipEventDisableTiming), ok = false);
    HIP_TRY(hipEventCreate(&P->ev_j0), ok = false);
    HIP_TRY(hipEventCreate(&P->ev_j1), ok = false);

    ok = ok && dev_upload(&P->d_seed_intrinsics,   intrinsics,                  (size_t)Ncameras_intrinsics*L.Nintrinsics);
    ok = ok && dev_upload(&P->d_seed_rt_cam_ref,   (const double*)rt_cam_ref,   (size_t)Ncameras_extrinsics*6);
    ok = ok && dev_upload(&P->d_seed_rt_ref_frame, (const double*)rt_ref_frame, (size_t)Nframes*6);
    ok = ok && dev_upload(&P->d_seed_points,       (const double*)points,       (size_t)Npoints*3);
    ok = ok && dev_upload(&P->d_board_meta,        bmeta.data(),                (size_t)Nboard_local);
    ok = ok && dev_upload(&P->d_board_pool,        (const double*)pool_src,     (size_t)Nboard_local*NPTS*3);
    ok = ok && dev_upload(&P->d_point_meta,        pmeta.data(),                (size_t)Npoint_local);
    ok = ok && dev_upload(&P->d_point_pool,        (const double*)point_pool_local.data(), (size_t)Npoint_local*3);
    ok = ok && dev_upload(&P->d_imagersizes,       imagersizes,                 (size_t)Ncameras_intrinsics*2);
    {
        const int Nt = Ntri_local;
        P->tri_obs0 = tri_o0;
        P->tri_meta_host = tmeta;
        P->tri_px_host.resize((size_t)3*Nt + 1);
        P->tri_outlier_host.resize((size_t)Nt + 1);
        for(int i=0;i<Nt;i++)
        {
            for(int j=0;j<3;j++) P->tri_px_host[3*i+j] = tri_local[i].px.xyz[j];
            P->tri_outlier_host[i] = tri_local[i].outlier ? 1 : 0;
        }
        ok = ok && dev_upload(&P->d_tri_meta,    tmeta.data(),                tmeta.size());
        ok = ok && dev_upload(&P->d_tri_px,      P->tri_px_host.data(),       (size_t)3*Nt);
        ok = ok && dev_upload(&P->d_tri_outlier, P->tri_outlier_host.data(),  (size_t)Nt);
    }
    ok = ok && dev_alloc (&P->op[0].b,  (size_t)L.Nstate);
    // + the unpacked intrinsics and warp (DeviceProblem::unpacked)
    ok = ok && dev_alloc (&P->d_joint,  (size_t)Nboard_local*JOINT_STRIDE + (size_t)Ncameras_intrinsics*L.Nintrinsics + 2);
    ok = ok && dev_alloc (&P->op[0].x,  (size_t)L.Nmeas);
    ok = ok && dev_alloc (&P->op[0].Jv, (size_t)innz);
    if(L.lensmodel.type == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC && Nboard_local > 0)
        ok = ok && dev_alloc(&P->op[0].spl_box, (size_t)4*Nboard_local);
    ok = ok && dev_alloc (&P->d_Jp,     (size_t)L.Nmeas+1);
    ok = ok && dev_alloc (&P->d_Ji,     (size_t)innz);
    if(!ok) { delete P; return NULL; }
    if(!problem_sync_ops(P)) { delete P; return NULL; }

    {
        NormalDims& nd = P->nd;
        nd.Nstate       = L.Nstate;
        nd.Nwarp        = L.Nstate_warp;
        nd.i_state_warp = L.i_state_warp;
        // Which pose blocks are eliminated (NormalDims, solver_kernels.hpp): the frames and points, unless the
        // extrinsics are the numerous ones - a moving camera against a stationary board, many rt_cam_ref and
        // few frames (test_calibration_helpers.py:422-493 builds such problems) - and every row touches at most
        // one of them, and only board rows touch them: a camera's block is then written whole by the workgroup
        // that sums its observations' Grams, as a frame's is (no triangulated pairs, no discrete points, no
        // unity_cam01 row). The splined models' assembly and the sharding know frames only.
        // mrcal_amd_set_elimination() overrides the choice where both are possible; without a call the environment
        // variable MRCAL_AMD_ELIMINATE=frames|extrinsics does (for a process that cannot make one)
        bool elimx = !sharded && lensmodel->type != MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC && Ntri_local == 0 &&
                     Npoint_local == 0 && !sel.do_apply_regularization_unity_cam01 && L.Nstate_extrinsics > 0;
        if(elimx)
        {
            const char* env = getenv("MRCAL_AMD_ELIMINATE");
            if(elimination_policy() == 1)              elimx = false;
            else if(elimination_policy() == 2)         elimx = true;
            else if(env && !strcmp(env, "frames"))     elimx = false;
            else if(env && !strcmp(env, "extrinsics")) elimx = true;
            else elimx = Ncameras_extrinsics >= 4 && L.Nstate_frames + L.Nstate_points < L.Nstate_extrinsics;
        }
        if(!elimx)
        {
            nd.Nc  = L.Nstate_intrinsics + L.Nstate_extrinsics + nd.Nwarp;
            nd.NE  = L.Nstate_frames + L.Nstate_points;
            nd.Nfb = L.Nstate_frames/6;
            nd.Npb = L.Nstate_points/3;
            normal_dims_set_partition(nd, L.Nstate_intrinsics + L.Nstate_extrinsics);
        }
        else
        {
            nd.NE  = L.Nstate_extrinsics;
            nd.Nc  = L.Nstate - nd.NE;
            nd.Nfb = L.Nstate_extrinsics/6;
            nd.Npb = 0;
            nd.S_split = L.Nstate_intrinsics; nd.S_shift = nd.NE; nd.E_state0 = L.Nstate_intrinsics;
            nd.elim_extrinsics = 1;
        }
        nd.NEb          = nd.Nfb + nd.Npb;
        P->is_leader    = is_shard_leader;
        P->br.frame_lo  = 0;  P->br.frame_hi = nd.Nfb;
        if(sharded && nd.Nfb > 0)
        {
            P->br.frame_lo = shard_begin_frame < 0 ? 0 : shard_begin_frame;
            P->br.frame_hi = (shard_end_frame < 0 || shard_end_frame > Nframes) ? Nframes : shard_end_frame;
        }
        // the point blocks (the variable points only) of the shard's point range
        {
            auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
            P->br.point_lo = nd.Nfb + clampi(shard_begin_point, 0, nd.Npb);
            P->br.point_hi = nd.Nfb + clampi(shard_end_point,   0, nd.Npb);
            if(P->br.point_hi < P->br.point_lo) P->br.point_hi = P->br.point_lo;
        }
    }

    DeviceProblem& D = P->D;
    memset(&D, 0, sizeof(D));
    D.lens_type   = (int)lensmodel->type;
    D.Nintrinsics = L.Nintrinsics;   D.Ncore = L.Ncore;        D.Ncore_state = L.Ncore_state;
    D.Ndist       = L.Ndist;         D.Ndist_state = L.Ndist_state; D.Nintr_state = L.Nintr_state;
    D.Ndist_row   = L.Nintr_per_row - (L.Ncore_state ? 2 : 0);
    D.i_state_intrinsics = L.i_state_intrinsics < 0 ? 0 : L.i_state_intrinsics;
    D.i_state_extrinsics = L.i_state_extrinsics;
    D.i_state_frames     = L.i_state_frames;
    D.i_state_points     = L.i_state_points;
    D.i_state_warp       = L.i_state_warp;
    D.Nstate = L.Nstate;  D.Nmeas = L.Nmeas;
    D.do_optimize_extrinsics = L.Nstate_extrinsics > 0;
    D.do_optimize_frames     = sel.do_optimize_frames;
    D.elim_extrinsics        = P->nd.elim_extrinsics;
    D.has_warp_state         = L.has_warp;
    D.has_warp_seed          = (calobject_warp != NULL);
    D.Ncameras_intrinsics = Ncameras_intrinsics; D.Ncameras_extrinsics = Ncameras_extrinsics;
    D.Nframes = Nframes; D.Npoints = Npoints; D.Npoints_fixed = Npoints_fixed;
    D.Nobs_board = Nboard_local; D.Nobs_point = Npoint_local;
    D.W = calibration_object_width_n; D.H = calibration_object_height_n;
    D.spacing = calibration_object_spacing;
    D.inv_Wm1 = 1.0/(double)(D.W - 1);      // (W = 1 or H = 1 with a warp: the reference divides by zero just the same)
    D.inv_Hm1 = 1.0/(double)(D.H - 1);
    if(calobject_warp) { D.seed_warp[0] = calobject_warp->x2; D.seed_warp[1] = calobject_warp->y2; }
    if(lensmodel->type == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC)
    {
        D.cfg.spline_order = lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.order;
        D.cfg.spline_Nx    = lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.Nx;
        D.cfg.spline_Ny    = lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.Ny;
        D.cfg.spline_segments_per_u =
            spline_segments_per_u(D.cfg.spline_order, D.cfg.spline_Nx,
                                  (double)lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.fov_x_deg);
    }
    if(lensmodel->type == MRCAL_LENSMODEL_CAHVORE)
        D.cfg.cahvore_linearity = lensmodel->LENSMODEL_CAHVORE__config.linearity;
    D.do_apply_regularization = sel.do_apply_regularization && is_shard_leader;
    D.has_unity_cam01         = L.has_unity_cam01;
    D.i_meas_regularization   = L.i_meas_regularization;
    D.i_nnz_regularization    = innz_reg;
    D.imager_width_cam0       = (Ncameras_intrinsics > 0) ? (double)imagersizes[0] : 1.0;
    D.seed_intrinsics   = P->d_seed_intrinsics;
    D.seed_rt_cam_ref   = P->d_seed_rt_cam_ref;
    D.seed_rt_ref_frame = P->d_seed_rt_ref_frame;
    D.seed_points       = P->d_seed_points;
    D.board_meta        = P->d_board_meta;
    D.board_pool        = P->d_board_pool;
    D.point_meta        = P->d_point_meta;
    D.point_pool        = P->d_point_pool;
    D.imagersizes       = P->d_imagersizes;
    D.Npairs_tri        = (int)P->tri_meta_host.size();
    D.tri_meta          = P->d_tri_meta;
    D.tri_px            = P->d_tri_px;
    D.tri_outlier       = P->d_tri_outlier;
    D.unpacked          = P->d_joint + (size_t)Nboard_local*JOINT_STRIDE;
    // (round 6) where the triangulated pairs ride in the board kernel's launch (board_tri_kernel) the launch the
    // benchmark times carries their bytes too: per pair two observation vectors and the record read, x and the
    // (up to) 12 partials written
    if(board_launch_takes_triangulated(D))
        for(const TriPairMeta& m : P->tri_meta_host)
            P->board_alg_bytes += 2*24 + (int64_t)sizeof(TriPairMeta) + 8 + 8*((m.i_state_extrinsics0 >= 0 ? 6 : 0) + (m.i_state_extrinsics1 >= 0 ? 6 : 0));

    // the seed state
    P->b_host.assign(L.Nstate > 0 ? L.Nstate : 1, 0.0);
    pack_state_from_arrays(P->b_host.data(), L, intrinsics, rt_cam_ref, rt_ref_frame, points, calobject_warp);
    HIP_TRY(hipMemcpyAsync(P->op[0].b, P->b_host.data(), (size_t)L.Nstate*sizeof(double),
                           hipMemcpyHostToDevice, P->stream), { delete P; return NULL; });

    // iteration-invariant CSR structure
    HIP_TRY(launch_structure(D, P->eval_buffers(0,false), P->stream), { delete P; return NULL; });
    if(L.Nmeas_regularization <= 0)
    {
        const int32_t last = (int32_t)innz;
        HIP_TRY(hipMemcpyAsync(&P->d_Jp[L.Nmeas], &last, sizeof(last), hipMemcpyHostToDevice, P->stream),
                { delete P; return NULL; });
    }
    HIP_TRY(hipStreamSynchronize(P->stream), { delete P; return NULL; });
    return P;
}

mrcal_amd_problem_t*
mrcal_amd_problem_create(const double*                 intrinsics,
                         const mrcal_pose_t*           rt_cam_ref,
                         const mrcal_pose_t*           rt_ref_frame,
                         const mrcal_point3_t*         points,
                         const mrcal_calobject_warp_t* calobject_warp,
                         int Ncameras_intrinsics, int Ncameras_extrinsics, int Nframes,
                         int Npoints, int Npoints_fixed,
                         const mrcal_observation_board_t* observations_board,
                         const mrcal_observation_point_t* observations_point,
                         int Nobservations_board,
                         int Nobservations_point,
                         const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                         int Nobservations_point_triangulated,
                         const mrcal_point3_t* observations_board_pool,
                         const mrcal_point3_t* observations_point_pool,
                         const mrcal_lensmodel_t* lensmodel,
                         const int* imagersizes,
                         mrcal_problem_selections_t problem_selections,
                         double calibration_object_spacing,
                         int calibration_object_width_n,
                         int calibration_object_height_n,
                         int shard_begin_frame, int shard_end_frame,
                         bool is_shard_leader)
{
    // the shard leader owns every discrete and triangulated point
    return mrcal_amd_problem_create_sharded(intrinsics, rt_cam_ref, rt_ref_frame, points, calobject_warp,
                                            Ncameras_intrinsics, Ncameras_extrinsics, Nframes, Npoints, Npoints_fixed,
                                            observations_board, observations_point, Nobservations_board, Nobservations_point,
                                            observations_point_triangulated, Nobservations_point_triangulated,
                                            observations_board_pool, observations_point_pool, lensmodel, imagersizes,
                                            problem_selections, calibration_object_spacing,
                                            calibration_object_width_n, calibration_object_height_n,
                                            shard_begin_frame, shard_end_frame, 0, -1, 0, -1, is_shard_leader);
}

void mrcal_amd_problem_destroy(mrcal_amd_problem_t* problem)
{
    delete problem;
}

int     mrcal_amd_problem_Nstate       (const mrcal_amd_problem_t* p) { return p->L.Nstate; }
int     mrcal_amd_problem_Nmeasurements(const mrcal_amd_problem_t* p) { return p->L.Nmeas;  }
int64_t mrcal_amd_problem_Nnz          (const mrcal_amd_problem_t* p) { return p->Nnz;      }
int64_t mrcal_amd_problem_jacobian_algorithmic_bytes(const mrcal_amd_problem_t* p) { return p->board_alg_bytes; }
bool    mrcal_amd_problem_synchronize  (mrcal_amd_problem_t* p)
{
    HIP_TRY(hipStreamSynchronize(p->stream), return false);
    return true;
}

double*  mrcal_amd_problem_dev_b_packed(mrcal_amd_problem_t* p) { return p->op[p->icur].b;  }
double*  mrcal_amd_problem_dev_x       (mrcal_amd_problem_t* p) { return p->op[p->icur].x;  }
int32_t* mrcal_amd_problem_dev_J_rowptr(mrcal_amd_problem_t* p) { return p->d_Jp; }
int32_t* mrcal_amd_problem_dev_J_colidx(mrcal_amd_problem_t* p) { return p->d_Ji; }
double*  mrcal_amd_problem_dev_J_values(mrcal_amd_problem_t* p) { return problem_ensure_jacobian(p) ? p->op[p->icur].Jv : NULL; }
// (round 6) stream != 0 (the default): every evaluation of the solver writes the CSR values of J to HBM, as the
// metric defines a step and as a caller who reads J between steps needs it. 0: mrcal_amd_problem_solve()
// / _run_steps() leave the stream out where nothing in the solve reads it (boards under a parametric lens model): the
// same x, b_packed, outliers - the same bits -, J made on demand (_get_J(), _dev_J_values(), _evaluate()) afterwards.
// Returns the previous setting
int mrcal_amd_problem_set_jacobian_stream(mrcal_amd_problem_t* p, int stream)
{
    const int old = p->solve_stores_jacobian ? 1 : 0;
    if((stream != 0) != p->solve_stores_jacobian)
    {
        p->solve_stores_jacobian = (stream != 0);
        // (the captured trial step has the board kernel's variant in it)
        for(int i = 0; i < 3; i++)
            if(p->step_graph[i]) { hipGraphExecDestroy(p->step_graph[i]); p->step_graph[i] = NULL; }
    }
    return old;
}
// does a solve of this problem go without the Jacobian stream when told to?
int mrcal_amd_problem_jacobian_stream_is_optional(mrcal_amd_problem_t* p)
{
    return (p->D.Nobs_board > 0 && p->D.lens_type != MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC) ? 1 : 0;
}
void*    mrcal_amd_problem_stream      (mrcal_amd_problem_t* p) { return (void*)p->stream; }

bool mrcal_amd_problem_set_b_packed(mrcal_amd_problem_t* p, const double* b)
{
    HIP_TRY(hipMemcpyAsync(p->op[p->icur].b, b, (size_t)p->L.Nstate*sizeof(double), hipMemcpyHostToDevice, p->stream), return false);
    HIP_TRY(hipStreamSynchronize(p->stream), return false);
    return true;
}
bool mrcal_amd_problem_get_b_packed(mrcal_amd_problem_t* p, double* b)
{
    HIP_TRY(hipMemcpyAsync(b, p->op[p->icur].b, (size_t)p->L.Nstate*sizeof(double), hipMemcpyDeviceToHost, p->stream), return false);
    HIP_TRY(hipStreamSynchronize(p->stream), return false);
    return true;
}
bool mrcal_amd_problem_get_x(mrcal_amd_problem_t* p, double* x)
{
    if(!device_to_host(x, p->op[p->icur].x, (size_t)p->L.Nmeas*sizeof(double), p->stream))
    {
        set_error("copying x to the host failed: %s", hipGetErrorString(hipGetLastError()));
        return false;
    }
    return true;
}
bool mrcal_amd_problem_get_J(mrcal_amd_problem_t* p, int32_t* rowptr, int32_t* colidx, double* values)
{
    if(values && !problem_ensure_jacobian(p)) return false;
    // (round 6: the big ones through a pinned ring and a pool of copying threads - host_copy.hpp: the caller's arrays are
    //  fresh pages, and one thread copying out of the runtime's staging buffer was 10 GB/s)
    bool ok = true;
    if(rowptr) ok = ok && device_to_host(rowptr, p->d_Jp, ((size_t)p->L.Nmeas+1)*sizeof(int32_t), p->stream);
    if(colidx) ok = ok && device_to_host(colidx, p->d_Ji, (size_t)p->Nnz*sizeof(int32_t), p->stream);
    if(values) ok = ok && device_to_host(values, p->op[p->icur].Jv, (size_t)p->Nnz*sizeof(double), p->stream);
    if(!ok) { set_error("copying the Jacobian to the host failed: %s", hipGetErrorString(hipGetLastError())); return false; }
    return true;
}

bool mrcal_amd_problem_evaluate(mrcal_amd_problem_t* p, bool with_jacobian, bool sync)
{
    if(!problem_evaluate_op(p, p->icur, with_jacobian, false)) return false;
    if(sync) HIP_TRY(hipStreamSynchronize(p->stream), return false);
    return true;
}

// Starts (capacity>0) or stops (capacity<=0) recording one HIP event pair
// around every Jacobian-kernel launch on the problem's stream
bool mrcal_amd_problem_jacobian_timing_begin(mrcal_amd_problem_t* p, int capacity)
{
    return mrcal_amd_problem_jacobian_timing_begin_strided(p, capacity, 1);
}
bool mrcal_amd_problem_jacobian_timing_begin_strided(mrcal_amd_problem_t* p, int capacity, int stride)
{
    p->ev_pool_used = 0;
    p->ev_pool_seen = 0;
    p->ev_pool_stride = stride > 0 ? stride : 1;
    p->ev_pool_enabled = capacity > 0;
    while((int)p->ev_pool.size() < 2*capacity)
    {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e), return false);
        p->ev_pool.push_back(e);
    }
    return true;
}
// Collects what was recorded since _begin(): number of launches and their
// total / min / max duration in ms. Stops the recording
bool mrcal_amd_problem_jacobian_timing_end(mrcal_amd_problem_t* p, int* Nlaunches,
                                           double* total_ms, double* min_ms, double* max_ms)
{
    HIP_TRY(hipStreamSynchronize(p->stream), return false);
    int n = 0; double tot = 0, mn = 1e300, mx = 0;
    for(int i=0; i+1<p->ev_pool_used; i+=2)
    {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, p->ev_pool[i], p->ev_pool[i+1]), return false);
        n++; tot += ms; if(ms < mn) mn = ms; if(ms > mx) mx = ms;
    }
    p->ev_pool_enabled = false;
    p->ev_pool_used = 0;
    if(Nlaunches) *Nlaunches = n;
    if(total_ms)  *total_ms  = tot;
    if(min_ms)    *min_ms    = n ? mn : 0;
    if(max_ms)    *max_ms    = mx;
    return true;
}

// dev tool: average duration (ms) of nrep back-to-back launches of the
// evaluation kernels alone, with the given debug_ablate bits
double mrcal_amd_problem_debug_time_evaluate(mrcal_amd_problem_t* p, bool with_gram, int ablate, int nrep)
{
    if(with_gram && !problem_prepare_solver(p)) return -1.0;
#ifdef MRCAL_AMD_DEV
    const int saved = p->D.debug_ablate;
    p->D.debug_ablate = ablate;
#else
    if(ablate != 0) { set_error("the ablation knob exists in measurement builds only (build.sh -DMRCAL_AMD_DEV)"); return -1.0; }
#endif
    const EvalBuffers B = p->eval_buffers(p->icur, with_gram);
    double total = 0.0;
    for(int i=0; i<nrep+1; i++)
    {
        if(launch_evaluate(p->D, B, true, p->lds_bytes, p->stream, p->ev_j0, p->ev_j1) != hipSuccess) return -1.0;
        if(hipStreamSynchronize(p->stream) != hipSuccess) return -1.0;
        float ms = 0;
        hipEventElapsedTime(&ms, p->ev_j0, p->ev_j1);
        if(i > 0) total += ms;
    }
#ifdef MRCAL_AMD_DEV
    p->D.debug_ablate = saved;
#endif
    return total/nrep;
}

#if defined(MRCAL_AMD_DEV) && defined(BOARD_TS)
// measurement builds (-DMRCAL_AMD_DEV -DBOARD_TS): per-observation phase timestamps of ONE board
// kernel launch, out[Nobs_board][8]. Returns the number of observations
int mrcal_amd_problem_debug_timestamps(mrcal_amd_problem_t* p, bool with_gram, long long* out)
{
    if(with_gram && !problem_prepare_solver(p)) return -1;
    const size_t n = (size_t)p->D.Nobs_board*10;
    long long* d = NULL;
    if(hipMalloc((void**)&d, n*sizeof(long long)) != hipSuccess) return -1;
    hipMemset(d, 0, n*sizeof(long long));
    const EvalBuffers B = p->eval_buffers(p->icur, with_gram);
    for(int i=0;i<3;i++)
    {
        p->D.debug_ts = (i == 2) ? d : NULL;
        if(launch_evaluate(p->D, B, true, p->lds_bytes, p->stream, p->ev_j0, p->ev_j1) != hipSuccess) return -1;
        hipStreamSynchronize(p->stream);
    }
    p->D.debug_ts = NULL;
    hipMemcpy(out, d, n*sizeof(long long), hipMemcpyDeviceToHost);
    hipFree(d);
    return p->D.Nobs_board;
}
#endif

double mrcal_amd_problem_last_jacobian_kernel_ms(mrcal_amd_problem_t* p)
{
    if(!p->have_jacobian_timing) return -1.0;
    if(hipEventSynchronize(p->ev_j1) != hipSuccess) return -1.0;
    float ms = -1.0f;
    if(hipEventElapsedTime(&ms, p->ev_j0, p->ev_j1) != hipSuccess) return -1.0;
    return (double)ms;
}

////////////////////////////////////////////////////////////////////////////////
// drop-in: one evaluation
////////////////////////////////////////////////////////////////////////////////
bool mrcal_optimizer_callback(double* b_packed, int buffer_size_b_packed,
                              double* x,        int buffer_size_x,
                              struct cholmod_sparse_struct* Jt,
                              const double*                 intrinsics,
                              const mrcal_pose_t*           rt_cam_ref,
                              const mrcal_pose_t*           rt_ref_frame,
                              const mrcal_point3_t*         points,
                              const mrcal_calobject_warp_t* calobject_warp,
                              int Ncameras_intrinsics, int Ncameras_extrinsics, int Nframes,
                              int Npoints, int Npoints_fixed,
                              const mrcal_observation_board_t* observations_board,
                              const mrcal_observation_point_t* observations_point,
                              int Nobservations_board,
                              int Nobservations_point,
                              const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                              int Nobservations_point_triangulated,
                              const mrcal_point3_t* observations_board_pool,
                              const mrcal_point3_t* observations_point_pool,
                              const mrcal_lensmodel_t* lensmodel,
                              const int* imagersizes,
                              mrcal_problem_selections_t       problem_selections,
                              const mrcal_problem_constants_t* problem_constants,
                              double calibration_object_spacing,
                              int calibration_object_width_n,
                              int calibration_object_height_n,
                              bool verbose)
{
    (void)problem_constants; (void)verbose;
    last_error_string().clear();

    if(observations_point_triangulated != NULL && Nobservations_point_triangulated &&
       !(!problem_selections.do_optimize_intrinsics_core &&
         !problem_selections.do_optimize_intrinsics_distortions &&
         problem_selections.do_optimize_extrinsics))
    {
        set_error("ERROR: We have triangulated points. At this time this is only allowed if we're NOT optimizing intrinsics AND if we ARE optimizing extrinsics.");
        return false;
    }
    if(Nobservations_board > 0 && problem_selections.do_optimize_calobject_warp && calobject_warp == NULL)
    {
        set_error("ERROR: We're optimizing the calibration object warp, so a buffer with a seed MUST be passed in.");
        return false;
    }
    const mrcal_problem_selections_t sel =
        effective_selections(problem_selections, *lensmodel, Nobservations_board);
    if(!sel.do_optimize_intrinsics_core && !sel.do_optimize_intrinsics_distortions &&
       !sel.do_optimize_extrinsics      && !sel.do_optimize_frames &&
       !sel.do_optimize_calobject_warp)
    {
        set_error("Not optimizing any of our variables!");
        return false;
    }

    const int Nstate =
        mrcal_num_states(Ncameras_intrinsics, Ncameras_extrinsics, Nframes,
                         Npoints, Npoints_fixed, Nobservations_board, sel, lensmodel);
    if(buffer_size_b_packed != Nstate*(int)sizeof(double))
    {
        set_error("The buffer passed to fill-in b_packed has the wrong size. Needed exactly %d bytes, but got %d bytes",
                  Nstate*(int)sizeof(double), buffer_size_b_packed);
        return false;
    }
    const int Nmeas =
        mrcal_num_measurements(Nobservations_board, Nobservations_point,
                               observations_point_triangulated, Nobservations_point_triangulated,
                               calibration_object_width_n, calibration_object_height_n,
                               Ncameras_intrinsics, Ncameras_extrinsics, Nframes,
                               Npoints, Npoints_fixed, sel, lensmodel);
    if(buffer_size_x != Nmeas*(int)sizeof(double))
    {
        set_error("The buffer passed to fill-in x has the wrong size. Needed exactly %d bytes, but got %d bytes",
                  Nmeas*(int)sizeof(double), buffer_size_x);
        return false;
    }

    mrcal_amd_problem_t* P =
        mrcal_amd_problem_create(intrinsics, rt_cam_ref, rt_ref_frame, points, calobject_warp,
                                 Ncameras_intrinsics, Ncameras_extrinsics, Nframes,
                                 Npoints, Npoints_fixed,
                                 observations_board, observations_point,
                                 Nobservations_board, Nobservations_point,
                                 observations_point_triangulated, Nobservations_point_triangulated,
                                 observations_board_pool, observations_point_pool,
                                 lensmodel, imagersizes, sel,
                                 calibration_object_spacing,
                                 calibration_object_width_n, calibration_object_height_n,
                                 0, -1, true);
    if(P == NULL) return false;

    bool ok = false;
    if(P->L.Nstate != Nstate || P->L.Nmeas != Nmeas)
    {
        set_error("internal error: layout mismatch (%d,%d) vs (%d,%d)", P->L.Nstate, P->L.Nmeas, Nstate, Nmeas);
        goto done;
    }
    memcpy(b_packed, P->b_host.data(), (size_t)Nstate*sizeof(double));
    if(!mrcal_amd_problem_evaluate(P, Jt != NULL, true)) goto done;
    if(!mrcal_amd_problem_get_x(P, x)) goto done;
    if(Jt != NULL)
        if(!mrcal_amd_problem_get_J(P, (int32_t*)Jt->p, (int32_t*)Jt->i, (double*)Jt->x)) goto done;
    ok = true;
 done:
    // (hipStreamSynchronize()d by the copies above: nothing of P is in flight)
    problem_destroy_later(P);
    return ok;
}

////////////////////////////////////////////////////////////////////////////////
// drop-in: stand-alone projection (mrcal.h:165-174)
////////////////////////////////////////////////////////////////////////////////
bool mrcal_project(mrcal_point2_t* q, mrcal_point3_t* dq_dp, double* dq_dintrinsics,
                   const mrcal_point3_t* p, int N,
                   const mrcal_lensmodel_t* lensmodel, const double* intrinsics)
{
    last_error_string().clear();
    if(mrcal_amd_device_count() <= 0)
    {
        set_error("no HIP device is visible: libmrcal_amd has no CPU fallback");
        return false;
    }
    if(!lens_supported(lensmodel->type))
    {
        set_error("mrcal_project(): lens model %d is not supported", (int)lensmodel->type);
        return false;
    }
    if(N <= 0) return true;
    const int Ni = lensmodel_num_params(*lensmodel);
    LensConfig cfg; memset(&cfg, 0, sizeof(cfg));
    if(lensmodel->type == MRCAL_LENSMODEL_SPLINED_STEREOGRAPHIC)
    {
        cfg.spline_order = lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.order;
        cfg.spline_Nx    = lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.Nx;
        cfg.spline_Ny    = lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.Ny;
        cfg.spline_segments_per_u =
            spline_segments_per_u(cfg.spline_order, cfg.spline_Nx,
                                  (double)lensmodel->LENSMODEL_SPLINED_STEREOGRAPHIC__config.fov_x_deg);
    }
    if(lensmodel->type == MRCAL_LENSMODEL_CAHVORE)
        cfg.cahvore_linearity = lensmodel->LENSMODEL_CAHVORE__config.linearity;

    double *d_p = NULL, *d_i = NULL, *d_q = NULL, *d_g = NULL, *d_gi = NULL;
    bool ok = true;
    ok = ok && dev_upload(&d_p, (const double*)p, (size_t)3*N);
    ok = ok && dev_upload(&d_i, intrinsics, (size_t)Ni);
    ok = ok && dev_alloc(&d_q, (size_t)2*N);
    if(dq_dp)          ok = ok && dev_alloc(&d_g,  (size_t)6*N);
    if(dq_dintrinsics) ok = ok && dev_alloc(&d_gi, (size_t)2*N*Ni);
    if(ok && d_gi) HIP_TRY(hipMemset(d_gi, 0, (size_t)2*N*Ni*sizeof(double)), ok = false);
    if(ok) HIP_TRY(launch_project_points((int)lensmodel->type, cfg, N, Ni, d_p, d_i, d_q, d_g, d_gi, NULL), ok = false);
    if(ok) HIP_TRY(hipMemcpy(q, d_q, (size_t)2*N*sizeof(double), hipMemcpyDeviceToHost), ok = false);
    if(ok && dq_dp)          HIP_TRY(hipMemcpy(dq_dp, d_g, (size_t)6*N*sizeof(double), hipMemcpyDeviceToHost), ok = false);
    if(ok && dq_dintrinsics) HIP_TRY(hipMemcpy(dq_dintrinsics, d_gi, (size_t)2*N*Ni*sizeof(double), hipMemcpyDeviceToHost), ok = false);
    hipFree(d_p); hipFree(d_i); hipFree(d_q); hipFree(d_g); hipFree(d_gi);
    return ok;
}

} // extern "C"
