// The dog-leg solver and the drop-in mrcal_optimize().
//
// Reference behaviour: mrcal_optimize() (mrcal.c:6179-6624) = [pack state] ->
// do { dogleg_optimize2() } while(outlier rejection found something) ->
// [unpack state, stats]. dogleg_optimize2() is libdogleg's (third party, not in
// the reference tree): Powell's dog leg with a trust region. Its algorithm is
// restated in oracle/dogleg_restated.c (CPU checker); THIS file is the
// product: the same algorithm with every vector/matrix operation AND the
// scalar trust-region decisions on the GPU (solver_device.hpp, "dog-leg
// control"). The host queues trial steps (one captured hipGraph each) and
// polls a pinned snapshot of the device's control block a few steps behind.
//
// mrcal's solver settings (mrcal.c:6296-6299): Jt_x_threshold 0,
// update_threshold 1e-7, trustregion_threshold 0, max_iterations 300; the rest
// are libdogleg's defaults: trustregion0 1e3, decrease 0.1 below rho 0.25,
// increase 2 above rho 0.75 (only if the step reached the trust-region edge).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <vector>
#include <stdlib.h>
#include "host_state.hpp"
#include "problem_object.hpp"
#include "dogleg_choose.hpp"
#include "triangulation.hpp"
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

namespace {

struct DoglegParameters
{
    int    max_iterations                 = 300;
    double trustregion0                   = 1.0e3;
    double trustregion_decrease_factor    = 0.1;
    double trustregion_decrease_threshold = 0.25;
    double trustregion_increase_factor    = 2.0;
    double trustregion_increase_threshold = 0.75;
    double Jt_x_threshold                 = 0.0;
    double update_threshold               = 1e-7;
    double trustregion_threshold          = 0.0;
};

// device scalars -> host, one sync
bool read_scalars(mrcal_amd_problem* P, const double* dev, int n, double* out)
{
    HIP_TRY(hipMemcpyAsync(P->h_scalars, dev, n*sizeof(double), hipMemcpyDeviceToHost, P->stream), return false);
    HIP_TRY(hipStreamSynchronize(P->stream), return false);
    for(int i=0;i<n;i++) out[i] = P->h_scalars[i];
    return true;
}

enum { CTL_RING = 8 };

bool ctl_prepare(mrcal_amd_problem* P)
{
    if(P->h_ctl_ring != NULL) return true;
    HIP_TRY(hipHostMalloc((void**)&P->h_ctl_ring, CTL_RING*sizeof(SolverCtl)), return false);
    for(int i=0;i<CTL_RING;i++)
    {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming), return false);
        P->ctl_events.push_back(e);
    }
    return true;
}

// (re)starts the device-side dog-leg at the point op[P->icur]
bool ctl_reset(mrcal_amd_problem* P, const DoglegParameters& prm, bool check_termination)
{
    if(!ctl_prepare(P)) return false;
    std::vector<char> buf(solver_ctl_bytes(), 0);
    SolverCtl& c = *(SolverCtl*)buf.data();
    c.trustregion_decrease_factor    = prm.trustregion_decrease_factor;
    c.trustregion_decrease_threshold = prm.trustregion_decrease_threshold;
    c.trustregion_increase_factor    = prm.trustregion_increase_factor;
    c.trustregion_increase_threshold = prm.trustregion_increase_threshold;
    c.update_threshold               = prm.update_threshold;
    c.trustregion_threshold          = prm.trustregion_threshold;
    c.max_iterations                 = prm.max_iterations;
    c.check_termination              = check_termination ? 1 : 0;
    c.trustregion                    = prm.trustregion0;
    c.lambda                         = P->stats.lambda;
    c.ib = P->icur; c.ia = 1 - P->icur;
    solver_ctl_init_flags(buf.data(), P->icur);
    HIP_TRY(hipMemsetAsync(P->F.status, 0, sizeof(int), P->stream), return false);
    static const unsigned long long spread0[2] = { 0x7ff0000000000000ull, 0ull };      // (+inf, 0)
    if(P->F.diag_minmax != NULL)
        HIP_TRY(hipMemcpyAsync(P->F.diag_minmax, spread0, sizeof(spread0), hipMemcpyHostToDevice, P->stream), return false);
    HIP_TRY(hipMemcpyAsync(P->d_ctl, buf.data(), buf.size(), hipMemcpyHostToDevice, P->stream), return false);
    HIP_TRY(hipStreamSynchronize(P->stream), return false);   // buf goes out of scope
    P->ctl_initialized = true;
    return true;
}

Step2Args step2_args(mrcal_amd_problem* P)
{
    Step2Args a;
    a.P = &P->D; a.nd = &P->nd; a.br = &P->br; a.plan = &P->plan;
    a.ops = P->d_ops; a.ctl = P->d_ctl; a.F = &P->F; a.gram = P->d_gram;
    a.Jp = P->d_Jp; a.Ji = P->d_Ji; a.step = P->d_step; a.is_leader = P->is_leader;
    a.comm2 = (P->comm != NULL || P->sharded_external) ? P->d_comm : NULL;
    a.snap  = P->capturing ? NULL : P->snap_target;
    a.side = P->side_stream; a.ev_fork = P->ev_fork; a.ev_join = P->ev_join;
    return a;
}

// The two sums over the ranks of a sharded trial step (include/mrcal_amd.h):
// 0 = [S | r | g_S | |x|^2 | status] behind launch_step2_reduce(), 1 = the four
// scalars behind launch_step2_factor(). Queued on the problem's stream; nothing
// to do on a single GPU, or when the caller brings its own collectives
bool step_collective(mrcal_amd_problem* P, int which)
{
    if(P->comm == NULL) return true;
    if(which == 0) return mrcal_amd_comm_allreduce_sum(P->comm, P->F.S, step2_comm1_doubles(P->nd), (void*)P->stream);
    return mrcal_amd_comm_allreduce_sum(P->comm, P->d_comm, 4, (void*)P->stream);
}

// x, J, the normal equations, g, |x|^2, the Cauchy step and (unless the Cauchy
// step already leaves the trust region) the Gauss-Newton step at the starting point
bool enqueue_initial_point(mrcal_amd_problem* P)
{
    const OpRef R = { P->d_ops, &P->d_ctl->ib, NULL };
    if(!problem_evaluate_ref(P, R, true, true, EVAL_PART_PROLOGUE | EVAL_PART_ZERO | EVAL_PART_BOARD | EVAL_PART_REST)) return false;
    const Step2Args a = step2_args(P);
    HIP_TRY(launch_step2_assemble(a, true, P->stream), return false);
    HIP_TRY(launch_step2_reduce(a, P->stream, 1), return false);
    if(!step_collective(P, 0)) return false;
    HIP_TRY(launch_step2_factor(a, true, P->stream), return false);
    if(!step_collective(P, 1)) return false;
    return true;
}

// SolverCtl::error, in words
static const char* solver_error_text(int error)
{
    return error == 4 ? "internal error: the workgroups of the large Cholesky's tail kernel did not all become resident (its grid barrier timed out): "
                        "is the GPU partitioned or CU-masked under this process? Nothing is known about the matrix" :
           error == 3 ? "internal error: a control point taken for uncovered by every board is coupled to other variables (spl_compact_kernel)" :
                        "could not make JtJ positive definite";
}
// One trial step of the dog-leg, entirely queued: every decision is taken on
// the device (step.hip, "the fused step"). segment: 0 = all of it;
// 1 = up to the board kernel, 2 = the board kernel alone, 3 = after it
bool enqueue_trial_step(mrcal_amd_problem* P, int segment)
{
    SolverCtl* ctl = P->d_ctl;
    const Step2Args a = step2_args(P);
    const OpRef Rto = { P->d_ops, &ctl->ia, solver_ctl_skip_eval2(ctl) };
    if(segment == 0 || segment == 1)
    {
        // the step from the current point (its Gauss-Newton step was computed when the
        // point was accepted); then the joint poses of the trial point
        if(prologue_takes_choose(P->D))
        {
            // ONE launch: the prologue's workgroups choose the trial point they evaluate (dogleg_choose.hpp)
            const ChooseArgs ca = step2_choose_args(a);
            if(!problem_evaluate_ref(P, Rto, true, true, EVAL_PART_PROLOGUE | EVAL_PART_ZERO, NULL, &ca)) return false;
        }
        else
        {
            HIP_TRY(launch_step2_choose(a, P->stream), return false);
            if(!problem_evaluate_ref(P, Rto, true, true, EVAL_PART_PROLOGUE | EVAL_PART_ZERO)) return false;
        }
    }
    if(segment == 0 || segment == 2)
        if(!problem_evaluate_ref(P, Rto, true, true, EVAL_PART_BOARD)) return false;
    if(segment == 0 || segment == 3)
    {
        if(!problem_evaluate_ref(P, Rto, true, true, EVAL_PART_REST)) return false;
        HIP_TRY(launch_step2_assemble(a, false, P->stream), return false);
        HIP_TRY(launch_step2_reduce(a, P->stream, 0), return false);
        if(!step_collective(P, 0)) return false;
        HIP_TRY(launch_step2_factor(a, false, P->stream), return false);
        if(!step_collective(P, 1)) return false;
    }
    return true;
}

bool capture_segment(mrcal_amd_problem* P, int segment, hipGraphExec_t* exec)
{
    hipGraph_t graph = NULL;
    HIP_TRY(hipStreamBeginCapture(P->stream, hipStreamCaptureModeThreadLocal), return false);
    P->capturing = true;
    const bool ok = enqueue_trial_step(P, segment);
    P->capturing = false;
    hipError_t e = hipStreamEndCapture(P->stream, &graph);
    if(!ok || e != hipSuccess || graph == NULL)
    {
        if(graph) hipGraphDestroy(graph);
        set_error("could not capture the solver step into a HIP graph: %s", hipGetErrorString(e));
        return false;
    }
    HIP_TRY(hipGraphInstantiate(exec, graph, NULL, NULL, 0), { hipGraphDestroy(graph); return false; });
    hipGraphDestroy(graph);
    return true;
}

// Queues one trial step: eagerly, or (MRCAL_AMD_GRAPH=1) as ONE captured graph;
// when the board kernel is being timed with per-launch events, as graph |
// event | kernel | event | graph. Graphs are captured on first use
bool queue_trial_step(mrcal_amd_problem* P)
{
    // Measured (8 cameras x 1000 frames, when a step was 17 launches): queued
    // eagerly 277 us, replayed as one graph 279 us, as graph|kernel|graph (when
    // the board kernel is timed with events) 297 us. Eager is the default;
    // MRCAL_AMD_GRAPH=1 selects the graph, which does not depend on the host
    // keeping up with the queue. (Single GPU only: the all-reduces of the sharded
    // step are not captured)
    static const bool use_graph = (getenv("MRCAL_AMD_GRAPH") != NULL);
    if(!use_graph || P->comm != NULL) return enqueue_trial_step(P, 0);
    if(!P->ev_pool_enabled)
    {
        if(P->step_graph[0] == NULL && !capture_segment(P, 0, &P->step_graph[0])) return false;
        HIP_TRY(hipGraphLaunch(P->step_graph[0], P->stream), return false);
        return true;
    }
    if(P->step_graph[1] == NULL && !capture_segment(P, 1, &P->step_graph[1])) return false;
    if(P->step_graph[2] == NULL && !capture_segment(P, 3, &P->step_graph[2])) return false;
    HIP_TRY(hipGraphLaunch(P->step_graph[1], P->stream), return false);
    if(!enqueue_trial_step(P, 2)) return false;
    HIP_TRY(hipGraphLaunch(P->step_graph[2], P->stream), return false);
    return true;
}

bool read_ctl(mrcal_amd_problem* P, SolverCtl* c)
{
    HIP_TRY(hipMemcpyAsync(&P->h_ctl_ring[0], P->d_ctl, sizeof(SolverCtl), hipMemcpyDeviceToHost, P->stream), return false);
    HIP_TRY(hipStreamSynchronize(P->stream), return false);
    *c = P->h_ctl_ring[0];
    return true;
}

void absorb_ctl(mrcal_amd_problem* P, const SolverCtl& c)
{
    P->icur = c.ib;
    P->stats.lambda  = c.lambda;
    P->stats.norm2_x = c.norm2_x[c.ib];
    P->op[0].have_normal = P->op[1].have_normal = true;
}

// The splined models' compacted camera block (LcholCompact): how many of its factorization's launches the host
// provides one by one - the panels of the coupled variables at THIS solve's first point; whatever a
// later point needs beyond that is lchol_tail_kernel's. A function of the inputs alone, and the result of a step
// does not depend on it
static bool learn_likely_size(mrcal_amd_problem* P)
{
    if(P->F.cperm_cur == NULL) return true;
    int n1 = 0;
    HIP_TRY(hipMemcpyAsync(&n1, P->op[P->icur].cperm + 2*P->nd.Nc, sizeof(int), hipMemcpyDeviceToHost, P->stream), return false);
    HIP_TRY(hipStreamSynchronize(P->stream), return false);
    if(n1 <= 0 || n1 > P->nd.Nc) n1 = P->nd.Nc;
    // (launch l = npanels is the closing one: launches 0 .. npanels of THIS size one by one)
    P->F.lchol_likely_panels = (n1 + 63)/64;
    // (the tests' hook: k instead - they make lchol_tail_kernel do the work with it)
    if(test_hooks().lchol_likely_panels > 0) P->F.lchol_likely_panels = test_hooks().lchol_likely_panels;
    // The nested-dissection order (lchol_nd_*): the evaluation of this solve's first point has made a plan - the best
    // strip there is at that point - whether launches for it are provided or not. Launches for THAT plan are what the
    // factorizations of this solve get: rounds for the longer side, a border a panel larger than the separator; the plans
    // of later points are used where they fit (else the point goes the ordinary way through the same launches). Where
    // that differs from what the last solve had, the trial step's graphs - the launches are in them - are made again
    // (a solve from the seed and the solve after an outlier pass may well differ: the boxes move with the state)
    if(P->F.ndMA != NULL && P->op[P->icur].ndp != NULL)
    {
        int h[NDH_WORDS];
        HIP_TRY(hipMemcpyAsync(h, P->op[P->icur].ndp, sizeof(h), hipMemcpyDeviceToHost, P->stream), return false);
        HIP_TRY(hipStreamSynchronize(P->stream), return false);
        NdLimits lim = { 0, 0 };
        int likely = 0;
        if(h[NDH_IDEAL_A] > 0 && h[NDH_IDEAL_B] > 0)
        {
            const int a = (h[NDH_IDEAL_A] + ND_PANEL - 1)/ND_PANEL, b = (h[NDH_IDEAL_B] + ND_PANEL - 1)/ND_PANEL;
            lim = NdLimits{ std::max(a, b), h[NDH_IDEAL_NS] + ND_PANEL };
            // (the tests' hook nd_rounds = k: k rounds instead - with fewer than the plan needs, every point goes the ordinary
            //  way through the dissection's launches: the tests hold that path to the bits of the path without them)
            if(test_hooks().nd_rounds > 0) lim.rounds = test_hooks().nd_rounds;
            if(ND_PANEL*lim.rounds > LCH_ND_WMAX) lim = NdLimits{ 0, 0 };
            likely = (h[NDH_IDEAL_NS] + ND_PANEL - 1)/ND_PANEL;
            if(test_hooks().lchol_likely_panels > 0) likely = test_hooks().lchol_likely_panels;
        }
        if(lim.rounds != P->F.nd_lim.rounds || lim.ns_max != P->F.nd_lim.ns_max || likely != P->F.nd_likely_panels)
        {
            P->F.nd_lim = lim; P->F.nd_likely_panels = likely;
            HIP_TRY(launch_nd_plans_off(P->d_ops, P->nd.Nc, P->stream), return false);
            HIP_TRY(hipMemcpyAsync(P->F.nd_lim_dev, &P->F.nd_lim, sizeof(NdLimits), hipMemcpyHostToDevice, P->stream), return false);
            HIP_TRY(hipStreamSynchronize(P->stream), return false);       // (the source is this problem's member: let it be read)
            for(int i = 0; i < 3; i++)
                if(P->step_graph[i]) { hipGraphExecDestroy(P->step_graph[i]); P->step_graph[i] = NULL; }
        }
    }
    return true;
}

// libdogleg's main loop. The host only keeps the queue fed and looks, a few
// steps behind, at whether the device has declared the solve finished. On
// return op[P->icur] is the final operating point
bool run_dogleg(mrcal_amd_problem* P, const DoglegParameters& prm)
{
    if(!ctl_reset(P, prm, true)) return false;
    if(!enqueue_initial_point(P)) return false;
    if(!learn_likely_size(P)) return false;

    const int LAG = 3;      // how many steps the host may run ahead of what it has seen
    // verbose (mrcal.c:6291 turns on libdogleg's per-iteration report with it), or the environment
    static const bool debug_env = (getenv("MRCAL_AMD_DEBUG_SOLVER") != NULL);
    const bool debug = debug_env || P->verbose;
    int  nqueued = 0;
    bool done = false;
    const int max_trials = 100*prm.max_iterations + 1000;   // runaway guard
    while(!done && nqueued < max_trials)
    {
        const int slot = nqueued % CTL_RING;
        // the step's last kernel leaves the snapshot in the pinned ring itself (eager queueing; a captured
        // graph has its arguments baked in and is followed by a copy instead)
        static const bool use_graph = (getenv("MRCAL_AMD_GRAPH") != NULL);
        const bool by_kernel = !use_graph || P->comm != NULL;
        P->snap_target = by_kernel ? &P->h_ctl_ring[slot] : NULL;
        const bool queued = queue_trial_step(P);
        P->snap_target = NULL;
        if(!queued) return false;
        if(!by_kernel)
            HIP_TRY(hipMemcpyAsync(&P->h_ctl_ring[slot], P->d_ctl, sizeof(SolverCtl), hipMemcpyDeviceToHost, P->stream), return false);
        HIP_TRY(hipEventRecord(P->ctl_events[slot], P->stream), return false);
        nqueued++;
        if(debug)
        {
            HIP_TRY(hipStreamSynchronize(P->stream), return false);
            const SolverCtl& s = P->h_ctl_ring[slot];
            int st = 0;
            hipMemcpy(&st, P->F.status, sizeof(int), hipMemcpyDeviceToHost);
            fprintf(stderr, "trial %3d: accepted %3d tr %-10.4g |x|^2 %.10g lambda %-8.3g refactor %d abort %d chol_status %d step_len %.3g expected %.6g kc %.3g kg %.3g done %d\n",
                    nqueued, s.Nsteps_accepted, s.trustregion, s.norm2_x[s.ib], s.lambda, s.refactor, s.abort_step, st,
                    sqrt(s.step_len_sq), s.expected_improvement, s.k_cauchy, s.k_gn, s.done);
        }
        // Sharded: every queued step carries two collectives, so every rank must queue
        // the SAME number of steps. The control block is replicated bit for bit, but
        // which snapshots have arrived when the host looks is a matter of timing: only
        // the snapshot exactly LAG steps back counts, and it is waited for
        if(P->comm != NULL)
        {
            if(nqueued >= LAG)
            {
                const int sj = (nqueued - LAG) % CTL_RING;
                HIP_TRY(hipEventSynchronize(P->ctl_events[sj]), return false);
                if(P->h_ctl_ring[sj].done) done = true;
            }
            continue;
        }
        // look at the newest snapshot that is at least LAG steps old, or any
        // newer one that happens to be complete
        for(int back = 1; back <= CTL_RING-1 && back <= nqueued; back++)
        {
            const int j = nqueued - back, sj = j % CTL_RING;
            hipError_t q = hipEventQuery(P->ctl_events[sj]);
            if(q == hipErrorNotReady)
            {
                if(back < LAG) continue;
                HIP_TRY(hipEventSynchronize(P->ctl_events[sj]), return false);
            }
            else if(q != hipSuccess)
            {
                set_error("hipEventQuery: %s", hipGetErrorString(q));
                return false;
            }
            if(P->h_ctl_ring[sj].done) done = true;
            break;
        }
    }
    SolverCtl c;
    if(!read_ctl(P, &c)) return false;
    P->last_ctl_error = c.error;
    if(c.error)
    {
        set_error("%s", solver_error_text(c.error));
        return false;
    }
    if(!c.done)
    {
        set_error("the dog-leg did not terminate within %d trial steps", max_trials);
        return false;
    }
    absorb_ctl(P, c);
    P->stats.Niterations     += c.Nsteps_accepted;
    P->stats.Nevaluations    += c.Nevaluations;
    P->stats.Nfactorizations += c.Nfactorizations;
    // (round 6) how far apart the diagonal entries of the big camera block's factors lay in this pass: the solve ends with
    // d = -Y^T z, Y = L^-1 formed explicitly, whose error grows like n eps max/min of that diagonal. A ratio
    // below 1e-10 - a step that is wrong in its fourth digit; at 5e-9, which ill-determined splined problems of the fuzz
    // sweeps reach, it is its fifth and the solves end where the reference's end - sends the problem to the backward sweep
    // for good, from the next pass on (mrcal_amd_problem_solve()): slower, backward stable
    if(P->F.diag_minmax != NULL)
    {
        unsigned long long mm[2];
        HIP_TRY(hipMemcpy(mm, P->F.diag_minmax, sizeof(mm), hipMemcpyDeviceToHost), return false);
        double lo, hi; memcpy(&lo, &mm[0], 8); memcpy(&hi, &mm[1], 8);
        if(hi > 0.0 && lo <= hi)
        {
            P->lchol_diag_ratio = lo/hi;
            const int lg = test_hooks().lchol_fallback_log10 ? test_hooks().lchol_fallback_log10 : -10;
            if(P->lchol_diag_ratio < pow(10.0, (double)lg) && !P->F.use_sweep && P->comm == NULL) P->sweep_fallback_wanted = true;
        }
    }
    return true;
}

// Host-driven Gauss-Newton step of op[i] (tests, and the reference for the
// device-controlled path): (JtJ + lambda I) d = -Jt x, adding lambda until the
// factorization succeeds. The normal equations of op[i] must be current
bool compute_gauss_newton(mrcal_amd_problem* P, int i)
{
    for(;;)
    {
        P->stats.Nfactorizations++;
        HIP_TRY(hipMemsetAsync(P->F.status, 0, sizeof(int), P->stream), return false);
        HIP_TRY(launch_factor_local(P->nd, P->br, P->opref(i), P->F, P->stats.lambda, NULL, true, P->stream), return false);
        HIP_TRY(launch_solve_backsub(P->nd, P->br, P->opref(i), P->F, NULL, true, P->stream), return false);
        double* sc = P->op[i].scalars;
        HIP_TRY(hipMemsetAsync(&sc[SC_TMP0], 0, sizeof(double), P->stream), return false);
        HIP_TRY(launch_dot(P->nd.Nstate, P->op[i].step_gn, P->op[i].step_gn, &sc[SC_TMP0], P->stream), return false);
        int status = 0;
        HIP_TRY(hipMemcpyAsync(P->h_scalars + 32, P->F.status, sizeof(int), hipMemcpyDeviceToHost, P->stream), return false);
        double lensq;
        if(!read_scalars(P, &sc[SC_TMP0], 1, &lensq)) return false;
        memcpy(&status, P->h_scalars + 32, sizeof(int));
        if(status == 0 && lensq == lensq) return true;
        // singular JtJ: regularize, like libdogleg does
        P->stats.lambda = (P->stats.lambda == 0.0) ? 1e-10 : P->stats.lambda*10.0;
        if(!(P->stats.lambda < 1e30))
        {
            set_error("could not make JtJ positive definite");
            return false;
        }
    }
}

// mrcal.c:3978-4402 markOutliers(): board corners on the GPU; the triangulated
// pairs on the host, because the reference's pass over them is sequential (an
// observation marked by one pair is an outlier for the pairs that follow).
// found: new outliers were marked
bool mark_outliers(mrcal_amd_problem* P, int* Noutliers_board, int* Noutliers_tri, bool* found)
{
    *found = false;
    *Noutliers_board = 0;
    *Noutliers_tri   = 0;
    const int Npts   = P->D.Nobs_board * P->D.W * P->D.H;
    const int Npairs = (int)P->tri_meta_host.size();
    if(Npts <= 0 && Npairs <= 0 && P->comm == NULL) return true;
    const double k0 = 4.0, k1 = 5.0;
    const double* x = P->op[P->icur].x;
    double* sums = P->op[P->icur].scalars + SC_TMP0;

    auto board_stats = [&](double thresh_sq, int* counts, double* sum) -> bool
    {
        counts[0] = counts[1] = counts[2] = counts[3] = 0; *sum = 0.0;
        if(Npts <= 0) return true;
        HIP_TRY(hipMemsetAsync(P->d_counts, 0, 4*sizeof(int), P->stream), return false);
        HIP_TRY(hipMemsetAsync(sums, 0, sizeof(double), P->stream), return false);
        HIP_TRY(launch_outlier_stats(Npts, thresh_sq, x, P->d_board_pool, P->d_counts, sums, P->d_outlier_part, P->stream), return false);
        HIP_TRY(hipMemcpyAsync(P->h_scalars + 32, P->d_counts, 4*sizeof(int), hipMemcpyDeviceToHost, P->stream), return false);
        if(!read_scalars(P, sums, 1, sum)) return false;
        memcpy(counts, P->h_scalars + 32, 4*sizeof(int));
        return true;
    };
    // sharded: the statistics are those of the whole problem (one tiny sum over the ranks per number)
    auto over_ranks = [&](double* v, int n) -> bool
    {
        if(P->comm == NULL) return true;
        double* d = P->d_comm + 8;
        HIP_TRY(hipMemcpyAsync(d, v, n*sizeof(double), hipMemcpyHostToDevice, P->stream), return false);
        if(!mrcal_amd_comm_allreduce_sum(P->comm, d, n, (void*)P->stream)) return false;
        HIP_TRY(hipMemcpyAsync(v, d, n*sizeof(double), hipMemcpyDeviceToHost, P->stream), return false);
        HIP_TRY(hipStreamSynchronize(P->stream), return false);
        return true;
    };
    // Sharded: every rank issues the SAME collectives whether or not its shard holds any board corners or
    // pairs (partition_frames() may leave a rank without observations): a rank that skipped one would sit in
    // gather_state()'s all-reduce while the others are still in here. Decisions are taken from the summed
    // numbers only, so they are the same everywhere.
    int counts[4]; double sum_board;
    if(!board_stats(-1.0, counts, &sum_board)) return false;

    // triangulated pairs (this rank's points): divergent ones are thrown out right away
    std::vector<double> x_tri(Npairs > 0 ? Npairs : 1), b(P->L.Nstate > 0 ? P->L.Nstate : 1);
    std::vector<int>&   out = P->tri_outlier_host;
    double sum_tri = 0.0;
    int    Nin_tri = 0, Nout_tri = 0;
    bool   marked_tri = false;
    if(Npairs > 0)
    {
        HIP_TRY(hipMemcpyAsync(x_tri.data(), x + P->L.i_meas_triangulated, (size_t)Npairs*sizeof(double), hipMemcpyDeviceToHost, P->stream), return false);
        HIP_TRY(hipMemcpyAsync(b.data(), P->op[P->icur].b, (size_t)P->L.Nstate*sizeof(double), hipMemcpyDeviceToHost, P->stream), return false);
        HIP_TRY(hipStreamSynchronize(P->stream), return false);
        auto rt_of = [&](int icam, double* rt) -> const double*
        {
            if(icam < 0) return NULL;
            const double* s = &b[P->L.i_state_extrinsics + 6*icam];
            for(int i=0;i<3;i++) { rt[i] = s[i]*SCALE_ROTATION_CAMERA; rt[3+i] = s[3+i]*SCALE_TRANSLATION_CAMERA; }
            return rt;
        };
        for(int ip = 0; ip < Npairs; ip++)
        {
            const TriPairMeta& m = P->tri_meta_host[ip];
            if(!(out[m.i0] || out[m.i1]))
            {
                double rt0[6], rt1[6];
                bool convergent = true;
                tri_pair_error<0>(&P->tri_px_host[3*m.i0], &P->tri_px_host[3*m.i1],
                                  rt_of(m.icam_extrinsics0, rt0), rt_of(m.icam_extrinsics1, rt1), &convergent);
                if(!convergent) { out[m.i0] = out[m.i1] = 1; marked_tri = true; }
            }
            if(out[m.i0] || out[m.i1]) Nout_tri++;
            else { sum_tri += x_tri[ip]*x_tri[ip]; Nin_tri++; }
        }
    }
    double Npts_all = (double)Npts;
    {
        double v[7] = { (double)counts[0], sum_board, Npts_all, sum_tri, (double)Nin_tri, (double)Nout_tri, marked_tri ? 1.0 : 0.0 };
        if(!over_ranks(v, 7)) return false;
        counts[0] = (int)v[0]; sum_board = v[1]; Npts_all = v[2];
        sum_tri = v[3]; Nin_tri = (int)v[4]; Nout_tri = (int)v[5]; marked_tri = v[6] > 0.0;
    }
    int Nout_board = counts[0];
    const int Nin_board = (int)Npts_all - Nout_board;
    *Noutliers_board = Nout_board;
    *Noutliers_tri   = Nout_tri;
    bool any = marked_tri;
    const int Ndenom = Nin_board*2 + Nin_tri;
    if(Ndenom > 0)
    {
        const double var = (sum_board + sum_tri)/(double)Ndenom;
        if(!any)
        {
            double dummy;
            if(!board_stats(k1*k1*var, counts, &dummy)) return false;
            double v[1] = { (double)counts[1] };
            for(int ip = 0; ip < Npairs && v[0] == 0.0; ip++)
            {
                const TriPairMeta& m = P->tri_meta_host[ip];
                if(!out[m.i0] && !out[m.i1] && x_tri[ip]*x_tri[ip] > k1*k1*var) v[0] = 1.0;
            }
            if(!over_ranks(v, 1)) return false;
            if(v[0] > 0) any = true;
        }
        if(any)
        {
            double v[2] = { 0.0, 0.0 };
            if(Npts > 0)
            {
                HIP_TRY(hipMemsetAsync(P->d_counts, 0, 4*sizeof(int), P->stream), return false);
                HIP_TRY(launch_mark_outliers(Npts, k0*k0*var, x, P->d_board_pool, P->d_counts, P->stream), return false);
                HIP_TRY(hipMemcpyAsync(P->h_scalars + 32, P->d_counts, 4*sizeof(int), hipMemcpyDeviceToHost, P->stream), return false);
                HIP_TRY(hipStreamSynchronize(P->stream), return false);
                memcpy(counts, P->h_scalars + 32, 4*sizeof(int));
                v[0] = (double)counts[0];
            }
            for(int ip = 0; ip < Npairs; ip++)
            {
                const TriPairMeta& m = P->tri_meta_host[ip];
                if(!out[m.i0] && !out[m.i1] && x_tri[ip]*x_tri[ip] > k0*k0*var)
                {
                    out[m.i0] = out[m.i1] = 1;
                    v[1] += 1.0;
                }
            }
            if(!over_ranks(v, 2)) return false;
            *Noutliers_board = Nout_board + (int)v[0];
            *Noutliers_tri   = Nout_tri   + (int)v[1];
        }
    }
    if(!any) return true;
    if(Npairs > 0)
        HIP_TRY(hipMemcpy(P->d_tri_outlier, out.data(), (size_t)(out.size()-1)*sizeof(int), hipMemcpyHostToDevice), return false);
    *found = true;
    return true;
}

// mrcal_optimize(check_gradient): libdogleg's dogleg_testGradient() (as restated in
// oracle/dogleg_restated.c, same columns) for every state variable, from the
// device's J and x. Small problems only: Nstate*Nmeasurements lines
bool test_gradients(mrcal_amd_problem* P)
{
    const int Nstate = P->L.Nstate, Nmeas = P->L.Nmeas;
    const double delta = 1e-6;
    std::vector<double>  b0(Nstate > 0 ? Nstate : 1), b(Nstate > 0 ? Nstate : 1), x0(Nmeas > 0 ? Nmeas : 1), x1(Nmeas > 0 ? Nmeas : 1);
    std::vector<int32_t> Jp((size_t)Nmeas + 1), Ji(P->Nnz > 0 ? P->Nnz : 1);
    std::vector<double>  Jx(P->Nnz > 0 ? P->Nnz : 1);
    if(!mrcal_amd_problem_get_b_packed(P, b0.data())) return false;
    if(!mrcal_amd_problem_evaluate(P, true, true)) return false;
    if(!mrcal_amd_problem_get_J(P, Jp.data(), Ji.data(), Jx.data())) return false;
    for(int var = 0; var < Nstate; var++)
    {
        b = b0; b[var] -= delta/2.0;
        if(!mrcal_amd_problem_set_b_packed(P, b.data()) || !mrcal_amd_problem_evaluate(P, false, true) ||
           !mrcal_amd_problem_get_x(P, x0.data())) return false;
        b[var] += delta;
        if(!mrcal_amd_problem_set_b_packed(P, b.data()) || !mrcal_amd_problem_evaluate(P, false, true) ||
           !mrcal_amd_problem_get_x(P, x1.data())) return false;
        if(var == 0)
            printf("# ivar imeasurement gradient_reported gradient_observed error error_relative\n");
        for(int j = 0; j < Nmeas; j++)
        {
            double g_rep = 0.0;
            for(int e = Jp[j]; e < Jp[j+1]; e++)
                if(Ji[e] == var) g_rep += Jx[e];
            const double g_obs = (x1[j] - x0[j]) / delta;
            const double err   = g_rep - g_obs;
            const double den   = (fabs(g_rep) + fabs(g_obs)) / 2.0;
            printf("%d %d %.6g %.6g %.6g %.6g\n", var, j, g_rep, g_obs, err, den > 0.0 ? fabs(err)/den : 0.0);
        }
    }
    fflush(stdout);
    return mrcal_amd_problem_set_b_packed(P, b0.data());
}

// mrcal_optimize(verbose): the regularization report of mrcal.c:6503-6598
void report_regularization(mrcal_amd_problem* P, const mrcal_problem_selections_t& sel)
{
    const Layout& L = P->L;
    if(L.Nmeas_regularization <= 0) return;
    std::vector<double> xreg(L.Nmeas_regularization);
    if(hipMemcpy(xreg.data(), P->op[P->icur].x + L.i_meas_regularization, xreg.size()*sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        return;
    const double norm2_error = P->stats.norm2_x;
    const int Ncore = L.Ncore;
    const int Ndist_reg  = sel.do_optimize_intrinsics_distortions ? L.dims.Ncameras_intrinsics*(L.Nintrinsics - Ncore) : 0;
    const int Ncenter    = sel.do_optimize_intrinsics_core ? L.dims.Ncameras_intrinsics*2 : 0;
    double n2d = 0.0, n2c = 0.0;
    int i = 0;
    for(int k = 0; k < Ndist_reg && i < (int)xreg.size(); k++, i++) n2d += xreg[i]*xreg[i];
    for(int k = 0; k < Ncenter   && i < (int)xreg.size(); k++, i++) n2c += xreg[i]*xreg[i];
    const double rd = n2d/norm2_error, rc = n2c/norm2_error;
    if(rd > 0.01) fprintf(stderr, "mrcal_amd: WARNING: regularization ratio for lens distortion exceeds 1%%. Is the scale factor too high? Ratio = %.3g/%.3g = %.3g\n", n2d, norm2_error, rd);
    if(rc > 0.01) fprintf(stderr, "mrcal_amd: WARNING: regularization ratio for the projection centerpixel exceeds 1%%. Is the scale factor too high? Ratio = %.3g/%.3g = %.3g\n", n2c, norm2_error, rc);
    fprintf(stderr, "mrcal_amd: reg err ratio (distortion,centerpixel): %.3g %.3g\n", rd, rc);
    if(L.has_unity_cam01 && i < (int)xreg.size())
        fprintf(stderr, "mrcal_amd: reg err ratio (unity_cam01): %.3g\n", xreg[i]*xreg[i]/norm2_error);
}

} // namespace

namespace { int& optimize_jacobian_stream_policy() { static int policy = 0; return policy; } }
namespace mrcal_amd { TestHooks& test_hooks() { static TestHooks h = {0, 0, 0, 0}; return h; } }

extern "C" {

// (round 6) does the drop-in mrcal_optimize() stream the CSR values of J to HBM in every step although nothing reads
// them? 0 (default): no - the same results to the bit, a shorter step. 1: yes, as the metric defines a step. For the
// calls AFTER this one; returns the previous setting
int mrcal_amd_set_optimize_jacobian_stream(int stream)
{
    const int old = optimize_jacobian_stream_policy();
    optimize_jacobian_stream_policy() = stream ? 1 : 0;
    return old;
}

// For the tests (include/mrcal_amd.h): "lchol_likely_panels", "nd_rounds", "lchol_sweep"; the previous value, -1 for an unknown name
int mrcal_amd_set_test_hook(const char* name, int value)
{
    TestHooks& h = test_hooks();
    int* p = !strcmp(name, "lchol_likely_panels") ? &h.lchol_likely_panels :
             !strcmp(name, "nd_rounds")           ? &h.nd_rounds :
             !strcmp(name, "lchol_sweep")         ? &h.lchol_sweep :
             !strcmp(name, "lchol_fallback_log10") ? &h.lchol_fallback_log10 : (int*)NULL;
    if(p == NULL) return -1;
    const int old = *p; *p = value;
    return old;
}

// min / max of the diagonal of the big camera block's Cholesky factors over the last dog-leg pass (1: no such factorization)
double mrcal_amd_problem_lchol_diag_ratio(mrcal_amd_problem_t* P) { return P->lchol_diag_ratio; }
int    mrcal_amd_problem_uses_sweep(mrcal_amd_problem_t* P) { return P->F.use_sweep; }

// Resident tier: the full solve on a resident problem. Returns rms error, <0
// on failure
double mrcal_amd_problem_solve(mrcal_amd_problem_t* P, int max_iterations,
                               int* Noutliers_board_out)
{
    last_error_string().clear();
    if(!problem_prepare_solver(P)) return -1.0;
    DoglegParameters prm;
    if(max_iterations > 0) prm.max_iterations = max_iterations;

    const auto t0 = std::chrono::steady_clock::now();
    P->stats = mrcal_amd_solver_stats();
    int Noutliers = 0, Noutliers_tri = 0;
    // (round 6: the evaluations queued from here on leave the Jacobian stream out if the problem was told so)
    struct JfreeScope { mrcal_amd_problem* P; JfreeScope(mrcal_amd_problem* p) : P(p) { P->jfree_now = !P->solve_stores_jacobian; }
                        ~JfreeScope() { P->jfree_now = false; } } jfree_scope(P);
    for(;;)
    {
        // the reference makes a new libdogleg context for every pass
        // (mrcal.c:6433-6439): the diagonal regularization starts over at 0
        P->stats.lambda = 0.0;
        P->last_ctl_error = 0;
        if(!run_dogleg(P, prm))
        {
            // the compaction / dissection of a splined camera block rest on a classification of the control
            // points (covered by a board's box or not; which side of the strip) that the reduction CHECKS against the
            // data: an entry with no place in the compacted matrix that is not zero is error 3. Then the classification
            // is wrong for this problem - a kind of row it does not know - and the answer is the ordinary reduction of
            // the whole camblock, not a failed solve: both off for good, the pass again from the point it reached (every
            // accepted point lowered the true cost). Loudly
            if(P->last_ctl_error == 3 && P->plan.spl_compact)
            {
                fprintf(stderr, "mrcal_amd: WARNING: %s. Solving this problem without the compaction of its camera block\n", solver_error_text(3));
                P->F.cperm_cur = NULL; P->plan.spl_compact = 0; P->plan.nd_lim = NULL; P->F.nd_lim.rounds = 0;
                for(int i = 0; i < 3; i++)
                    if(P->step_graph[i]) { hipGraphExecDestroy(P->step_graph[i]); P->step_graph[i] = NULL; }
                last_error_string().clear();
                continue;
            }
            return -1.0;
        }
        // (every evaluation of a pass that streams J wrote it, the final point's included)
        if(!P->jfree_now) P->jacobian_stale = false;
        if(P->sweep_fallback_wanted)
        {
            P->sweep_fallback_wanted = false;
            // (the pass that was just done stands - a second run of it would be 300 more iterations in front of the outlier
            //  marking where the iteration limit ends the passes, another trajectory than the reference's -; what follows it
            //  goes through the sweep)
            fprintf(stderr, "mrcal_amd: WARNING: the diagonal of the camera block's Cholesky factor spans %.1e: this problem's steps go through "
                            "the backward sweep from here on instead of the explicit inverse (slower, backward stable)\n", 1.0/P->lchol_diag_ratio);
            P->F.use_sweep = 1;
            P->F.cperm_cur = NULL; P->plan.spl_compact = 0; P->plan.nd_lim = NULL; P->F.nd_lim.rounds = 0;
            for(int i = 0; i < 3; i++)
                if(P->step_graph[i]) { hipGraphExecDestroy(P->step_graph[i]); P->step_graph[i] = NULL; }
        }
        if(!P->L.sel.do_apply_outlier_rejection) break;
        bool found;
        if(!mark_outliers(P, &Noutliers, &Noutliers_tri, &found)) return -1.0;
        P->stats.Noutliers_triangulated = Noutliers_tri;
        if(!found) break;
        P->stats.Noutlier_passes++;
        if(P->is_leader)
        fprintf(stderr, "mrcal_amd: Threw out some outliers. New count = %d/%d (%.1f%%). Going again\n",
                Noutliers, P->L.Nmeas_boards,
                (double)(Noutliers*100)/(double)(P->L.Nmeas_boards > 0 ? P->L.Nmeas_boards : 1));
    }
    if(Noutliers_board_out) *Noutliers_board_out = Noutliers;
    if(P->comm != NULL && !mrcal_amd_problem_gather_state(P)) return -1.0;
    P->stats.seconds_solve = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return sqrt(P->stats.norm2_x / (double)P->L.Nmeas);
}

// Exactly `Nsteps` accepted-or-rejected dog-leg steps from the current
// operating point, no termination tests, no outlier rejection: the unit the
// benchmark times. Each step = 1 evaluation of x,J + the normal equations + a
// factorization (when the trust region asks for the Gauss-Newton step).
// Returns the number of steps queued, <0 on error. Returns when they are done
int mrcal_amd_problem_run_steps(mrcal_amd_problem_t* P, int Nsteps, double* trustregion_inout)
{
    last_error_string().clear();
    if(!problem_prepare_solver(P)) return -1;
    DoglegParameters prm;
    if(trustregion_inout && *trustregion_inout > 0.0) prm.trustregion0 = *trustregion_inout;
    struct JfreeScope { mrcal_amd_problem* P; JfreeScope(mrcal_amd_problem* p) : P(p) { P->jfree_now = !P->solve_stores_jacobian; }
                        ~JfreeScope() { P->jfree_now = false; } } jfree_scope(P);
    if(!P->ctl_initialized || !(trustregion_inout && *trustregion_inout > 0.0))
    {
        if(!ctl_reset(P, prm, false)) return -1;
        if(!enqueue_initial_point(P)) return -1;
        if(!learn_likely_size(P)) return -1;
    }
    for(int n = 0; n < Nsteps; n++)
        if(!queue_trial_step(P)) return -1;
    SolverCtl c;
    if(!read_ctl(P, &c)) return -1;
    if(c.error) { set_error("%s", solver_error_text(c.error)); return -1; }
    absorb_ctl(P, c);
    P->stats.Nevaluations    = c.Nevaluations;
    P->stats.Nfactorizations = c.Nfactorizations;
    P->stats.Niterations     = c.Nsteps_accepted;
    if(trustregion_inout) *trustregion_inout = c.trustregion;
    return Nsteps;
}

void mrcal_amd_problem_solver_stats(mrcal_amd_problem_t* P,
                                    int* Niterations, int* Nevaluations, int* Nfactorizations,
                                    int* Noutlier_passes, double* norm2_x, double* lambda, double* seconds)
{
    if(Niterations)     *Niterations     = P->stats.Niterations;
    if(Nevaluations)    *Nevaluations    = P->stats.Nevaluations;
    if(Nfactorizations) *Nfactorizations = P->stats.Nfactorizations;
    if(Noutlier_passes) *Noutlier_passes = P->stats.Noutlier_passes;
    if(norm2_x)         *norm2_x         = P->stats.norm2_x;
    if(lambda)          *lambda          = P->stats.lambda;
    if(seconds)         *seconds         = P->stats.seconds_solve;
}

// Debug/test access to the normal equations of the current operating point:
// evaluates x, J and the blocks, copies A (Nc*Nc), Bt (NE*Nc), D (NEb*36), g
// (Nstate) to the host. Any pointer may be NULL
bool mrcal_amd_problem_get_normal_equations(mrcal_amd_problem_t* P,
                                            double* A, double* Bt, double* D, double* g,
                                            double* norm2_x,
                                            int* dims /* Nc, NE, NEb, Nfb, Nie, Nwarp */)
{
    last_error_string().clear();
    if(!problem_prepare_solver(P)) return false;
    if(!problem_evaluate_op(P, P->icur, true, true)) return false;
    const NormalDims& nd = P->nd;
    const mrcal_amd_oppoint& N = P->op[P->icur];
    if(A)  HIP_TRY(hipMemcpyAsync(A,  N.A,  (size_t)nd.Nc*nd.Nc*sizeof(double), hipMemcpyDeviceToHost, P->stream), return false);
    if(Bt) HIP_TRY(hipMemcpyAsync(Bt, N.Bt, (size_t)nd.NE*nd.Nc*sizeof(double), hipMemcpyDeviceToHost, P->stream), return false);
    if(D)  HIP_TRY(hipMemcpyAsync(D,  N.D,  (size_t)nd.NEb*36*sizeof(double),   hipMemcpyDeviceToHost, P->stream), return false);
    if(g)  HIP_TRY(hipMemcpyAsync(g,  N.g,  (size_t)nd.Nstate*sizeof(double),   hipMemcpyDeviceToHost, P->stream), return false);
    if(norm2_x) HIP_TRY(hipMemcpyAsync(norm2_x, &N.scalars[SC_NORM2_X], sizeof(double), hipMemcpyDeviceToHost, P->stream), return false);
    HIP_TRY(hipStreamSynchronize(P->stream), return false);
    // (the device keeps A's lower triangle: the upper one is written by some assemblies and not by others)
    if(A)
        for(int i = 0; i < nd.Nc; i++)
            for(int j = i + 1; j < nd.Nc; j++) A[(size_t)i*nd.Nc + j] = A[(size_t)j*nd.Nc + i];
    if(dims) { dims[0]=nd.Nc; dims[1]=nd.NE; dims[2]=nd.NEb; dims[3]=nd.Nfb; dims[4]=nd.S_split; dims[5]=nd.Nwarp; }
    return true;
}

// Solves (JtJ + lambda I) d = -Jt x at the current operating point; d (Nstate)
// to the host. The Gauss-Newton step, for tests
bool mrcal_amd_problem_gauss_newton_step(mrcal_amd_problem_t* P, double* step)
{
    last_error_string().clear();
    if(!problem_prepare_solver(P)) return false;
    if(!problem_evaluate_op(P, P->icur, true, true)) return false;
    if(!compute_gauss_newton(P, P->icur)) return false;
    HIP_TRY(hipMemcpyAsync(step, P->op[P->icur].step_gn, (size_t)P->nd.Nstate*sizeof(double), hipMemcpyDeviceToHost, P->stream), return false);
    HIP_TRY(hipStreamSynchronize(P->stream), return false);
    return true;
}

////////////////////////////////////////////////////////////////////////////////
// multi-GPU (include/mrcal_amd.h). The sharded step IS the single-GPU step with
// two all-reduces in it (enqueue_trial_step); what is here is the attachment of
// the communicator, the final gather of the state, and the same step cut at the
// collectives for a driver that brings its own (the protocol tests)
////////////////////////////////////////////////////////////////////////////////
bool mrcal_amd_problem_attach_comm(mrcal_amd_problem_t* P, mrcal_amd_comm_t* comm)
{
    last_error_string().clear();
    if(!problem_prepare_solver(P)) return false;
    P->comm = comm;
    P->ctl_initialized = false;
    // (the ranks of a sharded solve sum their camera blocks entry by entry: no rank puts its own in another order)
    P->F.cperm_cur = NULL; P->plan.spl_compact = 0; P->plan.nd_lim = NULL; P->F.nd_lim.rounds = 0;
    return true;
}
// dev / tests: the nested-dissection order of a splined problem's camera block. out[0] rounds the host provides launches for
// (0: none), out[1] the largest separator they serve, out[2..5] the current operating point's plan: used or not, columns of
// side A, side B (padded to whole panels), of the separator, out[6..8] what the best strip at that point would give (unpadded)
bool mrcal_amd_problem_dissection(mrcal_amd_problem_t* P, int* out /* [9] */)
{
    for(int i = 0; i < 9; i++) out[i] = 0;
    if(P->F.ndMA == NULL || P->op[P->icur].ndp == NULL) return true;
    out[0] = P->F.nd_lim.rounds; out[1] = P->F.nd_lim.ns_max;
    int h[NDH_WORDS];
    HIP_TRY(hipMemcpyAsync(h, P->op[P->icur].ndp, sizeof(h), hipMemcpyDeviceToHost, P->stream), return false);
    HIP_TRY(hipStreamSynchronize(P->stream), return false);
    out[2] = h[NDH_ACTIVE]; out[3] = h[NDH_NA]; out[4] = h[NDH_NB]; out[5] = h[NDH_NS];
    out[6] = h[NDH_IDEAL_A]; out[7] = h[NDH_IDEAL_B]; out[8] = h[NDH_IDEAL_NS];
    return true;
}
bool mrcal_amd_problem_gather_state(mrcal_amd_problem_t* P)
{
    last_error_string().clear();
    if(P->comm == NULL) return true;
    // everybody zeroes what it does not own (the camera block belongs to the leader), then the sum is the state
    HIP_TRY(launch_mask_state(P->nd, P->br, P->is_leader, P->op[P->icur].b, P->stream), return false);
    if(!mrcal_amd_comm_allreduce_sum(P->comm, P->op[P->icur].b, P->nd.Nstate, (void*)P->stream)) return false;
    HIP_TRY(hipStreamSynchronize(P->stream), return false);
    return true;
}
void mrcal_amd_problem_partition(mrcal_amd_problem_t* P, int* info)
{
    info[0] = P->nd.S_split; info[1] = P->nd.S_shift; info[2] = P->nd.E_state0; info[3] = P->nd.elim_extrinsics;
}
int mrcal_amd_problem_shard_info(mrcal_amd_problem_t* P, int* info, int Ninfo)
{
    const int all[MRCAL_AMD_SHARD_INFO_N] = {
        P->nd.Nstate, P->nd.S_split, P->nd.NE, P->nd.Nc,
        P->br.frame_lo, P->br.frame_hi, P->is_leader ? 1 : 0,
        P->D.Nobs_board * P->D.W * P->D.H,
        P->nd.Nfb, P->nd.Npb,
        P->br.point_lo, P->br.point_hi };      // point BLOCKS (indices >= Nfb) this shard owns
    for(int i = 0; i < Ninfo && i < MRCAL_AMD_SHARD_INFO_N; i++) info[i] = all[i];
    return MRCAL_AMD_SHARD_INFO_N;
}
// outlier statistics / marking on the local board observations
// (mrcal.c:3978-4402). counts (device int[4]) and sums (device double[1]) are
// accumulated into, not cleared
bool mrcal_amd_problem_phase_outlier_stats(mrcal_amd_problem_t* P, int iop, double thresh_sq,
                                           int* counts_dev, double* sums_dev)
{
    const int Npts = P->D.Nobs_board * P->D.W * P->D.H;
    HIP_TRY(launch_outlier_stats(Npts, thresh_sq, P->op[iop & 1].x, P->d_board_pool, counts_dev, sums_dev, P->d_outlier_part, P->stream), return false);
    return true;
}
bool mrcal_amd_problem_phase_mark_outliers(mrcal_amd_problem_t* P, int iop, double thresh_sq, int* counts_dev)
{
    const int Npts = P->D.Nobs_board * P->D.W * P->D.H;
    HIP_TRY(launch_mark_outliers(Npts, thresh_sq, P->op[iop & 1].x, P->d_board_pool, counts_dev, P->stream), return false);
    return true;
}
void mrcal_amd_problem_set_current(mrcal_amd_problem_t* P, int iop) { P->icur = iop & 1; }

bool mrcal_amd_problem_sharded_reset(mrcal_amd_problem_t* P, int check_termination, int max_iterations,
                                     double trustregion0)
{
    last_error_string().clear();
    if(!problem_prepare_solver(P)) return false;
    DoglegParameters prm;
    if(max_iterations > 0)  prm.max_iterations = max_iterations;
    if(trustregion0 > 0.0)  prm.trustregion0   = trustregion0;
    P->sharded_external = true;     // the caller sums comm_buffer(0), comm_buffer(1) over the shards itself
    P->F.cperm_cur = NULL; P->plan.spl_compact = 0; P->plan.nd_lim = NULL; P->F.nd_lim.rounds = 0;
    P->stats.lambda = 0.0;          // a new run starts unregularized, like a new libdogleg context
    return ctl_reset(P, prm, check_termination != 0);
}

// segment 0: up to the first sum over the shards; 1: from there to the second
bool mrcal_amd_problem_sharded_enqueue(mrcal_amd_problem_t* P, int initial, int segment)
{
    if(!P->ctl_initialized) { set_error("mrcal_amd_problem_sharded_reset() first"); return false; }
    SolverCtl* ctl = P->d_ctl;
    const bool init = initial != 0;
    const Step2Args a = step2_args(P);
    if(segment == 0)
    {
        if(init)
        {
            const OpRef R = { P->d_ops, &ctl->ib, NULL };
            if(!problem_evaluate_ref(P, R, true, true, EVAL_PART_PROLOGUE | EVAL_PART_ZERO | EVAL_PART_BOARD | EVAL_PART_REST)) return false;
        }
        else
        {
            const OpRef Rto = { P->d_ops, &ctl->ia, solver_ctl_skip_eval2(ctl) };
            const int parts = EVAL_PART_PROLOGUE | EVAL_PART_ZERO | EVAL_PART_BOARD | EVAL_PART_REST;
            if(prologue_takes_choose(P->D))
            {
                const ChooseArgs ca = step2_choose_args(a);
                if(!problem_evaluate_ref(P, Rto, true, true, parts, NULL, &ca)) return false;
            }
            else
            {
                HIP_TRY(launch_step2_choose(a, P->stream), return false);
                if(!problem_evaluate_ref(P, Rto, true, true, parts)) return false;
            }
        }
        HIP_TRY(launch_step2_assemble(a, init, P->stream), return false);
        HIP_TRY(launch_step2_reduce(a, P->stream), return false);
        return true;
    }
    if(segment == 1)
    {
        HIP_TRY(launch_step2_factor(a, init, P->stream), return false);
        return true;
    }
    set_error("mrcal_amd_problem_sharded_enqueue(): segment %d", segment);
    return false;
}

// the buffer to sum over the shards after segment 0 / 1
void* mrcal_amd_problem_sharded_comm_buffer(mrcal_amd_problem_t* P, int which, int64_t* Nelements)
{
    if(!problem_prepare_solver(P)) return NULL;
    void* p = NULL; int64_t n = 0;
    switch(which)
    {
    case 0: p = P->F.S;    n = step2_comm1_doubles(P->nd); break;
    case 1: p = P->d_comm; n = 4; break;
    default: set_error("mrcal_amd_problem_sharded_comm_buffer(): %d", which);
    }
    if(Nelements) *Nelements = n;
    return p;
}

// queues a copy of the control block into slot (0..7) of the pinned ring
bool mrcal_amd_problem_sharded_snapshot(mrcal_amd_problem_t* P, int slot)
{
    if(!ctl_prepare(P)) return false;
    slot = ((slot % CTL_RING) + CTL_RING) % CTL_RING;
    HIP_TRY(hipMemcpyAsync(&P->h_ctl_ring[slot], P->d_ctl, sizeof(SolverCtl), hipMemcpyDeviceToHost, P->stream), return false);
    HIP_TRY(hipEventRecord(P->ctl_events[slot], P->stream), return false);
    return true;
}
// waits for that copy. out: done, error, Nsteps_accepted, Ntrials
bool mrcal_amd_problem_sharded_wait(mrcal_amd_problem_t* P, int slot, int* out)
{
    slot = ((slot % CTL_RING) + CTL_RING) % CTL_RING;
    HIP_TRY(hipEventSynchronize(P->ctl_events[slot]), return false);
    const SolverCtl& c = P->h_ctl_ring[slot];
    out[0] = c.done; out[1] = c.error; out[2] = c.Nsteps_accepted; out[3] = c.Ntrials;
    return true;
}
// drains the stream and makes the final point current. out_i: Nsteps_accepted,
// Nevaluations, Nfactorizations, Ntrials, error; out_d: trustregion, |x|^2, lambda
bool mrcal_amd_problem_sharded_finish(mrcal_amd_problem_t* P, int* out_i, double* out_d)
{
    SolverCtl c;
    if(!read_ctl(P, &c)) return false;
    absorb_ctl(P, c);
    out_i[0] = c.Nsteps_accepted; out_i[1] = c.Nevaluations; out_i[2] = c.Nfactorizations;
    out_i[3] = c.Ntrials;         out_i[4] = c.error;
    out_d[0] = c.trustregion;     out_d[1] = c.norm2_x[c.ib]; out_d[2] = c.lambda;
    return true;
}
int mrcal_amd_problem_current(mrcal_amd_problem_t* P) { return P->icur; }

// The outlier bits of this problem's (this shard's) triangulated observations as mark_outliers() left them: flags[i]
// belongs to observations_point_triangulated[*first + i] of the array the problem was created from. Returns how many
int mrcal_amd_problem_get_triangulated_outliers(mrcal_amd_problem_t* P, int* flags, int Nflags, int* first)
{
    const int N = P->tri_outlier_host.empty() ? 0 : (int)P->tri_outlier_host.size() - 1;     // (one element of padding)
    if(first) *first = P->tri_obs0;
    for(int i = 0; i < N && i < Nflags; i++) flags[i] = P->tri_outlier_host[i];
    return N;
}

// copies the (possibly outlier-marked) board observation pool back
bool mrcal_amd_problem_get_board_pool(mrcal_amd_problem_t* P, mrcal_point3_t* pool_local)
{
    const size_t n = (size_t)P->D.Nobs_board*P->D.W*P->D.H;
    if(n == 0) return true;
    HIP_TRY(hipMemcpyAsync(pool_local, P->d_board_pool, n*sizeof(mrcal_point3_t), hipMemcpyDeviceToHost, P->stream), return false);
    HIP_TRY(hipStreamSynchronize(P->stream), return false);
    return true;
}

////////////////////////////////////////////////////////////////////////////////
// drop-in: the whole solve
////////////////////////////////////////////////////////////////////////////////
mrcal_stats_t
mrcal_optimize( double* b_packed, int buffer_size_b_packed,
                double* x,        int buffer_size_x,
                double*                 intrinsics,
                mrcal_pose_t*           rt_cam_ref,
                mrcal_pose_t*           rt_ref_frame,
                mrcal_point3_t*         points,
                mrcal_calobject_warp_t* calobject_warp,
                int Ncameras_intrinsics, int Ncameras_extrinsics, int Nframes,
                int Npoints, int Npoints_fixed,
                const mrcal_observation_board_t* observations_board,
                const mrcal_observation_point_t* observations_point,
                int Nobservations_board,
                int Nobservations_point,
                const mrcal_observation_point_triangulated_t* observations_point_triangulated,
                int Nobservations_point_triangulated,
                mrcal_point3_t* observations_board_pool,
                mrcal_point3_t* observations_point_pool,
                const mrcal_lensmodel_t* lensmodel,
                const int* imagersizes,
                mrcal_problem_selections_t       problem_selections,
                const mrcal_problem_constants_t* problem_constants,
                double calibration_object_spacing,
                int calibration_object_width_n,
                int calibration_object_height_n,
                bool verbose,
                bool check_gradient)
{
    (void)problem_constants;
    last_error_string().clear();
    const mrcal_stats_t failed = { -1.0, 0, 0 };
    if(Nobservations_board > 0 && problem_selections.do_optimize_calobject_warp && calobject_warp == NULL)
    {
        set_error("ERROR: We're optimizing the calibration object warp, so a buffer with a seed MUST be passed in.");
        return failed;
    }
    if(observations_point_triangulated != NULL && Nobservations_point_triangulated &&
       !(!problem_selections.do_optimize_intrinsics_core &&
         !problem_selections.do_optimize_intrinsics_distortions &&
         problem_selections.do_optimize_extrinsics))
    {
        set_error("ERROR: We have triangulated points. At this time this is only allowed if we're NOT optimizing intrinsics AND if we ARE optimizing extrinsics.");
        return failed;
    }
    const mrcal_problem_selections_t sel =
        effective_selections(problem_selections, *lensmodel, Nobservations_board);
    if(!sel.do_optimize_intrinsics_core && !sel.do_optimize_intrinsics_distortions &&
       !sel.do_optimize_extrinsics      && !sel.do_optimize_frames &&
       !sel.do_optimize_calobject_warp)
        fprintf(stderr, "mrcal_amd: Warning: Not optimizing any of our variables\n");

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
    if(P == NULL) return failed;

    mrcal_stats_t stats = failed;
    const int Nstate = P->L.Nstate, Nmeas = P->L.Nmeas;
    std::vector<double> b(Nstate > 0 ? Nstate : 1);
    int Noutliers = 0;
    double rms;

    if(b_packed != NULL && buffer_size_b_packed != Nstate*(int)sizeof(double))
    {
        set_error("The buffer passed to fill-in b_packed_final has the wrong size. Needed exactly %d bytes, but got %d bytes",
                  Nstate*(int)sizeof(double), buffer_size_b_packed);
        goto done;
    }
    if(x != NULL && buffer_size_x != Nmeas*(int)sizeof(double))
    {
        set_error("The buffer passed to fill-in x_final has the wrong size. Needed exactly %d bytes, but got %d bytes",
                  Nmeas*(int)sizeof(double), buffer_size_x);
        goto done;
    }
    if(Nmeas <= Nstate)
        fprintf(stderr, "mrcal_amd: WARNING: problem isn't overdetermined: Nmeasurements=%d, Nstate=%d. Solver may not converge, and if it does, the results aren't reliable\n",
                Nmeas, Nstate);

    P->verbose = verbose;
    // (round 6) mrcal_optimize() returns no Jacobian (mrcal.h:453-521) and the problem does not outlive the call:
    // nothing can read the 300 MB a step would stream, unless the process was told to stream them all the same
    // (mrcal_amd_set_optimize_jacobian_stream(): the benchmark's metric is defined with the stream)
    mrcal_amd_problem_set_jacobian_stream(P, optimize_jacobian_stream_policy() ? 1 : 0);
    if(check_gradient)
    {
        // mrcal.c:6600-6605: no solve; libdogleg's dogleg_testGradient() for every
        // state variable: the reported gradient (a column of J) beside a central
        // difference of x, one line per measurement, vnlog on stdout. The
        // reference then returns sqrt(-1/Nmeasurements) for the rms: not a number
        if(!test_gradients(P)) goto done;
        stats.rms_reproj_error__pixels     = sqrt(-1.0);
        stats.Noutliers_board              = 0;
        stats.Noutliers_triangulated_point = 0;
        goto done;
    }

    // input outliers count even if nothing new is found (mrcal.c:6418-6421)
    for(int i=0; i<P->L.Nmeas_boards/2; i++)
        if(observations_board_pool[i].z < 0.0) Noutliers++;

    {
        int Nout_solve = 0;
        rms = mrcal_amd_problem_solve(P, 0, &Nout_solve);
        if(rms < 0.0) goto done;
        if(sel.do_apply_outlier_rejection) Noutliers = Nout_solve;
    }

    if(!mrcal_amd_problem_get_b_packed(P, b.data())) goto done;
    unpack_state_to_arrays(b.data(), P->L, intrinsics, rt_cam_ref, rt_ref_frame, points, calobject_warp);
    if(b_packed) memcpy(b_packed, b.data(), (size_t)Nstate*sizeof(double));
    if(x && !mrcal_amd_problem_get_x(P, x)) goto done;
    // new outliers are reported by negated weights in the caller's array
    if(sel.do_apply_outlier_rejection && P->stats.Noutlier_passes > 0)
        if(!mrcal_amd_problem_get_board_pool(P, observations_board_pool)) goto done;
    // ... and by the outlier bit of the triangulated observations, which lives in the caller's (const) array: the
    // reference's markOutliers() writes it there through the same cast (mrcal.c:4225, 4375, 6467)
    if(sel.do_apply_outlier_rejection && observations_point_triangulated != NULL)
    {
        mrcal_observation_point_triangulated_t* o = (mrcal_observation_point_triangulated_t*)observations_point_triangulated;
        const int Nt = P->tri_outlier_host.empty() ? 0 : (int)P->tri_outlier_host.size() - 1;
        for(int i = 0; i < Nt && P->tri_obs0 + i < Nobservations_point_triangulated; i++)
            if(P->tri_outlier_host[i]) o[P->tri_obs0 + i].outlier = true;
    }

    stats.rms_reproj_error__pixels     = rms;
    stats.Noutliers_board              = Noutliers;
    stats.Noutliers_triangulated_point = P->stats.Noutliers_triangulated;
    if(verbose) report_regularization(P, sel);

 done:
    problem_destroy_later(P);
    return stats;
}

} // extern "C"
