// TEST INFRASTRUCTURE ONLY. Not part of the product; nothing under mrcal_amd/
// includes, links or calls this.
//
// Restatement of the third-party solver the reference links against but does
// not vendor: libdogleg (>= 0.15.3, doc/install.org:63; linked at
// Makefile:59) driving SuiteSparse CHOLMOD (Makefile:204). Call sites in the
// reference: mrcal.c:3234-3246 (dense, 2x2, inside mrcal_unproject) and
// mrcal.c:6289-6303, 6433-6439, 6603-6621 (sparse, the calibration solve).
//
// What is restated: Powell's dog-leg trust-region method as libdogleg
// publishes it (README / dogleg.c of github.com/dkogan/libdogleg):
//
//   operating point:  x, J at p;  g = Jt x
//   Cauchy step:      -(|g|^2 / |J g|^2) g
//   Gauss-Newton:     -(JtJ)^-1 g, Cholesky; if JtJ is not positive definite
//                     add lambda I with lambda = 1e-10, then x10 until it is
//   step selection:   Cauchy beyond the trust region -> scaled Cauchy step to
//                     the edge; else GN inside the trust region -> GN; else
//                     the point on the Cauchy->GN segment at the edge
//   acceptance:       rho = (|x|^2 - |x_new|^2) / (-2 g.s - |J s|^2);
//                     rho < decrease_threshold -> radius *= decrease_factor;
//                     rho > increase_threshold and step hit the edge ->
//                     radius *= increase_factor; rho > 0 -> accept
//   termination:      all |g_i| < Jt_x_threshold; step shorter than
//                     update_threshold; radius < trustregion_threshold;
//                     max_iterations
//   defaults:         max_iterations 100, trustregion0 1e3, decrease 0.1 @
//                     0.25, increase 2 @ 0.75, thresholds 1e-8
//
// CHOLMOD (supernodal=0, mrcal-pywrap.c:183) is replaced by a simplicial
// up-looking sparse Cholesky (the textbook algorithm of T. Davis, "Direct
// Methods for Sparse Linear Systems", ch. 4) with a static
// ascending-degree ordering, which is fill-optimal for the arrowhead
// structure of calibration problems.
//
// PARITY UNPINNED: no copy of libdogleg or CHOLMOD is available here, so the
// iteration-by-iteration trajectory cannot be pinned against the real thing.
// What IS pinned: the linear algebra (tests/test_oracle_dogleg.py checks the
// factorization against dense numpy solves, as the reference's
// test/test-CHOLMOD-factorization.py does) and convergence to the same
// optimum on the reference's callback.

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdbool.h>
#include "dogleg.h"

#define SAY(fmt, ...) fprintf(stderr, "dogleg_restated: " fmt "\n", ##__VA_ARGS__)

static int last_Nsteps, last_Ncallbacks, last_Nfactorizations;

// Oracle-only knob (not in libdogleg): caps the iterations of every
// subsequent dogleg_optimize2() call regardless of what the caller asked for,
// so that the benchmark's CPU baseline can time a BOUNDED number of the
// reference's iterations at full problem size. <=0: no cap
static int max_iterations_override = 0;
void dogleg_restated_set_max_iterations(int n) { max_iterations_override = n; }
// accumulated wall-clock seconds spent inside the callback / the factorization
static double seconds_callback, seconds_factorization;
void dogleg_restated_last_timing(double* callback, double* factorization)
{
    if(callback)      *callback      = seconds_callback;
    if(factorization) *factorization = seconds_factorization;
}
#include <time.h>
static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9*(double)ts.tv_nsec;
}
void dogleg_restated_last_counts(int* Nsteps, int* Ncallbacks, int* Nfactorizations)
{
    if(Nsteps)          *Nsteps          = last_Nsteps;
    if(Ncallbacks)      *Ncallbacks      = last_Ncallbacks;
    if(Nfactorizations) *Nfactorizations = last_Nfactorizations;
}

void dogleg_getDefaultParameters(dogleg_parameters2_t* parameters)
{
    parameters->max_iterations                 = 100;
    parameters->dogleg_debug                   = 0;
    parameters->trustregion0                   = 1.0e3;
    parameters->trustregion_decrease_factor    = 0.1;
    parameters->trustregion_decrease_threshold = 0.25;
    parameters->trustregion_increase_factor    = 2.0;
    parameters->trustregion_increase_threshold = 0.75;
    parameters->Jt_x_threshold                 = 1e-8;
    parameters->update_threshold               = 1e-8;
    parameters->trustregion_threshold          = 1e-8;
}

////////////////////////////////////////////////////////////////////////////////
// Simplicial sparse Cholesky of  C = P (Jt J + lambda I) Pt
////////////////////////////////////////////////////////////////////////////////
typedef struct dogleg_restated_factor_t
{
    int     n;
    int*    perm;    // new -> old
    int*    pinv;    // old -> new

    // the pattern of Jt this analysis was made for (pattern_matches())
    int     Nmeas, nnz;
    int*    Jt_p_analyzed;
    int*    Jt_i_analyzed;

    // J in CSC (= transpose of Jt); structure + a map into Jt's values
    int*    Jp;      // n+1
    int*    Jj;      // measurement index of each entry
    int*    Jsrc;    // index into Jt->x

    // upper triangle of C, CSC
    int*    Cp;
    int*    Ci;
    double* Cx;

    // L, CSC, diagonal first in each column
    int*    parent;
    int*    Lp;
    int*    Li;
    double* Lx;

    // workspace
    int*    w_int;   // 3n
    double* w_dbl;   // n
} factor_t;

static void factor_free(factor_t* F)
{
    if(F == NULL) return;
    free(F->perm); free(F->pinv);
    free(F->Jt_p_analyzed); free(F->Jt_i_analyzed);
    free(F->Jp); free(F->Jj); free(F->Jsrc);
    free(F->Cp); free(F->Ci); free(F->Cx);
    free(F->parent); free(F->Lp); free(F->Li); free(F->Lx);
    free(F->w_int); free(F->w_dbl);
    free(F);
}

// nonzero pattern of row k of L: reach of the entries of C(0:k-1,k) in the
// elimination tree. Returns top; pattern is s[top..n-1], topologically ordered
static int ereach(const factor_t* F, int k, int* s, int* mark /* n, ==k means marked */)
{
    const int n = F->n;
    int top = n;
    mark[k] = k;
    for(int p = F->Cp[k]; p < F->Cp[k+1]; p++)
    {
        int i = F->Ci[p];
        if(i > k) continue;
        int len = 0;
        for(; mark[i] != k; i = F->parent[i])
        {
            s[len++] = i;
            mark[i]  = k;
        }
        while(len > 0) s[--top] = s[--len];
    }
    return top;
}

static int cmp_degree(const void* a, const void* b)
{
    const long long* A = (const long long*)a;
    const long long* B = (const long long*)b;
    return (A[0] > B[0]) - (A[0] < B[0]);
}

// All the symbolic work. Jt is CSC (nrow=Nstate, ncol=Nmeas)
static factor_t* factor_analyze(int n, int Nmeas, const int* Jt_p, const int* Jt_i)
{
    factor_t* F = (factor_t*)calloc(1, sizeof(*F));
    F->n = n;
    const int nnz = Jt_p[Nmeas];
    F->Nmeas = Nmeas; F->nnz = nnz;
    F->Jt_p_analyzed = (int*)malloc(((size_t)Nmeas+1)*sizeof(int));
    F->Jt_i_analyzed = (int*)malloc((nnz>0?nnz:1)*sizeof(int));
    memcpy(F->Jt_p_analyzed, Jt_p, ((size_t)Nmeas+1)*sizeof(int));
    memcpy(F->Jt_i_analyzed, Jt_i, (size_t)nnz*sizeof(int));

    // J = transpose(Jt), structure only
    F->Jp   = (int*)calloc(n+1, sizeof(int));
    F->Jj   = (int*)malloc((nnz>0?nnz:1)*sizeof(int));
    F->Jsrc = (int*)malloc((nnz>0?nnz:1)*sizeof(int));
    for(int e=0; e<nnz; e++) F->Jp[Jt_i[e]+1]++;
    for(int i=0; i<n; i++)   F->Jp[i+1] += F->Jp[i];
    {
        int* next = (int*)malloc((n>0?n:1)*sizeof(int));
        memcpy(next, F->Jp, n*sizeof(int));
        for(int j=0; j<Nmeas; j++)
            for(int e=Jt_p[j]; e<Jt_p[j+1]; e++)
            {
                int q = next[Jt_i[e]]++;
                F->Jj[q]   = j;
                F->Jsrc[q] = e;
            }
        free(next);
    }

    F->w_int = (int*)   malloc((3*n>0?3*n:1)*sizeof(int));
    F->w_dbl = (double*)calloc((n>0?n:1),     sizeof(double));
    int* mark = F->w_int;

    // pass 1: degrees of the columns of JtJ, for the ordering
    long long* deg = (long long*)malloc((n>0?n:1)*2*sizeof(long long));
    for(int i=0; i<n; i++) mark[i] = -1;
    for(int b=0; b<n; b++)
    {
        long long count = 0;
        for(int q=F->Jp[b]; q<F->Jp[b+1]; q++)
        {
            int j = F->Jj[q];
            for(int e=Jt_p[j]; e<Jt_p[j+1]; e++)
            {
                int a = Jt_i[e];
                if(mark[a] != b) { mark[a] = b; count++; }
            }
        }
        deg[2*b+0] = count*(long long)n + b; // ties broken by index
        deg[2*b+1] = b;
    }
    qsort(deg, n, 2*sizeof(long long), cmp_degree);
    F->perm = (int*)malloc((n>0?n:1)*sizeof(int));
    F->pinv = (int*)malloc((n>0?n:1)*sizeof(int));
    for(int i=0; i<n; i++)
    {
        F->perm[i]               = (int)deg[2*i+1];
        F->pinv[(int)deg[2*i+1]] = i;
    }
    free(deg);

    // pass 2: pattern of upper(C), C = P JtJ Pt. The diagonal is always present
    F->Cp = (int*)calloc(n+1, sizeof(int));
    for(int pass=0; pass<2; pass++)
    {
        for(int i=0; i<n; i++) mark[i] = -1;
        int count_all = 0;
        for(int bnew=0; bnew<n; bnew++)
        {
            const int b = F->perm[bnew];
            if(pass) F->Ci[count_all] = bnew;
            count_all++;
            mark[bnew] = bnew;
            for(int q=F->Jp[b]; q<F->Jp[b+1]; q++)
            {
                int j = F->Jj[q];
                for(int e=Jt_p[j]; e<Jt_p[j+1]; e++)
                {
                    int anew = F->pinv[Jt_i[e]];
                    if(anew < bnew && mark[anew] != bnew)
                    {
                        mark[anew] = bnew;
                        if(pass) F->Ci[count_all] = anew;
                        count_all++;
                    }
                }
            }
            F->Cp[bnew+1] = count_all;
        }
        if(!pass)
        {
            F->Ci = (int*)   malloc((count_all>0?count_all:1)*sizeof(int));
            F->Cx = (double*)malloc((count_all>0?count_all:1)*sizeof(double));
        }
    }

    // elimination tree of C
    F->parent = (int*)malloc((n>0?n:1)*sizeof(int));
    {
        int* ancestor = F->w_int + n;
        for(int k=0; k<n; k++)
        {
            F->parent[k] = -1;
            ancestor[k]  = -1;
            for(int p=F->Cp[k]; p<F->Cp[k+1]; p++)
            {
                int i = F->Ci[p];
                while(i != -1 && i < k)
                {
                    int inext   = ancestor[i];
                    ancestor[i] = k;
                    if(inext == -1) F->parent[i] = k;
                    i = inext;
                }
            }
        }
    }

    // column counts of L by walking the row patterns
    F->Lp = (int*)calloc(n+1, sizeof(int));
    {
        int* s = F->w_int + n;
        for(int i=0; i<n; i++) mark[i] = -1;
        for(int k=0; k<n; k++)
        {
            F->Lp[k+1]++; // diagonal
            int top = ereach(F, k, s, mark);
            for(int t=top; t<n; t++) F->Lp[s[t]+1]++;
        }
        for(int k=0; k<n; k++) F->Lp[k+1] += F->Lp[k];
    }
    F->Li = (int*)   malloc((F->Lp[n]>0?F->Lp[n]:1)*sizeof(int));
    F->Lx = (double*)malloc((F->Lp[n]>0?F->Lp[n]:1)*sizeof(double));
    return F;
}

// Is this the pattern the analysis was made for? The splined lens models move a row's columns with the corner
// (which (order+1)^2 knots a projection lands on: mrcal.c:4718-4817), so Jt's pattern changes from one evaluation to
// the next. CHOLMOD's simplicial numeric factorization (cholmod_rowfac, which libdogleg selects with supernodal = 0)
// takes its elimination tree from the growing factor itself and is correct for ANY pattern under the analyzed
// ordering; this restatement's symbolic work is static (the map Jsrc into Jt's values, the pattern of C, the
// column pointers of L), so it is redone when the pattern moved. (Until round 4 it was not: the numeric phase then
// read values of the new Jt through the old map, found "not positive definite" matrices that LAPACK factors without
// trouble - tools/diag_splined_pd.py - and crawled through lambda: every disputed splined solve of
// profiles/r03_fuzz_parity.txt)
static bool pattern_matches(const factor_t* F, int Nmeas, const int* Jt_p, const int* Jt_i)
{
    return F->Nmeas == Nmeas && F->nnz == Jt_p[Nmeas] &&
           memcmp(F->Jt_p_analyzed, Jt_p, ((size_t)Nmeas+1)*sizeof(int)) == 0 &&
           memcmp(F->Jt_i_analyzed, Jt_i, (size_t)F->nnz*sizeof(int)) == 0;
}

// Numeric factorization. Returns false if not positive definite
static bool factor_numeric(factor_t* F, int Nmeas,
                           const int* Jt_p, const int* Jt_i, const double* Jt_x,
                           double lambda)
{
    const int n   = F->n;
    double*   x   = F->w_dbl;
    int*      mark= F->w_int;
    int*      s   = F->w_int + n;
    int*      c   = F->w_int + 2*n;

    // values of upper(C), Gustavson-style with a dense accumulator
    for(int i=0; i<n; i++) x[i] = 0.0;
    for(int bnew=0; bnew<n; bnew++)
    {
        const int b = F->perm[bnew];
        for(int q=F->Jp[b]; q<F->Jp[b+1]; q++)
        {
            const int    j   = F->Jj[q];
            const double Jjb = Jt_x[F->Jsrc[q]];
            for(int e=Jt_p[j]; e<Jt_p[j+1]; e++)
            {
                int anew = F->pinv[Jt_i[e]];
                if(anew <= bnew) x[anew] += Jt_x[e]*Jjb;
            }
        }
        for(int p=F->Cp[bnew]; p<F->Cp[bnew+1]; p++)
        {
            int i = F->Ci[p];
            F->Cx[p] = x[i] + (i==bnew ? lambda : 0.0);
            x[i] = 0.0;
        }
    }

    for(int k=0; k<n; k++) { c[k] = F->Lp[k]; mark[k] = -1; }
    for(int k=0; k<n; k++)
    {
        int top = ereach(F, k, s, mark);
        x[k] = 0.0;
        for(int p=F->Cp[k]; p<F->Cp[k+1]; p++)
            if(F->Ci[p] <= k) x[F->Ci[p]] = F->Cx[p];
        double d = x[k];
        x[k] = 0.0;
        for(; top<n; top++)
        {
            int    i   = s[top];
            double lki = x[i] / F->Lx[F->Lp[i]];
            x[i] = 0.0;
            for(int p=F->Lp[i]+1; p<c[i]; p++)
                x[F->Li[p]] -= F->Lx[p]*lki;
            d -= lki*lki;
            int p = c[i]++;
            F->Li[p] = k;
            F->Lx[p] = lki;
        }
        if(!(d > 0.0))
        {
            // clean up the accumulator before reporting failure
            for(int i=0; i<n; i++) x[i] = 0.0;
            return false;
        }
        int p = c[k]++;
        F->Li[p] = k;
        F->Lx[p] = sqrt(d);
    }
    return true;
}

// solves (JtJ + lambda I) x = b in place
static void factor_solve(const factor_t* F, double* b)
{
    const int n = F->n;
    double*   y = F->w_dbl;
    for(int i=0; i<n; i++) y[i] = b[F->perm[i]];
    for(int j=0; j<n; j++)
    {
        y[j] /= F->Lx[F->Lp[j]];
        for(int p=F->Lp[j]+1; p<F->Lp[j+1]; p++)
            y[F->Li[p]] -= F->Lx[p]*y[j];
    }
    for(int j=n-1; j>=0; j--)
    {
        for(int p=F->Lp[j]+1; p<F->Lp[j+1]; p++)
            y[j] -= F->Lx[p]*y[F->Li[p]];
        y[j] /= F->Lx[F->Lp[j]];
    }
    for(int i=0; i<n; i++) b[F->perm[i]] = y[i];
    for(int i=0; i<n; i++) y[i] = 0.0;
}

bool dogleg_restated_solve_JtJ(double* b, int Nrhs,
                               int Nstate, int Nmeas,
                               const int* Jt_p, const int* Jt_i, const double* Jt_x)
{
    factor_t* F = factor_analyze(Nstate, Nmeas, Jt_p, Jt_i);
    bool ok = factor_numeric(F, Nmeas, Jt_p, Jt_i, Jt_x, 0.0);
    if(ok)
        for(int i=0; i<Nrhs; i++)
            factor_solve(F, &b[(size_t)i*Nstate]);
    factor_free(F);
    return ok;
}

////////////////////////////////////////////////////////////////////////////////
// dense Cholesky, for dogleg_optimize_dense2 (2x2 problems in mrcal_unproject)
////////////////////////////////////////////////////////////////////////////////
static bool dense_cholesky(double* A /* n*n, lower used, overwritten by L */, int n)
{
    for(int j=0; j<n; j++)
    {
        double d = A[j*n+j];
        for(int k=0; k<j; k++) d -= A[j*n+k]*A[j*n+k];
        if(!(d > 0.0)) return false;
        d = sqrt(d);
        A[j*n+j] = d;
        for(int i=j+1; i<n; i++)
        {
            double v = A[i*n+j];
            for(int k=0; k<j; k++) v -= A[i*n+k]*A[j*n+k];
            A[i*n+j] = v/d;
        }
    }
    return true;
}
static void dense_cholesky_solve(const double* L, int n, double* b)
{
    for(int i=0; i<n; i++)
    {
        double v = b[i];
        for(int k=0; k<i; k++) v -= L[i*n+k]*b[k];
        b[i] = v / L[i*n+i];
    }
    for(int i=n-1; i>=0; i--)
    {
        double v = b[i];
        for(int k=i+1; k<n; k++) v -= L[k*n+i]*b[k];
        b[i] = v / L[i*n+i];
    }
}

////////////////////////////////////////////////////////////////////////////////
// operating points
////////////////////////////////////////////////////////////////////////////////
static dogleg_operatingPoint_t* allocOperatingPoint(int Nstate, int Nmeas, int NJnnz, int is_sparse)
{
    dogleg_operatingPoint_t* point = (dogleg_operatingPoint_t*)calloc(1, sizeof(*point));
    // one pool: p, x, Jt_x, updateCauchy, updateGN
    double* pool = (double*)calloc((size_t)Nstate*4 + Nmeas + 1, sizeof(double));
    point->p            = pool;
    point->x            = &pool[Nstate];
    point->Jt_x         = &pool[Nstate+Nmeas];
    point->updateCauchy = &pool[Nstate*2+Nmeas];
    point->updateGN     = &pool[Nstate*3+Nmeas];
    if(is_sparse)
    {
        cholmod_sparse* Jt = (cholmod_sparse*)calloc(1, sizeof(*Jt));
        Jt->nrow   = Nstate;
        Jt->ncol   = Nmeas;
        Jt->nzmax  = NJnnz;
        Jt->p      = calloc((size_t)Nmeas+1, sizeof(int));
        Jt->i      = calloc((size_t)(NJnnz>0?NJnnz:1), sizeof(int));
        Jt->x      = calloc((size_t)(NJnnz>0?NJnnz:1), sizeof(double));
        Jt->sorted = 1;
        Jt->packed = 1;
        point->Jt  = Jt;
    }
    else
        point->J_dense = (double*)calloc((size_t)Nmeas*Nstate, sizeof(double));
    return point;
}
static void freeOperatingPoint(dogleg_operatingPoint_t** point, int is_sparse)
{
    if(*point == NULL) return;
    free((*point)->p);
    if(is_sparse)
    {
        free((*point)->Jt->p); free((*point)->Jt->i); free((*point)->Jt->x);
        free((*point)->Jt);
    }
    else
        free((*point)->J_dense);
    free(*point);
    *point = NULL;
}

static double norm2(const double* v, int N)
{
    double s = 0.0;
    for(int i=0; i<N; i++) s += v[i]*v[i];
    return s;
}

// norm2(J v)
static double norm2_J_v(const dogleg_solverContext_t* ctx, const dogleg_operatingPoint_t* point, const double* v)
{
    double result = 0.0;
    if(ctx->is_sparse)
    {
        const int*    P = (const int*)   point->Jt->p;
        const int*    I = (const int*)   point->Jt->i;
        const double* X = (const double*)point->Jt->x;
        for(int j=0; j<ctx->Nmeasurements; j++)
        {
            double dot = 0.0;
            for(int e=P[j]; e<P[j+1]; e++) dot += X[e]*v[I[e]];
            result += dot*dot;
        }
    }
    else
        for(int j=0; j<ctx->Nmeasurements; j++)
        {
            double dot = 0.0;
            for(int i=0; i<ctx->Nstate; i++) dot += point->J_dense[(size_t)j*ctx->Nstate+i]*v[i];
            result += dot*dot;
        }
    return result;
}

// Evaluates the callback at point->p. Returns true if the gradient is below
// the Jt_x threshold everywhere
static bool computeCallbackOperatingPoint(dogleg_operatingPoint_t* point, dogleg_solverContext_t* ctx)
{
    ctx->Ncallbacks++;
    if(ctx->is_sparse)
    {
        const double t0 = now();
        (*ctx->f)(point->p, point->x, point->Jt, ctx->cookie);
        seconds_callback += now() - t0;
        const int*    P = (const int*)   point->Jt->p;
        const int*    I = (const int*)   point->Jt->i;
        const double* X = (const double*)point->Jt->x;
        for(int i=0; i<ctx->Nstate; i++) point->Jt_x[i] = 0.0;
        for(int j=0; j<ctx->Nmeasurements; j++)
        {
            const double xj = point->x[j];
            for(int e=P[j]; e<P[j+1]; e++) point->Jt_x[I[e]] += X[e]*xj;
        }
    }
    else
    {
        (*ctx->f_dense)(point->p, point->x, point->J_dense, ctx->cookie);
        for(int i=0; i<ctx->Nstate; i++) point->Jt_x[i] = 0.0;
        for(int j=0; j<ctx->Nmeasurements; j++)
            for(int i=0; i<ctx->Nstate; i++)
                point->Jt_x[i] += point->J_dense[(size_t)j*ctx->Nstate+i]*point->x[j];
    }

    point->norm2_x                    = norm2(point->x, ctx->Nmeasurements);
    point->updateCauchy_valid         = 0;
    point->updateGN_valid             = 0;
    point->didStepToEdgeOfTrustRegion = 0;

    for(int i=0; i<ctx->Nstate; i++)
        if(fabs(point->Jt_x[i]) >= ctx->parameters->Jt_x_threshold)
            return false;
    return true;
}

static void computeCauchyUpdate(dogleg_operatingPoint_t* point, const dogleg_solverContext_t* ctx)
{
    if(point->updateCauchy_valid) return;
    point->updateCauchy_valid = 1;

    const double norm2_Jt_x   = norm2(point->Jt_x, ctx->Nstate);
    const double norm2_J_Jt_x = norm2_J_v(ctx, point, point->Jt_x);
    const double k            = -norm2_Jt_x / norm2_J_Jt_x;

    point->updateCauchy_lensq = k*k * norm2_Jt_x;
    for(int i=0; i<ctx->Nstate; i++) point->updateCauchy[i] = k*point->Jt_x[i];

    if(ctx->parameters->dogleg_debug)
        SAY("cauchy step size %.6g", sqrt(point->updateCauchy_lensq));
}

// Oracle-only knob (not in libdogleg): DOGLEG_RESTATED_DUMP_NOTPD=<prefix> writes the sparse Jt and lambda of every
// factorization that was declared "not positive definite" to <prefix><k>.bin (int32 Nstate, Nmeas, nnz; double
// lambda; int32 p[Nmeas+1], i[nnz]; double x[nnz]) so that the decision can be put to LAPACK on the same matrix
// (tools/diag_splined_pd.py)
static void dump_not_positive_definite(const dogleg_solverContext_t* ctx, const dogleg_operatingPoint_t* point)
{
    static int ndumped = 0;
    const char* prefix = getenv("DOGLEG_RESTATED_DUMP_NOTPD");
    if(prefix == NULL || !ctx->is_sparse || ndumped >= 64) return;
    char path[1024];
    snprintf(path, sizeof(path), "%s%d.bin", prefix, ndumped++);
    FILE* f = fopen(path, "wb");
    if(f == NULL) return;
    const int    n = ctx->Nstate, m = ctx->Nmeasurements;
    const int*   p = (const int*)point->Jt->p;
    const int  nnz = p[m];
    fwrite(&n, sizeof(int), 1, f); fwrite(&m, sizeof(int), 1, f); fwrite(&nnz, sizeof(int), 1, f);
    fwrite(&ctx->lambda, sizeof(double), 1, f);
    fwrite(p, sizeof(int), (size_t)m + 1, f);
    fwrite(point->Jt->i, sizeof(int), (size_t)nnz, f);
    fwrite(point->Jt->x, sizeof(double), (size_t)nnz, f);
    fclose(f);
}

static void computeGaussNewtonUpdate(dogleg_operatingPoint_t* point, dogleg_solverContext_t* ctx)
{
    if(point->updateGN_valid) return;
    point->updateGN_valid = 1;

    const int n = ctx->Nstate;
    if(!ctx->factorization_valid)
    {
        ctx->factorization_valid = 1;
        while(1)
        {
            bool ok;
            const double t0 = now();
            ctx->Nfactorizations++;
            if(ctx->is_sparse)
            {
                if(ctx->factorization != NULL &&
                   !pattern_matches(ctx->factorization, ctx->Nmeasurements, (const int*)point->Jt->p, (const int*)point->Jt->i))
                {
                    factor_free(ctx->factorization);
                    ctx->factorization = NULL;
                }
                if(ctx->factorization == NULL)
                    ctx->factorization = factor_analyze(n, ctx->Nmeasurements,
                                                        (const int*)point->Jt->p,
                                                        (const int*)point->Jt->i);
                ok = factor_numeric(ctx->factorization, ctx->Nmeasurements,
                                    (const int*)   point->Jt->p,
                                    (const int*)   point->Jt->i,
                                    (const double*)point->Jt->x,
                                    ctx->lambda);
            }
            else
            {
                if(ctx->factorization_dense == NULL)
                    ctx->factorization_dense = (double*)malloc((size_t)n*n*sizeof(double));
                double* A = ctx->factorization_dense;
                for(int i=0; i<n; i++)
                    for(int k=0; k<=i; k++)
                    {
                        double s = (i==k) ? ctx->lambda : 0.0;
                        for(int j=0; j<ctx->Nmeasurements; j++)
                            s += point->J_dense[(size_t)j*n+i]*point->J_dense[(size_t)j*n+k];
                        A[i*n+k] = s;
                    }
                ok = dense_cholesky(A, n);
            }
            seconds_factorization += now() - t0;
            if(ok) break;
            dump_not_positive_definite(ctx, point);

            // singular JtJ. Raise lambda and go again
            if(ctx->lambda == 0.0) ctx->lambda = 1e-10;
            else                   ctx->lambda *= 10.0;
            if(ctx->parameters->dogleg_debug)
                SAY("singular JtJ. Have rank/full rank: ?/%d. Adding %g I from now on", n, ctx->lambda);
            if(!(ctx->lambda < 1e30))
            {
                SAY("giving up on making JtJ positive definite");
                break;
            }
        }
    }

    // solve JtJ*updateGN = Jt*x. Gauss-Newton step is then -updateGN
    memcpy(point->updateGN, point->Jt_x, n*sizeof(double));
    if(ctx->is_sparse) factor_solve(ctx->factorization, point->updateGN);
    else               dense_cholesky_solve(ctx->factorization_dense, n, point->updateGN);
    for(int i=0; i<n; i++) point->updateGN[i] = -point->updateGN[i];
    point->updateGN_lensq = norm2(point->updateGN, n);

    if(ctx->parameters->dogleg_debug)
        SAY("gn step size %.6g", sqrt(point->updateGN_lensq));
}

static void computeInterpolatedUpdate(double* update_dogleg, double* update_dogleg_lensq,
                                      const dogleg_operatingPoint_t* point,
                                      double trustregion, const dogleg_solverContext_t* ctx)
{
    // norm2(a + k*(b-a)) = dsq, a = Cauchy, b = GN:
    //   l2 k^2 + 2 c k + norm2(a) - dsq = 0,  c = at (b-a), l2 = norm2(b-a)
    // and I want the root with 0 <= k <= 1: k = (-c + sqrt(c^2 - l2 (norm2(a)-dsq)))/l2
    const double  dsq    = trustregion*trustregion;
    const double  norm2a = point->updateCauchy_lensq;
    const double* a      = point->updateCauchy;
    const double* b      = point->updateGN;
    double l2 = 0.0, neg_c = 0.0;
    for(int i=0; i<ctx->Nstate; i++)
    {
        const double d = a[i] - b[i];
        l2    += d*d;
        neg_c += d*a[i];
    }
    double discriminant = neg_c*neg_c - l2*(norm2a - dsq);
    if(discriminant < 0.0)
    {
        SAY("negative discriminant: %.6g!", discriminant);
        discriminant = 0.0;
    }
    const double k = (neg_c + sqrt(discriminant))/l2;
    *update_dogleg_lensq = 0.0;
    for(int i=0; i<ctx->Nstate; i++)
    {
        update_dogleg[i] = a[i] + k*(b[i] - a[i]);
        *update_dogleg_lensq += update_dogleg[i]*update_dogleg[i];
    }
    if(ctx->parameters->dogleg_debug)
        SAY("k_cauchy_to_gn %.6g, norm %.6g", k, sqrt(*update_dogleg_lensq));
}

// takes a step from pointFrom. Returns the squared length of the step; the
// new parameter vector goes to p_new
static double takeStepFrom(dogleg_operatingPoint_t* pointFrom, double* p_new, double* step_scratch,
                           double trustregion, double* expectedImprovement,
                           dogleg_solverContext_t* ctx)
{
    const int n = ctx->Nstate;
    double        step_len_sq;
    const double* step;

    computeCauchyUpdate(pointFrom, ctx);
    if(pointFrom->updateCauchy_lensq >= trustregion*trustregion)
    {
        // The Cauchy step leaves the trust region: gradient descent to the edge
        const double k = trustregion / sqrt(pointFrom->updateCauchy_lensq);
        for(int i=0; i<n; i++) step_scratch[i] = k*pointFrom->updateCauchy[i];
        step        = step_scratch;
        step_len_sq = trustregion*trustregion;
        pointFrom->didStepToEdgeOfTrustRegion = 1;
    }
    else
    {
        computeGaussNewtonUpdate(pointFrom, ctx);
        if(pointFrom->updateGN_lensq <= trustregion*trustregion)
        {
            step        = pointFrom->updateGN;
            step_len_sq = pointFrom->updateGN_lensq;
            pointFrom->didStepToEdgeOfTrustRegion = 0;
        }
        else
        {
            computeInterpolatedUpdate(step_scratch, &step_len_sq, pointFrom, trustregion, ctx);
            step = step_scratch;
            pointFrom->didStepToEdgeOfTrustRegion = 1;
        }
    }

    for(int i=0; i<n; i++) p_new[i] = pointFrom->p[i] + step[i];

    // F(0) - F(step) = norm2(x) - norm2(x + J step) = -2 (Jt x).step - norm2(J step)
    double dot = 0.0;
    for(int i=0; i<n; i++) dot += pointFrom->Jt_x[i]*step[i];
    *expectedImprovement = -2.0*dot - norm2_J_v(ctx, pointFrom, step);
    return step_len_sq;
}

static int runOptimizer(dogleg_solverContext_t* ctx)
{
    // oracle-only knob: DOGLEG_RESTATED_TRACE=1 prints the iteration trace even
    // when the caller did not ask for dogleg_debug (mrcal's verbose also floods
    // stderr with every observation)
    dogleg_parameters2_t Ptrace = *ctx->parameters;
    if(getenv("DOGLEG_RESTATED_TRACE") != NULL) Ptrace.dogleg_debug = 1;
    const dogleg_parameters2_t* P = &Ptrace;
    double  trustregion = P->trustregion0;
    int     stepCount   = 0;
    double* step_scratch = (double*)malloc((ctx->Nstate>0?ctx->Nstate:1)*sizeof(double));

    if(computeCallbackOperatingPoint(ctx->beforeStep, ctx))
        goto done;

    if(P->dogleg_debug)
        SAY("Initial operating point has norm2_x %.6g", ctx->beforeStep->norm2_x);

    ctx->factorization_valid = 0;

    const int max_iterations =
        (max_iterations_override > 0 && max_iterations_override < P->max_iterations) ?
        max_iterations_override : P->max_iterations;
    while(stepCount < max_iterations)
    {
        while(1)
        {
            double expectedImprovement;
            double step_len_sq =
                takeStepFrom(ctx->beforeStep, ctx->afterStep->p, step_scratch,
                             trustregion, &expectedImprovement, ctx);

            if(step_len_sq < P->update_threshold*P->update_threshold)
            {
                if(P->dogleg_debug)
                    SAY("Step size too small (%.6g). Giving up", sqrt(step_len_sq));
                goto done;
            }

            bool afterStepZeroGradient = computeCallbackOperatingPoint(ctx->afterStep, ctx);
            const double observedImprovement = ctx->beforeStep->norm2_x - ctx->afterStep->norm2_x;
            const double rho = observedImprovement / expectedImprovement;

            if(P->dogleg_debug)
                SAY("step %d: norm2_x %.15g -> %.15g; expected improvement %.6g, got %.6g; rho %.4g; trustregion %.6g",
                    stepCount, ctx->beforeStep->norm2_x, ctx->afterStep->norm2_x,
                    expectedImprovement, observedImprovement, rho, trustregion);

            if(rho < P->trustregion_decrease_threshold)
                trustregion *= P->trustregion_decrease_factor;
            else if(rho > P->trustregion_increase_threshold &&
                    ctx->beforeStep->didStepToEdgeOfTrustRegion)
                trustregion *= P->trustregion_increase_factor;

            if(rho > 0.0)
            {
                // accept the step
                dogleg_operatingPoint_t* tmp = ctx->afterStep;
                ctx->afterStep  = ctx->beforeStep;
                ctx->beforeStep = tmp;
                ctx->factorization_valid = 0;
                if(afterStepZeroGradient)
                {
                    if(P->dogleg_debug) SAY("Gradient low enough and we just improved. Done iterating");
                    goto done;
                }
                break;
            }

            // rejected. Try again with the smaller trust region
            if(trustregion < P->trustregion_threshold)
            {
                if(P->dogleg_debug) SAY("Trust region too small. Giving up");
                goto done;
            }
            // a degenerate case the thresholds=0 settings of mrcal would
            // otherwise spin on forever
            if(trustregion == 0.0 || !(trustregion == trustregion))
                goto done;
        }
        stepCount++;
    }
    if(P->dogleg_debug && stepCount == max_iterations)
        SAY("Exceeded max number of iterations");

 done:
    free(step_scratch);
    return stepCount;
}

static double optimize_generic(double* p, int Nstate, int Nmeas, int NJnnz, int is_sparse,
                               dogleg_callback_t* f, dogleg_callback_dense_t* f_dense,
                               void* cookie,
                               const dogleg_parameters2_t* parameters,
                               dogleg_solverContext_t** returnContext)
{
    dogleg_solverContext_t* ctx = (dogleg_solverContext_t*)calloc(1, sizeof(*ctx));
    ctx->f             = f;
    ctx->f_dense       = f_dense;
    ctx->cookie        = cookie;
    ctx->is_sparse     = is_sparse;
    ctx->Nstate        = Nstate;
    ctx->Nmeasurements = Nmeas;
    ctx->NJnnz         = NJnnz;
    ctx->parameters    = parameters;
    ctx->beforeStep    = allocOperatingPoint(Nstate, Nmeas, NJnnz, is_sparse);
    ctx->afterStep     = allocOperatingPoint(Nstate, Nmeas, NJnnz, is_sparse);

    memcpy(ctx->beforeStep->p, p, Nstate*sizeof(double));
    if(is_sparse) seconds_callback = seconds_factorization = 0.0;
    ctx->Nsteps = runOptimizer(ctx);
    memcpy(p, ctx->beforeStep->p, Nstate*sizeof(double));

    last_Nsteps          = ctx->Nsteps;
    last_Ncallbacks      = ctx->Ncallbacks;
    last_Nfactorizations = ctx->Nfactorizations;

    const double norm2_x = ctx->beforeStep->norm2_x;
    if(returnContext != NULL) *returnContext = ctx;
    else                      dogleg_freeContext(&ctx);
    return norm2_x;
}

double dogleg_optimize2(double* p, unsigned int Nstate,
                        unsigned int Nmeas, unsigned int NJnnz,
                        dogleg_callback_t* f, void* cookie,
                        const dogleg_parameters2_t* parameters,
                        dogleg_solverContext_t** returnContext)
{
    return optimize_generic(p, (int)Nstate, (int)Nmeas, (int)NJnnz, 1, f, NULL, cookie, parameters, returnContext);
}

double dogleg_optimize_dense2(double* p, unsigned int Nstate,
                              unsigned int Nmeas,
                              dogleg_callback_dense_t* f, void* cookie,
                              const dogleg_parameters2_t* parameters,
                              dogleg_solverContext_t** returnContext)
{
    return optimize_generic(p, (int)Nstate, (int)Nmeas, (int)(Nstate*Nmeas), 0, NULL, f, cookie, parameters, returnContext);
}

void dogleg_freeContext(dogleg_solverContext_t** ctx)
{
    if(*ctx == NULL) return;
    freeOperatingPoint(&(*ctx)->beforeStep, (*ctx)->is_sparse);
    freeOperatingPoint(&(*ctx)->afterStep,  (*ctx)->is_sparse);
    factor_free((*ctx)->factorization);
    free((*ctx)->factorization_dense);
    free(*ctx);
    *ctx = NULL;
}

// Reports the analytic and central-difference gradients of every measurement
// with respect to one state variable, in vnlog format on stdout (this is
// what test/test-gradients.py of the reference parses)
void dogleg_testGradient(unsigned int var, const double* p0,
                         unsigned int Nstate, unsigned int Nmeas, unsigned int NJnnz,
                         dogleg_callback_t* f, void* cookie)
{
    const double delta = 1e-6;
    dogleg_operatingPoint_t* pt0 = allocOperatingPoint(Nstate, Nmeas, NJnnz, 1);
    dogleg_operatingPoint_t* pt1 = allocOperatingPoint(Nstate, Nmeas, NJnnz, 1);
    double* x0 = (double*)malloc(Nmeas*sizeof(double));

    memcpy(pt0->p, p0, Nstate*sizeof(double));
    memcpy(pt1->p, p0, Nstate*sizeof(double));
    (*f)(pt0->p, pt0->x, pt0->Jt, cookie);
    pt1->p[var] -= delta/2.0;
    (*f)(pt1->p, x0, pt1->Jt, cookie);
    pt1->p[var] += delta;
    (*f)(pt1->p, pt1->x, pt1->Jt, cookie);

    const int*    P = (const int*)   pt0->Jt->p;
    const int*    I = (const int*)   pt0->Jt->i;
    const double* X = (const double*)pt0->Jt->x;
    if(var == 0)
        printf("# ivar imeasurement gradient_reported gradient_observed error error_relative\n");
    for(unsigned int j=0; j<Nmeas; j++)
    {
        double g_rep = 0.0;
        for(int e=P[j]; e<P[j+1]; e++)
            if(I[e] == (int)var) g_rep += X[e];
        const double g_obs = (pt1->x[j] - x0[j]) / delta;
        const double err   = g_rep - g_obs;
        const double den   = (fabs(g_rep) + fabs(g_obs)) / 2.0;
        printf("%u %u %.6g %.6g %.6g %.6g\n", var, j, g_rep, g_obs, err,
               den > 0.0 ? fabs(err)/den : 0.0);
    }
    free(x0);
    freeOperatingPoint(&pt0, 1);
    freeOperatingPoint(&pt1, 1);
}

// LAPACK's dgesdd_ is referenced by poseutils.c:1440 (procrustes fits), which
// is not on the hot path. LAPACK is not installed; abort if anyone gets here
int dgesdd_(void)
{
    SAY("dgesdd_() is not available in the oracle build");
    abort();
}
