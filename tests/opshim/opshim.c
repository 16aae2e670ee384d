/* TEST INFRASTRUCTURE: the operator-level plug-in of INTEGRATION.md section 3,
   compiled and run (tests/test_operator_boundary.py).

   libdogleg calls  void f(const double* p, double* x, cholmod_sparse* Jt, void* cookie)
   (dogleg_callback_t; the reference hands it its own optimizer_callback, cast, at
   mrcal.c:6435-6439). gpu_callback() below is a callback of exactly that shape
   over the resident tier of libmrcal_amd.so. Here it is handed to the CHECKER's
   restated dogleg_optimize2() (oracle/dogleg_restated.c) with mrcal's solver
   settings (mrcal.c:6289-6299): the reference's solver loop, the product's
   residuals and Jacobian. */
#include <stdbool.h>
#include <stdint.h>
#include <stddef.h>
#include "dogleg.h"          /* oracle/stubs: the interface mrcal.c compiles against */
#define MRCAL_AMD_HAVE_CHOLMOD_SPARSE     /* dogleg.h brought the type */
#include "mrcal_amd.h"

static int Ncalls, Ncalls_without_J;

/* verbatim from INTEGRATION.md section 3 */
static void gpu_callback(const double* p, double* x, cholmod_sparse* Jt, void* cookie)
{
    mrcal_amd_problem_t* P = (mrcal_amd_problem_t*)cookie;       /* created once */
    mrcal_amd_problem_set_b_packed(P, p);
    mrcal_amd_problem_evaluate(P, Jt != NULL, true);
    mrcal_amd_problem_get_x(P, x);
    if(Jt) mrcal_amd_problem_get_J(P, (int32_t*)Jt->p, (int32_t*)Jt->i, (double*)Jt->x);
    Ncalls++;
    if(Jt == NULL) Ncalls_without_J++;
}

/* the solve of mrcal.c:6289-6299 + 6435-6439 with gpu_callback in the place of optimizer_callback.
   p: the packed state, in/out. Returns |x|^2 at the solution (<0: failure) */
double opshim_optimize(mrcal_amd_problem_t* P, double* p, int Nstate, int Nmeas, int Nnz, int max_iterations)
{
    dogleg_parameters2_t prm;
    dogleg_getDefaultParameters(&prm);
    prm.dogleg_debug          = 0;
    prm.Jt_x_threshold        = 0;
    prm.update_threshold      = 1e-7;
    prm.trustregion_threshold = 0;
    prm.max_iterations        = max_iterations > 0 ? max_iterations : 300;
    Ncalls = Ncalls_without_J = 0;
    return dogleg_optimize2(p, (unsigned)Nstate, (unsigned)Nmeas, (unsigned)Nnz,
                            (dogleg_callback_t*)&gpu_callback, P, &prm, NULL);
}

/* the Jt == NULL form of the callback, as libdogleg may call it */
void opshim_residuals_only(mrcal_amd_problem_t* P, const double* p, double* x)
{
    gpu_callback(p, x, NULL, P);
}
void opshim_counts(int* n, int* n_without_J) { *n = Ncalls; *n_without_J = Ncalls_without_J; }
