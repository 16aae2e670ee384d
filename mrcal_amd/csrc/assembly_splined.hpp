// Launch parameters of the splined models' assembly kernels (assembly_splined.hip), for their launcher (assembly.hip)
#pragma once
#include "solver_device.hpp"

namespace mrcal_amd {

#define SPL_LDS_DOUBLES 8192

// can an observation of this problem take the fallback above? (The whole grid under one board is the largest box)
inline bool spl_fallback_possible(const DeviceProblem& P)
{
    const int order = P.cfg.spline_order, T = SPL_SUB_MAX - order;
    const int nsx = std::max(1, ((int)P.cfg.spline_Nx - order + T - 1)/T), nsy = std::max(1, ((int)P.cfg.spline_Ny - order + T - 1)/T);
    return nsx*nsy > SPL_MAXSUB || P.W*P.H > 1024;
}

#define SPLG_WAVES 8

// rows of the camera block that are knots (a workgroup each), and the others + the x row (SPLG_E workgroups each)
__host__ __device__ inline int splg_nknotrows(const DeviceProblem& P) { return P.Nintr_state > 0 ? P.Ncameras_intrinsics*(P.Nintr_state - P.Ncore_state) : 0; }
__host__ __device__ inline int splg_ndense(const DeviceProblem& P, const NormalDims& nd) { return nd.Nc + 1 - splg_nknotrows(P); }

#define SPLK_WAVES    4

#define SPLC_T 256

// ints of LDS spl_compact_body() needs
__host__ __device__ inline size_t spl_compact_lds_ints(int nknots_all, int Nx) { return (size_t)nknots_all + SPLC_T/64 + 8 + ((Nx + 1) & ~1) + 2*(SPLC_T/64) + 8; }

} // namespace mrcal_amd
