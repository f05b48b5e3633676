// elmk_kernels.h - host-callable launchers of the HIP kernels (defined in the k_*.hip files)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "elmk.h"

namespace elmk {

struct DevState;

// side streams of a context: independent launches of one wrapper run beside each other (fork from / join to the
// caller's stream with events, so the wrapper still behaves as one in-order operation on that stream)
constexpr int ELMK_NSIDE = 5;
struct SideStreams {
  hipStream_t s[ELMK_NSIDE];
  hipEvent_t fork;
  hipEvent_t join[ELMK_NSIDE];
  // launch shape of the leaf-temperature iteration (elmk_set_option ELMK_OPT_CF_HALF_WORKGROUPS): > 0 = k_cf_iterate_half with at most
  // this many 256-thread workgroups (the device's CU count: one per CU), 0 = the product's k_cf_iterate
  int cf_half_groups;
};

// the seven physics launches (one per reference L3 wrapper)
void launch_frac_wet(const DevState* S, int64_t n, hipStream_t st);
// classify = false: the caller has run stage 1 (soil albedo, SNICAR queues) itself (the fused step does it in k_fz_prep)
// final = false: the caller runs stage 3 (k_alb_final's body) itself (the fused step's k_fz_stream)
void launch_albedo_snicar(const DevState* S, int64_t n, hipStream_t st, const SideStreams* side, bool classify = true, bool final = true);
void launch_albedo_snicar_part(const DevState* S, int64_t n, hipStream_t st, int part, unsigned* snicar_grid);
void launch_canopy_hydrology(const DevState* S, int64_t n, double dt, hipStream_t st);
void launch_surface_radiation(const DevState* S, int64_t n, hipStream_t st);
void launch_canopy_temperature(const DevState* S, int64_t n, hipStream_t st);
// given (bit 0 forc_rho, bit 1 forc_po2, bit 2 forc_pco2): the L2-level entries elmk_*_given take these from DevState::cf_given
void launch_bareground_fluxes(const DevState* S, int64_t n, hipStream_t st, int given = 0);
void launch_canopy_fluxes(const DevState* S, int64_t n, double dt, hipStream_t st, int given = 0, const SideStreams* side = nullptr);
// elmk_timestep7_fused: the same seven wrappers as five launch groups (k_canopy_fluxes.hip):
//   0 k_fz_prep (frac_wet, list resets, canopy_fluxes class count)   1 albedo_snicar   2 k_fz_stream (canopy_hydrology ->
//   surface_radiation -> canopy_temperature -> bare-ground list -> canopy_fluxes initialize_flux, one pass per column)
//   3 the bare-ground flux list   4 the leaf-temperature iteration and compute_flux
constexpr int ELMK_FUSED_NSTAGE = 5;
void launch_fused_stage(const DevState* S, int64_t n, double dt, hipStream_t st, const SideStreams* side, int stage);
void launch_bareground_list(const DevState* S, int64_t n, hipStream_t st);
// next row after the seven: soil / snow column temperature
void launch_soil_temperature(const DevState* S, int64_t n, double dt, hipStream_t st);
// snow hydrology, aerosol masses, transpiration sink (k_snow_hydrology.hip)
void launch_snow_hydrology(const DevState* S, int64_t n, double dt, hipStream_t st);
// surface fluxes after the solve, conservation diagnostics reduced to (min, max, sum)
constexpr int ELMK_CONS_NPART = 512;  // stage-1 partials per diagnostic
void launch_surface_fluxes(const DevState* S, int64_t n, double dt, hipStream_t st);
void launch_init_timestep(const DevState* S, int64_t n, hipStream_t st);
// the per-column init functions of ELM::initialize_kokkos_elm (initialize_elm_kokkos.cc:373-428), k_init_state.hip
void launch_initialize_state(const DevState* S, int64_t n, hipStream_t st);
// the forcing and phenology functors kokkos_init_timestep runs first (k_forcing.hip)
void launch_get_forcing(const DevState* S, int64_t n, const double* wt1, const double* wt2, int qbot_is_rh, hipStream_t st);
void launch_phenology(const DevState* S, int64_t n, double wt1, double wt2, hipStream_t st);
void launch_conservation(const DevState* S, int64_t n, int64_t ld, double dt, const double* diag, double* part, double* out,
                         hipStream_t st);

// layout conversion between the reference's [column][level] host layout and device SoA [level][column]
// staging: dense buffer of n*nlev elements in device memory; elem = element size in bytes (1, 4 or 8)
void launch_cols_to_soa(const void* staging, void* field, int elem, int nlev, int64_t ld, int64_t col0, int64_t n,
                        hipStream_t st);
void launch_soa_to_cols(const void* field, void* staging, int elem, int nlev, int64_t ld, int64_t col0, int64_t n,
                        hipStream_t st);
// storage code of an fp64 state field kept as fp32 (ELMK_STATE_F32 builds), beside the public elmk_dtype values
constexpr int ELMK_F32_STORED = 16;
void launch_fill(void* field, int dtype, int nlev, int64_t ld, int64_t ncols, double value, hipStream_t st);
void launch_tile(void* field, int dtype, int nlev, int64_t ld, int64_t ncols, int64_t nbase, uint64_t seed,
                 int field_id, int mode, double amp, hipStream_t st);
void launch_flag_reduce(const uint32_t* flags, int64_t n, uint32_t* or_out, long long* first_bad, hipStream_t st);
void launch_copy(const double* src, double* dst, int64_t n, hipStream_t st, int shape = 0);
// up to COPY_JOBS_MAX device-to-device copies of 8-byte words in one launch; end[j] = words of jobs 0..j together
constexpr int COPY_JOBS_MAX = 16;
struct CopyJobs {
  const double* src[COPY_JOBS_MAX];
  double* dst[COPY_JOBS_MAX];
  int64_t end[COPY_JOBS_MAX];
  int n;
};
void launch_copy_multi(const CopyJobs& J, hipStream_t st);
void launch_math_eval(int fn, const double* x, const double* y, double* out, int64_t n, hipStream_t st);

}  // namespace elmk
