/*
 * elmk.h - C ABI of libelmk: MI355X-native (gfx950 / HIP) per-gridcell land-surface physics.
 *
 * Drop-in boundary for the reference's L3 dispatch layer.  The reference exposes, per physics group, a
 * C++ free function  void ELM::kokkos_<physics>(ELMStateType& S [, const double& dt])  declared in
 * driver/kokkos/<physics>_kokkos.hh and called only from ELMInterface::advance
 * (driver/kokkos/elm_kokkos_interface.cc:287-319).  Each elmk_<physics>() below replaces one of them;
 * the ELMState object becomes an opaque context that owns the same named per-column arrays on the GPU.
 *
 *   reference (file:line)                                             replacement
 *   ---------------------------------------------------------------   ---------------------------------
 *   ELMState(ncols, ...)            src/data/elm_state.h:184-192      elmk_create
 *   ~ELMState                                                         elmk_destroy
 *   S.<field> Views                 src/data/elm_state.h:53-180       elmk_upload / elmk_download /
 *                                                                     elmk_device_ptr  (ids: elmk_fields.def)
 *   S.Land                          src/data/land_data.h:36-44        elmk_set_land
 *   S.dewmx/oldfflag/dayl/max_dayl  src/data/elm_state.h:221-224      elmk_set_scalars
 *   S.pft_data (PFTData)            src/data/pft_data.h:20-31,35-90   elmk_set_pft
 *   S.albsat / S.albdry             src/data/elm_state.h:82           elmk_set_soilcolor
 *   S.snicar_data (SnicarData)      src/data/snicar_data.h:29-71      elmk_set_snicar
 *   kokkos_frac_wet(S)              canopy_hydrology_kokkos.hh:10     elmk_frac_wet
 *   kokkos_albedo_snicar(S)         albedo_kokkos.hh                  elmk_albedo_snicar
 *   kokkos_canopy_hydrology(S,dt)   canopy_hydrology_kokkos.hh:7      elmk_canopy_hydrology
 *   kokkos_surface_radiation(S)     surface_radiation_kokkos.hh       elmk_surface_radiation
 *   kokkos_canopy_temperature(S)    canopy_temperature_kokkos.hh      elmk_canopy_temperature
 *   kokkos_bareground_fluxes(S)     bareground_fluxes_kokkos.hh       elmk_bareground_fluxes
 *   kokkos_canopy_fluxes(S,dt)      canopy_fluxes_kokkos.hh           elmk_canopy_fluxes
 *   advance(): the 7 calls in order elm_kokkos_interface.cc:289-307   elmk_timestep7, elmk_timestep7_fused
 *   advance(): all per-column calls elm_kokkos_interface.cc:289-316   elmk_advance_physics
 *   get_forcing(S, dt, date)        atm_forcing_kokkos.cc:47-75       elmk_get_forcing
 *   update_phenology: ComputePhenology  phenology_kokkos.cc:59-62     elmk_phenology
 *   kokkos_init_timestep's kernel   init_timestep_kokkos.cc:55-75     elmk_init_timestep
 *   initialize_kokkos_elm's lambda  initialize_elm_kokkos.cc:373-428  elmk_initialize_state
 *   kokkos_soil_temperature(S,dt)   soil_temperature_kokkos.hh        elmk_soil_temperature
 *   kokkos_snow_hydrology(S,dt,t)   snow_hydrology_kokkos.hh          elmk_snow_hydrology
 *   S.snw_rds_table (SnwRdsTable)   src/data/snicar_data.h:75-84      elmk_set_snow_age_tables
 *   kokkos_surface_fluxes(S,dt)     surface_fluxes_kokkos.hh          elmk_surface_fluxes
 *   kokkos_evaluate_conservation    conserved_quantity_kokkos.hh      elmk_evaluate_conservation
 *   throw / assert inside physics   (list: SURVEY.md section 5)       per-column flag word, elmk_error_summary
 *
 * Conventions
 *   - every function returns 0 on success, a negative elmk_status on API misuse / HIP failure
 *     (elmk_last_error gives the text); physics never fails a call - reference throw/assert sites raise
 *     bits in a per-column flag word instead (ELMK_ERR_*), readable with elmk_error_summary or by
 *     downloading ELMK_FIELD_err_flags.
 *   - plain pointers and sizes only; host buffers are caller-owned and touched only inside
 *     upload/download/set_*; the context owns all device memory (allocated once in elmk_create).
 *   - kernel entry points enqueue on the context's HIP stream and return; elmk_sync / elmk_download wait.
 *   - one context per GPU; different contexts may be driven from different host threads.
 *   - there is no CPU fallback: without a usable HIP device elmk_create fails with ELMK_E_NO_DEVICE.
 */
#ifndef ELMK_H
#define ELMK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct elmk_ctx elmk_ctx;

/* dimensions (src/data/elm_constants.h:84-98) */
enum {
  ELMK_NLEVSNO = 5,
  ELMK_NLEVGRND = 15,
  ELMK_NLEVTOT = 20,
  ELMK_NUMRAD = 2,
  ELMK_NLEVCAN = 1,
  ELMK_NUMRAD_SNW = 5,
  ELMK_SNO_NBR_AER = 8,
  ELMK_MXPFT = 25,
  ELMK_NSOILCOL = 20,
  ELMK_MIE_N = 1471,
  ELMK_PSN_NPARAM = 27, /* members of ELM::PFTDataPSN, in declaration order (pft_data.h:20-24) */
  ELMK_ALB_NPARAM = 9   /* rhol[2] rhos[2] taul[2] taus[2] xl (pft_data.h:27-31) */
};

typedef enum {
  ELMK_OK = 0,
  ELMK_E_INVALID = -1,   /* bad argument (null pointer, unknown field, range) */
  ELMK_E_NO_DEVICE = -2, /* no HIP device / device id out of range */
  ELMK_E_HIP = -3,       /* a HIP runtime call failed */
  ELMK_E_NOMEM = -4
} elmk_status;

/* element types of fields */
typedef enum { ELMK_F64 = 0, ELMK_I32 = 1, ELMK_U8 = 2, ELMK_U32 = 3 } elmk_dtype;

/* host-side layouts accepted by upload/download */
typedef enum {
  ELMK_LAYOUT_COL_MAJOR = 0, /* reference layout: [column][level], level fastest (ELM::Array / LayoutRight) */
  ELMK_LAYOUT_SOA = 1        /* [level][column], column fastest, dense (stride = ncols of the transfer) */
} elmk_layout;

/* field ids: ELMK_FIELD_<name>, generated from elmk_fields.def; err_flags is appended */
typedef enum {
#define ELMK_FIELD(name, T, nlev) ELMK_FIELD_##name,
#include "elmk_fields.def"
#undef ELMK_FIELD
  ELMK_FIELD_err_flags,
  ELMK_NUM_FIELDS
} elmk_field;

/* per-column error flags, one bit per reference throw / assert site */
enum {
  ELMK_ERR_SURFRAD_LAYER_SUM = 1u << 0, /* surface_radiation_impl.hh:173 assert */
  ELMK_ERR_CANFLX_FORC_HGT = 1u << 1,   /* canopy_fluxes_impl.hh:178 assert(zldis >= 0) */
  ELMK_ERR_PSN_NEG_GS = 1u << 2,        /* photosynthesis_impl.hh:232 */
  ELMK_ERR_PSN_QUADRATIC = 1u << 3,     /* photosynthesis_impl.hh:289 */
  ELMK_ERR_PSN_BRENT_BRACKET = 1u << 4, /* photosynthesis_impl.hh:439 */
  ELMK_ERR_ALB_CANOPY_LAYERS = 1u << 5, /* surface_albedo_impl.hh:270 */
  ELMK_ERR_SNICAR_RDS = 1u << 6,        /* snow_snicar_impl.hh:76 */
  ELMK_ERR_SNICAR_FLAG = 1u << 7,       /* snow_snicar_impl.hh:99 */
  ELMK_ERR_SNICAR_NEG_ABS = 1u << 8,    /* snow_snicar_impl.hh:618 */
  ELMK_ERR_SNICAR_ENERGY = 1u << 9,     /* snow_snicar_impl.hh:658 */
  ELMK_ERR_SNICAR_ALBEDO = 1u << 10,    /* snow_snicar_impl.hh:664 */
  ELMK_WARN_PSN_BALL_BERRY = 1u << 11,  /* photosynthesis_impl.hh:240 (std::cout warning, not fatal) */
  /* elmk_snow_hydrology.  Two places where the reference reads outside an array, so that its own result is undefined; the
   * column continues with the documented choice and the flag records that it was taken (not fatal):
   *   snow_hydrology_impl.hh:388  snow_water reads vol_ice[i+i] (meant i+1) of a 5-element array; i <= 2 is in bounds and is
   *                               reproduced literally, for i = 3 (index 6) vol_ice[i+1] is used;
   *   snow_hydrology_impl.hh:871-885  combine_layers' shift loop copies element top-1 into top; with five snow layers that
   *                               is element -1 of the level arrays: 0.0 is used (the element ends up above the pack and
   *                               is reset by prune_snow_layers / the aerosol update / snow_aging).
   * and the path's two throw sites (fatal): */
  ELMK_WARN_SNOW_WATER_OOB = 1u << 12,
  ELMK_WARN_SNOW_COMBINE_OOB = 1u << 13,
  ELMK_ERR_SNOW_DIVIDE_RDS = 1u << 14,  /* snow_hydrology_impl.hh:1032, :1109, :1187, :1253 radius outside the Mie table */
  ELMK_ERR_SNOW_AGE_DRFRESH = 1u << 15  /* snow_hydrology_impl.hh:152 dr_fresh < 0 */
};
#define ELMK_ERR_FATAL_MASK 0xC7FFu

/* SNICAR lookup tables, names and extents of ELM::SnicarData (src/data/snicar_data.h:39-69); all row-major */
typedef struct {
  const double *ss_alb_oc1, *asm_prm_oc1, *ext_cff_mss_oc1;             /* [5] */
  const double *ss_alb_oc2, *asm_prm_oc2, *ext_cff_mss_oc2;             /* [5] */
  const double *ss_alb_dst1, *asm_prm_dst1, *ext_cff_mss_dst1;          /* [5] */
  const double *ss_alb_dst2, *asm_prm_dst2, *ext_cff_mss_dst2;          /* [5] */
  const double *ss_alb_dst3, *asm_prm_dst3, *ext_cff_mss_dst3;          /* [5] */
  const double *ss_alb_dst4, *asm_prm_dst4, *ext_cff_mss_dst4;          /* [5] */
  const double *ss_alb_snw_drc, *asm_prm_snw_drc, *ext_cff_mss_snw_drc; /* [5][1471] */
  const double *ss_alb_snw_dfs, *asm_prm_snw_dfs, *ext_cff_mss_snw_dfs; /* [5][1471] */
  const double *ss_alb_bc1, *asm_prm_bc1, *ext_cff_mss_bc1;             /* [10][5] */
  const double *ss_alb_bc2, *asm_prm_bc2, *ext_cff_mss_bc2;             /* [10][5] */
  const double *bcenh;                                                  /* [8][10][5] */
} elmk_snicar_tables;

/* one perturbation rule of elmk_tile_columns */
typedef struct {
  int32_t field;  /* elmk_field */
  int32_t mode;   /* 0: v *= (1 + amp*u), 1: v += amp*u,  u ~ U(-1,1) from a counter-based hash */
  double amp;
} elmk_perturb;

/* ---- lifetime ------------------------------------------------------------------------------- */
int elmk_create(int64_t ncols, int device_id, elmk_ctx **out);
int elmk_destroy(elmk_ctx *ctx);
const char *elmk_last_error(const elmk_ctx *ctx); /* ctx may be NULL: error of the last failed create */
int elmk_set_stream(elmk_ctx *ctx, void *hip_stream); /* hipStream_t; NULL restores the context's own stream */
int elmk_sync(elmk_ctx *ctx);
/* on != 0: elmk_timestep7 captures its ~22 kernel launches (and the side-stream fork / join inside albedo_snicar) once as
 * a HIP graph and replays it on every call with the same dt and stream.  Same kernels, same order, same results; what
 * it removes is per-launch host latency, which is all there is to a step of a few thousand columns. */
int elmk_set_graph(elmk_ctx *ctx, int on);
/* Launch options of a context (no effect on results, bit for bit).
 *   ELMK_OPT_CF_HALF_WORKGROUPS  value != 0: the leaf-temperature iteration of kokkos_canopy_fluxes (k_cf_iterate) runs in 256-thread
 *     workgroups, one per compute unit - half the registers and 68 KB of a CU's LDS stay free, so the kernels of ANOTHER context's
 *     stream are resident on the same CUs and use the memory pipeline this fp64-bound kernel leaves idle.  For a driver that steps two
 *     (or more) blocks of columns as separate contexts on separate streams; on its
 *     own the kernel is 1.5 x slower in this shape, which is why it is an option. */
enum { ELMK_OPT_CF_HALF_WORKGROUPS = 1 };
int elmk_set_option(elmk_ctx *ctx, int option, int value);
int64_t elmk_ncols(const elmk_ctx *ctx);
int64_t elmk_level_stride(const elmk_ctx *ctx); /* elements between consecutive levels of a device field */
int64_t elmk_device_bytes(const elmk_ctx *ctx);

/* ---- schema --------------------------------------------------------------------------------- */
int elmk_num_fields(void);
const char *elmk_field_name(int field);
int elmk_field_id(const char *name); /* -1 if unknown */
int elmk_field_info(int field, int *nlev, int *dtype);
/* bytes the device keeps per element of an fp64 field: 8 in the product (libelmk.so); 4 in libelmk_f32.so, the report-only
 * build of BASELINE config 5 ("fp32 state": every fp64 field stored as fp32, all arithmetic fp64, widen on load / round on
 * store).  The ABI is the same in both: uploads and downloads speak double. */
int elmk_state_real_bytes(void);

/* ---- data movement -------------------------------------------------------------------------- */
/* columns [col0, col0+n) of one field; host buffer holds n*nlev elements in the given layout */
int elmk_upload(elmk_ctx *ctx, int field, const void *host, int64_t col0, int64_t n, int layout);
int elmk_download(elmk_ctx *ctx, int field, void *host, int64_t col0, int64_t n, int layout);
int elmk_fill(elmk_ctx *ctx, int field, double value); /* every level of every column */
void *elmk_device_ptr(elmk_ctx *ctx, int field);      /* SoA base: element (lev, col) at [lev*stride + col] */
/* replicate columns [0, nbase) into [nbase, ncols) - column c takes column c % nbase - applying the given
 * perturbations (synthetic-workload generator for the benchmark) */
int elmk_tile_columns(elmk_ctx *ctx, int64_t nbase, uint64_t seed, int nrules, const elmk_perturb *rules);
/* Keep a device-side copy of the listed fields as they are now (replaces any earlier snapshot), and copy
 * them back later (a streaming copy kernel on the context's stream).  A driver uses this for
 * what the rest of the model would do between two calls of the hot path - e.g. the reference resets the
 * forcing heights every step (atm_physics_impl.hh:197-203) and other components move t_veg; the benchmark
 * uses it so that every timed step starts from the same, unconverged canopy state. */
int elmk_snapshot_fields(elmk_ctx *ctx, const int *fields, int nfields);
int elmk_restore_fields(elmk_ctx *ctx);

/* ---- parameters ----------------------------------------------------------------------------- */
int elmk_set_land(elmk_ctx *ctx, int ltype, int ctype, int vtype, int urbpoi, int lakpoi);
int elmk_set_scalars(elmk_ctx *ctx, double dewmx, int oldfflag, double dayl, double max_dayl);
/* psn[25][27], alb[25][9], z0mr[25], displar[25] */
int elmk_set_pft(elmk_ctx *ctx, const double *psn, const double *alb, const double *z0mr, const double *displar);
int elmk_set_soilcolor(elmk_ctx *ctx, const double *albsat /*[20][2]*/, const double *albdry /*[20][2]*/);
int elmk_set_snicar(elmk_ctx *ctx, const elmk_snicar_tables *t);
/* SnwRdsTable (src/data/snicar_data.h:75-84): the snow-aging best-fit parameters snowage_tau [hour], snowage_kappa and
 * snowage_drdt0 [um/hour], each [11][31][8] = [temperature][temperature gradient][density] index, row-major */
int elmk_set_snow_age_tables(elmk_ctx *ctx, const double *tau, const double *kappa, const double *drdt0);

/* ---- the physics wrappers (same names, order and arguments as driver/kokkos) ---------------- */
int elmk_frac_wet(elmk_ctx *ctx);
int elmk_albedo_snicar(elmk_ctx *ctx);
int elmk_canopy_hydrology(elmk_ctx *ctx, double dt);
int elmk_surface_radiation(elmk_ctx *ctx);
int elmk_canopy_temperature(elmk_ctx *ctx);
int elmk_bareground_fluxes(elmk_ctx *ctx);
int elmk_canopy_fluxes(elmk_ctx *ctx, double dt);
/* L2-level forms of the two flux wrappers: the forcing-derived scalars air density / O2 / CO2 partial pressure (per column,
 * host arrays of ncols doubles; NULL = derive it as the wrapper does, canopy_fluxes_kokkos.cc:47-49 /
 * bareground_fluxes_kokkos.cc:31) handed in, which is how the reference's unit tests drive the physics with the values ELM
 * itself used (test/test_CanFlux.cc, test/test_BGFlux.cc).  Everything else as elmk_canopy_fluxes / elmk_bareground_fluxes. */
int elmk_canopy_fluxes_given(elmk_ctx *ctx, double dt, const double *forc_rho, const double *forc_po2, const double *forc_pco2);
int elmk_bareground_fluxes_given(elmk_ctx *ctx, const double *forc_rho);
int elmk_timestep7(elmk_ctx *ctx, double dt);
/* The same seven calls (elm_kokkos_interface.cc:289-307) with the five streaming wrappers between albedo and the
 * leaf-temperature iteration - canopy_hydrology, surface_radiation, canopy_temperature, the streaming stage of
 * bareground_fluxes and canopy_fluxes' initialize_flux - fused into ONE pass per column: every state element the step
 * touches is read once and written once (3 405 B per column-step instead of 5 281; BASELINE.json
 * config 5's launch structure, fp64 state).  Results are bit-identical to elmk_timestep7 in every field.  elmk_set_graph
 * applies to it as well. */
int elmk_timestep7_fused(elmk_ctx *ctx, double dt);
/* next in ELMInterface::advance (elm_kokkos_interface.cc:310): kokkos_soil_temperature(S, dt),
 * soil_temperature_kokkos.cc:6-278 - thermal properties, the 21-row pentadiagonal temperature system of
 * snow / standing surface water / soil, its solve, phase change, ground temperature */
int elmk_soil_temperature(elmk_ctx *ctx, double dt);
/* kokkos_snow_hydrology(S, dt, time_plus_half_dt) (snow_hydrology_kokkos.cc:23-188; elm_kokkos_interface.cc:313, between
 * soil_temperature and surface_fluxes; the date argument is unused by the reference's wrapper): snow_water,
 * compute_aerosol_deposition, aerosol_phase_change, transpiration, snow_compaction, combine_layers, divide_layers,
 * prune_snow_layers, update_aerosol_mass_and_concen, snow_aging - five launches in the reference, one pass per column here.
 * Updates snl and the snow mesh (dz, zsoi, zisoi, t_soisno, h2osoi_ice/liq of the snow levels and of the top soil level),
 * snw_rds, the aerosol masses mss_* and concentrations cnc_*, h2osno, snow_depth, frac_sno(_eff), int_snow, qflx_snow_melt,
 * qflx_top_soil, qflx_sl_top_soil, qflx_snow2topsoi, mflx_*, qflx_rootsoi.  The reference has no fixture for this path; the
 * checker behind the parity tests is pinned bit for bit by the reference's own snow_hydrology.h for every function
 * but the two aerosol bookkeeping functions (those: parity unpinned).  See ELMK_WARN_SNOW_* for the two
 * places where the reference's own result is undefined. */
int elmk_snow_hydrology(elmk_ctx *ctx, double dt);
/* kokkos_surface_fluxes(S, dt) (surface_fluxes_kokkos.cc:5-107): flux corrections for the new ground temperature,
 * ground heat flux, total fluxes, dew / sublimation partition, outgoing longwave, soil energy balance */
int elmk_surface_fluxes(elmk_ctx *ctx, double dt);
/* the per-column kernel at the end of kokkos_init_timestep (init_timestep_kokkos.cc:55-75): h2osno_old,
 * dtbegin_column_h2o (what the conservation check starts from), snow capping flag, frac_veg_nosno, frac_iceold.
 * (The forcing / phenology readers before it in that wrapper are I/O and stay with the caller.) */
int elmk_init_timestep(elmk_ctx *ctx);
/* The "init functions" lambda ELM::initialize_kokkos_elm runs once per column after the input files are read
 * (driver/kokkos/initialize_elm_kokkos.cc:373-428): init_topo_slope / init_melt_factor / init_micro_sigma, init_snow_layers,
 * init_soil_hydraulics, init_vegrootfr, init_soil_temp, init_snow_state, init_soilh2o_state - the producer of the state the
 * physics calls consume.  Inputs: the fields topo_slope, topo_std, snow_depth, vtype, zsoi / zisoi / dz of the soil levels
 * and the surface-data soil texture pct_sand, pct_clay, organic (by soil level; wrapper-local Views in the reference,
 * :309-311), plus elmk_set_init_params: organic_max of the parameter file (:312) and PFTData::roota_par / rootb_par [25]
 * (pft_data.h:72-73).  Writes topo_slope, n_melt, micro_sigma, snl, dz / zsoi / zisoi of the snow levels, watsat, bsw, sucsat,
 * watdry, watopt, watfc, tkmg, tkdry, csol, rootfr, t_soisno, t_grnd, h2osno, int_snow, snow_depth, h2osfc, h2ocan,
 * frac_h2osfc, fwet, fdry, frac_sno, snw_rds, h2osoi_vol, h2osoi_liq, h2osoi_ice. */
int elmk_set_init_params(elmk_ctx *ctx, double organic_max, const double *roota_par, const double *rootb_par);
int elmk_initialize_state(elmk_ctx *ctx);
/* get_forcing (driver/kokkos/atm_forcing_kokkos.cc:47-75), called by kokkos_init_timestep (init_timestep_kokkos.cc:47):
 * the eight ComputeAtmForcing_* functors of src/physics/atm_physics_impl.hh:27-203 - TBOT, PBOT, QBOT|RH, FLDS, FSDS,
 * PREC, WIND, ZBOT - over the fields atm_tbot .. atm_wind (level 0 = forcing record t_idx, level 1 = t_idx + 1 of
 * AtmDataManager::data; upload `data + t_idx * ncells` with ELMK_LAYOUT_SOA) and coszen.  Writes forc_tbot, forc_thbot,
 * forc_pbot, forc_qbot, forc_lwrad, forc_solad, forc_solai, forc_rain, forc_snow, forc_u, forc_v, forc_hgt,
 * forc_hgt_{u,t,q}_patch.  wt1, wt2: [8] host doubles in that stream order = AtmDataManager::forcing_time_weights
 * (src/data/atm_data_impl.hh:191-199) of each stream (ignored for FSDS, PREC, ZBOT); qbot_is_rh != 0: the humidity
 * stream holds relative humidity in per cent (AtmForcType::RH).  Picking t_idx and the weights is date arithmetic on
 * the host (forc_t_idx_check_bounds, atm_data_impl.hh:147-169) and stays with the caller, as do the file readers. */
int elmk_get_forcing(elmk_ctx *ctx, const double *wt1, const double *wt2, int qbot_is_rh);
/* ComputePhenology (src/physics/phenology_physics_impl.hh:22-69), run by update_phenology (phenology_kokkos.cc:59-62,
 * called at init_timestep_kokkos.cc:43) over the fields mlai, msai, mhtop, mhbot (level 0 = month start_idx, level 1 =
 * start_idx + 1 of PhenologyDataManager).  Writes tlai, tsai, htop, hbot, elai, esai, frac_veg_nosno_alb. */
int elmk_phenology(elmk_ctx *ctx, double wt1, double wt2);
/* kokkos_evaluate_conservation(S, dt) (conserved_quantity_kokkos.cc:8-81).  The reference keeps its eight
 * diagnostics in wrapper-local Views and prints column 0; here min_max_sum[8][3] receives (min, max, sum) over the
 * context's columns of dtend_column_h2o, errh2o, errh2osno, dwb, errsol, errlon, errseb, netrad - what a multi-GPU
 * run all-reduces with MIN / MAX / SUM (the reference's min_max_sum, src/utils/min_max_sum.hh:57-66) - and
 * per_column (may be NULL) the values themselves, [8][ncols].  Synchronises. */
int elmk_evaluate_conservation(elmk_ctx *ctx, double dt, double *min_max_sum, double *per_column);
/* Everything ELMInterface::advance calls per column after kokkos_init_timestep, in its order
 * (elm_kokkos_interface.cc:289-316): the seven wrappers (as elmk_timestep7_fused), kokkos_soil_temperature,
 * kokkos_snow_hydrology, kokkos_surface_fluxes - one call, and with elmk_set_graph one HIP graph launch per model step.
 * Same bits as the ten calls.  elmk_set_snow_age_tables must have been called.  (kokkos_evaluate_conservation returns
 * values to the host and stays a call of its own.) */
int elmk_advance_physics(elmk_ctx *ctx, double dt);

/* ---- diagnostics ---------------------------------------------------------------------------- */
/* OR of all columns' flag words and the first column with a fatal bit (-1 if none); synchronises */
int elmk_error_summary(elmk_ctx *ctx, uint32_t *or_of_flags, int64_t *first_bad_col);
int elmk_clear_errors(elmk_ctx *ctx);
/* run `nsteps` timesteps with HIP events between the seven launches on the context's stream;
 * ms_per_kernel[7] (frac_wet, albedo_snicar, canopy_hydrology, surface_radiation, canopy_temperature,
 * bareground_fluxes, canopy_fluxes) receives the mean device time of each launch, *ms_total the mean
 * time of one whole timestep (first event to last).  If a snapshot exists (elmk_snapshot_fields) it is
 * restored before every step, outside the event brackets, so each profiled step does the same work. */
int elmk_profile_timestep7(elmk_ctx *ctx, double dt, int nsteps, float *ms_per_kernel, float *ms_total);
/* the same for elmk_timestep7_fused: ms_per_stage[5] = frac_wet + list resets + queue class count (k_fz_prep);
 * albedo_snicar; the fused streaming pass (k_fz_stream); the bare-ground flux list; the leaf-temperature iteration +
 * compute_flux (k_cf_iterate, k_cf_finish) */
int elmk_profile_timestep7_fused(elmk_ctx *ctx, double dt, int nsteps, float *ms_per_stage, float *ms_total);
/* the same for one wrapper: mean device time over nsteps launches, HIP events on the context's stream, the snapshot (if
 * any) restored before every launch outside the event brackets */
typedef enum {
  ELMK_WRAPPER_FRAC_WET = 0, ELMK_WRAPPER_ALBEDO_SNICAR, ELMK_WRAPPER_CANOPY_HYDROLOGY, ELMK_WRAPPER_SURFACE_RADIATION,
  ELMK_WRAPPER_CANOPY_TEMPERATURE, ELMK_WRAPPER_BAREGROUND_FLUXES, ELMK_WRAPPER_CANOPY_FLUXES,
  ELMK_WRAPPER_SOIL_TEMPERATURE, ELMK_WRAPPER_SURFACE_FLUXES, ELMK_WRAPPER_SNOW_HYDROLOGY,
  ELMK_WRAPPER_ADVANCE_PHYSICS /* elmk_advance_physics: all ten in the reference's order */
} elmk_wrapper;
int elmk_profile_wrapper(elmk_ctx *ctx, int wrapper, double dt, int nsteps, float *ms_mean);
/* device time of each of nsteps steps of elmk_timestep7 (fused = 0) or elmk_timestep7_fused (fused != 0), by HIP events
 * around every step on the context's stream, the snapshot restored before every step outside the brackets: what the
 * benchmark takes its median step time from */
int elmk_profile_steps(elmk_ctx *ctx, int fused, double dt, int nsteps, float *ms_each_step);
/* Read back context-owned scratch (diagnostics; not part of the state contract).
 *   ELMK_SCRATCH_CF_TRIPS: int32 per column - trips of the leaf-temperature iteration
 *                          (canopy_fluxes_impl.hh:233-450) in the last elmk_canopy_fluxes call, 0 = not vegetated
 *   ELMK_SCRATCH_WORK:     doubles of the work arrays, raw (development probes)
 * Synchronises the stream. */
enum {
  ELMK_SCRATCH_CF_TRIPS = 0,
  ELMK_SCRATCH_WORK = 1,
  ELMK_SCRATCH_CF_HINTS = 2,   /* int32 per column: the scheduling hint (decaying maximum of the trip count) */
  ELMK_SCRATCH_LIST_COUNTS = 3 /* uint32 x 2 per internal work list: entries, queue head.  Every list is left empty by the
                                  wrapper that filled it, so both read 0 between two calls (a test asserts it) */
};
int elmk_read_scratch(elmk_ctx *ctx, int kind, void *host, int64_t offset, int64_t count);
/* device-to-device copy bandwidth probe (read+write bytes / s) on this context's device, used as the
 * empirical HBM line next to the 8 TB/s datasheet peak */
int elmk_copy_bandwidth(elmk_ctx *ctx, int64_t bytes, int iters, double *gbytes_per_s);
/* the same probe in a chosen access shape: 0 = 8 bytes per lane, one load per thread (the shape of elmk_copy_bandwidth and of
 * the snapshot restore); 1 = 16 bytes per lane; 2 = 8 bytes per lane, four independent loads per thread; 3 = 16 bytes per
 * lane, four independent loads per thread; 4 = 8 bytes per lane, 64 separate streams read and 64 written by every thread
 * (the buffer seen as 64 fields, field-major like the state).  bench.py reports the best of 0..3 as roofline.empirical_peak
 * and shape 4 as the line of a many-field streaming kernel. */
int elmk_copy_bandwidth_shape(elmk_ctx *ctx, int64_t bytes, int iters, int shape, double *gbytes_per_s);
/* Evaluate one function of elmkernels_amd/csrc/elmk_math.h - the device restatement of the host libm's exp / log / log10 /
 * pow / atan / tanh / cos / erf / acos / expm1 (the <cmath> calls of src/physics headers) - on n host values: out[i] = fn(x[i]) or pow(x[i], y[i]).
 * ELMK_MATH_SQRT and ELMK_MATH_DIV (x[i] / y[i]) are the device's own IEEE operations, included so that the tests can
 * confirm they round correctly.  y is read only for ELMK_MATH_POW / ELMK_MATH_DIV.  Used by the parity tests to compare the device bits with the host libm's. */
typedef enum {
  ELMK_MATH_EXP = 0, ELMK_MATH_LOG, ELMK_MATH_LOG10, ELMK_MATH_ATAN, ELMK_MATH_SQRT, ELMK_MATH_TANH, ELMK_MATH_COS, ELMK_MATH_ERF,
  ELMK_MATH_ACOS, ELMK_MATH_EXPM1, ELMK_MATH_DIV, ELMK_MATH_POW
} elmk_math_fn;
int elmk_math_eval(elmk_ctx *ctx, int fn, const double *x, const double *y, double *out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* ELMK_H */
