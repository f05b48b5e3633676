/*
 * elm_oracle.h - CPU restatement (plain C) of the reference's per-column land-surface physics.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle for the HIP kernels in elmkernels_amd/csrc.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * library (libelmk.so) never links, loads or calls anything in this directory.
 *
 * Every function restates one function of the reference (path:line under /root/reference given at each
 * definition) on ONE column, with the reference's storage convention: level arrays are contiguous
 * per column ([col][lev] row-major, ELM::Array src/utils/array.hh:176-179).  The elmo_<wrapper>()
 * entry points restate the L3 dispatch wrappers driver/kokkos/*_kokkos.cc (call order, argument
 * wiring, zero-filled per-call temporaries) with an OpenMP loop over columns standing in for
 * Kokkos::parallel_for(RangePolicy<OpenMP>) (src/utils/invoke_kernel.hh:24-27).
 *
 * Pinning (see tests/test_oracle_golden.py, tests/test_oracle_vs_ref.py):
 *   - all seven reference fixture pairs test/data/<Module>_{IN,OUT}.txt (committed as tests/golden/*.npz);
 *   - the reference's own headers compiled under oracle/_ref/ (oracle/Makefile, four libraries) and run bit for bit against
 *     the restatement: all seven wrappers of the hot path - canopy_hydrology, surface_radiation, canopy_temperature,
 *     bareground_fluxes (libelmref.so) and, since round 3, surface_albedo + snow_snicar and canopy_fluxes + photosynthesis
 *     (libelmref_canopy.so; pft_data.h reaches netcdf.h only through the file readers, which ref_harness_canopy.cc skips by
 *     read_input.hh's own include guard - no stand-in for netcdf) - plus soil temperature, surface fluxes, init_timestep,
 *     the initialisation functions, and eight of the ten stages of snow hydrology.
 */
#ifndef ELM_ORACLE_H
#define ELM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* dimensions - src/data/elm_constants.h:84-98 */
enum {
  ELMO_NLEVSNO = 5,
  ELMO_NLEVGRND = 15,
  ELMO_NLEVTOT = 20,
  ELMO_NUMRAD = 2,
  ELMO_NLEVCAN = 1,
  ELMO_NUMRAD_SNW = 5,
  ELMO_SNO_NBR_AER = 8,
  ELMO_MXPFT = 25,
  ELMO_NSOILCOL = 20,
  ELMO_MIE_N = 1471,
  ELMO_SNOWAGE_N = 11 * 31 * 8
};

/* per-column error flag bits: one per reference throw / assert site */
enum {
  ELMO_ERR_SURFRAD_LAYER_SUM = 1u << 0,   /* surface_radiation_impl.hh:173 assert */
  ELMO_ERR_CANFLX_FORC_HGT = 1u << 1,     /* canopy_fluxes_impl.hh:178 assert(zldis >= 0) */
  ELMO_ERR_PSN_NEG_GS = 1u << 2,          /* photosynthesis_impl.hh:232 */
  ELMO_ERR_PSN_QUADRATIC = 1u << 3,       /* photosynthesis_impl.hh:289 */
  ELMO_ERR_PSN_BRENT_BRACKET = 1u << 4,   /* photosynthesis_impl.hh:439 */
  ELMO_ERR_ALB_CANOPY_LAYERS = 1u << 5,   /* surface_albedo_impl.hh:270 */
  ELMO_ERR_SNICAR_RDS = 1u << 6,          /* snow_snicar_impl.hh:76 */
  ELMO_ERR_SNICAR_FLAG = 1u << 7,         /* snow_snicar_impl.hh:99 */
  ELMO_ERR_SNICAR_NEG_ABS = 1u << 8,      /* snow_snicar_impl.hh:618 */
  ELMO_ERR_SNICAR_ENERGY = 1u << 9,       /* snow_snicar_impl.hh:658 */
  ELMO_ERR_SNICAR_ALBEDO = 1u << 10,      /* snow_snicar_impl.hh:664 */
  ELMO_WARN_PSN_BALL_BERRY = 1u << 11,    /* photosynthesis_impl.hh:240 (std::cout warning) */
  /* snow hydrology (elmo_physics_g.c): two places where the reference reads outside an array (its result is undefined;
   * the documented choice was taken) and its two throw sites */
  ELMO_WARN_SNOW_WATER_OOB = 1u << 12,    /* snow_hydrology_impl.hh:388 vol_ice[i+i], i = 3 */
  ELMO_WARN_SNOW_COMBINE_OOB = 1u << 13,  /* snow_hydrology_impl.hh:871-885 element -1 */
  ELMO_ERR_SNOW_DIVIDE_RDS = 1u << 14,    /* snow_hydrology_impl.hh:1032, :1109, :1187, :1253 */
  ELMO_ERR_SNOW_AGE_DRFRESH = 1u << 15    /* snow_hydrology_impl.hh:152 */
};

/* src/data/land_data.h:36-44 */
typedef struct {
  int ltype, ctype, vtype;
  int urbpoi, lakpoi;
} elmo_land;

/* src/data/pft_data.h:20-24 (same member order) */
typedef struct {
  double fnr, act25, kcha, koha, cpha, vcmaxha, jmaxha, tpuha, lmrha;
  double vcmaxhd, jmaxhd, tpuhd, lmrhd, lmrse, qe, theta_cj, bbbopt, mbbopt;
  double c3psn, slatop, leafcn, flnr, fnitr, dleaf, smpso, smpsc, tc_stress;
} elmo_pft_psn;

/* src/data/pft_data.h:27-31 */
typedef struct {
  double rhol[2], rhos[2], taul[2], taus[2];
  double xl;
} elmo_pft_alb;

/* src/data/snicar_data.h:29-71 - same names, flat row-major extents as allocated there */
typedef struct {
  double ss_alb_oc1[5], asm_prm_oc1[5], ext_cff_mss_oc1[5];
  double ss_alb_oc2[5], asm_prm_oc2[5], ext_cff_mss_oc2[5];
  double ss_alb_dst1[5], asm_prm_dst1[5], ext_cff_mss_dst1[5];
  double ss_alb_dst2[5], asm_prm_dst2[5], ext_cff_mss_dst2[5];
  double ss_alb_dst3[5], asm_prm_dst3[5], ext_cff_mss_dst3[5];
  double ss_alb_dst4[5], asm_prm_dst4[5], ext_cff_mss_dst4[5];
  double ss_alb_snw_drc[5 * ELMO_MIE_N], asm_prm_snw_drc[5 * ELMO_MIE_N], ext_cff_mss_snw_drc[5 * ELMO_MIE_N];
  double ss_alb_snw_dfs[5 * ELMO_MIE_N], asm_prm_snw_dfs[5 * ELMO_MIE_N], ext_cff_mss_snw_dfs[5 * ELMO_MIE_N];
  double ss_alb_bc1[10 * 5], asm_prm_bc1[10 * 5], ext_cff_mss_bc1[10 * 5];
  double ss_alb_bc2[10 * 5], asm_prm_bc2[10 * 5], ext_cff_mss_bc2[10 * 5];
  double bcenh[8 * 10 * 5];
} elmo_snicar;

/*
 * Column state: the subset of ELMStateViews (src/data/elm_state.h:53-180) + AerosolConcentrations
 * (src/data/aerosol_data.h:43-51) that the seven hot-path wrappers touch.
 * X(name, kind, nlev): kind D = double, I = int, B = unsigned char (bool).
 */
#define ELMO_FIELDS(X)                                                                                      \
  X(forc_tbot, D, 1) X(forc_thbot, D, 1) X(forc_pbot, D, 1) X(forc_qbot, D, 1) X(forc_lwrad, D, 1)           \
  X(forc_u, D, 1) X(forc_v, D, 1) X(forc_hgt_u_patch, D, 1) X(forc_hgt_t_patch, D, 1)                        \
  X(forc_hgt_q_patch, D, 1) X(forc_rain, D, 1) X(forc_snow, D, 1) X(forc_solai, D, 2) X(forc_solad, D, 2)    \
  X(tlai, D, 1) X(tsai, D, 1) X(elai, D, 1) X(esai, D, 1) X(htop, D, 1)                                      \
  X(frac_veg_nosno, I, 1)                                                                                   \
  X(watsat, D, 15) X(sucsat, D, 15) X(bsw, D, 15) X(watdry, D, 15) X(watopt, D, 15) X(watfc, D, 15)          \
  X(n_melt, D, 1) X(micro_sigma, D, 1)                                                                      \
  X(isoicol, I, 1)                                                                                          \
  X(snl, I, 1) X(snow_depth, D, 1) X(frac_sno, D, 1) X(int_snow, D, 1) X(snw_rds, D, 5) X(swe_old, D, 5)     \
  X(frac_iceold, D, 20) X(h2osoi_liq, D, 20) X(h2osoi_ice, D, 20) X(h2osoi_vol, D, 15)                       \
  X(h2ocan, D, 1) X(h2osno, D, 1) X(fwet, D, 1) X(fdry, D, 1) X(h2osfc, D, 1) X(frac_h2osfc, D, 1)           \
  X(frac_sno_eff, D, 1)                                                                                     \
  X(qflx_snwcp_liq, D, 1) X(qflx_snwcp_ice, D, 1) X(qflx_snow_grnd, D, 1) X(qflx_rain_grnd, D, 1)            \
  X(qflx_snow_melt, D, 1)                                                                                   \
  X(t_soisno, D, 20) X(t_grnd, D, 1)                                                                        \
  X(nrad, I, 1) X(laisun, D, 1) X(laisha, D, 1) X(parsun_z, D, 1) X(parsha_z, D, 1) X(laisun_z, D, 1)        \
  X(laisha_z, D, 1)                                                                                         \
  X(sabg_soil, D, 1) X(sabg_snow, D, 1) X(sabg, D, 1) X(sabv, D, 1) X(fsa, D, 1) X(fsr, D, 1)                \
  X(sabg_lyr, D, 6)                                                                                         \
  X(tlai_z, D, 1) X(fsun_z, D, 1) X(fabd_sun_z, D, 1) X(fabd_sha_z, D, 1) X(fabi_sun_z, D, 1)                \
  X(fabi_sha_z, D, 1)                                                                                       \
  X(ftdd, D, 2) X(ftid, D, 2) X(ftii, D, 2) X(fabd, D, 2) X(fabi, D, 2) X(albsod, D, 2) X(albsoi, D, 2)      \
  X(albgrd, D, 2) X(albgri, D, 2) X(flx_absdv, D, 6) X(flx_absdn, D, 6) X(flx_absiv, D, 6)                   \
  X(flx_absin, D, 6) X(albd, D, 2) X(albi, D, 2)                                                            \
  X(t_h2osfc, D, 1) X(t_h2osfc_bef, D, 1) X(soilbeta, D, 1) X(qg_snow, D, 1) X(qg_soil, D, 1) X(qg, D, 1)    \
  X(qg_h2osfc, D, 1) X(dqgdT, D, 1) X(htvp, D, 1) X(emg, D, 1) X(emv, D, 1) X(z0mg, D, 1) X(z0hg, D, 1)      \
  X(z0qg, D, 1) X(z0mv, D, 1) X(z0hv, D, 1) X(z0qv, D, 1) X(thv, D, 1) X(z0m, D, 1) X(displa, D, 1)          \
  X(thm, D, 1) X(eflx_sh_tot, D, 1) X(eflx_lh_tot, D, 1) X(eflx_sh_veg, D, 1) X(qflx_evap_tot, D, 1)         \
  X(qflx_evap_veg, D, 1) X(qflx_tran_veg, D, 1) X(tssbef, D, 20)                                            \
  X(dlrad, D, 1) X(ulrad, D, 1) X(eflx_sh_grnd, D, 1) X(eflx_sh_snow, D, 1) X(eflx_sh_soil, D, 1)            \
  X(eflx_sh_h2osfc, D, 1) X(qflx_evap_soi, D, 1) X(qflx_ev_snow, D, 1) X(qflx_ev_soil, D, 1)                 \
  X(qflx_ev_h2osfc, D, 1) X(t_ref2m, D, 1) X(q_ref2m, D, 1) X(rh_ref2m, D, 1) X(cgrnds, D, 1)                \
  X(cgrndl, D, 1) X(cgrnd, D, 1)                                                                            \
  X(altmax_indx, I, 1) X(altmax_lastyear_indx, I, 1) X(t10, D, 1) X(vcmaxcintsha, D, 1)                      \
  X(vcmaxcintsun, D, 1) X(btran, D, 1) X(t_veg, D, 1) X(rootfr, D, 15) X(rootr, D, 15)                       \
  X(eff_porosity, D, 15)                                                                                    \
  X(coszen, D, 1) X(fabd_sun, D, 2) X(fabd_sha, D, 2) X(fabi_sun, D, 2) X(fabi_sha, D, 2) X(albsnd, D, 2)    \
  X(albsni, D, 2)                                                                                           \
  X(dz, D, 20) X(zsoi, D, 20) X(zisoi, D, 21)                                                               \
  X(vtype, I, 1) X(veg_active, B, 1) X(do_capsnow, I, 1)                                                    \
  X(cnc_bcphi, D, 5) X(cnc_bcpho, D, 5) X(cnc_dst1, D, 5) X(cnc_dst2, D, 5) X(cnc_dst3, D, 5)                \
  X(cnc_dst4, D, 5)                                                                                         \
  /* soil_temperature (next row after the seven wrappers): elm_state.h:86,139-155 */                        \
  X(tkmg, D, 15) X(tkdry, D, 15) X(csol, D, 20) X(fact, D, 20) X(imelt, I, 20) X(xmf, D, 1)                  \
  X(xmf_h2osfc, D, 1) X(qflx_h2osfc_ice, D, 1) X(eflx_h2osfc_snow, D, 1) X(qflx_snofrz, D, 1)                \
  X(qflx_snomelt, D, 1) X(eflx_snomelt, D, 1) X(qflx_snofrz_lyr, D, 5) X(sabg_chk, D, 1)                      \
  /* surface_fluxes + conservation diagnostics: elm_state.h / elm_state_impl.hh:146,297-345 */               \
  X(eflx_soil_grnd, D, 1) X(eflx_lwrad_out, D, 1) X(eflx_lwrad_net, D, 1) X(qflx_evap_grnd, D, 1)            \
  X(qflx_sub_snow, D, 1) X(qflx_dew_snow, D, 1) X(qflx_dew_grnd, D, 1) X(soil_e_balance, D, 1)               \
  X(dtbegin_column_h2o, D, 1) X(h2osno_old, D, 1) X(qflx_sl_top_soil, D, 1)                                  \
  X(frac_veg_nosno_alb, I, 1)                                                                               \
  /* kokkos_init_timestep forcing + phenology functors: forcing records t_idx, t_idx + 1; months start_idx, + 1 */ \
  X(forc_hgt, D, 1) X(hbot, D, 1) X(atm_tbot, D, 2) X(atm_pbot, D, 2) X(atm_qbot, D, 2) X(atm_flds, D, 2)     \
  X(atm_fsds, D, 2) X(atm_prec, D, 2) X(atm_wind, D, 2) X(mlai, D, 2) X(msai, D, 2) X(mhtop, D, 2)            \
  X(mhbot, D, 2)                                                                                            \
  /* kokkos_snow_hydrology: AerosolMasses, AerosolFileInput (aerosol_data.h:11-41), elm_state.h:151,160 */   \
  X(mss_bcphi, D, 5) X(mss_bcpho, D, 5) X(mss_dst1, D, 5) X(mss_dst2, D, 5) X(mss_dst3, D, 5) X(mss_dst4, D, 5) \
  X(aer_bcphi, D, 1) X(aer_bcpho, D, 1) X(aer_bcdep, D, 1) X(aer_dst1_1, D, 1) X(aer_dst1_2, D, 1)            \
  X(aer_dst2_1, D, 1) X(aer_dst2_2, D, 1) X(aer_dst3_1, D, 1) X(aer_dst3_2, D, 1) X(aer_dst4_1, D, 1)         \
  X(aer_dst4_2, D, 1) X(qflx_top_soil, D, 1) X(mflx_neg_snow, D, 1) X(qflx_snow2topsoi, D, 1)                 \
  X(mflx_snowlyr_col, D, 1) X(qflx_rootsoi, D, 15)                                                            \
  /* initialize_kokkos_elm's per-column init functions: S.topo_slope, S.topo_std (elm_state.h) and the wrapper-local soil  \
     texture Views pct_sand, pct_clay, organic (initialize_elm_kokkos.cc:309-311) */                                      \
  X(topo_slope, D, 1) X(topo_std, D, 1) X(pct_sand, D, 15) X(pct_clay, D, 15) X(organic, D, 15)

#define ELMO_CT_D double
#define ELMO_CT_I int
#define ELMO_CT_B unsigned char

typedef struct elmo_state {
  int64_t ncols;
  /* scalars of ELMState (src/data/elm_state.h:194-224) */
  elmo_land land;
  double dewmx;
  int oldfflag;
  double dayl, max_dayl;
  /* shared tables */
  elmo_pft_psn pft_psn[ELMO_MXPFT];
  elmo_pft_alb pft_alb[ELMO_MXPFT];
  double z0mr[ELMO_MXPFT], displar[ELMO_MXPFT];
  double albsat[ELMO_NSOILCOL][2], albdry[ELMO_NSOILCOL][2];
  elmo_snicar snicar;
  /* SnwRdsTable (snicar_data.h:75-84): snowage_tau / kappa / drdt0 [idx_T 11][idx_Tgrd 31][idx_rhos 8], row-major */
  double snowage[3][ELMO_SNOWAGE_N];
  /* cold-start initialisation (initialize_elm_kokkos.cc:312, pft_data.h:72-73): organic_max of the parameter file and the
   * rooting distribution parameters by PFT */
  double organic_max;
  double roota_par[ELMO_MXPFT], rootb_par[ELMO_MXPFT];
  /* per-column fields, [col][lev] */
#define ELMO_DECL(name, kind, nlev) ELMO_CT_##kind *name;
  ELMO_FIELDS(ELMO_DECL)
#undef ELMO_DECL
  uint32_t *err_flags;
} elmo_state;

/* ---- state management / registry (elmo_driver.c) ---- */
elmo_state *elmo_create(int64_t ncols);
void elmo_destroy(elmo_state *S);
int elmo_num_fields(void);
const char *elmo_field_name(int i);
/* returns base pointer of field `name` (NULL if unknown); kind: 0 double, 1 int, 2 uchar */
void *elmo_field_ptr(elmo_state *S, const char *name, int *nlev, int *kind);
/* pointers to the shared-parameter blocks so Python can fill them with ctypes/numpy */
elmo_snicar *elmo_snicar_ptr(elmo_state *S);
elmo_pft_psn *elmo_pft_psn_ptr(elmo_state *S);
elmo_pft_alb *elmo_pft_alb_ptr(elmo_state *S);
double *elmo_z0mr_ptr(elmo_state *S);
double *elmo_displar_ptr(elmo_state *S);
double *elmo_albsat_ptr(elmo_state *S);
double *elmo_albdry_ptr(elmo_state *S);
void elmo_set_scalars(elmo_state *S, int ltype, int ctype, int vtype, int urbpoi, int lakpoi, double dewmx,
                      int oldfflag, double dayl, double max_dayl);
void elmo_set_threads(int n);
int elmo_get_max_threads(void);

/* ---- L3 wrappers, same names/order as driver/kokkos (elmo_driver.c) ---- */
void elmo_frac_wet(elmo_state *S);                        /* canopy_hydrology_kokkos.cc:98-112 */
void elmo_albedo_snicar(elmo_state *S);                   /* albedo_kokkos.cc:10-376 */
void elmo_canopy_hydrology(elmo_state *S, double dt);     /* canopy_hydrology_kokkos.cc:7-95 */
void elmo_surface_radiation(elmo_state *S);               /* surface_radiation_kokkos.cc:7-97 */
void elmo_canopy_temperature(elmo_state *S);              /* canopy_temperature_kokkos.cc:6-131 */
void elmo_bareground_fluxes(elmo_state *S);               /* bareground_fluxes_kokkos.cc:7-123 */
void elmo_canopy_fluxes(elmo_state *S, double dt);        /* canopy_fluxes_kokkos.cc:6-265 */
void elmo_canopy_fluxes_given(elmo_state *S, double dt, const double *rho_in, const double *po2_in,
                              const double *pco2_in, int *niter);
void elmo_bareground_fluxes_given(elmo_state *S, const double *rho_in);
void elmo_albedo_snicar_ex(elmo_state *S, double *fabd_sun_out, double *fabd_sha_out);
/* kokkos_snow_hydrology (snow_hydrology_kokkos.cc:23-188; ELMInterface::advance :313, between soil_temperature and
 * surface_fluxes); elmo_physics_g.c, whose header says what is pinned bit for bit by the reference's own functions (all the
 * snow functions but snow_aging) and what is not (snow_aging, the two aerosol bookkeeping functions). */
void elmo_snow_hydrology(elmo_state *S, double dt);
/* the same wrapper one stage at a time (0..ELMO_SNOW_HYDROLOGY_STAGES-1 in the wrapper's order: snow_water, aerosol
   deposition, aerosol_phase_change, transpiration, snow_compaction, combine_layers, divide_layers, prune_snow_layers,
   aerosol mass / concentration, snow_aging), over all columns: lets a test put the reference's own function of a stage
   (oracle/ref_harness_snow.cc) on the inputs the restatement's stage saw */
#define ELMO_SNOW_HYDROLOGY_STAGES 10
void elmo_snow_hydrology_stage(elmo_state *S, double dt, int stage);
/* the "init functions" lambda of ELM::initialize_kokkos_elm (initialize_elm_kokkos.cc:373-428); elmo_physics_h.c */
void elmo_initialize_state(elmo_state *S);
void elmo_set_init_params(elmo_state *S, double organic_max, const double *roota_par, const double *rootb_par);
double elmo_init_topo_slope(double raw_topo_slope);
double elmo_init_melt_factor(int ltype, double topo_std);
double elmo_init_micro_sigma(double topo_slope);
void elmo_init_snow_layers(double snow_depth, int lakpoi, int *snl_io, double *dz, double *z, double *zi);
void elmo_soil_hydraulic_params(double pct_sand, double pct_clay, double zsoi, double om_frac, double *watsat, double *bsw,
                                double *sucsat, double *watdry, double *watopt, double *watfc, double *tkmg, double *tkdry,
                                double *csol);
void elmo_init_soil_hydraulics(double organic_max, const double *pct_sand, const double *pct_clay, const double *organic,
                               const double *zsoi, double *watsat, double *bsw, double *sucsat, double *watdry,
                               double *watopt, double *watfc, double *tkmg, double *tkdry, double *csol);
void elmo_init_vegrootfr(int vtype, double roota_par, double rootb_par, const double *zi, double *rootfr);
void elmo_init_soil_temp(const elmo_land *L, int snl, double *t_soisno, double *t_grnd);
void elmo_init_snow_state(int urbpoi, int snl, double *h2osno, double *int_snow, double *snow_depth, double *h2osfc,
                          double *h2ocan, double *frac_h2osfc, double *fwet, double *fdry, double *frac_sno, double *snw_rds);
void elmo_init_soilh2o_state(const elmo_land *L, int snl, const double *watsat, const double *t_soisno, const double *dz,
                             double *h2osoi_vol, double *h2osoi_liq, double *h2osoi_ice);
double *elmo_snowage_ptr(elmo_state *S); /* 3 x ELMO_SNOWAGE_N: tau, kappa, drdt0 */
void elmo_snow_aging(int do_capsnow, int snl, double frac_sno, double dtime, double qflx_snwcp_ice, double qflx_snow_grnd,
                     double h2osno, const double *dz, const double *h2osoi_liq, const double *h2osoi_ice,
                     const double *t_soisno, const double *qflx_snofrz_lyr, const double *snowage_tau,
                     const double *snowage_kappa, const double *snowage_drdt0, double *snw_rds, uint32_t *err);
void elmo_snow_water(int do_capsnow, int snl, double dtime, double frac_sno_eff, double h2osno, double qflx_sub_snow,
                     double qflx_evap_grnd, double qflx_dew_snow, double qflx_dew_grnd, double qflx_rain_grnd,
                     double qflx_snomelt, double *qflx_snow_melt, double *qflx_top_soil, double *int_snow, double *frac_sno,
                     double *mflx_neg_snow, double *h2osoi_liq, double *h2osoi_ice, double *mss_bcphi, double *mss_bcpho,
                     double *mss_dst1, double *mss_dst2, double *mss_dst3, double *mss_dst4, double *dz, uint32_t *err);
void elmo_aerosol_phase_change(int snl, double dtime, double qflx_sub_snow, const double *h2osoi_liq, const double *h2osoi_ice,
                               double *mss_bcphi, double *mss_bcpho);
void elmo_transpiration(int veg_active, double qflx_tran_veg, const double *rootr, double *qflx_rootsoi);
void elmo_snow_compaction(int snl, int ltype, double dtime, double int_snow, double n_melt, double frac_sno, const int *imelt,
                          const double *swe_old, const double *h2osoi_liq, const double *h2osoi_ice, const double *t_soisno,
                          const double *frac_iceold, double *dz);
void elmo_combine_layers(int urbpoi, int ltype, double dtime, int *snl, double *h2osno, double *snow_depth,
                         double *frac_sno_eff, double *frac_sno, double *int_snow, double *qflx_sl_top_soil,
                         double *qflx_snow2topsoi, double *mflx_snowlyr_col, double *t_soisno, double *h2osoi_ice,
                         double *h2osoi_liq, double *snw_rds, double *mss_bcphi, double *mss_bcpho, double *mss_dst1,
                         double *mss_dst2, double *mss_dst3, double *mss_dst4, double *dz, double *z, double *zi, uint32_t *err);
void elmo_divide_layers(double frac_sno, int *snl, double *h2osoi_ice, double *h2osoi_liq, double *t_soisno, double *snw_rds,
                        double *mss_bcphi, double *mss_bcpho, double *mss_dst1, double *mss_dst2, double *mss_dst3,
                        double *mss_dst4, double *dz, double *z, double *zi, uint32_t *err);
void elmo_prune_snow_layers(int snl, double *h2osoi_ice, double *h2osoi_liq, double *t_soisno, double *dz, double *z, double *zi);
void elmo_aerosol_deposition(double dtime, int snl, const double *aer, double *mss_bcphi, double *mss_bcpho, double *mss_dst1,
                             double *mss_dst2, double *mss_dst3, double *mss_dst4);
void elmo_aerosol_mass_and_concen(double dtime, int snl, int do_capsnow, double qflx_snwcp_ice, const double *h2osoi_ice,
                                  const double *h2osoi_liq, double *const mss[6], double *const cnc[6]);
/* next row: soil_temperature_kokkos.cc:6-278 (follows the seven wrappers in ELMInterface::advance, :310) */
void elmo_soil_temperature(elmo_state *S, double dt);
/* the same with the intermediate system exposed, for the residual / energy-balance checks of the tests:
   lhs [ncols][21][5], rhs [ncols][21] (right-hand side before the solve), sol [ncols][21], cv [ncols][20],
   hs [ncols][4] = {hs_soil, hs_h2osfc, hs_top_snow, dhsdT}; any pointer may be NULL */
void elmo_soil_temperature_ex(elmo_state *S, double dt, double *lhs, double *rhs, double *sol, double *cv, double *hs);
/* surface_fluxes_kokkos.cc:5-107 and conserved_quantity_kokkos.cc:8-81.  The conservation wrapper of the reference
   keeps its eight diagnostics in wrapper-local Views (and prints column 0); here they are returned:
   diag [ncols][8] = dtend_column_h2o, errh2o, errh2osno, dwb, errsol, errlon, errseb, netrad */
void elmo_surface_fluxes(elmo_state *S, double dt);
/* the per-column kernel of kokkos_init_timestep (init_timestep_kokkos.cc:55-75): h2osno_old, dtbegin_column_h2o,
   ELM::init_timestep (init_timestep_impl.hh:7-42) */
void elmo_init_timestep(elmo_state *S);
/* get_forcing (driver/kokkos/atm_forcing_kokkos.cc:47-75): the eight ComputeAtmForcing_* functors of
   src/physics/atm_physics_impl.hh:27-203 in the wrapper's order (TBOT, PBOT, QBOT|RH, FLDS, FSDS, PREC, WIND, ZBOT).
   wt1, wt2 [8]: AtmDataManager::forcing_time_weights of each stream (atm_data_impl.hh:191-199; unused for FSDS, PREC,
   ZBOT); qbot_is_rh: the humidity stream holds relative humidity (AtmForcType::RH) */
void elmo_get_forcing(elmo_state *S, const double *wt1, const double *wt2, int qbot_is_rh);
/* ComputePhenology (src/physics/phenology_physics_impl.hh:22-69), as run by update_phenology
   (driver/kokkos/phenology_kokkos.cc:59-62) */
void elmo_phenology(elmo_state *S, double wt1, double wt2);
void elmo_evaluate_conservation(elmo_state *S, double dt, double *diag);
/* probes matching ref_harness.cc (the parts of this path the reference's headers build for) */
void elmo_soil_thermal(elmo_state *S, double *thk_out, double *tk_out, double *cv_out, double *scal_out);
void elmo_pdma(int64_t n, const int *snl, const double *lhs, double *rhs);
void elmo_phase_change(elmo_state *S, double dt, const double *dhsdT, const double *c_h2osfc);
void elmo_timestep7(elmo_state *S, double dt);            /* elm_kokkos_interface.cc:289-307 order */

/* ---- L2 physics, one column (elmo_physics.c) ---- */
void elmo_qsat(double T, double p, double *es, double *esdT, double *qs, double *qsdT);
double elmo_derive_forc_vp(double forc_qbot, double forc_pbot);
double elmo_derive_forc_rho(double forc_pbot, double forc_qbot, double forc_tbot);
double elmo_derive_forc_po2(double forc_pbot);
double elmo_derive_forc_pco2(double forc_pbot);

void elmo_ch_interception(const elmo_land *L, int frac_veg_nosno, double forc_rain, double forc_snow, double dewmx,
                          double elai, double esai, double dtime, double *h2ocan, double *qflx_candrip,
                          double *qflx_through_snow, double *qflx_through_rain, double *fracsnow, double *fracrain);
void elmo_ch_ground_flux(const elmo_land *L, int do_capsnow, int frac_veg_nosno, double forc_rain, double forc_snow,
                         double qflx_irrig, double qflx_candrip, double qflx_through_snow, double qflx_through_rain,
                         double fracsnow, double fracrain, double *qflx_snwcp_liq, double *qflx_snwcp_ice,
                         double *qflx_snow_grnd, double *qflx_rain_grnd);
void elmo_ch_fraction_wet(const elmo_land *L, int frac_veg_nosno, double dewmx, double elai, double esai,
                          double h2ocan, double *fwet, double *fdry);
void elmo_ch_snow_init(const elmo_land *L, double dtime, int do_capsnow, int oldfflag, double forc_t, double t_grnd,
                       double qflx_snow_grnd, double qflx_snow_melt, double n_melt, double *snow_depth,
                       double *h2osno, double *int_snow, double *swe_old, double *h2osoi_liq, double *h2osoi_ice,
                       double *t_soisno, double *frac_iceold, int *snl, double *dz, double *z, double *zi,
                       double *snw_rds, double *frac_sno_eff, double *frac_sno);
void elmo_ch_fraction_h2osfc(const elmo_land *L, double micro_sigma, double h2osno, double *h2osfc,
                             double *h2osoi_liq, double *frac_sno, double *frac_sno_eff, double *frac_h2osfc);

void elmo_sr_canopy_sunshade_fractions(const elmo_land *L, int nrad, double elai, const double *tlai_z,
                                       const double *fsun_z, const double *forc_solad, const double *forc_solai,
                                       const double *fabd_sun_z, const double *fabd_sha_z, const double *fabi_sun_z,
                                       const double *fabi_sha_z, double *parsun_z, double *parsha_z,
                                       double *laisun_z, double *laisha_z, double *laisun, double *laisha);
void elmo_sr_initialize_flux(const elmo_land *L, double *sabg_soil, double *sabg_snow, double *sabg, double *sabv,
                             double *fsa, double *sabg_lyr);
void elmo_sr_total_absorbed_radiation(const elmo_land *L, int snl, const double *ftdd, const double *ftid,
                                      const double *ftii, const double *forc_solad, const double *forc_solai,
                                      const double *fabd, const double *fabi, const double *albsod,
                                      const double *albsoi, const double *albsnd, const double *albsni,
                                      const double *albgrd, const double *albgri, double *sabv, double *fsa,
                                      double *sabg, double *sabg_soil, double *sabg_snow, double *trd, double *tri);
unsigned elmo_sr_layer_absorbed_radiation(const elmo_land *L, int snl, double sabg, double sabg_snow,
                                          double snow_depth, const double *flx_absdv, const double *flx_absdn,
                                          const double *flx_absiv, const double *flx_absin, const double *trd,
                                          const double *tri, double *sabg_lyr);
void elmo_sr_reflected_radiation(const elmo_land *L, const double *albd, const double *albi,
                                 const double *forc_solad, const double *forc_solai, double *fsr);

void elmo_ct_old_ground_temp(const elmo_land *L, double t_h2osfc, const double *t_soisno, double *t_h2osfc_bef,
                             double *tssbef);
void elmo_ct_ground_temp(const elmo_land *L, int snl, double frac_sno_eff, double frac_h2osfc, double t_h2osfc,
                         const double *t_soisno, double *t_grnd);
void elmo_ct_calc_soilalpha(const elmo_land *L, double frac_sno, double frac_h2osfc, const double *h2osoi_liq,
                            const double *h2osoi_ice, const double *dz, const double *t_soisno, const double *watsat,
                            const double *sucsat, const double *bsw, const double *watdry, const double *watopt,
                            double *qred, double *hr, double *soilalpha);
void elmo_ct_calc_soilbeta(const elmo_land *L, double frac_sno, double frac_h2osfc, const double *watsat,
                           const double *watfc, const double *h2osoi_liq, const double *h2osoi_ice, const double *dz,
                           double *soilbeta);
void elmo_ct_humidities(const elmo_land *L, int snl, double forc_q, double forc_pbot, double t_h2osfc, double t_grnd,
                        double frac_sno, double frac_sno_eff, double frac_h2osfc, double qred, double hr,
                        const double *t_soisno, double *qg_snow, double *qg_soil, double *qg, double *qg_h2osfc,
                        double *dqgdT);
void elmo_ct_ground_properties(const elmo_land *L, int snl, double frac_sno, double forc_th, double forc_q,
                               double elai, double esai, double htop, const double *displar, const double *z0mr,
                               const double *h2osoi_liq, const double *h2osoi_ice, double *emg, double *emv,
                               double *htvp, double *z0mg, double *z0hg, double *z0qg, double *z0mv, double *z0hv,
                               double *z0qv, double *thv, double *z0m, double *displa);
void elmo_ct_forcing_height(const elmo_land *L, int veg_active, int frac_veg_nosno, double z0m, double z0mg,
                            double forc_t, double displa, double *forc_hgt_u_patch, double *forc_hgt_t_patch,
                            double *forc_hgt_q_patch, double *thm);
void elmo_ct_init_energy_fluxes(const elmo_land *L, double *eflx_sh_tot, double *eflx_lh_tot, double *eflx_sh_veg,
                                double *qflx_evap_tot, double *qflx_evap_veg, double *qflx_tran_veg);

void elmo_fv_monin_obukhov_length(double ur, double thv, double dthv, double zldis, double z0m, double *um,
                                  double *obu);
void elmo_fv_wind(double forc_hgt_u_patch, double displa, double um, double obu, double z0m, double *ustar);
void elmo_fv_temp(double forc_hgt_t_patch, double displa, double obu, double z0h, double *temp1);
void elmo_fv_humidity(double forc_hgt_q_patch, double forc_hgt_t_patch, double displa, double obu, double z0h,
                      double z0q, double temp1, double *temp2);
void elmo_fv_temp2m(double obu, double z0h, double *temp12m);
void elmo_fv_humidity2m(double obu, double z0h, double z0q, double temp12m, double *temp22m);

void elmo_bg_initialize_flux(const elmo_land *L, int frac_veg_nosno, double forc_u, double forc_v, double forc_q,
                             double forc_th, double forc_hgt_u_patch, double thm, double thv, double t_grnd,
                             double qg, double z0mg, double *dlrad, double *ulrad, double *zldis, double *displa,
                             double *dth, double *dqh, double *obu, double *ur, double *um);
void elmo_bg_stability_iteration(const elmo_land *L, int frac_veg_nosno, double forc_hgt_t_patch,
                                 double forc_hgt_u_patch, double forc_hgt_q_patch, double z0mg, double zldis,
                                 double displa, double dth, double dqh, double ur, double forc_q, double forc_th,
                                 double thv, double *z0hg, double *z0qg, double *obu, double *um, double *temp1,
                                 double *temp2, double *temp12m, double *temp22m, double *ustar);
void elmo_bg_compute_flux(const elmo_land *L, int frac_veg_nosno, int snl, double forc_rho, double soilbeta,
                          double dqgdT, double htvp, double t_h2osfc, double qg_snow, double qg_soil,
                          double qg_h2osfc, const double *t_soisno, double forc_pbot, double dth, double dqh,
                          double temp1, double temp2, double temp12m, double temp22m, double ustar, double forc_q,
                          double thm, double *cgrnds, double *cgrndl, double *cgrnd, double *eflx_sh_grnd,
                          double *eflx_sh_tot, double *eflx_sh_snow, double *eflx_sh_soil, double *eflx_sh_h2osfc,
                          double *qflx_evap_soi, double *qflx_evap_tot, double *qflx_ev_snow, double *qflx_ev_soil,
                          double *qflx_ev_h2osfc, double *t_ref2m, double *q_ref2m, double *rh_ref2m);

void elmo_sms_calc_effective_soilporosity(const double *watsat, const double *h2osoi_ice, const double *dz,
                                          double *eff_por);
void elmo_sms_calc_volumetric_h2oliq(const double *eff_por, const double *h2osoi_liq, const double *dz,
                                     double *vol_liq);
void elmo_sms_calc_root_moist_stress(const double *h2osoi_liqvol, const double *rootfr, const double *t_soisno,
                                     double tc_stress, const double *sucsat, const double *watsat, const double *bsw,
                                     double smpso, double smpsc, const double *eff_porosity, int altmax_indx,
                                     int altmax_lastyear_indx, double *rootr, double *btran);

/* photosynthesis() alone over n independent inputs (elmo_driver.c; same interface as elmref_photosynthesis) */
void elmo_photosynthesis_batch(int64_t n, const elmo_pft_psn *table, const int *vtype, const int *nrad, const double *in,
                               double *out, unsigned *err);
/* test infrastructure: branch counters of the photosynthesis root find (elmo_physics_b.c) */
void elmo_psn_counters(unsigned long long *out4, int reset);
unsigned elmo_psn_photosynthesis(const elmo_pft_psn *psnveg, int nrad, double forc_pbot, double t_veg, double t10,
                                 double esat_tv, double eair, double oair, double cair, double rb, double btran,
                                 double dayl_factor, double thm, const double *tlai_z, double vcmaxcint,
                                 const double *par_z, const double *lai_z, double *ci_z, double *rs);

/* scratch that lives across the three canopy_fluxes calls (canopy_fluxes_kokkos.cc:11-40) */
typedef struct {
  double wtg, wtgq, wtalq, wtlq0, wtaq0, wtl0, wta0, wtal, dayl_factor, air, bir, cir, el, qsatl, qsatldT, taf, qaf,
      um, ur, dth, dqh, obu, zldis, temp1, temp2, temp12m, temp22m, tlbef, delq, dt_veg;
} elmo_cf_scratch;

unsigned elmo_cf_initialize_flux(const elmo_land *L, int snl, int frac_veg_nosno, double frac_sno,
                                 double forc_hgt_u_patch, double thm, double thv, double max_dayl, double dayl,
                                 int altmax_indx, int altmax_lastyear_indx, const double *t_soisno,
                                 const double *h2osoi_ice, const double *h2osoi_liq, const double *dz,
                                 const double *rootfr, double tc_stress, const double *sucsat, const double *watsat,
                                 const double *bsw, double smpso, double smpsc, double elai, double esai, double emv,
                                 double emg, double qg, double t_grnd, double forc_t, double forc_pbot,
                                 double forc_lwrad, double forc_u, double forc_v, double forc_q, double forc_th,
                                 double z0mg, double *btran, double *displa, double *z0mv, double *z0hv, double *z0qv,
                                 double *rootr, double *eff_porosity, elmo_cf_scratch *w, double *t_veg);
unsigned elmo_cf_stability_iteration(const elmo_land *L, double dtime, int snl, int frac_veg_nosno, double frac_sno,
                                     double forc_hgt_u_patch, double forc_hgt_t_patch, double forc_hgt_q_patch,
                                     double fwet, double fdry, double laisun, double laisha, double forc_rho,
                                     double snow_depth, double soilbeta, double frac_h2osfc, double t_h2osfc,
                                     double sabv, double h2ocan, double htop, const double *t_soisno, double displa,
                                     double elai, double esai, double t_grnd, double forc_pbot, double forc_q,
                                     double forc_th, double z0mg, double z0mv, double z0hv, double z0qv, double thm,
                                     double thv, double qg, const elmo_pft_psn *psn_pft, int nrad, double t10,
                                     const double *tlai_z, double vcmaxcintsha, double vcmaxcintsun,
                                     const double *parsha_z, const double *parsun_z, const double *laisha_z,
                                     const double *laisun_z, double forc_pco2, double forc_po2, double *btran,
                                     double *qflx_tran_veg, double *qflx_evap_veg, double *eflx_sh_veg,
                                     elmo_cf_scratch *w, double *t_veg, int *niter);
void elmo_cf_compute_flux(const elmo_land *L, double dtime, int snl, int frac_veg_nosno, double frac_sno,
                          const double *t_soisno, double frac_h2osfc, double t_h2osfc, double sabv, double qg_snow,
                          double qg_soil, double qg_h2osfc, double dqgdT, double htvp, const elmo_cf_scratch *w,
                          double t_veg, double t_grnd, double forc_pbot, double qflx_tran_veg, double qflx_evap_veg,
                          double eflx_sh_veg, double forc_q, double forc_rho, double thm, double emv, double emg,
                          double forc_lwrad, double *h2ocan, double *eflx_sh_grnd, double *eflx_sh_snow,
                          double *eflx_sh_soil, double *eflx_sh_h2osfc, double *qflx_evap_soi, double *qflx_ev_snow,
                          double *qflx_ev_soil, double *qflx_ev_h2osfc, double *dlrad, double *ulrad, double *cgrnds,
                          double *cgrndl, double *cgrnd, double *t_ref2m, double *q_ref2m, double *rh_ref2m);

/* soil / snow temperature (elmo_physics_d.c) */
void elmo_st_calc_soil_tk(int ltype, const double *h2osoi_liq, const double *h2osoi_ice, const double *t_soisno,
                          const double *dz, const double *watsat, const double *tkmg, const double *tkdry, double *thk);
void elmo_st_calc_snow_tk(int snl, double frac_sno, const double *h2osoi_liq, const double *h2osoi_ice,
                          const double *dz, double *thk);
void elmo_st_calc_face_tk(int snl, const double *thk, const double *z, const double *zi, double *tk);
void elmo_st_calc_soil_heat_capacity(int ltype, int snl, double h2osno, const double *watsat, const double *h2osoi_ice,
                                     const double *h2osoi_liq, const double *dz, const double *csol, double *cv);
void elmo_st_calc_snow_heat_capacity(int snl, double frac_sno, const double *h2osoi_ice, const double *h2osoi_liq,
                                     double *cv);
double elmo_st_calc_h2osfc_tk(double h2osfc, const double *thk, const double *z);
double elmo_st_calc_h2osfc_heat_capacity(int snl, double h2osfc, double frac_h2osfc);
double elmo_st_calc_h2osfc_height(int snl, double h2osfc, double frac_h2osfc);
double elmo_st_calc_surface_heat_flux(int frac_veg_nosno, double dlrad, double emg, double forc_lwrad, double htvp,
                                      double solar_abg, double temp, double eflx_sh, double qflx_ev);
double elmo_st_calc_dhsdT(double cgrnd, double emg, double t_grnd);
double elmo_st_check_absorbed_solar(double frac_sno_eff, double sabg_snow, double sabg_soil);
void elmo_st_calc_diffusive_heat_flux(int snl, const double *tk, const double *t_soisno, const double *z, double *fn);
void elmo_st_calc_heat_flux_matrix_factor(int snl, double dtime, const double *cv, const double *dz, const double *z,
                                          const double *zi, double *fact);
void elmo_st_update_temperature(int snl, double frac_h2osfc, const double *tvector, double *t_h2osfc, double *t_soisno);
void elmo_st_update_t_grnd(int snl, double frac_h2osfc, double frac_sno_eff, double t_h2osfc, const double *t_soisno,
                           double *t_grnd);
void elmo_st_set_rhs(double dtime, int snl, double hs_top_snow, double dhsdT, double hs_soil, double frac_sno_eff,
                     const double *t_soisno, const double *fact, const double *fn, const double *sabg_lyr,
                     const double *z, double tk_h2osfc, double t_h2osfc, double dz_h2osfc, double c_h2osfc,
                     double hs_h2osfc, double *rhs_vec);
void elmo_st_set_lhs(double dtime, int snl, double dz_h2osfc, double c_h2osfc, double tk_h2osfc, double frac_h2osfc,
                     double frac_sno_eff, double dhsdT, const double *z, const double *fact, const double *tk,
                     double *lhs);
void elmo_st_pdma(int snl, const double *LHS, double *A, double *B, double *Z, double *RHS);
void elmo_st_phase_change_h2osfc(int snl, double dtime, double frac_sno, double frac_h2osfc, double dhsdT,
                                 double c_h2osfc, double fact_sl1, double *t_h2osfc, double *h2osfc,
                                 double *xmf_h2osfc, double *qflx_h2osfc_to_ice, double *eflx_h2osfc_to_snow,
                                 double *h2osno, double *int_snow, double *snow_depth, double *h2osoi_ice_sl1,
                                 double *t_soisno_sl1);
void elmo_st_phase_change_soisno(int snl, int ltype, double dtime, double dhsdT, double frac_h2osfc,
                                 double frac_sno_eff, const double *fact, const double *watsat, const double *sucsat,
                                 const double *bsw, const double *dz, double *h2osno, double *snow_depth, double *xmf,
                                 double *qflx_snofrz, double *qflx_snow_melt, double *qflx_snomelt,
                                 double *eflx_snomelt, int *imelt, double *qflx_snofrz_lyr, double *h2osoi_ice,
                                 double *h2osoi_liq, double *t_soisno);

/* surface fluxes and conservation diagnostics (elmo_physics_e.c) */
void elmo_sf_initial_flux_calc(int urbpoi, int snl, double frac_sno_eff, double frac_h2osfc, double t_h2osfc_bef,
                               double tssbef_snotop, double tssbef_soitop, double t_grnd, double cgrnds, double cgrndl,
                               double *eflx_sh_grnd, double *qflx_evap_soi, double *qflx_ev_snow, double *qflx_ev_soil,
                               double *qflx_ev_h2osfc);
void elmo_sf_update_surface_fluxes(int urbpoi, int do_capsnow, int snl, double dtime, double t_grnd, double htvp,
                                   double frac_sno_eff, double frac_h2osfc, double t_h2osfc_bef, double sabg_soil,
                                   double sabg_snow, double dlrad, double frac_veg_nosno, double emg, double forc_lwrad,
                                   double tssbef_snotop, double tssbef_soitop, double h2osoi_ice_snotop,
                                   double h2osoi_liq_snotop, double eflx_sh_veg, double qflx_evap_veg,
                                   double *qflx_evap_soi, double *eflx_sh_grnd, double *qflx_ev_snow,
                                   double *qflx_ev_soil, double *qflx_ev_h2osfc, double *eflx_soil_grnd,
                                   double *eflx_sh_tot, double *qflx_evap_tot, double *eflx_lh_tot, double *qflx_evap_grnd,
                                   double *qflx_sub_snow, double *qflx_dew_snow, double *qflx_dew_grnd,
                                   double *qflx_snwcp_liq, double *qflx_snwcp_ice);
void elmo_sf_lwrad_outgoing(int urbpoi, int snl, int frac_veg_nosno, double forc_lwrad, double frac_sno_eff,
                            double tssbef_snotop, double tssbef_soitop, double frac_h2osfc, double t_h2osfc_bef,
                            double t_grnd, double ulrad, double emg, double *eflx_lwrad_out, double *eflx_lwrad_net);
double elmo_sf_soil_energy_balance(int ctype, int snl, double eflx_soil_grnd, double xmf, double xmf_h2osfc,
                                   double frac_h2osfc, double t_h2osfc, double t_h2osfc_bef, double dtime,
                                   double eflx_h2osfc_to_snow, double frac_sno_eff, const double *t_soisno,
                                   const double *tssbef, const double *fact);
double elmo_ce_column_water_mass(double h2ocan, double h2osno, double h2osfc, const double *h2osoi_ice,
                                 const double *h2osoi_liq);
double elmo_ce_dh2o_dt(double begwb, double endwb, double dtime);
double elmo_ce_column_water_balance_error(double begwb, double endwb, double hydrology_source_sink, double forc_rain,
                                          double forc_snow, double qflx_evap_tot, double qflx_snwcp_ice, double dtime);
double elmo_ce_snow_water_balance_error(int snl, double qflx_dew_snow, double qflx_dew_grnd, double qflx_sub_snow,
                                        double qflx_evap_grnd, double qflx_snow_melt, double qflx_snwcp_ice,
                                        double qflx_snwcp_liq, double qflx_sl_top_soil, double frac_sno_eff,
                                        double qflx_rain_grnd, double qflx_snow_grnd, double qflx_h2osfc_ice,
                                        double h2osno, double h2osno_old, double dtime, int do_capsnow);
double elmo_ce_solar_shortwave_balance_error(double fsa, double fsr, const double *forc_solad, const double *forc_solai);
double elmo_ce_solar_longwave_balance_error(double eflx_lwrad_out, double eflx_lwrad_net, double forc_lwrad);
double elmo_ce_surface_energy_balance_error(double sabv, double sabg_chk, double forc_lwrad, double eflx_lwrad_out,
                                            double eflx_sh_tot, double eflx_lh_tot, double eflx_soil_grnd);
double elmo_ce_net_radiation(double fsa, double eflx_lwrad_net);

/* surface albedo (surface_albedo_impl.hh) */
void elmo_sa_init_timestep(int urbpoi, double elai, const double *mss_cnc_bcphi, const double *mss_cnc_bcpho,
                           const double *mss_cnc_dst1, const double *mss_cnc_dst2, const double *mss_cnc_dst3,
                           const double *mss_cnc_dst4, double *vcmaxcintsun, double *vcmaxcintsha, double *albsod,
                           double *albsoi, double *albgrd, double *albgri, double *albd, double *albi, double *fabd,
                           double *fabd_sun, double *fabd_sha, double *fabi, double *fabi_sun, double *fabi_sha,
                           double *ftdd, double *ftid, double *ftii, double *flx_absdv, double *flx_absdn,
                           double *flx_absiv, double *flx_absin, double *mss_cnc_aer_in_fdb /*[5][8]*/);
void elmo_sa_soil_albedo(const elmo_land *L, int snl, double t_grnd, double coszen, const double *h2osoi_vol,
                         const double *albsat, const double *albdry, double *albsod, double *albsoi);
void elmo_sa_ground_albedo(int urbpoi, double coszen, double frac_sno, const double *albsod, const double *albsoi,
                           const double *albsnd, const double *albsni, double *albgrd, double *albgri);
void elmo_sa_flux_absorption_factor(const elmo_land *L, double coszen, double frac_sno, const double *albsod,
                                    const double *albsoi, const double *albsnd, const double *albsni,
                                    const double *flx_absd_snw /*[6][2]*/, const double *flx_absi_snw /*[6][2]*/,
                                    double *flx_absdv, double *flx_absdn, double *flx_absiv, double *flx_absin);
unsigned elmo_sa_canopy_layer_lai(int urbpoi, double elai, double esai, double tlai, double tsai, int *nrad,
                                  double *tlai_z, double *tsai_z, double *fsun_z, double *fabd_sun_z,
                                  double *fabd_sha_z, double *fabi_sun_z, double *fabi_sha_z);
void elmo_sa_two_stream_solver(const elmo_land *L, int nrad, double coszen, double t_veg, double fwet, double elai,
                               double esai, const double *tlai_z, const double *tsai_z, const double *albgrd,
                               const double *albgri, const elmo_pft_alb *alb_pft, double *vcmaxcintsun,
                               double *vcmaxcintsha, double *albd, double *ftid, double *ftdd, double *fabd,
                               double *fabd_sun, double *fabd_sha, double *albi, double *ftii, double *fabi,
                               double *fabi_sun, double *fabi_sha, double *fsun_z, double *fabd_sun_z,
                               double *fabd_sha_z, double *fabi_sun_z, double *fabi_sha_z);

/* SNICAR (snow_snicar_impl.hh) */
unsigned elmo_sn_init_timestep(int urbpoi, int flg_slr_in, double coszen, double h2osno, int snl,
                               const double *h2osoi_liq, const double *h2osoi_ice, const double *snw_rds,
                               int *snl_top, int *snl_btm, double *flx_abs_lcl /*[6][5]*/, double *flx_abs /*[6][2]*/,
                               int *flg_nosnl, double *h2osoi_ice_lcl, double *h2osoi_liq_lcl, int *snw_rds_lcl,
                               double *mu_not, double *flx_slrd_lcl, double *flx_slri_lcl);
void elmo_sn_snow_aerosol_mie_params(int urbpoi, int flg_slr_in, int snl_top, int snl_btm, double coszen,
                                     double h2osno, const int *snw_rds_lcl, const double *h2osoi_ice_lcl,
                                     const double *h2osoi_liq_lcl, const elmo_snicar *T,
                                     const double *mss_cnc_aer_in /*[5][8]*/, double *g_star /*[5][5]*/,
                                     double *omega_star, double *tau_star);
unsigned elmo_sn_snow_radiative_transfer_solver(int urbpoi, int flg_slr_in, int flg_nosnl, int snl_top, int snl_btm,
                                                double coszen, double h2osno, double mu_not,
                                                const double *flx_slrd_lcl, const double *flx_slri_lcl,
                                                const double *albsoi, const double *g_star,
                                                const double *omega_star, const double *tau_star, double *albout_lcl,
                                                double *flx_abs_lcl);
void elmo_sn_snow_albedo_radiation_factor(int urbpoi, int flg_slr_in, int snl_top, double coszen, double mu_not,
                                          double h2osno, const int *snw_rds_lcl, const double *albsoi,
                                          const double *albout_lcl, const double *flx_abs_lcl, double *albout,
                                          double *flx_abs);

#ifdef __cplusplus
}
#endif
#endif
