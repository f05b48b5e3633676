/* elmo_physics_f.c - TEST INFRASTRUCTURE (see elm_oracle.h): plain-C restatement of the per-column functors that
 * kokkos_init_timestep runs before its own kernel:
 *   - get_forcing (driver/kokkos/atm_forcing_kokkos.cc:47-75): ProcessTBOT/PBOT/QBOT/FLDS/FSDS/PREC/WIND/ZBOT,
 *     src/physics/atm_physics_impl.hh:27-245
 *   - ComputePhenology, src/physics/phenology_physics_impl.hh:22-69
 * Both reference headers compile here: tests/test_oracle_vs_ref.py pins every function bit for bit against them. */
#include <math.h>

#include "elm_oracle.h"
#include "elmo_const.h"

static inline double f_min(double a, double b) { return (b < a) ? b : a; } /* std::min */
static inline double f_max(double a, double b) { return (a < b) ? b : a; } /* std::max */

/* atm_physics_impl.hh:205-210 */
static inline double interp_forcing(double wt1, double wt2, double forc1, double forc2) { return forc1 * wt1 + forc2 * wt2; }
/* :212-214 */
static inline double tdc(double t) { return f_min(50.0, f_max(-50.0, (t - 273.15))); }
/* :216-230 */
static inline double esatw(double t)
{
  const double a0 = 6.107799961, a1 = 4.436518521e-01, a2 = 1.428945805e-02, a3 = 2.650648471e-04, a4 = 3.031240396e-06,
               a5 = 2.034080948e-08, a6 = 6.136820929e-11;
  return 100.0 * (a0 + t * (a1 + t * (a2 + t * (a3 + t * (a4 + t * (a5 + t * a6))))));
}
/* :232-245 */
static inline double esati(double t)
{
  const double b0 = 6.109177956, b1 = 5.034698970e-01, b2 = 1.886013408e-02, b3 = 4.176223716e-04, b4 = 5.824720280e-06,
               b5 = 4.838803174e-08, b6 = 1.838826904e-10;
  return 100.0 * (b0 + t * (b1 + t * (b2 + t * (b3 + t * (b4 + t * (b5 + t * b6))))));
}

void elmo_get_forcing(elmo_state *S, const double *wt1, const double *wt2, int qbot_is_rh)
{
  enum { TBOT, PBOT, QBOT, FLDS, FSDS, PREC, WIND, ZBOT };

#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    /* ProcessTBOT :38-42 */
    const double tbot = f_min(interp_forcing(wt1[TBOT], wt2[TBOT], S->atm_tbot[c * 2], S->atm_tbot[c * 2 + 1]), 323.0);
    S->forc_tbot[c] = tbot;
    S->forc_thbot[c] = tbot;
    /* ProcessPBOT :55-58 */
    const double pbot = f_max(interp_forcing(wt1[PBOT], wt2[PBOT], S->atm_pbot[c * 2], S->atm_pbot[c * 2 + 1]), 4.0e4);
    S->forc_pbot[c] = pbot;
    /* ProcessQBOT :73-81 */
    double qbot = f_max(interp_forcing(wt1[QBOT], wt2[QBOT], S->atm_qbot[c * 2], S->atm_qbot[c * 2 + 1]), 1.0e-9);
    if (qbot_is_rh) {
      const double e = (tbot > TFRZ) ? esatw(tdc(tbot)) : esati(tdc(tbot));
      const double qsat = 0.622 * e / (pbot - 0.378 * e);
      qbot *= qsat / 100.0;
    }
    S->forc_qbot[c] = qbot;
    /* ProcessFLDS :97-107 */
    const double flds = interp_forcing(wt1[FLDS], wt2[FLDS], S->atm_flds[c * 2], S->atm_flds[c * 2 + 1]);
    if (flds <= 50.0 || flds >= 600.0) {
      const double e = pbot * qbot / (0.622 + 0.378 * qbot);
      const double ea = 0.70 + 5.95e-5 * 0.01 * e * exp(1500.0 / tbot);
      S->forc_lwrad[c] = ea * STEBOL * pow(tbot, 4.0);
    } else {
      S->forc_lwrad[c] = flds;
    }
    /* ProcessFSDS :122-142 (record t_idx only) */
    {
      const double swndr = f_max(S->atm_fsds[c * 2] * S->coszen[c] * 0.5, 0.0);
      const double swndf = swndr, swvdr = swndr, swvdf = swndr;
      const double ratio_rvrf_vis =
          f_min(0.99, f_max(0.17639 + 0.00380 * swvdr - 9.0039e-06 * pow(swvdr, 2.0) + 8.1351e-09 * pow(swvdr, 3.0), 0.01));
      const double ratio_rvrf_nir =
          f_min(0.99, f_max(0.29548 + 0.00504 * swndr - 1.4957e-05 * pow(swndr, 2.0) + 1.4881e-08 * pow(swndr, 3.0), 0.01));
      S->forc_solad[c * 2 + 0] = ratio_rvrf_vis * swvdr;
      S->forc_solad[c * 2 + 1] = ratio_rvrf_nir * swndr;
      S->forc_solai[c * 2 + 0] = (1.0 - ratio_rvrf_vis) * swvdf;
      S->forc_solai[c * 2 + 1] = (1.0 - ratio_rvrf_nir) * swndf;
    }
    /* ProcessPREC :157-163 (record t_idx only) */
    {
      const double frac1 = (tbot - TFRZ) * 0.5;
      const double frac2 = f_min(1.0, f_max(0.0, frac1));
      S->forc_rain[c] = frac2 * f_max(S->atm_prec[c * 2], 0.0);
      S->forc_snow[c] = (1.0 - frac2) * f_max(S->atm_prec[c * 2], 0.0);
    }
    /* ProcessWIND :177-181 */
    S->forc_u[c] = interp_forcing(wt1[WIND], wt2[WIND], S->atm_wind[c * 2], S->atm_wind[c * 2 + 1]);
    S->forc_v[c] = 0.0;
    /* ProcessZBOT :195-203 */
    S->forc_hgt[c] = 30.0;
    S->forc_hgt_u_patch[c] = S->forc_hgt[c];
    S->forc_hgt_t_patch[c] = S->forc_hgt[c];
    S->forc_hgt_q_patch[c] = S->forc_hgt[c];
  }
}

/* phenology_physics_impl.hh:22-69 */
void elmo_phenology(elmo_state *S, double wt1, double wt2)
{
  const int noveg = 0, nbrdlf_dcd_brl_shrub = 11; /* elm_constants.h:56,67 */
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < S->ncols; c++) {
    const int vtype = S->vtype[c];
    if (vtype != noveg) {
      S->tlai[c] = wt1 * S->mlai[c * 2] + wt2 * S->mlai[c * 2 + 1];
      S->tsai[c] = wt1 * S->msai[c * 2] + wt2 * S->msai[c * 2 + 1];
      S->htop[c] = wt1 * S->mhtop[c * 2] + wt2 * S->mhtop[c * 2 + 1];
      S->hbot[c] = wt1 * S->mhbot[c * 2] + wt2 * S->mhbot[c * 2 + 1];
    } else {
      S->tlai[c] = 0.0;
      S->tsai[c] = 0.0;
      S->htop[c] = 0.0;
      S->hbot[c] = 0.0;
    }
    double fb;
    if (vtype > noveg && vtype <= nbrdlf_dcd_brl_shrub) {
      const double ol = f_min(f_max(S->snow_depth[c] - S->hbot[c], 0.0), S->htop[c] - S->hbot[c]);
      fb = 1.0 - ol / f_max(1.e-06, S->htop[c] - S->hbot[c]);
    } else {
      fb = 1.0 - f_max(f_min(S->snow_depth[c], 0.2), 0.0) / 0.2;
    }
    S->elai[c] = f_max(S->tlai[c] * (1.0 - S->frac_sno[c]) + S->tlai[c] * fb * S->frac_sno[c], 0.0);
    S->esai[c] = f_max(S->tsai[c] * (1.0 - S->frac_sno[c]) + S->tsai[c] * fb * S->frac_sno[c], 0.0);
    if (S->elai[c] < 0.05) S->elai[c] = 0.0;
    if (S->esai[c] < 0.05) S->esai[c] = 0.0;
    S->frac_veg_nosno_alb[c] = ((S->elai[c] + S->esai[c]) >= 0.05) ? 1 : 0;
  }
}
