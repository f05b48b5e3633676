// k_canopy_fluxes.hip - kokkos_canopy_fluxes (driver/kokkos/canopy_fluxes_kokkos.cc:6-265)
//
//   canopy_fluxes::initialize_flux      src/physics/canopy_fluxes_impl.hh:95-184
//     soil_moist_stress::calc_effective_soilporosity / calc_volumetric_h2oliq / calc_root_moist_stress
//                                       src/physics/soil_moist_stress_impl.hh:62-133
//   canopy_fluxes::stability_iteration  canopy_fluxes_impl.hh:187-452  (leaf-temperature Newton loop, <= 41 trips)
//     photosynthesis::photosynthesis    src/physics/photosynthesis_impl.hh:9-282 (called twice per trip)
//       hybrid :517, ci_func :308, brent :396, quadratic :286, ft/fth/fth25 :623-635
//   canopy_fluxes::compute_flux         canopy_fluxes_impl.hh:456-540
//
// One thread per column.  The kernel is the compute-bound one of the timestep (fp64 transcendentals and
// data-dependent trip counts), so the work is organised to cut instruction count without changing any
// rounded result:
//   * every quantity that the reference recomputes inside photosynthesis() from loop-invariant inputs
//     (PFT constants, t10, dayl_factor, forc_pbot, thm) is evaluated once per column (PsnInv);
//   * the t_veg-dependent Arrhenius factors, identical for the sunlit and shaded call of one trip, are
//     evaluated once per trip (PsnTemp) - 11 exp instead of 22;
//   * the 30 ViewD1(ncols) temporaries of the wrapper (:11-40) and the 15-level work arrays stay in registers.
// The arithmetic of each expression (operand order, parenthesisation) is the reference's.
#define ELMK_MATH_LDS 1  // exp / log / pow tables of elmk_math.h in LDS: every kernel below that evaluates them calls elmk_math_lds_init first
#include <stdlib.h>

#include "elmk_dev.h"
#include "elmk_kernels.h"
#include "elmk_stream.h"
#include "elmk_albedo_col.h"
#include "elmk_snicar.h"

#ifndef CF_PROBE
#define CF_PROBE 0  // 4/5: development timeline probes (tests/tools/cf_timeline.py), never set in the product build
#endif

namespace elmk {


// photosynthesis_impl.hh:623-635
__device__ __forceinline__ double psn_ft(double tl, double ha)
{
  return elmk_exp(ha / (RGAS * 1.0e-3 * (TFRZ + 25.0)) * (1.0 - (TFRZ + 25.0) / tl));
}
__device__ __forceinline__ double psn_fth(double tl, double hd, double se, double scaleFactor)
{
  return scaleFactor / (1.0 + elmk_exp((-hd + se * tl) / (RGAS * 1.0e-3 * tl)));
}
__device__ __forceinline__ double psn_fth25(double hd, double se)
{
  return 1.0 + elmk_exp((-hd + se * (TFRZ + 25.0)) / (RGAS * 1.0e-3 * (TFRZ + 25.0)));
}

// photosynthesis_impl.hh:286-302
__device__ __forceinline__ void psn_quadratic(double a, double b, double c, double& r1, double& r2, uint32_t& err)
{
  if (a == 0.0) err |= ELMK_ERR_PSN_QUADRATIC;
  double q;
  if (b >= 0.0) {
    q = -0.5 * (b + sqrt(b * b - 4.0 * a * c));
  } else {
    q = -0.5 * (b - sqrt(b * b - 4.0 * a * c));
  }
  r1 = q / a;
  if (q != 0.0) {
    r2 = c / q;
  } else {
    r2 = 1.0e36;
  }
}

// Iteration-invariant inputs of photosynthesis() (:22-61, :109-114, :135, :152-154), by where they vary:
//   * per plant functional type (and the day-length factor, one value per call): a row of PFT_N doubles that k_cf_iterate
//     evaluates once per workgroup into LDS (cf_pft_row).  The activation energies are stored divided by
//     RGAS * 1e-3 * (TFRZ + 25), the first operation of ft() (:623-625: ha / (...) * (1 - (TFRZ + 25) / tl), evaluated
//     left to right), so a trip spends no division on them;
//   * per column: PsnCol, assembled from the queue record when the column is loaded (cf_psn_column).
enum : int {
  PFT_lmrha_c, PFT_lmrhd, PFT_lmrse, PFT_vcmaxha_c, PFT_vcmaxhd, PFT_jmaxha_c, PFT_jmaxhd, PFT_tpuha_c, PFT_tpuhd, PFT_kcha_c,
  PFT_koha_c, PFT_cpha_c, PFT_qe, PFT_theta_cj, PFT_bbbopt, PFT_mbbopt, PFT_c3, PFT_vcmax25top, PFT_lmrc, PFT_sqrt_dleaf,
  PFT_N,
  PFT_STRIDE = PFT_N + 1  // odd: rows of different plant types start in different LDS banks
};
__device__ __forceinline__ void cf_pft_row(const DevState* __restrict__ S, int v, double* __restrict__ row)
{
  const double* __restrict__ P = S->pft_psn[v];
  const double k25 = RGAS * 1.0e-3 * (TFRZ + 25.0);
  row[PFT_lmrha_c] = P[P_lmrha] / k25;
  row[PFT_lmrhd] = P[P_lmrhd];
  row[PFT_lmrse] = P[P_lmrse];
  row[PFT_vcmaxha_c] = P[P_vcmaxha] / k25;
  row[PFT_vcmaxhd] = P[P_vcmaxhd];
  row[PFT_jmaxha_c] = P[P_jmaxha] / k25;
  row[PFT_jmaxhd] = P[P_jmaxhd];
  row[PFT_tpuha_c] = P[P_tpuha] / k25;
  row[PFT_tpuhd] = P[P_tpuhd];
  row[PFT_kcha_c] = P[P_kcha] / k25;
  row[PFT_koha_c] = P[P_koha] / k25;
  row[PFT_cpha_c] = P[P_cpha] / k25;
  row[PFT_qe] = P[P_qe];
  row[PFT_theta_cj] = P[P_theta_cj];
  row[PFT_bbbopt] = P[P_bbbopt];
  row[PFT_mbbopt] = P[P_mbbopt];
  row[PFT_c3] = (round(P[P_c3psn]) == 1) ? 1.0 : 0.0;
  // vcmax25top (:28-44): leaf nitrogen, the day-length factor and the nitrogen limitation
  const double dl = S->dayl, mdl = S->max_dayl;
  const double dayl_factor = dmin(1.0, dmax(0.01, (dl * dl) / (mdl * mdl)));
  const double lnc = 1.0 / (P[P_slatop] * P[P_leafcn]);
  const double act25 = P[P_act25] * 1000.0 / 60.0;
  double vcmax25top = lnc * P[P_flnr] * P[P_fnr] * act25 * dayl_factor;
  vcmax25top *= P[P_fnitr];
  row[PFT_vcmax25top] = vcmax25top;
  row[PFT_lmrc] = psn_fth25(P[P_lmrhd], P[P_lmrse]);
  row[PFT_sqrt_dleaf] = sqrt(P[P_dleaf]);
}

// per-trip temperature factors shared by the sunlit and shaded call
struct PsnTemp {
  double ft_vcmax, fth_vcmax, ft_jmax, fth_jmax, ft_tpu, fth_tpu, ft_lmr, fth_lmr;
  double p2, e_lmr_c4, e_vc4a, e_vc4b;  // C4 forms (:94-95, :120-124)
  double kc, ko, cp;
};
// what one phase of photosynthesis() reads besides PsnTemp
struct PsnCol {
  bool c3flag;
  double vcmax25top, jmax25top, cf, qe, theta_cj, bbbopt, mbbopt;
};

// the by-reference argument pack of ci_func / brent / hybrid
struct CiCtx {
  double gb_mol, je, cair, oair, lmr_z, par_z, rh_can, vcmax_z, forc_pbot, cp, kc, ko, qe, tpu_z, kp_z, theta_cj, bbb, mbb;
  // sub-expressions of ci_func that do not depend on ci, evaluated once per solve instead of once per call (same
  // operands, same operations: kc * (1.0 + oair / ko) of :316 and 1.4 / gb_mol of :352)
  double kc_o, r14_gb;
  bool c3flag;
  double gs_mol, ac, aj, ap, ag, an;
  uint32_t err;
#if CF_PROBE >= 4
  uint32_t nev;            // probe only: ci_func evaluations of this lane
  uint32_t* wev;           // probe only: this wave's count of executed ci_func bodies (LDS)
#endif
};

// photosynthesis_impl.hh:308-390
__device__ __forceinline__ double ci_func(double ci, CiCtx& k)
{
  const double theta_ip = 0.95;
#if CF_PROBE >= 4
  k.nev += 1u;
  {
    const unsigned long long m_ = __ballot(1);
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m_) - 1) *k.wev += 1u;
  }
#endif
  if (k.c3flag) {
    k.ac = k.vcmax_z * dmax(ci - k.cp, 0.0) / (ci + k.kc_o);
    k.aj = k.je * dmax(ci - k.cp, 0.0) / (4.0 * ci + 8.0 * k.cp);
    k.ap = 3.0 * k.tpu_z;
  } else {
    k.ac = k.vcmax_z;
    k.aj = k.qe * k.par_z * 4.6;
    k.ap = k.kp_z * dmax(ci, 0.0) / k.forc_pbot;
  }
  double r1, r2;
  psn_quadratic(k.theta_cj, -(k.ac + k.aj), k.ac * k.aj, r1, r2, k.err);
  const double ai = dmin(r1, r2);
  psn_quadratic(theta_ip, -(ai + k.ap), ai * k.ap, r1, r2, k.err);
  k.ag = dmin(r1, r2);
  k.an = k.ag - k.lmr_z;
  if (k.an < 0.0) return 0.0;
  double cs = k.cair - k.r14_gb * k.an * k.forc_pbot;
  cs = dmax(cs, 1.e-6);
  const double aquad = cs;
  const double bquad = cs * (k.gb_mol - k.bbb) - k.mbb * k.an * k.forc_pbot;
  const double cquad = -k.gb_mol * (cs * k.bbb + k.mbb * k.an * k.forc_pbot * k.rh_can);
  psn_quadratic(aquad, bquad, cquad, r1, r2, k.err);
  k.gs_mol = dmax(r1, r2);
  return ci - k.cair + k.an * k.forc_pbot * (1.4 * k.gs_mol + 1.6 * k.gb_mol) / (k.gb_mol * k.gs_mol);
}

// photosynthesis_impl.hh:396-511
#ifndef CF_BRENT_ATTR
#define CF_BRENT_ATTR __forceinline__  // not a call: see the note at psn_hybrid
#endif
__device__ CF_BRENT_ATTR double psn_brent(double x1, double x2, double f1, double f2, double tol, CiCtx& k)
{
  const int ITMAX = 20;
  const double EPS = 1.0e-2;
  double d = 0.0, e = 0.0, p, q, r, s, tol1, xm;
  double a = x1, b = x2, fa = f1, fb = f2;
  if ((fa > 0.0 && fb > 0.0) || (fa < 0.0 && fb < 0.0)) k.err |= ELMK_ERR_PSN_BRENT_BRACKET;
  double c = b, fc = fb;
  int iter = 0;
  while (iter != ITMAX) {
    iter += 1;
    if ((fb > 0.0 && fc > 0.0) || (fb < 0.0 && fc < 0.0)) {
      c = a;
      fc = fa;
      d = b - a;
      e = d;
    }
    if (fabs(fc) < fabs(fb)) {
      a = b;
      b = c;
      c = a;
      fa = fb;
      fb = fc;
      fc = fa;
    }
    tol1 = 2.0 * EPS * fabs(b) + 0.5 * tol;
    xm = 0.5 * (c - b);
    if (fabs(xm) <= tol1 || fb == 0.0) return b;
    if (fabs(e) >= tol1 && fabs(fa) > fabs(fb)) {
      s = fb / fa;
      if (a == c) {
        p = 2.0 * xm * s;
        q = 1.0 - s;
      } else {
        q = fa / fc;
        r = fb / fc;
        p = s * (2.0 * xm * q * (q - r) - (b - a) * (r - 1.0));
        q = (q - 1.0) * (r - 1.0) * (s - 1.0);
      }
      if (p > 0.0) q *= -1.0;
      p = fabs(p);
      if (2.0 * p < dmin(3.0 * xm * q - fabs(tol1 * q), fabs(e * q))) {
        e = d;
        d = p / q;
      } else {
        d = xm;
        e = d;
      }
    } else {
      d = xm;
      e = d;
    }
    a = b;
    fa = fb;
    if (fabs(d) > tol1) {
      b = b + d;
    } else {
      b = b + copysign(tol1, xm);
    }
    fb = ci_func(b, k);
    if (fb == 0.0) break;
  }
  return b;
}

// Brent's method is inlined: as a real call it sat inside divergent control flow of a kernel that spills SGPRs to
// VGPR lanes, and a build with a few more live SGPRs corrupted the state of the lanes that were inactive at the call.
// photosynthesis_impl.hh:517-620 (x0 is only an output in the reference; its final value is never read again,
// what survives the solve is the CiCtx state of the LAST ci_func evaluation)
__device__ __forceinline__ void psn_hybrid(double x0, CiCtx& k)
{
  const double eps = 1.0e-2;
  const double eps1 = 1.0e-4;
  const int itmax = 40;
  double f0 = ci_func(x0, k);
  if (f0 == 0.0) return;
  double minx = x0, minf = f0;
  double x1 = x0 * 0.99;
  double f1 = ci_func(x1, k);
  if (f1 == 0.0) return;
  if (f1 < minf) {
    minx = x1;
    minf = f1;
  }
  int iter = 0;
  for (;;) {
    iter += 1;
    const double dx = -f1 * (x1 - x0) / (f1 - f0);
    const double x = x1 + dx;
    const double tol = fabs(x) * eps;
    if (fabs(dx) < tol) break;
    x0 = x1;
    f0 = f1;
    x1 = x;
    f1 = ci_func(x1, k);
    if (f1 < minf) {
      minx = x1;
      minf = f1;
    }
    if (fabs(f1) <= eps1) break;
    if (f1 * f0 < 0.0) {
      (void)psn_brent(x0, x1, f0, f1, tol, k);
#if CF_PROBE >= 4
      k.err |= 0x80000000u;  // probe only: Brent was needed
#endif
      break;
    }
    if (iter > itmax) {
      (void)ci_func(minx, k);
      break;
    }
  }
}

// ft / fth with the pre-divided activation energy and the trip's (1 - (TFRZ + 25) / tl) (:623-630)
__device__ __forceinline__ double psn_ft_c(double ha_c, double fac) { return elmk_exp(ha_c * fac); }

// t_veg-dependent factors of one trip (photosynthesis_impl.hh:63-135); the carboxylation / electron-transport /
// Michaelis-Menten factors only feed the daytime branch (par_z > 0) of either phase.  R: the column's PFT row in LDS.
__device__ __forceinline__ PsnTemp psn_temp(const double* R, bool c3flag, double vcmaxse, double jmaxse, double vcmaxc,
                                            double jmaxc, double tpuc, double kc25, double ko25, double cp25, double t_veg,
                                            bool day)
{
  PsnTemp T;
  T.ft_lmr = T.fth_lmr = T.e_lmr_c4 = T.e_vc4a = T.e_vc4b = T.p2 = 0.0;
  T.ft_vcmax = T.fth_vcmax = T.ft_jmax = T.fth_jmax = T.ft_tpu = T.fth_tpu = T.kc = T.ko = T.cp = 0.0;
  // Night columns (par_z <= 0 in both phases, or no canopy layer) need none of this: their stomatal resistance is
  // min(rsmax0, 1 / bbb * cf) (:161-173) and the respiration rate lmr_z only enters an, which the wrapper does not keep.
  if (!day) return T;
  const double fac = (1.0 - (TFRZ + 25.0) / t_veg);
  if (c3flag) {
    T.ft_lmr = psn_ft_c(R[PFT_lmrha_c], fac);
    T.fth_lmr = psn_fth(t_veg, R[PFT_lmrhd], R[PFT_lmrse], R[PFT_lmrc]);
  } else {
    T.p2 = elmk_pow(2.0, ((t_veg - (TFRZ + 25.0)) / 10.0));
    T.e_lmr_c4 = elmk_exp(1.3 * (t_veg - (TFRZ + 55.0)));
  }
  {
    if (c3flag) {  // (kp_z = kp25 * 2^((t-25)/10) is only ever read by the C4 branch of ci_func)
      T.ft_vcmax = psn_ft_c(R[PFT_vcmaxha_c], fac);
      T.fth_vcmax = psn_fth(t_veg, R[PFT_vcmaxhd], vcmaxse, vcmaxc);
    } else {
      T.e_vc4a = elmk_exp(0.2 * ((TFRZ + 15.0) - t_veg));
      T.e_vc4b = elmk_exp(0.3 * (t_veg - (TFRZ + 40.0)));
    }
    T.ft_jmax = psn_ft_c(R[PFT_jmaxha_c], fac);
    T.fth_jmax = psn_fth(t_veg, R[PFT_jmaxhd], jmaxse, jmaxc);
    T.ft_tpu = psn_ft_c(R[PFT_tpuha_c], fac);
    T.fth_tpu = psn_fth(t_veg, R[PFT_tpuhd], vcmaxse, tpuc);  // tpuse = vcmaxse (:48)
    T.kc = kc25 * psn_ft_c(R[PFT_kcha_c], fac);
    T.ko = ko25 * psn_ft_c(R[PFT_koha_c], fac);
    T.cp = cp25 * psn_ft_c(R[PFT_cpha_c], fac);
  }
  return T;
}

// photosynthesis() for one phase (sunlit or shaded), nlevcan == 1 (:63-282), in two steps so that the trip's temperature
// factors (PsnTemp, 15 doubles) are consumed for BOTH phases before the first root find starts and do not sit in registers
// across it: psn_phase_inputs scales the rates of one phase (:89-127, :203-206), psn_phase_solve is the rest.
struct PsnPhaseIn {
  double lmr_z, vcmax_z, jmax_z, tpu_z, kp_z;
};
__device__ __forceinline__ PsnPhaseIn psn_phase_inputs(const PsnCol& I, const PsnTemp& T, double btran, double vcmaxcint,
                                                       double par_z)
{
  PsnPhaseIn q;
  const double nscaler = vcmaxcint;
  const double lmr25top = I.c3flag ? I.vcmax25top * 0.015 : I.vcmax25top * 0.025;  // (:56-61)
  const double lmr25 = lmr25top * nscaler;
  if (I.c3flag) {
    q.lmr_z = lmr25 * T.ft_lmr * T.fth_lmr;
  } else {
    q.lmr_z = lmr25 * T.p2;
    q.lmr_z /= (1.0 + T.e_lmr_c4);
  }
  if (par_z <= 0.0) {
    q.vcmax_z = 0.0;
    q.jmax_z = 0.0;
    q.tpu_z = 0.0;
    q.kp_z = 0.0;
  } else {
    const double vcmax25 = I.vcmax25top * nscaler;
    const double jmax25 = I.jmax25top * nscaler;
    const double tpu25 = (0.167 * I.vcmax25top) * nscaler;   // tpu25top (:42)
    const double kp25 = (20000.0 * I.vcmax25top) * nscaler;  // kp25top (:43)
    q.vcmax_z = vcmax25 * T.ft_vcmax * T.fth_vcmax;  // overwritten for C4 just below, as in the reference (:115-123)
    q.jmax_z = jmax25 * T.ft_jmax * T.fth_jmax;
    q.tpu_z = tpu25 * T.ft_tpu * T.fth_tpu;
    if (!I.c3flag) {
      q.vcmax_z = vcmax25 * T.p2;
      q.vcmax_z /= (1.0 + T.e_vc4a);
      q.vcmax_z /= (1.0 + T.e_vc4b);
    }
    q.kp_z = kp25 * T.p2;
  }
  q.vcmax_z *= btran;
  q.lmr_z *= btran;
  return q;
}

// what the solve of a phase reads of the column and the trip (everything else comes through PsnPhaseIn)
struct PsnSolveIn {
  bool c3flag;
  double cf, qe, theta_cj, bbbopt, mbbopt;  // column / plant type
  double cp, kc, ko;                        // trip (PsnTemp)
};
__device__ __forceinline__ double psn_phase_solve(const PsnSolveIn& I, const PsnPhaseIn& q, int nrad, double forc_pbot,
                                                  double esat_tv, double eair, double oair, double cair, double rb,
                                                  double btran, double par_z, double lai_z, uint32_t& err
#if CF_PROBE >= 4
                                                  , uint32_t* pr_wev, uint32_t& pr_nev
#endif
)
{
  if (nrad <= 0) return 0.0;  // laican == 0 -> rs = 0 (:266-281)
  const double fnps = 0.15;
  const double theta_psii = 0.7;
  const double gb = 1.0 / rb;
  const double gb_mol = gb * I.cf;
  const double bbb = dmax(I.bbbopt * btran, 1.0);
  const double rsmax0 = 2.0e4;
  double rs_z;
  if (par_z <= 0.0) {
    rs_z = dmin(rsmax0, 1.0 / bbb * I.cf);
  } else {
    const double ceair = dmin(eair, esat_tv);
    const double rh_can = ceair / esat_tv;
    const double qabs = 0.5 * (1.0 - fnps) * par_z * 4.6;
    double r1, r2;
    psn_quadratic(theta_psii, -(qabs + q.jmax_z), qabs * q.jmax_z, r1, r2, err);
    const double je = dmin(r1, r2);
    const double ci0 = I.c3flag ? 0.7 * cair : 0.4 * cair;
    CiCtx k;
    k.gb_mol = gb_mol;
    k.je = je;
    k.cair = cair;
    k.oair = oair;
    k.lmr_z = q.lmr_z;
    k.par_z = par_z;
    k.rh_can = rh_can;
    k.vcmax_z = q.vcmax_z;
    k.forc_pbot = forc_pbot;
    k.cp = I.cp;
    k.kc = I.kc;
    k.ko = I.ko;
    k.qe = I.qe;
    k.tpu_z = q.tpu_z;
    k.kp_z = q.kp_z;
    k.theta_cj = I.theta_cj;
    k.bbb = bbb;
    k.mbb = I.mbbopt;
    k.c3flag = I.c3flag;
    k.kc_o = k.kc * (1.0 + k.oair / k.ko);
    k.r14_gb = 1.4 / k.gb_mol;
    k.gs_mol = 0.0;
    k.ac = k.aj = k.ap = k.ag = k.an = 0.0;
    k.err = 0;
#if CF_PROBE >= 4
    k.nev = 0u;
    k.wev = pr_wev;
#endif
    psn_hybrid(ci0, k);
#if CF_PROBE >= 4
    pr_nev += k.nev;
#endif
    err |= k.err;
    double gs_mol = k.gs_mol;
    const double an = k.an;
    if (an < 0.0) gs_mol = bbb;
    double cs = cair - k.r14_gb * an * forc_pbot;
    cs = dmax(cs, 1.0e-6);
    const double gs = gs_mol / I.cf;
    rs_z = dmin(1.0 / gs, rsmax0);
    if (gs_mol < 0.0) err |= ELMK_ERR_PSN_NEG_GS;
    const double hs = (gb_mol * ceair + gs_mol * esat_tv) / ((gb_mol + gs_mol) * esat_tv);
    const double gs_mol_err = I.mbbopt * dmax(an, 0.0) * hs / cs * forc_pbot + bbb;
    if (fabs(gs_mol - gs_mol_err) > 1.0e-01) err |= ELMK_WARN_PSN_BALL_BERRY;
  }
  // canopy sums over the single layer (:255-281)
  double laican = 0.0, gscan = 0.0;
  gscan += lai_z / (rb + rs_z);
  laican += lai_z;
  if (laican > 0.0) return laican / gscan - rb;
  return 0.0;
}

// photosynthesis() of a phase without light (par_z <= 0, :161-173) or without a canopy layer: psn_phase_solve's branch of that
// case on its own - the same operations - for the lanes whose column has no light in EITHER phase (they run no root find)
__device__ __forceinline__ double psn_phase_dark(const PsnSolveIn& I, int nrad, double rb, double btran, double lai_z)
{
  if (nrad <= 0) return 0.0;  // laican == 0 -> rs = 0 (:266-281)
  const double bbb = dmax(I.bbbopt * btran, 1.0);
  const double rsmax0 = 2.0e4;
  const double rs_z = dmin(rsmax0, 1.0 / bbb * I.cf);
  double laican = 0.0, gscan = 0.0;
  gscan += lai_z / (rb + rs_z);
  laican += lai_z;
  if (laican > 0.0) return laican / gscan - rb;
  return 0.0;
}

// lane index of the r-th (0-based) set bit of a wave-uniform mask; r < popcount(m)
__device__ __forceinline__ int nth_set_bit(unsigned long long m, int r)
{
  uint32_t w = (uint32_t)m;
  int idx = 0;
  int c = __popc(w);
  if (r >= c) {
    r -= c;
    w = (uint32_t)(m >> 32);
    idx = 32;
  }
#pragma unroll
  for (int width = 16; width >= 1; width >>= 1) {
    const uint32_t lowmask = (1u << width) - 1u;
    c = __popc(w & lowmask);
    if (r >= c) {
      r -= c;
      w >>= width;
      idx += width;
    }
    w &= lowmask;
  }
  return idx;
}
#ifndef CF_PAIR_PHASES
#define CF_PAIR_PHASES 1  // 0: every lane runs both solves of its column itself, whatever the wave holds (development A/B)
#endif

// =====================================================================================================
// Launch structure.  The leaf-temperature iteration has data-dependent trip counts (3..41) and is fp64-compute
// bound; everything around it is streaming work.  So the wrapper is four launches:
//
//   k_cf_count   per column: scheduling class (bin of the previous call's trip count x day/night); every workgroup
//                takes a slice of each class with one atomic.  Classes are laid end to end in ONE work queue that
//                is sorted longest-expected-first (LPT keeps the tail of the persistent kernel short)
//   k_cf_init    coalesced, one thread per column: the bare-ground branch; for vegetated columns initialize_flux
//                (:129-182) and every loop-invariant quantity of the iteration, written as a RECORD at the column's
//                queue position (SoA by position: a refill of consecutive positions is a coalesced read)
//   k_cf_iterate persistent waves; each lane carries one column through stability_iteration (:187-452) and, when
//                it converges, stores the converged state at the queue position and takes the next position
//   k_cf_finish  coalesced, one thread per column: compute_flux (:456-540), the 2 m profile, state writes
// =====================================================================================================
#ifndef CF_REFILL_MIN_N
#define CF_REFILL_MIN_N 8  // (12 / 16 / 24 measured in round 4: profiles/r04_cf_refill_min_ab.txt)
#endif
constexpr int CF_REFILL_MIN = CF_REFILL_MIN_N;
#ifndef CF_PRIO_LEVEL
#define CF_PRIO_LEVEL 2  // s_setprio of a wave that carries a column past CF_PRIO_TRIPS trips (3 measured: profiles/r04_cf_phase_pairing_ab.txt)
#endif
#ifndef CF_PRIO_TRIPS
#define CF_PRIO_TRIPS 10  // trips after which a column makes its wave a priority wave (k_cf_iterate)
#endif
constexpr int CF_BLOCK_EXTRA = 24;  // queue positions a wave claims beyond what a refill needs (its private block)
// Measured and dropped in round 3 (profiles/r03_cf_probe_two_*_queue_tier*.txt): separating day from night waves.  With one
// head, day before night, every wave crosses from day to night columns mid-kernel and runs a dozen trips at the price of a day
// trip for the few day lanes it still has (24 % of the lanes of all day-priced wave-trips are night or idle lanes).  Two
// variants kept the kinds apart - a two-ended queue (day from the front, night from the back) and two parts with a head each,
// waves split by estimated work - and both did what they were built for (day-priced wave-trips 56 % -> 44-47 % of all
// wave-trips, active lanes in the root finds 73 % -> 90 %), but they postpone the short day classes to the end of the kernel,
// and that is where the few columns sit that jump from a handful of trips to the 41-trip limit from one call to the next: each
// of them then runs 41 trips past the end of the queue.  Tier A gained 1.4 % of the step, tier B lost 2-3 %.
// Day-first keeps every day column in the first 60 % of the kernel, which bounds that tail.

// doubles of a queue record (k_cf_init -> k_cf_iterate): the column's inputs of the iteration that are not recomputed
// from other record fields at a few instructions each when the column is loaded (w_lai, cf, cp25, the t10 terms, qsat of
// the start temperature) or per plant functional type at kernel start (PFT_* below).  Operands that only ever enter a trip
// combined are stored combined - same operations on the same operands, done by the writer: zl_x = hgt_x - displa of the three
// profiles (canopy_fluxes_impl.hh:235-238; zl_q repeats zl_t's subtraction when the two heights are equal, and that they are
// travels as bit 8 of the record's frac_veg_nosno word), rad_in = sabv + air and lw_term = cir * lw_grnd of :391-392 / :399-401.
#define CF_REC_FIELDS(X)                                                                                               \
  X(forc_pbot) X(forc_q) X(forc_th) X(forc_rho) X(thm) X(thv) X(elai) X(esai) X(qg) X(t_grnd) X(z0mg) X(z0mv)          \
  X(zl_u) X(zl_t) X(zl_q) X(ur) X(htop) X(fwet) X(fdry) X(laisun) X(laisha) X(rdl_num) X(soilbeta)                     \
  X(rad_in) X(h2ocan) X(bir) X(lw_term) X(vcmaxcintsun) X(vcmaxcintsha) X(parsun) X(parsha) X(lai_sun_z)               \
  X(lai_sha_z) X(t10) X(vcmaxc) X(jmaxc) X(tpuc) X(t_veg) X(btran) X(um) X(obu)                                       \
  X(forc_po2) X(forc_pco2) /* only written and read by the L2-level entry elmk_canopy_fluxes_given */
// doubles of a finish record (k_cf_iterate -> k_cf_finish)
#define CF_FIN_FIELDS(X)                                                                                               \
  X(t_veg) X(btran) X(qflx_tran_veg) X(qflx_evap_veg) X(eflx_sh_veg) X(wtg) X(wtl0) X(wta0) X(wtal) X(wtgq) X(wtalq)   \
  X(wtlq0) X(wtaq0) X(delq) X(qsatl) X(temp1) X(temp2) X(dth) X(dqh) X(tlbef) X(dt_veg) X(obu_trip) X(trips) X(err)

struct CfRec {
#define X(n) double n;
  CF_REC_FIELDS(X)
#undef X
};
enum : int {
#define X(n) REC_##n,
  CF_REC_FIELDS(X)
#undef X
  REC_COUNT
};
struct CfFin {
#define X(n) double n;
  CF_FIN_FIELDS(X)
#undef X
};
enum : int {
#define X(n) FIN_##n,
  CF_FIN_FIELDS(X)
#undef X
  FIN_COUNT
};
// Records are stored in blocks of 8 consecutive queue positions: block b holds field k of positions 8b..8b+7 at
// [b][k][0..7] (64 contiguous bytes), so all fields of neighbouring positions share a few DRAM pages, a refill batch
// of consecutive positions reads 64-byte runs, and the writers' partial runs merge in L2.
// CF_REC_AOS / CF_FIN_AOS (profiles/r04_record_layout_ab.txt): a record as CF_*_N consecutive doubles at its position - every
// lane reads or writes ONE contiguous run (16-byte accesses), whatever positions its neighbours hold.  The FINISH record is
// written by single lanes as they converge, trips apart from their neighbours: in blocks of 8 positions its 8-byte stores
// left half-written 64-byte runs behind (PMC: 403 bytes written per column for 192 of payload); per position the kernel writes
// 199 and the step moves 286 bytes per column less at the same time (k_cf_iterate -0.5 %, k_cf_finish +0.5 %): the product.
// The INPUT record is written by k_cf_init / k_fz_stream, whose neighbouring threads hold neighbouring positions: per position
// every wave store would go to 64 different lines (k_cf_init x 2) - it stays in blocks of 8.
#ifndef CF_REC_AOS
#define CF_REC_AOS 0
#endif
#ifndef CF_FIN_AOS
#define CF_FIN_AOS 1
#endif
#if CF_REC_AOS
#define CF_REC_BASE(pos) ((pos) * (int64_t)CF_REC_N)
#define CF_REC_K(k) (k)
#else
#define CF_REC_BASE(pos) (((pos) >> 3) * (int64_t)(CF_REC_N * 8) + ((pos)&7))
#define CF_REC_K(k) ((k)*8)
#endif
#if CF_FIN_AOS
#define CF_FIN_BASE(pos) ((pos) * (int64_t)CF_FIN_N)
#define CF_FIN_K(k) (k)
#else
#define CF_FIN_BASE(pos) (((pos) >> 3) * (int64_t)(CF_FIN_N * 8) + ((pos)&7))
#define CF_FIN_K(k) ((k)*8)
#endif
static_assert(REC_COUNT == CF_REC_N && FIN_COUNT == CF_FIN_N, "record sizes in elmk_dev.h out of date");
enum : int { IREC_vtype = 0, IREC_nrad, IREC_fvn };
constexpr int IREC_FVN_SAME_TQ = 1 << 8;  // in the IREC_fvn word: forc_hgt_q_patch == forc_hgt_t_patch (friction_velocity_humidity's short-cut)

// class totals of the current call: counters behind the list counters, one per 128-byte line; zero on entry
// (elmk_create clears them, k_cf_finish clears them again for the next call)
#define CF_CLASS_COUNT(S, k) ((S)->counters[(2 * NLISTS + (k)) * CPAD])

// scheduling class from (day, trip hint of the previous call: 0 = never iterated -> middle bin)
__device__ __forceinline__ int cf_class_of(const bool day, const int prev)
{
  int bin = 3;
  if (prev >= 22) {
    bin = 0;
  } else if (prev >= 16) {
    bin = 1;
  } else if (prev >= 12) {
    bin = 2;
  } else if (prev >= 9) {
    bin = 3;
  } else if (prev >= 6) {
    bin = 4;
  } else if (prev >= 1) {
    bin = 5;
  }
  // all day columns before the night ones: a day column can jump from a handful of trips to the 40-trip limit from
  // one call to the next (a few per 100000 do), a night column was never seen to; so a mispredicted long column
  // still starts in the first half of the queue, and the end of the queue is reliably short work.
  // (Separating C3 from C4 columns as well - 24 classes - was measured: no gain, the finer classes cost as much in
  // k_cf_init's scattered record writes and in queue tail as the C4 branches cost in mixed waves.)
  return (day ? 0 : CF_NCLS / 2) + bin;
}
// scheduling class of a column, -1 if it is not vegetated (lake land units are handled by the callers)
__device__ __forceinline__ int cf_class(const DevState* __restrict__ S, const Land& L, int64_t c, bool inside)
{
  if (!inside || L.urbpoi) return -1;
  if (S->frac_veg_nosno[c] == 0) return -1;
  const bool day = (S->nrad[c] > 0) && (S->parsun_z[c] > 0.0 || S->parsha_z[c] > 0.0);
  return cf_class_of(day, S->cf_niter[c] >> 16);  // scheduling hint (see k_cf_finish)
}

// One workgroup counts CF_COUNT_TILES tiles of 256 columns (a tile = one workgroup of k_cf_init) and takes the
// slices of all of them with ONE atomic per class: the class totals are 12 addresses, every atomic on them serialises.
#ifndef CF_COUNT_TILES_N
#define CF_COUNT_TILES_N 4
#endif
constexpr int CF_COUNT_TILES = CF_COUNT_TILES_N;
// (256 x CF_COUNT_TILES threads: the tiles of a workgroup are counted side by side, four waves each - one after the other
//  their loads were four dependent round trips)
__global__ __launch_bounds__(256 * CF_COUNT_TILES) void k_cf_count(const DevState* __restrict__ S)
{
  __shared__ uint32_t s_cnt[CF_COUNT_TILES][CF_NCLS];
  for (int i = threadIdx.x; i < CF_COUNT_TILES * CF_NCLS; i += blockDim.x) (&s_cnt[0][0])[i] = 0u;
  __syncthreads();
  const Land L = S->land;
  const int lane = threadIdx.x & 63;
  const int64_t tile0 = (int64_t)blockIdx.x * CF_COUNT_TILES;
  {
    const int t = (int)(threadIdx.x >> 8);
    const int64_t c = (tile0 + t) * 256 + (threadIdx.x & 255);
    const int cls = L.lakpoi ? -1 : cf_class(S, L, c, c < S->ncols);
#pragma unroll
    for (int k = 0; k < CF_NCLS; k++) {
      const unsigned long long m = __ballot(cls == k);
      if (lane == 0 && m) atomicAdd(&s_cnt[t][k], (uint32_t)__popcll(m));
    }
  }
  __syncthreads();
  // tile t's slice of class k: [value stored, + its count) relative to the start of the class
  if (threadIdx.x < CF_NCLS) {
    const int k = threadIdx.x;
    uint32_t n = 0u;
    for (int t = 0; t < CF_COUNT_TILES; t++) n += s_cnt[t][k];
    uint32_t base = n ? atomicAdd(ELMK_GENERIC(&CF_CLASS_COUNT(S, k)), n) : 0u;
    for (int t = 0; t < CF_COUNT_TILES; t++) {
      if (tile0 + t < S->cf_nblk) S->cf_blk[(int64_t)k * S->cf_nblk + tile0 + t] = base;
      base += s_cnt[t][k];
    }
  }
}

// Queue position of this thread's column: slice of (class, workgroup) from the count kernel (k_cf_count, or k_fz_prep in
// the fused step), then (wave, lane) order inside the workgroup.  All 256 threads of the workgroup call it (one barrier);
// cls < 0: not vegetated -> -1.  Thread 0 of workgroup 0 publishes the queue length.
__device__ __forceinline__ int64_t cf_queue_position(const DevState* __restrict__ S, const int cls, const int64_t tile = -1)
{
  const int64_t blk = tile >= 0 ? tile : (int64_t)blockIdx.x;  // the 256-column tile this workgroup is placing
  __shared__ uint32_t s_w[4][CF_NCLS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t rank = 0u;
#pragma unroll
  for (int k = 0; k < CF_NCLS; k++) {
    const unsigned long long m = __ballot(cls == k);
    if (lane == 0) s_w[wave][k] = (uint32_t)__popcll(m);
    if (cls == k) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  int64_t pos = -1;
  // start of each class in the queue = total of the classes before it (final: the count kernel has completed)
  uint32_t start = 0u, mine = 0u;
#pragma unroll
  for (int k = 0; k < CF_NCLS; k++) {
    if (k == cls) mine = start;
    start += CF_CLASS_COUNT(S, k);
  }
  if (blk == 0 && threadIdx.x == 0) {
    ELMK_LIST_COUNT(S, LIST_CF_QUEUE) = start;
    ELMK_LIST_HEAD(S, LIST_CF_QUEUE) = 0u;
  }
  if (cls >= 0) {
    uint32_t off = mine + S->cf_blk[(int64_t)cls * S->cf_nblk + blk];
    for (int w = 0; w < wave; w++) off += s_w[w][cls];
    pos = (int64_t)off + rank;
  }
  return pos;
}

// canopy_fluxes for one column up to the iteration: the bare branch, or initialize_flux (canopy_fluxes_impl.hh:95-184) and the
// queue record.  One source for k_cf_init (inputs from the state) and the fused streaming stage (FUSED: the values the
// earlier bodies of the same pass produced come through ColFwd).  pos: the column's queue position (-1: not vegetated).
// given: bit 0 / 1 / 2 = forc_rho / forc_po2 / forc_pco2 come from S->cf_given (elmk_canopy_fluxes_given) instead of being
// derived from the forcing as the wrapper does (canopy_fluxes_kokkos.cc:47-49)
// soil_moist_stress (soil_moist_stress_impl.hh:62-133) of one vegetated column: effective porosity, liquid volume and root
// moisture stress of the 15 soil levels -> eff_porosity, rootr, btran (state and queue record).  liq0: the liquid water of the
// top soil level as canopy_fluxes sees it (in the fused step's early kernel: what canopy_hydrology is about to leave there).
__device__ __forceinline__ void cf_root_stress_col(const DevState* __restrict__ S, const int64_t c, const int64_t ld, const int64_t pos,
                                                   const double liq0)
{
  const int vtype = S->vtype[c];
  const double* __restrict__ P = S->pft_psn[vtype];
  const double tc_stress = P[P_tc_stress], smpso = P[P_smpso], smpsc = P[P_smpsc];
  double btran = 0.0;  // btran0
  double rootr[NLEVGRND];
  // (Loading a level's inputs one level ahead of their use was measured: no gain - the kernel is bound by bandwidth, not by
  // latency - and it made the three loads of the unfrozen branch unconditional: +230 bytes per column.  They stay conditional.)
#pragma unroll
  for (int i = 0; i < NLEVGRND; i++) {
    const double watsat = LV(watsat, i);
    const double dzi = LV(dz, NLEVSNO + i);
    // calc_effective_soilporosity (soil_moist_stress_impl.hh:62-73)
    const double vol_ice = dmin(watsat, (LV(h2osoi_ice, NLEVSNO + i) / (DENICE * dzi)));
    const double eff_por = watsat - vol_ice;
    LV(eff_porosity, i) = eff_por;
    // calc_volumetric_h2oliq (:77-86)
    const double liq = (i == 0) ? liq0 : (double)LV(h2osoi_liq, NLEVSNO + i);
    const double liqvol = dmin(eff_por, (liq / (dzi * DENH2O)));
    // calc_root_moist_stress (:89-133), perchroot == perchroot_alt == 0
    const double tsoi = LV(t_soisno, NLEVSNO + i);
    if (liqvol <= 0.0 || tsoi <= TFRZ + tc_stress) {
      rootr[i] = 0.0;
    } else {
      const double s_node = dmax(liqvol / eff_por, 0.01);
      double smp_node = -LV(sucsat, i) * elmk_pow(s_node, (-LV(bsw, i)));
      smp_node = dmax(smpsc, smp_node);
      const double rresis = dmin((eff_por / watsat) * (smp_node - smpsc) / (smpso - smpsc), 1.0);
      rootr[i] = LV(rootfr, i) * rresis;
      btran += dmax(rootr[i], 0.0);
    }
  }
#pragma unroll
  for (int i = 0; i < NLEVGRND; i++) {
    double q = rootr[i];
    if (btran > 0.0) {
      q /= btran;
    } else {
      q = 0.0;
    }
    LV(rootr, i) = q;
  }
  S->btran[c] = btran;
  sc_st<4>(S->cf_rec + CF_REC_BASE(pos) + CF_REC_K(REC_btran), btran);
}

// ROOT_DONE: the fused step's early kernel (k_fz_pre) has already done cf_root_stress_col and the bare branch's rootr / btran
template <bool FUSED, bool ROOT_DONE = false>
__device__ __forceinline__ void cf_init_col(const DevState* __restrict__ S, const int64_t c, const int64_t ld, const Land& L,
                                            const int64_t pos, const ColFwd& w, const int given = 0)
{
  const bool inside = true, veg = pos >= 0;
  if (inside && !veg) {
    S->cf_niter[c] &= (int32_t)0xFFFF0000;  // trips of this call: 0 (not vegetated); the scheduling hint stays
    if (!L.urbpoi) {
      S->t_veg[c] = FW(forc_tbot, S->forc_tbot[c]);
      if (!ROOT_DONE) {
        S->btran[c] = 0.0;
#pragma unroll
        for (int i = 0; i < NLEVGRND; i++) LV(rootr, i) = 0.0;
      }
    }
    // (the cgrnd* = 0 of this branch, compute_flux :470-474, is written by k_cf_finish: in the fused step this body runs
    // BEFORE the bare-ground flux list, whose cgrnd* of the same columns the reference's call order overwrites)
  }
  if (!veg) return;

  // record fields are stored as soon as they are final (PUT), so few of them are live at any time
  CfRec r;
  const gptr<double> rec = S->cf_rec + CF_REC_BASE(pos);
#define PUT(n) sc_st<4>(rec + CF_REC_K(REC_##n), r.n);
  const int snl = FW(snl, S->snl[c]);
  const int vtype = S->vtype[c];
  const double* __restrict__ P = S->pft_psn[vtype];
  if (!ROOT_DONE) cf_root_stress_col(S, c, ld, pos, LV(h2osoi_liq, NLEVSNO));
  const double t_soi0 = LV(t_soisno, NLEVSNO);
  const double t_top = (snl > 0) ? LV(t_soisno, NLEVSNO - snl) : t_soi0;

  // canopy roughness blend (canopy_fluxes_impl.hh:141-147)
  r.elai = FW(elai, S->elai[c]);
  PUT(elai)
  r.esai = FW(esai, S->esai[c]);
  PUT(esai)
  r.z0mg = FW(z0mg, S->z0mg[c]);
  PUT(z0mg)
  const double tlsai_crit = 2.0;
  const double lt = dmin(r.elai + r.esai, tlsai_crit);
  const double egvf = (1.0 - elmk_exp(-lt)) / (1.0 - elmk_exp(-tlsai_crit));
  double displa = FW(displa, S->displa[c]);
  displa *= egvf;
  double z0mv = FW(z0m, S->z0mv[c]);
  z0mv = elmk_exp(egvf * elmk_log(z0mv) + (1.0 - egvf) * elmk_log(r.z0mg));
  S->displa[c] = displa;
  S->z0mv[c] = z0mv;
  S->z0hv[c] = z0mv;
  S->z0qv[c] = z0mv;
  r.z0mv = z0mv;
  PUT(z0mv)

  // forcing, derived forcing (atm_physics_impl.hh:246-272) and the wrapper-level inputs of the iteration
  r.forc_pbot = FW(forc_pbot, S->forc_pbot[c]);
  PUT(forc_pbot)
  r.forc_q = FW(forc_q, S->forc_qbot[c]);
  PUT(forc_q)
  r.forc_th = FW(forc_th, S->forc_thbot[c]);
  PUT(forc_th)
  r.forc_rho = derive_forc_rho(r.forc_pbot, r.forc_q, FW(forc_tbot, S->forc_tbot[c]));
  if (given & 1) r.forc_rho = S->cf_given[c];
  PUT(forc_rho)
  if (given & 6) {
    r.forc_po2 = (given & 2) ? S->cf_given[ld + c] : derive_forc_po2(r.forc_pbot);
    r.forc_pco2 = (given & 4) ? S->cf_given[2 * ld + c] : derive_forc_pco2(r.forc_pbot);
    PUT(forc_po2) PUT(forc_pco2)
  }
  r.thm = FW(thm, S->thm[c]);
  PUT(thm)
  r.thv = FW(thv, S->thv[c]);
  PUT(thv)
  r.qg = FW(qg, S->qg[c]);
  PUT(qg)
  r.t_grnd = FW(t_grnd, S->t_grnd[c]);
  PUT(t_grnd)
  const double hgt_u = FW(hgt_u, S->forc_hgt_u_patch[c]), hgt_t = FW(hgt_t, S->forc_hgt_t_patch[c]), hgt_q = FW(hgt_q, S->forc_hgt_q_patch[c]);
  const bool same_tq = (hgt_q == hgt_t);
  r.zl_u = hgt_u - displa;
  r.zl_t = hgt_t - displa;
  r.zl_q = same_tq ? (hgt_t - displa) : (hgt_q - displa);
  PUT(zl_u) PUT(zl_t) PUT(zl_q)
  const double forc_u = S->forc_u[c], forc_v = S->forc_v[c];
  r.ur = dmax(1.0, sqrt(forc_u * forc_u + forc_v * forc_v));
  PUT(ur)
  r.htop = FW(htop, S->htop[c]);
  PUT(htop)
  r.fwet = S->fwet[c];
  PUT(fwet)
  r.fdry = S->fdry[c];
  PUT(fdry)
  r.laisun = FW(laisun, S->laisun[c]);
  PUT(laisun)
  r.laisha = FW(laisha, S->laisha[c]);
  PUT(laisha)
  r.soilbeta = FW(soilbeta, S->soilbeta[c]);
  PUT(soilbeta)
  const double sabv = FW(sabv, S->sabv[c]);
  r.h2ocan = FW(h2ocan, S->h2ocan[c]);
  PUT(h2ocan)
  const double emv = FW(emv, S->emv[c]), emg = FW(emg, S->emg[c]);
  const double air = emv * (1.0 + (1.0 - emv) * (1.0 - emg)) * S->forc_lwrad[c];  // :360-362
  r.bir = -(2.0 - emv * (1.0 - emg)) * emv * STEBOL;
  r.rad_in = sabv + air;
  PUT(rad_in) PUT(bir)
  const double cir = emv * emg * STEBOL;
  r.vcmaxcintsun = S->vcmaxcintsun[c];
  PUT(vcmaxcintsun)
  r.vcmaxcintsha = S->vcmaxcintsha[c];
  PUT(vcmaxcintsha)
  const int nrad = FW(nrad, S->nrad[c]);
  r.parsun = r.parsha = r.lai_sun_z = r.lai_sha_z = 0.0;
  if (nrad > 0) {
    r.parsun = FW(parsun_z, S->parsun_z[c]);
    r.parsha = FW(parsha_z, S->parsha_z[c]);
    r.lai_sun_z = FW(laisun_z, S->laisun_z[c]);
    r.lai_sha_z = FW(laisha_z, S->laisha_z[c]);
  }
  PUT(parsun) PUT(parsha) PUT(lai_sun_z) PUT(lai_sha_z)
  // loop-invariant sub-expression of the iteration body (:301-303)
  {
    const double snow_depth_c = 0.05;
    const double fsno_dl = FW(snow_depth, S->snow_depth[c]) / snow_depth_c;
    const double elai_dl = 0.5 * (1.0 - dmin(fsno_dl, 1.0));
    r.rdl_num = (1.0 - elmk_exp(-elai_dl));
    PUT(rdl_num)
  }

  // initial flux profile and Monin-Obukhov length (:158-181)
  {
    const double taf = (r.t_grnd + r.thm) / 2.0;
    const double qaf = (r.forc_q + r.qg) / 2.0;
    const double dth = r.thm - taf;
    const double dqh = r.forc_q - qaf;
    const double dthv = dth * (1.0 + 0.61 * r.forc_q) + 0.61 * r.forc_th * dqh;
    const double zldis = r.zl_u;
    if (!(zldis >= 0.0)) S->err_flags[c] |= ELMK_ERR_CANFLX_FORC_HGT;
    monin_obukhov_length(r.ur, r.thv, dthv, zldis, z0mv, r.um, r.obu);
    PUT(um) PUT(obu)
  }

  // ground-emitted longwave (:366-367), loop-invariant
  const double frac_sno = FW(frac_sno, S->frac_sno[c]), frac_h2osfc = FW(frac_h2osfc, S->frac_h2osfc[c]);
  const double lw_grnd = (frac_sno * elmk_pow(t_top, 4.0) + (1.0 - frac_sno - frac_h2osfc) * elmk_pow(t_soi0, 4.0) +
                          frac_h2osfc * elmk_pow(FW(t_h2osfc, S->t_h2osfc[c]), 4.0));
  r.lw_term = cir * lw_grnd;
  PUT(lw_term)
  S->wk[(int64_t)WK_CF_LWGRND * ld + c] = lw_grnd;

  // iteration-invariant part of photosynthesis() that needs an exp per column (photosynthesis_impl.hh:46-55); the rest is
  // recomputed by k_cf_iterate when it loads the column (cf_psn_column) or once per plant functional type (cf_pft_row)
  {
    r.t10 = S->t10[c];
    PUT(t10)
    const double vcmaxse = 668.39 - 1.07 * dmin(dmax((r.t10 - TFRZ), 11.0), 35.0);
    const double jmaxse = 659.70 - 0.75 * dmin(dmax((r.t10 - TFRZ), 11.0), 35.0);
    r.vcmaxc = psn_fth25(P[P_vcmaxhd], vcmaxse);
    PUT(vcmaxc)
    r.jmaxc = psn_fth25(P[P_jmaxhd], jmaxse);
    PUT(jmaxc)
    r.tpuc = psn_fth25(P[P_tpuhd], vcmaxse);
    PUT(tpuc)
  }

  // iteration start value (canopy_fluxes_impl.hh:154-166 and :203-215)
  r.t_veg = S->t_veg[c];
  PUT(t_veg)

#undef PUT
  const gptr<int32_t> irec = S->cf_irec + pos;
  irec[(int64_t)IREC_vtype * ld] = vtype;
  irec[(int64_t)IREC_nrad * ld] = nrad;
  irec[(int64_t)IREC_fvn * ld] = FW(fvn, S->frac_veg_nosno[c]) | (same_tq ? IREC_FVN_SAME_TQ : 0);
}

__global__ __launch_bounds__(256) void k_cf_init(const DevState* __restrict__ S, const int given)
{
  elmk_math_lds_init<false>();
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ld = S->ld;
  const Land L = S->land;
  if (L.lakpoi) {  // uniform: the wrapper does nothing on lake land units
    if (blockIdx.x == 0 && threadIdx.x == 0) ELMK_LIST_COUNT(S, LIST_CF_QUEUE) = 0u;
    return;
  }
  const bool inside = c < S->ncols;
  const int64_t pos = cf_queue_position(S, cf_class(S, L, c, inside));
  if (!inside) return;
  S->cf_pos[c] = (int32_t)pos;
  ColFwd w;
  cf_init_col<false>(S, c, ld, L, pos, w, given);
}

// =====================================================================================================
// k_cf_iterate - persistent waves drain the queue.  Every lane carries one column through the leaf-temperature
// iteration; a lane whose column has converged stores the converged state at its queue position and takes the next
// position, so a wave never waits for its slowest column (trip counts range from 3 to 41).  Refill is batched
// (CF_REFILL_MIN idle lanes, or nothing left to do): the lanes of a batch take consecutive positions, so the record
// loads are coalesced.  Night columns ride in the idle lanes of waves busy with day columns: their trip is a subset.
//
// Two waves per SIMD.  A trip is long dependent fp64 chains (Horner forms, Newton steps of divisions and square roots,
// table look-ups in LDS): one wave alone keeps the fp64 pipe about half busy, a second one fills the gaps.  That needs
// the kernel inside 256 registers per lane, and a column brings ~45 constants with it - 90 registers that are each read
// once or twice per trip.  So the column's constants live in LDS (CF_LDS_FIELDS: one 8-byte slot per lane and field,
// [field][lane]: a wave reads 512 contiguous bytes, conflict-free) and only the few that every section of a trip uses
// stay in registers (CF_REG_FIELDS); what depends on the plant type alone is one LDS row per type (cf_pft_row).  Moving a
// field between the two lists changes nothing else: every access goes through C(name).
// =====================================================================================================
#define CF_REG_FIELDS(X) X(forc_pbot) X(t_grnd) X(thm) X(forc_q) X(qg) X(elai_esai) X(forc_rho)
#define CF_LDS_FIELDS(X)                                                                                               \
  X(forc_th) X(thv) X(elai) X(z0mg) X(z0mv) X(zl_u) X(zl_t) X(zl_q) X(ur) X(htop) X(fwet) X(fdry) X(laisun) X(laisha)  \
  X(rdl_num) X(soilbeta) X(rad_in) X(h2ocan_dt) X(bir) X(lw_term) X(vcmaxcintsun) X(vcmaxcintsha) X(parsun) X(parsha)  \
  X(lai_sun_z) X(lai_sha_z) X(w_lai) X(tc10) X(vcmaxc) X(jmaxc) X(tpuc) X(cf) X(cp25)
struct CfRegs {
#define X(n) double n;
  CF_REG_FIELDS(X)
#undef X
};
enum : int {
#define X(n) CL_##n,
  CF_LDS_FIELDS(X)
#undef X
  CF_NLDS
};
#ifndef CF_ITER_THREADS_N
#define CF_ITER_THREADS_N 512  // (256: one wave per SIMD, half the LDS - development, tests/tools/two_ctx_overlap.py)
#endif
constexpr int CF_ITER_THREADS = CF_ITER_THREADS_N;  // 8 waves = two per SIMD; one workgroup per CU shares one copy of the math tables
constexpr int CF_ITER_THREADS_HALF = 256;           // k_cf_iterate_half: one wave per SIMD, 92 KB of LDS - room for other kernels beside it
#define X(n)                                                                                              \
  template <class L> __device__ __forceinline__ double cf_get_##n(const CfRegs& R, const L* s, int t) { return R.n; } \
  template <class L> __device__ __forceinline__ void cf_set_##n(CfRegs& R, L* s, int t, double v) { R.n = v; }
CF_REG_FIELDS(X)
#undef X
#define X(n)                                                                                                       \
  template <class L> __device__ __forceinline__ double cf_get_##n(const CfRegs& R, const L* s, int t) { return s[CL_##n][t]; } \
  template <class L> __device__ __forceinline__ void cf_set_##n(CfRegs& R, L* s, int t, double v) { s[CL_##n][t] = v; }
CF_LDS_FIELDS(X)
#undef X
#define C(n) cf_get_##n(R, s_col, tid)
#define CSET(n, v) cf_set_##n(R, s_col, tid, (v))

// (the body as a template over the workgroup size: the per-lane LDS slots are [field][thread of the workgroup])
template <int THREADS>
__device__ __forceinline__ void cf_iterate_body(const DevState* __restrict__ S, double dtime, const int given)
{
  elmk_math_lds_init<true>();
  __shared__ double s_col[CF_NLDS][THREADS];
  __shared__ double s_pft[ELMK_MXPFT * PFT_STRIDE];
  __shared__ FvConst s_fv;  // (in LDS, not in registers: kernel-lifetime constants would be the first thing spilled)
  if (threadIdx.x < ELMK_MXPFT) cf_pft_row(S, threadIdx.x, s_pft + threadIdx.x * PFT_STRIDE);
  if (threadIdx.x == 64) s_fv = fv_const();
  __syncthreads();
  const int64_t ld = S->ld;
  const Land L = S->land;
  const int tid = threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool soy = (L.vtype == pft_nsoybean || L.vtype == pft_nsoybeanirrig);
  const uint32_t nq = ELMK_LIST_COUNT(S, LIST_CF_QUEUE);  // final: written by k_cf_init
  bool exhausted = false;  // wave-uniform: the queue is empty
  int64_t pos = -1;        // queue position owned by this lane (-1: idle)
  bool fresh = false;      // the lane has just received a position and must load its record
  uint32_t blk_next = 0, blk_end = 0;  // wave-uniform: the wave's private block of claimed queue positions
  uint32_t last_base = 0;              // queue head as this wave last saw it
  const uint32_t nwaves_total = gridDim.x * (blockDim.x >> 6);

  CfRegs R;
#define X(n) R.n = 0.0;
  CF_REG_FIELDS(X)
#undef X
  const double* PR = s_pft;  // the column's PFT row
  int nrad = 0, fvn = 0;
  bool day = false, c3flag = true, same_tq = true;
  uint32_t err = 0;
  // loop-carried state of stability_iteration
  double t_veg = 0.0, btran = 0.0, um = 0.0, obu = 0.0, taf = 0.0, qaf = 0.0, el = 0.0, qsatl = 0.0, qsatldT = 0.0;
  double delq = 0.0, del = 0.0, efeb = 0.0, obuold = 0.0;
  int itlef = 0, nmozsgn = 0;
#if CF_PROBE >= 4  // per-wave timeline: start, queue-exhausted and end time (100 MHz ticks), trips, active lane-trips
  const uint64_t pr_t0 = wall_clock64();
  uint64_t pr_texh = 0, pr_trips = 0, pr_lanes = 0, pr_refills = 0, pr_cols = 0, pr_brent = 0, pr_brent_trips = 0, pr_c4 = 0, pr_day = 0;
#endif
#if CF_PROBE >= 4
  __shared__ uint32_t s_wev[2][THREADS / 64];  // wave-level ci_func executions, [phase][wave]
  if (lane == 0) s_wev[0][threadIdx.x >> 6] = s_wev[1][threadIdx.x >> 6] = 0u;
  uint32_t pr_nev_sun = 0u, pr_nev_sha = 0u;  // per-lane ci_func evaluations
  uint64_t pr_daytrips = 0;                   // wave-trips with at least one day lane
#define PR_SOLVE_ARGS(ph, acc) , &s_wev[ph][threadIdx.x >> 6], acc
#else
#define PR_SOLVE_ARGS(ph, acc)
#endif
#if CF_PROBE == 5  // shader-clock cycles per section of the loop (wave-uniform accumulators)
  uint64_t pr_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t pr_last = clock64();
#define PR_T(i)                    \
  {                                \
    const uint64_t t_ = clock64(); \
    pr_acc[i] += t_ - pr_last;     \
    pr_last = t_;                  \
  }
#else
#define PR_T(i)
#endif

  for (;;) {
    // ---------------- refill ----------------
    // Idle lanes are refilled in batches (>= CF_REFILL_MIN of them, which includes the all-idle wave) with one
    // wave-aggregated atomic on the queue head.
    const int nidle = __popcll(__ballot(pos < 0));
    if (nidle >= CF_REFILL_MIN && (!exhausted || blk_next < blk_end)) {
      const bool take = (pos < 0);
      const unsigned long long m = __ballot(take);
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));  // set bits of m below this lane
      // positions come from the wave's private block first; one atomic on the queue head claims the next block
      // (what this refill still needs + CF_BLOCK_EXTRA for the refills to come): the atomic's round trip is then
      // paid by one refill in three or four instead of by every one
      const uint32_t have = blk_end - blk_next;
      uint32_t mine = 0xffffffffu;
      if (rank < have) mine = blk_next + rank;
      blk_next += ((uint32_t)nidle < have) ? (uint32_t)nidle : have;
      if ((uint32_t)nidle > have && !exhausted) {
        const uint32_t need = (uint32_t)nidle - have;
        // no hoarding near the end of the queue: with fewer than CF_BLOCK_EXTRA positions per wave left, claim exactly
        // ... and none at the head either: the first claim of every wave takes exactly its 64 positions, so that the whole
        // head of the queue - the longest classes - starts at time zero (positions held back there would start at the
        // first refill, ~22 trips in: a 41-trip column among them ended at trip 63 and set the kernel's tail, +0.5 ms on
        // the branch-mix tier)
        const uint32_t extra = (blk_end != 0u && nq - last_base > (uint32_t)CF_BLOCK_EXTRA * nwaves_total) ? (uint32_t)CF_BLOCK_EXTRA : 0u;
        const uint32_t want = need + extra;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(ELMK_GENERIC(&ELMK_LIST_HEAD(S, LIST_CF_QUEUE)), want);
        base = __shfl(base, 0, 64);
        last_base = (base < nq) ? base : nq;
        const uint32_t b1 = (base + want < nq) ? base + want : nq;
        const uint32_t b0 = (base < b1) ? base : b1;
        if (base + want >= nq) exhausted = true;
        if (take && rank >= have) {
          const uint32_t p = b0 + (rank - have);
          if (p < b1) mine = p;
        }
        blk_next = (b0 + need < b1) ? b0 + need : b1;
        blk_end = b1;
      }
      if (take && mine != 0xffffffffu) {
        pos = (int64_t)mine;
        fresh = true;
      }
#if CF_PROBE >= 4
      pr_refills++;
      if (exhausted && !pr_texh) pr_texh = wall_clock64();
#endif
    }
    if (fresh) {
      fresh = false;
      const gptr<const double> rec = S->cf_rec + CF_REC_BASE(pos);
#define LD(n) sc_ld<4>(rec + CF_REC_K(REC_##n))
      const gptr<const int32_t> irec = S->cf_irec + pos;
      const int vtype = irec[(int64_t)IREC_vtype * ld];
      nrad = irec[(int64_t)IREC_nrad * ld];
      fvn = irec[(int64_t)IREC_fvn * ld];
      same_tq = (fvn & IREC_FVN_SAME_TQ) != 0;
      fvn &= ~IREC_FVN_SAME_TQ;
      PR = s_pft + vtype * PFT_STRIDE;
      c3flag = PR[PFT_c3] != 0.0;
      const double pbot = LD(forc_pbot), thm = LD(thm), t_grnd = LD(t_grnd), forc_q = LD(forc_q), qg = LD(qg);
      const double elai = LD(elai), esai = LD(esai), t10 = LD(t10), parsun = LD(parsun), parsha = LD(parsha);
      CSET(forc_pbot, pbot);
      CSET(thm, thm);
      CSET(t_grnd, t_grnd);
      CSET(forc_q, forc_q);
      CSET(qg, qg);
      CSET(elai, elai);
      CSET(elai_esai, elai + esai);
      CSET(parsun, parsun);
      CSET(parsha, parsha);
#define CP(n) CSET(n, LD(n));
      CP(forc_th) CP(forc_rho) CP(thv) CP(z0mg) CP(z0mv) CP(ur) CP(htop) CP(fwet) CP(fdry) CP(laisun) CP(laisha) CP(rdl_num)
      CP(soilbeta) CP(bir) CP(vcmaxcintsun) CP(vcmaxcintsha) CP(lai_sun_z) CP(lai_sha_z) CP(vcmaxc) CP(jmaxc) CP(tpuc)
      // (operands that only ever enter a trip combined arrive combined: zl_x, rad_in, lw_term - see CF_REC_FIELDS; h2ocan / dtime
      // of :338 / :411-412 is formed here, the writer does not know the caller's dtime)
      CP(zl_u) CP(zl_t) CP(zl_q) CP(rad_in) CP(lw_term)
#undef CP
      CSET(h2ocan_dt, LD(h2ocan) / dtime);
      day = (nrad > 0) && (parsun > 0.0 || parsha > 0.0);
      // what k_cf_init does not hand over: a few instructions each, from the record's own fields
      CSET(w_lai, elmk_exp(-(elai + esai)));  // (canopy_fluxes_impl.hh:272)
      {
        // iteration-invariant part of photosynthesis() (photosynthesis_impl.hh:35-37, :46-50, :91, :152-154)
        CSET(tc10, dmin(dmax((t10 - TFRZ), 11.0), 35.0));
        CSET(cf, pbot / (RGAS * 1.0e-3 * thm) * 1.e06);
        const double sco = 0.5 * 0.209 / (42.75 / 1.e06);
        const double po2 = (given & 6) ? rec[CF_REC_K(REC_forc_po2)] : derive_forc_po2(pbot);
        CSET(cp25, 0.5 * po2 / sco);
      }
      // iteration start values (canopy_fluxes_impl.hh:154-166 and :203-215)
      btran = LD(btran);
      t_veg = LD(t_veg);
      um = LD(um);
      obu = LD(obu);
#undef LD
      {
        double deldT;
        qsat(t_veg, pbot, el, deldT, qsatl, qsatldT);
      }
      taf = (t_grnd + thm) / 2.0;
      qaf = (forc_q + qg) / 2.0;
      delq = qg - qaf;
      del = 0.0;
      efeb = 0.0;
      obuold = 0.0;
      itlef = 0;
      nmozsgn = 0;
      err = 0;
    }
    if (__ballot(pos >= 0) == 0ull) {
      if (exhausted && blk_next >= blk_end) break;
      continue;
    }
    PR_T(0)
#if CF_PROBE >= 4
    pr_trips++;
    pr_lanes += (uint64_t)__popcll(__ballot(pos >= 0));
#endif

    // A wave that carries a column deep into its iteration is on the kernel's critical path (the slowest columns take the
    // 41-trip limit, and often Brent's method in every trip: a millisecond of dependent work on their own): it gets issue
    // priority over the wave it shares the SIMD with, which then fills the gaps instead of competing for the slots.
    if (__ballot(pos >= 0 && itlef >= CF_PRIO_TRIPS) != 0ull) {
      __builtin_amdgcn_s_setprio(CF_PRIO_LEVEL);
    } else {
      __builtin_amdgcn_s_setprio(0);
    }

    // ---------------- one trip of the leaf-temperature iteration (:233-450) ----------------
    // The trip is in three parts: up to the inputs of the two photosynthesis root finds (lanes with a column), the root finds
    // (below: ALL lanes of the wave may take part), the energy balance and the stop test (lanes with a column).  What crosses
    // the parts is declared here.
    double ustar = 0.0, temp1 = 0.0, temp2 = 0.0, obu_trip = 0.0, zldis = 0.0, tlbef = 0.0, del2 = 0.0, rah0 = 0.0, raw0 = 0.0,
           uaf = 0.0, rb = 0.0, rah1 = 0.0, raw1 = 0.0, svpts = 0.0, eah = 0.0;
    PsnPhaseIn qsun, qsha;
    qsun.lmr_z = qsun.vcmax_z = qsun.jmax_z = qsun.tpu_z = qsun.kp_z = 0.0;
    qsha = qsun;
    PsnSolveIn J;
    J.c3flag = true;
    J.cf = J.qe = J.theta_cj = J.bbbopt = J.mbbopt = J.cp = J.kc = J.ko = 0.0;
    double btran_sun = 0.0, btran_sha = 0.0, forc_po2 = 0.0, forc_pco2 = 0.0;
    if (pos >= 0) {
      double unused12m = 0.0, unused22m = 0.0;
      obu_trip = obu;
      zldis = C(zl_u);
      // the 2 m profiles (:239-240) are only read by compute_flux: k_cf_finish evaluates them from obu_trip
      {
        const double z0mv = C(z0mv);
        friction_profiles_zl<true, false>(zldis, C(zl_t), C(zl_q), same_tq, um, obu, z0mv, z0mv, z0mv, s_fv, ustar, temp1, temp2,
                                          unused12m, unused22m);
      }
      PR_T(1)
      tlbef = t_veg;
      del2 = del;
      const double ram = 1.0 / (ustar * ustar / um);
      rah0 = 1.0 / (temp1 * ustar);
      raw0 = 1.0 / (temp2 * ustar);
      uaf = um * sqrt(1.0 / (ram * um));
      const double cf = 0.01 / (sqrt(uaf) * PR[PFT_sqrt_dleaf]);
      rb = 1.0 / (cf * uaf);
      {
        const double w = C(w_lai);
        const double csoilb = (VKC / (0.13 * elmk_pow((C(z0mg) * uaf / 1.5e-5), 0.45)));
        const double ri = (GRAV * C(htop) * (taf - C(t_grnd))) / (taf * elmk_sq(uaf));
        double csoilcn;
        if ((taf - C(t_grnd)) > 0.0) {
          const double ricsoilc = CSOILC / (1.0 + 0.5 * dmin(ri, 10.0));
          csoilcn = csoilb * w + ricsoilc * (1.0 - w);
        } else {
          csoilcn = csoilb * w + CSOILC * (1.0 - w);
        }
        rah1 = 1.0 / (csoilcn * uaf);
      }
      raw1 = rah1;
      svpts = el;
      eah = C(forc_pbot) * qaf / 0.622;
      PR_T(2)

      {
        {
          // temperature factors of this trip, shared by both phases; consumed here for both
          const double tc = C(tc10);
          const PsnTemp T = psn_temp(PR, c3flag, 668.39 - 1.07 * tc, 659.70 - 0.75 * tc, C(vcmaxc), C(jmaxc), C(tpuc),
                                     (404.9 / 1.e06) * C(forc_pbot), (278.4 / 1.e03) * C(forc_pbot), C(cp25), t_veg, day);
          PsnCol I;
          I.c3flag = c3flag;
          I.vcmax25top = PR[PFT_vcmax25top];
          I.jmax25top = (2.59 - 0.035 * tc) * PR[PFT_vcmax25top];
          I.cf = 0.0;  // (not read by psn_phase_inputs)
          I.qe = I.theta_cj = I.bbbopt = I.mbbopt = 0.0;
          // the soybean adjustment of btran runs once in front of each phase (:282-292), so the shaded phase sees it twice
          if (soy) btran = dmin(1.0, btran * 1.25);
          btran_sun = btran;
          if (soy) btran = dmin(1.0, btran * 1.25);
          btran_sha = btran;
          qsun = psn_phase_inputs(I, T, btran_sun, C(vcmaxcintsun), C(parsun));
          qsha = psn_phase_inputs(I, T, btran_sha, C(vcmaxcintsha), C(parsha));
          J.cp = T.cp;
          J.kc = T.kc;
          J.ko = T.ko;
        }
        J.c3flag = c3flag;
        J.cf = C(cf);
        J.qe = PR[PFT_qe];
        J.theta_cj = PR[PFT_theta_cj];
        J.bbbopt = PR[PFT_bbbopt];
        J.mbbopt = PR[PFT_mbbopt];
        forc_po2 = derive_forc_po2(C(forc_pbot));
        forc_pco2 = derive_forc_pco2(C(forc_pbot));
        if (given & 6) {  // (L2-level entry only: the column's own values, kept in its queue record)
          const gptr<const double> grec = S->cf_rec + CF_REC_BASE(pos);
          forc_po2 = grec[CF_REC_K(REC_forc_po2)];
          forc_pco2 = grec[CF_REC_K(REC_forc_pco2)];
        }
      }
      PR_T(3)
    }

    // ---------------- the two photosynthesis root finds of the trip (photosynthesis_impl.hh:9-283, called at :296 and :316) ----------------
    // The sunlit and the shaded solve of a column are independent and run one after the other on its lane.  A column that does not
    // converge takes 41 trips of ~20-30 us each (Brent's method in both phases of every trip) whatever the rest of the machine does,
    // and at a million columns that one serial chain IS the kernel's duration: 41-trip columns started at time zero end when the kernel
    // ends.  So whenever a wave holds no more day lanes than other lanes - the tail, and the passage from day to night columns -
    // every day lane borrows one of the others: the HELPER takes the shaded phase's inputs by shuffle, both lanes run the one solve
    // side by side, and the owner takes the result back.  Same function, same operands, another lane: the bits do not change.
    double rssun = 0.0, rssha = 0.0;
    {
      const bool has = pos >= 0;
      const bool dayl = has && day;
#if CF_PROBE >= 4
      if (has) {
        rssun = psn_phase_solve(J, qsun, nrad, C(forc_pbot), svpts, eah, forc_po2, forc_pco2, rb, btran_sun, C(parsun), C(lai_sun_z), err PR_SOLVE_ARGS(0, pr_nev_sun));
        PR_T(4)
        rssha = psn_phase_solve(J, qsha, nrad, C(forc_pbot), svpts, eah, forc_po2, forc_pco2, rb, btran_sha, C(parsha), C(lai_sha_z), err PR_SOLVE_ARGS(1, pr_nev_sha));
      }
#else
      if (has && !day) {  // no light in either phase (or no canopy layer): no root find (:161-173)
        rssun = psn_phase_dark(J, nrad, rb, btran_sun, C(lai_sun_z));
        rssha = psn_phase_dark(J, nrad, rb, btran_sha, C(lai_sha_z));
      }
      const unsigned long long md = __ballot(dayl);
      const int nd = __popcll(md);
      if (nd != 0) {  // (wave-uniform)
        const bool paired = CF_PAIR_PHASES && nd <= 32;
        // what the solve reads beside J / qsun, as this lane will hand it in: its own sunlit phase first
        double pb = has ? C(forc_pbot) : 0.0, rbi = rb, bt = btran_sun;
        double par = has ? C(parsun) : 0.0, lai = has ? C(lai_sun_z) : 0.0;
        int nr = nrad;
        int helper_of = lane;
        bool is_helper = false;
        if (paired) {
          // the r-th day lane pairs with the r-th lane that is not a day lane (there are at least as many)
          const int rd = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(md >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)md, 0u));
          const int rn = lane - rd;  // rank among the lanes that are not day lanes
          is_helper = !dayl && rn < nd;
          if (dayl) helper_of = nth_set_bit(~md, rd);
          const int src = is_helper ? nth_set_bit(md, rn) : lane;
          // Every lane takes part in the shuffles; only the helpers keep what they pull - the SHADED inputs of their owner - and
          // they keep it in the slots of their own sunlit inputs, which are dead by now (a helper's own column has no light: its
          // two results were formed above; rb and nrad, which its own trip still reads, go to copies).
          const double par_h = has ? C(parsha) : 0.0, lai_h = has ? C(lai_sha_z) : 0.0;
#define PULL(dst, own_shaded)                             \
  {                                                       \
    const double t_ = __shfl((own_shaded), src, 64);      \
    if (is_helper) dst = t_;                              \
  }
          PULL(qsun.lmr_z, qsha.lmr_z) PULL(qsun.vcmax_z, qsha.vcmax_z) PULL(qsun.jmax_z, qsha.jmax_z) PULL(qsun.tpu_z, qsha.tpu_z)
          PULL(qsun.kp_z, qsha.kp_z) PULL(bt, btran_sha) PULL(par, par_h) PULL(lai, lai_h)
          PULL(pb, pb) PULL(svpts, svpts) PULL(eah, eah) PULL(forc_po2, forc_po2) PULL(forc_pco2, forc_pco2) PULL(rbi, rb)
          PULL(J.cf, J.cf) PULL(J.qe, J.qe) PULL(J.theta_cj, J.theta_cj) PULL(J.bbbopt, J.bbbopt) PULL(J.mbbopt, J.mbbopt)
          PULL(J.cp, J.cp) PULL(J.kc, J.kc) PULL(J.ko, J.ko)
#undef PULL
          const int packed = __shfl((nrad << 1) | (J.c3flag ? 1 : 0), src, 64);
          if (is_helper) {
            nr = packed >> 1;
            J.c3flag = (packed & 1) != 0;
          }
        }
        const int nph = paired ? 1 : 2;
#pragma unroll 1
        for (int ph = 0; ph < nph; ph++) {  // ONE instance of the solve in the code: unpaired, the shaded phase is its second pass
          if (ph == 1) {
            qsun = qsha;
            bt = btran_sha;
            par = has ? C(parsha) : 0.0;
            lai = has ? C(lai_sha_z) : 0.0;
          }
          double r = 0.0;
          uint32_t e = 0u;
          if (dayl || is_helper) r = psn_phase_solve(J, qsun, nr, pb, svpts, eah, forc_po2, forc_pco2, rbi, bt, par, lai, e);
          if (paired) {  // the owner takes the shaded result and its flags back from its helper
            const double rh = __shfl(r, helper_of, 64);
            const uint32_t eh = (uint32_t)__shfl((int)e, helper_of, 64);
            if (dayl) {
              rssun = r;
              rssha = rh;
              err |= e | eh;
            }
          } else if (dayl) {
            if (ph == 0) rssun = r;
            else rssha = r;
            err |= e;
          }
        }
      }
#endif
    }
    if (pos >= 0) {
      PR_T(5)
#if CF_PROBE >= 4
      pr_brent += (uint64_t)__popcll(__ballot((err & 0x80000000u) != 0u));
      pr_brent_trips += (__ballot((err & 0x80000000u) != 0u) != 0ull) ? 1u : 0u;
      pr_c4 += (__ballot(!c3flag) != 0ull) ? 1u : 0u;
      pr_day += (uint64_t)__popcll(__ballot(day));
      pr_daytrips += (__ballot(day) != 0ull) ? 1u : 0u;
      err &= 0x7FFFFFFFu;
#endif

      const double wta = 1.0 / rah0;
      const double wtl = C(elai_esai) / rb;
      const double wtg = 1.0 / rah1;
      const double wtshi = 1.0 / (wta + wtl + wtg);
      const double wtl0 = wtl * wtshi;
      const double wtg0 = wtg * wtshi;
      const double wta0 = wta * wtshi;
      const double wtga = wta0 + wtg0;
      const double wtal = wta0 + wtl0;

      double rppdry;
      if (C(fdry) > 0.0) {
        rppdry = C(fdry) * rb * (C(laisun) / (rb + rssun) + C(laisha) / (rb + rssha)) / C(elai);
      } else {
        rppdry = 0.0;
      }
      const double forc_rho = C(forc_rho);
      const double h2ocan_dt = C(h2ocan_dt);
      double efpot = forc_rho * wtl * (qsatl - qaf);
      double rpp, qflx_tran_veg;
      if (efpot > 0.0) {
        if (btran > 0.0) {
          qflx_tran_veg = efpot * rppdry;
          rpp = rppdry + C(fwet);
        } else {
          rpp = C(fwet);
          qflx_tran_veg = 0.0;
        }
        rpp = dmin(rpp, (qflx_tran_veg + h2ocan_dt) / efpot);
      } else {
        rpp = 1.0;
        qflx_tran_veg = 0.0;
      }

      const double wtaq = fvn / raw0;
      const double wtlq = fvn * C(elai_esai) / rb * rpp;
      const double rdl = C(rdl_num) / (0.004 * uaf);
      double wtgq;
      if (delq < 0.0) {
        wtgq = fvn / (raw1 + rdl);
      } else {
        wtgq = C(soilbeta) * fvn / (raw1 + rdl);
      }
      const double wtsqi = 1.0 / (wtaq + wtlq + wtgq);
      const double wtgq0 = wtgq * wtsqi;
      const double wtlq0 = wtlq * wtsqi;
      const double wtaq0 = wtaq * wtsqi;
      const double wtgaq = wtaq0 + wtgq0;
      const double wtalq = wtaq0 + wtlq0;
      const double dc1 = forc_rho * CPAIR * wtl;
      const double dc2 = HVAP * forc_rho * wtlq;
      const double efsh = dc1 * (wtga * t_veg - wtg0 * C(t_grnd) - wta0 * C(thm));
      double efe = dc2 * (wtgaq * qsatl - wtgq0 * C(qg) - wtaq0 * C(forc_q));

      double erre = 0.0;
      if ((efe * efeb) < 0.0) {
        const double efeold = efe;
        efe = 0.1 * efeold;
        erre = efe - efeold;
      }
      const double tveg3 = elmk_pow(t_veg, 3.0);  // t_veg == tlbef here: also the pow(tlbef, 3.0) of errv below (:399)
      const double rad_in = C(rad_in);
      const double lw_term = C(lw_term);
      const double bir = C(bir);
      double dt_veg = (rad_in + bir * elmk_pow(t_veg, 4.0) + lw_term - efsh - efe) /
                      (-4.0 * bir * tveg3 + dc1 * wtga + dc2 * wtgaq * qsatldT);
      t_veg = tlbef + dt_veg;
      const double dels = dt_veg;
      del = fabs(dels);
      double errv = 0.0;
      if (del > 1.0) {
        dt_veg = dels / del;
        t_veg = tlbef + dt_veg;
        errv = rad_in + bir * tveg3 * (tlbef + 4.0 * dt_veg) + lw_term - (efsh + dc1 * wtga * dt_veg) -
               (efe + dc2 * wtgaq * qsatldT * dt_veg);
      }
      efpot = forc_rho * wtl * (wtgaq * (qsatl + qsatldT * dt_veg) - wtgq0 * C(qg) - wtaq0 * C(forc_q));
      double qflx_evap_veg = rpp * efpot;
      if (efpot > 0.0 && btran > 0.0) {
        qflx_tran_veg = efpot * rppdry;
      } else {
        qflx_tran_veg = 0.0;
      }
      const double ecidif = dmax(0.0, qflx_evap_veg - qflx_tran_veg - h2ocan_dt);
      qflx_evap_veg = dmin(qflx_evap_veg, qflx_tran_veg + h2ocan_dt);
      const double eflx_sh_veg = efsh + dc1 * wtga * dt_veg + errv + erre + HVAP * ecidif;
      PR_T(6)
      double deldT;
      qsat(t_veg, C(forc_pbot), el, deldT, qsatl, qsatldT);

      taf = wtg0 * C(t_grnd) + wta0 * C(thm) + wtl0 * t_veg;
      qaf = wtlq0 * qsatl + wtgq0 * C(qg) + C(forc_q) * wtaq0;
      const double dth = C(thm) - taf;
      const double dqh = C(forc_q) - qaf;
      delq = wtalq * C(qg) - wtlq0 * qsatl - wtaq0 * C(forc_q);
      const double tstar = temp1 * dth;
      const double qstar = temp2 * dqh;
      const double thvstar = tstar * (1.0 + 0.61 * C(forc_q)) + 0.61 * C(forc_th) * qstar;
      double zeta = zldis * VKC * GRAV * thvstar / (elmk_sq(ustar) * C(thv));
      if (zeta >= 0.0) {
        zeta = dmin(2.0, dmax(zeta, 0.01));
        um = dmax(C(ur), 0.1);
      } else {
        zeta = dmax(-100.0, dmin(zeta, -0.01));
        const double wc = 1.0 * elmk_pow((-GRAV * ustar * thvstar * 1000.0 / C(thv)), 0.333);
        um = sqrt(C(ur) * C(ur) + wc * wc);
      }
      obu = zldis / zeta;
      if (obuold * obu < 0.0) nmozsgn += 1;
      if (nmozsgn >= 4) obu = zldis / (-0.01);
      obuold = obu;

      itlef += 1;
      bool stop = false;
      if (itlef > 2) {  // itmin
        const double dele = fabs(efe - efeb);
        efeb = efe;
        const double det = dmax(del, del2);
        if ((det < 0.01) && (dele < 0.1)) stop = true;
      }
      if (itlef > 40) stop = true;  // itmax: while (itlef <= itmax && !stop)

      PR_T(7)
      // ---------------- converged: hand the state to k_cf_finish, release the lane ----------------
      if (stop) {
        const gptr<double> fin = S->cf_fin + CF_FIN_BASE(pos);
        sc_st<4>(fin + CF_FIN_K(FIN_t_veg), t_veg);
        sc_st<4>(fin + CF_FIN_K(FIN_btran), btran);
        sc_st<4>(fin + CF_FIN_K(FIN_qflx_tran_veg), qflx_tran_veg);
        sc_st<4>(fin + CF_FIN_K(FIN_qflx_evap_veg), qflx_evap_veg);
        sc_st<4>(fin + CF_FIN_K(FIN_eflx_sh_veg), eflx_sh_veg);
        sc_st<4>(fin + CF_FIN_K(FIN_wtg), wtg);
        sc_st<4>(fin + CF_FIN_K(FIN_wtl0), wtl0);
        sc_st<4>(fin + CF_FIN_K(FIN_wta0), wta0);
        sc_st<4>(fin + CF_FIN_K(FIN_wtal), wtal);
        sc_st<4>(fin + CF_FIN_K(FIN_wtgq), wtgq);
        sc_st<4>(fin + CF_FIN_K(FIN_wtalq), wtalq);
        sc_st<4>(fin + CF_FIN_K(FIN_wtlq0), wtlq0);
        sc_st<4>(fin + CF_FIN_K(FIN_wtaq0), wtaq0);
        sc_st<4>(fin + CF_FIN_K(FIN_delq), delq);
        sc_st<4>(fin + CF_FIN_K(FIN_qsatl), qsatl);
        sc_st<4>(fin + CF_FIN_K(FIN_temp1), temp1);
        sc_st<4>(fin + CF_FIN_K(FIN_temp2), temp2);
        sc_st<4>(fin + CF_FIN_K(FIN_dth), dth);
        sc_st<4>(fin + CF_FIN_K(FIN_dqh), dqh);
        sc_st<4>(fin + CF_FIN_K(FIN_tlbef), tlbef);
        sc_st<4>(fin + CF_FIN_K(FIN_dt_veg), dt_veg);
        sc_st<4>(fin + CF_FIN_K(FIN_obu_trip), obu_trip);
        sc_st<4>(fin + CF_FIN_K(FIN_trips), (double)itlef);
        sc_st<4>(fin + CF_FIN_K(FIN_err), (double)err);
        pos = -1;
      }
      PR_T(8)
#if CF_PROBE >= 4
      pr_cols += (uint64_t)__popcll(__ballot(stop));
#endif
    }
  }
#if CF_PROBE >= 4
  uint64_t pr_nev_sun_w = pr_nev_sun, pr_nev_sha_w = pr_nev_sha;  // wave sums of the per-lane evaluation counts
  for (int off = 32; off; off >>= 1) {
    pr_nev_sun_w += __shfl_xor(pr_nev_sun_w, off, 64);
    pr_nev_sha_w += __shfl_xor(pr_nev_sha_w, off, 64);
  }
  if (lane == 0) {
    const gptr<double> o = S->wk + (int64_t)WK_DEBUG * ld + ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32;
#if CF_PROBE == 5
    for (int i = 0; i < 12; i++) o[16 + i] = (double)pr_acc[i];
#endif
    o[8] = (double)pr_daytrips;
    o[28] = (double)pr_nev_sun_w;
    o[29] = (double)pr_nev_sha_w;
    o[30] = (double)s_wev[0][threadIdx.x >> 6];
    o[31] = (double)s_wev[1][threadIdx.x >> 6];
    o[0] = (double)pr_t0;
    o[1] = (double)pr_texh;
    o[2] = (double)wall_clock64();
    o[3] = (double)pr_trips;
    o[4] = (double)pr_lanes;
    o[5] = (double)pr_refills;
    o[6] = (double)pr_cols;
    o[7] = 1.0;
    o[15] = (double)pr_brent;
    o[14] = (double)pr_brent_trips;
    o[13] = (double)pr_c4;
    o[12] = (double)pr_day;
  }
#endif
}
#undef C
#undef CSET
__global__ __launch_bounds__(CF_ITER_THREADS, 2) void k_cf_iterate(const DevState* __restrict__ S, double dtime, const int given)
{
  cf_iterate_body<CF_ITER_THREADS>(S, dtime, given);
}
// The same iteration in 256-thread workgroups: one wave per SIMD and 92 KB of LDS, one workgroup per CU - slower on its own (the
// second wave per SIMD is what fills the fp64 pipe, 1.29 against 0.87 ms per million columns) but it leaves half the register file
// and 68 KB of LDS to the workgroups of OTHER kernels, which are then resident on the same CUs: the launch shape for a block of columns
// whose iteration runs beside another block's streaming kernels (elmk_set_option ELMK_OPT_CF_HALF_WORKGROUPS,
// profiles/r04_two_block_overlap.txt).
__global__ __launch_bounds__(CF_ITER_THREADS_HALF, 2) void k_cf_iterate_half(const DevState* __restrict__ S, double dtime, const int given)
{
  cf_iterate_body<CF_ITER_THREADS_HALF>(S, dtime, given);
}

// =====================================================================================================
// k_cf_finish - one thread per column, coalesced: compute_flux (canopy_fluxes_impl.hh:456-540) from the converged
// iteration state, the 2 m profiles of the last trip (friction_velocity_temp2m / _humidity2m, :239-240), state writes.
// =====================================================================================================
__global__ __launch_bounds__(256) void k_cf_finish(const DevState* __restrict__ S, double dtime, const int given)
{
  elmk_math_lds_init<false>();
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ld = S->ld;
  if (blockIdx.x == 0 && threadIdx.x < CF_NCLS) CF_CLASS_COUNT(S, threadIdx.x) = 0u;  // for the next call's k_cf_count
  if (S->land.lakpoi || c >= S->ncols) return;
  const int32_t pos = S->cf_pos[c];
  if (pos < 0) {  // not vegetated: what is left of the bare branch (see cf_init_col)
    S->cgrnd[c] = 0.0;
    S->cgrnds[c] = 0.0;
    S->cgrndl[c] = 0.0;
    return;
  }
  CfFin f;
  const gptr<const double> fin = S->cf_fin + CF_FIN_BASE((int64_t)pos);
#define X(n) f.n = sc_ld<4>(fin + CF_FIN_K(FIN_##n));
  CF_FIN_FIELDS(X)
#undef X
  const double t_veg = f.t_veg;
  S->btran[c] = f.btran;
  S->t_veg[c] = t_veg;
  S->qflx_tran_veg[c] = f.qflx_tran_veg;
  S->qflx_evap_veg[c] = f.qflx_evap_veg;
  S->eflx_sh_veg[c] = f.eflx_sh_veg;
  const double forc_pbot = S->forc_pbot[c], forc_q = S->forc_qbot[c];
  double forc_rho = derive_forc_rho(forc_pbot, forc_q, S->forc_tbot[c]);
  if (given & 1) forc_rho = S->cf_given[c];
  const double thm = S->thm[c], t_grnd = S->t_grnd[c];
  const int snl = S->snl[c];
  const double t_soi0 = LV(t_soisno, NLEVSNO);
  const double t_top = (snl > 0) ? LV(t_soisno, NLEVSNO - snl) : t_soi0;
  const double wtg = f.wtg, wtl0 = f.wtl0, wta0 = f.wta0, wtal = f.wtal;
  const double wtgq = f.wtgq, wtalq = f.wtalq, wtlq0 = f.wtlq0, wtaq0 = f.wtaq0, qsatl = f.qsatl;
  const double delt = wtal * t_grnd - wtl0 * t_veg - wta0 * thm;
  S->eflx_sh_grnd[c] = CPAIR * forc_rho * wtg * delt;
  const double delt_snow = wtal * t_top - wtl0 * t_veg - wta0 * thm;
  S->eflx_sh_snow[c] = CPAIR * forc_rho * wtg * delt_snow;
  const double delt_soil = wtal * t_soi0 - wtl0 * t_veg - wta0 * thm;
  S->eflx_sh_soil[c] = CPAIR * forc_rho * wtg * delt_soil;
  const double delt_h2osfc = wtal * S->t_h2osfc[c] - wtl0 * t_veg - wta0 * thm;
  S->eflx_sh_h2osfc[c] = CPAIR * forc_rho * wtg * delt_h2osfc;
  S->qflx_evap_soi[c] = forc_rho * wtgq * f.delq;
  const double delq_snow = wtalq * S->qg_snow[c] - wtlq0 * qsatl - wtaq0 * forc_q;
  S->qflx_ev_snow[c] = forc_rho * wtgq * delq_snow;
  const double delq_soil = wtalq * S->qg_soil[c] - wtlq0 * qsatl - wtaq0 * forc_q;
  S->qflx_ev_soil[c] = forc_rho * wtgq * delq_soil;
  const double delq_h2osfc = wtalq * S->qg_h2osfc[c] - wtlq0 * qsatl - wtaq0 * forc_q;
  S->qflx_ev_h2osfc[c] = forc_rho * wtgq * delq_h2osfc;
  // 2 m profiles of the last trip: z0hv == z0qv == z0mv here, so humidity2m takes temp2m's value (:153)
  const double z0mv = S->z0mv[c];
  const double temp12m = fv_profile<true>(2.0 + z0mv, f.obu_trip, z0mv);
  const double temp22m = temp12m;
  const double temp1 = f.temp1, temp2 = f.temp2;
  const double t_ref2m = thm + temp1 * f.dth * (1.0 / temp12m - 1.0 / temp1);
  const double q_ref2m = forc_q + temp2 * f.dqh * (1.0 / temp22m - 1.0 / temp2);
  double e_ref2m, de2mdT, qsat_ref2m, dqsat2mdT;
  qsat(t_ref2m, forc_pbot, e_ref2m, de2mdT, qsat_ref2m, dqsat2mdT);
  S->t_ref2m[c] = t_ref2m;
  S->q_ref2m[c] = q_ref2m;
  S->rh_ref2m[c] = dmin(100.0, (q_ref2m / qsat_ref2m) * 100.0);
  const double emv = S->emv[c], emg = S->emg[c], forc_lwrad = S->forc_lwrad[c];
  const double tlbef = f.tlbef, dt_veg = f.dt_veg;
  const double lw_grnd = S->wk[(int64_t)WK_CF_LWGRND * ld + c];
  S->dlrad[c] = (1.0 - emv) * emg * forc_lwrad + emv * emg * STEBOL * elmk_pow(tlbef, 3.0) * (tlbef + 4.0 * dt_veg);
  S->ulrad[c] = ((1.0 - emg) * (1.0 - emv) * (1.0 - emv) * forc_lwrad +
                 emv * (1.0 + (1.0 - emg) * (1.0 - emv)) * STEBOL * elmk_pow(tlbef, 3.0) * (tlbef + 4.0 * dt_veg) +
                 emg * (1.0 - emv) * STEBOL * lw_grnd);
  double cgrnds = 0.0, cgrndl = 0.0;
  cgrnds += CPAIR * forc_rho * wtg * wtal;
  cgrndl += forc_rho * wtgq * wtalq * S->dqgdT[c];
  S->cgrnds[c] = cgrnds;
  S->cgrndl[c] = cgrndl;
  S->cgrnd[c] = cgrnds + cgrndl * S->htvp[c];
  S->h2ocan[c] = dmax(0.0, S->h2ocan[c] + (f.qflx_tran_veg - f.qflx_evap_veg) * dtime);
  const uint32_t err = (uint32_t)f.err;
  if (err) S->err_flags[c] |= err;
  // low half: trips of this call (diagnostics); high half: scheduling hint = slowly decaying maximum of the trip
  // count, so a column that needed many trips recently keeps being scheduled early even if its last call was short
  {
    const int trips = (int)f.trips;
    const int decayed = (S->cf_niter[c] >> 16) - 1;
    S->cf_niter[c] = ((trips > decayed ? trips : decayed) << 16) | trips;
  }
}

// =====================================================================================================
// elmk_timestep7_fused: the seven wrappers of ELMInterface::advance (elm_kokkos_interface.cc:289-307) with the five
// streaming ones between albedo and the leaf-temperature iteration as ONE pass per column.
//
//   k_fz_prep    frac_wet (it must precede albedo, which reads fwet); resets of the work lists of the step; the scheduling
//                class of every column for the canopy_fluxes queue and the class counts (what k_cf_count does) - the class
//                is computed from inputs of the step (frac_veg_nosno, coszen, the incident visible flux, last call's trip
//                hint), because the queue slices must exist before the streaming pass places its records, and it is
//                stored per column so that count and placement agree by construction (a class only orders the queue: any
//                consistent choice gives the same results)
//   albedo       k_alb_classify -> k_alb_snicar<1..5> -> k_alb_final, unchanged: it reads snl / h2osno / frac_sno as they
//                are BEFORE canopy_hydrology, so it cannot join the pass
//   k_fz_stream  canopy_hydrology -> surface_radiation -> canopy_temperature -> bareground_fluxes' streaming stage (cgrnd*
//                reset, list of bare columns) -> canopy_fluxes' initialize_flux + queue record, in the reference's order, by
//                the bodies the separate kernels use (elmk_stream.h, cf_init_col); each later body takes what an earlier
//                one produced from registers: t_soisno[20] and some 45 scalars per column are neither re-read nor waited for
//   k_bg_flux, k_cf_iterate, k_cf_finish   as in the unfused step
// =====================================================================================================
#ifndef FZ_PREP_TILES_N
#define FZ_PREP_TILES_N 4
#endif
constexpr int FZ_PREP_TILES = FZ_PREP_TILES_N;  // tiles of 256 columns per workgroup of k_fz_prep
__global__ __launch_bounds__(256) void k_fz_prep(const DevState* __restrict__ S)
{
  elmk_math_lds_init<false>();  // (frac_wet: one pow per column)
  __shared__ uint32_t s_cnt[FZ_PREP_TILES][CF_NCLS];
  __shared__ uint32_t a_cnt[5], a_base[5];  // the SNICAR queues by snow-layer count (albedo stage 1)
  for (int i = threadIdx.x; i < FZ_PREP_TILES * CF_NCLS; i += blockDim.x) (&s_cnt[0][0])[i] = 0u;
  if (threadIdx.x < 5) a_cnt[threadIdx.x] = 0u;
  // (not the canopy queue's own counters: cf_queue_position; not the SNICAR queues, which this kernel fills and k_alb_final
  //  leaves empty for the next call)
  if (blockIdx.x == 0 && threadIdx.x < NLISTS && threadIdx.x != LIST_CF_QUEUE && !(threadIdx.x >= LIST_ALB_0 && threadIdx.x <= LIST_ALB_5)) {
    ELMK_LIST_COUNT(S, threadIdx.x) = 0u;
    ELMK_LIST_HEAD(S, threadIdx.x) = 0u;
  }
  __syncthreads();
  const Land L = S->land;
  const int64_t ld = S->ld;
  const int lane = threadIdx.x & 63;
  const int64_t tile0 = (int64_t)blockIdx.x * FZ_PREP_TILES;
  uint32_t apack[FZ_PREP_TILES];  // per tile: queue class << 16 | slot inside this workgroup's share of the queue
#pragma unroll
  for (int t = 0; t < FZ_PREP_TILES; t++) {  // (unrolled: the loads of several tiles are in flight together)
    const int64_t c = (tile0 + t) * 256 + threadIdx.x;
    const bool inside = c < S->ncols;
    int cls = -1, acl = -1;
    if (inside) {
      frac_wet_col(S, c, L);
      if (!L.lakpoi && !L.urbpoi && S->frac_veg_nosno[c] != 0) {
        const bool day = S->coszen[c] > 0.0 && (S->forc_solad[c] > 0.0 || S->forc_solai[c] > 0.0);  // (band 0: visible)
        cls = cf_class_of(day, S->cf_niter[c] >> 16);
      }
      S->cf_cls[c] = (int8_t)cls;
      if (!L.urbpoi) {  // stage 1 of albedo (k_alb_classify): soil albedo of the sunlit columns, SNICAR queue class
        const int nl = alb_main_column(S, c, ld, L);
        acl = nl >= 1 ? nl - 1 : -1;
      }
    }
#pragma unroll
    for (int k = 0; k < CF_NCLS; k++) {
      const unsigned long long m = __ballot(cls == k);
      if (lane == 0 && m) atomicAdd(&s_cnt[t][k], (uint32_t)__popcll(m));
    }
    uint32_t aoff = 0u;
#pragma unroll
    for (int k = 0; k < 5; k++) {
      const unsigned long long m = __ballot(acl == k);
      if (m != 0ull) {
        const int leader = __ffsll((long long)m) - 1;
        uint32_t off = 0u;
        if (lane == leader) off = atomicAdd(&a_cnt[k], (uint32_t)__popcll(m));
        off = __shfl(off, leader, 64);
        if (acl == k) aoff = off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      }
    }
    apack[t] = acl >= 0 ? ((uint32_t)acl << 16) | aoff : 0xffffffffu;
  }
  __syncthreads();
  if (threadIdx.x < CF_NCLS) {
    const int k = threadIdx.x;
    uint32_t n = 0u;
    for (int t = 0; t < FZ_PREP_TILES; t++) n += s_cnt[t][k];
    uint32_t base = n ? atomicAdd(ELMK_GENERIC(&CF_CLASS_COUNT(S, k)), n) : 0u;
    for (int t = 0; t < FZ_PREP_TILES; t++) {
      if (tile0 + t < S->cf_nblk) S->cf_blk[(int64_t)k * S->cf_nblk + tile0 + t] = base;
      base += s_cnt[t][k];
    }
  }
  if (threadIdx.x < 5) {
    const uint32_t n = a_cnt[threadIdx.x];
    a_base[threadIdx.x] = n ? atomicAdd(ELMK_GENERIC(&ELMK_LIST_COUNT(S, LIST_ALB_1 + threadIdx.x)), n) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < FZ_PREP_TILES; t++) {
    if (apack[t] != 0xffffffffu) {
      const int k = (int)(apack[t] >> 16);
      const int64_t at = (int64_t)a_base[k] + (apack[t] & 0xffffu);
      if (at < ld) S->lists[(int64_t)(LIST_ALB_1 + k) * ld + at] = (int32_t)((tile0 + t) * 256 + threadIdx.x);  // (bound: see block_classify_append)
    }
  }
}

// k_fz_pre - the part of the streaming pass that depends on nothing the albedo stage or canopy_hydrology produce, as a kernel
// of its own that runs BESIDE the albedo stage on a side stream: SNICAR and the two-stream solution are bound by fp64 issue,
// this kernel by HBM, and the two kinds of work share the CUs.  Per column: the queue position of the canopy_fluxes record,
// the root moisture stress of the 15 soil levels (soil_moist_stress: 105 level values read, rootr / eff_porosity written,
// fifteen pow) and old_ground_temp's copy of the soil levels of t_soisno into tssbef - 1.4 KB of the 3.1 KB per column that
// k_fz_stream used to move.  What lies between this kernel and k_fz_stream in the reference's call order touches none of it
// with one exception: fraction_h2osfc folds a vanishing pond (h2osfc <= 1e-8) into the top soil layer's liquid
// (canopy_hydrology_impl.hh:334-337), which calc_volumetric_h2oliq then reads - the same sum is formed here for that level.
__device__ __forceinline__ void fz_pre_tile(const DevState* __restrict__ S, const int64_t tile);
__global__ __launch_bounds__(256) void k_fz_pre(const DevState* __restrict__ S)
{
  elmk_math_lds_init<false>();
  if (S->land.lakpoi) {  // uniform: neither canopy_temperature's copy nor canopy_fluxes does anything on lake land units
    if (blockIdx.x == 0 && threadIdx.x == 0) ELMK_LIST_COUNT(S, LIST_CF_QUEUE) = 0u;
    return;
  }
  fz_pre_tile(S, (int64_t)blockIdx.x);
}
__device__ __forceinline__ void fz_pre_tile(const DevState* __restrict__ S, const int64_t tile)
{
  const int64_t c = tile * blockDim.x + threadIdx.x;
  const int64_t ld = S->ld;
  const Land L = S->land;
  const bool inside = c < S->ncols;
  // queue position of the canopy_fluxes record (every thread of the workgroup takes part in the barrier inside)
  const int64_t pos = cf_queue_position(S, inside ? (int)S->cf_cls[c] : -1, tile);
  if (!inside) return;
  S->cf_pos[c] = (int32_t)pos;
  const bool wall = (L.ctype == icol_sunwall || L.ctype == icol_shadewall || L.ctype == icol_roof);
#pragma unroll
  for (int i = NLEVSNO; i < NLEVTOT; i++) LV(tssbef, i) = (wall && i > 5) ? SPVAL : (double)LV(t_soisno, i);
  if (pos >= 0) {
    double liq0 = LV(h2osoi_liq, NLEVSNO);
    if (L.ltype == istsoil || L.ltype == istcrop) {
      const double h2osfc = S->h2osfc[c];
      if (!(h2osfc > 1.e-8)) liq0 = liq0 + h2osfc;
    }
    cf_root_stress_col(S, c, ld, pos, liq0);
  } else if (!L.urbpoi) {
    S->btran[c] = 0.0;
#pragma unroll
    for (int i = 0; i < NLEVGRND; i++) LV(rootr, i) = 0.0;
  }
}

// k_fz_snicar_pre - ONE kernel with two kinds of workgroups, dealt alternately: the single-layer SNICAR queue (k_alb_snicar<1>'s
// work: bound by fp64 issue, 0.25 ms on the fixture-tiled state) and k_fz_pre's tiles (bound by HBM, 0.29 ms).  As two kernels,
// even on two streams, they ran one after the other - each launch fills the machine, and the command processor hands out one
// kernel's workgroups at a time (step timeline in profiles/r03_tl_fused_split_two_streams_tierA.txt: k_fz_pre beside empty
// SNICAR queues, k_alb_snicar<1> after it).  Inside one kernel both kinds are resident on every CU at once, so the memory-bound
// waves fill the issue slots the arithmetic-bound ones leave and the other way round.  Registers and LDS are the maximum of
// the two bodies (SNICAR's 112 VGPRs).
__global__ __launch_bounds__(256, 2) void k_fz_snicar_pre(const DevState* __restrict__ S, const uint32_t g_snicar, const uint32_t g_pre)
{
  // workgroup b -> (kind, index): alternate while both kinds last, then the longer one's remainder
  const uint32_t b = blockIdx.x, m = g_snicar < g_pre ? g_snicar : g_pre;
  bool pre;
  uint32_t idx;
  if (b < 2u * m) {
    pre = (b & 1u) != 0u;
    idx = b >> 1;
  } else {
    pre = g_pre > g_snicar;
    idx = m + (b - 2u * m);
  }
  if (!pre) {
    snicar_workgroup<1>(S, idx, g_snicar);
    return;
  }
  elmk_math_lds_init<false>();
  if (S->land.lakpoi) {
    if (idx == 0 && threadIdx.x == 0) ELMK_LIST_COUNT(S, LIST_CF_QUEUE) = 0u;
    return;
  }
  fz_pre_tile(S, (int64_t)idx);
}

#ifndef FZ_SNICAR_GRID_BY_TILES
#define FZ_SNICAR_GRID_BY_TILES 1  // 0: k_alb_snicar<1>'s own grid for the SNICAR part of k_fz_snicar_pre (development A/B)
#endif
#ifndef FZ_SPLIT
#define FZ_SPLIT 2  // 2: k_fz_pre's tiles share a kernel with the single-layer SNICAR queue (k_fz_snicar_pre); 1: k_fz_pre beside
                    // the albedo stage on a side stream; 0: k_fz_stream does that work itself, as in round 2 (1, 0: development A/B)
#endif
#ifndef FZ_ALB_FWD
#define FZ_ALB_FWD 1  // 0: k_alb_final as its own launch in front of k_fz_stream, as in round 3 (development A/B)
#endif
__global__ __launch_bounds__(256, 3) void k_fz_stream(const DevState* __restrict__ S, double dtime)
{
  elmk_math_lds_init<false>();
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t ld = S->ld;
  const Land L = S->land;
  const bool inside = c < S->ncols;
#if !FZ_SPLIT
  int64_t pos0 = -1;
  if (!L.lakpoi) {
    pos0 = cf_queue_position(S, inside ? (int)S->cf_cls[c] : -1);
    if (inside) S->cf_pos[c] = (int32_t)pos0;
  } else if (blockIdx.x == 0 && threadIdx.x == 0) {
    ELMK_LIST_COUNT(S, LIST_CF_QUEUE) = 0u;
  }
#endif
  ColFwd w;
#if FZ_ALB_FWD
  // Stage 3 of albedo_snicar (k_alb_final's body, elmk_albedo_fin.h) runs HERE, between canopy_hydrology and surface_radiation,
  // so that surface_radiation takes the 56 doubles it reads of that stage's outputs from registers instead of from the state
  // (-496 bytes per column and one launch less).  The reference runs albedo BEFORE canopy_hydrology (elm_kokkos_interface.cc:
  // 292-295), which changes two of its inputs - frac_sno and h2osno (snow_init, fraction_h2osfc) - so their values of before are
  // taken first; nothing else the stage reads (coszen, elai, esai, t_veg, fwet, the soil albedos, the SNICAR products) is written
  // by canopy_hydrology, and nothing canopy_hydrology reads is written by the stage.
  const bool alb_here = inside && !L.urbpoi;  // (kokkos_albedo_snicar does nothing on urban land units)
  const double frac_sno_before = alb_here ? (double)S->frac_sno[c] : 0.0, h2osno_before = alb_here ? (double)S->h2osno[c] : 0.0;
  if (blockIdx.x == 0 && threadIdx.x < 6) {  // (k_alb_final's job: the SNICAR queues are drained by now, leave them empty)
    ELMK_LIST_COUNT(S, LIST_ALB_0 + threadIdx.x) = 0u;
    ELMK_LIST_HEAD(S, LIST_ALB_0 + threadIdx.x) = 0u;
  }
#endif
  canopy_hydrology_col<true>(S, c, ld, L, dtime, w, inside);  // (every thread: the pond solves are pooled per workgroup)
  if (inside) {
#if FZ_ALB_FWD
    AlbFwd a;
    double flx[6][4];
    if (alb_here) {
      const AlbIn x = alb_final_inputs(S, c, ld, frac_sno_before, h2osno_before);
      alb_ground(S, c, ld, x.day, x.frac_sno, x.albsod, x.albsoi, x.sd_alb, x.si_alb, a);
      alb_two_stream(S, c, ld, L, x.day, x.coszen, x.elai, x.esai, x.vcmaxcintsun, x.vcmaxcintsha, a);
      alb_flux_abs_all(S, c, ld, L, x, flx);
      surface_radiation_col<true, true>(S, c, ld, L, w, a, flx);
    } else {
      surface_radiation_col<true, false>(S, c, ld, L, w, a, flx);
    }
#else
    const AlbFwd a{};
    const double flx[6][4] = {};
    surface_radiation_col<true>(S, c, ld, L, w, a, flx);
#endif
    canopy_temperature_col<true, FZ_SPLIT != 0>(S, c, ld, L, w);
  }
  // bareground_fluxes, streaming stage (k_bg_main): compute_flux's unconditional cgrnd reset and the list of bare columns
  if (!L.lakpoi) {
    if (inside) {
      S->cgrnd[c] = 0.0;
      S->cgrnds[c] = 0.0;
      S->cgrndl[c] = 0.0;
    }
    const bool bare = inside && !L.urbpoi && w.fvn == 0;
    block_classify_append<1>(S->lists, ld, S->counters, LIST_BG, bare ? 0 : -1, (int32_t)c);
    // canopy_fluxes up to the iteration (queue position and root moisture stress: k_fz_pre)
    if (inside) cf_init_col<true, FZ_SPLIT != 0>(S, c, ld, L, (int64_t)S->cf_pos[c], w);
  }
}

#ifndef FZ_BG_OVERLAP
#define FZ_BG_OVERLAP 1
#endif
// Workgroups of the persistent k_cf_iterate: two per CU are launched (one is resident at its LDS footprint; the ones that start
// later find the queue empty).  ELMK_CF_GROUPS (development, read once) overrides the 512: a smaller grid leaves CUs to kernels
// of another context's stream (tests/tools/two_ctx_overlap.py).
static unsigned cf_iterate_groups(unsigned nblk)
{
  static const unsigned cap = [] {
    const char* e = getenv("ELMK_CF_GROUPS");
    const long v = e ? atol(e) : 0;
    return v > 0 ? (unsigned)v : 512u;
  }();
  return nblk < cap ? nblk : cap;
}

static void launch_cf_iterate(const DevState* S, unsigned nblk, double dt, hipStream_t st, const SideStreams* side, int given)
{
  if (side && side->cf_half_groups > 0) {  // (one workgroup per CU and no more: what is launched is resident)
    const unsigned g = nblk * 256u / CF_ITER_THREADS_HALF < (unsigned)side->cf_half_groups ? nblk * 256u / CF_ITER_THREADS_HALF : (unsigned)side->cf_half_groups;
    hipLaunchKernelGGL(k_cf_iterate_half, dim3(g ? g : 1u), dim3(CF_ITER_THREADS_HALF), 0, st, S, dt, given);
    return;
  }
  hipLaunchKernelGGL(k_cf_iterate, dim3(cf_iterate_groups(nblk)), dim3(CF_ITER_THREADS), 0, st, S, dt, given);
}

void launch_fused_stage(const DevState* S, int64_t n, double dt, hipStream_t st, const SideStreams* side, int stage)
{
  if (n <= 0) return;
  const unsigned nblk = (unsigned)((n + 255) / 256);
  switch (stage) {
    case 0: hipLaunchKernelGGL(k_fz_prep, dim3((nblk + FZ_PREP_TILES - 1) / FZ_PREP_TILES), dim3(256), 0, st, S); break;
    case 1:
      if (!FZ_SPLIT) {
        launch_albedo_snicar(S, n, st, side, false, !FZ_ALB_FWD);
      } else if (FZ_SPLIT == 2 && n >= 262144) {
        // the SNICAR queues of 5..2 layers, then the single-layer queue and k_fz_pre's tiles as ONE kernel, then k_alb_final
        unsigned gs = 0;
        launch_albedo_snicar_part(S, n, st, 0, &gs);
        // as many SNICAR workgroups as tiles (each walks the queue with that stride: two or three entries per wave), so that the
        // two kinds alternate for the whole length of the kernel; with k_alb_snicar<1>'s grid of 4 096 long-lived workgroups
        // the SNICAR part filled the machine first at 10 M columns and the tiles ran behind it
        if (FZ_SNICAR_GRID_BY_TILES && gs < nblk) gs = nblk;
        hipLaunchKernelGGL(k_fz_snicar_pre, dim3(gs + nblk), dim3(256), 0, st, S, gs, nblk);
        if (!FZ_ALB_FWD) launch_albedo_snicar_part(S, n, st, 1, nullptr);  // (else k_fz_stream runs stage 3 itself)
      } else if (n >= 262144) {
        // k_fz_pre (memory-bound) beside the albedo stage (fp64-issue-bound) on a side stream; joined before k_fz_stream
        (void)hipEventRecord(side->fork, st);
        (void)hipStreamWaitEvent(side->s[0], side->fork, 0);
        hipLaunchKernelGGL(k_fz_pre, dim3(nblk), dim3(256), 0, side->s[0], S);
        (void)hipEventRecord(side->join[0], side->s[0]);
        launch_albedo_snicar(S, n, st, side, false, !FZ_ALB_FWD);
        (void)hipStreamWaitEvent(st, side->join[0], 0);
      } else {  // (few columns: the SNICAR queues themselves fork onto the side streams)
        hipLaunchKernelGGL(k_fz_pre, dim3(nblk), dim3(256), 0, st, S);
        launch_albedo_snicar(S, n, st, side, false, !FZ_ALB_FWD);
      }
      break;
    case 2: hipLaunchKernelGGL(k_fz_stream, dim3(nblk), dim3(256), 0, st, S, dt); break;
    case 3:
      // The bare-ground Monin-Obukhov loop works on bare columns only and the canopy iteration on vegetated ones.  The one field
      // group both k_bg_flux and k_cf_finish would store to on a bare column is cgrnd / cgrnds / cgrndl (compute_flux of
      // canopy_fluxes resets them on every column after bareground_fluxes has set them, canopy_fluxes_impl.hh:474-479): in the
      // fused step k_bg_flux therefore leaves them alone (launch_bareground_list passes given = 4; k_fz_stream has stored the
      // zeros that are their final value), so the two kernels have no store in common and may run side by side.
      // With FZ_BG_OVERLAP the list kernel goes to a side stream behind k_cf_iterate's launch (case 4): a few hundred workgroups
      // of long dependent chains that cannot fill the machine on their own (VALU busy 0.31 on the branch-mix tier) run in the
      // CUs the persistent iteration kernel frees during its tail, instead of in front of it.
      if (FZ_BG_OVERLAP && n >= 262144) {
        (void)hipEventRecord(side->fork, st);
      } else {
        launch_bareground_list(S, n, st);
      }
      break;
    default: {
      launch_cf_iterate(S, nblk, dt, st, side, 0);
      if (FZ_BG_OVERLAP && n >= 262144) {
        (void)hipStreamWaitEvent(side->s[0], side->fork, 0);
        launch_bareground_list(S, n, side->s[0]);
        (void)hipEventRecord(side->join[0], side->s[0]);
      }
      hipLaunchKernelGGL(k_cf_finish, dim3(nblk), dim3(256), 0, st, S, dt, 0);
      if (FZ_BG_OVERLAP && n >= 262144) (void)hipStreamWaitEvent(st, side->join[0], 0);
    }
  }
}

void launch_canopy_fluxes(const DevState* S, int64_t n, double dt, hipStream_t st, int given, const SideStreams* side)
{
  if (n <= 0) return;
  const unsigned nblk = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_cf_count, dim3((nblk + CF_COUNT_TILES - 1) / CF_COUNT_TILES), dim3(256 * CF_COUNT_TILES), 0, st, S);
  hipLaunchKernelGGL(k_cf_init, dim3(nblk), dim3(256), 0, st, S, given);
  // persistent: two waves per SIMD are resident at this kernel's register and LDS footprint (2 workgroups per CU, 512 in
  // all); workgroups that start later find the queue empty
  launch_cf_iterate(S, nblk, dt, st, side, given);
  hipLaunchKernelGGL(k_cf_finish, dim3(nblk), dim3(256), 0, st, S, dt, given);
}

}  // namespace elmk
