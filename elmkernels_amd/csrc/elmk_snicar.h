// elmk_snicar.h - SNICAR for the sunlit snow-covered columns (snow_snicar_impl.hh): one (pass, band) solve per lane, the band
// combination across lanes, and the work of one workgroup of the queue-driven kernel.  Shared by k_albedo_snicar.hip (the
// kernels k_alb_snicar<NL>) and k_canopy_fluxes.hip (the fused step's k_fz_snicar_pre, which hosts streaming work beside it).
#pragma once
#include "elmk_dev.h"
#include "elmk_albedo_col.h"

namespace elmk {

constexpr int SN_RDS_MAX_TBL = 1500, SN_RDS_MIN_TBL = 30;
// exp(-argmax), argmax = 10 (snow_snicar_impl.hh:360): the reference's constexpr value, 0x1.7cd79b5647c9bp-15
constexpr double SN_EXP_MIN = 4.5399929762484854e-05;

struct SnowOut {
  double alb[2];         // albout (VIS, NIR)
  double fabs_[6][2];    // flx_abs(i, {VIS, NIR})
};

// One (pass, band) of SNICAR for an active column (coszen > 0, h2osno > min_snw) with NL (possibly fictitious) snow
// layers: snow_aerosol_mie_params (snow_snicar_impl.hh:107-305) and snow_radiative_transfer_solver (:313-667) of
// that band.  pass 0 = direct beam (flg_slr_in 1), 1 = diffuse (2).  Returns the band albedo and fl[i] = absorbed flux
// of layer i (i >= 5 - NL) and of the ground (fl[5]), after the reference's underflow clamp.
// The ten (pass, band) solves of a column are independent until snow_albedo_radiation_factor sums the bands, so they
// run on ten lanes; one lane then holds only one band of per-layer state (~150 VGPRs instead of 250-450 for a whole
// column), which lets three waves share a SIMD and hide the latency of the long dependent fp64 chains.
template <int NL>
__device__ __forceinline__ void snicar_band(const DevState* __restrict__ S, const int64_t c, const int64_t ld, const int pass,
                                            const int bnd, const double mu_not, const int snl, const double h2osno,
                                            const double albsoi_b, double& albedo, double (&fl)[6], uint32_t& err)
{
  constexpr int snl_top = 5 - NL;
  const gptr<const double> tab = S->snicar;
  const double difgauspt[8] = {0.9894009, 0.9445750, 0.8656312, 0.7554044, 0.6178762, 0.4580168, 0.2816036, 0.0950125};
  const double difgauswt[8] = {0.0271525, 0.0622535, 0.0951585, 0.1246290, 0.1495960, 0.1691565, 0.1826034, 0.1894506};
  const double puny = 1.0e-11;
  const double c0 = 0.0, c1 = 1.0, c3 = 3.0, c4 = 4.0, cp5 = 0.5, cp75 = 0.75, c1p5 = 1.5, trmin = 0.001;
  // incident irradiance (:88-98)
  const double flx_slrd = (pass == 0) ? 1.0 / (mu_not * ELM_PI) : 0.0;
  const double flx_slri = (pass == 0) ? 0.0 : 1.0;
  const gptr<const double> tsnw = tab + ((pass == 0) ? SN_SNW_DRC : SN_SNW_DFS);

  // aerosol species 2..7 of this band (:179-210); species 0/1 (BC) depend on the layer
  double ss_aer[8], asm_aer[8], ext_aer[8];
#pragma unroll
  for (int s = 0; s < 6; s++) {
    ss_aer[2 + s] = tab[SN_OC1 + s * SN_AER_STRIDE + 0 + bnd];
    asm_aer[2 + s] = tab[SN_OC1 + s * SN_AER_STRIDE + 5 + bnd];
    ext_aer[2 + s] = tab[SN_OC1 + s * SN_AER_STRIDE + 10 + bnd];
  }
  // ---- snow_aerosol_mie_params for this band: delta-transformed layer optics (:215-305)
  double g_star[5], omega_star[5], tau_star[5];
#pragma unroll
  for (int i = 0; i < 5; i++) {
    g_star[i] = omega_star[i] = tau_star[i] = 0.0;
    if (i >= snl_top) {
      // local copies of the layer (snow_snicar::init_timestep :9-60); snl == 0 is the fictitious fresh-snow layer
      double ice_i, liq_i;
      int rds_i;
      if (snl == 0) {  // only possible for NL == 1
        ice_i = h2osno;
        liq_i = 0.0;
        rds_i = (int)round(SNW_RDS_MIN);
      } else {
        liq_i = LV(h2osoi_liq, i);
        ice_i = LV(h2osoi_ice, i);
        rds_i = (int)round(LV(snw_rds, i));
      }
      if (rds_i < SN_RDS_MIN_TBL || rds_i > SN_RDS_MAX_TBL) {
        err |= ELMK_ERR_SNICAR_RDS;  // the reference throws (:74-78); clamp so the table gather stays in range
        rds_i = rds_i < SN_RDS_MIN_TBL ? SN_RDS_MIN_TBL : SN_RDS_MAX_TBL;
      }
      // aerosol mass concentrations (surface_albedo_impl.hh:141-150): OC species 2,3 are ignored; bands 3 and 4 see
      // no aerosol (:150-156)
      double mss[8];
      mss[0] = LV(cnc_bcphi, i);
      mss[1] = LV(cnc_bcpho, i);
      mss[2] = 0.0;
      mss[3] = 0.0;
      mss[4] = LV(cnc_dst1, i);
      mss[5] = LV(cnc_dst2, i);
      mss[6] = LV(cnc_dst3, i);
      mss[7] = LV(cnc_dst4, i);
      if (bnd == 4 || bnd == 3) {
#pragma unroll
        for (int j = 0; j < 8; j++) mss[j] = 0.0;
      }
      const int rds_idx = rds_i - SN_RDS_MIN_TBL;
      const double ss_snw = tsnw[(0 * 5 + bnd) * ELMK_MIE_N + rds_idx];
      const double asm_snw = tsnw[(1 * 5 + bnd) * ELMK_MIE_N + rds_idx];
      const double ext_snw = tsnw[(2 * 5 + bnd) * ELMK_MIE_N + rds_idx];
      int idx_ice;
      if (rds_i < 125) {
        const double tmp1 = rds_i / 50;  // integer division as in the reference (:250)
        idx_ice = (int)round(tmp1) - 1;
      } else if (rds_i < 175) {
        idx_ice = 1;
      } else {
        const double tmp1 = (rds_i / 250) + 2;  // integer division (:255)
        idx_ice = (int)round(tmp1) - 1;
      }
      const int idx_ncl = 1;  // round(100/50) - 1 for both within-ice and external BC (:260-261), inside [0, 9]
      if (idx_ice < 0) idx_ice = 0;
      if (idx_ice > 7) idx_ice = 7;
      const double enh_fct = tab[SN_BCENH + (idx_ice * 10 + idx_ncl) * 5 + bnd];
      ss_aer[0] = tab[SN_BC1 + 0 + idx_ncl * 5 + bnd];
      asm_aer[0] = tab[SN_BC1 + 50 + idx_ncl * 5 + bnd];
      ext_aer[0] = tab[SN_BC1 + 100 + idx_ncl * 5 + bnd] * enh_fct;
      ss_aer[1] = tab[SN_BC2 + 0 + idx_ncl * 5 + bnd];
      asm_aer[1] = tab[SN_BC2 + 50 + idx_ncl * 5 + bnd];
      ext_aer[1] = tab[SN_BC2 + 100 + idx_ncl * 5 + bnd];

      const double L_snw = ice_i + liq_i;
      const double tau_snw = L_snw * ext_snw;
      double tau_sum = 0.0, omega_sum = 0.0, g_sum = 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const double L_aer = L_snw * mss[j];
        const double tau_aer = L_aer * ext_aer[j];
        tau_sum += tau_aer;
        omega_sum += (tau_aer * ss_aer[j]);
        g_sum += (tau_aer * ss_aer[j] * asm_aer[j]);
      }
      const double tau = tau_sum + tau_snw;
      const double omega = (1.0 / tau) * (omega_sum + (ss_snw * tau_snw));
      const double g = (1.0 / (tau * omega)) * (g_sum + (asm_snw * ss_snw * tau_snw));
      g_star[i] = g / (1.0 + g);
      omega_star[i] = ((1.0 - elmk_sq(g)) * omega) / (1.0 - (omega * elmk_sq(g)));
      tau_star[i] = (1.0 - (omega * elmk_sq(g))) * tau;
    }
  }

  // ---- snow_radiative_transfer_solver for this band (:384-667)
  double trndir[6], trntdr[6], trndif[6], rdndif[6];
  double rdir[5], rdif_a[5], tdir[5], tdif_a[5], trnlay[5];  // rdif_b == rdif_a, tdif_b == tdif_a (:489-490)
#pragma unroll
  for (int i = 0; i < 6; i++) {
    trndir[i] = c0;
    trntdr[i] = c0;
    trndif[i] = c0;
    rdndif[i] = c0;
  }
#pragma unroll
  for (int i = 0; i < 5; i++) {
    if (i == snl_top) {
      trndir[i] = c1;
      trntdr[i] = c1;
      trndif[i] = c1;
      rdndif[i] = c0;
    }
    rdir[i] = c0;
    rdif_a[i] = c0;
    tdir[i] = c0;
    tdif_a[i] = c0;
    trnlay[i] = c0;
    if (i >= snl_top) {
      if (trntdr[i] > trmin) {
        const double ts = tau_star[i];
        const double ws = omega_star[i];
        const double gs = g_star[i];
        const double lm = sqrt(c3 * (c1 - ws) * (c1 - ws * gs));
        const double ue = c1p5 * (c1 - ws * gs) / lm;
        const double extins = dmax(SN_EXP_MIN, elmk_exp(-lm * ts));
        const double ne = ((ue + c1) * (ue + c1) / extins) - ((ue - c1) * (ue - c1) * extins);
        const double R1 = (elmk_sq(ue) - c1) * (c1 / extins - extins) / ne;
        const double T1 = c4 * ue / ne;
        trnlay[i] = dmax(SN_EXP_MIN, elmk_exp(-ts / mu_not));
        double alp = cp75 * ws * mu_not * ((c1 + gs * (c1 - ws)) / (c1 - lm * lm * mu_not * mu_not));
        double gam = cp5 * ws * ((c1 + c3 * gs * (c1 - ws) * mu_not * mu_not) / (c1 - lm * lm * mu_not * mu_not));
        double apg = alp + gam;
        double amg = alp - gam;
        rdir[i] = apg * R1 + amg * (T1 * trnlay[i] - c1);
        tdir[i] = apg * T1 + (amg * R1 - apg + c1) * trnlay[i];
        double swt = c0, smr = c0, smt = c0;
#pragma unroll
        for (int ng = 0; ng < 8; ++ng) {
          const double mu = difgauspt[ng];
          const double gwt = difgauswt[ng];
          swt = swt + mu * gwt;
          const double trn = dmax(SN_EXP_MIN, elmk_exp(-ts / mu));
          alp = cp75 * ws * mu * ((c1 + gs * (c1 - ws)) / (c1 - lm * lm * mu * mu));
          gam = cp5 * ws * ((c1 + c3 * gs * (c1 - ws) * mu * mu) / (c1 - lm * lm * mu * mu));
          apg = alp + gam;
          amg = alp - gam;
          const double rdr = apg * R1 + amg * T1 * trn - amg;
          const double tdr = apg * T1 + amg * R1 * trn - apg * trn + trn;
          smr = smr + mu * rdr * gwt;
          smt = smt + mu * tdr * gwt;
        }
        rdif_a[i] = smr / swt;
        tdif_a[i] = smt / swt;
      }
      trndir[i + 1] = trndir[i] * trnlay[i];
      const double refkm1 = c1 / (c1 - rdndif[i] * rdif_a[i]);
      const double tdrrdir = trndir[i] * rdir[i];
      const double tdndif = trntdr[i] - trndir[i];
      trntdr[i + 1] = trndir[i] * tdir[i] + (tdndif + tdrrdir * rdndif[i]) * refkm1 * tdif_a[i];
      rdndif[i + 1] = rdif_a[i] + (tdif_a[i] * rdndif[i] * refkm1 * tdif_a[i]);
      trndif[i + 1] = trndif[i] * refkm1 * tdif_a[i];
    }
  }

  // upward sweep from the ground interface (:506-524)
  double rupdir[6], rupdif[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    rupdir[i] = c0;
    rupdif[i] = c0;
  }
  rupdir[5] = albsoi_b;  // albsoi(VIS) for band 0, albsoi(NIR) for the other bands
  rupdif[5] = albsoi_b;
#pragma unroll
  for (int i = 4; i >= 0; --i) {
    if (i >= snl_top) {
      const double refkp1 = c1 / (c1 - rdif_a[i] * rupdif[i + 1]);
      rupdir[i] = rdir[i] + (trnlay[i] * rupdir[i + 1] + (tdir[i] - trnlay[i]) * rupdif[i + 1]) * refkp1 * tdif_a[i];
      rupdif[i] = rdif_a[i] + tdif_a[i] * rupdif[i + 1] * refkp1 * tdif_a[i];
    }
  }

  // net interface fluxes (:540-569); dftmp = dfdir (direct pass) or dfdif (diffuse pass) (:571-591)
  double dftmp[6];
  double F_sfc_pls = 0.0;
  albedo = 0.0;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    dftmp[i] = c0;
    if (i >= snl_top) {
      const double refk = c1 / (c1 - rdndif[i] * rupdif[i]);
      if (pass == 0) {
        double dfdir = trndir[i] + (trntdr[i] - trndir[i]) * (c1 - rupdif[i]) * refk -
                       trndir[i] * rupdir[i] * (c1 - rdndif[i]) * refk;
        if (dfdir < puny) dfdir = c0;
        dftmp[i] = dfdir;
      } else {
        double dfdif = trndif[i] * (c1 - rupdif[i]) * refk;
        if (dfdif < puny) dfdif = c0;
        dftmp[i] = dfdif;
      }
      if (i == snl_top) {
        if (pass == 0) {
          albedo = rupdir[i];
          F_sfc_pls = (trndir[i] * rupdir[i] + (trntdr[i] - trndir[i]) * rupdif[i]) * refk;
        } else {
          albedo = rupdif[i];
          F_sfc_pls = trndif[i] * rupdif[i] * refk;
        }
      }
    }
  }

  // absorbed flux per layer + ground (:594-650)
  double F_abs_sum = 0.0;
#pragma unroll
  for (int i = 0; i < 5; i++) {
    fl[i] = 0.0;
    if (i >= snl_top) {
      const double F_abs = dftmp[i] - dftmp[i + 1];
      fl[i] = F_abs;
      if (F_abs < -0.00001) err |= ELMK_ERR_SNICAR_NEG_ABS;
      F_abs_sum = F_abs_sum + F_abs;
    }
  }
  const double F_btm_net = dftmp[5];
  fl[5] = F_btm_net;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    if (i >= snl_top && fl[i] < 0.0) fl[i] = 0.0;  // underflow clamp (:640-644)
  }
  const double energy_sum = (mu_not * ELM_PI * flx_slrd) + flx_slri - (F_abs_sum + F_btm_net + F_sfc_pls);
  if (fabs(energy_sum) > 0.00001) err |= ELMK_ERR_SNICAR_ENERGY;
  if (albedo > 1.0) err |= ELMK_ERR_SNICAR_ALBEDO;
}

// snow_albedo_radiation_factor (:673-757) for one pass of one column, from the five band results held by five
// consecutive lanes (band b at lane g0 + b): VIS is band 0, NIR the flux-weighted sum of bands 1..4 in band order.
// Every lane of the group computes the same values; the caller lets one of them store.
// band_albedo(b) / band_flux(b, i): how this lane reads band b's results (shuffles from the lanes beside it, or LDS)
template <int NL, class FA, class FF>
__device__ __forceinline__ void snicar_combine_from(const int pass, const double mu_not, const int rds_top, const FA band_albedo,
                                                    const FF band_flux, SnowOut& out)
{
  constexpr int snl_top = 5 - NL;
  // 5-band flux weights (:710-723)
  const double w1 = (pass == 0) ? 0.49352158521175 : 0.58581507618433;
  const double w2 = (pass == 0) ? 0.18099494230665 : 0.20156903770812;
  const double w3 = (pass == 0) ? 0.12094898498813 : 0.10917889346386;
  const double w4 = (pass == 0) ? 0.20453448749347 : 0.10343699264369;
  const double flx_wgt[5] = {1.0, w1, w2, w3, w4};
  double alb_nir_sum = 0.0, wgt_sum = 0.0;
  double nir_sum[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  out.alb[0] = band_albedo(0);
#pragma unroll
  for (int i = 0; i < 6; i++) {
    out.fabs_[i][0] = band_flux(0, i);  // zero wherever the solver does not write (i < snl_top)
    out.fabs_[i][1] = 0.0;
  }
#pragma unroll
  for (int b = 1; b < 5; b++) {
    alb_nir_sum += flx_wgt[b] * band_albedo(b);
    wgt_sum += flx_wgt[b];
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const double f = band_flux(b, i);
      if (i >= snl_top) nir_sum[i] += flx_wgt[b] * f;
    }
  }
  out.alb[1] = alb_nir_sum / wgt_sum;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    if (i >= snl_top) out.fabs_[i][1] = nir_sum[i] / wgt_sum;
  }
  // near-IR direct albedo/absorption adjustment at high solar zenith angle (:748-757)
  if (pass == 0 && mu_not < 0.2588) {
    const double sza_c1 = 0.085730 + (-0.630883) * mu_not + 1.303723 * elmk_sq(mu_not);
    const double sza_c0 = 1.467291 + (-3.338043) * mu_not + 6.807489 * elmk_sq(mu_not);
    const double sza_factor = sza_c1 * (elmk_log10(rds_top * 1.0) - 6.0) + sza_c0;
    const double flx_sza_adjust = out.alb[1] * (sza_factor - 1.0) * wgt_sum;
    out.alb[1] *= sza_factor;
    out.fabs_[snl_top][1] -= flx_sza_adjust;
  }
}

// ... from the five band results held by five consecutive lanes (band b at lane g0 + b)
template <int NL>
__device__ __forceinline__ void snicar_combine(const int g0, const int pass, const double mu_not, const int rds_top,
                                               const double albedo, const double (&fl)[6], SnowOut& out)
{
  snicar_combine_from<NL>(
      pass, mu_not, rds_top, [&](const int b) { return __shfl(albedo, g0 + b, 64); },
      [&](const int b, const int i) { return __shfl(fl[i], g0 + b, 64); }, out);
}

// the work of one workgroup of k_alb_snicar<NL>: workgroup `block` of `nblocks` (a kernel that hosts other work beside SNICAR
// numbers its SNICAR workgroups itself: k_fz_snicar_pre, k_canopy_fluxes.hip)
template <int NL>
__device__ __forceinline__ void snicar_workgroup(const DevState* __restrict__ S, const uint32_t block, const uint32_t nblocks)
{
  uint32_t count = ELMK_LIST_COUNT(S, LIST_ALB_0 + NL);
  if ((int64_t)count > S->ld) count = (uint32_t)S->ld;  // (a list never holds more than every column: block_classify_append)
  // nothing in the queue for this workgroup (the whole launch, when no column has NL layers): leave before the table copy
  if ((uint64_t)block * (blockDim.x >> 6) * 6u >= count) return;
  elmk_math_lds_init<false>();
  const int64_t ld = S->ld;
  const gptr<const int32_t> list = S->lists + (int64_t)(LIST_ALB_0 + NL) * ld;
  constexpr int snl_top = NLEVSNO - NL;
  const int lane = threadIdx.x & 63;
  const int slot = lane / 10, task = lane - slot * 10;  // slot 6 (lanes 60..63): no column
  const int pass = task / 5, bnd = task - pass * 5;
  const int g0 = lane - bnd;  // first lane of this (column, pass) group
  const uint32_t nwaves = nblocks * (blockDim.x >> 6);
  for (uint32_t w = block * (blockDim.x >> 6) + (threadIdx.x >> 6); (uint64_t)w * 6u < count; w += nwaves) {
    const uint32_t q = w * 6u + (uint32_t)slot;
    const bool valid = slot < 6 && q < count;
    const int64_t c = list[valid ? q : w * 6u];  // lanes without a column shadow the wave's first one and store nothing
    uint32_t err = 0;
    const double mu_not = dmax(S->coszen[c], 0.01);
    const int snl = S->snl[c];
    const double h2osno = S->h2osno[c];
    const double albsoi_b = LV(albsoi, bnd == 0 ? 0 : 1);
    double albedo, fl[6];
    snicar_band<NL>(S, c, ld, pass, bnd, mu_not, snl, h2osno, albsoi_b, albedo, fl, err);
    int rds_top = (snl == 0) ? (int)round(SNW_RDS_MIN) : (int)round(LV(snw_rds, snl_top));
    rds_top = rds_top < SN_RDS_MIN_TBL ? SN_RDS_MIN_TBL : (rds_top > SN_RDS_MAX_TBL ? SN_RDS_MAX_TBL : rds_top);
    SnowOut out;
    snicar_combine<NL>(g0, pass, mu_not, rds_top, albedo, fl, out);
    if (valid && bnd == 0) {
      const gptr<double> o = S->alb_snow + (int64_t)(pass * 14) * ld + c;
      sc_st<2>(o, out.alb[0]);
      sc_st<2>(o + ld, out.alb[1]);
#pragma unroll
      for (int i = 0; i < 6; i++) {
        sc_st<2>(o + (int64_t)(2 + 2 * i) * ld, out.fabs_[i][0]);
        sc_st<2>(o + (int64_t)(3 + 2 * i) * ld, out.fabs_[i][1]);
      }
    }
    if (valid && err) atomicOr(ELMK_GENERIC(&S->err_flags[c]), err);
  }
}

// Measured and dropped in round 3 (profiles/r03_snicar_band_per_wave_ab.txt): one BAND per wave for the packs of two or more
// layers (a 320-thread workgroup = 5 bands x 64 columns of one pass, results combined through LDS).  It removes the divergence
// between the visible lanes, which work through every layer, and the near-infrared lanes, whose layers go dark (trntdr <=
// trmin, :414) early - VALU lane utilisation of k_alb_snicar<5> is 40 % - but at 188-236 VGPRs only one five-wave workgroup
// fits a CU, every wave gathers its own copy of the layer inputs and there is a barrier per 64 columns: 1.4-1.6 x slower.
// A second variant kept the product's occupancy - visible band and near-infrared bands in different waves (32 columns x 2
// passes of band 0; 8 columns x 2 passes x bands 1..4), no LDS, no barrier: -8 % .. +2 %, because a near-infrared wave still runs
// as deep as its deepest band-1 lane and there are more of them.  The ten-lane form stays.

}  // namespace elmk
