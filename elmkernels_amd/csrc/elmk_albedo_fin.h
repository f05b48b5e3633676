// elmk_albedo_fin.h - stage 3 of kokkos_albedo_snicar for one column (surface_albedo_impl.hh: ground_albedo :155-167,
// flux_absorption_factor :171-211, two_stream_solver :323-687 with nlevcan == 1; for a column without sun the values
// surface_albedo::init_timestep leaves, :90-151, and snow_albedo_radiation_factor's "no sun" branch, snow_snicar_impl.hh:758-765),
// in three pieces that are ONE piece of source for two users:
//   * k_alb_final (k_albedo_snicar.hip) calls them in the reference's order and every output goes to the state;
//   * the fused step's k_fz_stream (k_canopy_fluxes.hip) calls them around surface_radiation's body and ALSO keeps the outputs
//     in registers (AlbFwd), so kokkos_surface_radiation reads none of the 56 doubles back that this stage has just written.
// Every output is STORED BY ALL LANES TOGETHER: a wave that holds sunlit and dark columns would otherwise write every 128-byte
// line twice, half of it each time (measured: 904 instead of 490 bytes per column written on the fixture-tiled state).
#pragma once
#include "elmk_dev.h"
#include "elmk_albedo_col.h"

namespace elmk {

// the outputs of this stage that kokkos_surface_radiation reads (surface_radiation_kokkos.cc:7-97)
struct AlbFwd {
  double albsod[2], albsoi[2], albsnd[2], albsni[2], albgrd[2], albgri[2];
  double albd[2], albi[2], ftdd[2], ftid[2], ftii[2], fabd[2], fabi[2];
  double fsun_z, fabd_sun_z, fabd_sha_z, fabi_sun_z, fabi_sha_z;
};

// ---- ground_albedo (:155-167) and the soil / snow albedos as the wrapper leaves them
__device__ __forceinline__ void alb_ground(const DevState* __restrict__ S, const int64_t c, const int64_t ld, const bool day,
                                           const double frac_sno, const double (&albsod)[2], const double (&albsoi)[2],
                                           const double (&sd_alb)[2], const double (&si_alb)[2], AlbFwd& a)
{
#pragma unroll
  for (int ib = 0; ib < 2; ib++) {
    double o_sod = 0.0, o_soi = 0.0, o_snd = 0.0, o_sni = 0.0, g_d = 0.0, g_i = 0.0;
    if (day) {
      g_d = albsod[ib] * (1.0 - frac_sno) + sd_alb[ib] * frac_sno;
      g_i = albsoi[ib] * (1.0 - frac_sno) + si_alb[ib] * frac_sno;
      o_sod = albsod[ib];
      o_soi = albsoi[ib];
      o_snd = sd_alb[ib];
      o_sni = si_alb[ib];
    }
    a.albsod[ib] = o_sod;
    a.albsoi[ib] = o_soi;
    a.albsnd[ib] = o_snd;
    a.albsni[ib] = o_sni;
    a.albgrd[ib] = g_d;
    a.albgri[ib] = g_i;
    LV(albsod, ib) = o_sod;
    LV(albsoi, ib) = o_soi;
    LV(albsnd, ib) = o_snd;
    LV(albsni, ib) = o_sni;
    LV(albgrd, ib) = g_d;
    LV(albgri, ib) = g_i;
  }
}

// ---- flux_absorption_factor (:171-211, subgridflag == 1), level i of the four arrays; sdf / sif: flx_abs(i, {VIS, NIR}) of the
// direct and the diffuse SNICAR pass
__device__ __forceinline__ void alb_flux_abs_level(const Land& L, const bool day, const double frac_sno, const double (&albsod)[2],
                                                   const double (&albsoi)[2], const double (&sd_alb)[2], const double (&si_alb)[2],
                                                   const double (&sdf)[2], const double (&sif)[2], double& dv, double& dn, double& iv,
                                                   double& in)
{
  dv = dn = iv = in = 0.0;
  if (day) {
    if (L.ltype == istdlak) {
      dv = sdf[0] * frac_sno + ((1.0 - frac_sno) * (1.0 - albsod[0]) * (sdf[0] / (1.0 - sd_alb[0])));
      iv = sif[0] * frac_sno + ((1.0 - frac_sno) * (1.0 - albsoi[0]) * (sif[0] / (1.0 - si_alb[0])));
      dn = sdf[1] * frac_sno + ((1.0 - frac_sno) * (1.0 - albsod[1]) * (sdf[1] / (1.0 - sd_alb[1])));
      in = sif[1] * frac_sno + ((1.0 - frac_sno) * (1.0 - albsoi[1]) * (sif[1] / (1.0 - si_alb[1])));
    } else {
      dv = sdf[0] * (1.0 - sd_alb[0]);
      iv = sif[0] * (1.0 - si_alb[0]);
      dn = sdf[1] * (1.0 - sd_alb[1]);
      in = sif[1] * (1.0 - si_alb[1]);
    }
  }
}

// ---- two_stream_solver (:323-687), nlevcan == 1; reads a.albgrd / a.albgri, fills the rest of a and stores it
__device__ __forceinline__ void alb_two_stream(const DevState* __restrict__ S, const int64_t c, const int64_t ld, const Land& L,
                                               const bool day, const double coszen, const double elai, const double esai,
                                               double vcmaxcintsun, double vcmaxcintsha, AlbFwd& a)
{
  double (&albd)[2] = a.albd, (&albi)[2] = a.albi, (&ftdd)[2] = a.ftdd, (&ftid)[2] = a.ftid, (&ftii)[2] = a.ftii, (&fabd)[2] = a.fabd,
         (&fabi)[2] = a.fabi;
  double fabi_sun[2], fabi_sha[2];
  double& fsun_z = a.fsun_z;
  double &fabd_sun_z = a.fabd_sun_z, &fabd_sha_z = a.fabd_sha_z, &fabi_sun_z = a.fabi_sun_z, &fabi_sha_z = a.fabi_sha_z;
  fsun_z = fabd_sun_z = fabd_sha_z = fabi_sun_z = fabi_sha_z = 0.0;
  const double (&albgrd)[2] = a.albgrd, (&albgri)[2] = a.albgri;
  const bool soilcrop = (L.ltype == istsoil || L.ltype == istcrop);
  if (!day) {  // init_timestep's values (:117-135)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      fabd[ib] = 0.0;
      fabi[ib] = 0.0;
      fabi_sun[ib] = 0.0;
      fabi_sha[ib] = 0.0;
      ftdd[ib] = 0.0;
      ftid[ib] = 0.0;
      ftii[ib] = 0.0;
      albd[ib] = 1.0;
      albi[ib] = 1.0;
    }
  } else if (soilcrop && (elai + esai) > 0.0) {  // vegsol
    const double* __restrict__ A = S->pft_alb[S->vtype[c]];  // rhol[2] rhos[2] taul[2] taus[2] xl
    const double t_veg = S->t_veg[c], fwet = S->fwet[c];
    const double omegas[2] = {0.8, 0.4};
    const double betads = 0.5, betais = 0.5;
    const double wl = elai / dmax(elai + esai, SA_MPE);
    const double ws = esai / dmax(elai + esai, SA_MPE);
    const double cosz = dmax(0.001, coszen);
    double chil = dmin(dmax(A[8], -0.4), 0.6);
    if (fabs(chil) <= 0.01) chil = 0.01;
    const double phi1 = 0.5 - 0.633 * chil - 0.330 * chil * chil;
    const double phi2 = 0.877 * (1.0 - 2.0 * phi1);
    const double gdir = phi1 + phi2 * cosz;
    const double twostext = gdir / cosz;
    const double avmu = (1.0 - phi1 / phi2 * elmk_log((phi1 + phi2) / phi1)) / phi2;
    const double temp0 = gdir + phi2 * cosz;
    const double temp1 = phi1 * cosz;
    const double temp2 = (1.0 - temp1 / temp0 * elmk_log((temp1 + temp0) / temp1));
#pragma unroll
    for (int ib = 0; ib < 2; ib++) {
      const double rho = dmax(A[0 + ib] * wl + A[2 + ib] * ws, SA_MPE);
      const double tau = dmax(A[4 + ib] * wl + A[6 + ib] * ws, SA_MPE);
      const double omegal = rho + tau;
      const double asu = 0.5 * omegal * gdir / temp0 * temp2;
      const double betadl = (1.0 + avmu * twostext) / (omegal * avmu * twostext) * asu;
      const double betail = 0.5 * ((rho + tau) + (rho - tau) * elmk_sq(((1.0 + chil) / 2.0))) / omegal;
      double tmp0, tmp1, tmp2;
      if (t_veg > TFRZ) {
        tmp0 = omegal;
        tmp1 = betadl;
        tmp2 = betail;
      } else {
        tmp0 = (1.0 - fwet) * omegal + fwet * omegas[ib];
        tmp1 = ((1.0 - fwet) * omegal * betadl + fwet * omegas[ib] * betads) / tmp0;
        tmp2 = ((1.0 - fwet) * omegal * betail + fwet * omegas[ib] * betais) / tmp0;
      }
      const double omega = tmp0;
      const double betad = tmp1;
      const double betai = tmp2;
      const double b = 1.0 - omega + omega * betai;
      const double c1 = omega * betai;
      tmp0 = avmu * twostext;
      const double d = tmp0 * omega * betad;
      const double f = tmp0 * omega * (1.0 - betad);
      tmp1 = b * b - c1 * c1;
      const double h = sqrt(tmp1) / avmu;
      const double sigma = tmp0 * tmp0 - tmp1;
      const double p1 = b + avmu * h;
      const double p2 = b - avmu * h;
      const double p3 = b + tmp0;
      const double p4 = b - tmp0;
      double t1 = dmin(h * (elai + esai), 40.0);
      const double s1 = elmk_exp(-t1);
      t1 = dmin(twostext * (elai + esai), 40.0);
      const double s2 = elmk_exp(-t1);
      // direct beam
      double u1 = b - c1 / albgrd[ib];
      double u2 = b - c1 * albgrd[ib];
      const double u3 = f + c1 * albgrd[ib];
      tmp2 = u1 - avmu * h;
      double tmp3 = u1 + avmu * h;
      double d1 = p1 * tmp2 / s1 - p2 * tmp3 * s1;
      double tmp4 = u2 + avmu * h;
      double tmp5 = u2 - avmu * h;
      double d2 = tmp4 / s1 - tmp5 * s1;
      const double h1 = -d * p4 - c1 * f;
      const double tmp6 = d - h1 * p3 / sigma;
      const double tmp7 = (d - c1 - h1 / sigma * (u1 + tmp0)) * s2;
      const double h2 = (tmp6 * tmp2 / s1 - p2 * tmp7) / d1;
      const double h3 = -(tmp6 * tmp3 * s1 - p1 * tmp7) / d1;
      const double h4 = -f * p3 - c1 * d;
      const double tmp8 = h4 / sigma;
      const double tmp9 = (u3 - tmp8 * (u2 - tmp0)) * s2;
      const double h5 = -(tmp8 * tmp4 / s1 + tmp9) / d2;
      const double h6 = (tmp8 * tmp5 * s1 + tmp9) / d2;
      albd[ib] = h1 / sigma + h2 + h3;
      ftid[ib] = h4 * s2 / sigma + h5 * s1 + h6 / s1;
      ftdd[ib] = s2;
      fabd[ib] = 1.0 - albd[ib] - (1.0 - albgrd[ib]) * ftdd[ib] - (1.0 - albgri[ib]) * ftid[ib];
      double a1 = h1 / sigma * (1.0 - s2 * s2) / (2.0 * twostext) + h2 * (1.0 - s2 * s1) / (twostext + h) +
                  h3 * (1.0 - s2 / s1) / (twostext - h);
      double a2 = h4 / sigma * (1.0 - s2 * s2) / (2.0 * twostext) + h5 * (1.0 - s2 * s1) / (twostext + h) +
                  h6 * (1.0 - s2 / s1) / (twostext - h);
      const double fabd_sun = (1.0 - omega) * (1.0 - s2 + 1.0 / avmu * (a1 + a2));
      const double fabd_sha = fabd[ib] - fabd_sun;  // wrapper-local in the reference (albedo_kokkos.cc:27-28)
      // diffuse
      u1 = b - c1 / albgri[ib];
      u2 = b - c1 * albgri[ib];
      tmp2 = u1 - avmu * h;
      tmp3 = u1 + avmu * h;
      d1 = p1 * tmp2 / s1 - p2 * tmp3 * s1;
      tmp4 = u2 + avmu * h;
      tmp5 = u2 - avmu * h;
      d2 = tmp4 / s1 - tmp5 * s1;
      const double h7 = (c1 * tmp2) / (d1 * s1);
      const double h8 = (-c1 * tmp3 * s1) / d1;
      const double h9 = tmp4 / (d2 * s1);
      const double h10 = (-tmp5 * s1) / d2;
      albi[ib] = h7 + h8;
      ftii[ib] = h9 * s1 + h10 / s1;
      fabi[ib] = 1.0 - albi[ib] - (1.0 - albgri[ib]) * ftii[ib];
      a1 = h7 * (1.0 - s2 * s1) / (twostext + h) + h8 * (1.0 - s2 / s1) / (twostext - h);
      a2 = h9 * (1.0 - s2 * s1) / (twostext + h) + h10 * (1.0 - s2 / s1) / (twostext - h);
      fabi_sun[ib] = (1.0 - omega) / avmu * (a1 + a2);
      fabi_sha[ib] = fabi[ib] - fabi_sun[ib];
      if (ib == 0) {
        fsun_z = (1.0 - s2) / t1;
        const double laisum = elai + esai;
        fabd_sun_z = fabd_sun / (fsun_z * laisum);
        fabi_sun_z = fabi_sun[ib] / (fsun_z * laisum);
        fabd_sha_z = fabd_sha / ((1.0 - fsun_z) * laisum);
        fabi_sha_z = fabi_sha[ib] / ((1.0 - fsun_z) * laisum);
        const double extkb = twostext;
        vcmaxcintsun = (1.0 - elmk_exp(-(SA_EXTKN + extkb) * elai)) / (SA_EXTKN + extkb);
        vcmaxcintsha = (1.0 - elmk_exp(-SA_EXTKN * elai)) / SA_EXTKN - vcmaxcintsun;
        if (elai > 0.0) {
          vcmaxcintsun = vcmaxcintsun / (fsun_z * elai);
          vcmaxcintsha = vcmaxcintsha / ((1.0 - fsun_z) * elai);
        } else {
          vcmaxcintsun = 0.0;
          vcmaxcintsha = 0.0;
        }
      }
    }
  } else {  // novegsol (:672-686)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib) {
      fabd[ib] = 0.0;
      fabi[ib] = 0.0;
      fabi_sun[ib] = 0.0;
      fabi_sha[ib] = 0.0;
      ftdd[ib] = 1.0;
      ftid[ib] = 0.0;
      ftii[ib] = 1.0;
      albd[ib] = albgrd[ib];
      albi[ib] = albgri[ib];
    }
  }
#pragma unroll
  for (int ib = 0; ib < 2; ib++) {
    LV(albd, ib) = albd[ib];
    LV(albi, ib) = albi[ib];
    LV(ftdd, ib) = ftdd[ib];
    LV(ftid, ib) = ftid[ib];
    LV(ftii, ib) = ftii[ib];
    LV(fabd, ib) = fabd[ib];
    LV(fabi, ib) = fabi[ib];
    LV(fabi_sun, ib) = fabi_sun[ib];
    LV(fabi_sha, ib) = fabi_sha[ib];
  }
  S->vcmaxcintsun[c] = vcmaxcintsun;
  S->vcmaxcintsha[c] = vcmaxcintsha;
  S->fsun_z[c] = fsun_z;
  S->fabd_sun_z[c] = fabd_sun_z;
  S->fabd_sha_z[c] = fabd_sha_z;
  S->fabi_sun_z[c] = fabi_sun_z;
  S->fabi_sha_z[c] = fabi_sha_z;
}

// What the column brings to this stage: sun or not, the init_timestep values of the leaf-to-canopy scaling coefficients
// (:100-108; overwritten by two_stream where vegetated), the soil albedos stage 1 left in the state and the band albedos of the
// two SNICAR passes (or snow_albedo_radiation_factor's remaining branches, snow_snicar_impl.hh:758-765).  frac_sno / h2osno are
// the values the wrapper sees: the caller of the fused step hands in what they were BEFORE canopy_hydrology changed them.
struct AlbIn {
  bool day, snicar;  // snicar: the column went through SNICAR, its absorbed-flux factors are in alb_snow
  double coszen, elai, esai, frac_sno, vcmaxcintsun, vcmaxcintsha;
  double albsod[2], albsoi[2], sd_alb[2], si_alb[2];
};
__device__ __forceinline__ AlbIn alb_final_inputs(const DevState* __restrict__ S, const int64_t c, const int64_t ld, const double frac_sno_in,
                                                  const double h2osno_in)
{
  AlbIn x;
  x.coszen = S->coszen[c];
  x.elai = S->elai[c];
  x.vcmaxcintsun = 0.0;
  x.vcmaxcintsha = (1.0 - elmk_exp(-SA_EXTKN * x.elai)) / SA_EXTKN;
  if (x.elai > 0.0) {
    x.vcmaxcintsha /= x.elai;
  } else {
    x.vcmaxcintsha = 0.0;
  }
  x.day = x.coszen > 0.0;  // nothing after init_timestep runs without sun except snow_albedo_radiation_factor's "no sun" branch
  x.snicar = false;
  x.esai = 0.0;
  x.frac_sno = 0.0;
  x.albsod[0] = x.albsod[1] = x.albsoi[0] = x.albsoi[1] = 0.0;
  x.sd_alb[0] = x.sd_alb[1] = x.si_alb[0] = x.si_alb[1] = 0.0;
  if (x.day) {
    x.esai = S->esai[c];
    x.frac_sno = frac_sno_in;
    x.albsod[0] = LV(albsod, 0);  // written by stage 1
    x.albsod[1] = LV(albsod, 1);
    x.albsoi[0] = LV(albsoi, 0);
    x.albsoi[1] = LV(albsoi, 1);
    if (h2osno_in > SN_MIN_SNW) {
      const gptr<const double> o = S->alb_snow + c;
      x.snicar = true;
      x.sd_alb[0] = sc_ld<1>(o);
      x.sd_alb[1] = sc_ld<1>(o + ld);
      x.si_alb[0] = sc_ld<1>(o + (int64_t)14 * ld);
      x.si_alb[1] = sc_ld<1>(o + (int64_t)15 * ld);
    } else if (h2osno_in < SN_MIN_SNW && h2osno_in > 0.0) {
      // no snow radiative transfer: snow_albedo_radiation_factor's remaining branches (snow_snicar_impl.hh:758-765)
      x.sd_alb[0] = x.si_alb[0] = x.albsoi[0];
      x.sd_alb[1] = x.si_alb[1] = x.albsoi[1];
    }
  }
  return x;
}
// flux_absorption_factor for all six levels: the SNICAR factors come from alb_snow (zeros for a column that did not go through
// SNICAR), the four arrays go to the state and to flx[i] = {dv, dn, iv, in}
__device__ __forceinline__ void alb_flux_abs_all(const DevState* __restrict__ S, const int64_t c, const int64_t ld, const Land& L,
                                                 const AlbIn& x, double (&flx)[6][4])
{
  const gptr<const double> o = S->alb_snow + c;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double sdf[2] = {0.0, 0.0}, sif[2] = {0.0, 0.0};
    if (x.snicar) {
      sdf[0] = sc_ld<1>(o + (int64_t)(2 + 2 * i) * ld);
      sdf[1] = sc_ld<1>(o + (int64_t)(3 + 2 * i) * ld);
      sif[0] = sc_ld<1>(o + (int64_t)(16 + 2 * i) * ld);
      sif[1] = sc_ld<1>(o + (int64_t)(17 + 2 * i) * ld);
    }
    alb_flux_abs_level(L, x.day, x.frac_sno, x.albsod, x.albsoi, x.sd_alb, x.si_alb, sdf, sif, flx[i][0], flx[i][1], flx[i][2], flx[i][3]);
    LV(flx_absdv, i) = flx[i][0];
    LV(flx_absdn, i) = flx[i][1];
    LV(flx_absiv, i) = flx[i][2];
    LV(flx_absin, i) = flx[i][3];
  }
}

}  // namespace elmk
