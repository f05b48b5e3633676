// k_init_state.hip - cold-start initialisation of a column: the "init functions" lambda that ELM::initialize_kokkos_elm runs
// once per column after the input files are read (driver/kokkos/initialize_elm_kokkos.cc:373-428).  It is the producer of
// the state the hot path consumes - soil hydraulic / thermal parameters, root fractions, the initial snow mesh, soil
// temperature and water - so a driver that keeps its state on the GPU never builds it on the host.
//
//   init_topo_slope, init_melt_factor, init_micro_sigma     src/physics/init_topography_impl.hh:7-38
//   init_snow_layers                                        src/physics/init_snow_state_impl.hh:67-151
//   init_soil_hydraulics (soil_hydraulic_params, pedotransfer)   src/physics/soil_texture_hydraulic_model_impl.hh:7-123
//   init_vegrootfr, init_soil_temp, init_soilh2o_state      src/physics/init_soil_state_impl.hh:180-213, :11-54, :65-176
//   init_snow_state                                         src/physics/init_snow_state_impl.hh:11-63
//
// One thread per column, one launch; runs once, so nothing here is tuned beyond coalesced rows.  The oracle's restatement
// (oracle/elmo_physics_h.c) is pinned bit for bit against the reference's own headers; this kernel is bit-identical to it.
// Reference behaviour kept as it is (listed in the oracle file): init_snow_state zeroes snow_depth / h2osno after the
// layer mesh has been built from snow_depth; csol is written by soil index into a 20-level array that soil_temperature
// reads by level index; init_soilh2o_state's last loop overwrites the liquid / ice split of every layer.
#include "elmk_dev.h"
#include "elmk_kernels.h"

namespace elmk {

#define LV(f, lev) S->f[(int64_t)(lev) * ld + c]

namespace {
constexpr int NSOI = 10;  // nlevsoi, elm_constants.h:90
constexpr int NBED = 15;  // nlevbed :91
constexpr int NURB = 5;   // nlevurb :87
constexpr double BDSNO = 250.0;
constexpr double SECSPDAY = 86400.0;

// soil_texture_hydraulic_model_impl.hh:19-94 (pedotransfer :7-16 inlined)
__device__ __forceinline__ void soil_hydraulic_params(const double pct_sand, const double pct_clay, const double zsoi,
                                                      const double om_frac, double& watsat, double& bsw, double& sucsat,
                                                      double& watdry, double& watopt, double& watfc, double& tkmg, double& tkdry,
                                                      double& csol)
{
  const double zsapric = 0.5, pcalpha = 0.5, pcbeta = 0.139, om_tkd = 0.05, om_tkm = 0.25, om_csol = 2.5;
  watsat = 0.489 - 0.00126 * pct_sand;
  bsw = 2.91 + 0.159 * pct_clay;
  sucsat = 10.0 * elmk_pow_literal_base(10.0, (1.88 - 0.0131 * pct_sand));
  const double xksat = 0.0070556 * elmk_pow_literal_base(10.0, (-0.884 + 0.0153 * pct_sand));
  const double om_watsat = dmax(0.93 - 0.1 * (zsoi / zsapric), 0.83);
  const double om_b = dmin(2.7 + 9.3 * (zsoi / zsapric), 12.0);
  const double om_sucsat = dmin(10.3 - 0.2 * (zsoi / zsapric), 10.1);
  const double om_hksat = dmax(0.28 - 0.2799 * (zsoi / zsapric), 0.0001);

  const double bulk_den = (1.0 - watsat) * 2.7e3;
  const double tkm = (1.0 - om_frac) * (8.8 * pct_sand + 2.92 * pct_clay) / (pct_sand + pct_clay) + om_tkm * om_frac;
  watsat = (1.0 - om_frac) * watsat + om_watsat * om_frac;
  bsw = (1.0 - om_frac) * (2.91 + 0.159 * pct_clay) + om_frac * om_b;
  sucsat = (1.0 - om_frac) * sucsat + om_sucsat * om_frac;

  double perc_frac;
  if (om_frac > pcalpha) {
    const double perc_norm = 0x1.19e46a70188edp+0;  // pow(1 - pcalpha, -pcbeta): the compiler's and the libm's value agree
    perc_frac = perc_norm * elmk_pow((om_frac - pcalpha), pcbeta);
  } else {
    perc_frac = 0.0;
  }
  const double uncon_frac = (1.0 - om_frac) + (1.0 - perc_frac) * om_frac;
  double uncon_hksat;
  if (om_frac < 1.0) {
    uncon_hksat = uncon_frac / ((1.0 - om_frac) / xksat + ((1.0 - perc_frac) * om_frac) / om_hksat);
  } else {
    uncon_hksat = 0.0;
  }
  const double hksat = uncon_frac * uncon_hksat + (perc_frac * om_frac) * om_hksat;

  tkmg = elmk_pow(tkm, (1.0 - watsat));
  tkdry = ((0.135 * bulk_den + 64.7) / (2.7e3 - 0.947 * bulk_den)) * (1.0 - om_frac) + om_tkd * om_frac;
  csol = ((1.0 - om_frac) * (2.128 * pct_sand + 2.385 * pct_clay) / (pct_sand + pct_clay) + om_csol * om_frac) * 1.0e6;
  watdry = watsat * elmk_pow((316230.0 / sucsat), (-1.0 / bsw));
  watopt = watsat * elmk_pow((158490.0 / sucsat), (-1.0 / bsw));
  watfc = watsat * elmk_pow((0.1 / (hksat * SECSPDAY)), (1.0 / (2.0 * bsw + 3.0)));
}
}  // namespace

__global__ __launch_bounds__(256) void k_init_state(const DevState* __restrict__ S)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= S->ncols) return;
  const int64_t ld = S->ld;
  const Land L = S->land;

  // ---- init_topography_impl.hh:7-38
  const double topo_slope = dmax(S->topo_slope[c], 0.2);
  S->topo_slope[c] = topo_slope;
  S->n_melt[c] = (L.ltype == istice_mec) ? 10.0 : 200.0 / dmax(10.0, S->topo_std[c]);
  {
    const double slopebeta = 3.0;
    const double slope0 = 0x1.5b7209557b0edp+0;  // pow(0.4, -1.0 / 3.0): the compiler's and the libm's value agree
    S->micro_sigma[c] = elmk_pow((topo_slope + slope0), -slopebeta);
  }

  // ---- init_snow_layers (init_snow_state_impl.hh:67-151)
  int snl = S->snl[c];
  {
    const double snow_depth = S->snow_depth[c];
    double dz[NLEVSNO], z[NLEVSNO], zi[NLEVSNO + 1];
#pragma unroll
    for (int i = 0; i < NLEVSNO; i++) dz[i] = z[i] = zi[i] = SPVAL;
    zi[NLEVSNO] = LV(zisoi, NLEVSNO);
    if (!L.lakpoi) {
      if (snow_depth < 0.01) {
        snl = 0;
#pragma unroll
        for (int i = 0; i < NLEVSNO; i++) dz[i] = z[i] = zi[i] = 0.0;
        zi[NLEVSNO] = 0.0;
      } else {
        if ((snow_depth >= 0.01) && (snow_depth <= 0.03)) {
          snl = 1;
          dz[4] = snow_depth;
        } else if ((snow_depth > 0.03) && (snow_depth <= 0.04)) {
          snl = 2;
          dz[3] = snow_depth / 2.0;
          dz[4] = dz[3];
        } else if ((snow_depth > 0.04) && (snow_depth <= 0.07)) {
          snl = 2;
          dz[3] = 0.02;
          dz[4] = snow_depth - dz[3];
        } else if ((snow_depth > 0.07) && (snow_depth <= 0.12)) {
          snl = 3;
          dz[2] = 0.02;
          dz[3] = (snow_depth - 0.02) / 2.0;
          dz[4] = dz[3];
        } else if ((snow_depth > 0.12) && (snow_depth <= 0.18)) {
          snl = 3;
          dz[2] = 0.02;
          dz[3] = 0.05;
          dz[4] = snow_depth - dz[2] - dz[3];
        } else if ((snow_depth > 0.18) && (snow_depth <= 0.29)) {
          snl = 4;
          dz[1] = 0.02;
          dz[2] = 0.05;
          dz[3] = (snow_depth - dz[1] - dz[2]) / 2.0;
          dz[4] = dz[3];
        } else if ((snow_depth > 0.29) && (snow_depth <= 0.41)) {
          snl = 4;
          dz[1] = 0.02;
          dz[2] = 0.05;
          dz[3] = 0.11;
          dz[4] = snow_depth - dz[1] - dz[2] - dz[3];
        } else if ((snow_depth > 0.41) && (snow_depth <= 0.64)) {
          snl = 5;
          dz[0] = 0.02;
          dz[1] = 0.05;
          dz[2] = 0.11;
          dz[3] = (snow_depth - dz[0] - dz[1] - dz[2]) / 2.0;
          dz[4] = dz[3];
        } else if (snow_depth > 0.64) {
          snl = 5;
          dz[0] = 0.02;
          dz[1] = 0.05;
          dz[2] = 0.11;
          dz[3] = 0.23;
          dz[4] = snow_depth - dz[0] - dz[1] - dz[2] - dz[3];
        }
      }
#pragma unroll
      for (int j = NLEVSNO - 1; j >= 0; j--) {
        if (j >= NLEVSNO - snl) {
          z[j] = zi[j + 1] - 0.5 * dz[j];
          zi[j] = zi[j + 1] - dz[j];
        }
      }
    } else {
      snl = 0;
#pragma unroll
      for (int i = 0; i < NLEVSNO; i++) dz[i] = z[i] = zi[i] = 0.0;
      zi[NLEVSNO] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < NLEVSNO; i++) {
      LV(dz, i) = dz[i];
      LV(zsoi, i) = z[i];
      LV(zisoi, i) = zi[i];
    }
    LV(zisoi, NLEVSNO) = zi[NLEVSNO];
    S->snl[c] = snl;
  }

  // ---- init_soil_hydraulics (soil_texture_hydraulic_model_impl.hh:98-123)
  double watsat[NLEVGRND];
  {
    const double organic_max = S->organic_max;
    const double sand_b = LV(pct_sand, NSOI - 1), clay_b = LV(pct_clay, NSOI - 1);
#pragma unroll 1
    for (int i = 0; i < NLEVGRND; ++i) {
      double om_frac = 0.0, sand = sand_b, clay = clay_b;
      if (i < NSOI) {
        const double q = LV(organic, i) / organic_max;
        om_frac = elmk_sq(q);  // pow(q, 2.0), which the host compiler folds to q * q
        sand = LV(pct_sand, i);
        clay = LV(pct_clay, i);
      }
      double ws, bsw, sucsat, watdry, watopt, watfc, tkmg, tkdry, csol;
      soil_hydraulic_params(sand, clay, LV(zsoi, i + NLEVSNO), om_frac, ws, bsw, sucsat, watdry, watopt, watfc, tkmg, tkdry, csol);
      if (i >= NSOI) csol = 2.0e6;  // csol_bedrock
      LV(watsat, i) = ws;
      LV(bsw, i) = bsw;
      LV(sucsat, i) = sucsat;
      LV(watdry, i) = watdry;
      LV(watopt, i) = watopt;
      LV(watfc, i) = watfc;
      LV(tkmg, i) = tkmg;
      LV(tkdry, i) = tkdry;
      LV(csol, i) = csol;  // (by soil index, as the reference writes it)
    }
#pragma unroll
    for (int i = 0; i < NLEVGRND; ++i) watsat[i] = LV(watsat, i);
  }

  // ---- init_vegrootfr (init_soil_state_impl.hh:180-213)
  {
    const int vt = S->vtype[c];
    const double ra = S->roota_par[vt], rb = S->rootb_par[vt];
#pragma unroll
    for (int i = NSOI; i < NLEVGRND; ++i) LV(rootfr, i) = 0.0;
    if (vt != 0) {  // PFT::noveg()
      double ea = elmk_exp(-ra * LV(zisoi, NLEVSNO)), eb = elmk_exp(-rb * LV(zisoi, NLEVSNO));
#pragma unroll 1
      for (int i = 0; i < NSOI - 1; i++) {
        const double zi1 = LV(zisoi, i + 1 + NLEVSNO);
        const double ea1 = elmk_exp(-ra * zi1), eb1 = elmk_exp(-rb * zi1);
        LV(rootfr, i) = 0.5 * (ea + eb - ea1 - eb1);
        ea = ea1;
        eb = eb1;
      }
      LV(rootfr, NSOI - 1) = 0.5 * (ea + eb);
    } else {
#pragma unroll
      for (int i = 0; i < NSOI; i++) LV(rootfr, i) = 0.0;
    }
  }

  // ---- init_soil_temp (:11-54)
  double t_soil = 0.0;     // the uniform initial soil temperature of this land unit
  int n_t = 0;             // ... written to levels NLEVSNO .. NLEVSNO + n_t - 1
  if (!L.lakpoi) {
    if (L.ltype == istice || L.ltype == istice_mec) {
      t_soil = 250.0;
      n_t = NLEVGRND;
    } else if (L.ltype == istwet) {
      t_soil = 277.0;
      n_t = NLEVGRND;
    } else if (L.urbpoi) {
      if (L.ctype == icol_road_perv || L.ctype == icol_road_imperv) {
        t_soil = 274.0;
        n_t = NLEVGRND;
      } else if (L.ctype == icol_sunwall || L.ctype == icol_shadewall || L.ctype == icol_roof) {
        t_soil = 292.0;
        n_t = NURB;
      }
    } else {
      t_soil = 274.0;
      n_t = NLEVGRND;
    }
  }
  double t_lev[NLEVTOT];
#pragma unroll
  for (int i = 0; i < NLEVTOT; i++) {
    double t = LV(t_soisno, i);
    if (i < NLEVSNO) {
      if (snl > 0 && i >= NLEVSNO - snl) t = 250.0;
    } else if (i - NLEVSNO < n_t) {
      t = t_soil;
    }
    t_lev[i] = t;
    LV(t_soisno, i) = t;
  }
  if (!L.lakpoi) {
    double tg = t_lev[NLEVSNO];
#pragma unroll
    for (int i = 0; i < NLEVSNO; i++)
      if (i == NLEVSNO - snl) tg = t_lev[i];
    S->t_grnd[c] = tg;
  }

  // ---- init_snow_state (init_snow_state_impl.hh:11-63).  snow_depth and h2osno are zero from here on, so the snow-cover
  //      fraction is zero on every branch (min(0 / 0.05, 1) for urban points; the tanh branch is never reached).
  S->h2osno[c] = 0.0;
  S->int_snow[c] = 0.0;
  S->snow_depth[c] = 0.0;
  S->h2osfc[c] = 0.0;
  S->h2ocan[c] = 0.0;
  S->frac_h2osfc[c] = 0.0;
  S->fwet[c] = 0.0;
  S->fdry[c] = 0.0;
  S->frac_sno[c] = 0.0;
#pragma unroll
  for (int i = 0; i < NLEVSNO; i++) LV(snw_rds, i) = (snl > 0 && i >= NLEVSNO - snl) ? SNW_RDS_MIN : 0.0;

  // ---- init_soilh2o_state (init_soil_state_impl.hh:65-176)
  {
    double vol[NLEVGRND], liq[NLEVTOT], ice[NLEVTOT], dzl[NLEVTOT];
#pragma unroll
    for (int i = 0; i < NLEVGRND; ++i) vol[i] = SPVAL;
#pragma unroll
    for (int i = 0; i < NLEVTOT; ++i) {
      liq[i] = SPVAL;
      ice[i] = SPVAL;
      dzl[i] = LV(dz, i);
    }
    int nlevs = NLEVGRND;
    if (!L.lakpoi) {
      if (L.ltype == istsoil || L.ltype == istcrop) {
#pragma unroll
        for (int i = 0; i < NLEVGRND; ++i) vol[i] = (i >= NBED) ? 0.0 : 0.15;
      } else if (L.urbpoi) {
        if (L.ctype == icol_road_perv) {
#pragma unroll
          for (int i = 0; i < NLEVGRND; ++i) vol[i] = (i < NBED) ? 0.3 : 0.0;
        } else if (L.ctype == icol_road_imperv) {
#pragma unroll
          for (int i = 0; i < NLEVGRND; ++i) vol[i] = 0.0;
        } else {
          nlevs = NURB;
#pragma unroll
          for (int i = 0; i < NURB; ++i) vol[i] = 0.0;
        }
      } else if (L.ltype == istwet) {
#pragma unroll
        for (int i = 0; i < NLEVGRND; ++i) vol[i] = (i >= NBED) ? 0.0 : 1.0;
      } else if (L.ltype == istice || L.ltype == istice_mec) {
#pragma unroll
        for (int i = 0; i < NLEVGRND; ++i) vol[i] = 1.0;
      }
#pragma unroll
      for (int i = 0; i < NLEVGRND; ++i) {
        if (i < nlevs) {
          const int o = i + NLEVSNO;
          vol[i] = dmin(vol[i], watsat[i]);
          if (t_lev[o] <= TFRZ) {
            ice[o] = dzl[o] * DENICE * vol[i];
            liq[o] = 0.0;
          } else {
            ice[o] = 0.0;
            liq[o] = dzl[o] * DENH2O * vol[i];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NLEVSNO; ++i) {
        if (i >= NLEVSNO - snl) {
          ice[i] = dzl[i] * 250.0;
          liq[i] = 0.0;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NLEVSNO; ++i) {
        if (i >= NLEVSNO - snl) {
          ice[i] = dzl[i] * BDSNO;
          liq[i] = 0.0;
        }
      }
#pragma unroll
      for (int i = 0; i < NLEVGRND; ++i) {
        const int o = i + NLEVSNO;
        if (i < NSOI) {
          vol[i] = watsat[i];
          liq[o] = SPVAL;
          ice[o] = SPVAL;
        } else {
          vol[i] = 0.0;
        }
      }
    }
    // "for frozen layers" (:164-173): every layer once more
#pragma unroll
    for (int i = 0; i < NLEVGRND; ++i) {
      const int o = i + NLEVSNO;
      if (t_lev[o] <= TFRZ) {
        ice[o] = dzl[o] * DENICE * vol[i];
        liq[o] = 0.0;
      } else {
        ice[o] = 0.0;
        liq[o] = dzl[o] * DENH2O * vol[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NLEVGRND; ++i) LV(h2osoi_vol, i) = vol[i];
#pragma unroll
    for (int i = 0; i < NLEVTOT; ++i) {
      LV(h2osoi_liq, i) = liq[i];
      LV(h2osoi_ice, i) = ice[i];
    }
  }
}

void launch_initialize_state(const DevState* S, int64_t n, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_init_state, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, S);
}

}  // namespace elmk
