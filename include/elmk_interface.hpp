/* elmk_interface.hpp - the C++ host side above the C ABI of elmk.h: a header-only mirror of the reference's driver class
 * ELM::ELMInterface (driver/kokkos/elm_kokkos_interface.hh:11-28, elm_kokkos_interface.cc:38-358) for a caller that links
 * libelmk instead of the reference's Kokkos wrappers.  Same member names, same call order in advance(), same PrimaryVars
 * members (src/data/elm_state.h:17-48).  What the reference's class also does - opening the NetCDF surface / forcing /
 * parameter files and the date arithmetic of kokkos_init_timestep - stays with the caller (control plane, out of scope:
 * DESIGN.md section 8): the caller uploads the bracketing forcing / phenology records and passes the interpolation weights.
 *
 *   elmk::ELMInterface elm(ncols, gpu);                    // was ELM::ELMInterface elm(ncols);
 *   elm.setup(land, pft_psn, pft_alb, ..., snicar, age);   // was elm.setup();  (the files' contents, read by the caller)
 *   elm.upload("t_soisno", host_ptr);  ...                 // was initialize_kokkos_elm(*S_, files ...)
 *   bool failed = elm.advance(dt_seconds, w);              // was elm.advance(dt_start_date, dt_seconds);
 *   auto pv = elm.getPrimaryVars();                        // same
 *
 * Nothing here touches HIP or torch: plain C++17 over the extern "C" entry points. */
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "elmk.h"

namespace elmk {

/* The host scalars kokkos_init_timestep computes before its kernels (init_timestep_kokkos.cc:26-34): the cosine of the solar
 * zenith angle averaged over the step, the day length and its yearly maximum.  Plain <cmath> on the host, as in the
 * reference (src/physics/incident_shortwave.cc:14-121, day_length.cc:15-39), so the values are the reference's bits on the
 * same libm; checked against the reference's own sources compiled into oracle/_ref (tests/test_host_side.py). */
namespace solar {
constexpr double PI = 3.14159265358979323846;  // ELMconst::ELM_PI
constexpr double TWO_PI = PI * 2.0;
constexpr double PI_OVER_TWO = PI / 2.0;

/* incident_shortwave.cc:17 (lnd_import szenith() / shr_orb_cosz()) */
inline double declination_angle_sin(int doy) { return 23.45 * PI / 180.0 * std::sin(TWO_PI * (284.0 + doy) / 365.0); }
/* :29-31 */
inline double ensure_tan_defined(double var) { return (var == PI_OVER_TWO) ? var - 1.0e-05 : (var == -PI_OVER_TWO) ? var + 1.0e-05 : var; }
/* :37-42: start of the step as an hour angle on [-pi, pi) */
inline double dt_start_rad(double jday, double lonrad)
{
  const double t_start = (jday - std::floor(jday)) * TWO_PI + lonrad - PI;
  return (t_start >= PI) ? t_start - TWO_PI : (t_start < -PI) ? t_start + TWO_PI : t_start;
}
/* :52-56: half-day length [0, pi] */
inline double coshalfday(double latrad, double declin)
{
  const double cos_h = -std::tan(ensure_tan_defined(latrad)) * std::tan(ensure_tan_defined(declin));
  return (cos_h <= -1.0) ? PI : (cos_h >= 1.0) ? 0.0 : std::acos(cos_h);
}
/* :61-94 */
inline void avg_hourangle(double t_start, double t_end, double dtrad, double cos_h, double ha[4])
{
  auto clamp = [](double v, double lo, double hi) { return std::min(std::max(v, lo), hi); };
  if (t_end >= PI && t_start <= PI && PI - cos_h <= dtrad) {
    ha[0] = clamp(t_start, -cos_h, cos_h);
    ha[1] = cos_h;
    ha[2] = TWO_PI - cos_h;
    ha[3] = clamp(t_end, TWO_PI - cos_h, TWO_PI + cos_h);
  } else if (t_end >= -PI && t_start <= -PI && PI - cos_h <= dtrad) {
    ha[0] = clamp(t_start, -TWO_PI - cos_h, -TWO_PI + cos_h);
    ha[1] = -TWO_PI + cos_h;
    ha[2] = -cos_h;
    ha[3] = clamp(t_end, -cos_h, cos_h);
  } else {
    ha[0] = clamp((t_start > PI) ? t_start - TWO_PI : (t_start < -PI) ? t_start + TWO_PI : t_start, -cos_h, cos_h);
    ha[1] = clamp((t_end > PI) ? t_end - TWO_PI : (t_end < -PI) ? t_end + TWO_PI : t_end, -cos_h, cos_h);
    ha[2] = 0.0;
    ha[3] = 0.0;
  }
}
/* :98-121: Zhou et al. (2015) average of cos(zenith) over [t, t + dt] */
inline double average_cosz(double latrad, double lonrad, double dt, double jday)
{
  const double dtrad = dt * TWO_PI / 86400.0;
  const double t_start = dt_start_rad(jday, lonrad);
  const double t_end = t_start + dtrad;
  const double declin = declination_angle_sin(static_cast<int>(jday));
  const double cos_h = coshalfday(latrad, declin);
  const double aa = std::sin(latrad) * std::sin(declin);
  const double bb = std::cos(latrad) * std::cos(declin);
  double ha[4];
  avg_hourangle(t_start, t_end, dtrad, cos_h, ha);
  return (ha[1] > ha[0] || ha[3] > ha[2])
             ? (aa * (ha[1] - ha[0]) + bb * (std::sin(ha[1]) - std::sin(ha[0]))) / dtrad +
                   (aa * (ha[3] - ha[2]) + bb * (std::sin(ha[3]) - std::sin(ha[2]))) / dtrad
             : 0.0;
}
/* day_length.cc:15-34 (seconds); lat and decl in radians */
inline double daylength(double lat, double decl)
{
  const double secs_per_radian = 13750.9871;
  const double lat_epsilon = 10.0 * 2.220446049250313e-16;
  const double offset_pole = PI / 2.0 - lat_epsilon;
  const double my_lat = std::min(offset_pole, std::max(1.0 * offset_pole, lat));  // (the reference's own expression, :28)
  double temp = -(std::sin(my_lat) * std::sin(decl)) / (std::cos(my_lat) * std::cos(decl));
  temp = std::min(1.0, std::max(-1.0, temp));
  return 2.0 * secs_per_radian * std::acos(temp);
}
/* :39 */
inline double max_daylength(double lat) { return (lat < 0.0) ? daylength(lat, -0.409571) : daylength(lat, 0.409571); }
}  // namespace solar

/* ELM::PrimaryVars<ViewI1, ViewD1, ViewD2> (src/data/elm_state.h:17-48) on the host: [column][level], level fastest,
 * as the reference's Views are (src/utils/array.hh:176-179) */
struct PrimaryVars {
  explicit PrimaryVars(int64_t ncols)
      : snl(ncols), nrad(ncols), snow_depth(ncols), frac_sno(ncols), int_snow(ncols), h2ocan(ncols), h2osno(ncols),
        h2osfc(ncols), t_grnd(ncols), t_h2osfc(ncols), t_h2osfc_bef(ncols), snw_rds(ncols * 5), h2osoi_liq(ncols * 20),
        h2osoi_ice(ncols * 20), h2osoi_vol(ncols * 15), t_soisno(ncols * 20), dz(ncols * 20), zsoi(ncols * 20),
        zisoi(ncols * 21)
  {
  }
  std::vector<int32_t> snl, nrad;
  std::vector<double> snow_depth, frac_sno, int_snow, h2ocan, h2osno, h2osfc, t_grnd, t_h2osfc, t_h2osfc_bef;
  std::vector<double> snw_rds, h2osoi_liq, h2osoi_ice, h2osoi_vol, t_soisno, dz, zsoi, zisoi;
};

/* what kokkos_init_timestep's host part yields per step (init_timestep_kokkos.cc:17-52): the interpolation weights of
 * the two bracketing forcing records and of the two bracketing months; the records themselves are uploaded by the caller
 * (fields atm_* and mlai .. mhbot, ELMK_LAYOUT_SOA) whenever the bracket moves */
struct StepWeights {
  double forc_wt1[8], forc_wt2[8];  // per stream: TBOT, PBOT, QBOT|RH, FLDS, FSDS, PREC, WIND, ZBOT (atm_data_impl.hh:191-199)
  double month_wt1, month_wt2;
  int qbot_is_relative_humidity;
};

class ELMInterface {
 public:
  ELMInterface(int64_t ncols, int gpu = 0) : ncols_(ncols)
  {
    if (elmk_create(ncols, gpu, &ctx_) != ELMK_OK) throw std::runtime_error(elmk_last_error(nullptr));
  }
  ~ELMInterface() { (void)elmk_destroy(ctx_); }
  ELMInterface(const ELMInterface&) = delete;
  ELMInterface& operator=(const ELMInterface&) = delete;

  /* ELMInterface::setup (elm_kokkos_interface.cc:58-267) minus the file reads: parameters that the reference's state
   * object carries beside its Views */
  void setup(int ltype, int ctype, int vtype, int urbpoi, int lakpoi, double dewmx, int oldfflag, double dayl, double max_dayl,
             const double* pft_psn, const double* pft_alb, const double* z0mr, const double* displar, const double* albsat,
             const double* albdry, const elmk_snicar_tables* snicar, const double* age_tau, const double* age_kappa,
             const double* age_drdt0)
  {
    ok(elmk_set_land(ctx_, ltype, ctype, vtype, urbpoi, lakpoi));
    ok(elmk_set_scalars(ctx_, dewmx, oldfflag, dayl, max_dayl));
    ok(elmk_set_pft(ctx_, pft_psn, pft_alb, z0mr, displar));
    ok(elmk_set_soilcolor(ctx_, albsat, albdry));
    ok(elmk_set_snicar(ctx_, snicar));
    ok(elmk_set_snow_age_tables(ctx_, age_tau, age_kappa, age_drdt0));
    ok(elmk_set_graph(ctx_, 1));  // advance() replays one HIP graph per step
  }

  /* the per-column part of ELM::initialize_kokkos_elm (initialize_elm_kokkos.cc:373-428): cold-start state from the
   * uploaded topography, snow depth, soil texture (fields pct_sand, pct_clay, organic) and PFT; call after the uploads */
  void initialize(double organic_max, const double* roota_par, const double* rootb_par)
  {
    ok(elmk_set_init_params(ctx_, organic_max, roota_par, rootb_par));
    ok(elmk_initialize_state(ctx_));
  }

  /* one ELMStateViews member, host layout of the reference ([column][level]) */
  void upload(const char* field, const void* host) { ok(elmk_upload(ctx_, id(field), host, 0, ncols_, ELMK_LAYOUT_COL_MAJOR)); }
  void download(const char* field, void* host) { ok(elmk_download(ctx_, id(field), host, 0, ncols_, ELMK_LAYOUT_COL_MAJOR)); }

  /* ELMInterface::advance (elm_kokkos_interface.cc:269-322): kokkos_init_timestep's per-column work, then the ten physics
   * calls in the reference's order, then kokkos_evaluate_conservation.  Returns false like the reference ("failed"
   * flag); a raised throw / assert site of the reference's physics becomes one exception per step. */
  bool advance(double dt_seconds, const StepWeights& w)
  {
    ok(elmk_phenology(ctx_, w.month_wt1, w.month_wt2));
    ok(elmk_get_forcing(ctx_, w.forc_wt1, w.forc_wt2, w.qbot_is_relative_humidity));
    ok(elmk_init_timestep(ctx_));
    ok(elmk_advance_physics(ctx_, dt_seconds));
    ok(elmk_evaluate_conservation(ctx_, dt_seconds, &conservation_[0][0], nullptr));
    uint32_t flags = 0;
    int64_t col = -1;
    ok(elmk_error_summary(ctx_, &flags, &col));
    if (flags & ELMK_ERR_FATAL_MASK)
      throw std::runtime_error("ELM physics error flags " + std::to_string(flags) + ", first at column " + std::to_string(col));
    last_flags_ = flags;
    return false;
  }

  /* the host scalars of kokkos_init_timestep (init_timestep_kokkos.cc:26-34) for one location: coszen for every column
   * (Utils::assign(S.coszen, cosz)), S.dayl and S.max_dayl.  decday = Utils::decimal_doy(date) + 1.0, doy = date.doy. */
  void set_solar_geometry(double lat_r, double lon_r, double dt_seconds, double decday, int doy, double dewmx, int oldfflag)
  {
    ok(elmk_fill(ctx_, id("coszen"), solar::average_cosz(lat_r, lon_r, dt_seconds, decday)));
    ok(elmk_set_scalars(ctx_, dewmx, oldfflag, solar::daylength(lat_r, solar::declination_angle_sin(doy + 1)),
                        solar::max_daylength(lat_r)));
  }

  /* ELMInterface::copyPrimaryVars / getPrimaryVars (elm_kokkos_interface.cc:324-356) */
  void copyPrimaryVars(PrimaryVars& pv)
  {
    download("snl", pv.snl.data());
    download("snow_depth", pv.snow_depth.data());
    download("frac_sno", pv.frac_sno.data());
    download("int_snow", pv.int_snow.data());
    download("snw_rds", pv.snw_rds.data());
    download("h2osoi_liq", pv.h2osoi_liq.data());
    download("h2osoi_ice", pv.h2osoi_ice.data());
    download("h2osoi_vol", pv.h2osoi_vol.data());
    download("h2ocan", pv.h2ocan.data());
    download("h2osno", pv.h2osno.data());
    download("h2osfc", pv.h2osfc.data());
    download("t_soisno", pv.t_soisno.data());
    download("t_grnd", pv.t_grnd.data());
    download("t_h2osfc", pv.t_h2osfc.data());
    download("t_h2osfc_bef", pv.t_h2osfc_bef.data());
    download("nrad", pv.nrad.data());
    download("dz", pv.dz.data());
    download("zsoi", pv.zsoi.data());
    download("zisoi", pv.zisoi.data());
  }
  std::shared_ptr<PrimaryVars> getPrimaryVars()
  {
    auto pv = std::make_shared<PrimaryVars>(ncols_);
    copyPrimaryVars(*pv);
    return pv;
  }

  /* (min, max, sum) of the eight conservation diagnostics of the last advance() (the reference prints column 0's) */
  const double (&conservation() const)[8][3] { return conservation_; }
  uint32_t warning_flags() const { return last_flags_; }  // ELMK_WARN_* bits raised in the last step
  elmk_ctx* context() { return ctx_; }
  int64_t ncols() const { return ncols_; }

 private:
  void ok(int rc)
  {
    if (rc != ELMK_OK) throw std::runtime_error(elmk_last_error(ctx_));
  }
  int id(const char* field)
  {
    const int f = elmk_field_id(field);
    if (f < 0) throw std::runtime_error(std::string("unknown ELM state field ") + field);
    return f;
  }
  elmk_ctx* ctx_{nullptr};
  int64_t ncols_{0};
  double conservation_[8][3]{};
  uint32_t last_flags_{0};
};

}  // namespace elmk
