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
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "elmk.h"

namespace elmk {

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
