// elm_interface_demo.cc - the reference's driver loop (driver/kokkos/kokkos_driver.cc: construct ELMInterface, setup(),
// advance() per step, getPrimaryVars()) against libelmk through include/elmk_interface.hpp.
// The state and parameter arrays come from a flat binary file written by tests (tests/test_gpu_parity.py::
// test_cpp_interface_mirror): the demo has no file readers of its own, as the library has none.
//
//   g++ -std=c++17 -Iinclude examples/elm_interface_demo.cc -Lelmkernels_amd -lelmk -Wl,-rpath,$PWD/elmkernels_amd -o demo
//   ./demo state.bin nsteps out.bin
//
// state.bin: int64 ncols; then records until EOF: char name[32]; int32 kind (0 field [column][level], 1 parameter block);
//            int64 nbytes; payload.
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "elmk_interface.hpp"

static std::vector<char> read_all(const char* path)
{
  FILE* f = std::fopen(path, "rb");
  if (!f) throw std::runtime_error(std::string("cannot open ") + path);
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<char> b((size_t)n);
  if (std::fread(b.data(), 1, (size_t)n, f) != (size_t)n) throw std::runtime_error("short read");
  std::fclose(f);
  return b;
}

int main(int argc, char** argv)
{
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s state.bin nsteps out.bin\n", argv[0]);
    return 2;
  }
  try {
    const std::vector<char> blob = read_all(argv[1]);
    const int nsteps = std::atoi(argv[2]);
    const char* p = blob.data();
    const char* end = p + blob.size();
    int64_t ncols;
    std::memcpy(&ncols, p, 8);
    p += 8;
    std::map<std::string, const char*> fields, params;
    while (p < end) {
      char name[33] = {0};
      std::memcpy(name, p, 32);
      int32_t kind;
      int64_t nbytes;
      std::memcpy(&kind, p + 32, 4);
      std::memcpy(&nbytes, p + 36, 8);
      (kind == 0 ? fields : params)[name] = p + 44;
      p += 44 + nbytes;
    }
    auto D = [&](const char* k) { return reinterpret_cast<const double*>(params.at(k)); };
    auto I = [&](const char* k) { return reinterpret_cast<const int32_t*>(params.at(k)); };

    elmk::ELMInterface elm(ncols, 0);
    elmk_snicar_tables t;
    std::memset(&t, 0, sizeof t);
    {
      const double** slot = reinterpret_cast<const double**>(&t);  // the struct is 31 const double* members, in this order
      for (int i = 0; i < (int)(sizeof t / sizeof(double*)); i++) slot[i] = D(("snicar/" + std::to_string(i)).c_str());
    }
    const int32_t* land = I("land");
    const double* sc = D("scalars");
    elm.setup(land[0], land[1], land[2], land[3], land[4], sc[0], (int)sc[1], sc[2], sc[3], D("pft_psn"), D("pft_alb"), D("z0mr"),
              D("displar"), D("albsat"), D("albdry"), &t, D("age_tau"), D("age_kappa"), D("age_drdt0"));
    for (const auto& kv : fields) elm.upload(kv.first.c_str(), kv.second);

    elmk::StepWeights w;
    std::memcpy(w.forc_wt1, D("forc_wt1"), 64);
    std::memcpy(w.forc_wt2, D("forc_wt2"), 64);
    w.month_wt1 = D("month_wt")[0];
    w.month_wt2 = D("month_wt")[1];
    w.qbot_is_relative_humidity = 0;
    const double dt = sc[4];
    for (int s = 0; s < nsteps; s++) {
      if (elm.advance(dt, w)) return 1;
    }
    auto pv = elm.getPrimaryVars();
    FILE* o = std::fopen(argv[3], "wb");
    std::fwrite(pv->t_soisno.data(), 8, pv->t_soisno.size(), o);
    std::fwrite(pv->h2osoi_liq.data(), 8, pv->h2osoi_liq.size(), o);
    std::fwrite(pv->h2osoi_ice.data(), 8, pv->h2osoi_ice.size(), o);
    std::fwrite(pv->t_grnd.data(), 8, pv->t_grnd.size(), o);
    std::fwrite(pv->h2osno.data(), 8, pv->h2osno.size(), o);
    std::fwrite(pv->snl.data(), 4, pv->snl.size(), o);
    std::fwrite(&elm.conservation()[0][0], 8, 24, o);
    std::fclose(o);
    std::printf("advance x %d on %ld columns: ok, warning flags %#x\n", nsteps, (long)ncols, (unsigned)elm.warning_flags());
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
