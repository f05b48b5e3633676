// elmk_dev.h - device-side view of the column state and the small scalar helpers every kernel uses.
//
// Layout in HBM (DESIGN.md "Data layout"): every field is SoA [lev][column], column fastest; the level
// stride `ld` is ncols rounded up to 64 so each level row starts on a 512-byte boundary.  A wave64 that
// reads one level of one field therefore issues one fully coalesced 512-byte (fp64) request.
// One thread owns one column; kernels read each touched element once and write each result once.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "elmk.h"
#include "elmk_math.h"  // exp / log / pow / log10 / atan with the bits of the host libm the reference runs on

namespace elmk {

// ---- physical constants: values of src/data/elm_constants.h:18-53, formed by the same expressions ----
constexpr double TFRZ = 273.15;
constexpr double ELM_PI = 3.14159265358979323846;
constexpr double BOLTZ = 1.38065e-23;
constexpr double AVOGAD = 6.02214e26;
constexpr double MWWV = 18.016;
constexpr double RGAS = AVOGAD * BOLTZ;
constexpr double RWV = RGAS / MWWV;
constexpr double STEBOL = 5.67e-8;
constexpr double MWDAIR = 28.966;
constexpr double RAIR = RGAS / MWDAIR;
constexpr double GRAV = 9.80616;
constexpr double ROVERG = RWV / GRAV * 1000.;
constexpr double O2_MOLAR_CONST = 0.209;
constexpr double CO2_PPMV = 355.0;
constexpr double DENICE = 0.917e3;
constexpr double DENH2O = 1.000e3;
constexpr double HVAP = 2.501e6;
constexpr double HFUS = 3.337e5;
constexpr double HSUB = HVAP + HFUS;
constexpr double VKC = 0.4;
constexpr double CPAIR = 1.00464e3;
constexpr double CSOILC = 0.004;
constexpr double ZLND = 0.01;
constexpr double ZSNO = 0.0024;
constexpr double SNW_RDS_MIN = 54.526;
constexpr double SPVAL = 1.0e36;

// land-unit / column codes: src/data/land_data.h:8-31
enum : int {
  istsoil = 1,
  istcrop = 2,
  istice = 3,
  istice_mec = 4,
  istdlak = 5,
  istwet = 6,
  icol_roof = 71,
  icol_sunwall = 72,
  icol_shadewall = 73,
  icol_road_imperv = 74,
  icol_road_perv = 75,
  pft_nsoybean = 23,
  pft_nsoybeanirrig = 24
};

constexpr int NLEVSNO = ELMK_NLEVSNO;
constexpr int NLEVGRND = ELMK_NLEVGRND;
constexpr int NLEVTOT = ELMK_NLEVTOT;
constexpr int ELMK_SNOWAGE_N = 11 * 31 * 8;  // SnwRdsTable extents (snicar_data_impl.hh:42-47, snow_snicar.h:34-39)

// Device pointers of the parameter block are GLOBAL-address-space pointers.  A plain C++ pointer loaded from a struct in
// memory is a generic ("flat") pointer to the compiler: every access through it is a flat_load / flat_store, which may
// target LDS, so it cannot be reordered around LDS accesses, counts against both memory counters and returns out of order
// (each dependent use waits for ALL outstanding memory operations).  With the address space in the type the same source
// compiles to global_load / global_store with counted waits.
#define ELMK_GLOBAL __attribute__((address_space(1)))
template <typename T> using gptr = ELMK_GLOBAL T*;
// the atomic builtins of HIP take generic pointers: cast at the call (the operation itself is the same memory atomic)
template <typename T> __host__ __device__ __forceinline__ T* elmk_generic(ELMK_GLOBAL T* p) { return (T*)p; }
#define ELMK_GENERIC(p) elmk_generic(p)

template <int T> struct ctype_of;
template <> struct ctype_of<ELMK_F64> { using type = double; };
template <> struct ctype_of<ELMK_I32> { using type = int32_t; };
template <> struct ctype_of<ELMK_U8> { using type = uint8_t; };

// How a state field of elmk_fields.def is held on the device.  Product build: a global pointer to its element type.
// ELMK_STATE_F32 (the report-only build of BASELINE config 5, libelmk_f32.so): every fp64 field is STORED as fp32 - half the
// bytes per column - while all arithmetic stays fp64: an access widens on load and rounds to nearest on store, through a
// reference object, so the kernels' source is the same text in both builds.  Scratch (queue records, work arrays) and the
// shared parameter tables stay fp64 in both.
template <int T> struct field_of {
  using type = gptr<typename ctype_of<T>::type>;
  static __host__ __device__ type from(void* p) { return (type)p; }
};
#ifdef ELMK_STATE_F32
typedef float state_real;
struct F32Ref {
  gptr<float> p;
  __device__ __forceinline__ operator double() const { return (double)*p; }
  __device__ __forceinline__ const F32Ref& operator=(double v) const { *p = (float)v; return *this; }
  __device__ __forceinline__ const F32Ref& operator=(const F32Ref& o) const { *p = *o.p; return *this; }
  __device__ __forceinline__ const F32Ref& operator+=(double v) const { *p = (float)((double)*p + v); return *this; }
  __device__ __forceinline__ const F32Ref& operator-=(double v) const { *p = (float)((double)*p - v); return *this; }
  __device__ __forceinline__ const F32Ref& operator*=(double v) const { *p = (float)((double)*p * v); return *this; }
};
struct F32Field {
  gptr<float> p;
  __device__ __forceinline__ F32Ref operator[](int64_t i) const { return F32Ref{p + i}; }
  __device__ __forceinline__ F32Field operator+(int64_t off) const { return F32Field{p + off}; }
};
template <> struct field_of<ELMK_F64> {
  using type = F32Field;
  static __host__ __device__ type from(void* p) { return F32Field{(gptr<float>)p}; }
};
#elif defined(ELMK_STATE_NT)
// Accesses to an fp64 state field carry the nontemporal hint (global_load / global_store ... nt): a streamed column state is
// read once and written once per kernel, nothing of it is worth a cache line.  ELMK_STATE_NT is a bit mask chosen per
// translation unit (Makefile, FLAGS_<file>): 1 = loads, 2 = stores, 3 = both.  The type has the layout of the plain pointer,
// so DevState is the same block of memory for every translation unit.
typedef double state_real;
struct NTRef {
  gptr<double> p;
  static __device__ __forceinline__ double ld(gptr<double> q) { return ((ELMK_STATE_NT) & 1) ? __builtin_nontemporal_load(q) : *q; }
  static __device__ __forceinline__ void st(double v, gptr<double> q)
  {
    if ((ELMK_STATE_NT) & 2) __builtin_nontemporal_store(v, q);
    else *q = v;
  }
  __device__ __forceinline__ operator double() const { return ld(p); }
  __device__ __forceinline__ const NTRef& operator=(double v) const { st(v, p); return *this; }
  __device__ __forceinline__ const NTRef& operator=(const NTRef& o) const { st(ld(o.p), p); return *this; }
  __device__ __forceinline__ const NTRef& operator+=(double v) const { st(ld(p) + v, p); return *this; }
  __device__ __forceinline__ const NTRef& operator-=(double v) const { st(ld(p) - v, p); return *this; }
  __device__ __forceinline__ const NTRef& operator*=(double v) const { st(ld(p) * v, p); return *this; }
};
struct NTField {
  gptr<double> p;
  __device__ __forceinline__ NTRef operator[](int64_t i) const { return NTRef{p + i}; }
  __device__ __forceinline__ NTField operator+(int64_t off) const { return NTField{p + off}; }
};
template <> struct field_of<ELMK_F64> {
  using type = NTField;
  static __host__ __device__ type from(void* p) { return NTField{(gptr<double>)p}; }
};
static_assert(sizeof(NTField) == sizeof(gptr<double>) && alignof(NTField) == alignof(gptr<double>),
              "NTField must have the layout of the plain field pointer: DevState is shared by units built with and without the hint");
#else
typedef double state_real;
#endif
typedef field_of<ELMK_F64>::type dfield;

// Scratch streams that one kernel writes once and another reads once: which of their accesses carry the nontemporal hint
// (ELMK_SCRATCH_NT bit mask; interleaved A/B in profiles/r04_scratch_nt_ab.txt).  1: loads of the SNICAR products (alb_snow),
// 2: their stores, 4: the canopy queue records.  The records lose badly (their 8-byte stores into blocks of 8 positions
// depend on merging in L2: k_cf_init +65 %, k_cf_finish +70 %); reading the SNICAR products once, by column, gains.
#ifndef ELMK_SCRATCH_NT
#define ELMK_SCRATCH_NT 1
#endif
template <int BIT> __device__ __forceinline__ double sc_ld(gptr<const double> p)
{
  return ((ELMK_SCRATCH_NT) & BIT) ? __builtin_nontemporal_load(p) : *p;
}
template <int BIT> __device__ __forceinline__ void sc_st(gptr<double> p, double v)
{
  if ((ELMK_SCRATCH_NT) & BIT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// indices into one row of the PFT photosynthesis table (member order of ELM::PFTDataPSN, pft_data.h:20-24)
enum : int {
  P_fnr, P_act25, P_kcha, P_koha, P_cpha, P_vcmaxha, P_jmaxha, P_tpuha, P_lmrha, P_vcmaxhd, P_jmaxhd, P_tpuhd,
  P_lmrhd, P_lmrse, P_qe, P_theta_cj, P_bbbopt, P_mbbopt, P_c3psn, P_slatop, P_leafcn, P_flnr, P_fnitr, P_dleaf,
  P_smpso, P_smpsc, P_tc_stress
};

struct Land {
  int ltype, ctype, vtype, urbpoi, lakpoi;
};

// SNICAR tables on the device (one contiguous buffer; offsets in doubles)
struct SnicarDev {
  const double* base;
};
enum : int {
  SN_OC1 = 0,                      // ss_alb, asm_prm, ext_cff_mss: 3 x [5] each for oc1, oc2, dst1..4
  SN_AER_STRIDE = 15,              // per species block
  SN_SNW_DRC = 6 * 15,             // 3 x [5][1471]
  SN_SNW_DFS = SN_SNW_DRC + 3 * 5 * ELMK_MIE_N,
  SN_BC1 = SN_SNW_DFS + 3 * 5 * ELMK_MIE_N,  // 3 x [10][5]
  SN_BC2 = SN_BC1 + 150,
  SN_BCENH = SN_BC2 + 150,         // [8][10][5]
  SN_TOTAL = SN_BCENH + 400
};

// work arrays and work lists of the compacted (queue-driven) kernels
enum : int {
  WK_CF_LWGRND = 0,  // canopy_fluxes: ground-emitted longwave term (three pow(T,4)), by column
  WK_DEBUG,          // development probes only (per-wave timeline of k_cf_iterate, CF_PROBE builds)
  WK_N
};
enum : int {
  LIST_CF_QUEUE = 0,  // canopy_fluxes work queue: only its length and head counters are used (k_canopy_fluxes.hip)
  LIST_BG,            // bare-ground columns
  LIST_ALB_0,         // (unused slot: snow-free columns need no queue)
  LIST_ALB_1,         // sunlit snow-covered columns by number of (possibly fictitious) snow layers 1..5
  LIST_ALB_2,
  LIST_ALB_3,
  LIST_ALB_4,
  LIST_ALB_5,
  NLISTS
};

// canopy_fluxes queue records (k_canopy_fluxes.hip): per queue position, SoA [k][position] with stride ld
constexpr int CF_NCLS = 12;    // scheduling classes: 6 bins of the previous call's trip count x (day, night)
constexpr int CF_REC_N = 43;   // doubles a column carries into the iteration kernel
constexpr int CF_IREC_N = 3;   // int32: vtype, nrad, frac_veg_nosno
constexpr int CF_FIN_N = 24;   // doubles the iteration kernel hands to the finishing kernel

// Queue counters live one per 128-byte line (same-line atomics serialise in one L2 channel):
// length of list k at counters[k * CPAD], queue head of list k at counters[(NLISTS + k) * CPAD].
constexpr int CPAD = 32;
#define ELMK_LIST_COUNT(S, k) ((S)->counters[(k)*CPAD])
#define ELMK_LIST_HEAD(S, k) ((S)->counters[(NLISTS + (k)) * CPAD])

// Workgroup-aggregated classification append: every thread of the workgroup (any size) calls this with its class
// cls in [0, NCLS) or -1; column c goes to list first_list + cls.  One global atomic per class per workgroup;
// order inside a list follows (workgroup arrival, wave, lane).  All threads of the workgroup must call it.
template <int NCLS>
__device__ __forceinline__ void block_classify_append(gptr<int32_t> lists, int64_t ld, gptr<uint32_t> counters,
                                                      int first_list, int cls, int32_t c)
{
  __shared__ uint32_t s_cnt[NCLS];
  __shared__ uint32_t s_base[NCLS];
  if (threadIdx.x < NCLS) s_cnt[threadIdx.x] = 0u;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t my_off = 0u;
#pragma unroll
  for (int k = 0; k < NCLS; k++) {
    const unsigned long long m = __ballot(cls == k);
    if (m != 0ull) {
      const int leader = __ffsll((long long)m) - 1;
      uint32_t off = 0u;
      if (lane == leader) off = atomicAdd(&s_cnt[k], (uint32_t)__popcll(m));
      off = __shfl(off, leader, 64);
      if (cls == k) my_off = off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
  }
  __syncthreads();
  if (threadIdx.x < NCLS) {
    const uint32_t n = s_cnt[threadIdx.x];
    s_base[threadIdx.x] = n ? atomicAdd(ELMK_GENERIC(&counters[(first_list + threadIdx.x) * CPAD]), n) : 0u;
  }
  __syncthreads();
  // (a list holds at most every column once; the bound only matters if a caller's earlier step was cut short by a HIP error
  //  between filling and draining a list, when its counter is not back at zero)
  if (cls >= 0 && (int64_t)s_base[cls] + my_off < ld) lists[(int64_t)(first_list + cls) * ld + s_base[cls] + my_off] = c;
}

// Everything a kernel needs, resident in device memory (kernels get one pointer; all loads from this
// struct are wave-uniform and become scalar loads).
struct DevState {
  int64_t ncols;
  int64_t ld;  // level stride (elements)
  Land land;
  double dewmx;
  int oldfflag;
  double dayl, max_dayl;
  double pft_psn[ELMK_MXPFT][ELMK_PSN_NPARAM];
  double pft_alb[ELMK_MXPFT][ELMK_ALB_NPARAM];
  double z0mr[ELMK_MXPFT], displar[ELMK_MXPFT];
  double albsat[ELMK_NSOILCOL][2], albdry[ELMK_NSOILCOL][2];
  gptr<const double> snicar;  // SN_TOTAL doubles
  gptr<const double> snowage;  // 3 x ELMK_SNOWAGE_N: SnwRdsTable snowage_tau, snowage_kappa, snowage_drdt0 [11][31][8]
  // per-call scratch owned by the context (never part of the state contract):
  gptr<double> wk;          // WK_N work arrays, SoA [k][column] with the same level stride ld
  gptr<int32_t> lists;      // NLISTS column-index lists, each ld entries (work queues of the compacted kernels)
  gptr<uint32_t> counters;  // list lengths and queue heads (ELMK_LIST_COUNT / ELMK_LIST_HEAD)
  gptr<double> cons_diag;   // 8 x ld: conservation diagnostics per column (k_surface_fluxes.hip)
  gptr<double> alb_snow;    // 28 x ld: SNICAR products of the sunlit snow-covered columns (k_albedo_snicar.hip), by column
  gptr<int32_t> cf_niter;   // canopy_fluxes trip count of each column in the previous call (scheduling hint only)
  gptr<double> cf_rec;      // CF_REC_N x ld: inputs of the queued columns, by queue position
  gptr<double> cf_fin;      // CF_FIN_N x ld: converged iteration state, by queue position
  gptr<int32_t> cf_irec;    // CF_IREC_N x ld
  gptr<int32_t> cf_pos;     // queue position of each column (-1: not vegetated)
  gptr<double> cf_given;    // 3 x ld: forc_rho, forc_po2, forc_pco2 handed in by the L2-level entries (elmk_*_given), by column
  gptr<int8_t> cf_cls;      // scheduling class of each column as k_fz_prep counted it (fused step)
  gptr<uint32_t> cf_blk;    // CF_NCLS x cf_nblk: per-workgroup class counts, then exclusive offsets
  int64_t cf_nblk;     // workgroups of 256 columns
#define ELMK_FIELD(name, T, nlev) field_of<ELMK_##T>::type name;
#include "elmk_fields.def"
#undef ELMK_FIELD
  gptr<uint32_t> err_flags;
  // cold-start initialisation (k_init_state.hip): organic_max of the parameter file, PFTData::roota_par / rootb_par
  double organic_max;
  double roota_par[ELMK_MXPFT], rootb_par[ELMK_MXPFT];
};

// std::min / std::max of the reference (<algorithm>): first argument wins ties and NaNs
__device__ __forceinline__ double dmin(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ double dmax(double a, double b) { return (a < b) ? b : a; }

// ---- src/physics/qsat_impl.hh:7-78 ------------------------------------------------------------------
__device__ __forceinline__ void qsat(double T, double p, double& es, double& esdT, double& qs, double& qsdT)
{
  double td = T - TFRZ;
  if (td > 100.0) td = 100.0;
  if (td < -75.0) td = -75.0;
  if (td >= 0.0) {
    es = 6.11213476 +
         td * (0.444007856 +
               td * (0.143064234e-01 +
                     td * (0.264461437e-03 +
                           td * (0.305903558e-05 +
                                 td * (0.196237241e-07 + td * (0.892344772e-10 + td * (-0.373208410e-12 + td * 0.209339997e-15)))))));
    esdT = 0.444017302 +
           td * (0.286064092e-01 +
                 td * (0.794683137e-03 +
                       td * (0.121211669e-04 +
                             td * (0.103354611e-06 +
                                   td * (0.404125005e-09 + td * (-0.788037859e-12 + td * (-0.114596802e-13 + td * 0.381294516e-16)))))));
  } else {
    es = 6.11123516 +
         td * (0.503109514 +
               td * (0.188369801e-01 +
                     td * (0.420547422e-03 +
                           td * (0.614396778e-05 +
                                 td * (0.602780717e-07 + td * (0.387940929e-09 + td * (0.149436277e-11 + td * 0.262655803e-14)))))));
    esdT = 0.503277922 +
           td * (0.377289173e-01 +
                 td * (0.126801703e-02 +
                       td * (0.249468427e-04 +
                             td * (0.313703411e-06 +
                                   td * (0.257180651e-08 + td * (0.133268878e-10 + td * (0.394116744e-13 + td * 0.498070196e-16)))))));
  }
  es = es * 100.0;
  esdT = esdT * 100.0;
  const double vp = 1.0 / (p - 0.378 * es);
  const double vp1 = 0.622 * vp;
  const double vp2 = vp1 * vp;
  qs = es * vp1;
  qsdT = esdT * vp2 * p;
}

// ---- src/physics/atm_physics_impl.hh:246-272 --------------------------------------------------------
__device__ __forceinline__ double derive_forc_vp(double qbot, double pbot) { return qbot * pbot / (0.622 + 0.378 * qbot); }
__device__ __forceinline__ double derive_forc_rho(double pbot, double qbot, double tbot)
{
  return (pbot - 0.378 * derive_forc_vp(qbot, pbot)) / (RAIR * tbot);
}
__device__ __forceinline__ double derive_forc_po2(double pbot) { return O2_MOLAR_CONST * pbot; }
__device__ __forceinline__ double derive_forc_pco2(double pbot) { return CO2_PPMV * 1.0e-6 * pbot; }

// ---- src/physics/friction_velocity_impl.hh ----------------------------------------------------------
// :17-24, :27-33
__device__ __forceinline__ double stab1(double zeta)
{
  const double chik2 = sqrt(1.0 - 16.0 * zeta);
  const double chik = sqrt(chik2);
  return 2.0 * elmk_log((1.0 + chik) * 0.5) + elmk_log((1.0 + chik2) * 0.5) - 2.0 * elmk_atan(chik) + ELM_PI * 0.5;
}
__device__ __forceinline__ double stab2(double zeta)
{
  const double chik2 = sqrt(1.0 - 16.0 * zeta);
  return 2.0 * elmk_log((1.0 + chik2) * 0.5);
}

// :36-61
__device__ __forceinline__ void monin_obukhov_length(double ur, double thv, double dthv, double zldis, double z0m,
                                                     double& um, double& obu)
{
  const double wc = 0.5;
  if (dthv >= 0.0) {
    um = dmax(ur, 0.1);
  } else {
    um = sqrt(ur * ur + wc * wc);
  }
  const double rib = GRAV * zldis * dthv / (thv * um * um);
  double zeta;
  if (rib >= 0.0) {
    zeta = rib * elmk_log(zldis / z0m) / (1.0 - 5.0 * dmin(rib, 0.19));
    zeta = dmin(2.0, dmax(zeta, 0.01));
  } else {
    zeta = rib * elmk_log(zldis / z0m);
    zeta = dmax(-100.0, dmin(zeta, -0.01));
  }
  obu = zldis / zeta;
}

// :64-83
__device__ __forceinline__ double fv_wind(double forc_hgt_u, double displa, double um, double obu, double z0m)
{
  const double zetam = 1.574;
  const double zldis = forc_hgt_u - displa;
  const double zeta = zldis / obu;
  if (zeta < (-zetam)) {
    return VKC * um /
           (elmk_log(-zetam * obu / z0m) - stab1(-zetam) + stab1(z0m / obu) + 1.14 * (elmk_pow((-zeta), 0.333) - elmk_pow(zetam, 0.333)));
  } else if (zeta < 0.0) {
    return VKC * um / (elmk_log(zldis / z0m) - stab1(zeta) + stab1(z0m / obu));
  } else if (zeta <= 1.0) {
    return VKC * um / (elmk_log(zldis / z0m) + 5.0 * zeta - 5.0 * z0m / obu);
  }
  return VKC * um / (elmk_log(obu / z0m) + 5.0 - 5.0 * z0m / obu + (5.0 * elmk_log(zeta) + zeta - 1.0));
}

// the temperature/humidity profile relation shared by :86-172 (zetat = 0.465); GROUPED selects the
// "5.0 * (z0h / obu)" grouping that only friction_velocity_temp2m's last branch has (:148)
template <bool GROUPED>
__device__ __forceinline__ double fv_profile(double zldis, double obu, double z0)
{
  const double zetat = 0.465;
  const double zeta = zldis / obu;
  if (zeta < -zetat) {
    return VKC / (elmk_log(-zetat * obu / z0) - stab2(-zetat) + stab2(z0 / obu) + 0.8 * (elmk_pow(zetat, -0.333) - elmk_pow((-zeta), -0.333)));
  } else if (zeta < 0.0) {
    return VKC / (elmk_log(zldis / z0) - stab2(zeta) + stab2(z0 / obu));
  } else if (zeta <= 1.0) {
    return VKC / (elmk_log(zldis / z0) + 5.0 * zeta - 5.0 * z0 / obu);
  }
  if (GROUPED) return VKC / (elmk_log(obu / z0) + 5.0 - 5.0 * (z0 / obu) + (5.0 * elmk_log(zeta) + zeta - 1.0));
  return VKC / (elmk_log(obu / z0) + 5.0 - 5.0 * z0 / obu + (5.0 * elmk_log(zeta) + zeta - 1.0));
}

// stab1(x) and stab2(x) of the same argument: stab2(x) is twice the second logarithm of stab1(x) (:17-33)
__device__ __forceinline__ void stab12(double zeta, double& s1, double& s2)
{
  const double chik2 = sqrt(1.0 - 16.0 * zeta);
  const double chik = sqrt(chik2);
  const double lg2 = elmk_log((1.0 + chik2) * 0.5);
  s1 = 2.0 * elmk_log((1.0 + chik) * 0.5) + lg2 - 2.0 * elmk_atan(chik) + ELM_PI * 0.5;
  s2 = 2.0 * lg2;
}

// zetam^0.333 and zetat^-0.333 of the very unstable regime (:70, :91); evaluated once per kernel
struct FvConst {
  double pw_m, pw_t;
};
__device__ __forceinline__ FvConst fv_const()
{
  FvConst k;
  k.pw_m = elmk_pow(1.574, 0.333);
  k.pw_t = elmk_pow(0.465, -0.333);
  return k;
}

// The five calls that open every stability iteration (bareground_fluxes_impl.hh:52-57, canopy_fluxes_impl.hh:235-240):
// friction_velocity_wind :64, _temp :86, _humidity :107, _temp2m :134, _humidity2m :153.
//
// Each profile has four stability regimes with different transcendental calls, and the lanes of a wave sit in
// different regimes, so a call-by-call transcription executes nearly every branch of every call.  Here the three
// distinct profiles (wind, temperature, 2 m temperature; the humidity ones equal the temperature ones when their
// heights and roughness lengths do, as the reference itself short-cuts) are evaluated together: one logarithm per
// profile on a regime-selected argument, ONE unstable block (all three profiles are unstable together, zeta has
// the sign of obu) in which stab1/stab2 of z0/obu are shared, one very-unstable block for the pow terms and one
// very-stable block for log(zeta).  Every lane still evaluates exactly the reference's expression for its regime
// on the same operands, so results are bit-identical to the call-by-call form (fv_wind / fv_profile above).
// SAME_Z0: z0m, z0h and z0q are the same value (canopy); otherwise z0h == z0q is still checked at run time.
// WITH_2M = false leaves temp12m / temp22m untouched (the caller evaluates the 2 m profile later with fv_profile).
// friction_profiles_zl takes the three displaced heights zl_x = forc_hgt_x - displa (what the profiles read) and whether
// hgt_q == hgt_t (the reference's short-cut test of friction_velocity_humidity, :107-113)
template <bool SAME_Z0, bool WITH_2M = true>
__device__ __forceinline__ void friction_profiles_zl(double zl_u, double zl_t, double zl_q, bool same_tq, double um,
                                                     double obu, double z0m, double z0h, double z0q, const FvConst& K,
                                                     double& ustar, double& temp1, double& temp2, double& temp12m,
                                                     double& temp22m)
{
  const double zetam = 1.574, zetat = 0.465;
  const double ze_u = zl_u / obu;  // wind
  const double ze_t = zl_t / obu;  // temperature
  const double zl_2 = 2.0 + z0h, ze_2 = zl_2 / obu;       // 2 m temperature
  // regimes in the reference's test order: very unstable, unstable, stable (zeta <= 1), else very stable
  const bool u1 = ze_u < -zetam, u2 = !u1 && ze_u < 0.0, u3 = !u1 && !u2 && ze_u <= 1.0;
  const bool t1 = ze_t < -zetat, t2 = !t1 && ze_t < 0.0, t3 = !t1 && !t2 && ze_t <= 1.0;
  const bool b1 = WITH_2M && ze_2 < -zetat, b2 = WITH_2M && !b1 && ze_2 < 0.0, b3 = !WITH_2M || (!b1 && !b2 && ze_2 <= 1.0);

  const double au = u1 ? (-zetam * obu / z0m) : ((u2 || u3) ? (zl_u / z0m) : (obu / z0m));
  const double at = t1 ? (-zetat * obu / z0h) : ((t2 || t3) ? (zl_t / z0h) : (obu / z0h));
  const double a2 = b1 ? (-zetat * obu / z0h) : ((b2 || b3) ? (zl_2 / z0h) : (obu / z0h));
  const double Lu = elmk_log(au);
  double Lt = Lu;
  if (!(at == au)) Lt = elmk_log(at);
  double L2 = Lt;
  if (WITH_2M && !(a2 == at)) L2 = elmk_log(a2);

  double su = 0.0, st = 0.0, s2 = 0.0, sz1 = 0.0, sz2 = 0.0;
  if (u1 || u2 || t1 || t2 || b1 || b2) {
    if (SAME_Z0) {
      stab12(z0m / obu, sz1, sz2);
    } else {
      sz1 = stab1(z0m / obu);
      sz2 = stab2(z0h / obu);
    }
    su = stab1(u1 ? -zetam : ze_u);
    const double xt = t1 ? -zetat : ze_t;
    const double x2 = b1 ? -zetat : ze_2;
    st = stab2(xt);
    s2 = st;
    if (WITH_2M && !(x2 == xt)) s2 = stab2(x2);
  }
  double pu = 0.0, pt = 0.0, p2 = 0.0;
  if (u1 || t1 || b1) {
    if (u1) pu = elmk_pow((-ze_u), 0.333);
    if (t1) pt = elmk_pow((-ze_t), -0.333);
    p2 = pt;
    if (b1 && !(ze_2 == ze_t && t1)) p2 = elmk_pow((-ze_2), -0.333);
  }
  const bool u4 = !u1 && !u2 && !u3, t4 = !t1 && !t2 && !t3, b4 = !b1 && !b2 && !b3;
  double lu = 0.0, lt = 0.0, l2 = 0.0;
  if (u4 || t4 || b4) {
    if (u4) lu = elmk_log(ze_u);
    lt = lu;
    if (t4 && !(ze_t == ze_u && u4)) lt = elmk_log(ze_t);
    if (b4) l2 = elmk_log(ze_2);
  }

  double du, dt, d2;
  if (u1) {
    du = Lu - su + sz1 + 1.14 * (pu - K.pw_m);
  } else if (u2) {
    du = Lu - su + sz1;
  } else if (u3) {
    du = Lu + 5.0 * ze_u - 5.0 * z0m / obu;
  } else {
    du = Lu + 5.0 - 5.0 * z0m / obu + (5.0 * lu + ze_u - 1.0);
  }
  if (t1) {
    dt = Lt - st + sz2 + 0.8 * (K.pw_t - pt);
  } else if (t2) {
    dt = Lt - st + sz2;
  } else if (t3) {
    dt = Lt + 5.0 * ze_t - 5.0 * z0h / obu;
  } else {
    dt = Lt + 5.0 - 5.0 * z0h / obu + (5.0 * lt + ze_t - 1.0);
  }
  if (b1) {
    d2 = L2 - s2 + sz2 + 0.8 * (K.pw_t - p2);
  } else if (b2) {
    d2 = L2 - s2 + sz2;
  } else if (b3) {
    d2 = L2 + 5.0 * ze_2 - 5.0 * z0h / obu;
  } else {
    d2 = L2 + 5.0 - 5.0 * (z0h / obu) + (5.0 * l2 + ze_2 - 1.0);  // the grouping only friction_velocity_temp2m has (:148)
  }
  ustar = VKC * um / du;
  temp1 = VKC / dt;
  if (WITH_2M) temp12m = VKC / d2;
  if (same_tq && z0q == z0h) {  // friction_velocity_humidity :107
    temp2 = temp1;
  } else {
    temp2 = fv_profile<false>(zl_q, obu, z0q);
  }
  if (WITH_2M) {
    if (z0q == z0h) {  // friction_velocity_humidity2m :153
      temp22m = temp12m;
    } else {
      temp22m = fv_profile<false>(2.0 + z0q, obu, z0q);
    }
  }
}
template <bool SAME_Z0, bool WITH_2M = true>
__device__ __forceinline__ void friction_profiles(double hgt_u, double hgt_t, double hgt_q, double displa, double um,
                                                  double obu, double z0m, double z0h, double z0q, const FvConst& K,
                                                  double& ustar, double& temp1, double& temp2, double& temp12m,
                                                  double& temp22m)
{
  friction_profiles_zl<SAME_Z0, WITH_2M>(hgt_u - displa, hgt_t - displa, hgt_q - displa, hgt_q == hgt_t, um, obu, z0m, z0h, z0q,
                                         K, ustar, temp1, temp2, temp12m, temp22m);
}

}  // namespace elmk
