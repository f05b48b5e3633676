// elmk_api.cpp - host side of the C ABI declared in include/elmk.h.
//
// Owns the device arena (every field of elmk_fields.def as SoA [lev][column], level stride padded to 64
// columns, each field 256-byte aligned), the parameter block (DevState) mirrored into device memory, one
// HIP stream, and a staging buffer for layout conversion.  No physics lives here and there is no CPU path:
// every elmk_<physics>() is a single kernel launch on the context's stream.
#include "elmk.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <vector>

#include "elmk_dev.h"
#include "elmk_kernels.h"

using namespace elmk;

namespace {

struct FieldDesc {
  const char* name;
  int dtype;
  int nlev;
};

const FieldDesc g_fields[ELMK_NUM_FIELDS] = {
#define ELMK_FIELD(name, T, nlev) {#name, ELMK_##T, nlev},
#include "elmk_fields.def"
#undef ELMK_FIELD
    {"err_flags", ELMK_U32, 1},
};

inline int elem_size(int dtype) { return dtype == ELMK_F64 ? 8 : (dtype == ELMK_U8 ? 1 : 4); }
// bytes of one element as it is STORED on the device: the report-only ELMK_STATE_F32 build (libelmk_f32.so, BASELINE config 5)
// keeps every fp64 state field as fp32 (elmk_dev.h: field_of); the C ABI still speaks double
#ifdef ELMK_STATE_F32
constexpr bool kStateF32 = true;
#else
constexpr bool kStateF32 = false;
#endif
inline int store_size(int dtype) { return (kStateF32 && dtype == ELMK_F64) ? 4 : elem_size(dtype); }
inline int store_dtype(int dtype) { return (kStateF32 && dtype == ELMK_F64) ? ELMK_F32_STORED : dtype; }
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

thread_local std::string g_create_error;
constexpr int MAXLEV_STAGE = 21;  // widest field (zisoi)

}  // namespace

struct GraphSlot {
  hipGraphExec_t exec = nullptr;
  double dt = 0.0;
  hipStream_t stream = nullptr;
};

struct elmk_ctx {
  int dev = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  SideStreams side{};
  int64_t ncols = 0;
  int64_t ld = 0;
  DevState h;            // host mirror of the device parameter block
  DevState* d = nullptr; // device copy handed to kernels
  bool dirty = true;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  void* fptr[ELMK_NUM_FIELDS] = {};
  double* snicar = nullptr;
  double* snowage = nullptr;  // SnwRdsTable (elmk_set_snow_age_tables)
  char* scratch = nullptr;  // work arrays + work lists + queue counters of the compacted kernels
  size_t scratch_bytes = 0;
  char* staging = nullptr;  // device staging for layout conversion
  size_t staging_bytes = 0;
  std::vector<int> snap_fields;  // elmk_snapshot_fields
  std::vector<char*> snap_bufs;
  uint32_t* red_or = nullptr;  // device scalars for elmk_error_summary
  long long* red_first = nullptr;
  // elmk_set_graph: the seven wrappers of elmk_timestep7 captured once as a HIP graph (kernel nodes + the side-stream
  // fork / join of albedo_snicar) and replayed; key = (dt, stream)
  bool use_graph = false;
  bool have_init_params = false;
  GraphSlot graph[3];  // [0] elmk_timestep7, [1] elmk_timestep7_fused, [2] elmk_advance_physics
  // A HIP error may have cut a step short between the kernel that fills a work list and the one that drains it and leaves it
  // empty (the lists have no reset launch of their own): the next physics call zeroes every list counter first.
  bool lists_stale = false;
  char* counters_raw = nullptr;
  size_t counters_bytes = 0;
  std::string err;
};

namespace {

bool hip_fail(elmk_ctx* ctx, hipError_t e, const char* what)
{
  if (e == hipSuccess) return false;
  char buf[512];
  snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  if (ctx) {
    ctx->err = buf;
    ctx->lists_stale = true;
  }
  g_create_error = buf;
  return true;
}

#define HIPCHK(call)                                      \
  do {                                                    \
    if (hip_fail(ctx, (call), #call)) return ELMK_E_HIP;  \
  } while (0)

// Owners for the temporaries of the diagnostic entry points: released on every return path
struct EventList {
  std::vector<hipEvent_t> ev;
  hipError_t create(size_t n)
  {
    ev.reserve(n);
    for (size_t i = 0; i < n; i++) {
      hipEvent_t e = nullptr;
      const hipError_t rc = hipEventCreate(&e);
      if (rc != hipSuccess) return rc;
      ev.push_back(e);
    }
    return hipSuccess;
  }
  hipEvent_t& operator[](size_t i) { return ev[i]; }
  ~EventList()
  {
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  }
};
struct DevBuf {
  void* p = nullptr;
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
  ~DevBuf()
  {
    if (p) (void)hipFree(p);
  }
};

int invalid(elmk_ctx* ctx, const char* msg)
{
  if (ctx) ctx->err = msg;
  g_create_error = msg;
  return ELMK_E_INVALID;
}

int push_params(elmk_ctx* ctx)
{
  if (!ctx->dirty) return ELMK_OK;
  HIPCHK(hipMemcpyAsync(ctx->d, &ctx->h, sizeof(DevState), hipMemcpyHostToDevice, ctx->stream));
  // the source is pageable host memory: the runtime has staged it before returning, so h may change again
  ctx->dirty = false;
  return ELMK_OK;
}

int enter(elmk_ctx* ctx)
{
  if (!ctx) return ELMK_E_INVALID;
  HIPCHK(hipSetDevice(ctx->dev));
  return ELMK_OK;
}

bool field_ok(int f) { return f >= 0 && f < ELMK_NUM_FIELDS; }

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------------
// lifetime
// ---------------------------------------------------------------------------------------------------
int elmk_create(int64_t ncols, int device_id, elmk_ctx** out)
{
  if (!out || ncols < 0) return invalid(nullptr, "elmk_create: bad arguments");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_error = std::string("elmk_create: no HIP device (") + hipGetErrorString(e) +
                     "); libelmk has no CPU fallback";
    return ELMK_E_NO_DEVICE;
  }
  if (device_id < 0 || device_id >= ndev) {
    g_create_error = "elmk_create: device id out of range";
    return ELMK_E_NO_DEVICE;
  }
  elmk_ctx* ctx = new (std::nothrow) elmk_ctx();
  if (!ctx) return ELMK_E_NOMEM;
  ctx->dev = device_id;
  ctx->ncols = ncols;
  ctx->ld = (int64_t)align_up((size_t)(ncols > 0 ? ncols : 1), 64);

  auto fail = [&](int code) {
    elmk_destroy(ctx);
    return code;
  };
  if (hip_fail(ctx, hipSetDevice(device_id), "hipSetDevice")) return fail(ELMK_E_HIP);
  if (hip_fail(ctx, hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking), "hipStreamCreate"))
    return fail(ELMK_E_HIP);
  ctx->stream = ctx->own_stream;
  for (int i = 0; i < ELMK_NSIDE; i++) {
    if (hip_fail(ctx, hipStreamCreateWithFlags(&ctx->side.s[i], hipStreamNonBlocking), "hipStreamCreate(side)") ||
        hip_fail(ctx, hipEventCreateWithFlags(&ctx->side.join[i], hipEventDisableTiming), "hipEventCreate"))
      return fail(ELMK_E_HIP);
  }
  if (hip_fail(ctx, hipEventCreateWithFlags(&ctx->side.fork, hipEventDisableTiming), "hipEventCreate")) return fail(ELMK_E_HIP);

  // arena layout
  size_t off = 0;
  size_t foff[ELMK_NUM_FIELDS];
  for (int f = 0; f < ELMK_NUM_FIELDS; f++) {
    foff[f] = off;
    off += align_up((size_t)g_fields[f].nlev * (size_t)ctx->ld * store_size(g_fields[f].dtype), 256);
  }
  ctx->arena_bytes = off;
  if (hip_fail(ctx, hipMalloc((void**)&ctx->arena, off), "hipMalloc(state arena)")) return fail(ELMK_E_NOMEM);
  if (hip_fail(ctx, hipMemsetAsync(ctx->arena, 0, off, ctx->stream), "hipMemset(state arena)")) return fail(ELMK_E_HIP);
  for (int f = 0; f < ELMK_NUM_FIELDS; f++) ctx->fptr[f] = ctx->arena + foff[f];

  if (hip_fail(ctx, hipMalloc((void**)&ctx->snicar, SN_TOTAL * sizeof(double)), "hipMalloc(snicar)"))
    return fail(ELMK_E_NOMEM);
  if (hip_fail(ctx, hipMemsetAsync(ctx->snicar, 0, SN_TOTAL * sizeof(double), ctx->stream), "hipMemset(snicar)"))
    return fail(ELMK_E_HIP);
  if (hip_fail(ctx, hipMalloc((void**)&ctx->snowage, 3 * ELMK_SNOWAGE_N * sizeof(double)), "hipMalloc(snowage)"))
    return fail(ELMK_E_NOMEM);
  if (hip_fail(ctx, hipMemsetAsync(ctx->snowage, 0, 3 * ELMK_SNOWAGE_N * sizeof(double), ctx->stream), "hipMemset(snowage)"))
    return fail(ELMK_E_HIP);
  if (hip_fail(ctx, hipMalloc((void**)&ctx->d, sizeof(DevState)), "hipMalloc(params)")) return fail(ELMK_E_NOMEM);
  const size_t wk_bytes = align_up((size_t)WK_N * (size_t)ctx->ld * 8, 256);
  const size_t list_bytes = align_up((size_t)NLISTS * (size_t)ctx->ld * 4, 256);
  const size_t cnt_bytes = align_up((size_t)(2 * NLISTS + CF_NCLS) * CPAD * 4, 256);
  const size_t hint_bytes = align_up((size_t)ctx->ld * 4, 256);
  // canopy_fluxes queue records (k_canopy_fluxes.hip), by queue position
  const int64_t cf_nblk = (ncols + 255) / 256 > 0 ? (ncols + 255) / 256 : 1;
  const size_t rec_bytes = align_up((size_t)CF_REC_N * (size_t)(ctx->ld + 8) * 8, 256);
  const size_t fin_bytes = align_up((size_t)CF_FIN_N * (size_t)(ctx->ld + 8) * 8, 256);
  const size_t irec_bytes = align_up((size_t)CF_IREC_N * (size_t)ctx->ld * 4, 256);
  const size_t pos_bytes = align_up((size_t)ctx->ld * 4, 256);
  const size_t blk_bytes = align_up((size_t)CF_NCLS * (size_t)cf_nblk * 4, 256);
  const size_t cls_bytes = align_up((size_t)ctx->ld, 256);
  const size_t given_bytes = align_up((size_t)3 * (size_t)ctx->ld * 8, 256);
  const size_t snow_bytes = align_up((size_t)28 * (size_t)ctx->ld * 8, 256);
  const size_t cons_bytes = align_up((size_t)8 * (size_t)ctx->ld * 8 + (size_t)8 * ELMK_CONS_NPART * 3 * 8 + 8 * 3 * 8, 256);
  ctx->scratch_bytes = wk_bytes + list_bytes + cnt_bytes + hint_bytes + rec_bytes + fin_bytes + irec_bytes + pos_bytes + blk_bytes + cls_bytes + given_bytes + snow_bytes + cons_bytes;
  if (hip_fail(ctx, hipMalloc((void**)&ctx->scratch, ctx->scratch_bytes), "hipMalloc(scratch)")) return fail(ELMK_E_NOMEM);
  if (hip_fail(ctx, hipMemsetAsync(ctx->scratch, 0, ctx->scratch_bytes, ctx->stream), "hipMemset(scratch)"))
    return fail(ELMK_E_HIP);
  if (hip_fail(ctx, hipMalloc((void**)&ctx->red_or, 16), "hipMalloc(reduce)")) return fail(ELMK_E_NOMEM);
  ctx->red_first = (long long*)((char*)ctx->red_or + 8);

  // staging: up to 32 MiB, at least one 64-column tile of the widest field
  size_t want = (size_t)MAXLEV_STAGE * 8 * (size_t)(ncols > 0 ? ncols : 1);
  if (want > ((size_t)32 << 20)) want = (size_t)32 << 20;
  if (want < (size_t)MAXLEV_STAGE * 8 * 64) want = (size_t)MAXLEV_STAGE * 8 * 64;
  ctx->staging_bytes = want;
  if (hip_fail(ctx, hipMalloc((void**)&ctx->staging, want), "hipMalloc(staging)")) return fail(ELMK_E_NOMEM);

  // parameter block defaults: LandType() (land_data.h:38) and ELMState scalars (elm_state.h:221-224)
  DevState& h = ctx->h;
  memset(&h, 0, sizeof h);
  h.ncols = ncols;
  h.ld = ctx->ld;
  h.land = Land{1, 0, 2, 0, 0};
  h.dewmx = 0.1;
  h.oldfflag = 1;
  h.snicar = (gptr<const double>)ctx->snicar;
  h.snowage = (gptr<const double>)ctx->snowage;
  h.wk = (gptr<double>)ctx->scratch;
  h.lists = (gptr<int32_t>)(ctx->scratch + wk_bytes);
  h.counters = (gptr<uint32_t>)(ctx->scratch + wk_bytes + list_bytes);
  ctx->counters_raw = ctx->scratch + wk_bytes + list_bytes;
  ctx->counters_bytes = cnt_bytes;
  h.cf_niter = (gptr<int32_t>)(ctx->scratch + wk_bytes + list_bytes + cnt_bytes);
  {
    char* q = ctx->scratch + wk_bytes + list_bytes + cnt_bytes + hint_bytes;
    h.cf_rec = (gptr<double>)q;
    q += rec_bytes;
    h.cf_fin = (gptr<double>)q;
    q += fin_bytes;
    h.cf_irec = (gptr<int32_t>)q;
    q += irec_bytes;
    h.cf_pos = (gptr<int32_t>)q;
    q += pos_bytes;
    h.cf_blk = (gptr<uint32_t>)q;
    q += blk_bytes;
    h.cf_cls = (gptr<int8_t>)q;
    q += cls_bytes;
    h.cf_given = (gptr<double>)q;
    q += given_bytes;
    h.alb_snow = (gptr<double>)q;
    q += snow_bytes;
    h.cons_diag = (gptr<double>)q;
    h.cf_nblk = cf_nblk;
  }
  {
    int f = 0;
#define ELMK_FIELD(name, T, nlev) h.name = field_of<ELMK_##T>::from(ctx->fptr[f++]);
#include "elmk_fields.def"
#undef ELMK_FIELD
    h.err_flags = (gptr<uint32_t>)ctx->fptr[f];
  }
  ctx->dirty = true;
  if (hip_fail(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize")) return fail(ELMK_E_HIP);
  *out = ctx;
  return ELMK_OK;
}

int elmk_destroy(elmk_ctx* ctx)
{
  if (!ctx) return ELMK_OK;
  (void)hipSetDevice(ctx->dev);
  if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
  if (ctx->arena) (void)hipFree(ctx->arena);
  if (ctx->snicar) (void)hipFree(ctx->snicar);
  if (ctx->snowage) (void)hipFree(ctx->snowage);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->d) (void)hipFree(ctx->d);
  for (GraphSlot& g : ctx->graph)
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
  if (ctx->red_or) (void)hipFree(ctx->red_or);
  if (ctx->staging) (void)hipFree(ctx->staging);
  for (char* b : ctx->snap_bufs) (void)hipFree(b);
  for (int i = 0; i < ELMK_NSIDE; i++) {
    if (ctx->side.s[i]) {
      (void)hipStreamSynchronize(ctx->side.s[i]);
      (void)hipStreamDestroy(ctx->side.s[i]);
    }
    if (ctx->side.join[i]) (void)hipEventDestroy(ctx->side.join[i]);
  }
  if (ctx->side.fork) (void)hipEventDestroy(ctx->side.fork);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return ELMK_OK;
}

const char* elmk_last_error(const elmk_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int elmk_set_stream(elmk_ctx* ctx, void* hip_stream)
{
  if (int rc = enter(ctx)) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return ELMK_OK;
}

int elmk_set_graph(elmk_ctx* ctx, int on)
{
  if (int rc = enter(ctx)) return rc;
  ctx->use_graph = on != 0;
  if (!ctx->use_graph) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (GraphSlot& g : ctx->graph) {
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
      g.exec = nullptr;
    }
  }
  return ELMK_OK;
}

int elmk_set_option(elmk_ctx* ctx, int option, int value)
{
  if (int rc = enter(ctx)) return rc;
  if (option != ELMK_OPT_CF_HALF_WORKGROUPS) return invalid(ctx, "elmk_set_option: unknown option");
  int cus = 0;
  HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->dev));
  const int want = value ? (cus > 0 ? cus : 256) : 0;
  if (want != ctx->side.cf_half_groups) {
    // a captured graph holds the old launch shape
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (GraphSlot& g : ctx->graph) {
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
      g.exec = nullptr;
    }
    ctx->side.cf_half_groups = want;
  }
  return ELMK_OK;
}

int elmk_sync(elmk_ctx* ctx)
{
  if (int rc = enter(ctx)) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return ELMK_OK;
}

int64_t elmk_ncols(const elmk_ctx* ctx) { return ctx ? ctx->ncols : -1; }
int64_t elmk_level_stride(const elmk_ctx* ctx) { return ctx ? ctx->ld : -1; }
int64_t elmk_device_bytes(const elmk_ctx* ctx)
{
  return ctx ? (int64_t)(ctx->arena_bytes + ctx->staging_bytes + ctx->scratch_bytes + (SN_TOTAL + 3 * ELMK_SNOWAGE_N) * sizeof(double) + sizeof(DevState)) : -1;
}

// ---------------------------------------------------------------------------------------------------
// schema
// ---------------------------------------------------------------------------------------------------
int elmk_num_fields(void) { return ELMK_NUM_FIELDS; }
const char* elmk_field_name(int field) { return field_ok(field) ? g_fields[field].name : nullptr; }
int elmk_field_id(const char* name)
{
  if (!name) return -1;
  for (int f = 0; f < ELMK_NUM_FIELDS; f++)
    if (strcmp(name, g_fields[f].name) == 0) return f;
  return -1;
}
int elmk_field_info(int field, int* nlev, int* dtype)
{
  if (!field_ok(field)) return ELMK_E_INVALID;
  if (nlev) *nlev = g_fields[field].nlev;
  if (dtype) *dtype = g_fields[field].dtype;
  return ELMK_OK;
}

// ---------------------------------------------------------------------------------------------------
// data movement
// ---------------------------------------------------------------------------------------------------
static int xfer_stored(elmk_ctx* ctx, int field, void* host, int64_t col0, int64_t n, int layout, bool up);

static int xfer(elmk_ctx* ctx, int field, void* host, int64_t col0, int64_t n, int layout, bool up)
{
  if (int rc = enter(ctx)) return rc;
  if (!field_ok(field) || (!host && n > 0) || col0 < 0 || n < 0 || col0 + n > ctx->ncols)
    return invalid(ctx, "elmk_upload/download: bad field or column range");
  if (n == 0) return ELMK_OK;
  // snl indexes the level arrays (top = nlevsno - snl) in every snow and soil kernel, in global memory and in LDS packs: a
  // value outside 0..nlevsno is refused at the two doors host values come through (here and elmk_fill) instead of being read
  // out of bounds on the device (the reference has the same undefined behaviour, but no such door)
  if (up && field == ELMK_FIELD_snl) {
    const int32_t* v = (const int32_t*)host;
    for (int64_t i = 0; i < n; i++)
      if (v[i] < 0 || v[i] > NLEVSNO) return invalid(ctx, "elmk_upload: snl outside 0..nlevsno");
  }
  if (kStateF32 && g_fields[field].dtype == ELMK_F64) {
    // fp32-state build: the caller's doubles are rounded to the stored fp32 on the way in and widened on the way out (on the
    // host: this build is a measurement variant, its benchmark tiles a small uploaded block on the device)
    const size_t cnt = (size_t)n * (size_t)g_fields[field].nlev;
    std::vector<float> tmp(cnt);
    double* h = (double*)host;
    if (up)
      for (size_t i = 0; i < cnt; i++) tmp[i] = (float)h[i];
    const int rc = xfer_stored(ctx, field, tmp.data(), col0, n, layout, up);
    if (rc == ELMK_OK && !up)
      for (size_t i = 0; i < cnt; i++) h[i] = (double)tmp[i];
    return rc;
  }
  return xfer_stored(ctx, field, host, col0, n, layout, up);
}

// host elements already in the stored element type
static int xfer_stored(elmk_ctx* ctx, int field, void* host, int64_t col0, int64_t n, int layout, bool up)
{
  const int es = store_size(g_fields[field].dtype), nlev = g_fields[field].nlev;
  char* dev = (char*)ctx->fptr[field];
  if (layout == ELMK_LAYOUT_SOA || nlev == 1) {
    // rows of n elements <-> rows of ld elements
    if (up)
      HIPCHK(hipMemcpy2DAsync(dev + (size_t)col0 * es, (size_t)ctx->ld * es, host, (size_t)n * es, (size_t)n * es, nlev,
                              hipMemcpyHostToDevice, ctx->stream));
    else
      HIPCHK(hipMemcpy2DAsync(host, (size_t)n * es, dev + (size_t)col0 * es, (size_t)ctx->ld * es, (size_t)n * es, nlev,
                              hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return ELMK_OK;
  }
  if (layout != ELMK_LAYOUT_COL_MAJOR) return invalid(ctx, "elmk_upload/download: unknown layout");
  // reference layout [col][lev]: go through the device staging buffer in chunks of whole 64-column tiles
  int64_t chunk = (int64_t)(ctx->staging_bytes / ((size_t)nlev * es));
  chunk = chunk / 64 * 64;
  if (chunk <= 0) return invalid(ctx, "staging buffer too small");
  for (int64_t done = 0; done < n; done += chunk) {
    const int64_t m = (n - done) < chunk ? (n - done) : chunk;
    char* hp = (char*)host + (size_t)done * nlev * es;
    if (up) {
      HIPCHK(hipMemcpyAsync(ctx->staging, hp, (size_t)m * nlev * es, hipMemcpyHostToDevice, ctx->stream));
      launch_cols_to_soa(ctx->staging, dev, es, nlev, ctx->ld, col0 + done, m, ctx->stream);
    } else {
      launch_soa_to_cols(dev, ctx->staging, es, nlev, ctx->ld, col0 + done, m, ctx->stream);
      HIPCHK(hipMemcpyAsync(hp, ctx->staging, (size_t)m * nlev * es, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));  // staging is reused by the next chunk
  }
  return ELMK_OK;
}

int elmk_upload(elmk_ctx* ctx, int field, const void* host, int64_t col0, int64_t n, int layout)
{
  return xfer(ctx, field, const_cast<void*>(host), col0, n, layout, true);
}
int elmk_download(elmk_ctx* ctx, int field, void* host, int64_t col0, int64_t n, int layout)
{
  return xfer(ctx, field, host, col0, n, layout, false);
}

int elmk_fill(elmk_ctx* ctx, int field, double value)
{
  if (int rc = enter(ctx)) return rc;
  if (!field_ok(field)) return invalid(ctx, "elmk_fill: bad field");
  if (field == ELMK_FIELD_snl && !(value >= 0.0 && value <= (double)NLEVSNO))
    return invalid(ctx, "elmk_fill: snl outside 0..nlevsno");
  launch_fill(ctx->fptr[field], store_dtype(g_fields[field].dtype), g_fields[field].nlev, ctx->ld, ctx->ncols, value, ctx->stream);
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}

void* elmk_device_ptr(elmk_ctx* ctx, int field) { return (ctx && field_ok(field)) ? ctx->fptr[field] : nullptr; }

int elmk_tile_columns(elmk_ctx* ctx, int64_t nbase, uint64_t seed, int nrules, const elmk_perturb* rules)
{
  if (int rc = enter(ctx)) return rc;
  if (nbase <= 0 || nbase > ctx->ncols || nrules < 0 || (nrules > 0 && !rules))
    return invalid(ctx, "elmk_tile_columns: bad arguments");
  for (int f = 0; f < ELMK_NUM_FIELDS; f++) {
    int mode = -1;
    double amp = 0.0;
    for (int r = 0; r < nrules; r++) {
      if (rules[r].field == f) {
        mode = rules[r].mode;
        amp = rules[r].amp;
      }
    }
    launch_tile(ctx->fptr[f], store_dtype(g_fields[f].dtype), g_fields[f].nlev, ctx->ld, ctx->ncols, nbase, seed, f, mode, amp,
                ctx->stream);
  }
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}

int elmk_snapshot_fields(elmk_ctx* ctx, const int* fields, int nfields)
{
  if (int rc = enter(ctx)) return rc;
  if (nfields < 0 || (nfields > 0 && !fields)) return invalid(ctx, "elmk_snapshot_fields: bad arguments");
  for (int i = 0; i < nfields; i++)
    if (!field_ok(fields[i])) return invalid(ctx, "elmk_snapshot_fields: unknown field");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (char* b : ctx->snap_bufs) (void)hipFree(b);
  ctx->snap_bufs.clear();
  ctx->snap_fields.clear();
  // a field enters the snapshot only once its buffer exists and its copy has been enqueued; any failure drops the whole
  // snapshot, so that a later elmk_restore_fields never restores from a partly filled set
  auto drop = [&]() {
    (void)hipStreamSynchronize(ctx->stream);
    for (char* b : ctx->snap_bufs) (void)hipFree(b);
    ctx->snap_bufs.clear();
    ctx->snap_fields.clear();
  };
  for (int i = 0; i < nfields; i++) {
    const int f = fields[i];
    const size_t bytes = (size_t)g_fields[f].nlev * (size_t)ctx->ld * store_size(g_fields[f].dtype);
    char* b = nullptr;
    if (hip_fail(ctx, hipMalloc((void**)&b, bytes), "hipMalloc(snapshot)")) {
      drop();
      return ELMK_E_NOMEM;
    }
    if (hip_fail(ctx, hipMemcpyAsync(b, ctx->fptr[f], bytes, hipMemcpyDeviceToDevice, ctx->stream), "hipMemcpyAsync(snapshot)")) {
      (void)hipFree(b);
      drop();
      return ELMK_E_HIP;
    }
    ctx->snap_bufs.push_back(b);
    ctx->snap_fields.push_back(f);
  }
  if (hip_fail(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize(snapshot)")) {
    drop();
    return ELMK_E_HIP;
  }
  return ELMK_OK;
}

int elmk_restore_fields(elmk_ctx* ctx)
{
  if (int rc = enter(ctx)) return rc;
  // plain streaming kernels (hipMemcpyAsync device-to-device goes through the SDMA engines here, ~80 GB/s), up to
  // COPY_JOBS_MAX fields per launch: a small field's copy is all launch latency
  CopyJobs J;
  J.n = 0;
  int64_t words = 0;
  for (size_t i = 0; i < ctx->snap_fields.size(); i++) {
    const int f = ctx->snap_fields[i];
    const size_t bytes = (size_t)g_fields[f].nlev * (size_t)ctx->ld * store_size(g_fields[f].dtype);
    words += (int64_t)(bytes / 8);
    J.src[J.n] = (const double*)ctx->snap_bufs[i];
    J.dst[J.n] = (double*)ctx->fptr[f];
    J.end[J.n] = words;
    J.n++;
    if (J.n == COPY_JOBS_MAX || i + 1 == ctx->snap_fields.size()) {
      launch_copy_multi(J, ctx->stream);
      J.n = 0;
      words = 0;
    }
  }
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}

// ---------------------------------------------------------------------------------------------------
// parameters
// ---------------------------------------------------------------------------------------------------
int elmk_set_land(elmk_ctx* ctx, int ltype, int ctype, int vtype, int urbpoi, int lakpoi)
{
  if (!ctx) return ELMK_E_INVALID;
  if (vtype < 0 || vtype >= ELMK_MXPFT) return invalid(ctx, "elmk_set_land: vtype out of range");
  ctx->h.land = Land{ltype, ctype, vtype, urbpoi != 0, lakpoi != 0};
  ctx->dirty = true;
  return ELMK_OK;
}

int elmk_set_scalars(elmk_ctx* ctx, double dewmx, int oldfflag, double dayl, double max_dayl)
{
  if (!ctx) return ELMK_E_INVALID;
  ctx->h.dewmx = dewmx;
  ctx->h.oldfflag = oldfflag;
  ctx->h.dayl = dayl;
  ctx->h.max_dayl = max_dayl;
  ctx->dirty = true;
  return ELMK_OK;
}

int elmk_set_pft(elmk_ctx* ctx, const double* psn, const double* alb, const double* z0mr, const double* displar)
{
  if (!ctx || !psn || !alb || !z0mr || !displar) return invalid(ctx, "elmk_set_pft: null table");
  memcpy(ctx->h.pft_psn, psn, sizeof ctx->h.pft_psn);
  memcpy(ctx->h.pft_alb, alb, sizeof ctx->h.pft_alb);
  memcpy(ctx->h.z0mr, z0mr, sizeof ctx->h.z0mr);
  memcpy(ctx->h.displar, displar, sizeof ctx->h.displar);
  ctx->dirty = true;
  return ELMK_OK;
}

int elmk_set_init_params(elmk_ctx* ctx, double organic_max, const double* roota_par, const double* rootb_par)
{
  if (!ctx || !roota_par || !rootb_par) return invalid(ctx, "elmk_set_init_params: null table");
  if (!(organic_max > 0.0)) return invalid(ctx, "elmk_set_init_params: organic_max must be positive");
  ctx->h.organic_max = organic_max;
  memcpy(ctx->h.roota_par, roota_par, sizeof ctx->h.roota_par);
  memcpy(ctx->h.rootb_par, rootb_par, sizeof ctx->h.rootb_par);
  ctx->dirty = true;
  ctx->have_init_params = true;
  return ELMK_OK;
}

int elmk_set_soilcolor(elmk_ctx* ctx, const double* albsat, const double* albdry)
{
  if (!ctx || !albsat || !albdry) return invalid(ctx, "elmk_set_soilcolor: null table");
  memcpy(ctx->h.albsat, albsat, sizeof ctx->h.albsat);
  memcpy(ctx->h.albdry, albdry, sizeof ctx->h.albdry);
  ctx->dirty = true;
  return ELMK_OK;
}

int elmk_set_snicar(elmk_ctx* ctx, const elmk_snicar_tables* t)
{
  if (int rc = enter(ctx)) return rc;
  if (!t) return invalid(ctx, "elmk_set_snicar: null");
  std::vector<double> buf(SN_TOTAL, 0.0);
  const double* aer[6][3] = {
      {t->ss_alb_oc1, t->asm_prm_oc1, t->ext_cff_mss_oc1},    {t->ss_alb_oc2, t->asm_prm_oc2, t->ext_cff_mss_oc2},
      {t->ss_alb_dst1, t->asm_prm_dst1, t->ext_cff_mss_dst1}, {t->ss_alb_dst2, t->asm_prm_dst2, t->ext_cff_mss_dst2},
      {t->ss_alb_dst3, t->asm_prm_dst3, t->ext_cff_mss_dst3}, {t->ss_alb_dst4, t->asm_prm_dst4, t->ext_cff_mss_dst4}};
  for (int s = 0; s < 6; s++)
    for (int p = 0; p < 3; p++) {
      if (!aer[s][p]) return invalid(ctx, "elmk_set_snicar: null aerosol table");
      memcpy(&buf[SN_OC1 + s * SN_AER_STRIDE + p * 5], aer[s][p], 5 * sizeof(double));
    }
  const double* snw[2][3] = {{t->ss_alb_snw_drc, t->asm_prm_snw_drc, t->ext_cff_mss_snw_drc},
                             {t->ss_alb_snw_dfs, t->asm_prm_snw_dfs, t->ext_cff_mss_snw_dfs}};
  for (int k = 0; k < 2; k++)
    for (int p = 0; p < 3; p++) {
      if (!snw[k][p]) return invalid(ctx, "elmk_set_snicar: null Mie table");
      memcpy(&buf[(k == 0 ? SN_SNW_DRC : SN_SNW_DFS) + p * 5 * ELMK_MIE_N], snw[k][p], 5 * ELMK_MIE_N * sizeof(double));
    }
  const double* bc[2][3] = {{t->ss_alb_bc1, t->asm_prm_bc1, t->ext_cff_mss_bc1},
                            {t->ss_alb_bc2, t->asm_prm_bc2, t->ext_cff_mss_bc2}};
  for (int k = 0; k < 2; k++)
    for (int p = 0; p < 3; p++) {
      if (!bc[k][p]) return invalid(ctx, "elmk_set_snicar: null BC table");
      memcpy(&buf[(k == 0 ? SN_BC1 : SN_BC2) + p * 50], bc[k][p], 50 * sizeof(double));
    }
  if (!t->bcenh) return invalid(ctx, "elmk_set_snicar: null bcenh");
  memcpy(&buf[SN_BCENH], t->bcenh, 400 * sizeof(double));
  HIPCHK(hipMemcpyAsync(ctx->snicar, buf.data(), SN_TOTAL * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return ELMK_OK;
}

int elmk_set_snow_age_tables(elmk_ctx* ctx, const double* tau, const double* kappa, const double* drdt0)
{
  if (int rc = enter(ctx)) return rc;
  if (!tau || !kappa || !drdt0) return invalid(ctx, "elmk_set_snow_age_tables: null table");
  const double* src[3] = {tau, kappa, drdt0};
  for (int k = 0; k < 3; k++)
    HIPCHK(hipMemcpyAsync(ctx->snowage + (size_t)k * ELMK_SNOWAGE_N, src[k], ELMK_SNOWAGE_N * sizeof(double), hipMemcpyHostToDevice,
                          ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return ELMK_OK;
}

// ---------------------------------------------------------------------------------------------------
// physics wrappers: one launch each, same order/arguments as driver/kokkos
// ---------------------------------------------------------------------------------------------------
namespace {
int heal_lists(elmk_ctx* ctx)
{
  if (!ctx->lists_stale) return ELMK_OK;
  ctx->lists_stale = false;
  HIPCHK(hipMemsetAsync(ctx->counters_raw, 0, ctx->counters_bytes, ctx->stream));
  return ELMK_OK;
}
}  // namespace
#define PHYSICS_PROLOGUE()                 \
  if (int rc = enter(ctx)) return rc;      \
  if (int rc = heal_lists(ctx)) return rc; \
  if (int rc = push_params(ctx)) return rc

int elmk_frac_wet(elmk_ctx* ctx)
{
  PHYSICS_PROLOGUE();
  launch_frac_wet(ctx->d, ctx->ncols, ctx->stream);
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}
int elmk_albedo_snicar(elmk_ctx* ctx)
{
  PHYSICS_PROLOGUE();
  launch_albedo_snicar(ctx->d, ctx->ncols, ctx->stream, &ctx->side);
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}
int elmk_canopy_hydrology(elmk_ctx* ctx, double dt)
{
  PHYSICS_PROLOGUE();
  launch_canopy_hydrology(ctx->d, ctx->ncols, dt, ctx->stream);
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}
int elmk_surface_radiation(elmk_ctx* ctx)
{
  PHYSICS_PROLOGUE();
  launch_surface_radiation(ctx->d, ctx->ncols, ctx->stream);
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}
int elmk_canopy_temperature(elmk_ctx* ctx)
{
  PHYSICS_PROLOGUE();
  launch_canopy_temperature(ctx->d, ctx->ncols, ctx->stream);
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}
int elmk_bareground_fluxes(elmk_ctx* ctx)
{
  PHYSICS_PROLOGUE();
  launch_bareground_fluxes(ctx->d, ctx->ncols, ctx->stream);
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}
int elmk_canopy_fluxes(elmk_ctx* ctx, double dt)
{
  PHYSICS_PROLOGUE();
  launch_canopy_fluxes(ctx->d, ctx->ncols, dt, ctx->stream, 0, &ctx->side);
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}

// L2-level entries: the forcing-derived scalars handed in, as the reference's unit tests call the physics
// (test/test_CanFlux.cc:285-340, test/test_BGFlux.cc:200-260) instead of the wrapper's derive_forc_* (atm_physics_impl.hh:246-272)
namespace {
int stage_given(elmk_ctx* ctx, const double* rho, const double* po2, const double* pco2, int* mask)
{
  const double* src[3] = {rho, po2, pco2};
  *mask = 0;
  for (int k = 0; k < 3; k++) {
    if (!src[k] || ctx->ncols == 0) continue;
    HIPCHK(hipMemcpyAsync(ELMK_GENERIC(ctx->h.cf_given) + (size_t)k * ctx->ld, src[k], (size_t)ctx->ncols * 8, hipMemcpyHostToDevice,
                          ctx->stream));
    *mask |= 1 << k;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));  // the sources are caller-owned pageable host arrays
  return ELMK_OK;
}
}  // namespace

int elmk_canopy_fluxes_given(elmk_ctx* ctx, double dt, const double* forc_rho, const double* forc_po2, const double* forc_pco2)
{
  PHYSICS_PROLOGUE();
  int mask = 0;
  if (int rc = stage_given(ctx, forc_rho, forc_po2, forc_pco2, &mask)) return rc;
  launch_canopy_fluxes(ctx->d, ctx->ncols, dt, ctx->stream, mask, &ctx->side);
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}

int elmk_bareground_fluxes_given(elmk_ctx* ctx, const double* forc_rho)
{
  PHYSICS_PROLOGUE();
  int mask = 0;
  if (int rc = stage_given(ctx, forc_rho, nullptr, nullptr, &mask)) return rc;
  launch_bareground_fluxes(ctx->d, ctx->ncols, ctx->stream, mask);
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}

int elmk_soil_temperature(elmk_ctx* ctx, double dt)
{
  PHYSICS_PROLOGUE();
  launch_soil_temperature(ctx->d, ctx->ncols, dt, ctx->stream);
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}

int elmk_snow_hydrology(elmk_ctx* ctx, double dt)
{
  PHYSICS_PROLOGUE();
  launch_snow_hydrology(ctx->d, ctx->ncols, dt, ctx->stream);
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}

int elmk_init_timestep(elmk_ctx* ctx)
{
  PHYSICS_PROLOGUE();
  launch_init_timestep(ctx->d, ctx->ncols, ctx->stream);
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}

int elmk_get_forcing(elmk_ctx* ctx, const double* wt1, const double* wt2, int qbot_is_rh)
{
  PHYSICS_PROLOGUE();
  if (!wt1 || !wt2) return invalid(ctx, "elmk_get_forcing: null weights");
  launch_get_forcing(ctx->d, ctx->ncols, wt1, wt2, qbot_is_rh != 0, ctx->stream);
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}

int elmk_phenology(elmk_ctx* ctx, double wt1, double wt2)
{
  PHYSICS_PROLOGUE();
  launch_phenology(ctx->d, ctx->ncols, wt1, wt2, ctx->stream);
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}

int elmk_initialize_state(elmk_ctx* ctx)
{
  PHYSICS_PROLOGUE();
  if (!ctx->have_init_params) return invalid(ctx, "elmk_initialize_state: elmk_set_init_params has not been called");
  launch_initialize_state(ctx->d, ctx->ncols, ctx->stream);
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}

int elmk_surface_fluxes(elmk_ctx* ctx, double dt)
{
  PHYSICS_PROLOGUE();
  launch_surface_fluxes(ctx->d, ctx->ncols, dt, ctx->stream);
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}

int elmk_evaluate_conservation(elmk_ctx* ctx, double dt, double* min_max_sum, double* per_column)
{
  PHYSICS_PROLOGUE();
  if (!min_max_sum) return invalid(ctx, "elmk_evaluate_conservation: min_max_sum is NULL");
  double* diag = ELMK_GENERIC(ctx->h.cons_diag);
  double* part = diag + (size_t)8 * ctx->ld;
  double* out = part + (size_t)8 * ELMK_CONS_NPART * 3;
  launch_conservation(ctx->d, ctx->ncols, ctx->ld, dt, diag, part, out, ctx->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(min_max_sum, out, 8 * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (per_column) {  // [8][ncols], diagnostic-major
    for (int k = 0; k < 8; k++)
      HIPCHK(hipMemcpyAsync(per_column + (size_t)k * ctx->ncols, diag + (size_t)k * ctx->ld, (size_t)ctx->ncols * 8,
                            hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return ELMK_OK;
}

// ELMInterface::advance order (elm_kokkos_interface.cc:289-307).  ONE stage table per launch structure drives the plain
// path, the graph capture and the profiled path, so the three cannot drift apart.
namespace {
// roctx ranges named after the labels the reference gives its parallel_for launches (driver/kokkos/*_kokkos.cc:
// "kokkos_canhydro_fracwet_kernel", "kokkos_albedo_and_snicar", ...), so that a marker trace of this library reads like one of
// the reference (tracing).  Off unless ELMK_ROCTX=1 is set when the first context is created; the marker
// library is looked up at run time (no link dependency), and a missing one just leaves the ranges off.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx()
  {
    const char* e = getenv("ELMK_ROCTX");
    if (!e || e[0] != '1') return;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (push && pop) return;
        push = nullptr;
        pop = nullptr;
      }
    }
  }
};
const Roctx* roctx()
{
  static const Roctx r;
  return &r;
}
struct RoctxRange {
  explicit RoctxRange(const char* label)
  {
    if (roctx()->push) roctx()->push(label);
  }
  ~RoctxRange()
  {
    if (roctx()->pop) roctx()->pop();
  }
};
const char* const TS7_LABELS[7] = {"kokkos_canhydro_fracwet_kernel", "kokkos_albedo_and_snicar", "kokkos_canopy_hydrology", "kokkos_surface_radiation",
                                   "kokkos_canopy_temperature",     "kokkos_bareground_fluxes", "kokkos_canopy_fluxes"};
constexpr int TS7_NSTAGE = 7;
void launch_stage7(elmk_ctx* ctx, int k, double dt)
{
  const RoctxRange range(TS7_LABELS[k < 7 ? k : 6]);
  switch (k) {
    case 0: launch_frac_wet(ctx->d, ctx->ncols, ctx->stream); break;
    case 1: launch_albedo_snicar(ctx->d, ctx->ncols, ctx->stream, &ctx->side); break;
    case 2: launch_canopy_hydrology(ctx->d, ctx->ncols, dt, ctx->stream); break;
    case 3: launch_surface_radiation(ctx->d, ctx->ncols, ctx->stream); break;
    case 4: launch_canopy_temperature(ctx->d, ctx->ncols, ctx->stream); break;
    case 5: launch_bareground_fluxes(ctx->d, ctx->ncols, ctx->stream); break;
    default: launch_canopy_fluxes(ctx->d, ctx->ncols, dt, ctx->stream, 0, &ctx->side); break;
  }
}
// elmk_timestep7_fused: the same seven wrappers as ELMK_FUSED_NSTAGE launch groups (k_canopy_fluxes.hip)
void launch_stage_fused(elmk_ctx* ctx, int k, double dt) { launch_fused_stage(ctx->d, ctx->ncols, dt, ctx->stream, &ctx->side, k); }
typedef void (*stage_fn)(elmk_ctx*, int, double);

// all stages in order; marks (may be null): nstage + 1 events recorded around the stages on the context's stream
int enqueue_stages(elmk_ctx* ctx, stage_fn fn, int nstage, double dt, hipEvent_t* marks)
{
  for (int k = 0; k < nstage; k++) {
    if (marks) HIPCHK(hipEventRecord(marks[k], ctx->stream));
    fn(ctx, k, dt);
  }
  if (marks) HIPCHK(hipEventRecord(marks[nstage], ctx->stream));
  HIPCHK(hipGetLastError());
  return ELMK_OK;
}

// the stages captured once as a HIP graph (kernel nodes + the side-stream fork / join inside albedo_snicar) and replayed
int run_graph(elmk_ctx* ctx, GraphSlot& g, stage_fn fn, int nstage, double dt)
{
  if (!g.exec || g.dt != dt || g.stream != ctx->stream) {
    if (g.exec) {
      // dt or the stream changed: the old executable may still be running its last launch.  (Best effort: a caller that
      // destroyed the old stream has synchronised it itself, and the error of waiting on it is not this call's.)
      if (g.stream && hipStreamSynchronize(g.stream) != hipSuccess) (void)hipGetLastError();
      (void)hipGraphExecDestroy(g.exec);
      g.exec = nullptr;
    }
    hipGraph_t graph = nullptr;
    HIPCHK(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < nstage; k++) fn(ctx, k, dt);
    // a launch that failed during capture leaves its error in the runtime and may have invalidated the capture: read it,
    // and ALWAYS end the capture so that neither the stream nor the forked side streams stay in capture mode
    const hipError_t launch_err = hipGetLastError();
    const hipError_t end_err = hipStreamEndCapture(ctx->stream, &graph);
    if (launch_err != hipSuccess || end_err != hipSuccess) {
      if (graph) (void)hipGraphDestroy(graph);
      (void)hipGetLastError();
      hip_fail(ctx, launch_err != hipSuccess ? launch_err : end_err,
               launch_err != hipSuccess ? "kernel launch during graph capture" : "hipStreamEndCapture");
      return ELMK_E_HIP;
    }
    const hipError_t e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (hip_fail(ctx, e, "hipGraphInstantiate")) {
      g.exec = nullptr;
      return ELMK_E_HIP;
    }
    g.dt = dt;
    g.stream = ctx->stream;
  }
  HIPCHK(hipGraphLaunch(g.exec, ctx->stream));
  return ELMK_OK;
}

// nsteps profiled steps: HIP events between the stages on the context's stream; the snapshot (if any) is restored before
// every step outside the event brackets, so each profiled step does the same work as the caller's timed loop
int profile_stages(elmk_ctx* ctx, stage_fn fn, int nstage, double dt, int nsteps, float* ms_per_stage, float* ms_total,
                   float* ms_each_step = nullptr)
{
  EventList ev;
  HIPCHK(ev.create((size_t)nsteps * (nstage + 1)));
  for (int s = 0; s < nsteps; s++) {
    if (!ctx->snap_fields.empty())
      if (int rc = elmk_restore_fields(ctx)) return rc;
    if (int rc = enqueue_stages(ctx, fn, nstage, dt, &ev[(size_t)s * (nstage + 1)])) return rc;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::vector<double> acc((size_t)nstage, 0.0);
  double tot = 0.0;
  for (int s = 0; s < nsteps; s++) {
    hipEvent_t* e = &ev[(size_t)s * (nstage + 1)];
    for (int k = 0; k < nstage; k++) {
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, e[k], e[k + 1]));
      acc[k] += ms;
    }
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e[0], e[nstage]));
    tot += ms;
    if (ms_each_step) ms_each_step[s] = ms;
  }
  if (ms_per_stage)
    for (int k = 0; k < nstage; k++) ms_per_stage[k] = (float)(acc[k] / nsteps);
  if (ms_total) *ms_total = (float)(tot / nsteps);
  return ELMK_OK;
}
}  // namespace

int elmk_timestep7(elmk_ctx* ctx, double dt)
{
  PHYSICS_PROLOGUE();
  // ~22 dependent launches cost ~0.5 ms of launch latency however few columns there are; replaying them as one graph
  // removes the host side of that.  Kernel arguments are the device parameter block (fixed address) and dt.
  if (ctx->use_graph) return run_graph(ctx, ctx->graph[0], launch_stage7, TS7_NSTAGE, dt);
  return enqueue_stages(ctx, launch_stage7, TS7_NSTAGE, dt, nullptr);
}

int elmk_timestep7_fused(elmk_ctx* ctx, double dt)
{
  PHYSICS_PROLOGUE();
  if (ctx->use_graph) return run_graph(ctx, ctx->graph[1], launch_stage_fused, ELMK_FUSED_NSTAGE, dt);
  return enqueue_stages(ctx, launch_stage_fused, ELMK_FUSED_NSTAGE, dt, nullptr);
}

namespace {
// elmk_advance_physics: the fused seven, then the rest of ELMInterface::advance's per-column calls in its order
constexpr int ADV_NSTAGE = ELMK_FUSED_NSTAGE + 3;
void launch_stage_advance(elmk_ctx* ctx, int k, double dt)
{
  if (k < ELMK_FUSED_NSTAGE) {
    launch_stage_fused(ctx, k, dt);
    return;
  }
  switch (k - ELMK_FUSED_NSTAGE) {
    case 0: launch_soil_temperature(ctx->d, ctx->ncols, dt, ctx->stream); break;
    case 1: launch_snow_hydrology(ctx->d, ctx->ncols, dt, ctx->stream); break;
    default: launch_surface_fluxes(ctx->d, ctx->ncols, dt, ctx->stream); break;
  }
}
}  // namespace

int elmk_advance_physics(elmk_ctx* ctx, double dt)
{
  PHYSICS_PROLOGUE();
  if (ctx->use_graph) return run_graph(ctx, ctx->graph[2], launch_stage_advance, ADV_NSTAGE, dt);
  return enqueue_stages(ctx, launch_stage_advance, ADV_NSTAGE, dt, nullptr);
}

// ---------------------------------------------------------------------------------------------------
// diagnostics
// ---------------------------------------------------------------------------------------------------
int elmk_error_summary(elmk_ctx* ctx, uint32_t* or_of_flags, int64_t* first_bad_col)
{
  if (int rc = enter(ctx)) return rc;
  const long long none = 0x7fffffffffffffffll;
  HIPCHK(hipMemsetAsync(ctx->red_or, 0, 8, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->red_first, &none, 8, hipMemcpyHostToDevice, ctx->stream));
  launch_flag_reduce((const uint32_t*)ctx->fptr[ELMK_FIELD_err_flags], ctx->ncols, ctx->red_or, ctx->red_first,
                     ctx->stream);
  HIPCHK(hipGetLastError());
  uint32_t o = 0;
  long long first = none;
  HIPCHK(hipMemcpyAsync(&o, ctx->red_or, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipMemcpyAsync(&first, ctx->red_first, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (or_of_flags) *or_of_flags = o;
  if (first_bad_col) *first_bad_col = (first == none) ? -1 : (int64_t)first;
  return ELMK_OK;
}

int elmk_clear_errors(elmk_ctx* ctx) { return elmk_fill(ctx, ELMK_FIELD_err_flags, 0.0); }

int elmk_profile_timestep7(elmk_ctx* ctx, double dt, int nsteps, float* ms_per_kernel, float* ms_total)
{
  PHYSICS_PROLOGUE();
  if (nsteps <= 0) return invalid(ctx, "elmk_profile_timestep7: nsteps <= 0");
  return profile_stages(ctx, launch_stage7, TS7_NSTAGE, dt, nsteps, ms_per_kernel, ms_total);
}

int elmk_profile_timestep7_fused(elmk_ctx* ctx, double dt, int nsteps, float* ms_per_stage, float* ms_total)
{
  PHYSICS_PROLOGUE();
  if (nsteps <= 0) return invalid(ctx, "elmk_profile_timestep7_fused: nsteps <= 0");
  return profile_stages(ctx, launch_stage_fused, ELMK_FUSED_NSTAGE, dt, nsteps, ms_per_stage, ms_total);
}

int elmk_profile_steps(elmk_ctx* ctx, int fused, double dt, int nsteps, float* ms_each_step)
{
  PHYSICS_PROLOGUE();
  if (nsteps <= 0 || !ms_each_step) return invalid(ctx, "elmk_profile_steps: bad arguments");
  return fused ? profile_stages(ctx, launch_stage_fused, ELMK_FUSED_NSTAGE, dt, nsteps, nullptr, nullptr, ms_each_step)
               : profile_stages(ctx, launch_stage7, TS7_NSTAGE, dt, nsteps, nullptr, nullptr, ms_each_step);
}

namespace {
void launch_one_wrapper(elmk_ctx* ctx, int wrapper, double dt)
{
  const RoctxRange range(wrapper == ELMK_WRAPPER_SOIL_TEMPERATURE ? "kokkos_soil_temperature"
                         : wrapper == ELMK_WRAPPER_SNOW_HYDROLOGY ? "kokkos_snow_hydrology"
                         : wrapper == ELMK_WRAPPER_SURFACE_FLUXES ? "kokkos_surface_fluxes" : "elmk_wrapper");
  if (wrapper <= ELMK_WRAPPER_CANOPY_FLUXES)
    launch_stage7(ctx, wrapper, dt);
  else if (wrapper == ELMK_WRAPPER_SOIL_TEMPERATURE)
    launch_soil_temperature(ctx->d, ctx->ncols, dt, ctx->stream);
  else if (wrapper == ELMK_WRAPPER_SNOW_HYDROLOGY)
    launch_snow_hydrology(ctx->d, ctx->ncols, dt, ctx->stream);
  else if (wrapper == ELMK_WRAPPER_ADVANCE_PHYSICS)
    for (int k = 0; k < ADV_NSTAGE; k++) launch_stage_advance(ctx, k, dt);
  else
    launch_surface_fluxes(ctx->d, ctx->ncols, dt, ctx->stream);
}
}  // namespace

int elmk_profile_wrapper(elmk_ctx* ctx, int wrapper, double dt, int nsteps, float* ms_mean)
{
  PHYSICS_PROLOGUE();
  if (nsteps <= 0 || !ms_mean) return invalid(ctx, "elmk_profile_wrapper: bad arguments");
  if (wrapper < 0 || wrapper > ELMK_WRAPPER_ADVANCE_PHYSICS) return invalid(ctx, "elmk_profile_wrapper: unknown wrapper");
  EventList ev;
  HIPCHK(ev.create((size_t)nsteps * 2));
  for (int s = 0; s < nsteps; s++) {
    if (!ctx->snap_fields.empty())
      if (int rc = elmk_restore_fields(ctx)) return rc;
    HIPCHK(hipEventRecord(ev[(size_t)s * 2], ctx->stream));
    launch_one_wrapper(ctx, wrapper, dt);
    HIPCHK(hipEventRecord(ev[(size_t)s * 2 + 1], ctx->stream));
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  double acc = 0.0;
  for (int s = 0; s < nsteps; s++) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ev[(size_t)s * 2], ev[(size_t)s * 2 + 1]));
    acc += ms;
  }
  *ms_mean = (float)(acc / nsteps);
  return ELMK_OK;
}

int elmk_read_scratch(elmk_ctx* ctx, int kind, void* host, int64_t offset, int64_t count)
{
  if (int rc = enter(ctx)) return rc;
  if (!host || offset < 0 || count < 0) return invalid(ctx, "elmk_read_scratch: bad arguments");
  const void* src = nullptr;
  size_t esz = 0;
  int64_t limit = 0;
  if (kind == ELMK_SCRATCH_LIST_COUNTS) {  // (count, head) of every work list: the counters sit one per 128-byte line
    if (offset != 0 || count != 2 * NLISTS) return invalid(ctx, "elmk_read_scratch: the list counters are read whole (2 x the number of lists)");
    std::vector<uint32_t> raw((size_t)2 * NLISTS * CPAD);
    HIPCHK(hipMemcpyAsync(raw.data(), ELMK_GENERIC(ctx->h.counters), raw.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < NLISTS; k++) {
      ((uint32_t*)host)[2 * k] = raw[(size_t)k * CPAD];
      ((uint32_t*)host)[2 * k + 1] = raw[(size_t)(NLISTS + k) * CPAD];
    }
    return ELMK_OK;
  }
  if (kind == ELMK_SCRATCH_CF_TRIPS || kind == ELMK_SCRATCH_CF_HINTS) {
    src = ELMK_GENERIC(ctx->h.cf_niter);
    esz = 4;
    limit = ctx->ncols;
  } else if (kind == ELMK_SCRATCH_WORK) {
    src = ELMK_GENERIC(ctx->h.wk);
    esz = 8;
    limit = (int64_t)WK_N * ctx->ld;
  } else {
    return invalid(ctx, "elmk_read_scratch: unknown kind");
  }
  if (offset + count > limit) return invalid(ctx, "elmk_read_scratch: range exceeds the scratch array");
  HIPCHK(hipMemcpyAsync(host, (const char*)src + (size_t)offset * esz, (size_t)count * esz, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (kind == ELMK_SCRATCH_CF_TRIPS)  // the high half of each word is the scheduler's hint
    for (int64_t i = 0; i < count; i++) ((int32_t*)host)[i] &= 0xFFFF;
  if (kind == ELMK_SCRATCH_CF_HINTS)
    for (int64_t i = 0; i < count; i++) ((int32_t*)host)[i] >>= 16;
  return ELMK_OK;
}

int elmk_state_real_bytes(void) { return kStateF32 ? 4 : 8; }

int elmk_copy_bandwidth(elmk_ctx* ctx, int64_t bytes, int iters, double* gbytes_per_s)
{
  return elmk_copy_bandwidth_shape(ctx, bytes, iters, 0, gbytes_per_s);
}

int elmk_copy_bandwidth_shape(elmk_ctx* ctx, int64_t bytes, int iters, int shape, double* gbytes_per_s)
{
  if (int rc = enter(ctx)) return rc;
  if (bytes < 512 || iters <= 0 || shape < 0 || shape > 4 || !gbytes_per_s) return invalid(ctx, "elmk_copy_bandwidth: bad arguments");
  const int64_t n = (bytes / 512) * 64;  // whole 512-byte runs: every shape copies the same bytes
  DevBuf a, b;
  if (hip_fail(ctx, a.alloc((size_t)n * 8), "hipMalloc") || hip_fail(ctx, b.alloc((size_t)n * 8), "hipMalloc")) return ELMK_E_NOMEM;
  EventList ev;
  HIPCHK(ev.create(2));
  HIPCHK(hipMemsetAsync(a.p, 0, (size_t)n * 8, ctx->stream));
  launch_copy((const double*)a.p, (double*)b.p, n, ctx->stream, shape);  // warm-up
  HIPCHK(hipEventRecord(ev[0], ctx->stream));
  for (int i = 0; i < iters; i++) launch_copy((const double*)a.p, (double*)b.p, n, ctx->stream, shape);
  HIPCHK(hipEventRecord(ev[1], ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, ev[0], ev[1]));
  *gbytes_per_s = 2.0 * (double)n * 8.0 * iters / ((double)ms * 1e-3) / 1e9;
  return ELMK_OK;
}

int elmk_math_eval(elmk_ctx* ctx, int fn, const double* x, const double* y, double* out, int64_t n)
{
  if (int rc = enter(ctx)) return rc;
  if (fn < ELMK_MATH_EXP || fn > ELMK_MATH_POW || !x || !out || n < 0 || (fn >= ELMK_MATH_DIV && !y))
    return invalid(ctx, "elmk_math_eval: bad arguments");
  if (n == 0) return ELMK_OK;
  double* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, (size_t)n * 8 * 3));
  int rc = ELMK_OK;
  if (hip_fail(ctx, hipMemcpyAsync(d, x, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync")) rc = ELMK_E_HIP;
  if (!rc && fn >= ELMK_MATH_DIV &&
      hip_fail(ctx, hipMemcpyAsync(d + n, y, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync"))
    rc = ELMK_E_HIP;
  if (!rc) {
    launch_math_eval(fn, d, d + n, d + 2 * n, n, ctx->stream);
    if (hip_fail(ctx, hipMemcpyAsync(out, d + 2 * n, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream), "hipMemcpyAsync") ||
        hip_fail(ctx, hipStreamSynchronize(ctx->stream), "hipStreamSynchronize"))
      rc = ELMK_E_HIP;
  }
  (void)hipFree(d);
  return rc;
}

}  // extern "C"
