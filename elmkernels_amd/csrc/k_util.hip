// k_util.hip - data-movement kernels around the physics: layout conversion, fill, synthetic tiling,
// error-flag reduction and a copy-bandwidth probe.
#include "elmk_dev.h"
#include "elmk_kernels.h"

namespace elmk {

// ---------------------------------------------------------------------------------------------------
// [column][level] (reference host layout, level fastest) <-> SoA [level][column].
// A 64-column x nlev tile goes through LDS so both the dense staging side and the SoA side are accessed
// with consecutive lanes on consecutive addresses.  Tile rows are padded by one element (no bank conflicts
// on the transposed access).
// ---------------------------------------------------------------------------------------------------
constexpr int TCOLS = 64;
constexpr int MAXLEV = 21;

template <typename T, bool TO_SOA>
__global__ __launch_bounds__(256) void k_transpose(const T* __restrict__ src, T* __restrict__ dst, int nlev, int64_t ld,
                                                   int64_t col0, int64_t n)
{
  __shared__ T tile[TCOLS * (MAXLEV + 1)];
  const int64_t cbase = (int64_t)blockIdx.x * TCOLS;  // first column of this tile, relative to col0
  const int ncol = (int)((n - cbase) < TCOLS ? (n - cbase) : TCOLS);
  const int total = ncol * nlev;
  const int pitch = nlev + 1;
  if (TO_SOA) {
    // dense [col][lev] chunk is contiguous: element e -> (col = e / nlev, lev = e % nlev)
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int col = e / nlev, lev = e - col * nlev;
      tile[col * pitch + lev] = src[cbase * nlev + e];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int lev = e / ncol, col = e - lev * ncol;
      dst[(int64_t)lev * ld + col0 + cbase + col] = tile[col * pitch + lev];
    }
  } else {
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int lev = e / ncol, col = e - lev * ncol;
      tile[col * pitch + lev] = src[(int64_t)lev * ld + col0 + cbase + col];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int col = e / nlev, lev = e - col * nlev;
      dst[cbase * nlev + e] = tile[col * pitch + lev];
    }
  }
}

template <bool TO_SOA>
static void transpose_dispatch(const void* src, void* dst, int elem, int nlev, int64_t ld, int64_t col0, int64_t n,
                               hipStream_t st)
{
  if (n <= 0) return;
  const dim3 grid((unsigned)((n + TCOLS - 1) / TCOLS)), block(256);
  switch (elem) {
    case 8:
      hipLaunchKernelGGL((k_transpose<double, TO_SOA>), grid, block, 0, st, (const double*)src, (double*)dst, nlev, ld,
                         col0, n);
      break;
    case 4:
      hipLaunchKernelGGL((k_transpose<int32_t, TO_SOA>), grid, block, 0, st, (const int32_t*)src, (int32_t*)dst, nlev,
                         ld, col0, n);
      break;
    default:
      hipLaunchKernelGGL((k_transpose<uint8_t, TO_SOA>), grid, block, 0, st, (const uint8_t*)src, (uint8_t*)dst, nlev,
                         ld, col0, n);
      break;
  }
}

void launch_cols_to_soa(const void* staging, void* field, int elem, int nlev, int64_t ld, int64_t col0, int64_t n,
                        hipStream_t st)
{
  transpose_dispatch<true>(staging, field, elem, nlev, ld, col0, n, st);
}
void launch_soa_to_cols(const void* field, void* staging, int elem, int nlev, int64_t ld, int64_t col0, int64_t n,
                        hipStream_t st)
{
  transpose_dispatch<false>(field, staging, elem, nlev, ld, col0, n, st);
}

// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_fill(T* __restrict__ f, int nlev, int64_t ld, int64_t ncols, T v)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  for (int l = 0; l < nlev; l++) f[(int64_t)l * ld + c] = v;
}

void launch_fill(void* field, int dtype, int nlev, int64_t ld, int64_t ncols, double value, hipStream_t st)
{
  if (ncols <= 0) return;
  const dim3 grid((unsigned)((ncols + 255) / 256)), block(256);
  if (dtype == ELMK_F64)
    hipLaunchKernelGGL(k_fill<double>, grid, block, 0, st, (double*)field, nlev, ld, ncols, value);
  else if (dtype == ELMK_F32_STORED)
    hipLaunchKernelGGL(k_fill<float>, grid, block, 0, st, (float*)field, nlev, ld, ncols, (float)value);
  else if (dtype == ELMK_I32)
    hipLaunchKernelGGL(k_fill<int32_t>, grid, block, 0, st, (int32_t*)field, nlev, ld, ncols, (int32_t)value);
  else if (dtype == ELMK_U32)
    hipLaunchKernelGGL(k_fill<uint32_t>, grid, block, 0, st, (uint32_t*)field, nlev, ld, ncols, (uint32_t)value);
  else
    hipLaunchKernelGGL(k_fill<uint8_t>, grid, block, 0, st, (uint8_t*)field, nlev, ld, ncols, (uint8_t)value);
}

// ---------------------------------------------------------------------------------------------------
// synthetic workload: column c >= nbase takes column c % nbase; perturbed fields get a counter-based
// uniform u in (-1, 1) from splitmix64(seed, field, level, column) - reproducible on any grid shape.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

template <typename T>
__global__ __launch_bounds__(256) void k_tile(T* __restrict__ f, int nlev, int64_t ld, int64_t ncols, int64_t nbase,
                                              uint64_t seed, int field_id, int mode, double amp)
{
  const int64_t c = nbase + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  const int64_t src = c % nbase;
  for (int l = 0; l < nlev; l++) {
    T v = f[(int64_t)l * ld + src];
    if (mode >= 0) {
      const uint64_t h = splitmix64(seed ^ splitmix64(((uint64_t)field_id << 40) ^ ((uint64_t)l << 32) ^ (uint64_t)c));
      const double u = ((double)(h >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0;
      const double x = (double)v;
      v = (T)(mode == 0 ? x * (1.0 + amp * u) : x + amp * u);
    }
    f[(int64_t)l * ld + c] = v;
  }
}

void launch_tile(void* field, int dtype, int nlev, int64_t ld, int64_t ncols, int64_t nbase, uint64_t seed,
                 int field_id, int mode, double amp, hipStream_t st)
{
  const int64_t n = ncols - nbase;
  if (n <= 0) return;
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (dtype == ELMK_F64)
    hipLaunchKernelGGL(k_tile<double>, grid, block, 0, st, (double*)field, nlev, ld, ncols, nbase, seed, field_id, mode,
                       amp);
  else if (dtype == ELMK_F32_STORED)
    hipLaunchKernelGGL(k_tile<float>, grid, block, 0, st, (float*)field, nlev, ld, ncols, nbase, seed, field_id, mode, amp);
  else if (dtype == ELMK_I32 || dtype == ELMK_U32)
    hipLaunchKernelGGL(k_tile<int32_t>, grid, block, 0, st, (int32_t*)field, nlev, ld, ncols, nbase, seed, field_id, -1,
                       0.0);
  else
    hipLaunchKernelGGL(k_tile<uint8_t>, grid, block, 0, st, (uint8_t*)field, nlev, ld, ncols, nbase, seed, field_id, -1,
                       0.0);
}

// ---------------------------------------------------------------------------------------------------
// OR of all flag words + first column carrying a fatal bit
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flag_reduce(const uint32_t* __restrict__ flags, int64_t n, uint32_t* or_out,
                                                     long long* first_bad)
{
  uint32_t acc = 0;
  long long first = 0x7fffffffffffffffll;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t f = flags[c];
    acc |= f;
    if ((f & ELMK_ERR_FATAL_MASK) && c < first) first = c;
  }
  // wave64 butterfly, then one atomic per wave
  for (int off = 32; off > 0; off >>= 1) {
    acc |= __shfl_xor(acc, off, 64);
    const long long o = __shfl_xor(first, off, 64);
    first = o < first ? o : first;
  }
  if ((threadIdx.x & 63) == 0) {
    if (acc) atomicOr(or_out, acc);
    if (first != 0x7fffffffffffffffll) atomicMin(first_bad, first);
  }
}

void launch_flag_reduce(const uint32_t* flags, int64_t n, uint32_t* or_out, long long* first_bad, hipStream_t st)
{
  if (n <= 0) return;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_flag_reduce, dim3((unsigned)blocks), dim3(256), 0, st, flags, n, or_out, first_bad);
}

// ---------------------------------------------------------------------------------------------------
// streaming copy with the same access shape as the physics kernels (8 bytes per lane): empirical HBM line
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_copy(const double* __restrict__ src, double* __restrict__ dst, int64_t n)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// several such copies as ONE launch (elmk_restore_fields: a step of the benchmark restores four 8 MB fields, and four
// launches of 15 us each - latency, not bytes - were 2.6 % of the step)
// W = words of 8 bytes per thread (2: one 16-byte access per lane, half the workgroups - every field row is a multiple of
// 256 bytes on a 256-byte boundary, so the odd case only exists for callers with other buffers)
template <int W>
__global__ __launch_bounds__(256) void k_copy_multi(const CopyJobs J)
{
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * W;
  if (i >= J.end[J.n - 1]) return;
  int j = 0;
  while (i >= J.end[j]) j++;  // (n <= COPY_JOBS_MAX: a handful of scalar compares)
  const int64_t k = i - (j ? J.end[j - 1] : 0);
  if (W == 2) {
    *reinterpret_cast<double2*>(J.dst[j] + k) = *reinterpret_cast<const double2*>(J.src[j] + k);
  } else {
    J.dst[j][k] = J.src[j][k];
  }
}

void launch_copy_multi(const CopyJobs& J, hipStream_t st)
{
  if (J.n <= 0 || J.end[J.n - 1] <= 0) return;
  bool wide = true;
  for (int j = 0; j < J.n; j++)
    wide = wide && (J.end[j] % 2 == 0) && ((uintptr_t)J.src[j] % 16 == 0) && ((uintptr_t)J.dst[j] % 16 == 0);
  const int64_t n = J.end[J.n - 1];
  if (wide) {
    hipLaunchKernelGGL(k_copy_multi<2>, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, J);
  } else {
    hipLaunchKernelGGL(k_copy_multi<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, J);
  }
}

// The same copy in other access shapes (elmk_copy_bandwidth_shape): what bounds a streaming kernel on this chip is how many
// bytes a CU keeps in flight, and that is (bytes per load) x (independent loads per wave) x (resident waves).
//   shape 1: 16 bytes per lane, one load per thread          (the float4 copy the 6.29 TB/s figure was measured with)
//   shape 2:  8 bytes per lane, four independent loads per thread, each a contiguous 512-byte run per wave
//   shape 3: 16 bytes per lane, four independent loads per thread
__global__ __launch_bounds__(256) void k_copy16(const double2* __restrict__ src, double2* __restrict__ dst, int64_t n2)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n2) dst[i] = src[i];
}
template <typename T>
__global__ __launch_bounds__(256) void k_copy_x4(const T* __restrict__ src, T* __restrict__ dst, int64_t n)
{
  const int64_t i = (int64_t)blockIdx.x * (blockDim.x * 4) + threadIdx.x;
  if (i + 3 * 256 < n) {
    const T a = src[i], b = src[i + 256], c = src[i + 512], d = src[i + 768];
    dst[i] = a;
    dst[i + 256] = b;
    dst[i + 512] = c;
    dst[i + 768] = d;
  } else {
    for (int k = 0; k < 4; k++)
      if (i + k * 256 < n) dst[i + k * 256] = src[i + k * 256];
  }
}

//   shape 4:  8 bytes per lane, 64 separate streams read and 64 written by every thread (the n doubles seen as 64 fields of
//            n / 64 columns, field-major like the state): the line a many-field streaming kernel can reach - HBM sustains fewer
//            bytes per second over a hundred open streams than over two (profiles/r03_stream_layout_ubench.txt)
__global__ __launch_bounds__(256) void k_copy_streams(const double* __restrict__ src, double* __restrict__ dst, int64_t ncol)
{
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncol) return;
  double v[64];
#pragma unroll
  for (int k = 0; k < 64; k++) v[k] = src[(int64_t)k * ncol + c];
#pragma unroll
  for (int k = 0; k < 64; k++) dst[(int64_t)k * ncol + c] = v[k];
}

void launch_copy(const double* src, double* dst, int64_t n, hipStream_t st, int shape)
{
  if (n <= 0) return;
  switch (shape) {
    case 4: hipLaunchKernelGGL(k_copy_streams, dim3((unsigned)((n / 64 + 255) / 256)), dim3(256), 0, st, src, dst, n / 64); break;
    case 1: hipLaunchKernelGGL(k_copy16, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, (const double2*)src, (double2*)dst, n / 2); break;
    case 2: hipLaunchKernelGGL(k_copy_x4<double>, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, st, src, dst, n); break;
    case 3: hipLaunchKernelGGL(k_copy_x4<double2>, dim3((unsigned)((n / 2 + 1023) / 1024)), dim3(256), 0, st, (const double2*)src, (double2*)dst, n / 2); break;
    default: hipLaunchKernelGGL(k_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n);
  }
}

// ---------------------------------------------------------------------------------------------------
// elmk_math.h on the device, element-wise (parity check of the math functions themselves)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_math_eval(int fn, const double* __restrict__ x, const double* __restrict__ y,
                                                   double* __restrict__ out, int64_t n)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double a = x[i];
  double r;
  switch (fn) {
    case ELMK_MATH_EXP: r = elmk_exp(a); break;
    case ELMK_MATH_LOG: r = elmk_log(a); break;
    case ELMK_MATH_LOG10: r = elmk_log10(a); break;
    case ELMK_MATH_ATAN: r = elmk_atan(a); break;
    case ELMK_MATH_SQRT: r = sqrt(a); break;
    case ELMK_MATH_TANH: r = elmk_tanh(a); break;
    case ELMK_MATH_COS: r = elmk_cos(a); break;
    case ELMK_MATH_ERF: r = elmk_erf(a); break;
    case ELMK_MATH_ACOS: r = elmk_acos(a); break;
    case ELMK_MATH_EXPM1: r = elmk_expm1(a); break;
    case ELMK_MATH_DIV: r = a / y[i]; break;
    default: r = elmk_pow(a, y[i]); break;
  }
  out[i] = r;
}

void launch_math_eval(int fn, const double* x, const double* y, double* out, int64_t n, hipStream_t st)
{
  if (n > 0) hipLaunchKernelGGL(k_math_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, fn, x, y, out, n);
}

}  // namespace elmk
