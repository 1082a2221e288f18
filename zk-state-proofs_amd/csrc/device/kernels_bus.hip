// LogUp bus of the keccak chip, prover side ("plus
// lookup-argument constraints"; SP1 wires its precompile chips with the same argument,
// sp1-stark / sp1-core-machine 3.4.0, reference Cargo.lock:7485, :7130):
//   * the public I/O list as a limb matrix (what the transcript absorbs and the verifier
//     sums over),
//   * the running-sum trace phi: phi_0 = 0, phi_{i+1} = phi_i + export_i / f_i with
//     f = gamma + sum_j beta^j t_j over the row's 200-limb tuple, and its total S.
#include "air_keccak.hpp"
#include "kernels.h"

namespace zksp {

constexpr int kBusThreads = 256;

__device__ __forceinline__ uint64_t bus_rol64(uint64_t v, int n) { return n ? (v << n) | (v >> (64 - n)) : v; }

// one lane per permutation: limbs of the input state and of keccak-f(input)
__global__ __launch_bounds__(64) void keccak_io_kernel(const uint64_t* __restrict__ states, int max_perms,
                                                      const uint32_t* __restrict__ n_perms,
                                                      uint32_t* __restrict__ io, size_t io_stride) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= (int)n_perms[b]) return;
  const ka::Tables& T = ka::tables();
  uint64_t a[25];
  const uint64_t* s = states + ((size_t)b * max_perms + p) * 25;
  uint32_t* dst = io + (size_t)b * io_stride + (size_t)p * ka::kBusTuple;
#pragma unroll
  for (int j = 0; j < 25; ++j) {
    a[j] = s[j];
#pragma unroll
    for (int l = 0; l < 4; ++l) dst[4 * j + l] = Fp::from_canonical((uint32_t)((a[j] >> (16 * l)) & 0xffff)).v;
  }
  for (int r = 0; r < 24; ++r) {
    uint64_t c[5], d[5], bb[25];
#pragma unroll
    for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
    for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ bus_rol64(c[(x + 1) % 5], 1);
#pragma unroll
    for (int j = 0; j < 25; ++j) a[j] ^= d[j % 5];
#pragma unroll
    for (int x = 0; x < 5; ++x)
#pragma unroll
      for (int y = 0; y < 5; ++y) bb[y + 5 * ((2 * x + 3 * y) % 5)] = bus_rol64(a[x + 5 * y], T.rot[x][y]);
#pragma unroll
    for (int y = 0; y < 5; ++y)
#pragma unroll
      for (int x = 0; x < 5; ++x) a[x + 5 * y] = bb[x + 5 * y] ^ (~bb[(x + 1) % 5 + 5 * y] & bb[(x + 2) % 5 + 5 * y]);
    a[0] ^= T.rc[r];
  }
#pragma unroll
  for (int j = 0; j < 25; ++j)
#pragma unroll
    for (int l = 0; l < 4; ++l) dst[100 + 4 * j + l] = Fp::from_canonical((uint32_t)((a[j] >> (16 * l)) & 0xffff)).v;
}

void launch_keccak_io(hipStream_t stream, const uint64_t* states, int max_perms, const uint32_t* n_perms, uint32_t* io,
                      size_t io_stride, int batch) {
  (void)hipMemsetAsync(io, 0, (size_t)batch * io_stride * 4, stream);  // zero padding past 200 * n_perms
  hipLaunchKernelGGL(keccak_io_kernel, dim3((max_perms + 63) / 64, batch), dim3(64), 0, stream, states, max_perms,
                     n_perms, io, io_stride);
}

__device__ __forceinline__ Fp4 bus_load_fp4(const uint32_t* p) {
  uint4 v = *reinterpret_cast<const uint4*>(p);
  Fp4 r;
  r.c[0] = Fp::raw(v.x); r.c[1] = Fp::raw(v.y); r.c[2] = Fp::raw(v.z); r.c[3] = Fp::raw(v.w);
  return r;
}
__device__ __forceinline__ void bus_store_fp4(uint32_t* p, const Fp4& a) {
  *reinterpret_cast<uint4*>(p) = make_uint4(a.c[0].v, a.c[1].v, a.c[2].v, a.c[3].v);
}

// one lane per trace row: term = export / (gamma + sum_j beta^j t_j)
__global__ __launch_bounds__(kBusThreads) void bus_term_kernel(const uint32_t* __restrict__ trace,
                                                              const uint32_t* __restrict__ bus_ch,
                                                              const uint32_t* __restrict__ beta_pows,
                                                              uint32_t* __restrict__ terms, int logh) {
  const int h = 1 << logh;
  const int row = blockIdx.x * kBusThreads + threadIdx.x;
  if (row >= h) return;
  const int b = blockIdx.y;
  const uint32_t* t = trace + (size_t)b * ka::kWidth * h + row;
  const uint32_t* bp = beta_pows + (size_t)b * ka::kBusTuple * 4;
  Fp4 f = bus_load_fp4(bus_ch + (size_t)b * 8);  // gamma
  for (int j = 0; j < ka::kBusTuple; ++j) f += bus_load_fp4(bp + (size_t)j * 4) * Fp::raw(t[(size_t)ka::bus_tuple_col(j) * h]);
  const Fp m = Fp::raw(t[(size_t)ka::kExport * h]);
  bus_store_fp4(terms + ((size_t)b * h + row) * 4, f.inv() * m);
}

// one workgroup per proof: exclusive running sum of the terms -> phi columns, total -> cum_sum
__global__ __launch_bounds__(kBusThreads) void bus_scan_kernel(const uint32_t* __restrict__ terms,
                                                              uint32_t* __restrict__ phi,
                                                              uint32_t* __restrict__ cum_sum, int logh) {
  __shared__ Fp4 part[kBusThreads];
  const int h = 1 << logh;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int chunk = (h + kBusThreads - 1) / kBusThreads;
  const int r0 = tid * chunk, r1 = min(h, r0 + chunk);
  const uint32_t* tm = terms + (size_t)b * h * 4;
  Fp4 local = Fp4::zero();
  for (int r = r0; r < r1; ++r) local += bus_load_fp4(tm + (size_t)r * 4);
  part[tid] = local;
  __syncthreads();
  // inclusive Hillis-Steele scan over the 256 chunk sums
  for (int off = 1; off < kBusThreads; off <<= 1) {
    Fp4 v = part[tid];
    if (tid >= off) v += part[tid - off];
    __syncthreads();
    part[tid] = v;
    __syncthreads();
  }
  Fp4 acc = tid ? part[tid - 1] : Fp4::zero();
  uint32_t* ph = phi + (size_t)b * ka::kPermWidth * h;
  for (int r = r0; r < r1; ++r) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ph[(size_t)j * h + r] = acc.c[j].v;
    acc += bus_load_fp4(tm + (size_t)r * 4);
  }
  if (tid == kBusThreads - 1) bus_store_fp4(cum_sum + (size_t)b * 4, part[kBusThreads - 1]);
}

void launch_bus_perm_trace(hipStream_t stream, const uint32_t* trace, const uint32_t* bus_ch, const uint32_t* beta_pows,
                           uint32_t* terms, uint32_t* phi, uint32_t* cum_sum, int logh, int batch) {
  const int h = 1 << logh;
  hipLaunchKernelGGL(bus_term_kernel, dim3((h + kBusThreads - 1) / kBusThreads, batch), dim3(kBusThreads), 0, stream,
                     trace, bus_ch, beta_pows, terms, logh);
  hipLaunchKernelGGL(bus_scan_kernel, dim3(batch), dim3(kBusThreads), 0, stream, terms, phi, cum_sum, logh);
}

}  // namespace zksp
