// kernels_rel.hip — plan-independent relational kernels (gfx950, wave64), compiled ahead of time:
//   exclusive scan, selection-vector compaction (ballot + popcount rank), row gather ("take") for every
//   Arrow layout on the path, and the hash-join build / probe kernels.
// Reference operators they replace are cited per kernel (paths relative to /root/reference/qurious/src).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "device/qhip_status.h"
#include "device/qhip_device.hpp"
#include "common.hpp"
#include "kernels.hpp"

namespace qhip {

#define QH_NULL_IDX 0xFFFFFFFFu

static inline unsigned grid_for(uint64_t n, unsigned per_block = QH_BLOCK, unsigned cap = 4096) {
  uint64_t g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (unsigned)g;
}

// ================================================================ exclusive scan (u32)
// Three-phase scan: per-chunk sums -> scan of the sums (recursive) -> per-chunk scan + chunk offset. A chunk is
// 2048 elements (256 threads x 8); inside a chunk each wave scans with shuffles, waves are combined through LDS.
#define SCAN_ITEMS 8
#define SCAN_CHUNK (QH_BLOCK * SCAN_ITEMS)

__device__ __forceinline__ u32 wave_incl_scan_u32(u32 v) {
  const int lane = qh_lane();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    u32 t = (u32)__shfl_up((int)v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

__global__ __launch_bounds__(QH_BLOCK) void k_scan_chunk_sums(const u32* in, u64 n, u32* sums) {
  __shared__ u32 wsum[4];
  const u64 base = (u64)blockIdx.x * SCAN_CHUNK;
  u32 s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    const u64 i = base + (u64)k * QH_BLOCK + threadIdx.x;
    s += i < n ? in[i] : 0u;
  }
  const u32 w = (u32)qh_wave_sum_u64(s);
  if (qh_lane() == 0) wsum[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(QH_BLOCK) void k_scan_chunk_apply(const u32* in, u32* out, u64 n, const u32* chunk_offsets) {
  __shared__ u32 wtot[4];
  const u64 base = (u64)blockIdx.x * SCAN_CHUNK + (u64)threadIdx.x * SCAN_ITEMS;
  u32 v[SCAN_ITEMS];
  u32 s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    const u64 i = base + k;
    v[k] = i < n ? in[i] : 0u;
    s += v[k];
  }
  const u32 incl = wave_incl_scan_u32(s);
  const int wave = threadIdx.x >> 6;
  if (qh_lane() == 63) wtot[wave] = incl;
  __syncthreads();
  u32 off = chunk_offsets ? chunk_offsets[blockIdx.x] : 0u;
  for (int w = 0; w < wave; ++w) off += wtot[w];
  u32 run = off + incl - s;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    const u64 i = base + k;
    if (i < n) out[i] = run;
    run += v[k];
  }
}

// The middle phase for up to 16384 chunk sums (n <= 32M elements): ONE workgroup scans them in place, 2048 at a time
// with a running carry, and leaves the grand total behind the last sum and in *total — one launch instead of the
// recursion's five (sums, copy, apply, copy, copy), which matters when the whole scan is ~10 us of device time.
__global__ __launch_bounds__(QH_BLOCK) void k_scan_sums_inplace(u32* sums, u32 nchunks, u32* total) {
  __shared__ u32 wtot[4];
  __shared__ u32 carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (u32 base = 0; base < nchunks; base += SCAN_CHUNK) {
    const u32 first = base + threadIdx.x * SCAN_ITEMS;
    u32 v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { v[k] = first + k < nchunks ? sums[first + k] : 0u; s += v[k]; }
    const u32 incl = wave_incl_scan_u32(s);
    const int wave = threadIdx.x >> 6;
    if (qh_lane() == 63) wtot[wave] = incl;
    __syncthreads();
    u32 off = carry_s;
    for (int w = 0; w < wave; ++w) off += wtot[w];
    u32 run = off + incl - s;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { if (first + k < nchunks) sums[first + k] = run; run += v[k]; }
    __syncthreads();
    if (threadIdx.x == QH_BLOCK - 1) carry_s = run;
    __syncthreads();
  }
  if (threadIdx.x == 0) { sums[nchunks] = carry_s; if (total) *total = carry_s; }
}

// up to 64 k values in ONE launch of one workgroup (thread t owns ceil(n / 1024) consecutive values): the three launches of
// the chunked scan below cost ~15 us of stream time whatever n is
__global__ __launch_bounds__(1024) void k_scan_small(const u32* in, u32* out, u32 n, u32* total) {
  __shared__ u32 wsum[16];
  typedef u32 v4u __attribute__((ext_vector_type(4)));
  constexpr int MAXV = 16;                                  // 16 x 4 values per thread: n <= 65536
  const u32 tid = threadIdx.x;
  const u32 nv = ((n + 1023u) / 1024u + 3u) / 4u;           // 16-byte vectors per thread (wave-uniform)
  const u32 first = tid * nv * 4u;
  v4u v[MAXV];
  // all loads first (independent, 16 bytes each), then the sums: a dependent chain of 4-byte loads costs ~1 us per step
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const u32 i = first + (u32)k * 4u;
    v[k] = (v4u)(0u);
    if ((u32)k < nv) {
      if (i + 3u < n) v[k] = *(const v4u*)(in + i);
      else { v[k].x = i < n ? in[i] : 0u; v[k].y = i + 1u < n ? in[i + 1] : 0u; v[k].z = i + 2u < n ? in[i + 2] : 0u; }
    }
  }
  u32 sum = 0;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) sum += v[k].x + v[k].y + v[k].z + v[k].w;
  const u32 incl = wave_incl_scan_u32(sum);
  if ((tid & 63u) == 63u) wsum[tid >> 6] = incl;
  __syncthreads();
  u32 run = incl - sum;
  for (u32 w = 0; w < (tid >> 6); ++w) run += wsum[w];
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const u32 i = first + (u32)k * 4u;
    if ((u32)k < nv && i < n) {
      v4u o;
      o.x = run; o.y = run + v[k].x; o.z = o.y + v[k].y; o.w = o.z + v[k].z;
      if (i + 3u < n) *(v4u*)(out + i) = o;
      else { out[i] = o.x; if (i + 1u < n) out[i + 1] = o.y; if (i + 2u < n) out[i + 2] = o.z; }
    }
    run += v[k].x + v[k].y + v[k].z + v[k].w;
  }
  if (tid == 1023u && total) *total = run;
}
// out[i] = sum(in[0..i)); out may alias in. *total (device) receives the grand total when non-null.
void exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint64_t n, uint32_t* total_dev, hipStream_t s) {
  if (n == 0) {
    if (total_dev) hipMemsetAsync(total_dev, 0, 4, s);
    return;
  }
  if (n <= 65536 && (((uintptr_t)in | (uintptr_t)out) & 15u) == 0) {   // (16-byte loads / stores)
    hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, s, (const u32*)in, (u32*)out, (u32)n, (u32*)total_dev);
    return;
  }
  const uint64_t nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  DevBuf sums_buf((nchunks + 1) * sizeof(uint32_t));   // pooled; reuse is stream-ordered (single stream per context)
  uint32_t* sums = sums_buf.as<uint32_t>();
  hipLaunchKernelGGL(k_scan_chunk_sums, dim3((unsigned)nchunks), dim3(QH_BLOCK), 0, s, (const u32*)in, (u64)n, (u32*)sums);
  if (nchunks <= 16384) {
    hipLaunchKernelGGL(k_scan_sums_inplace, dim3(1), dim3(QH_BLOCK), 0, s, (u32*)sums, (u32)nchunks, (u32*)total_dev);
  } else {
    // scan the chunk sums in place (recursion depth <= 2 for n < 2^32)
    exclusive_scan_u32(sums, sums, nchunks, sums + nchunks, s);
    if (total_dev) hipMemcpyAsync(total_dev, sums + nchunks, 4, hipMemcpyDeviceToDevice, s);
  }
  hipLaunchKernelGGL(k_scan_chunk_apply, dim3((unsigned)nchunks), dim3(QH_BLOCK), 0, s, (const u32*)in, (u32*)out, (u64)n, (const u32*)sums);
}

// ================================================================ selection vector from a keep-mask
// filter_record_batch (physical/plan/filter.rs:34, datasource/memory.rs:92): the rank of a kept row inside its
// wavefront's 64-bit ballot mask (v_mbcnt) plus the scanned per-wave offset is its output position.
__global__ __launch_bounds__(QH_BLOCK) void k_select_indices(const u64* mask, const u32* wave_offset, u64 nrows, u32* sel) {
  const u64 nwords = (nrows + 63) / 64;
  const u64 wave_global = ((u64)blockIdx.x * QH_BLOCK + threadIdx.x) >> 6;
  const u64 nwaves = ((u64)gridDim.x * QH_BLOCK) >> 6;
  const int lane = qh_lane();
  // 64 mask words per trip: lane k loads word j0 + k and its offset (two coalesced loads for 4 096 rows), then the
  // wavefront walks the NON-ZERO words out of registers. (One word per trip made a selective filter's pass a chain of
  // ~1 us memory round trips with one or two lanes working: 61 us for 60 M rows keeping 1 %.)
  for (u64 j0 = wave_global * 64; j0 < nwords; j0 += nwaves * 64) {
    const u64 j = j0 + (u64)lane;
    const u64 mine = j < nwords ? mask[j] : 0ULL;
    const u32 off = j < nwords ? wave_offset[j] : 0u;
    u64 todo = qh_ballot(mine != 0);
    while (todo) {
      const int k = __builtin_ctzll(todo);
      todo &= todo - 1;
      const u64 m = qh_readlane64(mine, k);
      const u32 o = qh_readlane32(off, k);
      if ((m >> lane) & 1) sel[o + (u32)qh_rank(m)] = (u32)((j0 + (u64)k) * 64 + (u64)lane);
    }
  }
}

// number of kept rows before each boundary row (batch starts): one output batch per input batch (filter.rs:29-43)
__global__ void k_mask_prefix_at(const u64* mask, const u32* wave_offset, const u64* rows, u32 n, u64 nrows, u32 total, u32* out) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const u64 r = rows[k];
  if (r >= nrows) { out[k] = total; return; }
  const u64 j = r >> 6;
  const u64 below = (r & 63) ? (mask[j] & ((1ULL << (r & 63)) - 1)) : 0ULL;
  out[k] = wave_offset[j] + (u32)__builtin_popcountll(below);
}

// keep-mask from a bitmap of "visited" build rows (hash_join.rs:277-343): want_set selects visited / unvisited
__global__ __launch_bounds__(QH_BLOCK) void k_mask_from_bits(const u32* bits, u64 nrows, int want_set, u64* mask, u32* wave_count) {
  const u64 nwords = (nrows + 63) / 64;
  for (u64 j = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; j < nwords; j += (u64)gridDim.x * QH_BLOCK) {
    u64 m = (u64)bits[2 * j] | ((u64)bits[2 * j + 1] << 32);
    if (!want_set) m = ~m;
    const u64 rem = nrows - j * 64;
    if (rem < 64) m &= (1ULL << rem) - 1;
    mask[j] = m;
    wave_count[j] = (u32)__builtin_popcountll(m);
  }
}

// ================================================================ gather ("take") kernels
// compute::take (hash.rs:65,79; hash_join.rs:192-199; utils/batch.rs:46,53): out[k] = in[idx[k]], NULL index -> NULL.
template <class T>
__global__ __launch_bounds__(QH_BLOCK) void k_gather_fixed(const T* in, const u32* idx, T* out, u64 m) {
  for (u64 k = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; k < m; k += (u64)gridDim.x * QH_BLOCK) {
    const u32 i = idx[k];
    T v;
    if (i == QH_NULL_IDX) memset(&v, 0, sizeof(T)); else v = in[i];
    out[k] = v;
  }
}
// ... several columns in ONE launch (blockIdx.y = the column): what an aggregate / join reads from a join output are a few
// short fixed-width columns, each a ~5 us launch of its own otherwise
template <class T>
__device__ __forceinline__ void qh_gather_loop(const void* in_, const u32* idx, void* out_, u64 m, u32 null_ones) {
  const T* in = (const T*)in_;
  T* out = (T*)out_;
  for (u64 k = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; k < m; k += (u64)gridDim.x * QH_BLOCK) {
    const u32 i = idx[k];
    T v;
    if (i == QH_NULL_IDX) memset(&v, null_ones ? 0xFF : 0, sizeof(T)); else v = in[i];
    out[k] = v;
  }
}
__global__ __launch_bounds__(QH_BLOCK) void k_gather_multi(GatherBatch b) {
  const GatherDesc d = b.d[blockIdx.y];
  switch (d.width) {
    case 1: qh_gather_loop<u8>(d.in, d.idx, d.out, d.m, d.null_ones); break;
    case 2: qh_gather_loop<u16>(d.in, d.idx, d.out, d.m, d.null_ones); break;
    case 4: qh_gather_loop<u32>(d.in, d.idx, d.out, d.m, d.null_ones); break;
    case 8: qh_gather_loop<u64>(d.in, d.idx, d.out, d.m, d.null_ones); break;
    default: qh_gather_loop<u128>(d.in, d.idx, d.out, d.m, d.null_ones); break;
  }
}
// Mask-driven compaction of a fixed-width column (Filter with a predicate that keeps a good part of the rows): wavefront
// word j of the keep mask covers rows 64 j .. 64 j + 63; a kept row's value goes to wave_offset[j] + its rank inside the
// word. The input is read in row order (fully coalesced, the dropped rows' bytes ride along), the output is written in
// runs — no selection vector is read per column (the index gather reads 4 bytes of index per kept value and column).
template <class T>
__global__ __launch_bounds__(QH_BLOCK) void k_compact_fixed(const T* in, const u64* mask, const u32* wave_offset, T* out, u64 nrows) {
  const u64 nwords = (nrows + 63) / 64;
  const u64 wave_global = ((u64)blockIdx.x * QH_BLOCK + threadIdx.x) >> 6;
  const u64 nwaves = ((u64)gridDim.x * QH_BLOCK) >> 6;
  const int lane = qh_lane();
  constexpr int U = 4;   // mask words per trip: their loads are issued together
  for (u64 j0 = wave_global * U; j0 < nwords; j0 += nwaves * U) {
    u64 m[U];
    u32 off[U];
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const u64 j = j0 + u < nwords ? j0 + u : nwords - 1;
      m[u] = j0 + u < nwords ? mask[j] : 0ULL;
      off[u] = wave_offset[j];
      const u64 i = j * 64 + lane;
      v[u] = in[i < nrows ? i : nrows - 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if ((m[u] >> lane) & 1) out[off[u] + (u32)qh_rank(m[u])] = v[u];
  }
}
// validity (or Boolean values) of the gathered rows as one ballot word per 64 output rows, plus the set count
__global__ __launch_bounds__(QH_BLOCK) void k_gather_bits(const u8* bitmap /* null = all set */, const u32* idx, u64 m, u64* out_words,
                                                         u32* set_count) {
  const u64 nwords = (m + 63) / 64;
  const u64 wave_global = ((u64)blockIdx.x * QH_BLOCK + threadIdx.x) >> 6;
  const u64 nwaves = ((u64)gridDim.x * QH_BLOCK) >> 6;
  const int lane = qh_lane();
  u32 local = 0;
  for (u64 j = wave_global; j < nwords; j += nwaves) {
    const u64 k = j * 64 + lane;
    bool bit = false;
    if (k < m) {
      const u32 i = idx[k];
      bit = i != QH_NULL_IDX && (!bitmap || qh_bit(bitmap, i));
    }
    const u64 w = qh_ballot(bit);
    if (lane == 0) { out_words[j] = w; local += (u32)__builtin_popcountll(w); }
  }
  if (lane == 0 && local) atomicAdd(set_count, local);
}
__global__ __launch_bounds__(QH_BLOCK) void k_gather_utf8_lengths(const int* offsets, const u32* idx, u64 m, u32* out_len) {
  for (u64 k = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; k < m; k += (u64)gridDim.x * QH_BLOCK) {
    const u32 i = idx[k];
    out_len[k] = i == QH_NULL_IDX ? 0u : (u32)(offsets[i + 1] - offsets[i]);
  }
}
// one wavefront per 64 output rows; every row's bytes are copied by its own lane (short strings: TPC-H flags,
// segments), rows longer than 64 bytes are copied cooperatively by the whole wavefront afterwards
__global__ __launch_bounds__(QH_BLOCK) void k_gather_utf8_bytes(const int* offsets, const u8* data, const u32* idx, u64 m, const u32* out_off,
                                                               u8* out_data) {
  const u64 nwords = (m + 63) / 64;
  const u64 wave_global = ((u64)blockIdx.x * QH_BLOCK + threadIdx.x) >> 6;
  const u64 nwaves = ((u64)gridDim.x * QH_BLOCK) >> 6;
  const int lane = qh_lane();
  for (u64 j = wave_global; j < nwords; j += nwaves) {
    const u64 k = j * 64 + lane;
    u32 len = 0, src = 0, dst = 0;
    if (k < m) {
      const u32 i = idx[k];
      if (i != QH_NULL_IDX) { src = (u32)offsets[i]; len = (u32)offsets[i + 1] - src; dst = out_off[k]; }
    }
    if (len <= 64) for (u32 b = 0; b < len; ++b) out_data[dst + b] = data[src + b];
    u64 big = qh_ballot(len > 64);
    while (big) {
      const int l = __builtin_ctzll(big);
      big &= big - 1;
      const u32 s2 = qh_readlane32(src, l), d2 = qh_readlane32(dst, l), n2 = qh_readlane32(len, l);
      for (u32 b = lane; b < n2; b += 64) out_data[d2 + b] = data[s2 + b];
    }
  }
}
__global__ void k_store_u32(u32* p, u32 v) { *p = v; }
// longest value of a Utf8 column (sizes the packed key words of group / join keys)
__global__ __launch_bounds__(QH_BLOCK) void k_utf8_max_len(const int* offsets, u64 n, u32* out) {
  u32 m = 0;
  for (u64 i = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * QH_BLOCK) {
    const u32 l = (u32)(offsets[i + 1] - offsets[i]);
    m = l > m ? l : m;
  }
  m = (u32)qh_wave_max_u64(m);
  if (qh_lane() == 0 && m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
}

// largest |value| of an Int64 (WORDS = 1) / Decimal128 (WORDS = 2) column: out[0] = max |v| over the values that fit 63 bits,
// out[1] != 0 when some value does not. NULL slots count too (an upper bound is all the callers need). One atomic per
// wavefront, none when it would change nothing.
template <int WORDS>
__global__ __launch_bounds__(QH_BLOCK) void k_value_maxabs(const u64* v, u64 n, u64* out, u32* narrow32) {
  u64 m = 0;
  u32 big = 0;
  for (u64 i = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * QH_BLOCK) {
    const u64 lo = v[(size_t)i * WORDS];
    const u64 hi = WORDS == 2 ? v[(size_t)i * WORDS + 1] : (u64)((i64)lo >> 63);
    // (speculative 4-byte narrow copy in the same pass: adopted by the host when the maximum turns out to fit 31 bits)
    if (narrow32) narrow32[i] = (u32)lo;
    const bool fits = hi == (u64)((i64)lo >> 63);
    const u64 a = (lo >> 63) ? (u64)0 - lo : lo;
    if (fits && a < (1ULL << 63)) m = a > m ? a : m; else big = 1;
  }
  m = qh_wave_max_u64(m);
  const u64 anybig = qh_ballot(big != 0);
  if (qh_lane() == 0) {
    if (m > __hip_atomic_load(&out[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) (void)__hip_atomic_fetch_max(&out[0], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (anybig && !__hip_atomic_load(&out[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) (void)__hip_atomic_fetch_or(&out[1], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// the narrow copy of a Decimal128 column whose every value fits T (DevColumn::narrow): the low bytes of the two's complement value
template <class T, int WORDS>
__global__ __launch_bounds__(QH_BLOCK) void k_narrow_decimal(const u64* v, u64 n, T* out) {
  for (u64 i = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * QH_BLOCK) out[i] = (T)v[WORDS * i];
}

// Value range of an integer-like column (sign-extended to 64 bits): out[0] = max of (v ^ sign bit), out[1] = max of
// ~(v ^ sign bit), i.e. the order-preserving unsigned images of max and min, both gathered with atomic max from zero-filled
// words (one atomic pair per wavefront, none when it would change nothing). NULL slots take part with whatever their value
// slot holds: the range may only be wider for it. Decides whether a hash join can address its table by the key itself.
template <class T>
__global__ __launch_bounds__(QH_BLOCK) void k_value_range(const T* v, u64 n, u64* out) {
  u64 hi = 0, lo = 0;
  for (u64 i = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * QH_BLOCK) {
    const u64 img = (u64)(i64)v[i] ^ 0x8000000000000000ULL;
    hi = img > hi ? img : hi;
    lo = ~img > lo ? ~img : lo;
  }
  hi = qh_wave_max_u64(hi);
  lo = qh_wave_max_u64(lo);
  if (qh_lane() == 0) {
    if (hi > __hip_atomic_load(&out[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) (void)__hip_atomic_fetch_max(&out[0], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lo > __hip_atomic_load(&out[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) (void)__hip_atomic_fetch_max(&out[1], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ================================================================ hash join
// JoinHashMap (physical/plan/join/hash_join.rs:39-107) keeps `hash -> last row + 1` and a `next` chain, built in
// reverse so that chains ascend. Here: an open-addressing table keyed by the REAL key words (so there is no
// candidate re-check, hash_join.rs:191-215) maps every distinct build key to a slot; rows are then grouped by slot
// (stable radix sort => ascending row order inside a group, the order the reverse-built chains produce) into a
// CSR layout (start/count per slot). NULL keys are never inserted and never probe (eq of NULL is NULL).
// Slot = [state][key words W]. state: 0 empty, 1 being written, v >= 2 ready with v - 2 = the build row that inserted the
// key — for unique build keys (every FK -> PK join) that row IS the whole match list, so neither the per-slot counts nor
// a slot -> row array are ever touched (a build row then costs two random accesses: the slot and its filter bit).
// Further rows of a key only count themselves in extra[slot] and raise the duplicates flag (status[QS_MAXCOUNT] = 2).
template <int W>
__global__ __launch_bounds__(QH_BLOCK) void k_join_build_insert(const u64* keys, const u64* keyvalid, u64 n, u64* table, u32 nslots,
                                                               u32* row_slot, u32* extra, u64* bloom, u32 bloom_mask, u32* status) {
  u32 flags = 0;
  for (u64 i = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * QH_BLOCK) {
    u32 sid = nslots;   // NULL key: sorts behind every real slot, never probed
    if ((keyvalid[i >> 6] >> (i & 63)) & 1) {
      u64 k[W];
#pragma unroll
      for (int w = 0; w < W; ++w) k[w] = keys[(size_t)w * n + i];
      const u64 h = qh_key_hash<W>(k);
      u32 s = (u32)h & (nslots - 1);
      u32 probes = 0;
      while (probes < nslots) {
        u64* slot = table + (size_t)s * (1 + W);
        const u64 st = qh_ld64<MemHbm>(slot);
        if (st >= QH_READY) {
          bool eq = true;
#pragma unroll
          for (int w = 0; w < W; ++w) eq &= qh_ld64<MemHbm>(slot + 1 + w) == k[w];
          if (eq) { sid = s; atomicAdd(&extra[s], 1u); flags |= 1u << QS_MAXCOUNT; break; }
          s = (s + 1) & (nslots - 1); ++probes;
        } else if (st == QH_EMPTY) {
          if (qh_cas64<MemHbm>(slot, QH_EMPTY, QH_BUSY)) {
#pragma unroll
            for (int w = 0; w < W; ++w) qh_st64<MemHbm>(slot + 1 + w, k[w]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // key words visible before the slot reads as ready
            qh_st64<MemHbm>(slot, (u64)i + 2);
            (void)__hip_atomic_fetch_or(&bloom[qh_filter_word(h, bloom_mask)], qh_filter_mask((u32)h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sid = s;
            break;
          }
          // lost the claim: look at the same slot again
        }
        // QH_BUSY: the claimer is between claim and publish; look again (a lane never spins inside an iteration)
      }
      if (probes >= nslots) flags |= 1u << QS_OVERFLOW;
    }
    row_slot[i] = sid;
  }
  // one flag update per wavefront, none when the flag is already up
  const u64 dup = qh_ballot((flags >> QS_MAXCOUNT) & 1u), ovf = qh_ballot((flags >> QS_OVERFLOW) & 1u);
  if (qh_lane() == 0) {
    if (dup && __hip_atomic_load(&status[QS_MAXCOUNT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 2u) atomicMax(&status[QS_MAXCOUNT], 2u);
    if (ovf) atomicOr(&status[QS_OVERFLOW], 1u);
  }
}
// LDS-staged build, step 2 (step 1: qh_join_scatter_body in device/qhip_device.hpp): ONE workgroup per region. Thread w
// collects the region's entries [row + 2 | key words] that step-1 workgroup w (w + 512, ...) left in its own range of
// `entries` (their positions: first[region][w], first[region + 1][w]) and inserts them into an open-addressing image of
// the region in LDS (DS compare-and-swap on the state word, probe sequence wrapping inside the region); the region's
// slice of the hash filter is assembled beside it, and both are stored as whole lines — the table is never memset and
// never sees an HBM atomic. A second row with an equal key (the unique-key speculation failed) or a region with more
// entries than 7/8 of its slots raises status[QS_MAXCOUNT] = 2: the host then builds the legacy layout.
#define QH_REGION_BLOCK 512
// one entry into the LDS image `lt` of a region of S slots (+ its filter bits into `lb`); true when an equal key is
// already there
template <int W>
__device__ __forceinline__ bool region_insert(u64* lt, u64* lb, u32 S, u32 slot_bits, u32 bword_bits, const u64* ent) {
  const u64 h = qh_key_hash<W>(ent + 1);
  u32 s = (u32)h & (S - 1);
  u32 probes = 0;
  while (probes < S) {
    u64* slot = lt + (size_t)s * (1 + W);
    // claim first, look afterwards: the compare-and-swap returns what the slot held (at load <= 1/2 most first looks find
    // it empty: one LDS operation less per insert — the inserts are 48 of this kernel's 60 us on a 1.5 M-row build side)
    u64 st = QH_EMPTY;
    if (__hip_atomic_compare_exchange_strong(slot, &st, (u64)QH_BUSY, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
#pragma unroll
      for (int w = 0; w < W; ++w) qh_st64<MemLds>(slot + 1 + w, ent[1 + w]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // key words before the slot reads as ready
      qh_st64<MemLds>(slot, ent[0]);
      (void)__hip_atomic_fetch_or(&lb[qh_rfilter_word(h, slot_bits, bword_bits)], qh_rfilter_mask(h, slot_bits, bword_bits), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return false;
    }
    if (st >= QH_READY) {
      bool eq = true;
#pragma unroll
      for (int w = 0; w < W; ++w) eq &= qh_ld64<MemLds>(slot + 1 + w) == ent[1 + w];
      if (eq) return true;
      s = (s + 1) & (S - 1); ++probes;
    }
    // QH_BUSY: the claimer is between claim and publish; look again (a lane never spins inside an iteration)
  }
  return false;
}

template <int W>
__global__ __launch_bounds__(QH_REGION_BLOCK) void k_join_region_build(const u64* entries, const u32* first, u32 n_wgs, u32 n_regions,
                                                                        u32 rows_per_wg, u64* table, u64* bloom, u32 slot_bits,
                                                                        u32 bword_bits, u32* status, u32 dbg) {
  constexpr u32 RB = QH_REGION_BLOCK;
  const u32 S = 1u << slot_bits, BW = 1u << bword_bits, tid = threadIdx.x;
  // XCD-aware region order: workgroups b, b + 8, b + 16, ... share an XCD (and its L2), so they take CONSECUTIVE regions. A
  // step-1 workgroup's entries of neighbouring regions lie next to each other (4 regions per 128-byte line, 16 per line
  // of `first`): with the plain order every line was fetched by 4 to 8 different L2s.
  const u32 per_xcd = (n_regions + 7) / 8, reg = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (reg >= n_regions) return;               // (workgroup-uniform; the grid is 8 * per_xcd)
  u64* lt = (u64*)qh_dyn_lds;                 // [S][1 + W]
  u64* lb = lt + (size_t)S * (1 + W);         // [BW]
  __shared__ u32 total;
  // this thread's share: the region's entries of step-1 workgroups tid, tid + RB, ... (the loads of the first one are
  // issued before the LDS image is cleared)
  u32 lo = 0, hi = 0;
  const u32* const f0 = first + (size_t)reg * n_wgs;   // first[region][workgroup]: two contiguous rows
  const u32* const f1 = f0 + n_wgs;
  if (tid < n_wgs) { lo = f0[tid]; hi = f1[tid]; }
  if (tid == 0) total = 0;
  for (u32 k = tid; k < S * (1 + W); k += RB) lt[k] = 0;
  for (u32 k = tid; k < BW; k += RB) lb[k] = 0;
  __syncthreads();
  u32 mine = hi - lo;
  for (u32 w = tid + RB; w < n_wgs; w += RB) mine += f1[w] - f0[w];
  if (mine) atomicAdd(&total, mine);
  __syncthreads();
  const bool overfull = total > S - (S >> 3);   // workgroup-uniform
  bool careful = overfull;
  if (!overfull && !(dbg & 4u)) {
    for (u32 w = tid; w < n_wgs; w += RB) {
      if (w != tid) { lo = f0[w]; hi = f1[w]; }
      const u64* src = entries + ((size_t)w * rows_per_wg + lo) * (1 + W);
      const u32 n = hi - lo;
      // four entries at a time: their loads are issued together (a thread has two entries on average, rarely more than four)
      for (u32 e0 = 0; e0 < n; e0 += 4) {
        u64 ent[4][1 + W];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32 e = e0 + j < n ? e0 + j : n - 1;
#pragma unroll
          for (int x = 0; x < 1 + W; ++x) ent[j][x] = src[(size_t)e * (1 + W) + x];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (e0 + j < n && !(dbg & 1u)) careful |= region_insert<W>(lt, lb, S, slot_bits, bword_bits, ent[j]);
          if (dbg & 1u) lt[(ent[j][0] ^ ent[j][1]) & 7] = 1;
        }
      }
    }
  }
  __syncthreads();
  // an overfull region is stored EMPTY: every state word the (speculatively launched) probe reads is a valid one
  u64* gr = table + ((size_t)reg << slot_bits) * (1 + W);
  typedef u64 v2u64 __attribute__((ext_vector_type(2)));
  const u32 pairs = S * (1 + W) / 2;           // S is even
  for (u32 k = tid; k < ((dbg & 2u) ? 8u : pairs); k += RB) {
    v2u64 v;
    v.x = overfull ? 0ULL : lt[2 * k];
    v.y = overfull ? 0ULL : lt[2 * k + 1];
    ((v2u64*)gr)[k] = v;
  }
  u64* gb = bloom + ((size_t)reg << bword_bits);
  for (u32 k = tid; k < BW; k += RB) gb[k] = overfull ? 0ULL : lb[k];
  const u64 any = qh_ballot(careful);
  if (any && qh_lane() == 0 && __hip_atomic_load(&status[QS_MAXCOUNT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 2u) atomicMax(&status[QS_MAXCOUNT], 2u);
}

// duplicated build keys: rows per slot = (slot occupied ? 1 : 0) + extra[slot]
template <int W>
__global__ __launch_bounds__(QH_BLOCK) void k_join_full_counts(const u64* table, u32 nslots, u32* count) {
  for (u32 s = blockIdx.x * QH_BLOCK + threadIdx.x; s < nslots; s += gridDim.x * QH_BLOCK)
    if (table[(size_t)s * (1 + W)] >= QH_READY) count[s] += 1u;
}

// The dense join build's byte map -> its bitmap (round 4): thread w turns bytes[32 w .. 32 w + 31] into bitmap word w (bit i =
// "byte i carries this execution's stamp") and the grid counts the stamped bytes; the LAST workgroup to finish compares the
// count with the rows the build kernel inserted — fewer stamped bytes = two rows stamped the same byte = a duplicate build key
// (status[QS_MAXCOUNT] = 2, the same signal the atomic form raises; the join is then re-run with the chained layout).
// counters: [0] rows inserted (build kernel), [1] stamped bytes, [2] workgroups done. The byte map is padded to whole words.
__global__ __launch_bounds__(QH_BLOCK) void k_bytes_to_bits(const u8* bytes, u32 gen, u32* bits, u32 nwords, u32* counters, u32* status) {
  const u32 w = blockIdx.x * QH_BLOCK + threadIdx.x;
  u32 word = 0;
  if (w < nwords) {
    typedef u32 v4 __attribute__((ext_vector_type(4)));
    const v4* src = (const v4*)(bytes + (size_t)w * 32);
    const v4 a = __builtin_nontemporal_load(src), b = __builtin_nontemporal_load(src + 1);
    const u32 d[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const u32 g4 = gen * 0x01010101u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const u32 t = d[k] ^ g4;                                                     // a zero byte = a stamped one
      const u32 eq = ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u;        // exact per byte: 0x80 where the byte of t is zero
      word |= (((eq >> 7) * 0x10204080u) >> 28) << (4 * k);                         // bytes 0..3 -> bits 0..3
    }
    bits[w] = word;
  }
  const u32 mine = (u32)qh_wave_sum_u64((u64)__builtin_popcount(word));
  __shared__ u32 wg_total;
  if (threadIdx.x == 0) wg_total = 0;
  __syncthreads();
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&wg_total, mine);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (wg_total) atomicAdd(&counters[1], wg_total);
    __threadfence();
    const u32 ticket = atomicAdd(&counters[2], 1u);
    if (ticket == gridDim.x - 1) {
      const u32 stamped = __hip_atomic_load(&counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const u32 inserted = __hip_atomic_load(&counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (stamped < inserted) atomicMax(&status[QS_MAXCOUNT], 2u);
    }
  }
}

// probe pass 2 (get_matches_indices + probe_hash_table's index vectors, hash_join.rs:70-107,177-216): turn the
// (slot, probe row) entries that pass 1 (qk_join_probe, device/qhip_device.hpp) compacted per tile into (build row,
// probe row) pairs — probe-row major, build rows ascending (hash_join.rs:475-512 pins that order). A wavefront owns the
// same tile as in pass 1; tile_off is the exclusive scan of pass 1's per-tile pair counts. Unique build keys
// (start == nullptr): the entry already holds the one build row. cnt_out / pair_off (Right / Full joins) receive the
// pair count and first pair position of every MATCHING probe row (cnt_out is zero-filled by the caller).
__global__ __launch_bounds__(QH_BLOCK) void k_join_emit(const u32* ent_slot, const u32* ent_row, const u32* chunk_nent, const u32* chunk_off,
                                                       const u32* count, const u32* start, const u32* rows, const u32* row_of, u64 nchunks, u64 chunk_rows, u32* b_idx,
                                                       u32* p_idx, u32* pair_off, u32* cnt_out, u32* visited, u32 cap, const u32* stat_block,
                                                       u32* publish, u32* rows_out, i64* key_out, u64 key_min) {
  // a join of deferred size (cap = the room its output has; stat_block = [build status | probe status | pair total]): rows
  // [total, cap) of the index vectors repeat row 0 of both sides (valid to gather, never counted), the total goes to the
  // output table's device-side row count and the status block to page-locked host memory, where the consumer's
  // synchronisation finds it — all in this launch instead of a fill, a copy and a device-to-device copy behind it
  if (stat_block) {
    const u32 total = stat_block[2 * QS_WORDS];
    for (u64 k = (u64)total + (u64)blockIdx.x * QH_BLOCK + threadIdx.x; k < cap; k += (u64)gridDim.x * QH_BLOCK) { b_idx[k] = 0; p_idx[k] = 0; }
    if (blockIdx.x == 0 && threadIdx.x <= 2 * QS_WORDS) {
      publish[threadIdx.x] = stat_block[threadIdx.x];
      if (threadIdx.x == 2 * QS_WORDS) *rows_out = total;
      __threadfence_system();
    }
  }
  const int lane = qh_lane();
  // a wavefront per chunk (the consecutive tiles one probe wavefront owned): its entries are one run
  for (u64 ch = (u64)blockIdx.x * (QH_BLOCK / 64) + (threadIdx.x >> 6); ch < nchunks; ch += (u64)gridDim.x * (QH_BLOCK / 64)) {
    const u32 n = chunk_nent[ch];
    u32 base = chunk_off[ch];
    const u64 e0 = ch * chunk_rows;
    if (!start && !pair_off && !visited) {
      // unique build keys, nothing but the pairs wanted (every Inner FK -> PK join): entry j of the chunk IS pair base + j, so
      // there is no scan and no dependence between the trips — four of them in flight (Q3's join 1: 1 640 entries per chunk
      // were 26 dependent trips of two chained loads each, 20 us for 3 M pairs)
      for (u32 j0 = 0; j0 < n; j0 += 256) {
        u32 sid[4], p[4];
        bool live[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const u32 j = j0 + (u32)u * 64u + (u32)lane;
          live[u] = j < n;
          sid[u] = live[u] ? ent_slot[e0 + j] : 0u;
          p[u] = live[u] ? ent_row[e0 + j] : 0u;
        }
        // (dense layout, Int64 key: the entry IS key - min — the join key of every output row is written here for nothing, so that
        // an aggregate grouping by it streams a plain column instead of gathering lineitem.l_orderkey through p_idx)
        if (key_out) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const u32 o = base + j0 + (u32)u * 64u + (u32)lane;
            if (live[u] && o < cap) key_out[o] = (i64)(key_min + (u64)sid[u]);
          }
        }
        if (row_of) {
#pragma unroll
          for (int u = 0; u < 4; ++u) sid[u] = live[u] ? row_of[sid[u]] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const u32 o = base + j0 + (u32)u * 64u + (u32)lane;
          if (live[u] && o < cap) { b_idx[o] = sid[u]; p_idx[o] = p[u]; }
        }
      }
      continue;
    }
    for (u32 j0 = 0; j0 < n; j0 += 64) {
      const u32 j = j0 + lane;
      const bool live = j < n;
      u32 sid = live ? ent_slot[e0 + j] : 0u;
      const u32 p = live ? ent_row[e0 + j] : 0u;
      // dense (direct-address) layout: the entry holds key - min; the build row is looked up HERE, for the matching rows only
      // (qh_join_probe_dense_body keeps the lookup out of its streaming loop)
      if (row_of && live) sid = row_of[sid];
      u32 c = start ? count[sid] : 1u;
      c = live ? c : 0u;
      const u32 incl = wave_incl_scan_u32(c);
      const u32 o = base + incl - c;
      base += qh_readlane32(incl, 63);
      if (live) {
        if (pair_off) { pair_off[p] = o; cnt_out[p] = c; }
        const u32 s0 = start ? start[sid] : 0u;
        for (u32 q = 0; q < c; ++q) {
          const u32 b = start ? rows[s0 + q] : sid;
          if (o + q >= cap) break;   // (deferred sizing: more pairs than the output has room for — the host finds out later)
          b_idx[o + q] = b;
          p_idx[o + q] = p;
          if (visited) atomicOr(&visited[b >> 5], 1u << (b & 31));
        }
      }
    }
  }
}
// visited bitmap (hash_join.rs:166-167,253-255) and surviving-pair count per probe row
__global__ __launch_bounds__(QH_BLOCK) void k_join_mark(const u32* b_idx, const u32* p_idx, u64 m, u32* visited_bits, u32* cnt_per_probe) {
  for (u64 k = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; k < m; k += (u64)gridDim.x * QH_BLOCK) {
    const u32 b = b_idx[k];
    if (b != QH_NULL_IDX) atomicOr(&visited_bits[b >> 5], 1u << (b & 31));
    if (cnt_per_probe) atomicAdd(&cnt_per_probe[p_idx[k]], 1u);
  }
}
// adjust_right_indices (join/mod.rs:176-207): probe rows without a surviving pair are emitted once with a NULL build index
__global__ __launch_bounds__(QH_BLOCK) void k_join_out_counts(const u32* cnt, u64 np, u32* out_cnt) {
  for (u64 p = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; p < np; p += (u64)gridDim.x * QH_BLOCK) out_cnt[p] = cnt[p] ? cnt[p] : 1u;
}
__global__ __launch_bounds__(QH_BLOCK) void k_join_adjust_right(const u32* b_in, const u32* cnt, const u32* in_off, const u32* out_off, u64 np,
                                                               u32* b_out, u32* p_out) {
  for (u64 p = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; p < np; p += (u64)gridDim.x * QH_BLOCK) {
    const u32 c = cnt[p], o = out_off[p];
    if (!c) { b_out[o] = QH_NULL_IDX; p_out[o] = (u32)p; continue; }
    const u32 i0 = in_off[p];
    for (u32 k = 0; k < c; ++k) { b_out[o + k] = b_in[i0 + k]; p_out[o + k] = (u32)p; }
  }
}
// first position in the ascending array `a` (m values) whose value is >= bound[k] — output-batch boundaries of a join
// from the probe rows of its pairs (hash_join.rs:363-372: one output batch per probe batch)
// (m_dev: the number of values lives on the device — a join output of deferred size — and m is its capacity)
__global__ __launch_bounds__(QH_BLOCK) void k_lower_bound_u32(const u32* a, u64 m, const u32* m_dev, const u64* bound, u32 nb, u32* pos) {
  const u32 k = blockIdx.x * QH_BLOCK + threadIdx.x;
  if (k >= nb) return;
  const u64 v = bound[k];
  if (m_dev) m = *m_dev < m ? *m_dev : m;
  u64 lo = 0, hi = m;
  while (lo < hi) { const u64 mid = (lo + hi) >> 1; if ((u64)a[mid] < v) lo = mid + 1; else hi = mid; }
  pos[k] = (u32)lo;
}
// ================================================================ sort (physical/plan/sort.rs:48-82)
// lexsort_to_indices as a sequence of stable LSD radix passes over order-preserving key images, least significant key
// word first; the implicit last key of the reference (the row number, sort.rs:62-73) is the initial order of `idx`.
__global__ __launch_bounds__(QH_BLOCK) void k_sort_gather_img(const u64* img, const u32* idx, u64 n, u64 flip, u64* out) {
  for (u64 k = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; k < n; k += (u64)gridDim.x * QH_BLOCK) out[k] = img[idx[k]] ^ flip;
}
// null placement pass: out[k] = 1 for the rows that go LAST in this key (nulls_first: the valid ones)
__global__ __launch_bounds__(QH_BLOCK) void k_sort_gather_valid(const u64* validwords, const u32* idx, u64 n, u32 nulls_first, u64* out) {
  for (u64 k = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; k < n; k += (u64)gridDim.x * QH_BLOCK) {
    const u32 i = idx[k];
    const u32 v = (u32)((validwords[i >> 6] >> (i & 63)) & 1);
    out[k] = nulls_first ? v : (v ^ 1u);
  }
}
// Utf8 key: chunk c >= 0 -> bytes [8c, 8c+8) of the value, big-endian, zero-padded (unsigned compare of the chunk sequence
// = bytewise compare up to trailing NULs); c < 0 -> the length (orders "ab" before "ab\0"). NULL values -> 0.
__global__ __launch_bounds__(QH_BLOCK) void k_sort_utf8_chunk(const int* offsets, const u8* data, const u8* validity, const u32* idx, u64 n,
                                                             int chunk, u64 flip, u64* out) {
  for (u64 k = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; k < n; k += (u64)gridDim.x * QH_BLOCK) {
    const u32 i = idx[k];
    u64 v = 0;
    if (!validity || qh_bit(validity, i)) {
      const int b = offsets[i], len = offsets[i + 1] - b;
      if (chunk < 0) v = (u64)(u32)len;
      else
        for (int j = 0; j < 8; ++j) { const int p = chunk * 8 + j; v = (v << 8) | (p < len ? (u64)data[b + p] : 0ULL); }
    }
    out[k] = v ^ flip;
  }
}

// number of set bits among the first n bits of `words` (one atomic per workgroup)
__global__ __launch_bounds__(QH_BLOCK) void k_count_bits(const u64* words, u64 n, u32* out) {
  __shared__ u32 part[QH_BLOCK / 64];
  const u64 nwords = (n + 63) / 64;
  u32 c = 0;
  for (u64 j = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; j < nwords; j += (u64)gridDim.x * QH_BLOCK) {
    u64 w = words[j];
    if (j == nwords - 1 && (n & 63)) w &= (1ULL << (n & 63)) - 1;
    c += (u32)__popcll(w);
  }
  c = (u32)qh_wave_sum_u64(c);
  if (qh_lane() == 0) part[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) { u32 t = 0; for (int w = 0; w < QH_BLOCK / 64; ++w) t += part[w]; if (t) atomicAdd(out, t); }
}

// pair k of a nested-loop / cross join block: minor[k] = minor0 + k % n_minor, major[k] = major0 + k / n_minor
__global__ __launch_bounds__(QH_BLOCK) void k_pair_indices(u32* minor, u32* major, u64 n, u32 n_minor, u32 minor0, u32 major0) {
  for (u64 k = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; k < n; k += (u64)gridDim.x * QH_BLOCK) {
    minor[k] = minor0 + (u32)(k % n_minor);
    major[k] = major0 + (u32)(k / n_minor);
  }
}

// Utf8 upload: the int32 offsets of a batch arrive as they are on the host; adding (position of the batch's bytes in the
// concatenated data buffer - first offset of the batch) rebases them — on the device instead of a host loop over every row
__global__ __launch_bounds__(QH_BLOCK) void k_add_i32(int* p, u64 n, int delta) {
  for (u64 i = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * QH_BLOCK) p[i] += delta;
}

// Bitmap concatenation (validity / Boolean values of exchanged or concatenated tables): bits [0, nbits) of `src` (all ones
// when src is null: a peer without NULLs ships no bitmap) are OR-ed into bits [dst_pos, dst_pos + nbits) of the zeroed
// `dst`. One thread per destination word of the range; the sections of one destination are appended in stream order, so
// the two boundary words shared with the neighbouring sections never race.
__global__ __launch_bounds__(QH_BLOCK) void k_bits_append(u32* dst, u64 dst_pos, const unsigned char* src, u64 nbits) {
  const u64 w = (dst_pos >> 5) + (u64)blockIdx.x * QH_BLOCK + threadIdx.x;
  const u64 end = dst_pos + nbits;
  if (w * 32 >= end) return;
  const u64 lo = w * 32 > dst_pos ? w * 32 : dst_pos;
  const u64 hi = w * 32 + 32 < end ? w * 32 + 32 : end;
  const u32 cnt = (u32)(hi - lo);
  const u32 mask = cnt == 32 ? 0xffffffffu : ((1u << cnt) - 1u);
  u32 bits = mask;
  if (src) {
    const u64 sb = lo - dst_pos, b0 = sb >> 3, nbytes = (nbits + 7) >> 3;
    u64 v = 0;
    for (u32 j = 0; j < 5; ++j)
      if (b0 + j < nbytes) v |= (u64)src[b0 + j] << (8 * j);
    bits = (u32)(v >> (sb & 7)) & mask;
  }
  if (bits) dst[w] |= bits << (u32)(lo & 31);
}

// ... and for MANY batches in one launch: row i belongs to the batch b with starts[b] <= i < starts[b + 1]
__global__ __launch_bounds__(QH_BLOCK) void k_add_i32_batched(int* p, u64 n, const u64* starts, const int* shifts, u32 nb) {
  for (u64 i = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * QH_BLOCK) {
    u32 lo = 0, hi = nb;                    // last b with starts[b] <= i
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (starts[mid] <= i) lo = mid; else hi = mid; }
    p[i] += shifts[lo];
  }
}

// index-vector composition for deferred gathers: out[k] = inner[idx[k]], NULL stays NULL
__global__ __launch_bounds__(QH_BLOCK) void k_gather_u32_nullable(const u32* inner, const u32* idx, u32* out, u64 m) {
  for (u64 k = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; k < m; k += (u64)gridDim.x * QH_BLOCK) {
    const u32 i = idx[k];
    out[k] = i == QH_NULL_IDX ? QH_NULL_IDX : inner[i];
  }
}
__global__ __launch_bounds__(QH_BLOCK) void k_iota_u32(u32* out, u64 n, u32 first) {
  for (u64 i = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * QH_BLOCK) out[i] = first + (u32)i;
}
__global__ __launch_bounds__(QH_BLOCK) void k_iota_stride_u32(u32* out, u64 n, u32 stride) {
  for (u64 i = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * QH_BLOCK) out[i] = (u32)(i * stride);
}
__global__ __launch_bounds__(QH_BLOCK) void k_fill_u32(u32* out, u64 n, u32 v) {
  for (u64 i = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * QH_BLOCK) out[i] = v;
}
// out[k] = off[rows[k]] for boundary rows (batch starts), rows[k] == n -> total
__global__ void k_lookup_u32(const u32* off, const u64* rows, u32 n, u64 nrows, u32 total, u32* out) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = rows[k] >= nrows ? total : off[rows[k]];
}
// partition id per row from its key words (exchange, generic path): qh_part_hash, the function the fused pass 1 evaluates
#define QH_MAX_PARTS 1024
template <int W>
__global__ __launch_bounds__(QH_BLOCK) void k_partition_ids(const u64* keys, u64 n, u32 nparts, u32* part, u32* hist) {
  // histogram in LDS, one global atomic per workgroup and part (the parts are few: per-row atomics would all land on
  // the same handful of words)
  __shared__ u32 lh[QH_MAX_PARTS];
  for (u32 p = threadIdx.x; p < nparts; p += QH_BLOCK) lh[p] = 0;
  __syncthreads();
  for (u64 i = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * QH_BLOCK) {
    u64 k[W];
#pragma unroll
    for (int w = 0; w < W; ++w) k[w] = keys[(size_t)w * n + i];   // (a NULL key's words are zero already, qh_eval_keys_body)
    const u32 pid = qh_part_hash<W>(k, true, nparts);
    part[i] = pid;
    atomicAdd(&lh[pid], 1u);
  }
  __syncthreads();
  for (u32 p = threadIdx.x; p < nparts; p += QH_BLOCK) if (lh[p]) atomicAdd(&hist[p], lh[p]);
}

// ================================================================ exchange, pass 2: rows -> per-part runs
// (pass 1 and the design: device/qhip_device.hpp qh_part_ids_body.) The wavefront that counted a row range in pass 1 reads it
// again: the part bytes, then one column after the other. A tile of 64 * R rows is ranked per part with ballots (stable: a
// part keeps the input's row order), ordered by part in the wavefront's OWN LDS area and written out so that consecutive
// lanes store consecutive values of a part's run. Every column is ONE buffer over all parts (part p = positions
// [runs[p * n_units], runs[(p + 1) * n_units]) of it): a part's column is a slice, never a copy. No workgroup barrier, no
// atomic: the LDS traffic of a wavefront is ordered by the hardware (DS operations of one wavefront execute in order).
// NPT = 8 / 16: the parts' counters are unrolled SGPR arrays; NPT = 0: up to 255 parts, the distinct parts of a tile row are
// walked with readfirstlane + ballot and the counters live in LDS.
template <class T>
__device__ __forceinline__ void qh_scatter_column(const PartCol& col, const i64 tb, const i64 last, const int lane, const u32 (&id)[4], const u32 (&pos)[4],
                                                  const u32 total, u8* sval, const u32* sdst) {
  constexpr int R = 4;
  T v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    i64 row = tb + r * 64 + lane;
    row = row < last ? row : last - 1;
    if (col.kind == 1) {   // the row number itself (T = u32)
      if constexpr (sizeof(T) == 4) v[r] = (T)(u32)row;
      else v[r] = T{};
    } else {
      const u64 src = col.idx ? (u64)col.idx[row] : (u64)row;
      v[r] = col.idx ? ((const T*)col.src)[src] : __builtin_nontemporal_load((const T*)col.src + src);
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) if (id[r] != 0xFFu) ((T*)sval)[pos[r]] = v[r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const u32 j = (u32)k * 64u + (u32)lane;
    if (j < total) ((T*)col.out)[sdst[j]] = ((const T*)sval)[j];
  }
  asm volatile("" ::: "memory");
}

template <int NPT>
__global__ __launch_bounds__(QH_BLOCK) void k_part_scatter(PartScatterArgs A) {
  constexpr int R = 4, TILE = 64 * R, NW = QH_BLOCK / 64, NPL = NPT > 0 ? NPT : 256;
  __shared__ __attribute__((aligned(16))) u8 s_val[NW][TILE * 16];
  __shared__ u32 s_dst[NW][TILE];
  __shared__ u32 s_cur[NW][NPL];   // per part: position (in the column buffers) of this wavefront's next row of the part
  __shared__ u32 s_tf[NW][NPL];    // per part: first position of the part inside the ordered tile
  const int lane = qh_lane();
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const u32 unit = blockIdx.x * NW + (u32)wv;
  if (unit >= A.n_units) return;
  i64 nrows = (i64)A.nrows;
  if (A.nrows_dev) { const i64 d = (i64)*A.nrows_dev; nrows = d < nrows ? d : nrows; }
  const i64 first = (i64)unit * A.rows_per_unit;
  const i64 last = first + A.rows_per_unit < nrows ? first + A.rows_per_unit : nrows;
  if (first >= last) return;
  const u32 np = A.n_parts;
  for (u32 p = (u32)lane; p < np; p += 64) s_cur[wv][p] = A.runs[(size_t)p * A.n_units + unit];
  u32 idn[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { const i64 row = first + r * 64 + lane; idn[r] = row < last ? (u32)A.ids[row] : 0xFFu; }
  for (i64 tb = first; tb < last; tb += TILE) {
    u32 id[R], q[R], pos[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { id[r] = idn[r]; q[r] = 0; pos[r] = 0; }
    if (tb + TILE < last) {   // the next tile's part bytes fly while this tile is ranked and moved
#pragma unroll
      for (int r = 0; r < R; ++r) { const i64 row = tb + TILE + r * 64 + lane; idn[r] = row < last ? (u32)A.ids[row] : 0xFFu; }
    }
    u32 total = 0, my_cnt = 0;   // my_cnt: rows of part `lane` (NPT = 0: parts lane * 4 .. lane * 4 + 3 summed) in this tile
    if (NPT > 0) {
      u32 run[NPT > 0 ? NPT : 1];
#pragma unroll
      for (int p = 0; p < NPT; ++p) run[p] = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int p = 0; p < NPT; ++p) {
          const u64 m = qh_ballot(id[r] == (u32)p);
          q[r] = id[r] == (u32)p ? run[p] + (u32)qh_rank(m) : q[r];
          run[p] += (u32)__builtin_popcountll(m);
        }
      }
      u32 my_tf = 0;
#pragma unroll
      for (int p = 0; p < NPT; ++p) { my_tf = lane == p ? total : my_tf; my_cnt = lane == p ? run[p] : my_cnt; total += run[p]; }
      if (lane < NPT) s_tf[wv][lane] = my_tf;
    } else {
      for (u32 p = (u32)lane; p < np; p += 64) s_tf[wv][p] = 0;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < R; ++r) {
        u64 todo = qh_ballot(id[r] != 0xFFu);
        while (todo) {   // wave-uniform
          const int l = __builtin_ctzll(todo);
          const u32 p = qh_readlane32(id[r], l);
          const u64 m = qh_ballot(id[r] == p);
          const u32 base = s_tf[wv][p];
          if (id[r] == p) q[r] = base + (u32)qh_rank(m);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (lane == l) s_tf[wv][p] = base + (u32)__builtin_popcountll(m);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          todo &= ~m;
        }
      }
      // counts -> first positions: lane l owns parts 4 l .. 4 l + 3
      u32 c[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { const u32 p = (u32)lane * 4u + (u32)k; c[k] = p < np ? s_tf[wv][p] : 0u; }
      my_cnt = c[0] + c[1] + c[2] + c[3];
      const u32 incl = wave_incl_scan_u32(my_cnt);
      total = qh_readlane32(incl, 63);
      u32 at = incl - my_cnt;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 4; ++k) { const u32 p = (u32)lane * 4u + (u32)k; if (p < np) s_tf[wv][p] = at; at += c[k]; }
      // (the counts are needed again below, per part: keep them where s_cur is advanced)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (id[r] != 0xFFu) { pos[r] = s_tf[wv][id[r]] + q[r]; s_dst[wv][pos[r]] = s_cur[wv][id[r]] + q[r]; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 4; ++k) { const u32 p = (u32)lane * 4u + (u32)k; if (p < np) s_cur[wv][p] += c[k]; }
    }
    if (NPT > 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (id[r] != 0xFFu) { pos[r] = s_tf[wv][id[r]] + q[r]; s_dst[wv][pos[r]] = s_cur[wv][id[r]] + q[r]; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane < NPT) s_cur[wv][lane] += my_cnt;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (total) {   // wave-uniform
      for (u32 cix = 0; cix < A.n_cols; ++cix) {
        const PartCol col = A.cols[cix];
        switch (col.width) {
          case 1: qh_scatter_column<u8>(col, tb, last, lane, id, pos, total, s_val[wv], s_dst[wv]); break;
          case 2: qh_scatter_column<u16>(col, tb, last, lane, id, pos, total, s_val[wv], s_dst[wv]); break;
          case 4: qh_scatter_column<u32>(col, tb, last, lane, id, pos, total, s_val[wv], s_dst[wv]); break;
          case 8: qh_scatter_column<u64>(col, tb, last, lane, id, pos, total, s_val[wv], s_dst[wv]); break;
          default: qh_scatter_column<qh_v4u>(col, tb, last, lane, id, pos, total, s_val[wv], s_dst[wv]); break;
        }
      }
    }
  }
}
// the exchange's metadata words that only the DEVICE knows when the collective starts (qhip_shuffle_tables): rows per part from
// the parts' first positions, and whether a hash join of deferred size below (its status block in a page-locked slot the device
// can read) met duplicate build keys / had too little room — a flag every rank will see and act on together
__global__ void k_shuffle_meta(const u32* starts, u32 n_parts, long long* rows_out) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_parts) rows_out[k] = (long long)(starts[k + 1] - starts[k]);
}
__global__ void k_pending_flags(PendingSlots P, long long* flag) {
  if (blockIdx.x || threadIdx.x) return;
  long long f = 0;
  for (u32 k = 0; k < P.n; ++k) {
    const volatile u32* sl = (const volatile u32*)P.slot[k];
    if (sl[QS_MAXCOUNT] > 1u || sl[QS_OVERFLOW] || (u64)sl[2 * QS_WORDS] > P.cap[k]) f = 1;
  }
  *flag |= f;   // (bit 0 of the metadata row's flag word; the host-known bits were copied in ahead of this launch)
}
// out[k] = in[k * stride] (the parts' first positions out of the scanned histogram: what the host reads back)
__global__ void k_gather_stride_u32(const u32* in, u32 stride, u32 n, u32* out) {
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = in[(size_t)k * stride];
}

// ================================================================ aggregate output assembly on the device
// GroupAccumulator::output (physical/plan/aggregate/hash.rs:89-107) + Accumulator::evaluate for every group: one
// wavefront per 64 dense slots, one pass per output column (data-driven by FinCol descriptors, so no JIT is involved).
// Used when there are too many groups to finish on the host (Q3: ~10^5 groups at SF10).
__device__ __forceinline__ double qh_ord_to_f64(u64 k) { return qh_ord_f64(k); }

struct FinColsArg { FinCol c[kFinColsByValue]; };
__global__ __launch_bounds__(QH_BLOCK) void k_agg_finalize(const u64* dense, u32 G_cap, const u32* g_dev, int slot_words, int null_mask_word,
                                                          FinColsArg byval, const FinCol* cols_dev, int ncols, u32* null_counts, u32* status) {
  const FinCol* cols = cols_dev ? cols_dev : byval.c;
  // g_dev: the group count is still on the device (speculative launch behind the compaction); never beyond the capacity
  const u32 G = g_dev ? (*g_dev < G_cap ? *g_dev : G_cap) : G_cap;
  const u64 nwords = ((u64)G + 63) / 64;
  const u64 wave_global = ((u64)blockIdx.x * QH_BLOCK + threadIdx.x) >> 6;
  const u64 nwaves = ((u64)gridDim.x * QH_BLOCK) >> 6;
  const int lane = qh_lane();
  for (u64 j = wave_global; j < nwords; j += nwaves) {
    const u64 g = j * 64 + lane;
    const bool live = g < G;
    const u64* slot = dense + (size_t)(live ? g : 0) * slot_words;
    const u64 nullmask = null_mask_word ? slot[1] : 0ULL;
    for (int c = 0; c < ncols; ++c) {
      const FinCol fc = cols[c];
      bool valid = live;
      if (fc.cnt_word >= 0) valid = valid && slot[fc.cnt_word] != 0;
      if (fc.key_index >= 0) valid = valid && !((nullmask >> fc.key_index) & 1);
      const u64 w0 = slot[fc.src_word >= 0 ? fc.src_word : 0];
      const u64 w1 = slot[(fc.src_word >= 0 ? fc.src_word : 0) + 1 < slot_words ? (fc.src_word >= 0 ? fc.src_word : 0) + 1 : 0];
      u64 lo = 0, hi = 0;
      switch (fc.kind) {
        case F_KEY_FIXED: lo = w0; break;
        case F_KEY_DEC: lo = w0; hi = w1; break;
        case F_KEY_UTF8_LEN: lo = valid ? (slot[fc.src_word + fc.pad - 1] >> 56) : 0ULL; break;   // pad = words of the packed key
        case F_SUM64: lo = w0; break;
        case F_SUM128: lo = w0; hi = w1; break;
        case F_COUNT: lo = slot[fc.cnt_word]; valid = live; break;
        case F_AVG_F64: { const double s = qh_f64(w0); const u64 n = slot[fc.cnt_word]; lo = (u64)__double_as_longlong(n ? s / (double)n : 0.0); break; }
        case F_AVG_DEC: {
          // avg.rs:91-116: (sum * 10^(s_out - s)) checked, precision check on the scaled sum, truncating division
          const i128 sum = qh_mk128(w0, (i64)w1), mul = qh_mk128(fc.mul_lo, (i64)fc.mul_hi), lim = qh_mk128(fc.lim_lo, (i64)fc.lim_hi);
          i128 value = 0;
          const bool of = __builtin_mul_overflow(sum, mul, &value);
          const u64 n = slot[fc.cnt_word];
          if (valid && (of || value >= lim || value <= -lim)) atomicOr(&status[QS_ARITH_OVERFLOW], 1u);
          const i128 q = n ? value / (i128)n : (i128)0;
          lo = (u64)(u128)q; hi = (u64)((u128)q >> 64);
          break;
        }
        case F_MM_INT: {
          const u64 o = fc.is_min ? ~w0 : w0;
          lo = fc.is_signed ? (o ^ 0x8000000000000000ULL) : o;
          // no non-null value seen (all-zero cell): the seed of the column's own type (i32::MAX, not i64::MAX truncated)
          if (w0 == 0 && fc.is_signed && fc.width < 8) lo = fc.is_min ? ((1ULL << (8 * fc.width - 1)) - 1) : (1ULL << (8 * fc.width - 1));
          valid = live;
          break;
        }
        case F_MM_F64:
        case F_MM_F32: {
          // PrimitiveAccumulator seeded with NATIVE::MAX / MIN and PartialOrd merging (aggregate/mod.rs:60-84, min.rs:12-28)
          const double big = fc.kind == F_MM_F32 ? 3.4028234663852886e38 : 1.7976931348623157e308;
          double v = w0 == 0 ? (fc.is_min ? big : -big) : qh_ord_to_f64(fc.is_min ? ~w0 : w0);
          if (v != v) v = fc.is_min ? big : -big;
          if (fc.is_min && v > big) v = big;
          if (!fc.is_min && v < -big) v = -big;
          if (fc.kind == F_MM_F32) { const float f = (float)v; lo = (u64)__float_as_uint(f); } else lo = (u64)__double_as_longlong(v);
          valid = live;
          break;
        }
        case F_MM_DEC: { u128 o = ((u128)w1 << 64) | w0; if (fc.is_min) o = ~o; o ^= (u128)1 << 127; lo = (u64)o; hi = (u64)(o >> 64); valid = live; break; }
      }
      if (live) {
        u8* out = (u8*)fc.out_values;
        switch (fc.width) {
          case 1: ((u8*)out)[g] = (u8)lo; break;
          case 2: ((u16*)out)[g] = (u16)lo; break;
          case 4: ((u32*)out)[g] = (u32)lo; break;
          case 8: ((u64*)out)[g] = lo; break;
          default: ((u64*)out)[2 * g] = lo; ((u64*)out)[2 * g + 1] = hi; break;
        }
      }
      const u64 vb = qh_ballot(valid);
      if (lane == 0) {
        fc.out_valid[j] = vb;
        const u32 live_n = (u32)(((u64)G - j * 64) < 64 ? ((u64)G - j * 64) : 64);
        const u32 nulls = live_n - (u32)__builtin_popcountll(vb);
        if (nulls) atomicAdd(&null_counts[c], nulls);
      }
    }
  }
}
// bytes of packed (<= 7 byte) Utf8 group keys: word = bytes | len << 56
__global__ __launch_bounds__(QH_BLOCK) void k_agg_utf8_key_bytes(const u64* dense, u32 G, int slot_words, int src_word, const u32* offsets, u8* data) {
  for (u64 g = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; g < G; g += (u64)gridDim.x * QH_BLOCK) {
    const u64* w = dense + (size_t)g * slot_words + src_word;
    const u32 o = offsets[g], len = offsets[g + 1] - o;
    for (u32 b = 0; b < len; ++b) data[o + b] = (u8)(w[b >> 3] >> (8 * (b & 7)));
  }
}

// ================================================================ host launchers
void launch_select_indices(const uint64_t* mask, const uint32_t* wave_offset, uint64_t nrows, uint32_t* sel, hipStream_t s) {
  if (!nrows) return;
  hipLaunchKernelGGL(k_select_indices, dim3(grid_for((nrows + 63) / 64 * 64)), dim3(QH_BLOCK), 0, s, (const u64*)mask, (const u32*)wave_offset,
                     (u64)nrows, (u32*)sel);
}
void launch_mask_prefix_at(const uint64_t* mask, const uint32_t* wave_offset, const uint64_t* rows, uint32_t n, uint64_t nrows, uint32_t total,
                           uint32_t* out, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(k_mask_prefix_at, dim3((n + 255) / 256), dim3(256), 0, s, (const u64*)mask, (const u32*)wave_offset, (const u64*)rows, n,
                     (u64)nrows, total, (u32*)out);
}
void launch_mask_from_bits(const uint32_t* bits, uint64_t nrows, int want_set, uint64_t* mask, uint32_t* wave_count, hipStream_t s) {
  if (!nrows) return;
  hipLaunchKernelGGL(k_mask_from_bits, dim3(grid_for((nrows + 63) / 64)), dim3(QH_BLOCK), 0, s, (const u32*)bits, (u64)nrows, want_set, (u64*)mask,
                     (u32*)wave_count);
}
void launch_gather_fixed(const void* in, const uint32_t* idx, void* out, uint64_t m, int width, hipStream_t s) {
  if (!m) return;
  const dim3 g(grid_for(m)), b(QH_BLOCK);
  switch (width) {
    case 1: hipLaunchKernelGGL(k_gather_fixed<u8>, g, b, 0, s, (const u8*)in, (const u32*)idx, (u8*)out, (u64)m); break;
    case 2: hipLaunchKernelGGL(k_gather_fixed<u16>, g, b, 0, s, (const u16*)in, (const u32*)idx, (u16*)out, (u64)m); break;
    case 4: hipLaunchKernelGGL(k_gather_fixed<u32>, g, b, 0, s, (const u32*)in, (const u32*)idx, (u32*)out, (u64)m); break;
    case 8: hipLaunchKernelGGL(k_gather_fixed<u64>, g, b, 0, s, (const u64*)in, (const u32*)idx, (u64*)out, (u64)m); break;
    default: hipLaunchKernelGGL(k_gather_fixed<u128>, g, b, 0, s, (const u128*)in, (const u32*)idx, (u128*)out, (u64)m); break;
  }
}
void launch_gather_multi(const GatherBatch& b, int n, hipStream_t s) {
  uint64_t m = 0;
  for (int k = 0; k < n; ++k) m = std::max<uint64_t>(m, b.d[k].m);
  if (!m || n <= 0) return;
  hipLaunchKernelGGL(k_gather_multi, dim3(grid_for(m), (unsigned)n), dim3(QH_BLOCK), 0, s, b);
}
void launch_compact_fixed(const void* in, const uint64_t* mask, const uint32_t* wave_offset, void* out, uint64_t nrows, int width, hipStream_t s) {
  if (!nrows) return;
  const dim3 g(grid_for((nrows + 63) / 64 * 64, QH_BLOCK * 4, 8192)), b(QH_BLOCK);
  switch (width) {
    case 1: hipLaunchKernelGGL(k_compact_fixed<u8>, g, b, 0, s, (const u8*)in, (const u64*)mask, (const u32*)wave_offset, (u8*)out, (u64)nrows); break;
    case 2: hipLaunchKernelGGL(k_compact_fixed<u16>, g, b, 0, s, (const u16*)in, (const u64*)mask, (const u32*)wave_offset, (u16*)out, (u64)nrows); break;
    case 4: hipLaunchKernelGGL(k_compact_fixed<u32>, g, b, 0, s, (const u32*)in, (const u64*)mask, (const u32*)wave_offset, (u32*)out, (u64)nrows); break;
    case 8: hipLaunchKernelGGL(k_compact_fixed<u64>, g, b, 0, s, (const u64*)in, (const u64*)mask, (const u32*)wave_offset, (u64*)out, (u64)nrows); break;
    default: hipLaunchKernelGGL(k_compact_fixed<u128>, g, b, 0, s, (const u128*)in, (const u64*)mask, (const u32*)wave_offset, (u128*)out, (u64)nrows); break;
  }
}
void launch_gather_bits(const uint8_t* bitmap, const uint32_t* idx, uint64_t m, uint64_t* out_words, uint32_t* set_count, hipStream_t s) {
  if (!m) return;
  hipLaunchKernelGGL(k_gather_bits, dim3(grid_for((m + 63) / 64 * 64)), dim3(QH_BLOCK), 0, s, (const u8*)bitmap, (const u32*)idx, (u64)m,
                     (u64*)out_words, (u32*)set_count);
}
void launch_gather_utf8_lengths(const int32_t* offsets, const uint32_t* idx, uint64_t m, uint32_t* out_len, hipStream_t s) {
  if (!m) return;
  hipLaunchKernelGGL(k_gather_utf8_lengths, dim3(grid_for(m)), dim3(QH_BLOCK), 0, s, (const int*)offsets, (const u32*)idx, (u64)m, (u32*)out_len);
}
void launch_gather_utf8_bytes(const int32_t* offsets, const uint8_t* data, const uint32_t* idx, uint64_t m, const uint32_t* out_off,
                              uint8_t* out_data, hipStream_t s) {
  if (!m) return;
  hipLaunchKernelGGL(k_gather_utf8_bytes, dim3(grid_for((m + 63) / 64 * 64)), dim3(QH_BLOCK), 0, s, (const int*)offsets, (const u8*)data,
                     (const u32*)idx, (u64)m, (const u32*)out_off, (u8*)out_data);
}
void launch_utf8_max_len(const int32_t* offsets, uint64_t n, uint32_t* out, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_utf8_max_len, dim3(grid_for(n, QH_BLOCK, 1024)), dim3(QH_BLOCK), 0, s, (const int*)offsets, (u64)n, (u32*)out);
}
void launch_value_maxabs(const void* values, uint64_t n, int words, uint64_t* out, hipStream_t s, uint32_t* narrow32) {
  if (!n) return;
  const dim3 g(grid_for(n, QH_BLOCK * 8, 2048)), b(QH_BLOCK);
  if (words == 2) hipLaunchKernelGGL(k_value_maxabs<2>, g, b, 0, s, (const u64*)values, (u64)n, (u64*)out, (u32*)narrow32);
  else hipLaunchKernelGGL(k_value_maxabs<1>, g, b, 0, s, (const u64*)values, (u64)n, (u64*)out, (u32*)narrow32);
}
// field `offset` of `stride`-byte records <- a contiguous array of 4- / 8-byte values (ColRange::rec_buf)
template <class T>
__global__ __launch_bounds__(QH_BLOCK) void k_pack_field(const T* __restrict__ in, u8* __restrict__ out, u64 n, u32 stride, u32 offset) {
  for (u64 i = (u64)blockIdx.x * QH_BLOCK + threadIdx.x; i < n; i += (u64)gridDim.x * QH_BLOCK)
    *(T*)(out + i * stride + offset) = __builtin_nontemporal_load(&in[i]);
}
void launch_pack_field(const void* values, uint64_t n, int width, void* records, uint32_t stride, uint32_t offset, hipStream_t s) {
  if (!n) return;
  const dim3 g(grid_for(n, QH_BLOCK * 4, 4096)), b(QH_BLOCK);
  if (width == 4) hipLaunchKernelGGL(k_pack_field<u32>, g, b, 0, s, (const u32*)values, (u8*)records, (u64)n, (u32)stride, (u32)offset);
  else hipLaunchKernelGGL(k_pack_field<u64>, g, b, 0, s, (const u64*)values, (u8*)records, (u64)n, (u32)stride, (u32)offset);
}
void launch_narrow_decimal(const void* values, uint64_t n, int bytes, void* out, hipStream_t s, int src_words) {
  if (!n) return;
  const dim3 g(grid_for(n, QH_BLOCK * 8, 2048)), b(QH_BLOCK);
  if (src_words == 1) hipLaunchKernelGGL((k_narrow_decimal<u32, 1>), g, b, 0, s, (const u64*)values, (u64)n, (u32*)out);   // Int64 -> 4 bytes
  else if (bytes == 4) hipLaunchKernelGGL((k_narrow_decimal<u32, 2>), g, b, 0, s, (const u64*)values, (u64)n, (u32*)out);
  else hipLaunchKernelGGL((k_narrow_decimal<u64, 2>), g, b, 0, s, (const u64*)values, (u64)n, (u64*)out);
}
void launch_value_range(const void* values, uint64_t n, int width, bool is_signed, uint64_t* out, hipStream_t s) {
  if (!n) return;
  const dim3 g(grid_for(n, QH_BLOCK * 8, 2048)), b(QH_BLOCK);
  switch (width * 2 + (is_signed ? 1 : 0)) {
    case 2: hipLaunchKernelGGL(k_value_range<unsigned char>, g, b, 0, s, (const unsigned char*)values, (u64)n, (u64*)out); break;
    case 3: hipLaunchKernelGGL(k_value_range<signed char>, g, b, 0, s, (const signed char*)values, (u64)n, (u64*)out); break;
    case 4: hipLaunchKernelGGL(k_value_range<unsigned short>, g, b, 0, s, (const unsigned short*)values, (u64)n, (u64*)out); break;
    case 5: hipLaunchKernelGGL(k_value_range<short>, g, b, 0, s, (const short*)values, (u64)n, (u64*)out); break;
    case 8: hipLaunchKernelGGL(k_value_range<unsigned int>, g, b, 0, s, (const unsigned int*)values, (u64)n, (u64*)out); break;
    case 9: hipLaunchKernelGGL(k_value_range<int>, g, b, 0, s, (const int*)values, (u64)n, (u64*)out); break;
    default: hipLaunchKernelGGL(k_value_range<i64>, g, b, 0, s, (const i64*)values, (u64)n, (u64*)out); break;   // (UInt64 keys are not accepted by create_hashes)
  }
}
void launch_store_u32(uint32_t* p, uint32_t v, hipStream_t s) { hipLaunchKernelGGL(k_store_u32, dim3(1), dim3(1), 0, s, (u32*)p, v); }
void launch_iota_u32(uint32_t* out, uint64_t n, hipStream_t s, uint32_t first) {
  if (n) hipLaunchKernelGGL(k_iota_u32, dim3(grid_for(n)), dim3(QH_BLOCK), 0, s, (u32*)out, (u64)n, first);
}
void launch_iota_stride_u32(uint32_t* out, uint64_t n, uint32_t stride, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_iota_stride_u32, dim3(grid_for(n)), dim3(QH_BLOCK), 0, s, (u32*)out, (u64)n, stride);
}
void launch_fill_u32(uint32_t* out, uint64_t n, uint32_t v, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(n)), dim3(QH_BLOCK), 0, s, (u32*)out, (u64)n, v);
}
void launch_lookup_u32(const uint32_t* off, const uint64_t* rows, uint32_t n, uint64_t nrows, uint32_t total, uint32_t* out, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_lookup_u32, dim3((n + 255) / 256), dim3(256), 0, s, (const u32*)off, (const u64*)rows, n, (u64)nrows, total, (u32*)out);
}

#define DISPATCH_W(W, CALL)                \
  switch (W) {                             \
    case 1: { constexpr int KW = 1; CALL; } break; \
    case 2: { constexpr int KW = 2; CALL; } break; \
    case 3: { constexpr int KW = 3; CALL; } break; \
    case 4: { constexpr int KW = 4; CALL; } break; \
    case 5: { constexpr int KW = 5; CALL; } break; \
    case 6: { constexpr int KW = 6; CALL; } break; \
    case 7: { constexpr int KW = 7; CALL; } break; \
    default: { constexpr int KW = 8; CALL; } break; \
  }

void launch_join_build_insert(int W, const uint64_t* keys, const uint64_t* keyvalid, uint64_t n, uint64_t* table, uint32_t nslots,
                              uint32_t* row_slot, uint32_t* extra, uint64_t* bloom, uint32_t bloom_mask, uint32_t* status, hipStream_t s) {
  if (!n) return;
  DISPATCH_W(W, hipLaunchKernelGGL(k_join_build_insert<KW>, dim3(grid_for(n)), dim3(QH_BLOCK), 0, s, (const u64*)keys, (const u64*)keyvalid, (u64)n,
                                   (u64*)table, nslots, (u32*)row_slot, (u32*)extra, (u64*)bloom, bloom_mask, (u32*)status));
}
void launch_join_region_build(int W, const uint64_t* entries, const uint32_t* first, uint32_t n_wgs, uint32_t rows_per_wg, uint64_t* table,
                              uint64_t* bloom, uint32_t n_regions, uint32_t slot_bits, uint32_t bword_bits, uint32_t* status, hipStream_t s) {
  if (!n_regions) return;
  const size_t lds = ((size_t)8 * (1 + (size_t)W) << slot_bits) + ((size_t)8 << bword_bits);
  DISPATCH_W(W, hipLaunchKernelGGL(k_join_region_build<KW>, dim3(8 * ((n_regions + 7) / 8)), dim3(QH_REGION_BLOCK), lds, s, (const u64*)entries, (const u32*)first,
                                   n_wgs, n_regions, rows_per_wg, (u64*)table, (u64*)bloom, slot_bits, bword_bits, (u32*)status,
                                   (u32)((getenv("QHIP_REGION_DBG") && n_regions < 1450 && n_regions > 1400) ? atoi(getenv("QHIP_REGION_DBG")) : 0)));
}
void launch_join_full_counts(int W, const uint64_t* table, uint32_t nslots, uint32_t* count, hipStream_t s) {
  DISPATCH_W(W, hipLaunchKernelGGL(k_join_full_counts<KW>, dim3(grid_for(nslots)), dim3(QH_BLOCK), 0, s, (const u64*)table, nslots, (u32*)count));
}
void launch_add_i32(int32_t* p, uint64_t n, int32_t delta, hipStream_t s) {
  if (n && delta) hipLaunchKernelGGL(k_add_i32, dim3(grid_for(n, QH_BLOCK * 8, 1024)), dim3(QH_BLOCK), 0, s, (int*)p, (u64)n, (int)delta);
}
void launch_add_i32_batched(int32_t* p, uint64_t n, const uint64_t* starts, const int32_t* shifts, uint32_t nb, hipStream_t s) {
  if (n && nb) hipLaunchKernelGGL(k_add_i32_batched, dim3(grid_for(n, QH_BLOCK * 4, 2048)), dim3(QH_BLOCK), 0, s, (int*)p, (u64)n, (const u64*)starts, (const int*)shifts, nb);
}
void launch_bits_append(uint32_t* dst, uint64_t dst_pos, const uint8_t* src, uint64_t nbits, hipStream_t s) {
  if (!nbits) return;
  const uint64_t words = ((dst_pos + nbits + 31) >> 5) - (dst_pos >> 5);
  hipLaunchKernelGGL(k_bits_append, dim3((unsigned)((words + QH_BLOCK - 1) / QH_BLOCK)), dim3(QH_BLOCK), 0, s, (u32*)dst, (u64)dst_pos, (const unsigned char*)src, (u64)nbits);
}
void launch_pair_indices(uint32_t* minor, uint32_t* major, uint64_t n, uint32_t n_minor, uint32_t minor0, uint32_t major0, int, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_pair_indices, dim3(grid_for(n)), dim3(QH_BLOCK), 0, s, (u32*)minor, (u32*)major, (u64)n, n_minor, minor0, major0);
}
void launch_count_bits(const uint64_t* words, uint64_t nbits, uint32_t* out, hipStream_t s) {
  if (nbits) hipLaunchKernelGGL(k_count_bits, dim3(grid_for((nbits + 63) / 64, QH_BLOCK, 256)), dim3(QH_BLOCK), 0, s, (const u64*)words, (u64)nbits, (u32*)out);
}
void launch_sort_gather_img(const uint64_t* img, const uint32_t* idx, uint64_t n, uint64_t flip, uint64_t* out, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_sort_gather_img, dim3(grid_for(n)), dim3(QH_BLOCK), 0, s, (const u64*)img, (const u32*)idx, (u64)n, (u64)flip, (u64*)out);
}
void launch_sort_gather_valid(const uint64_t* validwords, const uint32_t* idx, uint64_t n, bool nulls_first, uint64_t* out, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_sort_gather_valid, dim3(grid_for(n)), dim3(QH_BLOCK), 0, s, (const u64*)validwords, (const u32*)idx, (u64)n,
                            nulls_first ? 1u : 0u, (u64*)out);
}
void launch_sort_utf8_chunk(const int32_t* offsets, const uint8_t* data, const uint8_t* validity, const uint32_t* idx, uint64_t n, int chunk,
                            uint64_t flip, uint64_t* out, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_sort_utf8_chunk, dim3(grid_for(n)), dim3(QH_BLOCK), 0, s, (const int*)offsets, (const u8*)data, (const u8*)validity,
                            (const u32*)idx, (u64)n, chunk, (u64)flip, (u64*)out);
}
void launch_gather_u32_nullable(const uint32_t* inner, const uint32_t* idx, uint32_t* out, uint64_t m, hipStream_t s) {
  if (m) hipLaunchKernelGGL(k_gather_u32_nullable, dim3(grid_for(m)), dim3(QH_BLOCK), 0, s, (const u32*)inner, (const u32*)idx, (u32*)out, (u64)m);
}
void launch_lower_bound_u32(const uint32_t* a, uint64_t m, const uint32_t* m_dev, const uint64_t* bound, uint32_t nb, uint32_t* pos, hipStream_t s) {
  if (nb) hipLaunchKernelGGL(k_lower_bound_u32, dim3((nb + QH_BLOCK - 1) / QH_BLOCK), dim3(QH_BLOCK), 0, s, (const u32*)a, (u64)m, (const u32*)m_dev, (const u64*)bound, nb, (u32*)pos);
}
void launch_bytes_to_bits(const uint8_t* bytes, uint32_t gen, uint32_t* bits, uint32_t nwords, uint32_t* counters, uint32_t* status, hipStream_t s) {
  if (!nwords) return;
  hipLaunchKernelGGL(k_bytes_to_bits, dim3((nwords + QH_BLOCK - 1) / QH_BLOCK), dim3(QH_BLOCK), 0, s, (const u8*)bytes, (u32)gen, (u32*)bits, (u32)nwords,
                     (u32*)counters, (u32*)status);
}
void launch_join_emit(const uint32_t* ent_slot, const uint32_t* ent_row, const uint32_t* chunk_nent, const uint32_t* chunk_off, const uint32_t* count,
                      const uint32_t* start, const uint32_t* rows, const uint32_t* row_of, uint64_t nchunks, uint64_t chunk_rows, uint32_t* b_idx, uint32_t* p_idx,
                      uint32_t* pair_off, uint32_t* cnt_out, uint32_t* visited, uint32_t cap, const uint32_t* stat_block, uint32_t* publish,
                      uint32_t* rows_out, hipStream_t s, int64_t* key_out, uint64_t key_min) {
  if (!nchunks) return;
  hipLaunchKernelGGL(k_join_emit, dim3(grid_for(nchunks * 64, QH_BLOCK)), dim3(QH_BLOCK), 0, s, (const u32*)ent_slot, (const u32*)ent_row,
                     (const u32*)chunk_nent, (const u32*)chunk_off, (const u32*)count, (const u32*)start, (const u32*)rows, (const u32*)row_of, (u64)nchunks,
                     (u64)chunk_rows, (u32*)b_idx, (u32*)p_idx, (u32*)pair_off, (u32*)cnt_out, (u32*)visited, (u32)cap, (const u32*)stat_block,
                     (u32*)publish, (u32*)rows_out, (i64*)key_out, (u64)key_min);
}
void launch_join_mark(const uint32_t* b_idx, const uint32_t* p_idx, uint64_t m, uint32_t* visited_bits, uint32_t* cnt_per_probe, hipStream_t s) {
  if (!m) return;
  hipLaunchKernelGGL(k_join_mark, dim3(grid_for(m)), dim3(QH_BLOCK), 0, s, (const u32*)b_idx, (const u32*)p_idx, (u64)m, (u32*)visited_bits,
                     (u32*)cnt_per_probe);
}
void launch_join_out_counts(const uint32_t* cnt, uint64_t np, uint32_t* out_cnt, hipStream_t s) {
  if (np) hipLaunchKernelGGL(k_join_out_counts, dim3(grid_for(np)), dim3(QH_BLOCK), 0, s, (const u32*)cnt, (u64)np, (u32*)out_cnt);
}
void launch_join_adjust_right(const uint32_t* b_in, const uint32_t* cnt, const uint32_t* in_off, const uint32_t* out_off, uint64_t np,
                              uint32_t* b_out, uint32_t* p_out, hipStream_t s) {
  if (np) hipLaunchKernelGGL(k_join_adjust_right, dim3(grid_for(np)), dim3(QH_BLOCK), 0, s, (const u32*)b_in, (const u32*)cnt, (const u32*)in_off,
                             (const u32*)out_off, (u64)np, (u32*)b_out, (u32*)p_out);
}
void launch_part_scatter(const PartScatterArgs& a, hipStream_t s) {
  if (!a.n_units || !a.n_cols) return;
  const dim3 g((a.n_units + QH_BLOCK / 64 - 1) / (QH_BLOCK / 64)), b(QH_BLOCK);
  if (a.n_parts <= 8) hipLaunchKernelGGL(k_part_scatter<8>, g, b, 0, s, a);
  else if (a.n_parts <= 16) hipLaunchKernelGGL(k_part_scatter<16>, g, b, 0, s, a);
  else hipLaunchKernelGGL(k_part_scatter<0>, g, b, 0, s, a);
}
void launch_shuffle_meta(const uint32_t* starts, uint32_t n_parts, int64_t* rows_out, hipStream_t s) {
  hipLaunchKernelGGL(k_shuffle_meta, dim3((n_parts + 255) / 256), dim3(256), 0, s, (const u32*)starts, n_parts, (long long*)rows_out);
}
void launch_pending_flags(const PendingSlots& p, int64_t* flag, hipStream_t s) {
  hipLaunchKernelGGL(k_pending_flags, dim3(1), dim3(64), 0, s, p, (long long*)flag);
}
void launch_gather_stride_u32(const uint32_t* in, uint32_t stride, uint32_t n, uint32_t* out, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_gather_stride_u32, dim3((n + 255) / 256), dim3(256), 0, s, (const u32*)in, stride, n, (u32*)out);
}
void launch_partition_ids(int W, const uint64_t* keys, uint64_t n, uint32_t nparts, uint32_t* part, uint32_t* hist, hipStream_t s) {
  if (!n) return;
  DISPATCH_W(W, hipLaunchKernelGGL(k_partition_ids<KW>, dim3(grid_for(n, QH_BLOCK * 16, 1024)), dim3(QH_BLOCK), 0, s, (const u64*)keys, (u64)n, nparts, (u32*)part, (u32*)hist));
}

void launch_agg_finalize(const uint64_t* dense, uint32_t G_cap, const uint32_t* g_dev, int slot_words, int null_mask_word, const FinCol* cols_host,
                         const FinCol* cols_dev, int ncols, uint32_t* null_counts, uint32_t* status, hipStream_t s) {
  if (!G_cap) return;
  FinColsArg byval;
  memset(&byval, 0, sizeof byval);
  if (!cols_dev) for (int k = 0; k < ncols && k < kFinColsByValue; ++k) byval.c[k] = cols_host[k];
  hipLaunchKernelGGL(k_agg_finalize, dim3(grid_for(((uint64_t)G_cap + 63) / 64 * 64)), dim3(QH_BLOCK), 0, s, (const u64*)dense, G_cap, (const u32*)g_dev,
                     slot_words, null_mask_word, byval, cols_dev, ncols, (u32*)null_counts, (u32*)status);
}
void launch_agg_utf8_key_bytes(const uint64_t* dense, uint32_t G, int slot_words, int src_word, const uint32_t* offsets, uint8_t* data,
                               hipStream_t s) {
  if (!G) return;
  hipLaunchKernelGGL(k_agg_utf8_key_bytes, dim3(grid_for(G)), dim3(QH_BLOCK), 0, s, (const u64*)dense, G, slot_words, src_word,
                     (const u32*)offsets, (u8*)data);
}

// stable sort of (key, value) pairs on the low `bits` bits of the key (rocPRIM LSD radix sort): groups build rows by
// hash-table slot / rows by partition id while keeping ascending row order inside a group
void stable_sort_pairs_u32(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, uint64_t n, int bits,
                           hipStream_t s) {
  if (!n) return;
  size_t tmp_bytes = 0;
  rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned)bits, s);
  DevBuf tmp(tmp_bytes);
  rocprim::radix_sort_pairs(tmp.ptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned)bits, s);
}

void stable_sort_pairs_u64(const uint64_t* keys_in, uint64_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, uint64_t n, int bits,
                           hipStream_t s) {
  if (!n) return;
  size_t tmp_bytes = 0;
  rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned)bits, s);
  DevBuf tmp(tmp_bytes);
  rocprim::radix_sort_pairs(tmp.ptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned)bits, s);
}

}  // namespace qhip
