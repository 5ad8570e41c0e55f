// radix_sort.hip — stable LSD radix sort of (u64 key, u32 value) pairs, item count on the device.
//
// Replaces sort_sig (Large_INDEL/extract_contig_signature_Hifi.py:170-179: np.argsort over positions)
// with the canonical STABLE order (SURVEY.md §7 hard part 1): ties keep input order. The same sort
// "radix-buckets" signatures by (tid, hap, type, pos): the list id sits in the key's high bits.
//
// 8-bit digits, 3 kernels per pass (tile histogram -> exclusive scan -> stable scatter). Tiles are
// 4 waves x 16 rounds x 64 items; inside a wave the stable rank of an item among equal digits is
// popcount(match_mask & lanes_below) where match_mask comes from 8 wave64 ballots, plus a per-wave LDS
// running counter across rounds — no atomics, fully deterministic.
#include "vsv_device.h"

namespace {

constexpr int RS_ROUNDS = 16;
constexpr int RS_WAVES = 4;
constexpr int RS_TILE = RS_WAVES * RS_ROUNDS * 64;  // 4096 items per tile

__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid) {
  // mask of lanes (valid ones) holding the same 8-bit digit
  uint64_t m = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const uint64_t bal = __ballot((d >> b) & 1u);
    m &= ((d >> b) & 1u) ? bal : ~bal;
  }
  return m;
}

__device__ __forceinline__ uint32_t n_tiles_of(uint32_t n) { return (n + RS_TILE - 1) / RS_TILE; }

// hist[d * ntiles + tile] = number of items of this tile with digit d
__global__ __launch_bounds__(256) void rs_hist(const uint64_t* __restrict__ key, const uint32_t* __restrict__ d_n, int shift,
                                               uint32_t* __restrict__ hist) {
  __shared__ uint32_t cnt[256];
  const uint32_t n = *d_n, ntiles = n_tiles_of(n);
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = tile * RS_TILE;
    for (int k = 0; k < RS_TILE / 256; ++k) {
      const uint32_t i = base + k * 256 + threadIdx.x;
      if (i < n) atomicAdd(&cnt[(uint32_t)(key[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * ntiles + tile] = cnt[threadIdx.x];
    __syncthreads();
  }
}

// single-block exclusive scan over the 256*ntiles histogram entries
__global__ __launch_bounds__(1024) void rs_scan(uint32_t* __restrict__ hist, const uint32_t* __restrict__ d_n) {
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry;
  const uint32_t total = 256u * n_tiles_of(*d_n);
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (uint32_t base = 0; base < total; base += 4096) {
    // each thread owns 4 consecutive entries
    const uint32_t i0 = base + threadIdx.x * 4;
    uint32_t v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = (i0 + k < total) ? hist[i0 + k] : 0; s += v[k]; }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const uint32_t t = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    const uint32_t c = carry;
    uint32_t run = c + sh[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (i0 + k < total) hist[i0 + k] = run; run += v[k]; }
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + sh[1023];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void rs_scatter(const uint64_t* __restrict__ key, const uint32_t* __restrict__ val,
                                                  const uint32_t* __restrict__ d_n, int shift,
                                                  const uint32_t* __restrict__ hist, uint64_t* __restrict__ key_out,
                                                  uint32_t* __restrict__ val_out) {
  __shared__ uint32_t wcnt[RS_WAVES][256];   // per-wave running digit counters
  __shared__ uint32_t gbase[256];            // global base of (digit, tile)
  const uint32_t n = *d_n, ntiles = n_tiles_of(n);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t lt = (1ull << lane) - 1ull;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) wcnt[w][threadIdx.x] = 0;
    gbase[threadIdx.x] = hist[threadIdx.x * ntiles + tile];
    __syncthreads();
    const uint32_t wbase = tile * RS_TILE + wv * (RS_ROUNDS * 64);
    uint64_t k_[RS_ROUNDS];
    uint32_t v_[RS_ROUNDS], rk[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
      const uint32_t i = wbase + r * 64 + lane;
      const bool ok = i < n;
      k_[r] = ok ? key[i] : 0;
      v_[r] = ok ? val[i] : 0;
      const uint32_t d = (uint32_t)(k_[r] >> shift) & 255u;
      const uint64_t m = match_digit(d, ok);
      // every lane of the match group reads the counter, then the group's lowest lane bumps it;
      // LDS operations of one wave execute in order.
      const uint32_t old = ok ? wcnt[wv][d] : 0;
      __builtin_amdgcn_wave_barrier();
      if (ok && (m & lt) == 0) wcnt[wv][d] = old + (uint32_t)__popcll(m);
      __builtin_amdgcn_wave_barrier();
      rk[r] = ok ? old + (uint32_t)__popcll(m & lt) : 0xFFFFFFFFu;
    }
    __syncthreads();
    {  // digit t: exclusive prefix of its count over the waves, folded into the global base
      uint32_t run = gbase[threadIdx.x];
#pragma unroll
      for (int w = 0; w < RS_WAVES; ++w) { const uint32_t c = wcnt[w][threadIdx.x]; wcnt[w][threadIdx.x] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
      if (rk[r] != 0xFFFFFFFFu) {
        const uint32_t d = (uint32_t)(k_[r] >> shift) & 255u;
        const uint32_t dst = wcnt[wv][d] + rk[r];
        key_out[dst] = k_[r];
        val_out[dst] = v_[r];
      }
    }
    __syncthreads();
  }
}

__global__ void rs_copy(const uint64_t* __restrict__ k_in, const uint32_t* __restrict__ v_in, const uint32_t* __restrict__ d_n,
                        uint64_t* __restrict__ k_out, uint32_t* __restrict__ v_out) {
  const uint32_t n = *d_n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { k_out[i] = k_in[i]; v_out[i] = v_in[i]; }
}

}  // namespace

void vsv_radix_sort_pairs(hipStream_t st, uint64_t* key, uint32_t* val, const uint32_t* d_n, int64_t max_n, int nbits,
                          const SortWork& w) {
  int64_t max_tiles = (max_n + RS_TILE - 1) / RS_TILE;
  int grid = (int)(max_tiles < 1024 ? (max_tiles < 1 ? 1 : max_tiles) : 1024);
  uint64_t* kin = key; uint32_t* vin = val;
  uint64_t* kout = w.key_alt; uint32_t* vout = w.val_alt;
  for (int shift = 0; shift < nbits; shift += 8) {
    rs_hist<<<grid, 256, 0, st>>>(kin, d_n, shift, w.hist);
    rs_scan<<<1, 1024, 0, st>>>(w.hist, d_n);
    rs_scatter<<<grid, 256, 0, st>>>(kin, vin, d_n, shift, w.hist, kout, vout);
    uint64_t* tk = kin; kin = kout; kout = tk;
    uint32_t* tv = vin; vin = vout; vout = tv;
  }
  if (kin != key) rs_copy<<<256, 256, 0, st>>>(kin, vin, d_n, key, val);
}

int64_t vsv_radix_hist_entries(int64_t max_n) { return 256 * ((max_n + RS_TILE - 1) / RS_TILE); }
