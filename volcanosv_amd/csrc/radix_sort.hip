// radix_sort.hip — stable LSD radix sort of (u64 key, u32 value) pairs, item count on the device.
//
// Replaces sort_sig (Large_INDEL/extract_contig_signature_Hifi.py:170-179: np.argsort over positions)
// with the canonical STABLE order (SURVEY.md §7 hard part 1): ties keep input order. The same sort
// "radix-buckets" signatures by (tid, hap, type, pos): the list id sits in the key's high bits.
//
// 8-11 bit digits (fewest passes for the key width), 3 kernels per pass (tile histogram -> exclusive scan -> stable scatter). Tiles are
// 4 waves x 16 rounds x 64 items; inside a wave the stable rank of an item among equal digits is
// popcount(match_mask & lanes_below) where match_mask comes from 8 wave64 ballots, plus a per-wave LDS
// running counter across rounds — no atomics, fully deterministic.
#include "vsv_device.h"

namespace {

// Tiles are RS_WAVES waves x ROUNDS rounds x 64 items. ROUNDS = 16 (4096 items) amortises the per-tile histogram
// traffic on large inputs; ROUNDS = 4 (1024 items) spreads a ~100 k-row table over ~100 CUs instead of ~25, which is
// what bounds the scatter on the signature tables of a read-shaped run (n ~ 1 % of the records).
constexpr int RS_WAVES = 4;
constexpr int RS_ROUNDS_BIG = 16, RS_ROUNDS_SMALL = 4;
template <int ROUNDS> constexpr uint32_t rs_tile() { return RS_WAVES * ROUNDS * 64; }

template <int BITS>
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid) {
  // mask of lanes (valid ones) holding the same BITS-bit digit
  uint64_t m = __ballot(valid);
#pragma unroll
  for (int b = 0; b < BITS; ++b) {
    const uint64_t bal = __ballot((d >> b) & 1u);
    m &= ((d >> b) & 1u) ? bal : ~bal;
  }
  return m;
}

template <int ROUNDS> __device__ __forceinline__ uint32_t n_tiles_of(uint32_t n) { return (n + rs_tile<ROUNDS>() - 1) / rs_tile<ROUNDS>(); }

// hist[tile * BINS + d] = number of items of this tile with digit d
template <int BITS, int ROUNDS>
__global__ __launch_bounds__(256) void rs_hist(const uint64_t* __restrict__ key, const uint32_t* __restrict__ d_n, int shift,
                                               uint32_t* __restrict__ hist, uint32_t* __restrict__ totals) {
  constexpr int BINS = 1 << BITS;
  __shared__ uint32_t cnt[BINS];
  const uint32_t n = *d_n, ntiles = n_tiles_of<ROUNDS>(n);
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (int d = threadIdx.x; d < BINS; d += 256) cnt[d] = 0;
    __syncthreads();
    const uint32_t base = tile * rs_tile<ROUNDS>();
    for (int k = 0; k < (int)rs_tile<ROUNDS>() / 256; ++k) {
      const uint32_t i = base + k * 256 + threadIdx.x;
      if (i < n) atomicAdd(&cnt[(uint32_t)(key[i] >> shift) & (BINS - 1)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < BINS; d += 256) {
      const uint32_t c = cnt[d];
      hist[(size_t)tile * BINS + d] = c;
      if (totals && c) atomicAdd(&totals[d], c);   // per-digit totals of the whole input (zeroed at run start)
    }
    __syncthreads();
  }
}

// single-block exclusive scan of the [tile][digit] histogram in (digit, tile) order: thread t owns DPT digits, walks
// the tiles (coalesced rows) for the per-digit totals, one block-wide scan over the BINS totals, second walk writes
// the offsets.
template <int BITS, int ROUNDS>
__global__ __launch_bounds__(1024) void rs_scan(uint32_t* __restrict__ hist, const uint32_t* __restrict__ d_n) {
  constexpr int BINS = 1 << BITS;
  constexpr int DPT = BINS >= 1024 ? BINS / 1024 : 1;
  __shared__ uint32_t sh[1024];
  const uint32_t ntiles = n_tiles_of<ROUNDS>(*d_n);
  const int t = threadIdx.x;
  const bool active = t * DPT < BINS;
  uint32_t tot[DPT];
#pragma unroll
  for (int k = 0; k < DPT; ++k) tot[k] = 0;
  if (active) {
#pragma unroll 16
    for (uint32_t tile = 0; tile < ntiles; ++tile)
#pragma unroll
      for (int k = 0; k < DPT; ++k) tot[k] += hist[(size_t)tile * BINS + t * DPT + k];
  }
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < DPT; ++k) mine += tot[k];
  sh[t] = mine;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const uint32_t v = t >= d ? sh[t - d] : 0;
    __syncthreads();
    sh[t] += v;
    __syncthreads();
  }
  uint32_t run = sh[t] - mine;   // exclusive base of this thread's first digit
  if (active) {
    uint32_t base[DPT];
#pragma unroll
    for (int k = 0; k < DPT; ++k) { base[k] = run; run += tot[k]; }
#pragma unroll 16
    for (uint32_t tile = 0; tile < ntiles; ++tile)
#pragma unroll
      for (int k = 0; k < DPT; ++k) {
        const size_t idx = (size_t)tile * BINS + t * DPT + k;
        const uint32_t c = hist[idx];
        hist[idx] = base[k];
        base[k] += c;
      }
  }
}

// Multi-block scan (used whenever the hist kernel produced the digit totals): a block owns SCAN_DG digits x SCAN_TG tile
// groups (1024 threads). Digit base = exclusive scan of the totals (digits before the block + in-block scan); thread
// (g, d) sums digit d over its tile group, the group sums are exchanged through LDS, and a second sweep over the same
// (L2-hot) column entries writes the offsets. The sequential depth is ntiles / SCAN_TG loads in batches of 16; the
// single-group version was the slowest kernel of a pass once the tiles became small (~100 tiles: 14 us -> see DESIGN).
constexpr int SCAN_DG = 128, SCAN_TG = 8;
template <int BITS, int ROUNDS>
__global__ __launch_bounds__(1024) void rs_scan_mb(uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals,
                                                   const uint32_t* __restrict__ d_n) {
  constexpr int BINS = 1 << BITS;
  __shared__ uint32_t red[1024];
  __shared__ uint32_t dig[SCAN_DG];
  __shared__ uint32_t part[SCAN_TG][SCAN_DG];
  const uint32_t ntiles = n_tiles_of<ROUNDS>(*d_n);
  const int t = threadIdx.x, dl = t & (SCAN_DG - 1), g = t >> 7;
  const int d0 = blockIdx.x * SCAN_DG, d = d0 + dl;
  uint32_t before = 0;
  for (int i = t; i < d0; i += 1024) before += totals[i];
  red[t] = before;
  __syncthreads();
  for (int k = 512; k > 0; k >>= 1) { if (t < k) red[t] += red[t + k]; __syncthreads(); }
  const uint32_t prev = red[0];
  const uint32_t mine = (t < SCAN_DG && d < BINS) ? totals[d] : 0;
  if (t < SCAN_DG) dig[t] = mine;
  __syncthreads();
  for (int k = 1; k < SCAN_DG; k <<= 1) {
    const uint32_t v = (t < SCAN_DG && t >= k) ? dig[t - k] : 0;
    __syncthreads();
    if (t < SCAN_DG) dig[t] += v;
    __syncthreads();
  }
  if (t < SCAN_DG) dig[t] = prev + dig[t] - mine;        // exclusive digit base
  const uint32_t per = (ntiles + SCAN_TG - 1) / SCAN_TG;
  const uint32_t t0 = (uint32_t)g * per, t1 = min(ntiles, t0 + per);
  uint32_t sum = 0;
  if (d < BINS) {
    for (uint32_t tb = t0; tb < t1; tb += 16) {
      uint32_t c[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) c[k] = (tb + k < t1) ? hist[(size_t)(tb + k) * BINS + d] : 0;
#pragma unroll
      for (int k = 0; k < 16; ++k) sum += c[k];
    }
  }
  part[g][dl] = sum;
  __syncthreads();
  if (d >= BINS) return;
  uint32_t run = dig[dl];
  for (int gg = 0; gg < g; ++gg) run += part[gg][dl];
  for (uint32_t tb = t0; tb < t1; tb += 16) {
    uint32_t c[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) c[k] = (tb + k < t1) ? hist[(size_t)(tb + k) * BINS + d] : 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (tb + k < t1) hist[(size_t)(tb + k) * BINS + d] = run;
      run += c[k];
    }
  }
}

template <int BITS, int RS_ROUNDS>
__global__ __launch_bounds__(256) void rs_scatter(const uint64_t* __restrict__ key, const uint32_t* __restrict__ val,
                                                  const uint32_t* __restrict__ d_n, int shift,
                                                  const uint32_t* __restrict__ hist, uint64_t* __restrict__ key_out,
                                                  uint32_t* __restrict__ val_out, const uint32_t* __restrict__ totals) {
  constexpr int BINS = 1 << BITS;
  __shared__ uint32_t wcnt[RS_WAVES][BINS];   // per-wave running digit counters, then (wave, digit) bases
  (void)totals;
  constexpr uint32_t RS_TILE = rs_tile<RS_ROUNDS>();
  const uint32_t n = *d_n, ntiles = n_tiles_of<RS_ROUNDS>(n);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t lt = (1ull << lane) - 1ull;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (int d = threadIdx.x; d < BINS; d += 256)
#pragma unroll
      for (int w = 0; w < RS_WAVES; ++w) wcnt[w][d] = 0;
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
      const uint32_t d = (uint32_t)(k_[r] >> shift) & (BINS - 1);
      const uint64_t m = match_digit<BITS>(d, ok);
      // every lane of the match group reads the counter, then the group's lowest lane bumps it;
      // LDS operations of one wave execute in order.
      const uint32_t old = ok ? wcnt[wv][d] : 0;
      __builtin_amdgcn_wave_barrier();
      if (ok && (m & lt) == 0) wcnt[wv][d] = old + (uint32_t)__popcll(m);
      __builtin_amdgcn_wave_barrier();
      rk[r] = ok ? old + (uint32_t)__popcll(m & lt) : 0xFFFFFFFFu;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < BINS; d += 256) {  // exclusive prefix of the digit's count over the waves + global base
      uint32_t run = hist[(size_t)tile * BINS + d];
#pragma unroll
      for (int w = 0; w < RS_WAVES; ++w) { const uint32_t c = wcnt[w][d]; wcnt[w][d] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
      if (rk[r] != 0xFFFFFFFFu) {
        const uint32_t d = (uint32_t)(k_[r] >> shift) & (BINS - 1);
        const uint32_t dst = wcnt[wv][d] + rk[r];
        key_out[dst] = k_[r];
        val_out[dst] = v_[r];
      }
    }
    __syncthreads();
  }
}

template <int BITS, int ROUNDS>
void one_pass(hipStream_t st, int64_t max_n, const uint64_t* kin, const uint32_t* vin, const uint32_t* d_n, int shift, uint32_t* hist,
              uint64_t* kout, uint32_t* vout, uint32_t* totals) {
  const int64_t max_tiles = (max_n + rs_tile<ROUNDS>() - 1) / rs_tile<ROUNDS>();
  const int grid = (int)(max_tiles < 1024 ? (max_tiles < 1 ? 1 : max_tiles) : 1024);
  rs_hist<BITS, ROUNDS><<<grid, 256, 0, st>>>(kin, d_n, shift, hist, totals);
  if (totals) rs_scan_mb<BITS, ROUNDS><<<(1 << BITS) / SCAN_DG, 1024, 0, st>>>(hist, totals, d_n);
  else rs_scan<BITS, ROUNDS><<<1, 1024, 0, st>>>(hist, d_n);
  rs_scatter<BITS, ROUNDS><<<grid, 256, 0, st>>>(kin, vin, d_n, shift, hist, kout, vout, totals);
}

template <int ROUNDS>
SortResult sort_pairs(hipStream_t st, uint64_t* key, uint32_t* val, uint64_t* key_scratch, uint32_t* val_scratch,
                      const uint32_t* d_n, int64_t max_n, int nbits, const SortWork& w) {
  uint64_t* kin = key; uint32_t* vin = val;
  uint64_t* kout = key_scratch; uint32_t* vout = val_scratch;
  if (nbits < 1) nbits = 1;
  const int passes = (nbits + 10) / 11;
  int shift = 0;
  for (int p = 0; p < passes; ++p) {
    const int left = nbits - shift;
    const int bits = (left + (passes - p) - 1) / (passes - p);   // spread the bits evenly over the remaining passes
    // the multi-block scan needs a zeroed 2048-entry totals slot per pass (SortWork::totals, zeroed once per run)
    uint32_t* totals = nullptr;
    if (w.totals && *w.pass_cursor < w.max_passes) { totals = w.totals + (size_t)(*w.pass_cursor) * 2048; ++*w.pass_cursor; }
    switch (bits <= 8 ? 8 : bits) {
      case 8: one_pass<8, ROUNDS>(st, max_n, kin, vin, d_n, shift, w.hist, kout, vout, totals); shift += 8; break;
      case 9: one_pass<9, ROUNDS>(st, max_n, kin, vin, d_n, shift, w.hist, kout, vout, totals); shift += 9; break;
      case 10: one_pass<10, ROUNDS>(st, max_n, kin, vin, d_n, shift, w.hist, kout, vout, totals); shift += 10; break;
      default: one_pass<11, ROUNDS>(st, max_n, kin, vin, d_n, shift, w.hist, kout, vout, totals); shift += 11; break;
    }
    uint64_t* tk = kin; kin = kout; kout = tk;
    uint32_t* tv = vin; vin = vout; vout = tv;
  }
  return SortResult{kin, vin};   // the sorted pairs live in (key,val) or in the scratch pair, no copy back
}

}  // namespace

// Digit widths are chosen per sort so that the pass count is minimal with digits of at most 11 bits
// (e.g. 33 key bits -> 3 passes of 11; 26 bits -> 3 passes of 9). SortWork::small picks the tile size (both exact).
SortResult vsv_radix_sort_pairs(hipStream_t st, uint64_t* key, uint32_t* val, uint64_t* key_scratch, uint32_t* val_scratch,
                                const uint32_t* d_n, int64_t max_n, int nbits, const SortWork& w) {
  return w.small_tiles ? sort_pairs<RS_ROUNDS_SMALL>(st, key, val, key_scratch, val_scratch, d_n, max_n, nbits, w)
                 : sort_pairs<RS_ROUNDS_BIG>(st, key, val, key_scratch, val_scratch, d_n, max_n, nbits, w);
}

int64_t vsv_radix_hist_entries(int64_t max_n) { return 2048 * ((max_n + rs_tile<RS_ROUNDS_SMALL>() - 1) / rs_tile<RS_ROUNDS_SMALL>()); }
