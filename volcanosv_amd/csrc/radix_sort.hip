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
#include <stdlib.h>

#include <type_traits>

#include "vsv_device.h"

namespace {

// Tiles are RS_WAVES waves x ROUNDS rounds x 64 items. ROUNDS = 16 (4096 items) amortises the per-tile histogram
// traffic on large inputs; ROUNDS = 4 (1024 items) spreads a ~100 k-row table over ~100 CUs instead of ~25, which is
// what bounds the scatter on the signature tables of a read-shaped run (n ~ 1 % of the records).
constexpr int RS_WAVES = 4;
constexpr int RS_ROUNDS_BIG = 16, RS_ROUNDS_SMALL = 4;
template <int ROUNDS> constexpr uint32_t rs_tile() { return RS_WAVES * ROUNDS * 64; }
constexpr uint32_t RS_GROUP = 16;       // tiles per group of the self-scanning bucket pass (rs_scatter<..., SELF>)

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

// digit of a key: LSD passes take BITS bits at `shift`; the bucket pass (below) maps the whole key range [0, kmax) linearly onto
// BINS - 1 buckets (monotone, so the buckets are contiguous key ranges) and the all-ones "dead" key onto the last one
struct LsdDigit {
  int shift;
  template <int BITS> __device__ __forceinline__ uint32_t get(uint64_t k) const { return (uint32_t)(k >> shift) & ((1u << BITS) - 1u); }
};
struct BucketDigit {
  int s;          // k >> s fits 32 bits
  uint32_t M;     // bucket = umulhi(k >> s, M) < BINS - 1 for k < kmax
  template <int BITS> __device__ __forceinline__ uint32_t get(uint64_t k) const {
    if (k == VSV_KEY_DEAD) return (1u << BITS) - 1u;
    const uint64_t kk = k >> s;
    const uint32_t b = __umulhi(kk > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)kk, M);
    return b < (1u << BITS) - 2u ? b : (1u << BITS) - 2u;
  }
};

// hist[tile * BINS + d] = number of items of this tile with digit d
// where the (key, value) pairs of a pass come from: arrays, or straight from the rows of a signature / call table (the keys of
// the bucket pass are then never materialised before the scatter: no key-building launch)
struct KeyArr {
  const uint64_t* key; const uint32_t* val;
  __device__ __forceinline__ uint64_t key_at(uint32_t i, bool) const { return key[i]; }
  __device__ __forceinline__ uint32_t val_at(uint32_t i) const { return val[i]; }
  const uint32_t* val_lut() const { return val; }
};
struct SigKey {      // stage key of a signature row (vsv_key_stage; stage 5 = the READS final order, reads.py:281-286)
  const vsv_sig* rows; int stage, pb, tid_lo, tid_bits; uint32_t* err;
  __device__ __forceinline__ uint64_t key_at(uint32_t i, bool check) const {
    const vsv_sig v = rows[i];
    uint64_t k;
    if (stage == 5) {
      k = (v.meta & VSV_M_DEAD) ? VSV_KEY_DEAD
          : ((uint64_t)(uint32_t)(v.tid - tid_lo) << (pb + 2)) | (vsv_kpos(v.pos) << 2) | ((v.meta & VSV_M_SPLIT) ? 2u : 0u) | ((v.meta & VSV_M_DEL) ? 0u : 1u);
    } else k = vsv_key_stage(v, stage, pb, tid_lo);
    if (check && !(v.meta & VSV_M_DEAD) && ((pb < 32 && (vsv_kpos(v.pos) >> pb) != 0) || ((uint32_t)(v.tid - tid_lo) >> tid_bits) != 0))
      atomicOr(err, ERRB_RANGE);            // max_pos hint too small / tid outside [tid_lo, n_tids)
    return k;
  }
  __device__ __forceinline__ uint32_t val_at(uint32_t i) const { return i; }
  const uint32_t* val_lut() const { return nullptr; }
};
// (tid, pos) key of a call row: the final order of pair_sig's output (H:594). The hp1 rows' calls were written by the pairing
// kernels; an hp2 row's call is derived here from the merged table and the pairing state — unpaired -> a 0/1 call of its own
// (H:588-592), paired -> dead — so the pass that used to write those rows and all keys (pair_finish) is not launched.
struct CallKey {
  const vsv_call* rows; int pb, tid_lo; const vsv_sig* m; const int32_t* st2;
  __device__ __forceinline__ uint64_t key_at(uint32_t i, bool) const {
    if (m) {
      const vsv_sig me = m[i];
      if (me.meta & VSV_M_HP2) return st2[i] == -1 ? vsv_key_stage(me, 4, pb, tid_lo) : VSV_KEY_DEAD;
    }
    return vsv_key_stage(rows[i].sig, 4, pb, tid_lo);
  }
  __device__ __forceinline__ uint32_t val_at(uint32_t i) const { return i; }
  const uint32_t* val_lut() const { return nullptr; }
};

// key of a split-pair slot, straight from the candidates sorted by (tid, hap, name): candidate j followed by another one of the same
// name heads a pair slot keyed (tid, hap, record of the name's first candidate); every other slot is dead (H:421-457). What the
// separate split_mark_pairs pass computed in front of the sort.
struct PairKey {
  const uint64_t* ckey; const uint32_t* crec; int qid_bits, rec_bits; const uint32_t* d_n;
  __device__ __forceinline__ uint64_t key_at(uint32_t j, bool) const {
    const uint32_t n = *d_n;
    const uint64_t k = ckey[j];
    if (j + 1 < n && ckey[j + 1] == k) {
      uint32_t g = j;
      while (g > 0 && ckey[g - 1] == k) --g;
      return ((k >> qid_bits) << rec_bits) | crec[g];
    }
    return VSV_KEY_DEAD;
  }
  __device__ __forceinline__ uint32_t val_at(uint32_t j) const { return j; }
  const uint32_t* val_lut() const { return nullptr; }
};

template <int BITS, int ROUNDS, typename DIGIT, typename SRC>
__global__ __launch_bounds__(256) void rs_hist(SRC src, const uint32_t* __restrict__ d_n, DIGIT dg,
                                               uint32_t* __restrict__ hist, uint32_t* __restrict__ totals, uint32_t* __restrict__ groups = nullptr) {
  constexpr int BINS = 1 << BITS;
  __shared__ uint32_t cnt[BINS];
  const uint32_t n = *d_n, ntiles = n_tiles_of<ROUNDS>(n);
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (int d = threadIdx.x; d < BINS; d += 256) cnt[d] = 0;
    __syncthreads();
    const uint32_t base = tile * rs_tile<ROUNDS>();
    for (int k = 0; k < (int)rs_tile<ROUNDS>() / 256; ++k) {
      const uint32_t i = base + k * 256 + threadIdx.x;
      if (i < n) atomicAdd(&cnt[dg.template get<BITS>(src.key_at(i, true))], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < BINS; d += 256) {
      const uint32_t c = cnt[d];
      hist[(size_t)tile * BINS + d] = c;
      if (totals && c) atomicAdd(&totals[d], c);   // per-digit totals of the whole input (zeroed at run start)
      // ... and of every group of RS_GROUP tiles (a table with more groups than the slot holds is refused by bk_lds_sort)
      if (groups && c && tile / RS_GROUP < (uint32_t)VSV_RS_MAX_GROUPS) atomicAdd(&groups[(size_t)(tile / RS_GROUP) * BINS + d], c);
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
// SCAN_DG digits x SCAN_TG tile groups = 1024 threads: 128 x 8 for the 11-bit LSD passes; the bucket pass (256-2048 buckets over
// a few hundred small tiles) takes 32 x 32, i.e. four times the blocks and a quarter of the sequential tile walk per thread.
template <int BITS, int ROUNDS, int SCAN_DG, int SCAN_TG>
__global__ __launch_bounds__(1024) void rs_scan_mb(uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals,
                                                   const uint32_t* __restrict__ d_n) {
  static_assert(SCAN_DG * SCAN_TG == 1024, "one thread per (digit, tile group)");
  constexpr int BINS = 1 << BITS;
  __shared__ uint32_t red[1024];
  __shared__ uint32_t dig[SCAN_DG];
  __shared__ uint32_t part[SCAN_TG][SCAN_DG];
  const uint32_t ntiles = n_tiles_of<ROUNDS>(*d_n);
  const int t = threadIdx.x, dl = t & (SCAN_DG - 1), g = t / SCAN_DG;
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

// SELF: `hist` still holds the raw per-tile counts (no scan launch ran): the block derives its tile's offsets itself — digit bases
// from an in-block exclusive scan of the per-digit totals, plus the counts in front of its tile: the sums of the whole groups of
// RS_GROUP tiles (accumulated by rs_hist) and the tiles of its own group — at most VSV_RS_MAX_GROUPS + RS_GROUP - 1 L2-hot rows of
// BINS entries, whatever the table's size (a walk over every earlier tile is bound by the CU's 64 B/clk L2 port: 4 us at 270 tiles).
template <int BITS, int RS_ROUNDS, typename DIGIT, typename SRC, bool SELF = false>
__global__ __launch_bounds__(256) void rs_scatter(SRC src,
                                                  const uint32_t* __restrict__ d_n, DIGIT dg,
                                                  const uint32_t* __restrict__ hist, uint64_t* __restrict__ key_out,
                                                  uint32_t* __restrict__ val_out, const uint32_t* __restrict__ totals,
                                                  const uint32_t* __restrict__ groups = nullptr) {
  constexpr int BINS = 1 << BITS;
  constexpr int DPT = BINS / 256;             // digits per thread in the self-scan (BINS >= 256)
  __shared__ uint32_t wcnt[RS_WAVES][BINS];   // per-wave running digit counters, then (wave, digit) bases
  __shared__ uint32_t dbase[SELF ? BINS : 1];
  __shared__ uint32_t wsum[RS_WAVES];
  constexpr uint32_t RS_TILE = rs_tile<RS_ROUNDS>();
  const uint32_t n = *d_n, ntiles = n_tiles_of<RS_ROUNDS>(n);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t lt = (1ull << lane) - 1ull;
  if (SELF && ntiles > (uint32_t)VSV_RS_MAX_GROUPS * RS_GROUP) return;   // more groups than the slot of group sums holds: see bk_lds_sort
  if (SELF) {                                  // exclusive scan of the digit totals: thread t owns digits [t * DPT, (t + 1) * DPT)
    uint32_t tt[DPT], mine = 0;
#pragma unroll
    for (int k = 0; k < DPT; ++k) { tt[k] = totals[threadIdx.x * DPT + k]; mine += tt[k]; }
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
    for (int w = 0; w < wv; ++w) run += wsum[w];
#pragma unroll
    for (int k = 0; k < DPT; ++k) { dbase[threadIdx.x * DPT + k] = run; run += tt[k]; }
    __syncthreads();
  }
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
      k_[r] = ok ? src.key_at(i, false) : 0;
      v_[r] = ok ? src.val_at(i) : 0;
      const uint32_t d = dg.template get<BITS>(k_[r]);
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
    if (SELF) {
      uint32_t col[DPT];
#pragma unroll
      for (int k = 0; k < DPT; ++k) col[k] = dbase[threadIdx.x * DPT + k];
      const uint32_t g_me = tile / RS_GROUP;
      const uint32_t* gp = groups + threadIdx.x * DPT;
      const uint32_t* hp = hist + (size_t)g_me * RS_GROUP * BINS + threadIdx.x * DPT;
      {                                            // the tiles of the own group in front of this one: at most RS_GROUP - 1 rows, all in flight
        const uint32_t in_group = tile - g_me * RS_GROUP;
        uint32_t c[RS_GROUP - 1][DPT];
#pragma unroll
        for (uint32_t u = 0; u < RS_GROUP - 1; ++u)
#pragma unroll
          for (int k = 0; k < DPT; ++k) c[u][k] = u < in_group ? hp[(size_t)u * BINS + k] : 0u;
#pragma unroll
        for (uint32_t u = 0; u < RS_GROUP - 1; ++u)
#pragma unroll
          for (int k = 0; k < DPT; ++k) col[k] += c[u][k];
      }
      uint32_t gb = 0;
      for (; gb + 8 <= g_me; gb += 8) {            // whole groups in front: eight rows in flight
        uint32_t c[8][DPT];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int k = 0; k < DPT; ++k) c[u][k] = gp[(size_t)(gb + u) * BINS + k];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int k = 0; k < DPT; ++k) col[k] += c[u][k];
      }
      for (; gb < g_me; ++gb)
#pragma unroll
        for (int k = 0; k < DPT; ++k) col[k] += gp[(size_t)gb * BINS + k];
#pragma unroll
      for (int k = 0; k < DPT; ++k) {
        const int d = threadIdx.x * DPT + k;
        uint32_t run = col[k];
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) { const uint32_t c = wcnt[w][d]; wcnt[w][d] = run; run += c; }
      }
    } else {
      for (int d = threadIdx.x; d < BINS; d += 256) {  // exclusive prefix of the digit's count over the waves + global base
        uint32_t run = hist[(size_t)tile * BINS + d];
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) { const uint32_t c = wcnt[w][d]; wcnt[w][d] = run; run += c; }
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
      if (rk[r] != 0xFFFFFFFFu) {
        const uint32_t d = dg.template get<BITS>(k_[r]);
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
              uint64_t* kout, uint32_t* vout, uint32_t* totals, uint64_t hint_rows) {
  const int64_t max_tiles = (max_n + rs_tile<ROUNDS>() - 1) / rs_tile<ROUNDS>();
  const int grid = (int)(max_tiles < 1024 ? (max_tiles < 1 ? 1 : max_tiles) : 1024);
  const LsdDigit dg{shift};
  rs_hist<BITS, ROUNDS, LsdDigit, KeyArr><<<grid, 256, 0, st>>>(KeyArr{kin, vin}, d_n, dg, hist, totals);
  // tables of hundreds of tiles and more (10^6-10^7 rows: config 3's candidates, the contig shape's raw signatures): 16 digits x 64
  // tile groups per block instead of 128 x 8 — eight times the blocks and an eighth of the sequential tile walk per thread (the
  // 128 x 8 form ran on 4-8 CUs for 0.5-0.9 ms per pass there)
  const bool wide = hint_rows >= 256ull * rs_tile<ROUNDS>();
  if (totals && wide) rs_scan_mb<BITS, ROUNDS, 16, 64><<<(1 << BITS) / 16, 1024, 0, st>>>(hist, totals, d_n);
  else if (totals) rs_scan_mb<BITS, ROUNDS, 128, 8><<<(1 << BITS) / 128, 1024, 0, st>>>(hist, totals, d_n);
  else rs_scan<BITS, ROUNDS><<<1, 1024, 0, st>>>(hist, d_n);
  rs_scatter<BITS, ROUNDS, LsdDigit, KeyArr, false><<<grid, 256, 0, st>>>(KeyArr{kin, vin}, d_n, dg, hist, kout, vout, totals);
}

// ---- bucket sort: ONE counting pass into <= 2048 key-range buckets, then one workgroup sorts each bucket in LDS -----------
// The signature tables of a chromosome hold 10^4-10^6 rows: an LSD sort of their 33-bit keys is 3-4 passes x 3 launches over a table
// that fits the chip's LDS many times. Here the keys are cut into BINS - 1 contiguous key ranges by a linear (monotone) map, the
// (stable) counting pass moves every row to its range, and a workgroup then sorts its range completely in LDS: 32-bit keys
// relative to the range's smallest key, stable LSD passes of 8 bits over the bits that actually differ (2-3 passes), ranks from
// wave ballots as in rs_scatter. 4 launches instead of 9-12. A range that does not fit (more than BK_CAP rows, or keys more than
// 2^32 apart) raises ERRB_SORT_FALLBACK: the caller runs the table again through the LSD passes (same result, the usual speed).
constexpr int BK_CAP = 4096;        // rows a workgroup sorts in LDS
// Workgroup of the LDS sort: 512 threads (two waves per SIMD, 9-bit digits) when the handle has the GPU to itself; 256 threads
// (one wave per SIMD like the CIGAR scan's workgroups, 8-bit digits, no register-resident copy of the bucket) when several
// handles share it (vsv_params.split_overlap = VSV_OVERLAP_OFF): a workgroup of 8 waves x 64 VGPRs finds no room on a CU whose
// SIMDs another engine's scan keeps refilling one wave at a time and waits for that scan's tail (86 us instead of 12 for the
// candidate sort). Same box, config 2: one engine 0.61-0.65 ms per step with 512 threads against 0.66-0.68 with 256; three
// engines 0.46-0.47 against 0.45.
__device__ __forceinline__ uint64_t match_digit_rt(uint32_t d, bool valid, int bits) {
  uint64_t m = __ballot(valid);
  for (int b = 0; b < bits; ++b) {
    const uint64_t bal = __ballot((d >> b) & 1u);
    m &= ((d >> b) & 1u) ? bal : ~bal;
  }
  return m;
}
// ROW = void: the sorted (key, value) pairs are the result. ROW = vsv_sig / vsv_call: the values index `rows_in` and the sorted ROWS
// are the result (plus their keys): rows_out[i] = rows_in[value_i] for the alive rows, *d_alive = their count, *n_long = 0 and,
// if given, fill[0, n) = -1 (the state array of the stage that follows) — the gather launch that used to follow a sort.
template <typename ROW>
struct RowIO {
  const ROW* in; ROW* out; uint32_t* d_alive; uint32_t* n_long; int32_t* fill;
  __device__ __forceinline__ ROW fetch(uint32_t v) const { return in[v]; }
};
template <>
struct RowIO<void> {};
template <>
struct RowIO<vsv_call> {                     // m != nullptr: the calls of hp2 rows are built on the fly (CallKey)
  const vsv_call* in; vsv_call* out; uint32_t* d_alive; uint32_t* n_long; int32_t* fill; const vsv_sig* m;
  __device__ __forceinline__ vsv_call fetch(uint32_t v) const {
    if (m) {
      const vsv_sig me = m[v];
      if (me.meta & VSV_M_HP2) { vsv_call c; c.sig = me; c.a = -1; c.b = (int32_t)v; c.gt = 1; c.pad = 0; return c; }   // (alive rows only: unpaired)
    }
    return in[v];
  }
};
template <typename ROW, int BK_THREADS>
__global__ __launch_bounds__(BK_THREADS) void bk_lds_sort(const uint64_t* __restrict__ key, const uint32_t* __restrict__ val,
                                                           const uint32_t* __restrict__ base, const uint32_t* __restrict__ totals, int nbuckets,
                                                           const uint32_t* __restrict__ d_n,
                                                           uint64_t* __restrict__ key_out, uint32_t* __restrict__ val_out, uint32_t* __restrict__ err, uint32_t cap,
                                                           uint32_t max_tile_rows, RowIO<ROW> io) {
  constexpr bool ROWS = !std::is_same<ROW, void>::value;
  constexpr int BK_WAVES = BK_THREADS / 64, BK_DBITS = BK_THREADS >= 512 ? 9 : 8, BK_PER = BK_THREADS >= 512 ? BK_CAP / BK_THREADS : 1;
  constexpr bool IN_REGS = BK_THREADS >= 512;     // the bucket waits in registers between the min / max reduction and the LDS fill
  __shared__ uint32_t sk[2][BK_CAP], sv[2][BK_CAP];
  __shared__ uint32_t wcnt[BK_WAVES][1 << BK_DBITS];
  __shared__ uint32_t tot[BK_WAVES];
  __shared__ unsigned long long s_min, s_max;
  __shared__ uint32_t s_lo;
  const uint32_t n = *d_n;
  const int b = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (totals && (n + max_tile_rows - 1) / max_tile_rows > (uint32_t)VSV_RS_MAX_GROUPS * RS_GROUP) {
    // the table outgrew the group sums of the self-scanning pass (sized from the previous run): nothing was scattered
    if (t == 0) atomicOr(err, ERRB_SORT_FALLBACK);
    return;
  }
  if (totals) {          // no scan launch ran (self-scanning scatter): the bucket's first row = the totals of the buckets in front of it
    uint32_t part = 0;
    for (int d = t; d < b; d += BK_THREADS) part += totals[d];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) part += (uint32_t)__shfl_xor((int)part, d, 64);
    if (t == 0) s_lo = 0;
    __syncthreads();
    if (lane == 0 && part) atomicAdd(&s_lo, part);
    __syncthreads();
  }
  {  // the last bucket holds the dead rows (all keys equal, nothing to sort): every workgroup moves a slice of it
    uint32_t dlo = totals ? n - min(n, totals[nbuckets - 1]) : base[nbuckets - 1];
    if (dlo > n) dlo = n;
    const uint32_t dm = n - dlo, per = (dm + gridDim.x - 1) / gridDim.x;
    const uint32_t a = dlo + min(dm, (uint32_t)b * per), e = dlo + min(dm, (uint32_t)(b + 1) * per);
    for (uint32_t i = a + t; i < e; i += BK_THREADS) { key_out[i] = VSV_KEY_DEAD; if (!ROWS) val_out[i] = val[i]; }
    if constexpr (ROWS) {
      if (b == 0 && t == 0) { *io.d_alive = dlo; *io.n_long = 0; }
      if (io.fill) {
        const uint32_t fper = (n + gridDim.x - 1) / gridDim.x;
        const uint32_t fa = min(n, (uint32_t)b * fper), fe = min(n, fa + fper);
        for (uint32_t i = fa + t; i < fe; i += BK_THREADS) io.fill[i] = -1;
      }
    }
  }
  if (b == nbuckets - 1) return;
  uint32_t lo = totals ? s_lo : base[b], hi = totals ? lo + totals[b] : base[b + 1];
  if (lo > n) lo = n;
  if (hi > n) hi = n;
  if (hi <= lo) return;
  const uint32_t m = hi - lo;
  if (m > cap) { if (t == 0) atomicOr(err, ERRB_SORT_FALLBACK); return; }
  if (t == 0) { s_min = ~0ull; s_max = 0ull; }
  __syncthreads();
  uint64_t kr[BK_PER];
  uint32_t vr[BK_PER];
  uint64_t kmin = ~0ull, kmax = 0;
  if constexpr (IN_REGS) {
#pragma unroll
    for (int j = 0; j < BK_PER; ++j) {
      const uint32_t i = (uint32_t)t + (uint32_t)j * BK_THREADS;
      kr[j] = i < m ? key[lo + i] : 0ull;
      vr[j] = i < m ? val[lo + i] : 0u;
    }
#pragma unroll
    for (int j = 0; j < BK_PER; ++j) {
      if ((uint32_t)t + (uint32_t)j * BK_THREADS < m) { kmin = kr[j] < kmin ? kr[j] : kmin; kmax = kr[j] > kmax ? kr[j] : kmax; }
    }
  } else {
    for (uint32_t i = (uint32_t)t; i < m; i += BK_THREADS) { const uint64_t k = key[lo + i]; kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax; }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const uint64_t a = __shfl_xor(kmin, d, 64), c = __shfl_xor(kmax, d, 64);
    kmin = a < kmin ? a : kmin; kmax = c > kmax ? c : kmax;
  }
  if (lane == 0) { atomicMin(&s_min, (unsigned long long)kmin); atomicMax(&s_max, (unsigned long long)kmax); }
  __syncthreads();
  kmin = s_min; kmax = s_max;
  const uint64_t width = kmax - kmin;
  if (width > 0xFFFFFFFFull) { if (t == 0) atomicOr(err, ERRB_SORT_FALLBACK); return; }
  if (m == 1 || width == 0) {                            // one row, or all keys equal: already in order
    for (uint32_t i = (uint32_t)t; i < m; i += BK_THREADS) {
      key_out[lo + i] = key[lo + i];
      if constexpr (ROWS) io.out[lo + i] = io.fetch(val[lo + i]); else val_out[lo + i] = val[lo + i];
    }
    return;
  }
  if constexpr (IN_REGS) {
#pragma unroll
    for (int j = 0; j < BK_PER; ++j) { const uint32_t i = (uint32_t)t + (uint32_t)j * BK_THREADS; if (i < m) { sk[0][i] = (uint32_t)(kr[j] - kmin); sv[0][i] = vr[j]; } }
  } else {
    for (uint32_t i = (uint32_t)t; i < m; i += BK_THREADS) { sk[0][i] = (uint32_t)(key[lo + i] - kmin); sv[0][i] = val[lo + i]; }
  }
  int src = 0;
  // stable LSD passes over the bits that differ inside the bucket, digits as equal as possible and at most BK_DBITS wide
  const int wbits = 64 - __builtin_clzll(width);
  const int passes = (wbits + BK_DBITS - 1) / BK_DBITS;
  const int db = (wbits + passes - 1) / passes;
  const uint32_t dmask = (1u << db) - 1u, nbins = 1u << db;
  // rows of wave w: [w * per, (w + 1) * per) — wave-major chunks keep the input order inside equal digits
  const uint32_t per = ((m + BK_WAVES - 1) / BK_WAVES + 63u) & ~63u;
  const uint32_t c0 = min(m, (uint32_t)wv * per), c1 = min(m, c0 + per);
  const uint64_t lt = (1ull << lane) - 1ull;
  for (int pass = 0, shift = 0; pass < passes; ++pass, shift += db) {
    for (uint32_t d = t; d < BK_WAVES * nbins; d += BK_THREADS) wcnt[d / nbins][d % nbins] = 0;
    __syncthreads();
    for (uint32_t i = c0 + lane; i < c1; i += 64) atomicAdd(&wcnt[wv][(sk[src][i] >> shift) & dmask], 1u);
    __syncthreads();
    // per-digit totals -> exclusive scan over the digits (thread t owns digit t: in-wave shuffle scan + wave sums) -> (wave, digit) bases
    uint32_t mine = 0, incl = 0;
    if ((uint32_t)t < nbins) {
#pragma unroll
      for (int w = 0; w < BK_WAVES; ++w) mine += wcnt[w][t];
    }
    incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) tot[wv] = incl;
    __syncthreads();
    if ((uint32_t)t < nbins) {
      uint32_t run = incl - mine;
      for (int w = 0; w < wv; ++w) run += tot[w];
#pragma unroll
      for (int w = 0; w < BK_WAVES; ++w) { const uint32_t c = wcnt[w][t]; wcnt[w][t] = run; run += c; }
    }
    __syncthreads();
    for (uint32_t i0 = c0; i0 < c1; i0 += 64) {            // whole waves iterate together (c0, c1 are wave-uniform)
      const uint32_t i = i0 + lane;
      const bool ok = i < c1;
      const uint32_t k = ok ? sk[src][i] : 0u, v = ok ? sv[src][i] : 0u;
      const uint32_t d = (k >> shift) & dmask;
      const uint64_t mm = match_digit_rt(d, ok, db);
      const uint32_t old = ok ? wcnt[wv][d] : 0u;
      __builtin_amdgcn_wave_barrier();
      if (ok && (mm & lt) == 0) wcnt[wv][d] = old + (uint32_t)__popcll(mm);
      __builtin_amdgcn_wave_barrier();
      if (ok) { const uint32_t dst = old + (uint32_t)__popcll(mm & lt); sk[src ^ 1][dst] = k; sv[src ^ 1][dst] = v; }
    }
    __syncthreads();
    src ^= 1;
  }
  for (uint32_t i = t; i < m; i += BK_THREADS) {
    key_out[lo + i] = kmin + sk[src][i];
    if constexpr (ROWS) io.out[lo + i] = io.fetch(sv[src][i]); else val_out[lo + i] = sv[src][i];
  }
}

// ---- the same bucket sort in TWO launches: no histogram launch in front of the counting pass ------------------------------------------
// Every bucket owns a fixed region of `caps` (key, input index) slots in the scratch pair. A tile ranks its rows per bucket as
// rs_scatter does, reserves room for each bucket it touches with ONE atomicAdd on the bucket's counter (the per-pass digit totals,
// zeroed with the run) and writes its rows there: a bucket's region is a sequence of per-tile chunks, each in input order, the chunks
// in the order the tiles happened to arrive. The LDS sort puts the chunks back into tile order first (stable passes on the tile number
// of the input index; skipped when the indices already ascend: position-clustered input feeds a bucket from one or two tiles), then
// sorts by key as before: the result is the stable sort, bit for bit. Dead rows (the last bucket) are only counted. A bucket with more
// rows than its region raises ERRB_SORT_FALLBACK like a bucket beyond LDS does.
template <int BITS, int RS_ROUNDS, typename DIGIT, typename SRC>
__global__ __launch_bounds__(256) void rs_slot_scatter(SRC src, const uint32_t* __restrict__ d_n, DIGIT dg, uint32_t* __restrict__ totals, uint32_t caps,
                                                       uint64_t* __restrict__ slot_key, uint32_t* __restrict__ slot_idx, uint32_t* __restrict__ slot_val,
                                                       uint32_t* __restrict__ err) {
  constexpr int BINS = 1 << BITS;
  __shared__ uint32_t wcnt[RS_WAVES][BINS];   // per-wave running digit counters, then (wave, digit) bases inside the bucket's region
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
    uint32_t rk[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
      const uint32_t i = wbase + r * 64 + lane;
      const bool ok = i < n;
      k_[r] = ok ? src.key_at(i, true) : 0;
      const uint32_t d = dg.template get<BITS>(k_[r]);
      const uint64_t m = match_digit<BITS>(d, ok);
      const uint32_t old = ok ? wcnt[wv][d] : 0;
      __builtin_amdgcn_wave_barrier();
      if (ok && (m & lt) == 0) wcnt[wv][d] = old + (uint32_t)__popcll(m);
      __builtin_amdgcn_wave_barrier();
      rk[r] = ok ? old + (uint32_t)__popcll(m & lt) : 0xFFFFFFFFu;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < BINS; d += 256) {      // room in the bucket's region for this tile's rows, then the waves' shares of it
      uint32_t c[RS_WAVES], tot = 0;
#pragma unroll
      for (int w = 0; w < RS_WAVES; ++w) { c[w] = wcnt[w][d]; tot += c[w]; }
      uint32_t run = tot ? atomicAdd(&totals[d], tot) : 0u;
      if (tot && d != BINS - 1 && run + tot > caps) atomicOr(err, ERRB_SORT_FALLBACK);
#pragma unroll
      for (int w = 0; w < RS_WAVES; ++w) { wcnt[w][d] = run; run += c[w]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
      if (rk[r] != 0xFFFFFFFFu) {
        const uint32_t d = dg.template get<BITS>(k_[r]);
        const uint32_t dst = wcnt[wv][d] + rk[r];
        if (d != BINS - 1 && dst < caps) {
          const uint32_t i = wbase + r * 64 + lane;
          const size_t at = (size_t)d * caps + dst;
          slot_key[at] = k_[r]; slot_idx[at] = i;
          if (slot_val) slot_val[at] = src.val_at(i);      // (values that are not the input index travel beside it: the sort may run in place)
        }
      }
    }
    __syncthreads();
  }
}

// slot_val: the values of the rows where they are not the input indices themselves (KeyArr), or nullptr
template <typename ROW, int BK_THREADS>
__global__ __launch_bounds__(BK_THREADS) void bk_slot_sort(const uint64_t* __restrict__ slot_key, const uint32_t* __restrict__ slot_idx, const uint32_t* __restrict__ slot_val, uint32_t caps,
                                                            const uint32_t* __restrict__ totals, int nbuckets, const uint32_t* __restrict__ d_n,
                                                            uint64_t* __restrict__ key_out, uint32_t* __restrict__ val_out, uint32_t* __restrict__ err, uint32_t cap,
                                                            uint32_t tile_rows, uint32_t tie_max, RowIO<ROW> io) {
  constexpr bool ROWS = !std::is_same<ROW, void>::value;
  constexpr int BK_WAVES = BK_THREADS / 64, BK_DBITS = BK_THREADS >= 512 ? 9 : 8;
  __shared__ uint32_t sk[2][BK_CAP];
  __shared__ uint16_t sv[2][BK_CAP];          // the row's place in the bucket's region
  __shared__ uint32_t wcnt[BK_WAVES][1 << BK_DBITS];
  __shared__ uint32_t tot[BK_WAVES];
  __shared__ unsigned long long s_min, s_max;
  __shared__ uint32_t s_lo, s_unordered;
  const uint32_t n = *d_n;
  const int b = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  {                      // the bucket's first output row = the totals of the buckets in front of it
    uint32_t part = 0;
    for (int d = t; d < b; d += BK_THREADS) part += totals[d];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) part += (uint32_t)__shfl_xor((int)part, d, 64);
    if (t == 0) { s_lo = 0; s_unordered = 0; s_min = ~0ull; s_max = 0ull; }
    __syncthreads();
    if (lane == 0 && part) atomicAdd(&s_lo, part);
    __syncthreads();
  }
  {  // the last bucket counts the dead rows: every workgroup writes a slice of the dead tail
    uint32_t dlo = n - min(n, totals[nbuckets - 1]);
    const uint32_t dm = n - dlo, per = (dm + gridDim.x - 1) / gridDim.x;
    const uint32_t a = dlo + min(dm, (uint32_t)b * per), e = dlo + min(dm, (uint32_t)(b + 1) * per);
    for (uint32_t i = a + t; i < e; i += BK_THREADS) { key_out[i] = VSV_KEY_DEAD; if (!ROWS) val_out[i] = 0; }
    if constexpr (ROWS) {
      if (b == 0 && t == 0) { *io.d_alive = dlo; *io.n_long = 0; }
      if (io.fill) {
        const uint32_t fper = (n + gridDim.x - 1) / gridDim.x;
        const uint32_t fa = min(n, (uint32_t)b * fper), fe = min(n, fa + fper);
        for (uint32_t i = fa + t; i < fe; i += BK_THREADS) io.fill[i] = -1;
      }
    }
  }
  if (b == nbuckets - 1) return;
  const uint32_t lo = min(n, s_lo), m_all = totals[b];
  if (m_all == 0) return;
  if (m_all > cap || m_all > caps || lo + m_all > n) { if (t == 0) atomicOr(err, ERRB_SORT_FALLBACK); return; }
  const uint32_t m = m_all;
  const uint64_t* kin = slot_key + (size_t)b * caps;
  const uint32_t* iin = slot_idx + (size_t)b * caps;
  uint64_t kmin = ~0ull, kmax = 0;
  for (uint32_t i = (uint32_t)t; i < m; i += BK_THREADS) { const uint64_t k = kin[i]; kmin = k < kmin ? k : kmin; kmax = k > kmax ? k : kmax; }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const uint64_t a = __shfl_xor(kmin, d, 64), c = __shfl_xor(kmax, d, 64);
    kmin = a < kmin ? a : kmin; kmax = c > kmax ? c : kmax;
  }
  if (lane == 0) { atomicMin(&s_min, (unsigned long long)kmin); atomicMax(&s_max, (unsigned long long)kmax); }
  __syncthreads();
  kmin = s_min; kmax = s_max;
  const uint64_t width = kmax - kmin;
  if (width > 0xFFFFFFFFull) { if (t == 0) atomicOr(err, ERRB_SORT_FALLBACK); return; }
  bool bad = false;                                      // input indices that do not ascend: chunks of several tiles, out of order
  for (uint32_t i = (uint32_t)t; i + 1 < m; i += BK_THREADS) if (iin[i + 1] < iin[i]) bad = true;
  if (bad) s_unordered = 1;
  for (uint32_t i = (uint32_t)t; i < m; i += BK_THREADS) { sk[0][i] = (uint32_t)(kin[i] - kmin); sv[0][i] = (uint16_t)i; }
  __syncthreads();
  const bool unordered = s_unordered != 0;               // (block-uniform)
  int src = 0;
  const uint32_t per = ((m + BK_WAVES - 1) / BK_WAVES + 63u) & ~63u;
  const uint32_t c0 = min(m, (uint32_t)wv * per), c1 = min(m, c0 + per);
  const uint64_t lt = (1ull << lane) - 1ull;
  // stable LSD passes over bits [0, nb) of sk (digits as equal as possible, at most BK_DBITS wide), (sk, sv) moving together; called
  // by the whole block
  auto lsd = [&](int nb) {
    const int passes = (nb + BK_DBITS - 1) / BK_DBITS;
    const int db = passes ? (nb + passes - 1) / passes : 0;
    const uint32_t dmask = (1u << db) - 1u, nbins = 1u << db;
    for (int pass = 0, shift = 0; pass < passes; ++pass, shift += db) {
      for (uint32_t d = t; d < BK_WAVES * nbins; d += BK_THREADS) wcnt[d / nbins][d % nbins] = 0;
      __syncthreads();
      for (uint32_t i = c0 + lane; i < c1; i += 64) atomicAdd(&wcnt[wv][(sk[src][i] >> shift) & dmask], 1u);
      __syncthreads();
      uint32_t mine = 0, incl = 0;
      if ((uint32_t)t < nbins) {
#pragma unroll
        for (int w = 0; w < BK_WAVES; ++w) mine += wcnt[w][t];
      }
      incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
      if (lane == 63) tot[wv] = incl;
      __syncthreads();
      if ((uint32_t)t < nbins) {
        uint32_t run = incl - mine;
        for (int w = 0; w < wv; ++w) run += tot[w];
#pragma unroll
        for (int w = 0; w < BK_WAVES; ++w) { const uint32_t c = wcnt[w][t]; wcnt[w][t] = run; run += c; }
      }
      __syncthreads();
      for (uint32_t i0 = c0; i0 < c1; i0 += 64) {            // whole waves iterate together (c0, c1 are wave-uniform)
        const uint32_t i = i0 + lane;
        const bool ok = i < c1;
        const uint32_t k = ok ? sk[src][i] : 0u;
        const uint16_t v = ok ? sv[src][i] : (uint16_t)0;
        const uint32_t d = (k >> shift) & dmask;
        const uint64_t mm = match_digit_rt(d, ok, db);
        const uint32_t old = ok ? wcnt[wv][d] : 0u;
        __builtin_amdgcn_wave_barrier();
        if (ok && (mm & lt) == 0) wcnt[wv][d] = old + (uint32_t)__popcll(mm);
        __builtin_amdgcn_wave_barrier();
        if (ok) { const uint32_t dst = old + (uint32_t)__popcll(mm & lt); sk[src ^ 1][dst] = k; sv[src ^ 1][dst] = v; }
      }
      __syncthreads();
      src ^= 1;
    }
  };
  const int wbits = width ? 64 - __builtin_clzll(width) : 0;
  lsd(wbits);
  if (unordered) {
    // The passes kept the order the rows had in the region: rows of equal key whose chunks arrived out of order are out of order still.
    // Every such run (<= tie_max = 64 rows: beyond that the whole bucket is sorted again, tile order first) is put into input order by rank.
    const uint32_t TIE_MAX = tie_max;
    __syncthreads();
    if (t == 0) s_unordered = 0;                           // (now: "a run too long for the ranks")
    __syncthreads();
    for (uint32_t i = (uint32_t)t; i < m; i += BK_THREADS) {
      const uint32_t k = sk[src][i];
      uint32_t dst = i;
      if ((i > 0 && sk[src][i - 1] == k) || (i + 1 < m && sk[src][i + 1] == k)) {
        uint32_t a = i, e = i + 1;
        while (a > 0 && sk[src][a - 1] == k && i - a < TIE_MAX) --a;
        while (e < m && sk[src][e] == k && e - i < TIE_MAX) ++e;
        if (e - a > TIE_MAX || (a > 0 && sk[src][a - 1] == k) || (e < m && sk[src][e] == k)) s_unordered = 1;
        else {
          const uint32_t me = iin[sv[src][i]];
          uint32_t r = 0;
          for (uint32_t x = a; x < e; ++x) r += iin[sv[src][x]] < me ? 1u : 0u;
          dst = a + r;
        }
      }
      sk[src ^ 1][i] = k;                                  // (keys inside a run are equal: a row's key stays where it is)
      sv[src ^ 1][dst] = sv[src][i];
    }
    __syncthreads();
    src ^= 1;
    if (s_unordered) {                                     // (block-uniform) a pile of equal keys: tile order first, then the keys again
      const uint32_t ntiles = (n + tile_rows - 1) / tile_rows;
      for (uint32_t i = (uint32_t)t; i < m; i += BK_THREADS) { sk[src][i] = iin[i] / tile_rows; sv[src][i] = (uint16_t)i; }
      __syncthreads();
      lsd(ntiles > 1 ? 32 - __builtin_clz(ntiles - 1) : 0);
      for (uint32_t i = (uint32_t)t; i < m; i += BK_THREADS) sk[src][i] = (uint32_t)(kin[sv[src][i]] - kmin);
      __syncthreads();
      lsd(wbits);
    }
  }
  const uint32_t* vin = slot_val ? slot_val + (size_t)b * caps : iin;
  for (uint32_t i = t; i < m; i += BK_THREADS) {
    key_out[lo + i] = kmin + sk[src][i];
    const uint32_t j = sv[src][i];
    if constexpr (ROWS) io.out[lo + i] = io.fetch(iin[j]); else val_out[lo + i] = vin[j];
  }
}

template <int BITS, int ROUNDS, typename SRC, typename ROW>
SortResult bucket_sort(hipStream_t st, SRC src, uint64_t* key, uint32_t* val, uint64_t* key_scratch, uint32_t* val_scratch, const uint32_t* d_n,
                       int64_t max_n, uint64_t kmax, const SortWork& w, RowIO<ROW> io) {
  int s = 0;
  while ((kmax >> s) > 0xFFFFFFFFull) ++s;
  BucketDigit dg;
  dg.s = s;
  dg.M = (uint32_t)((((uint64_t)((1u << BITS) - 1u)) << 32) / ((kmax >> s) + 1ull));
  uint32_t* totals = w.totals + (size_t)(*w.pass_cursor) * 2048;
  ++*w.pass_cursor;
  const int64_t max_tiles = (max_n + rs_tile<ROUNDS>() - 1) / rs_tile<ROUNDS>();
  const int grid = (int)(max_tiles < 1024 ? (max_tiles < 1 ? 1 : max_tiles) : 1024);
  static const int cap_env = vsv_dbg_env("VSV_BK_CAP") ? atoi(vsv_dbg_env("VSV_BK_CAP")) : BK_CAP;      // tests force the fallback with a tiny capacity
  const uint32_t cap = (uint32_t)(cap_env < 2 ? 2 : cap_env > BK_CAP ? BK_CAP : cap_env);
  // two launches: a fixed region of slots per bucket in the scratch pair (when it holds 2^BITS regions of a useful size)
  static const char* slots_env = vsv_dbg_env("VSV_BK_SLOTS");      // timing experiments / tests: "0" = the three-launch forms
  const int64_t caps64 = max_n >> BITS;
  const uint32_t caps = (uint32_t)(caps64 > BK_CAP ? BK_CAP : caps64);
  if (w.slots_ok && caps >= 2048 && (int64_t)caps << BITS <= w.max_items && !(slots_env && slots_env[0] == '0')) {
    uint32_t* slot_val = src.val_lut() ? w.hist : nullptr;        // (the histogram buffer is free in this form: 2 words per row of capacity)
    static const int tie_env = vsv_dbg_env("VSV_BK_TIEMAX") ? atoi(vsv_dbg_env("VSV_BK_TIEMAX")) : 64;      // tests: 0 = every bucket whose chunks arrived out of order takes the tile-number passes
    const uint32_t tie_max = (uint32_t)(tie_env < 0 ? 0 : tie_env > 64 ? 64 : tie_env);
    rs_slot_scatter<BITS, ROUNDS, BucketDigit, SRC><<<grid, 256, 0, st>>>(src, d_n, dg, totals, caps, key_scratch, val_scratch, slot_val, w.err);
    if (w.shared_gpu) bk_slot_sort<ROW, 256><<<1 << BITS, 256, 0, st>>>(key_scratch, val_scratch, slot_val, caps, totals, 1 << BITS, d_n, key, val, w.err, cap, rs_tile<ROUNDS>(), tie_max, io);
    else bk_slot_sort<ROW, 512><<<1 << BITS, 512, 0, st>>>(key_scratch, val_scratch, slot_val, caps, totals, 1 << BITS, d_n, key, val, w.err, cap, rs_tile<ROUNDS>(), tie_max, io);
    return SortResult{key, val};
  }
  // three launches. The self-scanning form needs a zeroed slot of group sums and a table of at most VSV_RS_MAX_GROUPS groups of tiles
  // (the row count of the handle's previous run; a table that grew past it is caught below: the sort falls back like an overflowing
  // bucket); bucket bases = the scanned histogram row of tile 0 (offset of the first tile's rows of every bucket), or the totals in front
  static const char* scan_env = vsv_dbg_env("VSV_BK_SCAN");      // timing experiments / tests: "self" | "launch"
  uint32_t* groups = nullptr;
  if (w.groups && *w.group_cursor < w.max_group_slots && w.hint_rows > 0 &&
      w.hint_rows + w.hint_rows / 2 <= (uint64_t)VSV_RS_MAX_GROUPS * RS_GROUP * rs_tile<ROUNDS>() && !(scan_env && scan_env[0] == 'l')) {
    groups = w.groups + (size_t)(*w.group_cursor) * VSV_RS_MAX_GROUPS * 2048;
    ++*w.group_cursor;
  }
  const bool self = groups != nullptr;
  rs_hist<BITS, ROUNDS, BucketDigit, SRC><<<grid, 256, 0, st>>>(src, d_n, dg, w.hist, totals, groups);
  if (self) rs_scatter<BITS, ROUNDS, BucketDigit, SRC, true><<<grid, 256, 0, st>>>(src, d_n, dg, w.hist, key_scratch, val_scratch, totals, groups);
  else {
    rs_scan_mb<BITS, ROUNDS, 32, 32><<<(1 << BITS) / 32, 1024, 0, st>>>(w.hist, totals, d_n);
    rs_scatter<BITS, ROUNDS, BucketDigit, SRC, false><<<grid, 256, 0, st>>>(src, d_n, dg, w.hist, key_scratch, val_scratch, totals);
  }
  if (w.shared_gpu) bk_lds_sort<ROW, 256><<<1 << BITS, 256, 0, st>>>(key_scratch, val_scratch, w.hist, self ? totals : nullptr, 1 << BITS, d_n, key, val, w.err, cap, rs_tile<ROUNDS>(), io);
  else bk_lds_sort<ROW, 512><<<1 << BITS, 512, 0, st>>>(key_scratch, val_scratch, w.hist, self ? totals : nullptr, 1 << BITS, d_n, key, val, w.err, cap, rs_tile<ROUNDS>(), io);
  return SortResult{key, val};
}

// 0: the LSD passes; 8..11: bucket sort with that many bucket bits
int bucket_bits_for(const SortWork& w, int nbits, uint64_t& kmax) {
  if (!(w.bucket_bits >= 8 && w.err && w.totals && *w.pass_cursor < w.max_passes && nbits > 12 && nbits - 1 <= 63)) return 0;
  if (kmax == 0) kmax = 1ull << (nbits - 1);      // unless the caller knows better: the top key bit is the dead flag
  const int b = w.bucket_bits > 11 ? 11 : w.bucket_bits;
  return (kmax >> b) < 0xFFFFFFFFull ? b : 0;
}
template <typename SRC, typename ROW>
SortResult bucket_sort_any(hipStream_t st, int b, SRC src, uint64_t* key, uint32_t* val, uint64_t* key_scratch, uint32_t* val_scratch,
                           const uint32_t* d_n, int64_t max_n, uint64_t kmax, const SortWork& w, RowIO<ROW> io) {
#define BK_CASE(B)                                                                                                                         \
  case B: return w.small_tiles ? bucket_sort<B, RS_ROUNDS_SMALL, SRC, ROW>(st, src, key, val, key_scratch, val_scratch, d_n, max_n, kmax, w, io) \
                               : bucket_sort<B, RS_ROUNDS_BIG, SRC, ROW>(st, src, key, val, key_scratch, val_scratch, d_n, max_n, kmax, w, io);
  switch (b) {
    BK_CASE(8) BK_CASE(9) BK_CASE(10)
    default: return w.small_tiles ? bucket_sort<11, RS_ROUNDS_SMALL, SRC, ROW>(st, src, key, val, key_scratch, val_scratch, d_n, max_n, kmax, w, io)
                                  : bucket_sort<11, RS_ROUNDS_BIG, SRC, ROW>(st, src, key, val, key_scratch, val_scratch, d_n, max_n, kmax, w, io);
  }
#undef BK_CASE
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
      case 8: one_pass<8, ROUNDS>(st, max_n, kin, vin, d_n, shift, w.hist, kout, vout, totals, w.hint_rows); shift += 8; break;
      case 9: one_pass<9, ROUNDS>(st, max_n, kin, vin, d_n, shift, w.hist, kout, vout, totals, w.hint_rows); shift += 9; break;
      case 10: one_pass<10, ROUNDS>(st, max_n, kin, vin, d_n, shift, w.hist, kout, vout, totals, w.hint_rows); shift += 10; break;
      default: one_pass<11, ROUNDS>(st, max_n, kin, vin, d_n, shift, w.hist, kout, vout, totals, w.hint_rows); shift += 11; break;
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
                                const uint32_t* d_n, int64_t max_n, int nbits, const SortWork& w, uint64_t kmax) {
  // bucket sort when the caller allows it (bucket_bits from the row counts of the handle's previous run), the keys of one bucket
  // can fit 32 bits and a zeroed totals slot is left; anything else takes the LSD passes
  const int b = bucket_bits_for(w, nbits, kmax);
  if (b) return bucket_sort_any<KeyArr, void>(st, b, KeyArr{key, val}, key, val, key_scratch, val_scratch, d_n, max_n, kmax, w, RowIO<void>{});
  return w.small_tiles ? sort_pairs<RS_ROUNDS_SMALL>(st, key, val, key_scratch, val_scratch, d_n, max_n, nbits, w)
                 : sort_pairs<RS_ROUNDS_BIG>(st, key, val, key_scratch, val_scratch, d_n, max_n, nbits, w);
}

// Sorted copy of a signature table by its stage key / of a call table by (tid, pos), in 4 launches: keys come straight from the rows
// (hist + scatter), the LDS sort writes the rows. Returns the sorted keys (key_out), or nullptr when the bucket sort does not apply
// (the caller then builds keys, runs vsv_radix_sort_pairs and gathers).
const uint64_t* vsv_bucket_sort_sigs(hipStream_t st, const vsv_sig* in, const uint32_t* d_n, int stage, int pb, int tid_lo, int tid_bits, int nbits,
                                     uint64_t kmax, vsv_sig* sorted, uint64_t* key_out, uint32_t* d_alive, uint32_t* n_long, int32_t* fill,
                                     const SortWork& w, int64_t max_n) {
  const int b = bucket_bits_for(w, nbits, kmax);
  if (!b) return nullptr;
  bucket_sort_any<SigKey, vsv_sig>(st, b, SigKey{in, stage, pb, tid_lo, tid_bits, w.err}, key_out, nullptr, w.key_alt, w.val_alt, d_n, max_n, kmax, w,
                                   RowIO<vsv_sig>{in, sorted, d_alive, n_long, fill});
  return key_out;
}
const uint64_t* vsv_bucket_sort_calls(hipStream_t st, const vsv_call* in, const uint32_t* d_n, int pb, int tid_lo, int nbits, uint64_t kmax,
                                      vsv_call* sorted, uint64_t* key_out, uint32_t* d_alive, uint32_t* n_long, const SortWork& w, int64_t max_n,
                                      const vsv_sig* merged, const int32_t* st2) {
  const int b = bucket_bits_for(w, nbits, kmax);
  if (!b) return nullptr;
  bucket_sort_any<CallKey, vsv_call>(st, b, CallKey{in, pb, tid_lo, merged, st2}, key_out, nullptr, w.key_alt, w.val_alt, d_n, max_n, kmax, w,
                                     RowIO<vsv_call>{in, sorted, d_alive, n_long, nullptr, merged});
  return key_out;
}

// pair slots sorted by (tid, hap, record) without materialising their keys first; {nullptr, nullptr} = not applicable (the caller runs
// split_mark_pairs + vsv_radix_sort_pairs)
SortResult vsv_bucket_sort_pair_slots(hipStream_t st, const uint64_t* ckey, const uint32_t* crec, int qid_bits, int rec_bits, const uint32_t* d_n,
                                      uint64_t* okey, uint32_t* oval, uint64_t* key_scratch, uint32_t* val_scratch, int64_t max_n, int nbits, const SortWork& w) {
  uint64_t kmax = 0;
  const int b = bucket_bits_for(w, nbits, kmax);
  if (!b) return SortResult{nullptr, nullptr};
  return bucket_sort_any<PairKey, void>(st, b, PairKey{ckey, crec, qid_bits, rec_bits, d_n}, okey, oval, key_scratch, val_scratch, d_n, max_n, kmax, w, RowIO<void>{});
}

int64_t vsv_radix_hist_entries(int64_t max_n) { return 2048 * ((max_n + rs_tile<RS_ROUNDS_SMALL>() - 1) / rs_tile<RS_ROUNDS_SMALL>()); }
