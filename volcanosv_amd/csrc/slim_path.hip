// slim_path.hip — the stages behind the scan for LARGE signature tables (10^6-10^7 rows: contig alignments piled on one
// chromosome; config 3's ONT tables):
//   sort_sig                   H:170-179     -> sl_hist / sl_scan / sl_scatter (stable LSD passes over 16-byte elements)
//   cluster_del / cluster_ins  H:196-288     -> sl_cluster
//   merge_all                  H:478-499     -> the same two kernels on the stage-2 / stage-3 keys
//   pair_sig                   H:548-603     -> sl_pair_index + sl_pair_lds (or sl_pair_prep + sl_pair_round* on dense piles) + the call sort
// (paths relative to bin/VolcanoSV-vc/Large_INDEL/extract_contig_signature_Hifi.py in the reference).
//
// Everything a stage decides depends on (list, pos, svlen, type) only, and every row a stage passes on is a verbatim copy of a
// row of the stage-1 input table (the fold's output + the split rows): cluster representatives, merged rows and the kept
// signature of a call are all rows of that table. So the stages run on SLIM elements — {stage key, svlen, row index | type} =
// 16 bytes instead of 32-byte rows + 8-byte keys + 4-byte indices — and the 32-byte rows are gathered exactly once, for the call
// table (and for VSV_T_CLUSTER1 / VSV_T_MERGED when a caller asks for them). A stage that drops a row writes a dead element; the
// next sort's first pass skips dead elements, so its output is compact and its element count is the live count. Sorting is
// stable LSD with 8-9 bit digits over 4096-element tiles: 16 elements of a digit per tile = 256 contiguous bytes per (tile,
// digit), where the 11-bit digits of radix_sort.hip (tuned for the launch-bound small tables) write 2 elements = partial lines.
#include "vsv_device.h"

namespace {

struct __align__(16) Slim {
  uint64_t key;     // stage key (vsv_key_stage layout), VSV_KEY_DEAD = dropped
  int32_t svlen;    // (call elements: index of the hp2 mate in the merged table, -1 = none)
  uint32_t idx;     // row of the stage-1 input table | SL_DEL  (call elements: slot in the merged table)
};
constexpr uint32_t SL_DEL = 0x80000000u;
constexpr uint32_t SL_ROW = 0x7FFFFFFFu;
// key bit that carries an element's CLASS from the stage that drops a list bit to the sort behind it (sl_merge_sort below): the value of
// the dropped bit (source / type / haplotype). Set only when that sort runs as rank-inside-the-class + merge, which clears it again.
constexpr uint64_t SL_CLASS = 1ull << 62;
__device__ __forceinline__ uint64_t sl_stash(uint64_t key, uint64_t drop, uint64_t stash) { return (key & ~drop) | ((key & drop) ? stash : 0ull); }
// geometry of that sort (sl_merge_sort, further down; the cluster kernel counts its live outputs per tile for it)
constexpr int MS_T = 2048, MS_H = 128, MS_W = MS_T + 2 * MS_H, MS_R = MS_W / 256;      // 2304 window slots = 9 rounds of a 256-thread block
constexpr int MS_MT = 4096;                                                            // outputs per merge tile
constexpr int MS_GROUP = 64, MS_MAX_GROUPS = 1016;                                     // tile counts are also summed per group of 64 tiles
struct MsCtl { uint32_t nA, nB, pad[14]; uint32_t grp[MS_MAX_GROUPS][2]; };            // lives in a zeroed per-pass totals slot (2048 words)
static_assert(sizeof(MsCtl) == 2048 * 4, "one totals slot");
static_assert(MS_H % 64 == 0 && MS_W % 256 == 0, "window rounds are whole waves");

// per-tile counts of a producer's live outputs: tcnt[2 tile + class], the group sums and class totals of ctl (all zeroed beforehand)
struct MsCount {
  uint32_t* tcnt; MsCtl* ctl;
  __device__ __forceinline__ void add(uint32_t tile, uint32_t cls, uint32_t v) const {
    atomicAdd(&tcnt[2 * tile + cls], v);
    atomicAdd(&ctl->grp[tile / MS_GROUP][cls], v);
    atomicAdd(cls ? &ctl->nB : &ctl->nA, v);
  }
};

__device__ __forceinline__ Slim ld_slim(const Slim* p) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  Slim s;
  s.key = (uint64_t)v.x | ((uint64_t)v.y << 32); s.svlen = (int32_t)v.z; s.idx = v.w;
  return s;
}
__device__ __forceinline__ void st_slim(Slim* p, const Slim& s) {
  *reinterpret_cast<uint4*>(p) = make_uint4((uint32_t)s.key, (uint32_t)(s.key >> 32), (uint32_t)s.svlen, s.idx);
}
__device__ __forceinline__ Slim dead_slim() { Slim s; s.key = VSV_KEY_DEAD; s.svlen = 0; s.idx = 0; return s; }

// 8-byte element of the split stage's candidate sorts on large inputs (config 3: 10^7 candidates): a key of at most 31 bits and the
// row's ordinal; what the consumers need beyond that (record, candidate ordinals) is gathered by ordinal when the sorted list is unpacked.
// The passes below take either element type (SRC::at decides).
struct Slim8 { uint32_t key, val; };
constexpr uint32_t SL8_DEAD = 0xFFFFFFFFu;
__device__ __forceinline__ Slim8 ld_slim(const Slim8* p) { const uint2 v = *reinterpret_cast<const uint2*>(p); return Slim8{v.x, v.y}; }
__device__ __forceinline__ void st_slim(Slim8* p, const Slim8& s) { *reinterpret_cast<uint2*>(p) = make_uint2(s.key, s.val); }
__device__ __forceinline__ bool sl_dead(const Slim& e) { return e.key == VSV_KEY_DEAD; }
__device__ __forceinline__ bool sl_dead(const Slim8& e) { return e.key == SL8_DEAD; }
template <typename E> __device__ __forceinline__ E sl_dead_elem();
template <> __device__ __forceinline__ Slim sl_dead_elem<Slim>() { return dead_slim(); }
template <> __device__ __forceinline__ Slim8 sl_dead_elem<Slim8>() { return Slim8{SL8_DEAD, 0u}; }

struct KeyFmt {      // bit layout of the run's keys
  int pb;            // position bits; hap at pb + 2, type at pb + 1, source at pb, tid above pb + 3
  __device__ __forceinline__ int32_t pos(uint64_t k) const {
    const uint32_t kp = pb >= 32 ? (uint32_t)k : (uint32_t)k & ((1u << pb) - 1u);
    return (int32_t)(kp - (uint32_t)VSV_POS_BIAS);
  }
};

// exact integer form of the reference's ratio predicates, as vsv_match (vsv_device.h) on (pos, svlen, type) — in 32 bits: the element
// path keeps only inputs whose positions (+ bias) fit 30 bits and whose lengths lie in [0, 2^30) (checked where the elements are
// built: ERRB_SLIM_FALLBACK sends anything else back to the row path), so no sum or difference below leaves int32.
// The cluster and pairing kernels are bound by instructions issued; the 64-bit form of this predicate was a third of them.
__device__ __forceinline__ bool sl_match(int32_t p1, int32_t v1, int32_t p2, int32_t v2, bool del, int max_shift) {
  const int32_t d = p1 - p2;
  if ((d < 0 ? -d : d) > max_shift) return false;
  const int32_t mn = v1 < v2 ? v1 : v2, mx = v1 < v2 ? v2 : v1;
  if (2 * mn < mx) return false;
  if (del) {
    const int32_t e1 = p1 + v1, e2 = p2 + v2;
    const int32_t ov = (e1 < e2 ? e1 : e2) - (p1 > p2 ? p1 : p2);
    if (2 * ov < mn) return false;
  }
  return true;
}

// ---- rows -> slim elements (stage-1 keys) ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sl_from_rows(const vsv_sig* __restrict__ rows, const uint32_t* __restrict__ d_n, int pb, int tid_lo, int tid_bits,
                                                    Slim* __restrict__ out, uint32_t* __restrict__ err) {
  const uint32_t n = *d_n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const vsv_sig v = rows[i];
    Slim s;
    s.key = vsv_key_stage(v, 1, pb, tid_lo);
    s.svlen = v.svlen;
    s.idx = i | ((v.meta & VSV_M_DEL) ? SL_DEL : 0u);
    if (!(v.meta & VSV_M_DEAD) && ((pb < 32 && (vsv_kpos(v.pos) >> pb) != 0) || ((uint32_t)(v.tid - tid_lo) >> tid_bits) != 0))
      atomicOr(err, ERRB_RANGE);            // max_pos hint too small / tid outside [tid_lo, n_tids)
    if (!(v.meta & VSV_M_DEAD) && ((uint32_t)v.svlen >= (1u << 30) || (vsv_kpos(v.pos) >> 30) != 0)) atomicOr(err, ERRB_SLIM_FALLBACK);
    st_slim(out + i, s);
  }
}

// ---- stable LSD radix sort of slim elements ----------------------------------------------------------------------------------
// A tile = 4096 elements, worked by a block of WAVES waves (4 or 16: a table of a few million elements has only some hundred tiles,
// and 4 waves per tile leave the chip at two waves per SIMD with nothing to hide the dependent LDS / ballot chains behind).
constexpr uint32_t SL_TILE = 4096;

// where the elements of a sort's first pass come from: an array, or the pairing state (the call elements are never materialised)
struct SrcSlim {
  using elem = Slim;
  const Slim* p;
  __device__ __forceinline__ Slim at(uint32_t i) const { return ld_slim(p + i); }
};
struct SrcSlim8 {
  using elem = Slim8;
  const Slim8* p;
  __device__ __forceinline__ Slim8 at(uint32_t i) const { return ld_slim(p + i); }
};
template <typename E> struct SrcOf;
template <> struct SrcOf<Slim> { using type = SrcSlim; };
template <> struct SrcOf<Slim8> { using type = SrcSlim8; };
template <typename SRC> using SlElem = typename SRC::elem;       // the element type a source yields
// Call element of merged slot i (pair_sig's output rows, H:571-592): an hp1 row is a call whatever happened — alone (0/1) or with
// its mate (1/1: the longer of the two signatures is kept, the hp1 one on ties, H:583-586); an hp2 row is a call iff nobody took
// it. Key = (tid, pos) of the kept signature; svlen field = the mate's slot; idx = the slot itself.
struct SrcCalls {
  using elem = Slim;
  const Slim* m; const int32_t* st; int hap_bit; uint64_t stash;
  __device__ __forceinline__ Slim at(uint32_t i) const {
    const Slim me = ld_slim(m + i);
    const uint64_t hb = 1ull << hap_bit;
    Slim c;
    c.idx = i;
    const int32_t s = st[i];
    if (me.key & hb) {
      c.svlen = -1;
      c.key = s == -1 ? ((me.key & ~hb) | stash) : VSV_KEY_DEAD;
    } else {
      c.svlen = s;
      uint64_t k = me.key;
      if (s >= 0) { const Slim mate = ld_slim(m + s); if (!(me.svlen > mate.svlen)) k = mate.key & ~hb; }
      c.key = k;
    }
    return c;
  }
  // (sl_merge_sort) 0 = no call in this slot, 1 = class A (an hp1 row), 2 = class B (an hp2 row nobody took)
  __device__ __forceinline__ uint32_t lc(uint32_t i) const {
    return ((m[i].key >> hap_bit) & 1ull) ? (st[i] == -1 ? 2u : 0u) : 1u;
  }
  // ... the call element and the slot's ANCHOR: the slot's own key, which the slots ascend by inside a class (a call's key lies
  // within pair_shift of it: its own or its mate's)
  __device__ __forceinline__ Slim at2(uint32_t i, uint64_t& anchor) const {
    const uint64_t hb = 1ull << hap_bit, k = m[i].key;
    anchor = (k & ~hb) | ((k & hb) ? SL_CLASS : 0ull);
    return at(i);
  }
  __device__ __forceinline__ uint64_t key2(uint32_t i, uint64_t& anchor) const { return at2(i, anchor).key; }
};

// The keys of the later stages carry list bits that are zero for every element (source after the first clustering, type after the
// second, haplotype in the call keys): the passes sort on the key with those bits squeezed out, which saves a pass (30 bits = three
// 10-bit passes for a single-chromosome shard).
struct KeyCmp {
  int zlo, zbits;      // `zbits` zero bits at bit `zlo` are not sorted on (zbits = 0: the key as it is)
  __device__ __forceinline__ uint64_t operator()(uint64_t k) const {
    return zbits ? ((k & ((1ull << zlo) - 1ull)) | ((k >> (zlo + zbits)) << zlo)) : k;
  }
};

// the digit a pass sorts on: a bit field of the (squeezed) key for the LSD passes, or — the first sort's one counting pass in front
// of its LDS sort per bucket — a monotone map of the whole stage-1 key onto contiguous buckets (sl_bucket_sort1 below)
struct LsdDigit {
  int shift; KeyCmp kc; uint32_t mask;
  __device__ __forceinline__ LsdDigit prep() const { return *this; }
  __device__ __forceinline__ uint32_t operator()(uint64_t key) const { return (uint32_t)(kc(key) >> shift) & mask; }
};
// list pair p = key >> (pb + 1) owns buckets [p << bp_log, (p + 1) << bp_log): b0 for its cigar list, nb1 for its split list, each cut
// into equal position ranges between the smallest and the largest position the table holds (mm: sl_minmax or the in-place fold) — a shard that covers a part of its chromosome, or contigs that reach beyond the max_pos hint, fill all buckets alike
struct MsdPrepared {
  int pb, bp_log; uint32_t b0, nb1, mul0, mul1, kmin;
  __device__ __forceinline__ uint32_t operator()(uint64_t key) const {
    const uint32_t list = (uint32_t)(key >> pb), src = list & 1u;
    const uint32_t kq = (uint32_t)key & ((1u << pb) - 1u), kp = kq > kmin ? kq - kmin : 0u;
    uint32_t b = (uint32_t)(((uint64_t)kp * (src ? mul1 : mul0)) >> 32);
    const uint32_t nb = src ? nb1 : b0;
    b = b < nb ? b : nb - 1u;
    return ((list >> 1) << bp_log) + (src ? b0 : 0u) + b;
  }
};
// mm[0..63] = max kpos, mm[64..127] = max ~kpos over the live elements, one slot per residue of the writing block's index (thousands of
// atomics on ONE address take their turns at the L2, ~10 ns each, and in a position-sorted table every block brings a new maximum)
struct MsdDigit {
  int pb, bp_log; uint32_t b0, nb1; const uint32_t* mm;
  __device__ __forceinline__ MsdPrepared prep() const {          // called by whole waves
    const int lane = threadIdx.x & 63;
    uint32_t kmax = mm[lane], lo = mm[64 + lane];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, d, 64)); lo = max(lo, (uint32_t)__shfl_xor((int)lo, d, 64)); }
    const uint32_t kmin = ~lo;
    const uint64_t range = kmax >= kmin ? (uint64_t)(kmax - kmin) + 1u : 1u;        // (no live element: nothing is mapped)
    // floor(b << 32 / range) as a 32-bit multiplier; a range smaller than the bucket count maps one position per bucket
    const uint64_t m0 = ((uint64_t)b0 << 32) / range, m1 = ((uint64_t)nb1 << 32) / range;
    return MsdPrepared{pb, bp_log, b0, nb1, m0 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)m0, m1 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)m1, kmin};
  }
};

// the position range of the live elements (both halves of mm zeroed with the run's sort scratch)
__global__ __launch_bounds__(256) void sl_minmax(const Slim* __restrict__ e, const uint32_t* __restrict__ d_n, int pb, uint32_t* __restrict__ mm) {
  __shared__ uint32_t s_hi, s_lo;
  const uint32_t n = *d_n, pmask = (1u << pb) - 1u;
  if (threadIdx.x == 0) { s_hi = 0; s_lo = 0; }
  __syncthreads();
  uint32_t hi = 0, lo = 0;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const uint64_t k = e[i].key;
    if (k != VSV_KEY_DEAD) { const uint32_t kp = (uint32_t)k & pmask; hi = max(hi, kp); lo = max(lo, ~kp); }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { hi = max(hi, (uint32_t)__shfl_xor((int)hi, d, 64)); lo = max(lo, (uint32_t)__shfl_xor((int)lo, d, 64)); }
  if ((threadIdx.x & 63) == 0) { atomicMax(&s_hi, hi); atomicMax(&s_lo, lo); }
  __syncthreads();
  if (threadIdx.x == 0) { if (s_hi) atomicMax(&mm[blockIdx.x & 63], s_hi); if (s_lo) atomicMax(&mm[64 + (blockIdx.x & 63)], s_lo); }
}

template <int BITS>
__device__ __forceinline__ uint64_t sl_match_digit(uint32_t d, bool valid) {
  uint64_t m = __ballot(valid);
#pragma unroll
  for (int b = 0; b < BITS; ++b) {
    const uint64_t bal = __ballot((d >> b) & 1u);
    m &= ((d >> b) & 1u) ? bal : ~bal;
  }
  return m;
}
__device__ __forceinline__ uint32_t sl_tiles(uint32_t n) { return (n + SL_TILE - 1) / SL_TILE; }

// hist[tile][d] = elements of the tile with digit d (dead elements of a first pass do not count); totals[d] += the same
template <int BITS, typename SRC, bool SKIP_DEAD, int WAVES, typename DIG>
__global__ __launch_bounds__(WAVES * 64) void sl_hist(SRC src, const uint32_t* __restrict__ d_n, DIG dig_, uint32_t* __restrict__ hist, uint32_t* __restrict__ totals) {
  constexpr int BINS = 1 << BITS, T = WAVES * 64;
  __shared__ uint32_t cnt[BINS];
  const uint32_t n = *d_n, ntiles = sl_tiles(n);
  const auto dig = dig_.prep();
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (int d = threadIdx.x; d < BINS; d += T) cnt[d] = 0;
    __syncthreads();
    const uint32_t base = tile * SL_TILE;
#pragma unroll 4
    for (int k = 0; k < (int)SL_TILE / T; ++k) {
      const uint32_t i = base + k * T + threadIdx.x;
      if (i < n) {
        const auto e = src.at(i);
        if (!SKIP_DEAD || !sl_dead(e)) atomicAdd(&cnt[dig(e.key)], 1u);
      }
    }
    __syncthreads();
    for (int d = threadIdx.x; d < BINS; d += T) {
      const uint32_t c = cnt[d];
      hist[(size_t)tile * BINS + d] = c;
      if (c) atomicAdd(&totals[d], c);
    }
    __syncthreads();
  }
}

// hist[tile][d] -> exclusive offsets in (digit, tile) order. A block owns 16 digits x 64 groups of tiles (1024 threads); digit bases
// from the totals. Block 0 also publishes the pass's element count (= live elements, for a first pass) where asked.
template <int BITS>
__global__ __launch_bounds__(1024) void sl_scan(uint32_t* __restrict__ hist, const uint32_t* __restrict__ totals, const uint32_t* __restrict__ d_n,
                                                uint32_t* __restrict__ d_total) {
  constexpr int BINS = 1 << BITS, DG = 16, TG = 64;
  __shared__ uint32_t red[1024];
  __shared__ uint32_t dig[DG];
  __shared__ uint32_t part[TG][DG];
  const uint32_t ntiles = sl_tiles(*d_n);
  const int t = threadIdx.x, dl = t & (DG - 1), g = t / DG;
  const int d0 = blockIdx.x * DG, d = d0 + dl;
  auto block_sum = [&](uint32_t v) -> uint32_t {
    __syncthreads();
    red[t] = v;
    __syncthreads();
    for (int k = 512; k > 0; k >>= 1) { if (t < k) red[t] += red[t + k]; __syncthreads(); }
    return red[0];
  };
  uint32_t before = 0;
  for (int i = t; i < d0; i += 1024) before += totals[i];
  const uint32_t prev = block_sum(before);          // elements with a digit in front of this block's
  if (blockIdx.x == 0 && d_total) {                  // (uniform per block)
    uint32_t all = 0;
    for (int i = t; i < BINS; i += 1024) all += totals[i];
    all = block_sum(all);
    if (t == 0) *d_total = all;
  }
  const uint32_t mine = t < DG ? totals[d] : 0;
  if (t < DG) dig[t] = mine;
  __syncthreads();
  for (int k = 1; k < DG; k <<= 1) {
    const uint32_t v = (t < DG && t >= k) ? dig[t - k] : 0;
    __syncthreads();
    if (t < DG) dig[t] += v;
    __syncthreads();
  }
  if (t < DG) dig[t] = prev + dig[t] - mine;        // exclusive digit base
  const uint32_t per = (ntiles + TG - 1) / TG;
  const uint32_t t0 = min(ntiles, (uint32_t)g * per), t1 = min(ntiles, t0 + per);
  uint32_t sum = 0;
  for (uint32_t tb = t0; tb < t1; tb += 16) {
    uint32_t c[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) c[k] = (tb + k < t1) ? hist[(size_t)(tb + k) * BINS + d] : 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += c[k];
  }
  part[g][dl] = sum;
  __syncthreads();
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

// stable scatter of one pass: ranks inside a wave from ballot matches, per-wave running counters in LDS (as rs_scatter)
template <int BITS, typename SRC, bool SKIP_DEAD, int WAVES, typename DIG>
__global__ __launch_bounds__(WAVES * 64) void sl_scatter(SRC src, const uint32_t* __restrict__ d_n, DIG dig_, const uint32_t* __restrict__ hist, SlElem<SRC>* __restrict__ out) {
  using E = SlElem<SRC>;
  constexpr int BINS = 1 << BITS, T = WAVES * 64, ROUNDS = (int)SL_TILE / T;
  __shared__ uint32_t wcnt[WAVES][BINS];
  const uint32_t n = *d_n, ntiles = sl_tiles(n);
  const auto dig = dig_.prep();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t lt = (1ull << lane) - 1ull;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (int d = threadIdx.x; d < BINS; d += T)
#pragma unroll
      for (int w = 0; w < WAVES; ++w) wcnt[w][d] = 0;
    __syncthreads();
    const uint32_t wbase = tile * SL_TILE + wv * (ROUNDS * 64);
    E e_[ROUNDS];
    uint32_t rk[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const uint32_t i = wbase + r * 64 + lane;
      bool ok = i < n;
      if (ok) e_[r] = src.at(i); else e_[r] = sl_dead_elem<E>();
      if (SKIP_DEAD) ok = ok && !sl_dead(e_[r]);
      const uint32_t d = ok ? dig(e_[r].key) : 0u;
      const uint64_t m = sl_match_digit<BITS>(d, ok);
      const uint32_t old = ok ? wcnt[wv][d] : 0;
      __builtin_amdgcn_wave_barrier();
      if (ok && (m & lt) == 0) wcnt[wv][d] = old + (uint32_t)__popcll(m);
      __builtin_amdgcn_wave_barrier();
      rk[r] = ok ? old + (uint32_t)__popcll(m & lt) : 0xFFFFFFFFu;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < BINS; d += T) {  // exclusive prefix of the digit's count over the waves + global base
      uint32_t run = hist[(size_t)tile * BINS + d];
#pragma unroll
      for (int w = 0; w < WAVES; ++w) { const uint32_t c = wcnt[w][d]; wcnt[w][d] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      if (rk[r] != 0xFFFFFFFFu) {
        const uint32_t d = dig(e_[r].key);
        st_slim(out + wcnt[wv][d] + rk[r], e_[r]);
      }
    }
    __syncthreads();
  }
}

// The counting pass of the first sort (11 / 12-bit digits: sl_bucket_sort1) in blocks of 8 waves: the 4-wave form above keeps 16 elements
// per lane in registers (196 VGPRs: two waves per SIMD) next to 64 KB of per-wave counters (two blocks per CU). Here a lane holds 8
// elements, the per-wave counters are 16 bits wide (a wave counts at most 512 of a digit, a tile 4096) and the tile's digit bases sit in
// an array of their own: 80 KB per block, two blocks = 16 waves per CU.
template <int BITS, typename SRC, typename DIG>
__global__ __launch_bounds__(512) void sl_scatter_msd(SRC src, const uint32_t* __restrict__ d_n, DIG dig_, const uint32_t* __restrict__ hist, SlElem<SRC>* __restrict__ out) {
  using E = SlElem<SRC>;
  constexpr int BINS = 1 << BITS, WAVES = 8, T = WAVES * 64, ROUNDS = (int)SL_TILE / T;
  __shared__ uint16_t wcnt[WAVES][BINS];
  __shared__ uint32_t base[BINS];
  const uint32_t n = *d_n, ntiles = sl_tiles(n);
  const auto dig = dig_.prep();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint64_t lt = (1ull << lane) - 1ull;
  uint32_t* wz = reinterpret_cast<uint32_t*>(&wcnt[0][0]);
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (int d = threadIdx.x; d < WAVES * BINS / 2; d += T) wz[d] = 0;
    __syncthreads();
    const uint32_t wbase = tile * SL_TILE + wv * (ROUNDS * 64);
    E e_[ROUNDS];
    uint32_t rk[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      const uint32_t i = wbase + r * 64 + lane;
      bool ok = i < n;
      if (ok) e_[r] = src.at(i); else e_[r] = sl_dead_elem<E>();
      ok = ok && !sl_dead(e_[r]);
      const uint32_t d = ok ? dig(e_[r].key) : 0u;
      const uint64_t m = sl_match_digit<BITS>(d, ok);
      const uint32_t old = ok ? wcnt[wv][d] : 0;
      __builtin_amdgcn_wave_barrier();
      if (ok && (m & lt) == 0) wcnt[wv][d] = (uint16_t)(old + (uint32_t)__popcll(m));
      __builtin_amdgcn_wave_barrier();
      rk[r] = ok ? old + (uint32_t)__popcll(m & lt) : 0xFFFFFFFFu;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < BINS; d += T) {    // the waves' shares of the digit's piece, and where the piece starts
      uint32_t run = 0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) { const uint32_t c = wcnt[w][d]; wcnt[w][d] = (uint16_t)run; run += c; }
      base[d] = hist[(size_t)tile * BINS + d];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
      if (rk[r] != 0xFFFFFFFFu) {
        const uint32_t d = dig(e_[r].key);
        st_slim(out + base[d] + wcnt[wv][d] + rk[r], e_[r]);
      }
    }
    __syncthreads();
  }
}

// ---- seeded greedy clustering on the sorted elements (H:196-288) ------------------------------------------------------------
// A run = maximal stretch of one list whose consecutive positions differ by <= max_shift; no match crosses a run boundary, so
// runs are independent and the reference's full scans reduce to a scan inside the run. The lane that holds a run's first element
// performs the sequential greedy exactly (seed = first unassigned element, members = unassigned elements that match the SEED,
// representative = first longest member, H:236-247); runs of more than SL_LONG_RUN elements are taken over by the lane's whole
// wave. Slot i of the output = the representative (with the next stage's key: `drop` cleared) if element i is a seed, dead
// otherwise — seed order. A block stages its 1024 slots (+ a halo) in LDS: run heads, run ends and the greedy's inner loop read
// there instead of chasing dependent global loads.
constexpr uint32_t SL_LONG_RUN = 48;
constexpr int CL_TILE = 1024, CL_HALO = 127;       // LDS holds slots [t0 - 1, t0 + CL_TILE + CL_HALO)

__device__ __forceinline__ int32_t ld_i32(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_i32(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One wave on a long run [i, e): seeds stay sequential, the scan of a seed's window and the search for the next seed are 64-wide.
// cl[] (one word per slot) is accessed with agent-scope relaxed atomics. Called by ALL 64 lanes with wave-uniform (i, e).
__device__ __forceinline__ void sl_cluster_long(const Slim* __restrict__ s, int max_shift, KeyFmt kf, uint64_t drop, uint64_t stash, int32_t* __restrict__ cl,
                                                Slim* __restrict__ out, const uint32_t i, const uint32_t e, const int lane, MsCount mc) {
  for (uint32_t k = i + lane; k < e; k += 64) st_i32(&cl[k], -1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  uint32_t a = i;
  while (a < e) {
    const Slim s1 = ld_slim(s + a);
    const int32_t p1 = kf.pos(s1.key);
    const bool del = (s1.idx & SL_DEL) != 0;
    if (lane == 0) st_i32(&cl[a], (int32_t)a);
    uint64_t best = ((uint64_t)(uint32_t)s1.svlen << 32) | (0xFFFFFFFFu - a);   // max length, then lowest index
    for (uint32_t b0 = a + 1; b0 < e; b0 += 64) {
      const uint32_t b = b0 + lane;
      const bool valid = b < e;
      Slim s2 = s1;
      if (valid) s2 = ld_slim(s + b);
      const int32_t p2 = kf.pos(s2.key);
      const bool inwin = valid && (int64_t)p2 - p1 <= max_shift;
      if (__ballot(inwin) == 0) break;          // positions ascend: the whole tile is beyond the window
      if (inwin && ld_i32(&cl[b]) == -1 && sl_match(p1, s1.svlen, p2, s2.svlen, del, max_shift)) {
        st_i32(&cl[b], (int32_t)a);
        const uint64_t c = ((uint64_t)(uint32_t)s2.svlen << 32) | (0xFFFFFFFFu - b);
        if (c > best) best = c;
      }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const uint64_t o = __shfl_xor(best, d, 64); if (o > best) best = o; }
    if (lane == 0) {
      Slim rep = ld_slim(s + (0xFFFFFFFFu - (uint32_t)best)); rep.key = sl_stash(rep.key, drop, stash); st_slim(out + a, rep);
      if (mc.tcnt) mc.add(a / (uint32_t)MS_T, (rep.key & SL_CLASS) ? 1u : 0u, 1u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t nxt = e;                            // next seed: first unassigned element after a
    for (uint32_t b0 = a + 1; b0 < e; b0 += 64) {
      const uint32_t b = b0 + lane;
      const uint64_t free_m = __ballot(b < e && ld_i32(&cl[b]) == -1);
      if (free_m) { nxt = b0 + (uint32_t)__builtin_ctzll(free_m); break; }
    }
    a = nxt;
  }
  for (uint32_t k = i + lane; k < e; k += 64) if (ld_i32(&cl[k]) != (int32_t)k) st_slim(out + k, dead_slim());
}

// The kernel is bound by the instructions it issues (a lane per run with data-dependent loops keeps a wave busy for its longest run
// and pays scalar mask bookkeeping per trip), so the tile is staged pre-digested — position, list, svlen, index as four LDS arrays —
// and the work is split by run length: every lane looks eight elements ahead of a run's first element in straight-line code and
// files the run under "at most 4", "at most 8" or "longer"; the first two classes (95 % of the runs of a pile, all of them in
// sparse data) are then clustered by fully unrolled code on registers, dense lanes first; only the rest takes the loops.
template <int N>
__device__ __forceinline__ void sl_cluster_small(const int32_t* __restrict__ l_pos, const int32_t* __restrict__ l_len, const uint32_t* __restrict__ l_lid,
                                                 const uint32_t* __restrict__ l_idx, uint32_t rel0, uint32_t len, uint32_t i, int max_shift, KeyFmt kf, uint64_t drop,
                                                 uint64_t stash, Slim* __restrict__ out, uint32_t tb, uint32_t (&cn)[4]) {
  int32_t p[N], v[N];
#pragma unroll
  for (int u = 0; u < N; ++u) { p[u] = l_pos[rel0 + u]; v[u] = l_len[rel0 + u]; }
  const bool del = (l_idx[rel0] & SL_DEL) != 0;
  const uint32_t lid = l_lid[rel0];
  uint32_t assigned = ~((1u << len) - 1u);          // elements behind the run never join
#pragma unroll
  for (int a = 0; a < N; ++a) {
    const bool seed = !((assigned >> a) & 1u);
    int best = a;
    int32_t best_len = v[a];
#pragma unroll
    for (int b = a + 1; b < N; ++b) {
      const bool m = seed && !((assigned >> b) & 1u) && sl_match(p[a], v[a], p[b], v[b], del, max_shift);     // (shift > max_shift fails the match)
      assigned |= (m ? 1u : 0u) << b;
      if (m && v[b] > best_len) { best = b; best_len = v[b]; }
    }
    if ((uint32_t)a < len) {
      Slim r = dead_slim();
      if (seed) {
        r.key = sl_stash((((uint64_t)lid) << kf.pb) | vsv_kpos(l_pos[rel0 + best]), drop, stash);
        r.svlen = best_len;
        r.idx = l_idx[rel0 + best];
        const uint32_t d = ((i + (uint32_t)a) / (uint32_t)MS_T) - tb, cb = (r.key & SL_CLASS) ? 1u : 0u;     // (a run of <= 8 slots ends in this tile or the next)
        cn[0] += (d == 0u && cb == 0u) ? 1u : 0u; cn[1] += (d == 0u && cb != 0u) ? 1u : 0u;
        cn[2] += (d != 0u && cb == 0u) ? 1u : 0u; cn[3] += (d != 0u && cb != 0u) ? 1u : 0u;
      }
      st_slim(out + i + a, r);
    }
  }
}

__global__ __launch_bounds__(256) void sl_cluster(const Slim* __restrict__ s, const uint32_t* __restrict__ d_n, int max_shift, KeyFmt kf, int drop_bit,
                                                  uint64_t stash, Slim* __restrict__ out, int32_t* __restrict__ cl, MsCount mc) {
  constexpr int LDS_N = CL_TILE + CL_HALO + 1;
  __shared__ uint32_t lcnt[4];                                    // live outputs of this tile's runs: [this sort tile | the next][class]
  __shared__ uint16_t mid_i[128];                                 // runs of 9 .. SL_LONG_RUN elements: tile-relative first slot, length
  __shared__ uint8_t mid_len[128];
  __shared__ uint32_t nmid;
  __shared__ int32_t l_pos[LDS_N], l_len[LDS_N];
  __shared__ uint32_t l_lid[LDS_N], l_idx[LDS_N];
  __shared__ uint16_t h4[CL_TILE], h8[CL_TILE], hx[CL_TILE];      // run heads (tile-relative slot) by class
  __shared__ uint32_t n4, n8, nx;
  const uint32_t n = *d_n;
  const int lane = threadIdx.x & 63;
  const uint64_t drop = 1ull << drop_bit;
  const uint32_t ntiles = (n + CL_TILE - 1) / CL_TILE;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint32_t t0 = tile * CL_TILE;
    __syncthreads();
    if (threadIdx.x == 0) { n4 = 0; n8 = 0; nx = 0; nmid = 0; }
    if (threadIdx.x < 4) lcnt[threadIdx.x] = 0;
    const uint32_t tb = t0 / (uint32_t)MS_T;
    uint32_t cn[4] = {0, 0, 0, 0};
    for (int k = threadIdx.x; k < LDS_N; k += 256) {          // LDS index k = slot t0 - 1 + k
      const int64_t slot = (int64_t)t0 - 1 + k;
      if (slot >= 0 && slot < (int64_t)n) {
        const Slim x = ld_slim(s + slot);
        l_pos[k] = kf.pos(x.key); l_len[k] = x.svlen; l_lid[k] = (uint32_t)(x.key >> kf.pb); l_idx[k] = x.idx;
      } else { l_pos[k] = 0; l_len[k] = 0; l_lid[k] = 0xFFFFFFFFu; l_idx[k] = 0; }     // (no list has this id: it ends every run)
    }
    __syncthreads();
    // ---- run heads, and how far their runs go (eight elements of look-ahead, straight-line) ----
#pragma unroll 1
    for (int r = 0; r < CL_TILE / 256; ++r) {
      const uint32_t rel = (uint32_t)r * 256u + threadIdx.x + 1u, i = t0 + rel - 1u;
      if (i >= n) continue;
      const uint32_t lid = l_lid[rel];
      const bool head = !(l_lid[rel - 1] == lid && l_pos[rel] - l_pos[rel - 1] <= max_shift) || i == 0;
      if (!head) continue;
      uint32_t cont = 0;
#pragma unroll
      for (int u = 1; u <= 8; ++u)
        cont |= (l_lid[rel + u] == lid && l_pos[rel + u] - l_pos[rel + u - 1] <= max_shift ? 1u : 0u) << (u - 1);
      const uint32_t len = 1u + (uint32_t)__builtin_ctz(~cont | 0x100u);          // 1..9 (9: the run goes on)
      if (len <= 4u) h4[atomicAdd(&n4, 1u)] = (uint16_t)((rel - 1u) | ((len - 1u) << 12));
      else if (len <= 8u) h8[atomicAdd(&n8, 1u)] = (uint16_t)((rel - 1u) | ((len - 5u) << 12));
      else hx[atomicAdd(&nx, 1u)] = (uint16_t)(rel - 1u);
    }
    __syncthreads();
    // ---- runs of at most 4 and at most 8 elements: unrolled greedy on registers ----
    for (uint32_t h = threadIdx.x; h < n4; h += 256) {
      const uint32_t w = h4[h], r0 = w & 0xFFFu, len = (w >> 12) + 1u;
      sl_cluster_small<4>(l_pos, l_len, l_lid, l_idx, r0 + 1u, len, t0 + r0, max_shift, kf, drop, stash, out, tb, cn);
    }
    for (uint32_t h = threadIdx.x; h < n8; h += 256) {
      const uint32_t w = h8[h], r0 = w & 0xFFFu, len = (w >> 12) + 5u;
      sl_cluster_small<8>(l_pos, l_len, l_lid, l_idx, r0 + 1u, len, t0 + r0, max_shift, kf, drop, stash, out, tb, cn);
    }
    // ---- longer runs: the sequential greedy with loops; beyond SL_LONG_RUN elements the lane's whole wave ----
    auto get = [&](uint32_t k) -> Slim {            // element k: from LDS when the block staged it
      const uint32_t rel = k + 1u - t0;
      if (rel < (uint32_t)LDS_N && l_lid[rel] != 0xFFFFFFFFu) {
        Slim x;
        x.key = ((uint64_t)l_lid[rel] << kf.pb) | vsv_kpos(l_pos[rel]); x.svlen = l_len[rel]; x.idx = l_idx[rel];
        return x;
      }
      return ld_slim(s + k);
    };
    const uint32_t nxx = nx;
    for (uint32_t h0 = 0; h0 < nxx; h0 += 256) {               // (block-uniform trip count: the wave take-over below needs whole waves)
      const uint32_t h = h0 + threadIdx.x;
      bool is_long = false;
      uint32_t i = 0, e = 0;
      if (h < nxx) {
        i = t0 + hx[h];
        const Slim me = get(i);
        const uint64_t lk = me.key >> kf.pb;
        e = i + 1;
        int32_t last = kf.pos(me.key);
        while (e < n && e - i <= SL_LONG_RUN) {
          const Slim x = get(e);
          const int32_t px = kf.pos(x.key);
          if ((x.key >> kf.pb) != lk || (int64_t)px - last > max_shift) break;
          last = px; ++e;
        }
        if (e - i > SL_LONG_RUN) {
          is_long = true;
          while (e < n) {
            const Slim x = get(e);
            const int32_t px = kf.pos(x.key);
            if ((x.key >> kf.pb) != lk || (int64_t)px - last > max_shift) break;
            last = px; ++e;
          }
        } else {                                       // 9 .. SL_LONG_RUN elements, all of them staged: filed for the 16-lane groups below
          const uint32_t q = atomicAdd(&nmid, 1u);
          mid_i[q] = (uint16_t)(i - t0); mid_len[q] = (uint8_t)(e - i);
        }
      }
      uint64_t lm = __ballot(is_long);
      while (lm) {
        const int src = __builtin_ctzll(lm);
        lm &= lm - 1;
        sl_cluster_long(s, max_shift, kf, drop, stash, cl, out, (uint32_t)__shfl((int)i, src, 64), (uint32_t)__shfl((int)e, src, 64), lane, mc);
      }
    }
    // ---- runs of 9 .. SL_LONG_RUN elements: a lane per run spent ~L^2 / 2 match evaluations one after the other while its wave
    // waited (12 % of the pile's elements, 40 % of the kernel's time). Sixteen lanes per run instead, four runs per wave: the seeds stay
    // sequential (H:236-247), a seed's members are tested sixteen at a time, the assigned set is a 64-bit mask every lane of the group
    // keeps, the representative (first longest member) a packed maximum over the group. Loops and ballots are wave-uniform.
    __syncthreads();
    {
      const uint32_t nm = nmid;
      const int g = lane >> 4, gl = lane & 15, wv = threadIdx.x >> 6;
      for (uint32_t base = (uint32_t)wv * 4u; base < nm; base += 16u) {
        const uint32_t ridx = base + (uint32_t)g;
        const bool has = ridx < nm;
        const uint32_t r0 = has ? mid_i[ridx] : 0u, len = has ? mid_len[ridx] : 0u;
        uint32_t maxlen = len;
        maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, 16, 64));
        maxlen = max(maxlen, (uint32_t)__shfl_xor((int)maxlen, 32, 64));
        const uint32_t rel0 = r0 + 1u;                             // (LDS index k = slot t0 - 1 + k)
        const bool del = (l_idx[rel0] & SL_DEL) != 0;
        const uint32_t lid = l_lid[rel0];
        uint64_t assigned = 0;
        for (uint32_t a = 0; a < maxlen; ++a) {
          const bool act = a < len;
          const bool seed = act && !((assigned >> a) & 1ull);
          const uint32_t ra = rel0 + (act ? a : 0u);
          const int32_t p1 = l_pos[ra], v1 = l_len[ra];
          uint64_t best = ((uint64_t)(uint32_t)v1 << 32) | (uint64_t)(0xFFFFFFFFu - a);          // max length, then lowest index
          const uint32_t rounds = __ballot(seed) ? (maxlen - a - 1u + 15u) / 16u : 0u;          // (wave-uniform: no group holds a seed here)
          for (uint32_t q = 0; q < rounds; ++q) {
            const uint32_t b = a + 1u + q * 16u + (uint32_t)gl;
            bool m = false;
            int32_t v2 = 0;
            if (seed && b < len && !((assigned >> b) & 1ull)) {
              v2 = l_len[rel0 + b];
              m = sl_match(p1, v1, l_pos[rel0 + b], v2, del, max_shift);       // (a member beyond the shift fails the match)
            }
            const uint64_t bits = (__ballot(m) >> (g * 16)) & 0xFFFFull;
            assigned |= bits << (a + 1u + q * 16u);                          // (< 64: a run holds at most SL_LONG_RUN elements)
            if (m) { const uint64_t c = ((uint64_t)(uint32_t)v2 << 32) | (uint64_t)(0xFFFFFFFFu - b); if (c > best) best = c; }
          }
          if (rounds) {
#pragma unroll
            for (int d = 8; d > 0; d >>= 1) { const uint64_t o = __shfl_xor(best, d, 64); if (o > best) best = o; }
          }
          if (gl == 0 && act) {
            Slim r = dead_slim();
            if (seed) {
              const uint32_t rb = rel0 + (0xFFFFFFFFu - (uint32_t)best);
              r.key = sl_stash((((uint64_t)lid) << kf.pb) | vsv_kpos(l_pos[rb]), drop, stash);
              r.svlen = l_len[rb]; r.idx = l_idx[rb];
              const uint32_t d = (t0 + r0 + a) / (uint32_t)MS_T - tb, cb = (r.key & SL_CLASS) ? 1u : 0u;     // (at most SL_LONG_RUN slots behind the tile)
              cn[0] += (d == 0u && cb == 0u) ? 1u : 0u; cn[1] += (d == 0u && cb != 0u) ? 1u : 0u;
              cn[2] += (d != 0u && cb == 0u) ? 1u : 0u; cn[3] += (d != 0u && cb != 0u) ? 1u : 0u;
            }
            st_slim(out + t0 + r0 + a, r);
          }
        }
      }
    }
    if (mc.tcnt) {                                                // (block-uniform)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t v = cn[q];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
        if (lane == 0 && v) atomicAdd(&lcnt[q], v);
      }
      __syncthreads();
      if (threadIdx.x < 4 && lcnt[threadIdx.x]) mc.add(tb + (threadIdx.x >> 1), threadIdx.x & 1u, lcnt[threadIdx.x]);
    }
  }
}

// ---- haplotype pairing (pair_sig, H:548-592) on the merged elements --------------------------------------------------------------
// Merged table m sorted by (tid, hap, pos). The reference walks the hp1 rows in order and gives each the first free hp2 row of its
// type within pair_shift that matches — first come, first served. Only rows of one type compete for an hp2 row, so the walk
// decomposes by type, and inside a type it can be cut between two consecutive hp1 rows a < b wherever no hp2 row of the type lies
// in [pos_b - shift, pos_a + right]: whatever the rows up to a can reach (pos <= pos_a + right) lies in front of whatever the rows
// from b on can reach (pos >= pos_b - shift). Such a stretch is walked by one lane with the sequential rule (sl_pair_lds, below).
// sl_pair_prep finds every hp1 row's first candidate (jlo) for the round-based pairing of dense piles and looks for giant stretches.
// st[]: hp1 slot -> its mate's slot or -1; hp2 slot -> the hp1 slot that took it or -1.
__device__ __forceinline__ uint32_t sl_lower_bound(const Slim* __restrict__ m, uint32_t n, uint64_t target) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (m[mid].key >= target) hi = mid; else lo = mid + 1; }
  return lo;
}
constexpr uint32_t PJ_HEAD = 0x80000000u;

// The kernel also looks for GIANT stretches (a pile of contigs so dense that a type's rows go on for thousands of rows without a
// cut): a window of 2048 consecutive slots that holds >= 512 hp1 rows of a type and no stretch start among them reports
// max_stretch = 4096, which makes the handle pair its NEXT run in rounds (and a window with starts everywhere lets it go back).
constexpr uint32_t PJ_WINDOW = 2048;
__global__ __launch_bounds__(256) void sl_pair_prep(const Slim* __restrict__ m, const uint32_t* __restrict__ d_n, KeyFmt kf, int pair_shift, int right,
                                                    uint32_t* __restrict__ hj, int32_t* __restrict__ st, uint32_t* __restrict__ max_stretch) {
  __shared__ uint32_t rows_t[2], heads_t[2];
  const uint32_t n = *d_n;
  const int sh_hap = kf.pb + 2, sh_tid = kf.pb + 3;
  for (uint32_t base = blockIdx.x * PJ_WINDOW; base < n; base += gridDim.x * PJ_WINDOW) {
    if (threadIdx.x < 2) { rows_t[threadIdx.x] = 0; heads_t[threadIdx.x] = 0; }
    __syncthreads();
    for (uint32_t k8 = 0; k8 < PJ_WINDOW / 256; ++k8) {
      const uint32_t i = base + k8 * 256u + threadIdx.x;
      if (i >= n) continue;
      const Slim me = ld_slim(m + i);
      st[i] = -1;
      if ((me.key >> sh_hap) & 1ull) { hj[i] = 0; continue; }
      const int32_t pb_ = kf.pos(me.key);
      const uint64_t tk = me.key >> sh_tid;
      const uint32_t jlo = sl_lower_bound(m, n, (tk << sh_tid) | (1ull << sh_hap) | vsv_kpos((int32_t)max((int64_t)pb_ - pair_shift, (int64_t)-VSV_POS_BIAS)));
      bool head = true;
      const uint32_t t = me.idx & SL_DEL;
      const uint64_t hp1_prefix = me.key >> sh_hap, hp2_prefix = hp1_prefix | 1ull;
      for (uint32_t k = i; k-- > 0;) {
        const Slim x = ld_slim(m + k);
        if ((x.key >> sh_hap) != hp1_prefix) break;
        const int32_t pa = kf.pos(x.key);
        if ((int64_t)pb_ - pa > (int64_t)pair_shift + right) break;      // nothing earlier shares a candidate with this row
        if ((x.idx & SL_DEL) != t) continue;
        // the previous row of this type: the walk can be cut here iff no hp2 row of the type lies in [pos_b - shift, pos_a + right]
        for (uint32_t j = jlo; j < n; ++j) {
          const Slim y = ld_slim(m + j);
          if ((y.key >> sh_hap) != hp2_prefix || (int64_t)kf.pos(y.key) - pa > right) break;
          if ((y.idx & SL_DEL) == t) { head = false; break; }
        }
        break;
      }
      hj[i] = jlo | (head ? PJ_HEAD : 0u);
      atomicAdd(&rows_t[t ? 1 : 0], 1u);
      if (head) atomicAdd(&heads_t[t ? 1 : 0], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 2 && rows_t[threadIdx.x] >= 512u && heads_t[threadIdx.x] == 0u) atomicMax(max_stretch, 4096u);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void sl_fill_i32(int32_t* __restrict__ p, int32_t v, const uint32_t* __restrict__ d_n) {
  const uint32_t n = *d_n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}

// ---- the walk from LDS -----------------------------------------------------------------------------------------------------
// The walks are bound by instructions issued, not by bytes: a lane per stretch makes its wave wait for the longest stretch, a group of
// lanes per stretch spends ~500 wave instructions per row. sl_pair_lds gives ONE wave a window of 256 slots: the window's hp1
// rows (+ halo) and the hp2 rows they can reach are staged in LDS as (pos, svlen | type) pairs — 8 bytes per row, list membership
// resolved at staging time — then (1) every hp1 row gets its first candidate by bisection in LDS and its stretch-start test, the
// starts go to a list; (2) the lanes take stretches from the list until it is empty (a lane that finishes a short stretch takes the
// next one: the wave is not held up by its longest stretch) and walk them a row per step: one sweep over the row's window advances
// the candidate pointer, decides the cut in front of the row (no hp2 row of the type at or in front of prev + right) and finds the
// mate. Rows outside the staged windows are read from global memory by the same accessors. st[] must be -1 before the launch.
constexpr int QW = 256, QW_BACK = 64, QW_FWD = 64, QA = QW_BACK + QW + QW_FWD, QB = 512;

// first index in [0, n) whose key is >= target, by the 64 lanes of a wave: 65-ary steps (4 dependent loads for 10^7 rows)
__device__ __forceinline__ uint32_t sl_wave_lower_bound(const Slim* __restrict__ m, uint32_t n, uint64_t target, int lane) {
  uint32_t lo = 0, hi = n;                        // the answer lies in [lo, hi]; keys in front of lo are < target, keys from hi on >= target
  while (hi - lo > 64u) {
    const uint64_t span = (uint64_t)(hi - lo);
    const uint32_t probe = lo + (uint32_t)(span * (uint32_t)(lane + 1) / 65u);      // lo <= probe < hi, ascending with the lane
    const uint64_t bal = __ballot(m[probe].key >= target);
    const int f = bal ? __builtin_ctzll(bal) : 64;        // lanes below f hold keys < target, lane f a key >= target
    const uint32_t nlo = f == 0 ? lo : (lo + (uint32_t)(span * (uint32_t)f / 65u)) + 1u;
    const uint32_t nhi = f == 64 ? hi : lo + (uint32_t)(span * (uint32_t)(f + 1) / 65u);
    lo = nlo; hi = nhi;
  }
  const uint32_t i = lo + (uint32_t)lane;
  const uint64_t bal = __ballot(i < hi && m[i].key >= target);
  return bal ? lo + (uint32_t)__builtin_ctzll(bal) : hi;
}

__global__ __launch_bounds__(64) void sl_pair_lds(const Slim* __restrict__ m, const uint32_t* __restrict__ d_n, KeyFmt kf, int pair_shift, int right,
                                                  int32_t* __restrict__ st, const uint32_t* __restrict__ widx, uint32_t* __restrict__ max_stretch) {
  __shared__ int32_t a_pos[QA], b_pos[QB + 8];
  __shared__ uint32_t a_lt[QA], b_lt[QB + 8];        // svlen | type << 31
  __shared__ uint32_t a_pre[QA];                     // list of the slot: key >> hap bit (tid, hap)
  __shared__ uint8_t b_taken[QB + 8];
  __shared__ uint32_t h_a[QW], h_j[QW];
  __shared__ uint32_t n_heads, next_head, rows_t[2], heads_t[2];
  const uint32_t n = *d_n;
  const int lane = threadIdx.x;
  const int sh_hap = kf.pb + 2;
  const int64_t reach = (int64_t)pair_shift + right;
  for (uint32_t base = blockIdx.x * QW; base < n; base += gridDim.x * QW) {
    const uint32_t w_hi = min(n, base + QW);
    // ---- stage the window's slots (+ halo): one round trip, everything else about them is decided from LDS ----
    const uint32_t a_lo = base >= (uint32_t)QW_BACK ? base - QW_BACK : 0u, a_hi = min(n, w_hi + QW_FWD), a_cnt = a_hi - a_lo;
    __syncthreads();
    for (uint32_t r = lane; r < a_cnt; r += 64) {
      const Slim x = ld_slim(m + a_lo + r);
      a_pos[r] = kf.pos(x.key);
      a_lt[r] = ((uint32_t)x.svlen & 0x7FFFFFFFu) | (x.idx & SL_DEL);
      a_pre[r] = (uint32_t)(x.key >> sh_hap);
    }
    __syncthreads();
    uint32_t sub = base;
    while (sub < w_hi) {
      // ---- the next run of hp1 rows of one list inside the window: [s_lo, s_hi) ----
      uint32_t s_lo = w_hi, s_hi = w_hi;
      for (uint32_t q = sub; q < w_hi && s_lo == w_hi; q += 64) {
        const uint32_t k = q + lane;
        const uint64_t bal = __ballot(k < w_hi && !(a_pre[k - a_lo] & 1u));
        if (bal) s_lo = q + (uint32_t)__builtin_ctzll(bal);
      }
      if (s_lo >= w_hi) break;
      const uint32_t P1 = a_pre[s_lo - a_lo], P2 = P1 | 1u;
      for (uint32_t q = s_lo; q < w_hi && s_hi == w_hi; q += 64) {
        const uint32_t k = q + lane;
        const uint64_t bal = __ballot(k < w_hi && a_pre[k - a_lo] != P1);
        if (bal) s_hi = q + (uint32_t)__builtin_ctzll(bal);
      }
      sub = s_hi;
      // rows of list P1 among the staged slots: [a_vlo, a_vhi) (the list is contiguous)
      uint32_t a_vlo = s_lo, a_vhi = s_hi;
      for (uint32_t q = 0; q < a_cnt; q += 64) {
        const uint32_t r = q + lane;
        const uint64_t bal = __ballot(r < a_cnt && a_pre[r] == P1);
        if (bal) {
          const uint32_t first = a_lo + q + (uint32_t)__builtin_ctzll(bal), last = a_lo + q + 63u - (uint32_t)__builtin_clzll(bal);
          if (first < a_vlo) a_vlo = first;
          if (last + 1 > a_vhi) a_vhi = last + 1;
        }
      }
      // ---- the hp2 rows the staged rows can reach start at j0 (precomputed for the window's first staged slot); QB rows from there ----
      uint32_t j0;
      if (a_vlo == a_lo && widx) j0 = widx[base / QW];
      else {
        const int64_t p_first = (int64_t)a_pos[a_vlo - a_lo] - pair_shift;
        j0 = sl_wave_lower_bound(m, n, ((uint64_t)P2 << sh_hap) | vsv_kpos((int32_t)max(p_first, (int64_t)-VSV_POS_BIAS)), lane);
      }
      __syncthreads();
      if (lane < 2) { rows_t[lane] = 0; heads_t[lane] = 0; }
      if (lane == 0) { n_heads = 0; next_head = 0; }
      uint32_t b_cnt = QB;                           // leading staged rows that belong to list P2
      for (uint32_t q = 0; q < (uint32_t)QB; q += 64) {
        const uint32_t r = q + lane, j = j0 + r;
        bool ok = j < n;
        if (ok) {
          const Slim y = ld_slim(m + j);
          ok = (uint32_t)(y.key >> sh_hap) == P2;
          b_pos[r] = kf.pos(y.key);
          b_lt[r] = ((uint32_t)y.svlen & 0x7FFFFFFFu) | (y.idx & SL_DEL);
          b_taken[r] = 0;
        }
        const uint64_t bad = ~__ballot(ok);
        if (bad && b_cnt == (uint32_t)QB) b_cnt = q + (uint32_t)__builtin_ctzll(bad);
      }
      const bool clipped = b_cnt == (uint32_t)QB;    // the list may go on behind the staged rows (else it ends there)
      // a list that ends inside the staged rows is followed by eight rows "beyond every window": the batched sweep may read past its end
      if (!clipped && lane < 8) { b_pos[b_cnt + lane] = 0x7FFFFFFF; b_lt[b_cnt + lane] = 0; b_taken[b_cnt + lane] = 1; }
      const uint32_t b_pad = clipped ? b_cnt : b_cnt + 8u;
      __syncthreads();
      auto getA = [&](uint32_t k, int32_t& p, uint32_t& lt) -> bool {          // row k of list P1?
        const uint32_t rel = k - a_lo;
        if (rel < a_cnt) { p = a_pos[rel]; lt = a_lt[rel]; return k >= a_vlo && k < a_vhi; }
        if (k >= n) return false;
        const Slim x = ld_slim(m + k);
        p = kf.pos(x.key); lt = ((uint32_t)x.svlen & 0x7FFFFFFFu) | (x.idx & SL_DEL);
        return (uint32_t)(x.key >> sh_hap) == P1;
      };
      auto getB = [&](uint32_t j, int32_t& p, uint32_t& lt) -> bool {          // hp2 row j of list P2?
        const uint32_t rel = j - j0;
        if (rel < b_cnt) { p = b_pos[rel]; lt = b_lt[rel]; return true; }
        if (!clipped || j >= n) return false;
        const Slim y = ld_slim(m + j);
        p = kf.pos(y.key); lt = ((uint32_t)y.svlen & 0x7FFFFFFFu) | (y.idx & SL_DEL);
        return (uint32_t)(y.key >> sh_hap) == P2;
      };
      // ---- phase 1: first candidate and stretch-start test of the window's rows ----
      for (uint32_t i = s_lo + lane; i < s_hi; i += 64) {
        const int32_t pb_ = a_pos[i - a_lo];
        const uint32_t t = a_lt[i - a_lo] & SL_DEL;
        const int32_t tp = (int32_t)max((int64_t)pb_ - pair_shift, (int64_t)-VSV_POS_BIAS);
        uint32_t jlo;
        if (b_cnt == 0) jlo = j0;
        else if (b_pos[0] >= tp) jlo = j0;                                           // (j0 is the answer for the smallest target among the staged rows)
        else if (b_pos[b_cnt - 1] >= tp) {
          uint32_t lo = 0, hi = b_cnt - 1;                                           // pos[lo] < tp <= pos[hi]
          while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (b_pos[mid] >= tp) hi = mid; else lo = mid; }
          jlo = j0 + hi;
        } else if (!clipped) jlo = j0 + b_cnt;                                       // behind the list's last row
        else jlo = sl_lower_bound(m, n, ((uint64_t)P2 << sh_hap) | vsv_kpos(tp));
        bool head = true;
        for (uint32_t k = i; k-- > 0;) {
          int32_t pa; uint32_t lt;
          if (!getA(k, pa, lt)) break;
          if ((int64_t)pb_ - pa > reach) break;                          // nothing earlier shares a candidate with this row
          if ((lt & SL_DEL) != t) continue;
          for (uint32_t j = jlo;; ++j) {                                 // an hp2 row of the type in [pos_b - shift, pos_a + right]?
            int32_t py; uint32_t ly;
            if (!getB(j, py, ly) || py - pa > right) break;
            if ((ly & SL_DEL) == t) { head = false; break; }
          }
          break;
        }
        atomicAdd(&rows_t[t ? 1 : 0], 1u);
        if (head) { atomicAdd(&heads_t[t ? 1 : 0], 1u); const uint32_t s = atomicAdd(&n_heads, 1u); h_a[s] = i; h_j[s] = jlo; }
      }
      __syncthreads();
      if (lane < 2 && rows_t[lane] >= 96u && heads_t[lane] == 0u) atomicMax(max_stretch, 4096u);        // a giant stretch passes through
      // ---- phase 2: lanes take stretches until the list is empty; a row per step ----
      const uint32_t nh = n_heads;
      bool active = false;
      uint32_t a = 0, jl = 0, cnt = 0, t = 0, l1 = 0;
      int32_t p1 = 0, prev = 0;
      for (;;) {
        if (!active) {
          const uint32_t h = atomicAdd(&next_head, 1u);
          if (h < nh) {
            a = h_a[h]; jl = h_j[h]; cnt = 0;
            p1 = a_pos[a - a_lo]; l1 = a_lt[a - a_lo]; t = l1 & SL_DEL;
            active = true;
          }
        }
        if (!__ballot(active)) break;
        if (!active) continue;
        // One sweep over the row's window: rows in front of pos - shift are passed for good, then the cut test, then the mate.
        // The kernel is bound by instructions ISSUED (divergent loops cost scalar mask bookkeeping per trip), so the sweep takes the
        // staged candidates eight at a time in straight-line code: five predicate masks, first-set-bit arithmetic, no branch per
        // candidate. (Positions ascend: "passed" is a prefix of the batch, "beyond" and "past" are suffixes.)
        bool seen_t = cnt == 0, cut = false, done = false;
        int32_t mate = -1;
        const int32_t v1 = (int32_t)(l1 & 0x7FFFFFFFu);
        while (!done && jl - j0 + 8u <= b_pad) {
          const uint32_t jb = jl, r0 = jb - j0;
          uint32_t m_passed = 0, m_beyond = 0, m_past = 0, m_t = 0, m_ok = 0;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int32_t p2 = b_pos[r0 + u];
            const uint32_t l2 = b_lt[r0 + u];
            const bool ist = (l2 & SL_DEL) == t;
            m_passed |= (p1 - p2 > pair_shift ? 1u : 0u) << u;
            m_beyond |= (p2 - p1 > right ? 1u : 0u) << u;
            m_past |= (p2 - prev > right ? 1u : 0u) << u;
            m_t |= (ist ? 1u : 0u) << u;
            m_ok |= (ist && b_taken[r0 + u] == 0 && sl_match(p1, v1, p2, (int32_t)(l2 & 0x7FFFFFFFu), t != 0, pair_shift) ? 1u : 0u) << u;
          }
          const uint32_t np = (uint32_t)__builtin_ctz(~m_passed | 0x100u);           // rows in front of pos - shift: passed for good
          if (np == 8u) { jl = jb + 8u; continue; }
          jl = jb + np;
          const uint32_t e = (uint32_t)__builtin_ctz(m_beyond | 0x100u);
          const uint32_t live = ((1u << e) - 1u) & ~((1u << np) - 1u);              // the row's candidates in this batch
          const uint32_t ft = (uint32_t)__builtin_ctz((m_t & live) | 0x100u), fp = (uint32_t)__builtin_ctz((m_past & live) | 0x100u);
          const uint32_t fo = (uint32_t)__builtin_ctz((m_ok & live) | 0x100u);
          if (!seen_t) {
            if (fp < 8u && fp <= ft) cut = true;                  // no row of the type at or in front of prev + right
            else if (ft < 8u) seen_t = true;
          }
          done = true;
          if (cut) break;
          if (fo < 8u) { mate = (int32_t)(jb + fo); break; }                                // H:560-569
          if (e < 8u) break;                                      // the window ended inside the batch
          for (uint32_t j = jb + 8u;; ++j) {                     // it goes on behind the batch: the rest of it, a row at a time
            int32_t p2; uint32_t l2;
            if (!getB(j, p2, l2)) break;
            if (!seen_t && p2 - prev > right) { cut = true; break; }
            if (p2 - p1 > right) break;
            if ((l2 & SL_DEL) != t) continue;
            seen_t = true;
            const uint32_t rel = j - j0;
            if ((rel < b_cnt ? b_taken[rel] == 0 : st[j] == -1) && sl_match(p1, v1, p2, (int32_t)(l2 & 0x7FFFFFFFu), t != 0, pair_shift)) { mate = (int32_t)j; break; }
          }
        }
        if (!done) {                                              // candidates outside the staged rows: the same sweep, a row at a time
          for (uint32_t j = jl;; ++j) {
            int32_t p2; uint32_t l2;
            if (!getB(j, p2, l2)) break;
            if (p1 - p2 > pair_shift) { jl = j + 1; continue; }
            if (!seen_t && p2 - prev > right) { cut = true; break; }
            if (p2 - p1 > right) break;
            if ((l2 & SL_DEL) != t) continue;
            seen_t = true;
            const uint32_t rel = j - j0;
            if ((rel < b_cnt ? b_taken[rel] == 0 : st[j] == -1) && sl_match(p1, v1, p2, (int32_t)(l2 & 0x7FFFFFFFu), t != 0, pair_shift)) { mate = (int32_t)j; break; }
          }
        }
        if (cut || !seen_t) {                         // no hp2 row of the type in [pos - shift, prev + right]: the row starts its own stretch
          if (cnt > 1024u) atomicMax(max_stretch, cnt);
          active = false;
          continue;
        }
        st[a] = mate;
        if (mate >= 0) { st[mate] = (int32_t)a; const uint32_t rel = (uint32_t)mate - j0; if (rel < b_cnt) b_taken[rel] = 1; }
        ++cnt; prev = p1;
        bool more = false, ended = false;             // the list's next row of this type: four staged slots at a time
        uint32_t k = a + 1;
        while (k - a_lo + 4u <= a_cnt) {
          const uint32_t r0 = k - a_lo;
          uint32_t m_stop = 0, m_t = 0;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            m_stop |= (a_pre[r0 + u] != P1 ? 1u : 0u) << u;
            m_t |= ((a_lt[r0 + u] & SL_DEL) == t ? 1u : 0u) << u;
          }
          const uint32_t fs = (uint32_t)__builtin_ctz(m_stop | 0x10u), ft = (uint32_t)__builtin_ctz(m_t | 0x10u);
          if (ft < fs) { a = k + ft; p1 = a_pos[r0 + ft]; l1 = a_lt[r0 + ft]; more = true; break; }
          if (fs < 4u) { ended = true; break; }
          k += 4u;
        }
        if (!more && !ended) {
          for (;; ++k) {
            int32_t pk; uint32_t lk;
            if (!getA(k, pk, lk)) break;
            if ((lk & SL_DEL) == t) { a = k; p1 = pk; l1 = lk; more = true; break; }
          }
        }
        if (!more) { if (cnt > 1024u) atomicMax(max_stretch, cnt); active = false; }
      }
    }
  }
}

// st[] = -1 for the pairing, and for every window of sl_pair_lds the first hp2 row its first staged slot can reach (one bisection per
// window here, in parallel, instead of a chain of dependent loads at the start of every window's wave)
__global__ __launch_bounds__(256) void sl_pair_index(const Slim* __restrict__ m, const uint32_t* __restrict__ d_n, KeyFmt kf, int pair_shift,
                                                     int32_t* __restrict__ st, uint32_t* __restrict__ widx) {
  const uint32_t n = *d_n;
  const int sh_hap = kf.pb + 2;
  const uint32_t nwin = (n + QW - 1) / QW;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    st[i] = -1;
    if (i < nwin) {
      const uint32_t base = i * QW, s = base >= (uint32_t)QW_BACK ? base - QW_BACK : 0u;
      const Slim x = ld_slim(m + s);
      uint32_t r = 0;
      if (!((x.key >> sh_hap) & 1ull))
        r = sl_lower_bound(m, n, (((x.key >> sh_hap) | 1ull) << sh_hap) | vsv_kpos((int32_t)max((int64_t)kf.pos(x.key) - pair_shift, (int64_t)-VSV_POS_BIAS)));
      widx[i] = r;
    }
  }
}

// ---- pairing in rounds: dense piles (deterministic reservations; see sig_stages.hip "Pairing in rounds") -------------------------
// Every undecided hp1 row reserves ALL its free candidates with its index as priority (atomicMin; the round in the high word so
// nothing needs clearing), then takes its FIRST free candidate if it holds the reservation there: no earlier undecided row can then
// ever want that hp2 row and the decided rows are final, so this is exactly what the sequential walk gives it; a row without a free
// candidate is unpaired for good. What the rounds leave goes through the sequential rule in chains (sl_pair_leftover).
__device__ __forceinline__ bool sl_candidate(const Slim& s1, int32_t p1, const Slim& s2, int32_t p2, int pair_shift) {
  return ((s1.idx ^ s2.idx) & SL_DEL) == 0 && sl_match(p1, s1.svlen, p2, s2.svlen, (s1.idx & SL_DEL) != 0, pair_shift);
}
template <bool COMMIT>
__global__ __launch_bounds__(256) void sl_pair_round(const Slim* __restrict__ m, const uint32_t* __restrict__ d_n, KeyFmt kf, int pair_shift, int right,
                                                     int32_t* __restrict__ st, const uint32_t* __restrict__ hj, uint32_t* __restrict__ done1,
                                                     uint64_t* __restrict__ res, uint32_t round) {
  const uint32_t n = *d_n;
  const int sh_hap = kf.pb + 2;
  const uint64_t stamp = (uint64_t)(~round) << 32;          // newer rounds compare smaller: old reservations lose by themselves
  for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a < n; a += gridDim.x * blockDim.x) {
    const Slim s1 = ld_slim(m + a);
    if (((s1.key >> sh_hap) & 1ull) || done1[a]) continue;
    const uint64_t hp2_prefix = (s1.key >> sh_hap) | 1ull;
    const int32_t p1 = kf.pos(s1.key);
    int32_t first = -1;
    for (uint32_t j = hj[a] & ~PJ_HEAD; j < n; ++j) {
      const Slim s2 = ld_slim(m + j);
      if ((s2.key >> sh_hap) != hp2_prefix) break;
      const int32_t p2 = kf.pos(s2.key);
      if ((int64_t)p2 - p1 > right) break;
      if (ld_i32(&st[j]) != -1 || !sl_candidate(s1, p1, s2, p2, pair_shift)) continue;
      if (!COMMIT) atomicMin((unsigned long long*)&res[j], (unsigned long long)(stamp | a));
      else { first = (int32_t)j; break; }
    }
    if (!COMMIT) continue;
    if (first < 0) { done1[a] = 1; }                                                  // H:575-576 (st[a] stays -1)
    else if (res[first] == (stamp | a)) { st_i32(&st[first], (int32_t)a); st_i32(&st[a], first); done1[a] = 1; }
  }
}
// (rounds count from 1: the reservation words start at all ones, which no round's stamp exceeds)
__global__ __launch_bounds__(256) void sl_pair_rounds_init(const uint32_t* __restrict__ d_n, uint32_t* __restrict__ done1, uint64_t* __restrict__ res) {
  const uint32_t n = *d_n;
  for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a < n; a += gridDim.x * blockDim.x) { done1[a] = 0; res[a] = ~0ull; }
}
__global__ __launch_bounds__(256) void sl_pair_leftover(const Slim* __restrict__ m, const uint32_t* __restrict__ d_n, KeyFmt kf, int pair_shift, int right,
                                                        int32_t* __restrict__ st, const uint32_t* __restrict__ hj, const uint32_t* __restrict__ done1) {
  const uint32_t n = *d_n;
  const int sh_hap = kf.pb + 2;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const Slim me = ld_slim(m + i);
    if (((me.key >> sh_hap) & 1ull) || done1[i]) continue;
    const uint64_t hp1_prefix = me.key >> sh_hap, hp2_prefix = hp1_prefix | 1ull;
    const int32_t pme = kf.pos(me.key);
    bool head = true;
    for (uint32_t k = i; k-- > 0;) {                        // an undecided hp1 row of this list within 2 * pair_shift in front?
      const Slim x = ld_slim(m + k);
      if ((x.key >> sh_hap) != hp1_prefix || (int64_t)pme - kf.pos(x.key) > 2 * (int64_t)pair_shift) break;
      if (!done1[k]) { head = false; break; }
    }
    if (!head) continue;
    int32_t last_pos = pme;
    for (uint32_t a = i; a < n; ++a) {
      const Slim s1 = ld_slim(m + a);
      if ((s1.key >> sh_hap) != hp1_prefix) break;
      const int32_t p1 = kf.pos(s1.key);
      if ((int64_t)p1 - last_pos > 2 * (int64_t)pair_shift) break;      // the next undecided row, if any, leads its own chain
      if (done1[a]) continue;
      last_pos = p1;
      int32_t mate = -1;
      for (uint32_t j = hj[a] & ~PJ_HEAD; j < n; ++j) {
        const Slim s2 = ld_slim(m + j);
        if ((s2.key >> sh_hap) != hp2_prefix) break;
        const int32_t p2 = kf.pos(s2.key);
        if ((int64_t)p2 - p1 > right) break;
        if (ld_i32(&st[j]) == -1 && sl_candidate(s1, p1, s2, p2, pair_shift)) { mate = (int32_t)j; st_i32(&st[j], (int32_t)a); break; }
      }
      st_i32(&st[a], mate);                  // (done1 stays as the rounds left it: the other lanes' head tests read it meanwhile)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
}

// ---- rows out ------------------------------------------------------------------------------------------------------------------
// sorted call elements -> vsv_call rows (H:571-592): the kept signature's row comes from the stage-1 input table
__global__ __launch_bounds__(256) void sl_calls_out(const Slim* __restrict__ cs, const uint32_t* __restrict__ d_n, const Slim* __restrict__ m, int hap_bit,
                                                    const vsv_sig* __restrict__ rows, vsv_call* __restrict__ out) {
  const uint32_t n = *d_n;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const Slim c = ld_slim(cs + k);
    const Slim me = ld_slim(m + c.idx);
    vsv_call o;
    o.pad = 0;
    uint32_t row = me.idx & SL_ROW;
    if ((me.key >> hap_bit) & 1ull) { o.a = -1; o.b = (int32_t)c.idx; o.gt = 1; }                  // H:588-592
    else {
      o.a = (int32_t)c.idx; o.b = c.svlen;
      if (c.svlen < 0) o.gt = 1;                                                                   // H:575-576
      else { o.gt = 2; const Slim mate = ld_slim(m + c.svlen); if (!(me.svlen > mate.svlen)) row = mate.idx & SL_ROW; }   // H:583-586
    }
    o.sig = rows[row];
    out[k] = o;
  }
}
// slim table -> signature rows (VSV_T_CLUSTER1 / VSV_T_MERGED on request): a dead element gives a dead row
__global__ __launch_bounds__(256) void sl_rows_out(const Slim* __restrict__ e, uint32_t n, const vsv_sig* __restrict__ rows, vsv_sig* __restrict__ out) {
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const Slim x = ld_slim(e + k);
    vsv_sig r;
    if (x.key == VSV_KEY_DEAD) { r.pos = 0; r.svlen = 0; r.q_start = 0; r.q_end = 0; r.rec = 0; r.rec2 = 0; r.meta = VSV_M_DEAD; r.tid = 0; }
    else r = rows[x.idx & SL_ROW];
    out[k] = r;
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
template <int BITS, typename SRC, bool SKIP>
void sl_pass(hipStream_t st, SRC src, const uint32_t* d_n, int shift, KeyCmp kc, SlElem<SRC>* out, uint32_t* hist, uint32_t* totals, uint32_t* d_total, int grid, bool wide) {
  const LsdDigit dig{shift, kc, (1u << BITS) - 1u};
  if (wide) sl_hist<BITS, SRC, SKIP, 16><<<grid, 1024, 0, st>>>(src, d_n, dig, hist, totals);
  else sl_hist<BITS, SRC, SKIP, 4><<<grid, 256, 0, st>>>(src, d_n, dig, hist, totals);
  sl_scan<BITS><<<(1 << BITS) / 16, 1024, 0, st>>>(hist, totals, d_n, d_total);
  if (wide) sl_scatter<BITS, SRC, SKIP, 16><<<grid, 1024, 0, st>>>(src, d_n, dig, hist, out);
  else sl_scatter<BITS, SRC, SKIP, 4><<<grid, 256, 0, st>>>(src, d_n, dig, hist, out);
}

// Stable sort of the live elements of `src` (n slots, *d_slots) by key bits [0, nbits): the first pass skips dead elements and
// publishes the live count (*d_live), the later passes run on that count. Ping-pong between a and b; returns the buffer that holds
// the result (nothing is copied back).
// digit plan of a sort on `nbits` significant bits: 8-bit passes, or 10-bit ones when that saves a pass (a 10-bit pass costs ~10 % more
// than an 8-bit one: 64-byte instead of 256-byte pieces per tile and digit, a four times larger histogram)
struct DigitPlan { int bits, passes; };
DigitPlan sl_plan(int nbits) {
  if (nbits < 1) nbits = 1;
  const int p8 = (nbits + 7) / 8, p10 = (nbits + 9) / 10;
  static const int force_bits = vsv_dbg_env("VSV_SLIM_BITS") ? atoi(vsv_dbg_env("VSV_SLIM_BITS")) : 0;     // timing experiments
  if (force_bits == 8) return DigitPlan{8, p8};
  if (force_bits == 10) return DigitPlan{10, p10};
  return p10 < p8 ? DigitPlan{10, p10} : DigitPlan{8, p8};
}
template <typename SRC>
SlElem<SRC>* sl_sort(hipStream_t st, SRC src, const uint32_t* d_slots, uint32_t* d_live, int nbits, KeyCmp kc, SlElem<SRC>* a, SlElem<SRC>* b, const SlimWork& w, int64_t hint) {
  using E = SlElem<SRC>;
  using SrcE = typename SrcOf<E>::type;
  const DigitPlan dp = sl_plan(nbits - kc.zbits);
  const int bits = dp.bits, passes = dp.passes;
  const int64_t max_tiles = (w.cap + SL_TILE - 1) / SL_TILE;
  const int grid = (int)(max_tiles < 2048 ? (max_tiles < 1 ? 1 : max_tiles) : 2048);
  // blocks of 16 waves where the previous run's tables say the tiles alone cannot fill the chip
  static const int force_waves = vsv_dbg_env("VSV_SLIM_WAVES") ? atoi(vsv_dbg_env("VSV_SLIM_WAVES")) : 0;     // timing experiments
  const bool wide = force_waves ? force_waves == 16 : hint <= (int64_t)2048 * SL_TILE;
  E* dst = a;
  E* other = b;
  for (int p = 0; p < passes; ++p) {
    uint32_t* totals = w.totals + (size_t)(*w.pass_cursor) * 2048;
    ++*w.pass_cursor;
    const int shift = p * bits;
    if (p == 0) {
      if (bits == 8) sl_pass<8, SRC, true>(st, src, d_slots, shift, kc, dst, w.hist, totals, d_live, grid, wide);
      else sl_pass<10, SRC, true>(st, src, d_slots, shift, kc, dst, w.hist, totals, d_live, grid, wide);
    } else {
      const SrcE in{other};
      if (bits == 8) sl_pass<8, SrcE, false>(st, in, d_live, shift, kc, dst, w.hist, totals, nullptr, grid, wide);
      else sl_pass<10, SrcE, false>(st, in, d_live, shift, kc, dst, w.hist, totals, nullptr, grid, wide);
    }
    E* t = dst; dst = other; other = t;
  }
  return other;        // the buffer the last pass wrote
}


// ---- the FIRST sort: one counting pass into position buckets + an LDS sort per bucket ----------------------------------------------
// The stage-1 elements arrive in (record, op) order, and a coordinate-sorted BAM's records ascend by position: a tile of 4096 of them
// lies within a few megabases, i.e. in a handful of the ~2000-4000 buckets a monotone map cuts the populated key range into — a list
// pair (tid, hap, type) owns 2^bp_log consecutive buckets, most of them for its cigar list, the others for its split list (by the
// share of split elements the handle expects), equally wide in position between the smallest and the largest position the table
// holds (sl_minmax: one read of the keys). So ONE stable counting pass (the kernels of the LSD passes with that map as the digit)
// writes long contiguous pieces, and every bucket — one list, a position range — is then sorted completely in LDS by position (keys
// relative to the bucket's minimum, stable passes of <= 9 bits) and written in order: five launches that move the table twice, where
// the four 8-bit passes take twelve and move it eight times. A bucket of more than SB_CAP elements (a
// pile far from uniform, a stale size hint) is sorted by the same block with the same passes on the elements themselves, ping-pong
// between the bucket's own ranges of the two buffers (sl_bucket_global: correct whatever the bucket holds, slow), and raises
// ERRB_BUCKET1_SLOW: no table is ever left unsorted, the handle just takes the LSD passes for its next runs.
constexpr int SB_CAP = 4096, SB_DBITS = 9;

// digit totals of the waves' chunks (wcnt[w][d], counted) -> first destination of every (wave, digit): digits ascending, waves
// ascending inside a digit (wave-major chunks: the input order inside equal digits is kept). Called by the whole block.
template <int SB_THREADS>
__device__ __forceinline__ void sb_bases(uint32_t (*wcnt)[1 << SB_DBITS], uint32_t* tot, uint32_t nbins) {
  constexpr int SB_WAVES = SB_THREADS / 64;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  uint32_t mine = 0, incl = 0;
  if ((uint32_t)t < nbins) {
#pragma unroll
    for (int w = 0; w < SB_WAVES; ++w) mine += wcnt[w][t];
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
    for (int w = 0; w < SB_WAVES; ++w) { const uint32_t c = wcnt[w][t]; wcnt[w][t] = run; run += c; }
  }
  __syncthreads();
}

// a bucket too large for LDS: the passes on global memory, a = the bucket's range of the counting pass's output (consumed),
// b = its range of the sorted table. Block-uniform arguments; any m.
template <int SB_THREADS>
__device__ void sl_bucket_global(Slim* __restrict__ a, Slim* __restrict__ b, uint32_t m, uint32_t pmask, uint32_t kmin, uint32_t kmax,
                                 uint32_t (*wcnt)[1 << SB_DBITS], uint32_t* tot) {
  constexpr int SB_WAVES = SB_THREADS / 64;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const uint32_t width = kmax - kmin;
  Slim* src = a; Slim* dst = b;
  if (width != 0) {
    const int wbits = 32 - __builtin_clz(width);
    const int passes = (wbits + SB_DBITS - 1) / SB_DBITS;
    const int db = (wbits + passes - 1) / passes;
    const uint32_t dmask = (1u << db) - 1u, nbins = 1u << db;
    const uint32_t per = ((m + SB_WAVES - 1) / SB_WAVES + 63u) & ~63u;
    const uint32_t c0 = min(m, (uint32_t)wv * per), c1 = min(m, c0 + per);
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int pass = 0, shift = 0; pass < passes; ++pass, shift += db) {
      for (uint32_t d = t; d < SB_WAVES * nbins; d += SB_THREADS) wcnt[d / nbins][d % nbins] = 0;
      __syncthreads();
      for (uint32_t i = c0 + lane; i < c1; i += 64) atomicAdd(&wcnt[wv][((((uint32_t)src[i].key & pmask) - kmin) >> shift) & dmask], 1u);
      __syncthreads();
      sb_bases<SB_THREADS>(wcnt, tot, nbins);
      for (uint32_t i0 = c0; i0 < c1; i0 += 64) {            // whole waves iterate together (c0, c1 are wave-uniform)
        const uint32_t i = i0 + lane;
        const bool ok = i < c1;
        Slim e = dead_slim();
        if (ok) e = ld_slim(src + i);
        const uint32_t d = ok ? ((((uint32_t)e.key & pmask) - kmin) >> shift) & dmask : 0u;
        uint64_t mm = __ballot(ok);
        for (int bit = 0; bit < db; ++bit) { const uint64_t bal = __ballot((d >> bit) & 1u); mm &= ((d >> bit) & 1u) ? bal : ~bal; }
        const uint32_t old = ok ? wcnt[wv][d] : 0u;
        __builtin_amdgcn_wave_barrier();
        if (ok && (mm & lt) == 0) wcnt[wv][d] = old + (uint32_t)__popcll(mm);
        __builtin_amdgcn_wave_barrier();
        if (ok) st_slim(dst + old + (uint32_t)__popcll(mm & lt), e);
      }
      __threadfence();                                       // the next pass reads what other waves of the block wrote
      __syncthreads();
      Slim* x = src; src = dst; dst = x;
    }
  }
  if (src != b) for (uint32_t i = t; i < m; i += SB_THREADS) st_slim(b + i, ld_slim(src + i));
}

template <int SB_THREADS>
__global__ __launch_bounds__(SB_THREADS) void sl_bucket_lds(Slim* __restrict__ in, const uint32_t* __restrict__ offs, int nbuckets, const uint32_t* __restrict__ d_live,
                                                            int pb, uint32_t cap, Slim* __restrict__ out, uint32_t* __restrict__ err) {
  constexpr int SB_WAVES = SB_THREADS / 64;
  __shared__ uint32_t sk[2][SB_CAP];
  __shared__ uint16_t sv[2][SB_CAP];
  __shared__ uint32_t wcnt[SB_WAVES][1 << SB_DBITS];
  __shared__ uint32_t tot[SB_WAVES];
  __shared__ uint32_t s_min, s_max;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const uint32_t n = *d_live;
  uint32_t lo = offs[b], hi = b + 1 < nbuckets ? offs[b + 1] : n;      // (tile 0's row of the scanned histogram: the buckets' first elements)
  if (lo > n) lo = n;
  if (hi > n) hi = n;
  if (hi <= lo) return;
  const uint32_t m = hi - lo;
  const bool slow = m > cap;                                           // (block-uniform)
  const uint32_t pmask = pb >= 32 ? 0xFFFFFFFFu : (1u << pb) - 1u;
  if (t == 0) { s_min = 0xFFFFFFFFu; s_max = 0u; }
  __syncthreads();
  uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
  for (uint32_t i = t; i < m; i += SB_THREADS) {
    const uint32_t k = (uint32_t)in[lo + i].key & pmask;               // a bucket lies in one list: the position orders it
    if (!slow) { sk[0][i] = k; sv[0][i] = (uint16_t)i; }
    kmin = min(kmin, k); kmax = max(kmax, k);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, d, 64)); kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, d, 64)); }
  if (lane == 0) { atomicMin(&s_min, kmin); atomicMax(&s_max, kmax); }
  __syncthreads();
  kmin = s_min; kmax = s_max;
  if (slow) {
    if (t == 0) atomicOr(err, ERRB_BUCKET1_SLOW);
    sl_bucket_global<SB_THREADS>(in + lo, out + lo, m, pmask, kmin, kmax, wcnt, tot);
    return;
  }
  const uint32_t width = kmax - kmin;
  int src = 0;
  if (width != 0) {
    const int wbits = 32 - __builtin_clz(width);
    const int passes = (wbits + SB_DBITS - 1) / SB_DBITS;
    const int db = (wbits + passes - 1) / passes;
    const uint32_t dmask = (1u << db) - 1u, nbins = 1u << db;
    // elements of wave w: [w * per, (w + 1) * per) — wave-major chunks keep the input order inside equal digits
    const uint32_t per = ((m + SB_WAVES - 1) / SB_WAVES + 63u) & ~63u;
    const uint32_t c0 = min(m, (uint32_t)wv * per), c1 = min(m, c0 + per);
    const uint64_t lt = (1ull << lane) - 1ull;
    for (int pass = 0, shift = 0; pass < passes; ++pass, shift += db) {
      for (uint32_t d = t; d < SB_WAVES * nbins; d += SB_THREADS) wcnt[d / nbins][d % nbins] = 0;
      __syncthreads();
      for (uint32_t i = c0 + lane; i < c1; i += 64) atomicAdd(&wcnt[wv][((sk[src][i] - kmin) >> shift) & dmask], 1u);
      __syncthreads();
      sb_bases<SB_THREADS>(wcnt, tot, nbins);
      for (uint32_t i0 = c0; i0 < c1; i0 += 64) {            // whole waves iterate together (c0, c1 are wave-uniform)
        const uint32_t i = i0 + lane;
        const bool ok = i < c1;
        const uint32_t k = ok ? sk[src][i] : 0u;
        const uint16_t v = ok ? sv[src][i] : (uint16_t)0;
        const uint32_t d = ((k - kmin) >> shift) & dmask;
        uint64_t mm = __ballot(ok);
        for (int bit = 0; bit < db; ++bit) { const uint64_t bal = __ballot((d >> bit) & 1u); mm &= ((d >> bit) & 1u) ? bal : ~bal; }
        const uint32_t old = ok ? wcnt[wv][d] : 0u;
        __builtin_amdgcn_wave_barrier();
        if (ok && (mm & lt) == 0) wcnt[wv][d] = old + (uint32_t)__popcll(mm);
        __builtin_amdgcn_wave_barrier();
        if (ok) { const uint32_t dst = old + (uint32_t)__popcll(mm & lt); sk[src ^ 1][dst] = k; sv[src ^ 1][dst] = v; }
      }
      __syncthreads();
      src ^= 1;
    }
  }
  for (uint32_t i = t; i < m; i += SB_THREADS) st_slim(out + lo + i, ld_slim(in + lo + sv[src][i]));       // (the bucket's 64 KB were just read: L2)
}

// ---- the sorts behind a stage that drops a list bit: rank inside the class + ONE merge -------------------------------------------
// Sorts 2 and 3 and the call sort do not meet unordered input. Their slots come in two CLASSES — the value of the key bit the stage
// in front dropped: source for the merge of the cigar and split lists (H:478-490), type for the final list (H:492-499), haplotype for
// the calls (H:571-594) — and inside a class the slots are in the order of the sorted table the stage worked on: slot j has an
// ANCHOR a_j (the key of the stage's input element j, dropped bit cleared) that ascends with j, and the element a slot holds lies
// within [a_j - dlo, a_j + dhi] (a cluster representative is a later element within cluster_shift of its seed, H:236-247; a call's kept
// signature is the hp1 row's own or its mate's within pair_shift, H:583-586). So
//   rank of element i inside its class = live class elements in front of slot i
//                                        - those of them with a larger key      (only slots j < i with a_j + dhi > K_i can hold one)
//                                        + those behind it with a smaller key   (only slots j > i with a_j - dlo < K_i),
// both counted from an LDS window around the tile (anchors and keys of MS_T slots + MS_H on either side), and the two ranked
// classes — each exactly sorted now, ties in slot order — are merged once (class A first on ties: its slots lie in front of class
// B's inside every list). Three launches that move the table twice, where the LSD passes take nine and move it six times; the result
// is the stable sort of the slots by key, bit for bit. Whatever the window cannot decide (a scan that reaches its edge: thousands of
// slots within one shift) or an element outside its anchor's range raises ERRB_MERGE_FALLBACK: the run repeats on the LSD passes.
// Keys of at most 31 bits after squeezing (KeyCmp): a single-chromosome shard has 30.
// a stage-2 / stage-3 sort: the cluster kernel's output slots with the sorted table it worked on as anchors
struct SrcAnch {
  const Slim* e; const Slim* anc; uint64_t drop;
  __device__ __forceinline__ uint32_t lc(uint32_t i) const { const uint64_t k = e[i].key; return k == VSV_KEY_DEAD ? 0u : ((k & SL_CLASS) ? 2u : 1u); }
  __device__ __forceinline__ Slim at2(uint32_t i, uint64_t& anchor) const { anchor = sl_stash(anc[i].key, drop, SL_CLASS); return ld_slim(e + i); }
  __device__ __forceinline__ uint64_t key2(uint32_t i, uint64_t& anchor) const { anchor = sl_stash(anc[i].key, drop, SL_CLASS); return e[i].key; }
  __device__ __forceinline__ Slim at(uint32_t i) const { return ld_slim(e + i); }
};

// tcnt[2 t + c] = live elements of class c in tile t; ctl->nA / nB += the same (a zeroed slot)
template <typename SRC>
__global__ __launch_bounds__(256) void sl_ms_count(SRC src, const uint32_t* __restrict__ d_n, uint32_t* __restrict__ tcnt, MsCtl* __restrict__ ctl) {
  __shared__ uint32_t cA, cB;
  const uint32_t n = *d_n, ntiles = (n + MS_T - 1) / MS_T;
  const int lane = threadIdx.x & 63;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    if (threadIdx.x == 0) { cA = 0; cB = 0; }
    __syncthreads();
    uint32_t a = 0, b = 0;
#pragma unroll
    for (int k = 0; k < MS_T / 256; ++k) {
      const uint32_t i = tile * MS_T + k * 256 + threadIdx.x;
      const uint32_t c = i < n ? src.lc(i) : 0u;
      a += c == 1u ? 1u : 0u; b += c == 2u ? 1u : 0u;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    if (lane == 0) { atomicAdd(&cA, a); atomicAdd(&cB, b); }
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t ta = cA, tb = cB;
      tcnt[2 * tile] = ta; tcnt[2 * tile + 1] = tb;
      if (ta) { atomicAdd(&ctl->nA, ta); atomicAdd(&ctl->grp[tile / MS_GROUP][0], ta); }
      if (tb) { atomicAdd(&ctl->nB, tb); atomicAdd(&ctl->grp[tile / MS_GROUP][1], tb); }
    }
    __syncthreads();
  }
}

__device__ __forceinline__ uint32_t sl_wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(v, d, 64); if (lane >= d) v += o; }
  return v;
}

// every live element to its rank inside its class: class A at out[0, nA), class B at out[nA, nA + nB)
template <typename SRC>
__global__ __launch_bounds__(256) void sl_ms_local(SRC src, const uint32_t* __restrict__ d_n, KeyCmp kc, uint32_t dlo, uint32_t dhi, const uint32_t* __restrict__ tcnt,
                                                   const MsCtl* __restrict__ ctl, Slim* __restrict__ out, uint32_t* __restrict__ d_live, uint32_t* __restrict__ err) {
  __shared__ uint32_t la[MS_W], lk[MS_W];          // per window slot: anchor | class << 31 (0xFFFFFFFF outside the table), key (0xFFFFFFFF: no element)
  __shared__ uint32_t wc[2][64];                   // live elements per (round, wave) and class -> exclusive prefixes
  __shared__ uint32_t below[2];                    // live elements of the tiles in front
  __shared__ uint8_t lr[MS_W];                     // live elements of the slot's class in front of it in its wave
  const uint32_t n = *d_n, ntiles = (n + MS_T - 1) / MS_T;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t nA = ctl->nA;
  if (blockIdx.x == 0 && threadIdx.x == 0) *d_live = nA + ctl->nB;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t w0 = (int64_t)tile * MS_T - MS_H;            // slot of window index 0
    __syncthreads();
#pragma unroll
    for (int r = 0; r < MS_R; ++r) {               // anchors and keys of the window (the elements themselves are fetched when their rank is known)
      const uint32_t w = (uint32_t)r * 256u + threadIdx.x;
      const int64_t slot = w0 + w;
      uint32_t a = 0xFFFFFFFFu, k = 0xFFFFFFFFu, cls = 0;
      if (slot >= 0 && slot < (int64_t)n) {
        uint64_t anc;
        const uint64_t key = src.key2((uint32_t)slot, anc);
        cls = (uint32_t)(anc >> 62) & 1u;
        a = ((uint32_t)kc(anc & ~SL_CLASS) & 0x7FFFFFFFu) | (cls << 31);
        if (key != VSV_KEY_DEAD) k = (uint32_t)kc(key & ~SL_CLASS) & 0x7FFFFFFFu;
      }
      la[w] = a; lk[w] = k;
      const bool live = k != 0xFFFFFFFFu;
      const uint64_t mA = __ballot(live && cls == 0u), mB = __ballot(live && cls != 0u);
      const uint64_t mine = cls ? mB : mA;
      lr[w] = (uint8_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
      if (lane == 0) { wc[0][r * 4 + wv] = (uint32_t)__popcll(mA); wc[1][r * 4 + wv] = (uint32_t)__popcll(mB); }
    }
    if (wv == 3) {  // live elements of either class in the tiles in front of this one: whole groups + the tiles of its own group
      const uint32_t g = tile / MS_GROUP;
      uint32_t ba = 0, bb = 0;
      for (uint32_t q = lane; q < g; q += 64) { ba += ctl->grp[q][0]; bb += ctl->grp[q][1]; }
      const uint32_t t = g * MS_GROUP + (uint32_t)lane;
      if (t < tile) { const uint2 c = *reinterpret_cast<const uint2*>(tcnt + 2 * (size_t)t); ba += c.x; bb += c.y; }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) { ba += __shfl_xor(ba, d, 64); bb += __shfl_xor(bb, d, 64); }
      if (lane == 0) { below[0] = ba; below[1] = bb; }
    }
    __syncthreads();
    if (wv < 2) {                                              // exclusive prefixes over the (round, wave) counts, one wave per class
      const uint32_t v = lane < MS_R * 4 ? wc[wv][lane] : 0u;
      const uint32_t incl = sl_wave_incl_scan(v, lane);
      wc[wv][lane] = incl - v;
    }
    __syncthreads();
    const bool open_left = w0 > 0, open_right = w0 + MS_W < (int64_t)n;      // the table goes on beyond the window
    const uint32_t base[2] = {below[0] - wc[0][MS_H / 64], below[1] - wc[1][MS_H / 64] + nA};
    bool trouble = false;
#pragma unroll 1
    for (int r = 0; r < MS_R; ++r) {               // (a rolled loop of per-lane work: nine copies of the two scans cost the kernel half of its wave slots)
      const uint32_t w = (uint32_t)r * 256u + threadIdx.x;
      if (w < (uint32_t)MS_H || w >= (uint32_t)(MS_H + MS_T)) continue;                        // halo
      const uint32_t K = lk[w];
      if (K == 0xFFFFFFFFu) continue;                                                          // no element
      const uint32_t aw = la[w], cls = aw >> 31, A = aw & 0x7FFFFFFFu, myrank = lr[w];
      if (K + dlo < A || K > A + dhi) trouble = true;                        // not what the caller promised
      { const uint32_t ap = la[w - 1u]; if (ap != 0xFFFFFFFFu && (ap >> 31) == cls && (ap & 0x7FFFFFFFu) > A) trouble = true; }   // ... nor this: anchors ascend inside a class
      // four slots either way in straight-line code (a pile with a signature per 45 bp looks two or three slots far; the loops behind
      // take over where that does not reach a slot that ends the scan: every trip of a divergent loop costs scalar mask bookkeeping)
      uint32_t back = 0, fwd = 0;
      int jb, jf;
      {
        uint32_t stop = 0, inv = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {                              // (w >= MS_H: the slots exist)
          const uint32_t aj = la[w - 1u - u], kj = lk[w - 1u - u];
          stop |= ((aj == 0xFFFFFFFFu || (aj >> 31) != cls || (aj & 0x7FFFFFFFu) + dhi <= K) ? 1u : 0u) << u;
          inv |= ((kj != 0xFFFFFFFFu && kj > K) ? 1u : 0u) << u;
        }
        const uint32_t f = (uint32_t)__builtin_ctz(stop | 16u);
        back = (uint32_t)__popc(inv & ((1u << f) - 1u));
        jb = f < 4u ? -2 : (int)w - 5;                             // -2: decided
      }
      {
        uint32_t stop = 0, inv = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {                              // (w < MS_H + MS_T: the slots exist)
          const uint32_t aj = la[w + 1u + u], kj = lk[w + 1u + u];
          stop |= ((aj == 0xFFFFFFFFu || (aj >> 31) != cls || (aj & 0x7FFFFFFFu) >= K + dlo) ? 1u : 0u) << u;
          inv |= ((kj != 0xFFFFFFFFu && kj < K) ? 1u : 0u) << u;
        }
        const uint32_t f = (uint32_t)__builtin_ctz(stop | 16u);
        fwd = (uint32_t)__popc(inv & ((1u << f) - 1u));
        jf = f < 4u ? -2 : (int)w + 5;
      }
      if (jb != -2) for (int j = jb;; --j) {
        if (j < 0) { trouble = trouble || open_left; break; }
        const uint32_t aj = la[j];
        if (aj == 0xFFFFFFFFu || (aj >> 31) != cls) break;                   // the table's start / the class's previous list lies in front
        if ((aj & 0x7FFFFFFFu) + dhi <= K) break;                            // nothing at or in front of j exceeds K
        const uint32_t kj = lk[j];
        back += (kj != 0xFFFFFFFFu && kj > K) ? 1u : 0u;
      }
      if (jf != -2) for (int j = jf;; ++j) {
        if (j >= MS_W) { trouble = trouble || open_right; break; }
        const uint32_t aj = la[j];
        if (aj == 0xFFFFFFFFu || (aj >> 31) != cls) break;
        if ((aj & 0x7FFFFFFFu) >= K + dlo) break;                            // nothing at or behind j lies below K
        const uint32_t kj = lk[j];
        fwd += (kj != 0xFFFFFFFFu && kj < K) ? 1u : 0u;
      }
      const uint32_t rank = (cls ? base[1] : base[0]) + wc[cls][r * 4 + wv] + myrank - back + fwd;
      st_slim(out + rank, src.at((uint32_t)(w0 + w)));
    }
    if (trouble) atomicOr(err, ERRB_MERGE_FALLBACK);
  }
}

// the two ranked classes merged: a tile of MS_MT outputs finds its two diagonals by wave-wide 64-ary searches (four dependent rounds
// over millions of elements), ranks its elements against the other class's keys in LDS and writes them with the class bit cleared
__device__ __forceinline__ uint32_t sl_ms_key(const Slim* __restrict__ p, KeyCmp kc) { return (uint32_t)kc(p->key & ~SL_CLASS) & 0x7FFFFFFFu; }
__device__ __forceinline__ uint32_t sl_ms_path(const Slim* __restrict__ A, uint32_t nA, const Slim* __restrict__ B, uint32_t nB, uint32_t d, KeyCmp kc, int lane) {
  // (lo, hi: wave-uniform by construction, and told so — the ballot sits in a loop the compiler can then branch on the scalar unit)
  uint32_t lo = __builtin_amdgcn_readfirstlane(d > nB ? d - nB : 0u), hi = __builtin_amdgcn_readfirstlane(d < nA ? d : nA);
  while (lo < hi) {                                  // a* = elements of A among the first d outputs: A[a] <= B[d - 1 - a] holds for a < a*
    const uint32_t s = hi - lo, step = (s + 63u) / 64u;
    const uint32_t p = lo + (uint32_t)lane * step;
    bool q = false;
    if (p < hi) q = sl_ms_key(A + p, kc) <= sl_ms_key(B + (d - 1u - p), kc);
    const uint32_t t = (uint32_t)__popcll(__ballot(q));
    if (t == 0u) { hi = lo; }
    else {
      const uint32_t nlo = lo + (t - 1u) * step + 1u;
      const uint32_t cand = lo + t * step;
      hi = __builtin_amdgcn_readfirstlane((t < 64u && cand < hi) ? cand : hi);
      lo = __builtin_amdgcn_readfirstlane(nlo);
    }
  }
  return lo;
}
__global__ __launch_bounds__(512) void sl_ms_merge(const Slim* __restrict__ tmp, const MsCtl* __restrict__ ctl, KeyCmp kc, Slim* __restrict__ out) {
  __shared__ uint32_t lkeys[MS_MT];
  __shared__ uint32_t sh_a[2];
  const uint32_t nA = ctl->nA, nB = ctl->nB, n = nA + nB;
  const Slim* A = tmp; const Slim* B = tmp + nA;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t ntiles = (n + MS_MT - 1) / MS_MT;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint32_t d0 = tile * MS_MT, d1 = min(n, d0 + MS_MT);
    __syncthreads();
    if (wv < 2) {                                                   // (wave-uniform: whole waves enter the search)
      const uint32_t a = sl_ms_path(A, nA, B, nB, __builtin_amdgcn_readfirstlane(wv ? d1 : d0), kc, lane);
      if (lane == 0) sh_a[wv] = a;
    }
    __syncthreads();
    const uint32_t a0 = sh_a[0], a1 = sh_a[1], b0 = d0 - a0, cA = a1 - a0, cT = d1 - d0;
    Slim e[MS_MT / 512];
#pragma unroll
    for (int k = 0; k < MS_MT / 512; ++k) {
      const uint32_t c = (uint32_t)k * 512u + threadIdx.x;
      if (c < cT) {
        e[k] = ld_slim(c < cA ? A + a0 + c : B + b0 + (c - cA));
        lkeys[c] = (uint32_t)kc(e[k].key & ~SL_CLASS) & 0x7FFFFFFFu;
      }
    }
    __syncthreads();
    const uint32_t cB = cT - cA;
#pragma unroll
    for (int k = 0; k < MS_MT / 512; ++k) {
      const uint32_t c = (uint32_t)k * 512u + threadIdx.x;
      if (c >= cT) continue;
      const uint32_t K = lkeys[c];
      uint32_t r;
      if (c < cA) {                                   // + elements of B below K
        uint32_t lo = 0, hi = cB;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (lkeys[cA + mid] < K) lo = mid + 1; else hi = mid; }
        r = c + lo;
      } else {                                        // + elements of A at or below K
        uint32_t lo = 0, hi = cA;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (lkeys[mid] <= K) lo = mid + 1; else hi = mid; }
        r = (c - cA) + lo;
      }
      e[k].key &= ~SL_CLASS;
      st_slim(out + d0 + r, e[k]);
    }
  }
}

bool sl_merge_ok(const SlimWork& w, int nbits, KeyCmp kc) {
  return w.merge_sorts && nbits - kc.zbits <= 31 && (w.cap + MS_T - 1) / MS_T <= (int64_t)MS_GROUP * MS_MAX_GROUPS;
}

// a zeroed control slot for one such sort; `counted`: the tile counts are zeroed too, for a producer that counts its own outputs (sl_cluster)
MsCtl* sl_ms_begin(hipStream_t st, const SlimWork& w, bool counted) {
  MsCtl* ctl = (MsCtl*)(w.totals + (size_t)(*w.pass_cursor) * 2048);
  ++*w.pass_cursor;
  if (counted) (void)hipMemsetAsync(w.hist, 0, (size_t)((w.cap + MS_T - 1) / MS_T) * 2 * sizeof(uint32_t), st);
  return ctl;
}
template <typename SRC>
Slim* sl_merge_sort(hipStream_t st, SRC src, const uint32_t* d_slots, uint32_t* d_live, KeyCmp kc, uint32_t dlo, uint32_t dhi, Slim* tmp, Slim* out, const SlimWork& w,
                    MsCtl* counted) {
  MsCtl* ctl = counted ? counted : sl_ms_begin(st, w, false);
  const int64_t tiles = (w.cap + MS_T - 1) / MS_T, mtiles = (w.cap + MS_MT - 1) / MS_MT;
  const int64_t hint_tiles = (w.rows_hint + MS_T - 1) / MS_T;
  const int64_t want = hint_tiles < tiles ? hint_tiles : tiles;
  const int gc = (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
  const int gl = (int)(tiles < 1 ? 1 : (tiles > 8192 ? 8192 : tiles));
  const int gm = (int)(mtiles < 1 ? 1 : (mtiles > 4096 ? 4096 : mtiles));
  if (!counted) sl_ms_count<<<gc, 256, 0, st>>>(src, d_slots, w.hist, ctl);
  sl_ms_local<<<gl, 256, 0, st>>>(src, d_slots, kc, dlo, dhi, w.hist, ctl, tmp, d_live, w.err);
  sl_ms_merge<<<gm, 512, 0, st>>>(tmp, ctl, kc, out);
  return out;
}

}  // namespace

// sort 1 as counting pass + LDS sort per bucket; nullptr when the shape does not take it (the caller runs the LSD passes)
template <int BITS>
Slim* sl_bucket_sort1_bits(hipStream_t st, const Slim* in, const uint32_t* d_slots, uint32_t* d_live, int pb, int tid_bits, Slim* tmp, Slim* out, const SlimWork& w) {
  const int pairs_log = 2 + tid_bits, bp_log = BITS - pairs_log;                 // list pairs (tid, hap, type); buckets per pair
  if (bp_log < 3 || pb > 31) return nullptr;
  // a pair's buckets go to its cigar and its split list by the share of split elements the handle expects (+25 %; at least one each)
  const uint32_t bp = 1u << bp_log;
  double want1 = w.split_share * 1.25 * (double)bp;
  uint32_t nb1 = want1 < 1.0 ? 1u : (uint32_t)(want1 + 0.5);
  if (nb1 > bp / 2) nb1 = bp / 2;
  const uint32_t b0 = bp - nb1;
  constexpr int SLOTS = (1 << BITS) > 2048 ? (1 << BITS) / 2048 : 1;              // totals slots of 2048 words (+ one for the position range)
  uint32_t* totals = w.totals + (size_t)(*w.pass_cursor) * 2048;
  uint32_t* mm_own = totals + (size_t)SLOTS * 2048;
  *w.pass_cursor += SLOTS + 1;
  const MsdDigit dig{pb, bp_log, b0, nb1, w.mm ? w.mm : mm_own};     // (w.mm: the fold measured the cigar elements; a split element outside lands in an end bucket)
  const int64_t max_tiles = (w.cap + SL_TILE - 1) / SL_TILE;
  const int grid = (int)(max_tiles < 2048 ? (max_tiles < 1 ? 1 : max_tiles) : 2048);
  static const int cap_env = vsv_dbg_env("VSV_SORT1_CAP") ? atoi(vsv_dbg_env("VSV_SORT1_CAP")) : 0;      // tests: buckets above this many elements take the global-memory form
  const uint32_t cap = cap_env > 0 && cap_env < SB_CAP ? (uint32_t)cap_env : (uint32_t)SB_CAP;
  if (!w.mm) sl_minmax<<<grid < 1024 ? grid : 1024, 256, 0, st>>>(in, d_slots, pb, mm_own);
  sl_hist<BITS, SrcSlim, true, 4><<<grid, 256, 0, st>>>(SrcSlim{in}, d_slots, dig, w.hist, totals);
  sl_scan<BITS><<<(1 << BITS) / 16, 1024, 0, st>>>(w.hist, totals, d_slots, d_live);
  static const char* sc_form = vsv_dbg_env("VSV_SORT1_SCATTER");      // timing experiments: "4" = the 4-wave scatter of the LSD passes
  if (sc_form && sc_form[0] == '4') sl_scatter<BITS, SrcSlim, true, 4><<<grid, 256, 0, st>>>(SrcSlim{in}, d_slots, dig, w.hist, tmp);
  else sl_scatter_msd<BITS, SrcSlim><<<grid, 512, 0, st>>>(SrcSlim{in}, d_slots, dig, w.hist, tmp);
  sl_bucket_lds<512><<<1 << BITS, 512, 0, st>>>(tmp, w.hist, 1 << BITS, d_live, pb, cap, out, w.err);
  return out;
}
Slim* sl_bucket_sort1(hipStream_t st, const Slim* in, const uint32_t* d_slots, uint32_t* d_live, int pb, int tid_bits, Slim* tmp, Slim* out, const SlimWork& w) {
  static const int forced = vsv_dbg_env("VSV_SORT1_BITS") ? atoi(vsv_dbg_env("VSV_SORT1_BITS")) : 0;       // timing experiments / tests: 11 | 12 (0: by the size hint), -1: the passes
  if (!w.bucket_sort1 || forced < 0) return nullptr;
  // ~2000 or ~4000 buckets: the fewest that keep the previous run's table (+25 %) at <= ~1600 elements per bucket on average; beyond
  // ~2000 per bucket (LDS holds 4096: a list pair or a region twice as dense as the average still fits) the passes
  const bool big = forced ? forced == 12 : w.rows_hint > (int64_t)1600 * 2048;
  if (!forced && w.rows_hint > (int64_t)2048 * 4096) return nullptr;
  return big ? sl_bucket_sort1_bits<12>(st, in, d_slots, d_live, pb, tid_bits, tmp, out, w) : sl_bucket_sort1_bits<11>(st, in, d_slots, d_live, pb, tid_bits, tmp, out, w);
}

// ================================================== entry points (capi.hip) ==================================================
int vsv_slim_sort_passes(int nbits) { return (nbits + 7) / 8; }     // (upper bound: a 10-bit plan never takes more passes)

// stage 1: rows -> elements, sort by (tid, hap, type, source, pos), cluster per list. Returns the cluster output (slots = *d_alive1).
void* vsv_slim_stage1(hipStream_t st, const vsv_sig* s1in, const uint32_t* d_n_s1, uint32_t* d_alive1, int pb, int tid_lo, int tid_bits, int cluster_shift,
                      const SlimWork& w, Counters* ctr, bool prebuilt, const void** sorted1, void** ctl2) {
  Slim* b0 = (Slim*)w.buf[0]; Slim* b1 = (Slim*)w.buf[1]; Slim* e2 = (Slim*)w.buf[2];
  // (a fused run has the elements already: fold_kernel / split_eval wrote them next to the rows, vsv_slim_emit)
  if (!prebuilt) sl_from_rows<<<w.grid, 256, 0, st>>>(s1in, d_n_s1, pb, tid_lo, tid_bits, b0, &ctr->err);
  Slim* sorted = sl_bucket_sort1(st, b0, d_n_s1, d_alive1, pb, tid_bits, b1, b0, w);       // (in: b0, bucket order: b1, sorted: b0 again)
  if (!sorted) sorted = sl_sort(st, SrcSlim{b0}, d_n_s1, d_alive1, pb + 3 + tid_bits, KeyCmp{0, 0}, b1, b0, w, w.rows_hint);
  const int64_t tiles = (w.cap + CL_TILE - 1) / CL_TILE;
  const bool ms = sl_merge_ok(w, pb + 3 + tid_bits, KeyCmp{pb, 1});       // the sort behind these clusters: rank + merge (sl_merge_sort)
  MsCtl* ctl = ms ? sl_ms_begin(st, w, true) : nullptr;                   // ... whose tile counts the cluster kernel leaves in w.hist
  sl_cluster<<<(int)(tiles < 4096 ? (tiles < 1 ? 1 : tiles) : 4096), 256, 0, st>>>(sorted, d_alive1, cluster_shift, KeyFmt{pb}, pb, ms ? SL_CLASS : 0ull, e2, w.cl,
                                                                                    MsCount{ms ? w.hist : nullptr, ctl});
  *sorted1 = sorted;         // (the anchors of that sort: stays in scratch buffer 0 / 1 until vsv_slim_merge has run)
  *ctl2 = ctl;
  return e2;
}

// merge_all: sort the stage-1 representatives by (tid, hap, type, pos), cluster, sort by (tid, hap, pos). Returns the merged elements.
void* vsv_slim_merge(hipStream_t st, const void* e2, const void* sorted1, void* ctl2, const uint32_t* d_alive1, uint32_t* d_alive2, uint32_t* d_alive3, int pb, int tid_bits,
                     int shift1, int cluster_shift, const SlimWork& w) {
  Slim* b0 = (Slim*)w.buf[0]; Slim* b1 = (Slim*)w.buf[1]; Slim* e3 = (Slim*)w.buf[3]; Slim* m0 = (Slim*)w.buf[4]; Slim* m1 = (Slim*)w.buf[5];
  const int nbits = pb + 3 + tid_bits;
  const int64_t tiles = (w.cap + CL_TILE - 1) / CL_TILE;
  const int cgrid = (int)(tiles < 4096 ? (tiles < 1 ? 1 : tiles) : 4096);
  if (ctl2 && sl_merge_ok(w, nbits, KeyCmp{pb, 1})) {       // (ctl2: the stage-1 clusters carry their class bits and left their tile counts)
    // classes by source / by type; anchors = the table the clusters were cut from; a representative lies at or behind its seed
    Slim* free01 = (const Slim*)sorted1 == b0 ? b1 : b0;
    Slim* s2 = sl_merge_sort(st, SrcAnch{(const Slim*)e2, (const Slim*)sorted1, 1ull << pb}, d_alive1, d_alive2, KeyCmp{pb, 1}, 0u, (uint32_t)shift1, free01, m0, w,
                             (MsCtl*)ctl2);      // (shift1: what the stage-1 clusters were cut with)
    MsCtl* ctl3 = sl_ms_begin(st, w, true);
    sl_cluster<<<cgrid, 256, 0, st>>>(s2, d_alive2, cluster_shift, KeyFmt{pb}, pb + 1, SL_CLASS, e3, w.cl, MsCount{w.hist, ctl3});
    return sl_merge_sort(st, SrcAnch{e3, s2, 1ull << (pb + 1)}, d_alive2, d_alive3, KeyCmp{pb, 2}, 0u, (uint32_t)cluster_shift, b0, m1, w, ctl3);
  }
  Slim* s2 = sl_sort(st, SrcSlim{(const Slim*)e2}, d_alive1, d_alive2, nbits, KeyCmp{pb, 1}, b0, b1, w, w.rows_hint);
  sl_cluster<<<cgrid, 256, 0, st>>>(s2, d_alive2, cluster_shift, KeyFmt{pb}, pb + 1, 0ull, e3, w.cl, MsCount{nullptr, nullptr});
  // the result must outlive the pairing stage and the readback: it lands in one of the two buffers reserved for it
  return sl_sort(st, SrcSlim{e3}, d_alive2, d_alive3, nbits, KeyCmp{pb, 2}, m0, m1, w, w.rows_hint);
}

// pair_sig + the final order: pairing state in w.cl, call elements sorted by (tid, pos), call rows gathered from s1in
void vsv_slim_pair(hipStream_t st, const void* merged, const uint32_t* d_alive3, uint32_t* d_ncalls, int pb, int tid_bits, int pair_shift, int pair_window,
                   const vsv_sig* s1in, vsv_call* calls, bool dense, const SlimWork& w, Counters* ctr) {
  const Slim* m = (const Slim*)merged;
  const KeyFmt kf{pb};
  const int right = pair_shift < pair_window ? pair_shift : pair_window;
  // scratch: buffers 0 / 1 hold nothing that is still needed (buffer 2 = the stage-1 clusters, 4 / 5 = the merged elements)
  Slim* b0 = (Slim*)w.buf[0]; Slim* b1 = (Slim*)w.buf[1];
  if (dense) {
    sl_pair_prep<<<w.grid, 256, 0, st>>>(m, d_alive3, kf, pair_shift, right, w.hj, w.cl, &ctr->max_stretch);
    uint32_t* done1 = w.done1; uint64_t* res = (uint64_t*)w.buf[3];         // (the stage-2 clusters are consumed)
    sl_pair_rounds_init<<<w.grid, 256, 0, st>>>(d_alive3, done1, res);
    for (uint32_t r = 0; r < 10; ++r) {
      sl_pair_round<false><<<w.grid, 256, 0, st>>>(m, d_alive3, kf, pair_shift, right, w.cl, w.hj, done1, res, r + 1);
      sl_pair_round<true><<<w.grid, 256, 0, st>>>(m, d_alive3, kf, pair_shift, right, w.cl, w.hj, done1, res, r + 1);
    }
    sl_pair_leftover<<<w.grid, 256, 0, st>>>(m, d_alive3, kf, pair_shift, right, w.cl, w.hj, done1);
  } else {
    sl_pair_index<<<w.grid, 256, 0, st>>>(m, d_alive3, kf, pair_shift, w.cl, w.hj);       // (w.hj: one word per window)
    const int64_t wins = (w.cap + QW - 1) / QW;
    sl_pair_lds<<<(int)(wins < 32768 ? (wins < 1 ? 1 : wins) : 32768), 64, 0, st>>>(m, d_alive3, kf, pair_shift, right, w.cl, w.hj, &ctr->max_stretch);
  }
  // the call sort: classes by haplotype, anchors = the merged slots' own keys, a call's key within pair_shift of its slot's either way
  Slim* cs = sl_merge_ok(w, pb + 3 + tid_bits, KeyCmp{pb, 3})
      ? sl_merge_sort(st, SrcCalls{m, w.cl, pb + 2, SL_CLASS}, d_alive3, d_ncalls, KeyCmp{pb, 3}, (uint32_t)pair_shift, (uint32_t)pair_shift, b0, b1, w, nullptr)
      : sl_sort(st, SrcCalls{m, w.cl, pb + 2, 0ull}, d_alive3, d_ncalls, pb + 3 + tid_bits, KeyCmp{pb, 3}, b0, b1, w, w.rows_hint);
  sl_calls_out<<<w.grid, 256, 0, st>>>(cs, d_ncalls, m, pb + 2, s1in, calls);
}

// ---- (key, value) pair arrays through the same passes: the split stage's candidate sorts on large inputs ---------------------------
namespace {
struct SrcPairs {
  using elem = Slim;
  const uint64_t* key; const uint32_t* val;
  __device__ __forceinline__ Slim at(uint32_t i) const { Slim s; s.key = key[i]; s.svlen = (int32_t)i; s.idx = val[i]; return s; }     // (svlen: the input ordinal)
};
// key of a split-pair slot, straight from the candidates sorted by (tid, hap, name) (as PairKey of radix_sort.hip): candidate j followed
// by another one of the same name heads a pair slot keyed (tid, hap, record of the name's first candidate); every other slot is dead
// With `cord` (the candidates' ordinals in record order) a slot carries its two candidates instead of its index: svlen = ordinal of
// candidate j, idx = ordinal of candidate j + 1 (split_eval_info, sig_stages.hip).
struct SrcPairSlots {
  using elem = Slim;
  const uint64_t* ckey; const uint32_t* crec; int qid_bits, rec_bits; const uint32_t* d_n; const uint32_t* cord;
  __device__ __forceinline__ Slim at(uint32_t j) const {
    const uint32_t n = *d_n;
    const uint64_t k = ckey[j];
    Slim s; s.svlen = 0; s.idx = j; s.key = VSV_KEY_DEAD;
    if (j + 1 < n && ckey[j + 1] == k) {
      uint32_t g = j;
      while (g > 0 && ckey[g - 1] == k) --g;
      s.key = ((k >> qid_bits) << rec_bits) | crec[g];
      if (cord) { s.svlen = (int32_t)cord[j]; s.idx = cord[j + 1]; }
    }
    return s;
  }
};
// sorted elements -> (key, value) arrays of n slots: the live ones in order, dead keys behind them
__global__ __launch_bounds__(256) void sl_unpack_pairs(const Slim* __restrict__ e, const uint32_t* __restrict__ d_live, const uint32_t* __restrict__ d_n,
                                                       uint64_t* __restrict__ key, uint32_t* __restrict__ val, uint32_t* __restrict__ aux) {
  const uint32_t n = *d_n, live = *d_live;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (i < live) { const Slim x = ld_slim(e + i); key[i] = x.key; val[i] = x.idx; if (aux) aux[i] = (uint32_t)x.svlen; }
    else { key[i] = VSV_KEY_DEAD; val[i] = 0; if (aux) aux[i] = 0; }
  }
}
// the same two sorts on 8-byte elements (keys of at most 31 bits): (key, ordinal) travels through the passes — half the bytes — and the
// unpack kernels gather what the 16-byte elements carried (the record; the two candidates' ordinals) by ordinal
struct SrcPairs8 {
  using elem = Slim8;
  const uint64_t* key;
  __device__ __forceinline__ Slim8 at(uint32_t i) const { return Slim8{(uint32_t)key[i], i}; }
};
struct SrcPairSlots8 {
  using elem = Slim8;
  const uint64_t* ckey; const uint32_t* crec; int qid_bits, rec_bits; const uint32_t* d_n;
  __device__ __forceinline__ Slim8 at(uint32_t j) const {
    const uint32_t n = *d_n;
    const uint64_t k = ckey[j];
    Slim8 s{SL8_DEAD, j};
    if (j + 1 < n && ckey[j + 1] == k) {
      uint32_t g = j;
      while (g > 0 && ckey[g - 1] == k) --g;
      s.key = (uint32_t)(((k >> qid_bits) << rec_bits) | crec[g]);
    }
    return s;
  }
};
__global__ __launch_bounds__(256) void sl_unpack_pairs8(const Slim8* __restrict__ e, const uint32_t* __restrict__ d_live, const uint32_t* __restrict__ d_n,
                                                        const uint32_t* __restrict__ val_in, uint64_t* __restrict__ key, uint32_t* __restrict__ val, uint32_t* __restrict__ ord) {
  const uint32_t n = *d_n, live = *d_live;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (i < live) { const Slim8 x = ld_slim(e + i); key[i] = x.key; val[i] = val_in[x.val]; if (ord) ord[i] = x.val; }
    else { key[i] = VSV_KEY_DEAD; val[i] = 0; if (ord) ord[i] = 0; }
  }
}
__global__ __launch_bounds__(256) void sl_unpack_slots8(const Slim8* __restrict__ e, const uint32_t* __restrict__ d_live, const uint32_t* __restrict__ d_n,
                                                        const uint32_t* __restrict__ cord, uint64_t* __restrict__ key, uint32_t* __restrict__ val, uint32_t* __restrict__ c1) {
  const uint32_t n = *d_n, live = *d_live;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (i < live) {
      const Slim8 x = ld_slim(e + i);
      key[i] = x.key;
      if (cord) { c1[i] = cord[x.val]; val[i] = cord[x.val + 1]; } else val[i] = x.val;
    } else { key[i] = VSV_KEY_DEAD; val[i] = 0; if (cord) c1[i] = 0; }
  }
}
bool sl_use8(int nbits) {
  static const char* e8 = vsv_dbg_env("VSV_SLIM8");       // timing experiments / tests: "0" = 16-byte elements for the candidate sorts too
  return nbits <= 31 && !(e8 && e8[0] == '0');
}
}  // namespace

SortResult vsv_slim_sort_pairs(hipStream_t st, const uint64_t* key, const uint32_t* val, const uint32_t* d_n, int nbits, uint64_t* out_key, uint32_t* out_val,
                               uint32_t* d_live, const SlimWork& w, uint32_t* out_ord) {
  if (sl_use8(nbits)) {
    Slim8* r = sl_sort(st, SrcPairs8{key}, d_n, d_live, nbits, KeyCmp{0, 0}, (Slim8*)w.buf[3], (Slim8*)w.buf[4], w, w.cand_hint);
    sl_unpack_pairs8<<<w.grid, 256, 0, st>>>(r, d_live, d_n, val, out_key, out_val, out_ord);
    return SortResult{out_key, out_val};
  }
  Slim* r = sl_sort(st, SrcPairs{key, val}, d_n, d_live, nbits, KeyCmp{0, 0}, (Slim*)w.buf[3], (Slim*)w.buf[4], w, w.cand_hint);
  sl_unpack_pairs<<<w.grid, 256, 0, st>>>(r, d_live, d_n, out_key, out_val, out_ord);
  return SortResult{out_key, out_val};
}
SortResult vsv_slim_sort_pair_slots(hipStream_t st, const uint64_t* ckey, const uint32_t* crec, int qid_bits, int rec_bits, const uint32_t* d_n, int nbits,
                                    uint64_t* out_key, uint32_t* out_val, uint32_t* d_live, const SlimWork& w, const uint32_t* cord, uint32_t* out_c1) {
  if (sl_use8(nbits)) {
    Slim8* r = sl_sort(st, SrcPairSlots8{ckey, crec, qid_bits, rec_bits, d_n}, d_n, d_live, nbits, KeyCmp{0, 0}, (Slim8*)w.buf[3], (Slim8*)w.buf[4], w, w.cand_hint);
    sl_unpack_slots8<<<w.grid, 256, 0, st>>>(r, d_live, d_n, cord, out_key, out_val, out_c1);
    return SortResult{out_key, out_val};
  }
  Slim* r = sl_sort(st, SrcPairSlots{ckey, crec, qid_bits, rec_bits, d_n, cord}, d_n, d_live, nbits, KeyCmp{0, 0}, (Slim*)w.buf[3], (Slim*)w.buf[4], w, w.cand_hint);
  sl_unpack_pairs<<<w.grid, 256, 0, st>>>(r, d_live, d_n, out_key, out_val, cord ? out_c1 : nullptr);
  return SortResult{out_key, out_val};
}

void vsv_slim_rows(hipStream_t st, const void* elems, uint32_t n, const vsv_sig* s1in, vsv_sig* out) {
  if (n == 0) return;
  const uint32_t g = (n + 255) / 256;
  sl_rows_out<<<g < 4096 ? g : 4096, 256, 0, st>>>((const Slim*)elems, n, s1in, out);
}
