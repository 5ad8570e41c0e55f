// vsv_device.h — shared device-side definitions for the gfx950 kernels (wave64, CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/volcanosv.h"
#include "vsv_env.h"

#define VSV_WAVE 64

// device-side error bits (OR-ed into Counters::err by kernels)
enum : uint32_t {
  ERRB_EMPTY_CIGAR = 1u << 0,
  ERRB_REFEND = 1u << 1,
  ERRB_READLEN = 1u << 2,
  ERRB_UNSORTED = 1u << 3,
  ERRB_ZERODIV = 1u << 4,
  ERRB_CAPACITY = 1u << 5,
  ERRB_RANGE = 1u << 6,  // a 64-record batch spans >= 2^31 CIGAR ops
  ERRB_SEQLEN = 1u << 8,         // a walked record carries VSV_F_SEQ_MISMATCH (H:397-398)
  ERRB_CLR_FALLBACK = 1u << 9,   // a part too long for the gate state of the fused CLR scan: the run is repeated with the separate gate pass
  ERRB_SORT_FALLBACK = 1u << 7,  // a bucket of the bucket sort did not fit in LDS: the run is repeated with the LSD passes
  ERRB_MERGE_FALLBACK = 1u << 11, // a rank-and-merge sort of the element path met a window it cannot decide (thousands of slots within one shift): the run is repeated with the LSD passes
  ERRB_BATCH_FALLBACK = 1u << 13, // (host side) a part of the scan filed an overflow batch in a run that skipped their placement launch: the run is repeated with it
  ERRB_BUCKET1_SLOW = 1u << 14,  // (not an error) a bucket of the element path's first sort (one counting pass + an LDS sort per bucket) held more than fits in LDS and was sorted in global memory: the handle's next runs take the LSD passes
  ERRB_LOOKBACK = 1u << 12,      // a part of the scan waited for its predecessors' row counts beyond every plausible time (cigar_scan.hip, look-back): internal error
  ERRB_SLIM_FALLBACK = 1u << 10, // an svlen outside [0, 2^30) or a position of 2^30 and more met the element path (32-bit predicates): the run is repeated on rows
};

// One cache line of device counters, zeroed at the start of every run.
struct Counters {
  uint32_t n_pool;     // rows the emit pool would need (max shard use x shards); > capacity => VSV_E_CAPACITY
  uint32_t n_raw;      // min(n_pool, cap)
  uint32_t n_cand;     // split candidates
  uint32_t n_pairs;    // split pair slots
  uint32_t n_s1;       // rows entering stage 1 (n_raw + n_pairs)
  uint32_t n_alive1;   // alive rows after stage-1 sort
  uint32_t n_alive2;
  uint32_t n_alive3;   // == rows of T_MERGED
  uint32_t n_calls;
  uint32_t n_reads;
  uint32_t err;
  uint32_t n_long;     // (unused: long runs / stretches are taken over by the discovering lane's wave)
  uint32_t max_stretch; // longest pairing stretch one wave walked in this run: next run's hint for the round-based pairing
  uint32_t pad[3];
};

struct RecView {
  int64_t n_records;
  int64_t n_ops;
  const int32_t* pos;
  const int32_t* tid;
  const uint32_t* qid;
  const uint64_t* cigar_off;
  const uint8_t* mapq;
  const uint8_t* flag;
  const uint32_t* cigar;
  int32_t n_qids;
  int32_t tid_lo;   // keys carry tid - tid_lo
};

// CIGAR range [a, b) of record r, clamped to the caller's n_ops: offsets that do not ascend or run past the array are
// reported by the scan (ERRB_EMPTY_CIGAR), and no other kernel follows them outside the array meanwhile.
__device__ __forceinline__ void vsv_op_range(const RecView& rv, int64_t r, uint64_t& a, uint64_t& b) {
  const uint64_t n = (uint64_t)rv.n_ops;
  b = rv.cigar_off[r + 1]; if (b > n) b = n;
  a = rv.cigar_off[r]; if (a > b) a = b;
}

// ---- sort keys: [tid | hap | type | source | pos], list id in the high bits ------------------------
#define VSV_KEY_DEAD 0xFFFFFFFFFFFFFFFFull
#define VSV_POS_BIAS 65536   // keys order positions >= -65536 (split DEL positions can dip slightly below 0)
// key = (tid - tid_lo) << (pb+3) | hap << (pb+2) | type << (pb+1) | source << pb | (pos + bias); pb = position bits of the run
// (32 without a max_pos hint). Same ORDER as the oracle's keys; only the bit packing differs.
__host__ __device__ inline uint64_t vsv_kpos(int32_t pos) { return (uint32_t)(pos + VSV_POS_BIAS); }
__host__ __device__ inline uint64_t vsv_key_stage(const vsv_sig& s, int stage, int pb, int tid_lo) {
  if (s.meta & VSV_M_DEAD) return VSV_KEY_DEAD;
  uint64_t hap = (s.meta & VSV_M_HP2) ? 1 : 0, del = (s.meta & VSV_M_DEL) ? 1 : 0, sp = (s.meta & VSV_M_SPLIT) ? 1 : 0;
  uint64_t k = ((uint64_t)(uint32_t)(s.tid - tid_lo) << (pb + 3)) | vsv_kpos(s.pos);
  if (stage == 1) return k | (hap << (pb + 2)) | (del << (pb + 1)) | (sp << pb);   // list = (tid,hap,type,src)
  if (stage == 2) return k | (hap << (pb + 2)) | (del << (pb + 1));                // list = (tid,hap,type)
  if (stage == 3) return k | (hap << (pb + 2));                                    // list = (tid,hap)
  return k;                                                                        // (tid,pos)
}

// exact integer form of the reference's ratio predicates (SURVEY.md §7 hard part 3):
// shift<=max_shift, min(l)/max(l)>=0.5, DEL: (min(e)-max(s))/min(l)>=0.5  (H:200-226, 515-546)
__host__ __device__ inline bool vsv_match(const vsv_sig& a, const vsv_sig& b, int max_shift) {
  int64_t shift = (int64_t)a.pos - b.pos;
  if (shift < 0) shift = -shift;
  if (shift > max_shift) return false;
  int64_t l1 = a.svlen, l2 = b.svlen, mn = l1 < l2 ? l1 : l2, mx = l1 < l2 ? l2 : l1;
  if (2 * mn < mx) return false;
  if (a.meta & VSV_M_DEL) {
    int64_t s1 = a.pos, e1 = s1 + l1, s2 = b.pos, e2 = s2 + l2;
    int64_t ov = (e1 < e2 ? e1 : e2) - (s1 > s2 ? s1 : s2);
    if (2 * ov < mn) return false;
  }
  return true;
}

// 16-byte element of the large-table path (slim_path.hip): {stage-1 key, svlen, row | type}. The kernels that write the stage-1
// input rows (fold, split_eval) emit it next to the row when the run works on elements, so no pass re-reads the rows for it.
struct SlimOut { void* base; int pb, tid_lo, tid_bits; uint32_t* err; uint32_t* mm = nullptr; };     // base == nullptr: the run does not use elements; mm: where fold_elems leaves the position range (max kpos, max ~kpos)
#define VSV_SL_DEL 0x80000000u
__device__ __forceinline__ void vsv_slim_emit(const SlimOut& so, uint32_t i, const vsv_sig& v) {
  if (!so.base) return;
  const uint64_t k = vsv_key_stage(v, 1, so.pb, so.tid_lo);
  if (!(v.meta & VSV_M_DEAD) && ((so.pb < 32 && (vsv_kpos(v.pos) >> so.pb) != 0) || ((uint32_t)(v.tid - so.tid_lo) >> so.tid_bits) != 0))
    atomicOr(so.err, ERRB_RANGE);            // max_pos hint too small / tid outside [tid_lo, n_tids)
  if (!(v.meta & VSV_M_DEAD) && ((uint32_t)v.svlen >= (1u << 30) || (vsv_kpos(v.pos) >> 30) != 0)) atomicOr(so.err, ERRB_SLIM_FALLBACK);      // the element kernels compute in 32 bits
  reinterpret_cast<uint4*>(so.base)[i] = make_uint4((uint32_t)k, (uint32_t)(k >> 32), (uint32_t)v.svlen, i | ((v.meta & VSV_M_DEL) ? VSV_SL_DEL : 0u));
}

// ---- decoupled look-back on ONE 32-bit value per block: an exclusive scan (sum or maximum) inside one launch -----------------
// Every block publishes its own value, looks back over its predecessors' words until it meets one that holds an inclusive prefix, and
// publishes its own inclusive prefix. A word is valid by itself: [63:40] epoch of the run (words of earlier runs are "not yet": the
// buffer is zeroed when allocated, never per run), [39:38] kind (1 own value / 2 inclusive prefix), [31:0] value; relaxed agent-scope
// atomics, no fence. Workgroups are dispatched in index order and a block waits for lower blocks only, so the waits end; they are
// bounded all the same (ERRB_LOOKBACK). Called by all 64 lanes of ONE wave of the block, wave-uniform arguments.
__device__ __forceinline__ uint64_t vsv_lb1_ld(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void vsv_lb1_st(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool MAX>
__device__ __forceinline__ uint32_t vsv_lb1_exclusive(uint64_t* __restrict__ words, uint32_t epoch, uint32_t idx, uint32_t own, int lane, uint32_t* err) {
  const uint64_t tag = (uint64_t)epoch << 40;
  if (lane == 0) vsv_lb1_st(words + idx, tag | (1ull << 38) | own);
  uint32_t acc = 0;
  if (idx != 0) {
    int64_t j = (int64_t)idx - 1;
    uint32_t spins = 0;
    for (;;) {
      const int64_t pj = j - lane;
      uint64_t w = tag | (2ull << 38);             // in front of block 0: an empty inclusive prefix
      if (pj >= 0) w = vsv_lb1_ld(words + pj);
      const uint32_t kind = (uint32_t)(w >> 40) == epoch ? (uint32_t)(w >> 38) & 3u : 0u;
      const uint64_t m_inv = __ballot(kind == 0u), m_pre = __ballot(kind == 2u);
      const uint32_t lp = m_pre ? (uint32_t)__builtin_ctzll(m_pre) : 64u, li = m_inv ? (uint32_t)__builtin_ctzll(m_inv) : 64u;
      if (li < lp) {                               // a block between this one and the nearest prefix has not published yet
        if (++spins > (1u << 20)) { if (lane == 0) atomicOr(err, ERRB_LOOKBACK); break; }
        __builtin_amdgcn_s_sleep(8);
        continue;
      }
      uint32_t v = (uint32_t)lane <= lp ? (uint32_t)w : 0u;
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64); v = MAX ? (v > o ? v : o) : v + o; }
      acc = MAX ? (acc > v ? acc : v) : acc + v;
      if (lp < 64u) break;
      j -= 64;
    }
  }
  const uint32_t incl = MAX ? (acc > own ? acc : own) : acc + own;
  if (lane == 0) vsv_lb1_st(words + idx, tag | (2ull << 38) | incl);
  return acc;
}

// ---- launch wrappers implemented in the .hip files ------------------------------------------------
struct SortWork {       // scratch for vsv_radix_sort_pairs
  uint64_t* key_alt;
  uint32_t* val_alt;
  uint32_t* hist;       // [2048 * max_tiles]
  int64_t max_items;
  // per-pass digit totals for the multi-block scan, zeroed once per run by the caller
  uint32_t* totals;     // [max_passes * 2048]
  int* pass_cursor;     // host-side index of the next free totals slot (reset per run)
  int max_passes;
  bool small_tiles;     // 1024-row instead of 4096-row tiles; chosen by the caller from the row count of the previous run (both exact)
  int bucket_bits;      // 8..11: bucket sort (one counting pass + LDS sort per bucket) with 2^bits buckets; 0: LSD passes only
  uint32_t* err;        // device error word (ERRB_SORT_FALLBACK)
  bool shared_gpu;      // several handles keep the GPU busy (vsv_params.split_overlap = OFF): LDS sort in 256-thread workgroups
  uint64_t hint_rows;   // rows of the largest table of the handle's previous run (0 = unknown): speed decisions only
  // bucket sorts without a scan launch: zeroed slots of [VSV_RS_MAX_GROUPS][2048] sums over groups of tiles, one per sort
  uint32_t* groups;
  int* group_cursor;
  int max_group_slots;
  bool slots_ok = false; // the caller's consumers read nothing but the key of a dead row: the two-launch bucket sort may be used (rs_slot_scatter)
};
constexpr int VSV_RS_MAX_GROUPS = 64;
struct StageBufs {
  uint64_t* key;        // sort keys of the current stage (kept sorted for cluster / pair kernels)
  uint32_t* idx;
  int32_t* cl;          // cluster ids / pairing state
  int tid_lo, tid_bits; // keys carry tid - tid_lo on tid_bits bits
  uint64_t kmax;        // exclusive upper bound of the alive stage keys (0 = unknown): balances the bucket sort
  int grid;             // blocks of the row-parallel kernels (sized from the row counts of the handle's previous run; the kernels grid-stride)
};

// radix_sort.hip: stable LSD radix sort of (key,val) pairs on bits [0,nbits); n on the device. The result is left in
// whichever of the two buffer pairs the last pass wrote (returned); nothing is copied back.
struct SortResult { uint64_t* key; uint32_t* val; };
SortResult vsv_radix_sort_pairs(hipStream_t st, uint64_t* key, uint32_t* val, uint64_t* key_scratch, uint32_t* val_scratch,
                                const uint32_t* d_n, int64_t max_n, int nbits, const SortWork& w, uint64_t kmax = 0);
int64_t vsv_radix_hist_entries(int64_t max_n);
SortResult vsv_bucket_sort_pair_slots(hipStream_t st, const uint64_t* ckey, const uint32_t* crec, int qid_bits, int rec_bits, const uint32_t* d_n,
                                      uint64_t* okey, uint32_t* oval, uint64_t* key_scratch, uint32_t* val_scratch, int64_t max_n, int nbits, const SortWork& w);
// rows sorted by their stage key in 4 launches (bucket sort with the keys taken from the rows and the rows gathered by the LDS sort);
// nullptr = not applicable, use build-keys + vsv_radix_sort_pairs + gather
const uint64_t* vsv_bucket_sort_sigs(hipStream_t st, const vsv_sig* in, const uint32_t* d_n, int stage, int pb, int tid_lo, int tid_bits, int nbits,
                                     uint64_t kmax, vsv_sig* sorted, uint64_t* key_out, uint32_t* d_alive, uint32_t* n_long, int32_t* fill,
                                     const SortWork& w, int64_t max_n);
const uint64_t* vsv_bucket_sort_calls(hipStream_t st, const vsv_call* in, const uint32_t* d_n, int pb, int tid_lo, int nbits, uint64_t kmax,
                                      vsv_call* sorted, uint64_t* key_out, uint32_t* d_alive, uint32_t* n_long, const SortWork& w, int64_t max_n,
                                      const vsv_sig* merged = nullptr, const int32_t* st2 = nullptr);

// cigar_scan.hip
void vsv_launch_stream_read(hipStream_t st, const void* src, size_t bytes, uint32_t* sink);
void vsv_launch_stream_copy(hipStream_t st, const void* src, void* dst, size_t bytes);
int vsv_cigar_parts(int64_t n_ops, int ops_per_part);
void vsv_scan_u32_exclusive(hipStream_t st, const uint32_t* in, int n, uint32_t* out, uint32_t* tmp);
struct LongScanBufs {   // scratch of the scan launcher beyond the part tables
  void* agg; uint32_t* carry_r; uint32_t* carry_q; void* tile_sum;   // long-record scan (cigar_scan_long): aggregates, carries, tile sums
  uint32_t* tile_cnt; int tile_cnt_cap;    // read-shaped scan: zeroed per run, the waves add their part counts per 2048-part tile
  bool arena_zeroed;                       // the caller zeroed shard_cnt together with the counters
  bool clr_fused;                          // CLR, read-shaped scan: the gate is computed inside the scan (flags arrive ungated)
  uint64_t* lbw; uint32_t epoch;           // look-back words of the placement kernel's scan (3 per block of parts), and this run's epoch
  void* prec;                              // long-record scan: {location, descriptors} of every part's last batch
  void* fused_rows; SlimOut so;            // place the rows here (the stage-1 input table) with their elements, not in `raw`
  bool skip_batches;                       // no launch for the overflow batches: the handle's last runs filed none (Counters::pad[0] tells whether this one did)
};
void vsv_launch_place(hipStream_t st, const RecView& rv, const vsv_params& p, int n_parts, vsv_sig* pool, uint64_t* pool_key, uint32_t cap, uint32_t* part_count,
                      uint32_t* part_off, vsv_sig* rows, Counters* ctr, uint32_t* shard_cnt, const LongScanBufs& lb, uint32_t epoch, const SlimOut& so,
                      bool with_batches = true);
int vsv_scan_parts(const RecView& rv, const vsv_params& p, int ops_per_part);
size_t vsv_lookback_bytes(int64_t n_ops, int ops_per_part);
bool vsv_scan_is_long(const RecView& rv, const vsv_params& p);
size_t vsv_long_scan_bytes(int64_t n_ops, int which, int ops_per_part);
void vsv_launch_cigar_scan(hipStream_t st, const RecView& rv, const vsv_params& p, uint32_t* part_rb, int n_parts,
                           int ops_per_part, vsv_sig* pool, uint64_t* pool_key, uint32_t cap, uint32_t* part_count,
                           uint32_t* part_off, uint32_t* scan_tmp, vsv_sig* raw, Counters* ctr, uint32_t* shard_cnt,
                           hipEvent_t ev0, hipEvent_t ev1, const LongScanBufs& lb);

struct SlimWork;
// sig_stages.hip
void vsv_launch_clr_gate(hipStream_t st, const RecView& rv, uint8_t* gflag, Counters* ctr);   // CLR: gated flag bytes for the scan
void vsv_launch_fold(hipStream_t st, const vsv_sig* raw, vsv_sig* s1in, const RecView& rv, const vsv_params& p, Counters* ctr, int grid,
                     const SlimOut& so = SlimOut{nullptr, 0, 0, 0, nullptr});
void vsv_launch_fold_elems(hipStream_t st, vsv_sig* rows, Counters* ctr, int grid, const SlimOut& so);   // the fold in place, on rows placed with their elements
struct SplitSorted {      // candidates by name / pairs by record
  const uint64_t* ckey; const uint32_t* crec; const uint64_t* okey; const uint32_t* oval;
  const uint32_t* oc1 = nullptr; const void* cinfo = nullptr;      // large read-shaped inputs: a slot's two candidates (oc1, oval) into cinfo
  bool live_only = false;   // the slot sort dropped the dead slots: Counters::n_pairs live ones, in front (else n_cand slots, dead ones behind)
};
struct CandBufs { void* cinfo; uint32_t* cord; uint32_t* oc1; };     // 32 / 4 / 4 bytes per candidate (null: the record arrays are gathered per pair)
struct CandLb {              // the two scans inside the candidate kernels: look-back words (one per 2048 records each), this run's epoch, and
  uint64_t* qwords; uint64_t* cwords; uint32_t epoch;      // the name table of the NEXT run, which this run's kernel clears (its first
  uint32_t* tab_next; uint32_t clear_words;                 // clear_words words: what the run before the last left in it)
};
SplitSorted vsv_launch_split_candidates(hipStream_t st, const RecView& rv, const vsv_params& p, int n_tids, uint32_t* tab, uint32_t tab_size,
                                        uint32_t* blk_cnt, uint32_t* blk_off, uint32_t* scan_tmp, uint64_t* ckey, uint32_t* crec,
                                        uint64_t* okey, uint32_t* oval, uint64_t* key2, uint32_t* idx2, const SortWork& sw,
                                        uint32_t cap, Counters* ctr, uint8_t* cmask, int grid, const struct SlimWork* slim = nullptr,
                                        const CandBufs& cb = CandBufs{nullptr, nullptr, nullptr}, int phase = 0,
                                        const CandLb& lb = CandLb{nullptr, nullptr, 0, nullptr, 0});
void vsv_launch_split_eval(hipStream_t st, const RecView& rv, const vsv_params& p, int n_tids, const SplitSorted& so, vsv_sig* s1in,
                           uint32_t cap, Counters* ctr, int grid, const SlimOut& sl = SlimOut{nullptr, 0, 0, 0, nullptr});
// returns the sorted key array (kept for the cluster / pair kernel that follows)
const uint64_t* vsv_launch_sort_stage(hipStream_t st, const vsv_sig* in, const uint32_t* d_n, int stage, int pb, int nbits,
                                      vsv_sig* sorted, uint32_t* d_alive, const StageBufs& b, const SortWork& sw, int64_t cap,
                                      Counters* ctr, int32_t* fill_minus1 = nullptr);
void vsv_launch_cluster(hipStream_t st, const vsv_sig* sorted, const uint64_t* sorted_key, const uint32_t* d_alive, int max_shift,
                        int pb, vsv_sig* out, const StageBufs& b, uint64_t* long_list, Counters* ctr);
void vsv_launch_pair(hipStream_t st, const vsv_sig* merged, const uint64_t* merged_key, const uint32_t* d_alive3, int pair_shift, int pair_window,
                     vsv_call* calls_tmp, vsv_call* calls, uint32_t* d_ncalls, const StageBufs& b, uint64_t* key2, uint32_t* idx2,
                     const SortWork& sw, int pb, int nbits, int64_t cap, Counters* ctr, bool dense);

// slim_path.hip: the stages behind the scan on 16-byte elements, for tables of 10^6-10^7 rows (capi.hip picks the path per run)
struct SlimWork {
  void* buf[6];         // element buffers, cap x 16 bytes each: 0 / 1 scratch of the sorts, 2 stage-1 clusters (kept for VSV_T_CLUSTER1),
                        // 3 stage-2 clusters / pairing reservations, 4 / 5 merged elements (kept for VSV_T_MERGED and the calls)
  int64_t cap;          // elements per buffer (= row capacity of the handle)
  int64_t rows_hint, cand_hint;   // signature rows / split candidates of the handle's previous run + 25 % (a first run: cap) - speed decisions only
  uint32_t* hist;       // [1024 * tiles of 4096]
  uint32_t* totals;     // zeroed per-pass digit totals, 2048 entries per slot (SortWork::totals)
  int* pass_cursor;
  int grid;             // blocks of the row-parallel kernels
  int32_t* cl;          // one word per slot: long-run cluster state, then the pairing state
  uint32_t* hj;         // pairing: first candidate of every hp1 row | stretch-start flag
  uint32_t* done1;      // pairing in rounds: decided flags
  bool merge_sorts;     // sorts 2 / 3 / calls as rank-inside-the-class + merge (sl_merge_sort); false: the LSD passes
  bool bucket_sort1;    // sort 1 as one counting pass into position buckets + an LDS sort per bucket (sl_bucket_sort1); false: the LSD passes
  const uint32_t* mm;   // position range of the stage-1 cigar elements where the in-place fold measured it (max kpos, max ~kpos), or nullptr
  double split_share;   // expected share of split-list elements among the stage-1 elements (previous run / cold wait): their part of the buckets
  uint32_t* err;        // device error word (ERRB_MERGE_FALLBACK)
};
int vsv_slim_sort_passes(int nbits);
void* vsv_slim_stage1(hipStream_t st, const vsv_sig* s1in, const uint32_t* d_n_s1, uint32_t* d_alive1, int pb, int tid_lo, int tid_bits, int cluster_shift,
                      const SlimWork& w, Counters* ctr, bool prebuilt, const void** sorted1, void** ctl2);
void* vsv_slim_merge(hipStream_t st, const void* e2, const void* sorted1, void* ctl2, const uint32_t* d_alive1, uint32_t* d_alive2, uint32_t* d_alive3, int pb, int tid_bits, int shift1, int cluster_shift,
                     const SlimWork& w);
void vsv_slim_pair(hipStream_t st, const void* merged, const uint32_t* d_alive3, uint32_t* d_ncalls, int pb, int tid_bits, int pair_shift, int pair_window,
                   const vsv_sig* s1in, vsv_call* calls, bool dense, const SlimWork& w, Counters* ctr);
void vsv_slim_rows(hipStream_t st, const void* elems, uint32_t n, const vsv_sig* s1in, vsv_sig* out);
// (key, value) arrays through the element passes (element buffers 3 / 4 as scratch): the split stage's candidate sorts on large inputs.
// Dead keys are dropped by the first pass and written back behind the live ones; *d_live = the live count.
SortResult vsv_slim_sort_pairs(hipStream_t st, const uint64_t* key, const uint32_t* val, const uint32_t* d_n, int nbits, uint64_t* out_key, uint32_t* out_val,
                               uint32_t* d_live, const SlimWork& w, uint32_t* out_ord = nullptr);
SortResult vsv_slim_sort_pair_slots(hipStream_t st, const uint64_t* ckey, const uint32_t* crec, int qid_bits, int rec_bits, const uint32_t* d_n, int nbits,
                                    uint64_t* out_key, uint32_t* out_val, uint32_t* d_live, const SlimWork& w, const uint32_t* cord = nullptr,
                                    uint32_t* out_c1 = nullptr);

// bnd.hip
void vsv_launch_bnd_segments(hipStream_t st, const vsv_segments& s, const vsv_bnd_params& p, vsv_bnd* cand, uint32_t cap, Counters* ctr);
void vsv_launch_bnd_pair(hipStream_t st, const vsv_bnd* cand, const int32_t* contig_rank, int rank_bits, const vsv_bnd_params& p,
                         vsv_bnd* sorted, vsv_bnd* calls, Counters* ctr, const StageBufs& b, const SortWork& sw, int64_t cap);

// support.hip: FP_filter_v1.eval_sig as a sorted window join (wave per call)
void vsv_launch_support_join(hipStream_t st, const int32_t* call_pos, const int32_t* call_len, int64_t n_calls, const int32_t* sig_pos,
                             const int32_t* sig_len, int64_t n_sigs, const vsv_support_params& p, uint32_t* support, uint32_t* err);
void vsv_launch_cov_ins(hipStream_t st, const int32_t* call_pos, int64_t n_calls, const int32_t* sig_pos, const int32_t* sig_len,
                        int64_t n_sigs, int32_t flanking, int64_t* cov, uint32_t* err);
void vsv_launch_cov_del(hipStream_t st, const int32_t* call_start, const int32_t* call_end, int64_t n_calls, const int32_t* sig_start,
                        const int32_t* sig_end, const int32_t* sig_svlen, int64_t n_sigs, int32_t flanking, int64_t* cov,
                        uint32_t* err /* [0] error bits, [1] max signature span (scratch, zeroed by the caller) */);

// cutesv.hip: sig_extract.py analysis_split_read (INS/DEL branches), lane per read
void vsv_launch_cutesv_split(hipStream_t st, const vsv_segments& sg, const int32_t* read_len, const uint32_t* read_rec, int sv_size,
                             int max_size, int max_parts, vsv_sig* out, uint32_t cap, uint32_t* cnt, uint32_t* off, uint32_t* scan_tmp,
                             Counters* ctr, uint8_t* tra);

// redundancy.hip: remove_redundancy.py candidate pairs / DEL predicate, and the edit-distance similarity of INS pairs
void vsv_launch_rr_pairs(hipStream_t st, bool write, const int32_t* pos, const int32_t* svlen, int64_t n, int is_del, int64_t dist,
                         double size_thr, double overlap_thr, uint32_t* cnt, const uint32_t* off, uint32_t* pairs, uint32_t cap);
void vsv_launch_rr_edit_sim(hipStream_t st, const uint32_t* pairs, uint32_t n_pairs, const uint8_t* seq, const uint64_t* seq_off,
                            const uint64_t* hoff, int32_t* hbuf, double seq_sim_thr, uint8_t* flag, uint32_t* dist_out);

// inflate.hip: BGZF members (raw deflate) -> bytes, one lane per member
void vsv_launch_bgzf_inflate(hipStream_t st, const uint8_t* comp, const uint64_t* comp_off, const uint64_t* out_off, int64_t n, uint8_t* out,
                             int32_t* status);
void vsv_crc32_tables(uint32_t* t);      // 256 + 17 x 32 words for vsv_launch_bgzf_crc32 (host-side construction)
void vsv_launch_bgzf_crc32(hipStream_t st, const uint8_t* data, const uint64_t* out_off, int64_t n, const uint32_t* tables, uint32_t* crc);

// bam_device.hip: BAM record parse on the device
void vsv_bamdev_speculate(hipStream_t st, const uint8_t* s, const uint64_t* moff, int64_t n_members, uint64_t first, int32_t n_ref, uint64_t* spec);
void vsv_bamdev_chain(hipStream_t st, bool write, const uint8_t* s, const uint64_t* moff, int64_t n_members, const uint64_t* spec, uint32_t* count,
                      uint64_t* land, const uint64_t* base, uint64_t* rec_off);
void vsv_bamdev_fields(hipStream_t st, const uint8_t* s, const uint64_t* rec_off, int64_t n, int32_t want_tid, int32_t* pos, int32_t* tid, uint8_t* mapq,
                       uint8_t* flag, uint32_t* l_seq, uint32_t* sam_flag, uint32_t* n_cig_out, uint64_t* cg_src, uint64_t* hash, uint32_t* keep, uint32_t* err);
void vsv_bamdev_scan64(hipStream_t st, const uint32_t* v, const uint32_t* keep, int64_t n, uint64_t* sums, uint64_t* out, uint64_t* total);
void vsv_bamdev_emit(hipStream_t st, const uint8_t* s, int64_t n, const uint32_t* keep, const uint32_t* kidx, const int32_t* pos, const int32_t* tid,
                     const uint8_t* mapq, const uint8_t* flag, const uint32_t* l_seq, const uint32_t* sam_flag, const uint32_t* n_cig_out,
                     const uint64_t* cg_src, const uint64_t* hash, const uint64_t* cig_off_in, const uint64_t* rec_off, int32_t* o_pos, int32_t* o_tid,
                     uint8_t* o_mapq, uint8_t* o_flag, uint32_t* o_l_seq, uint32_t* o_sam_flag, uint64_t* o_cig_off, uint32_t* o_cigar, uint64_t* o_hash,
                     uint64_t* w_rec_off, uint64_t k0, uint64_t c0, uint64_t ops_hint);
void vsv_bamdev_iota(hipStream_t st, uint32_t* p, int64_t n);
void vsv_bamdev_mark_first(hipStream_t st, const uint64_t* skey, const uint32_t* sval, int64_t n, uint32_t* is_first);
void vsv_bamdev_assign(hipStream_t st, const uint64_t* skey, const uint32_t* sval, int64_t n, const uint32_t* first_rank, const uint8_t* names,
                       const uint64_t* nm_off, const uint32_t* nm_len, uint32_t* qid, uint32_t* err);
void vsv_bamdev_win_name_lens(hipStream_t st, const uint8_t* s, const uint64_t* w_rec_off, int64_t nk, uint32_t* len);
void vsv_bamdev_win_name_store(hipStream_t st, const uint8_t* s, const uint64_t* w_rec_off, const uint32_t* loff, int64_t nk, uint64_t k0, uint64_t n0,
                               uint8_t* names, uint64_t* nm_off, uint32_t* nm_len);
void vsv_bamdev_win_seq_find(hipStream_t st, const uint8_t* s, const uint64_t* w_rec_off, int64_t nk, uint64_t* seq_off, uint32_t* seq_len);
void vsv_bamdev_win_seq_store(hipStream_t st, const uint8_t* s, const uint64_t* seq_off, const uint32_t* seq_len, const uint32_t* loff, int64_t nk,
                              uint64_t k0, uint64_t q0, uint8_t* blob, uint64_t* rec_seq_off);
void vsv_bamdev_seq_slices(hipStream_t st, const uint8_t* blob, const uint64_t* rec_seq_off, const uint32_t* l_seq, const uint32_t* rec,
                           const uint32_t* start, const uint32_t* len, const uint8_t* rev, int64_t n, const uint64_t* out_off, uint8_t* out);
void vsv_bamdev_win_sa_find(hipStream_t st, const uint8_t* s, const uint64_t* w_rec_off, int64_t nk, uint64_t* sa_off, uint32_t* sa_len);
void vsv_bamdev_win_sa_store(hipStream_t st, const uint8_t* s, const uint64_t* sa_off, const uint32_t* sa_len, const uint32_t* loff, int64_t nk,
                             uint64_t s0, uint8_t* blob);
void vsv_bamdev_name_lens(hipStream_t st, const uint32_t* nm_len, const uint32_t* is_first, int64_t n, uint32_t* len);
void vsv_bamdev_name_copy(hipStream_t st, const uint8_t* names, const uint64_t* nm_off, const uint32_t* nm_len, const uint32_t* is_first, const uint32_t* noff,
                          int64_t n, uint8_t* blob);

// support.hip: GT-correction joins
void vsv_launch_gt_support(hipStream_t st, const int32_t* vpos, const int32_t* vlen, const int32_t* blo, const int32_t* bhi, int64_t nv, const int32_t* spos,
                           const int32_t* slen, const int32_t* scnt, double shift_ratio, double size_sim, int64_t* sum, int32_t* lo, int32_t* hi);
void vsv_launch_span_count(hipStream_t st, const RecView& rv, int32_t* rend, uint32_t* max_span, const int32_t* qt, const int32_t* qa, const int32_t* qb,
                           int64_t nq, uint32_t* out);
