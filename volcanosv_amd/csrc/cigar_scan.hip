// cigar_scan.hip — K0 partition + K1 cigar_scan_emit + ordered placement of the emit stream.
//
// Replaces the per-record Python walk of extract_sig_from_cigar
//   (Large_INDEL/extract_contig_signature_Hifi.py:53-85, extract_reads_signature.py:47-83,
//    Complex_SV/svim-asm-1.0.2/src/svim_asm/SVIM_intra.py:8-30)
// and the filter loop around it (H:386-400, RS:107-125, SV/SVIM_COLLECT.py:67).
//
// Design (gfx950, wave64; HBM-bound, no MFMA):
//  * The packed CIGAR array (u32 len<<4|op) is one flat stream. K0 cuts it into parts of ~ops_per_part
//    ops at record boundaries (binary search over cigar_off); one wavefront owns one part, so no
//    inter-wave communication exists anywhere in K1.
//  * A wave walks its part in 256-op chunks: every lane loads one aligned dwordx4 (4 ops, 1 KiB per
//    wave-instruction, fully coalesced), four chunks in flight. The streaming path only asks "does this chunk
//    hold an I/D op of at least min_svlen (or an op the table forbids)?": per op one shift of a constant mask,
//    a sign test and one compare (~25 VALU per chunk), so the kernel is bound by HBM, not by issue slots.
//  * Everything else is lazy, in a wave-uniform slow path for chunks with a candidate (1 in 14 on read-shaped
//    input): decode with bit-field extracts against constant op masks, ONE plain (unsegmented) DPP prefix sum
//    per advance (mod 2^32, differences are exact), record lookup in a 64-record window of the part's
//    record-start table (LDS), and P(op) - P(record start) through a per-record checkpoint that is advanced at
//    the end of every slow chunk (a record that started in an earlier chunk re-reads at most the L2-hot chunks
//    since its checkpoint). A signature's reference position is pos[rec] + (P(op) - P(record start)).
//  * Emission applies the mapq/hp filters (record header through the scalar cache, kept while consecutive
//    signatures share the record) and appends to a pool sharded over 256 cursors. Each row carries (part,
//    ordinal), and place_raw scatters rows to part_off[part]+ordinal, so T_RAW is in (record, op) order whatever
//    the atomic order was.
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "vsv_device.h"

namespace {

// ---- op tables as 16-bit masks (bit op set => op advances / emits) ---------------------------------
// contig (H:72-85): M both; S query; D ref+emit; I query+emit; everything else ignored.
// reads  (RS:66-81): M,=,X both; N ref. svim (SVIM_intra.py:13-29): M,=,X both; no H offset.
template <int CLS> struct OpTab;
template <> struct OpTab<0> { static constexpr uint32_t REF = 0x005, QRY = 0x013, BAD = 0x188; static constexpr bool HC = true; };
template <> struct OpTab<1> { static constexpr uint32_t REF = 0x18D, QRY = 0x193, BAD = 0x000; static constexpr bool HC = true; };
template <> struct OpTab<2> { static constexpr uint32_t REF = 0x185, QRY = 0x193, BAD = 0x000; static constexpr bool HC = false; };
// sig_extract.py parse_read (SE:449-472): M,=,X,D advance shift_del / shift_ins; every op but D advances shift_ins_read.
template <> struct OpTab<3> { static constexpr uint32_t REF = 0x185, QRY = 0x1FB, BAD = 0x000; static constexpr bool HC = false; };
constexpr uint32_t EMIT_MASK = 0x006;  // I(1), D(2)

constexpr uint32_t rev32(uint32_t v) {
  uint32_t r = 0;
  for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i);
  return r;
}

__device__ __forceinline__ uint32_t bfe_mask(uint32_t table, uint32_t op) {
  // 0xFFFFFFFF if bit `op` of table is set else 0 (v_bfe_i32 with width 1 sign-extends)
  return (uint32_t)__builtin_amdgcn_sbfe(table, op, 1);
}

// ---- DPP inclusive prefix sum over the 64 lanes (row_shr 1,2,4,8 then row_bcast15/31) --------------
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) {
  // lanes without a source (or masked rows) receive 0
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
  x += dpp0<0x111>(x);            // row_shr:1
  x += dpp0<0x112>(x);            // row_shr:2
  x += dpp0<0x114>(x);            // row_shr:4
  x += dpp0<0x118>(x);            // row_shr:8
  x += dpp0<0x142, 0xA>(x);       // row_bcast:15 -> rows 1,3
  x += dpp0<0x143, 0xC>(x);       // row_bcast:31 -> rows 2,3
  return x;
}

__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}

// ---- K0: part boundaries -----------------------------------------------------------------------
// rb[p] = first record whose start offset >= p*ops_per_part; rb[n_parts] = n_records.
// first r in [0, n] with cigar_off[r] >= target. Interpolated guess (records are about equally long), a galloping bracket around
// it, then the bisection: ~12 dependent loads instead of log2(n) = 23 on a 10 M-record shard. Invariant: cigar_off[lo] < target
// (or lo = -1) and cigar_off[hi] >= target (or hi = n); whatever the offsets are, the result lies in [0, n].
__device__ __forceinline__ int64_t first_record_at(const uint64_t* __restrict__ cigar_off, int64_t n_records, uint64_t target, int64_t n_ops) {
  int64_t g = n_ops > 0 ? (int64_t)((double)target / (double)n_ops * (double)n_records) : 0;
  g = g < 0 ? 0 : g > n_records ? n_records : g;
  int64_t lo, hi;
  if (cigar_off[g] >= target) {
    hi = g; lo = g;
    for (int64_t step = 64; ; step *= 8) { lo = hi - step; if (lo < 0) { lo = -1; break; } if (cigar_off[lo] < target) break; hi = lo; }
  } else {
    lo = g; hi = g;
    for (int64_t step = 64; ; step *= 8) { hi = lo + step; if (hi >= n_records) { hi = n_records; break; } if (cigar_off[hi] >= target) break; lo = hi; }
  }
  while (hi - lo > 1) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (cigar_off[mid] >= target) hi = mid; else lo = mid;
  }
  return hi;
}
// a search per part: fine up to a few hundred thousand parts (config 2: 80 k parts, 13 us)
__global__ void partition_search(const uint64_t* __restrict__ cigar_off, int64_t n_records, uint32_t* __restrict__ rb,
                                 int n_parts, int ops_per_part, int64_t n_ops, int stride) {
  const int64_t p = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * stride;
  if (p > n_parts) return;
  rb[p] = p == n_parts ? (uint32_t)n_records : (uint32_t)first_record_at(cigar_off, n_records, (uint64_t)p * (uint64_t)ops_per_part, n_ops);
}
// Millions of parts (config 3: 2.2 M): a search per part touches a different cache line per lane and step — 1.7 GB of traffic,
// 0.52 ms. Instead partition_search finds every 64th boundary, and here one wave per 64 consecutive parts reads the offsets between
// its two known boundaries once (64 parts x ~22 records on read-shaped input), coalesced, into LDS, where every lane bisects. A range
// that does not fit (records of a few ops) is searched the old way.
constexpr int PK_RANGE = 1024;
__global__ __launch_bounds__(64) void partition_fill(const uint64_t* __restrict__ cigar_off, int64_t n_records, uint32_t* __restrict__ rb,
                                                     int n_parts, int ops_per_part, int64_t n_ops) {
  __shared__ uint64_t sh[PK_RANGE];
  const int lane = threadIdx.x;
  const int64_t p0 = (int64_t)blockIdx.x * 64, p = p0 + lane;
  const int64_t ra = rb[p0], rb_end = p0 + 64 >= n_parts ? n_records : (int64_t)rb[p0 + 64];
  const int64_t cnt = rb_end - ra + 1;                       // offsets [ra, rb_end]: every answer of the wave lies there
  const bool ranged = cnt >= 1 && cnt <= 64 * (int64_t)PK_RANGE;   // (wave-uniform; cnt < 1 only if the offsets are not monotone)
  const bool mine = lane != 0 && p < n_parts;                 // (every 64th boundary is partition_search's)
  const uint64_t target = (uint64_t)p * (uint64_t)ops_per_part;
  int64_t res = -1;
  if (ranged) {
    for (int64_t base = 0; base < cnt; base += PK_RANGE) {   // the range in LDS-sized pieces; a lane's answer lies in the first piece
      const int64_t m = cnt - base < PK_RANGE ? cnt - base : PK_RANGE;     // whose last offset reaches its target
      for (int64_t i = lane; i < m; i += 64) sh[i] = cigar_off[ra + base + i];
      __builtin_amdgcn_wave_barrier();
      if (mine && res < 0 && sh[m - 1] >= target) {
        int64_t lo = -1, hi = m - 1;                          // sh[lo] < target <= sh[hi] with sh[-1] = -inf
        while (hi - lo > 1) { const int64_t mid = lo + ((hi - lo) >> 1); if (sh[mid] >= target) hi = mid; else lo = mid; }
        res = ra + base + hi;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (p == n_parts && lane != 0) { rb[p] = (uint32_t)n_records; return; }
  if (!mine) return;
  if (res < 0) res = first_record_at(cigar_off, n_records, target, n_ops);   // offsets that are not monotone, or a huge range
  if (res > n_records) res = n_records;
  rb[p] = (uint32_t)res;
}

// Emit pool = K1_SHARDS independent sub-pools (cursor s on its own 64-byte line). A single cursor word saturates at
// ~88 returning atomics/us (MI355X_MICROARCH.md "dequeue"), which made 88 k emissions cost 0.8 ms; a part uses
// shard = part % K1_SHARDS and takes its slots in one allocation when it has streamed its ops.
constexpr int K1_SHARDS = 256;
struct EmitCtx {
  vsv_sig* pool;         // K1_SHARDS shards of shard_cap rows
  uint32_t* shard_cnt;   // [K1_SHARDS * 16]: cursor s at [16 s]; [8]: cursor of the overflow list
  uint32_t shard_cap;    // rows per shard
  Counters* ctr;
  uint2* prec;           // per part: {location, rows} of its last (normally only) batch
  uint4* batches; uint32_t batch_cap;     // overflow list {part, first ordinal, location, rows}
};
constexpr int K1E_STAGE = 32;      // rows a wave of the read-shaped scan stages before a batch leaves (1 KiB)

// five wave-uniform dword loads through the scalar cache (lgkmcnt): they do not drain the in-flight vector loads
__device__ __forceinline__ void sload5(const void* p0, const void* p1, const void* p2, const void* p3, const void* p4,
                                       uint32_t& v0, uint32_t& v1, uint32_t& v2, uint32_t& v3, uint32_t& v4) {
  asm volatile(
      "s_load_dword %0, %5, 0x0\n\ts_load_dword %1, %6, 0x0\n\ts_load_dword %2, %7, 0x0\n\ts_load_dword %3, %8, 0x0\n\t"
      "s_load_dword %4, %9, 0x0\n\ts_waitcnt lgkmcnt(0)"
      : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3), "=&s"(v4)
      : "s"(p0), "s"(p1), "s"(p2), "s"(p3), "s"(p4)
      : "memory");
}
__device__ __forceinline__ uint32_t byte_of(uint32_t word, const void* p) { return (word >> (8u * ((uintptr_t)p & 3u))) & 0xFFu; }
__device__ __forceinline__ const void* align4(const void* p) { return (const void*)((uintptr_t)p & ~(uintptr_t)3); }

// ---- K1 ------------------------------------------------------------------------------------------
// Per-wave state while streaming a part:
//   chunk grid    256-op chunks on absolute 16-byte boundaries, continuous over the whole part. DEPTH chunks
//                 (DEPTH KiB per wave) are always in flight: the loop is unrolled by DEPTH so no in-flight register is
//                 ever copied, and chunk loads go through a raw buffer descriptor clipped to the part.
//   record table  the part's record starts (relative, u32) staged in LDS (wave-private slice, K1_RMAX records; parts
//                 with more records restage). Only the slow path reads it, through a 64-record window (lane i <->
//                 record wbase+i) that moves forward with the lookups.
//   checkpoint    (record, chunk, P(chunk start) - P(record start)) of the record that was open at the end of the last
//                 slow chunk; the streaming path keeps no running prefix at all.
constexpr int K1_RMAX = 320;   // record starts staged per wave (8192-op parts of ~33-op reads hold ~250)
constexpr int K1_WAVES = 4;

// GATE (CLR, C:53-70, 422-433): the extractor computes ins_pct / var_dist of EVERY haplotype-tagged record first — a record without
// M ops divides by zero there whatever its mapq — and walks the CIGAR (signatures, reference_end and SEQ asserts) only where
// ins_pct <= 0.13 or var_dist >= 200. Instead of a second pass over the CIGARs in front of the scan (clr_gate_records), the scan
// carries the gate: every chunk leaves four ballots (which lanes hold an M op in op slot 0..3: the compares write them straight
// into SGPRs) and a running count in LDS; at the end of the part one lane per record counts the M ops between its record's
// start and end (eight mask reads and popcounts) —
// a tagged record without M ops raises ZeroDivisionError; and the few records that emit (or carry a forbidden op, or a
// SEQ-length mismatch) get their exact sums on demand, by the wave, from the L2-hot chunks. A part of more than K1G_CH chunks
// (a record of thousands of ops in a read-shaped input), or one that holds an EMPTY M / I op (the only way to a zero ins_pct
// denominator with M ops present), raises ERRB_CLR_FALLBACK: the run is repeated with the separate gate pass.
constexpr int K1G_CH = 96;      // chunks (24576 ops) of a part whose M-op masks are kept in LDS (32 bytes each)

template <int CLS, int DEPTH, bool GATE = false>
__global__ __launch_bounds__(256) void cigar_scan_emit(RecView rv, const uint32_t* __restrict__ rb, int n_parts,
                                                        int min_svlen, int min_mapq, EmitCtx ec,
                                                        uint32_t* __restrict__ part_count, int ablate) {
  using T = OpTab<CLS>;
  static_assert(!GATE || CLS == 0, "the gate belongs to the contig op table");
  __shared__ uint32_t sh_off[K1_WAVES][K1_RMAX + 1];
  __shared__ uint4 sh_rows[K1_WAVES][2 * K1E_STAGE];            // rows of the part, staged (wave-private) until they leave as one batch
  __shared__ uint64_t sh_gm[GATE ? K1_WAVES : 1][GATE ? K1G_CH : 1][4];    // per chunk: which lanes hold an M op in op slot k (lane l: ops 4l..4l+3)
  __shared__ uint32_t sh_cp[GATE ? K1_WAVES : 1][GATE ? K1G_CH + 1 : 1];   // per chunk: M ops of the part in front of it
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int part = blockIdx.x * K1_WAVES + wv;
  if (part >= n_parts) return;
  const uint32_t r0 = __builtin_amdgcn_readfirstlane(rb[part]), r1 = __builtin_amdgcn_readfirstlane(rb[part + 1]);
  uint32_t ord = 0;  // signatures emitted by this part so far (wave-uniform)
  if (r0 >= r1) { if (lane == 0) { part_count[part] = 0; ec.prec[part] = make_uint2(0u, 0u); } return; }
  const uint64_t o0 = rv.cigar_off[r0], o1 = rv.cigar_off[r1];
  const uint64_t cb0 = o0 & ~3ull;  // absolute op index of rel 0 (16-byte aligned)
  if (o1 <= o0 || o1 - cb0 >= 0x3FFFFF00ull) {  // relative BYTE offsets must fit 32 bits
    if (lane == 0) { atomicOr(&ec.ctr->err, o1 <= o0 ? ERRB_EMPTY_CIGAR : ERRB_RANGE); part_count[part] = 0; ec.prec[part] = make_uint2(0u, 0u); }
    return;
  }
  const uint32_t ob_rel = (uint32_t)(o0 - cb0), oe_rel = (uint32_t)(o1 - cb0);

  // Chunk loads go through a raw buffer descriptor whose range is exactly this part: [cb0, o1). The hardware
  // range check returns 0 for every dword at or beyond the end, so the prefetch that runs past the part needs no
  // branch, touches no memory and yields op 0 / len 0 (advances nothing).
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const uint32_t* part_base = rv.cigar + cb0;
  const uint32_t base_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)part_base);
  const uint32_t base_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)part_base >> 32));
  const uint32_t part_bytes = __builtin_amdgcn_readfirstlane(oe_rel * 4u);
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(((uintptr_t)base_hi << 32) | (uintptr_t)base_lo), (short)0, (int)part_bytes, 0x00020000);
  auto load_chunk = [&](uint32_t cb) -> uint4 {
    const uint32_t voff = (cb + 4u * (uint32_t)lane) * 4u;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
  };
  uint4 wa = load_chunk(0), wb = load_chunk(256), wc = load_chunk(512), wd = make_uint4(0, 0, 0, 0);
  if (DEPTH == 4) wd = load_chunk(768);

  // Only the low dwords of cigar_off are read: relative offsets are < 2^30, so (lo - cb0_lo) mod 2^32 is exact.
  const uint32_t* __restrict__ off_lo = reinterpret_cast<const uint32_t*>(rv.cigar_off);
  const uint32_t cb0_lo = (uint32_t)cb0;
  uint32_t* my_off = sh_off[wv];
  uint32_t tbase = r0;      // first record of the staged table
  uint32_t n_tab = 0;       // records in the table
  bool bad = false;
  auto stage = [&](uint32_t first) {
    tbase = first;
    n_tab = min((uint32_t)K1_RMAX, r1 - first);
    bool mono = true;
    for (uint32_t i = lane; i <= n_tab; i += 64) {
      const uint32_t v = off_lo[2 * (size_t)(first + i)] - cb0_lo;
      my_off[i] = v;
      if (i < n_tab) {
        const uint32_t nx = off_lo[2 * (size_t)(first + i) + 2] - cb0_lo;
        mono = mono && (int32_t)(nx - v) > 0;   // empty CIGAR => reference IndexError at H:63
        if (!GATE && (CLS == 0 || CLS == 1)) {  // a walked record whose SEQ length differs from its CIGAR's: H:397-398, RS:123-124
          const uint32_t fl = rv.flag[first + i];
          if ((fl & VSV_F_SEQ_MISMATCH) && rv.mapq[first + i] >= (uint32_t)min_mapq && (CLS == 1 || (fl & (VSV_F_HP1 | VSV_F_HP2))))
            atomicOr(&ec.ctr->err, ERRB_SEQLEN);
        }                                       // (GATE: whether such a record is walked depends on its gate: end of the part)
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (__ballot(!mono)) { if (lane == 0) atomicOr(&ec.ctr->err, ERRB_EMPTY_CIGAR); bad = true; }
  };
  stage(r0);
  // ---- CLR gate state (GATE only) ----
  uint64_t (*my_gm)[4] = sh_gm[GATE ? wv : 0];
  uint32_t* my_cp = sh_cp[GATE ? wv : 0];
  uint32_t gate_run_g = 0;
  bool gate_over = false;
  if (GATE && lane == 0) my_cp[0] = 0;
  // exact gate of the record whose ops are [s_r, s_e) (relative): all 64 lanes, the record's chunks come from L2
  auto gate_of = [&](uint32_t s_r, uint32_t s_e) -> bool {
    int64_t m = 0, nm = 0, ins = 0;
    for (uint32_t c = s_r & ~255u; c < s_e; c += 256u) {
      const uint4 v = load_chunk(c);
      const uint32_t ww[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t x = c + 4u * (uint32_t)lane + (uint32_t)k;
        if (x >= s_r && x < s_e) {
          const uint32_t op = ww[k] & 15u, len = ww[k] >> 4;
          if (op == 0u) { m += len; ++nm; } else if (op == 1u) ins += len;
        }
      }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { m += __shfl_xor(m, d, 64); nm += __shfl_xor(nm, d, 64); ins += __shfl_xor(ins, d, 64); }
    // (every lane holds the same sums after the butterfly; readfirstlane tells the compiler that the verdict is wave-uniform)
    const bool zero = m + ins == 0 || nm == 0;                                                            // C:61, C:70
    const bool pass = !zero && ((100 * ins <= 13 * (m + ins)) || (m >= 200 * nm));                        // C:427 in exact integers
    if (__builtin_amdgcn_readfirstlane(zero ? 1 : 0)) { if (lane == 0) atomicOr(&ec.ctr->err, ERRB_ZERODIV); }
    return __builtin_amdgcn_readfirstlane(pass ? 1 : 0) != 0;
  };

  // ---- record lookup (slow path only): 64-record window of the LDS table, one start per lane -------------
  uint32_t wbase = 0;                  // window = table entries [wbase, wbase+64)
  uint32_t nwin = 0, nxt = 0, s_rel = 0;
  auto load_window = [&]() {
    nwin = min(64u, n_tab - wbase);
    s_rel = my_off[wbase + min((uint32_t)lane, nwin - 1u)];
    nxt = __builtin_amdgcn_readfirstlane(my_off[wbase + nwin]);   // first start after the window (oe_rel at the end)
  };
  load_window();
  // Lookups ascend within a part, so the window only moves forward. Returns the number of window records whose start
  // is <= xo (>= 1): op xo belongs to record tbase + wbase + cnt - 1.
  auto lookup = [&](uint32_t xo) -> uint32_t {
    while (xo >= nxt && !bad) {
      wbase += 64;
      if (wbase >= n_tab) { stage(tbase + n_tab); wbase = 0; }   // part with > K1_RMAX records: restage
      load_window();
    }
    return (uint32_t)__popcll(__ballot((uint32_t)lane < nwin && s_rel <= xo));
  };

  // Checkpoint of the lazily evaluated prefix: for record ck_rec, (ck_r, ck_q) = P(start of chunk ck_chunk) - P(record
  // start). The streaming path keeps NO running prefix; a signature whose record started in an earlier chunk walks the
  // (L2-hot) chunks from the checkpoint (or from the record's start chunk) up to the current one.
  uint32_t ck_rec = 0xFFFFFFFFu, ck_chunk = 0, ck_r = 0, ck_q = 0;

  // rows collect in LDS and leave as ONE batch at the end of the part (a part of a read-shaped input emits a row or two): one slot
  // allocation in the part's shard of the pool, coalesced stores, and the part files where its batch lies for the placement kernel
  // (k1l_place). A part with more rows than the stage holds files the batches in front of its last one in the overflow list.
  const uint32_t shard = (uint32_t)part % K1_SHARDS;
  uint4* my_rows = sh_rows[wv];
  uint32_t n_staged = 0;
  auto flush = [&](bool last) {
    uint32_t loc = 0, n = 0;
    if (n_staged != 0) {
      uint32_t b = 0;
      if (lane == 0) b = atomicAdd(&ec.shard_cnt[shard * 16], n_staged);
      const uint32_t slot = __builtin_amdgcn_readfirstlane(b);
      if (slot + n_staged <= ec.shard_cap) {                    // (else: the placement pass reports the shard's use, the rows of this batch stay unwritten)
        loc = shard * ec.shard_cap + slot; n = n_staged;
        uint4* dst = reinterpret_cast<uint4*>(ec.pool + loc);
        for (uint32_t i = lane; i < 2u * n_staged; i += 64) dst[i] = my_rows[i];
      }
      if (!last && n != 0 && lane == 0) {
        const uint32_t e = atomicAdd(&ec.shard_cnt[8], 1u);
        if (e < ec.batch_cap) ec.batches[e] = make_uint4((uint32_t)part, ord - n_staged, loc, n);
        else atomicOr(&ec.ctr->err, ERRB_CAPACITY);
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (last && lane == 0) ec.prec[part] = make_uint2(loc, n);
    n_staged = 0;
  };

  // decoded chunk: per-op (ref, query) advances, their wave-exclusive prefix per lane and the chunk totals
  struct Dec { uint32_t ar[4], aq[4], excl_r, excl_q, tot_r, tot_q; };
  auto decode = [&](const uint32_t (&w)[4], Dec& d) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t op = w[k] & 15u, len = w[k] >> 4;
      d.ar[k] = len & bfe_mask(T::REF, op);
      d.aq[k] = len & bfe_mask(T::QRY, op);
    }
    const uint32_t sum_r = d.ar[0] + d.ar[1] + d.ar[2] + d.ar[3];
    const uint32_t sum_q = d.aq[0] + d.aq[1] + d.aq[2] + d.aq[3];
    const uint32_t incl_r = wave_incl_scan(sum_r), incl_q = wave_incl_scan(sum_q);
    d.excl_r = incl_r - sum_r; d.excl_q = incl_q - sum_q;
    d.tot_r = rdlane(incl_r, 63); d.tot_q = rdlane(incl_q, 63);
  };
  // in-chunk exclusive prefix at rel op index s (wave-uniform, inside chunk cb)
  auto prefix_at = [&](const Dec& d, uint32_t cb, uint32_t s, uint32_t& b_r, uint32_t& b_q) {
    const uint32_t ls = (s - cb) >> 2, ss = (s - cb) & 3u;
    b_r = rdlane(d.excl_r, ls); b_q = rdlane(d.excl_q, ls);
    if (ss > 0) { b_r += rdlane(d.ar[0], ls); b_q += rdlane(d.aq[0], ls); }
    if (ss > 1) { b_r += rdlane(d.ar[1], ls); b_q += rdlane(d.aq[1], ls); }
    if (ss > 2) { b_r += rdlane(d.ar[2], ls); b_q += rdlane(d.aq[2], ls); }
  };
  // make (ck_r, ck_q) = P(cb) - P(s_r) for record rec, which started at s_r < cb
  auto base_for = [&](uint32_t rec, uint32_t s_r, uint32_t cb) {
    if (ck_rec != rec) {
      const uint32_t c_s = s_r & ~255u;
      const uint4 v = load_chunk(c_s);
      const uint32_t ww[4] = {v.x, v.y, v.z, v.w};
      Dec e; decode(ww, e);
      uint32_t b_r, b_q;
      prefix_at(e, c_s, s_r, b_r, b_q);
      ck_rec = rec; ck_chunk = c_s + 256u; ck_r = e.tot_r - b_r; ck_q = e.tot_q - b_q;
    }
    while (ck_chunk < cb) {
      const uint4 v = load_chunk(ck_chunk);
      const uint32_t ww[4] = {v.x, v.y, v.z, v.w};
      Dec e; decode(ww, e);
      ck_r += e.tot_r; ck_q += e.tot_q; ck_chunk += 256u;
    }
  };

  const uint32_t thr = (uint32_t)min_svlen;
  const uint32_t thr16 = thr >= (1u << 28) ? 0xFFFFFFFFu : thr << 4;
  constexpr uint32_t EMIT_R = rev32(EMIT_MASK | (EMIT_MASK << 16));   // bit-reversed, duplicated: sign(EMIT_R << (w & 31)) = bit op
  constexpr uint32_t BAD_R = rev32(T::BAD | (T::BAD << 16));
  uint32_t hd_rec = 0xFFFFFFFFu, hd_fl = 0, hd_mq = 0, hd_tid = 0, hd_first = 0, hd_pos = 0;   // cached record header
  bool hd_gate = true;                                                                          // ... and its CLR gate (GATE)

  // ---- slow path: the chunk holds at least one candidate op (wave-uniform) ------------------------------
  auto slow = [&](const uint4& wcur, const uint32_t cb) {
    const uint32_t x = cb + 4u * (uint32_t)lane;  // rel index of this lane's first op
    uint32_t w[4] = {wcur.x, wcur.y, wcur.z, wcur.w};
    if (cb == 0) {                                // ops before ob_rel belong to the previous part
#pragma unroll
      for (int k = 0; k < 4; ++k) if (x + k < ob_rel) w[k] = 15u;
    }
    Dec d; decode(w, d);
    uint32_t em = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {   // exact emit predicate: len >= min_svlen <=> packed word >= min_svlen << 4
      const bool e = ((int32_t)(EMIT_R << (w[k] & 31u)) < 0 && w[k] >= thr16) ||
                     (T::BAD != 0 && (int32_t)(BAD_R << (w[k] & 31u)) < 0 && w[k] >= 16u);
      em |= (e ? 1u : 0u) << k;
    }
    // per-lane exclusive prefixes in front of each of the lane's four ops
    uint32_t pr[4], pq[4];
    pr[0] = d.excl_r; pq[0] = d.excl_q;
#pragma unroll
    for (int k = 1; k < 4; ++k) { pr[k] = pr[k - 1] + d.ar[k - 1]; pq[k] = pq[k - 1] + d.aq[k - 1]; }
    // one scalar iteration per candidate op, in (lane, sub) = op order
    uint64_t anym = __ballot(em != 0);
    while (anym) {
      const uint32_t l = (uint32_t)__builtin_ctzll(anym);
      anym &= anym - 1;
      uint32_t eml = rdlane(em, l);
      while (eml) {
        const uint32_t sub = (uint32_t)__builtin_ctz(eml);
        eml &= eml - 1;
        const uint32_t wl = rdlane(sub == 0 ? w[0] : sub == 1 ? w[1] : sub == 2 ? w[2] : w[3], l);
        const uint32_t px_r = rdlane(sub == 0 ? pr[0] : sub == 1 ? pr[1] : sub == 2 ? pr[2] : pr[3], l);
        const uint32_t px_q = rdlane(sub == 0 ? pq[0] : sub == 1 ? pq[1] : sub == 2 ? pq[2] : pq[3], l);
        const uint32_t xo = cb + 4u * l + sub;
        const uint32_t op = wl & 15u, len = wl >> 4;
        const uint32_t cnt = lookup(xo);
        if (bad) return;
        const uint32_t rec = tbase + wbase + cnt - 1u;
        const uint32_t s_r = rdlane(s_rel, cnt - 1u);
        uint32_t a_r, a_q;
        if (s_r >= cb) {                 // record starts inside this chunk
          uint32_t b_r, b_q;
          prefix_at(d, cb, s_r, b_r, b_q);
          a_r = px_r - b_r;
          a_q = px_q - b_q;
        } else {                         // record was open at the chunk start
          base_for(rec, s_r, cb);
          a_r = ck_r + px_r;
          a_q = ck_q + px_q;
        }
        if (rec != hd_rec) {             // header of the record (scalar cache); long records emit many times
          sload5(align4(rv.flag + rec), align4(rv.mapq + rec), rv.tid + rec, rv.cigar + cb0 + s_r, rv.pos + rec, hd_fl, hd_mq, hd_tid, hd_first, hd_pos);
          hd_rec = rec;
          if (GATE) {                    // C:425-427: the gate of a tagged record with enough mapq, once per record
            const uint32_t gfl = byte_of(hd_fl, rv.flag + rec), gmq = byte_of(hd_mq, rv.mapq + rec);
            hd_gate = true;
            if ((gfl & (VSV_F_HP1 | VSV_F_HP2)) && gmq >= (uint32_t)min_mapq)
              hd_gate = gate_of(s_r, cnt < nwin ? rdlane(s_rel, cnt) : nxt);
          }
        }
        const uint32_t fl = byte_of(hd_fl, rv.flag + rec), mq = byte_of(hd_mq, rv.mapq + rec);
        uint32_t hapbits;
        if (CLS == 0) hapbits = (mq >= (uint32_t)min_mapq && (!GATE || hd_gate)) ? ((fl >> 2) & 3u) : 0u;   // H:392-394 (C:427)
        else if (CLS == 1) hapbits = (mq >= (uint32_t)min_mapq) ? 1u : 0u;                        // RS:120
        else if (CLS == 3) hapbits = (!(fl & VSV_F_SKIP) && mq >= (uint32_t)min_mapq) ? 1u : 0u; // SE:439, 446
        else hapbits = (!(fl & (VSV_F_UNMAPPED | VSV_F_SECONDARY)) && mq >= (uint32_t)min_mapq) ? 1u : 0u;
        if (!hapbits) continue;
        if (op != 1u && op != 2u) {      // N/=/X on the contig table: assert offset_ref==reference_end (H:396)
          if (lane == 0) atomicOr(&ec.ctr->err, ERRB_REFEND);
          continue;
        }
        const uint32_t hc = (T::HC && (hd_first & 15u) == 5u) ? (hd_first >> 4) : 0u;             // H:63-65
        const uint32_t nemit = (hapbits == 3u) ? 2u : 1u;
        if (n_staged + nemit > (uint32_t)K1E_STAGE) flush(false);
        if (lane < (int)nemit) {
          const uint32_t qs = a_q + hc;
          const uint32_t hp2 = (CLS == 0) ? ((hapbits == 3u) ? (uint32_t)lane : (hapbits >> 1)) : 0u;
          my_rows[2 * (n_staged + lane)] = make_uint4(hd_pos + a_r, len, qs, (CLS == 1 || CLS == 3) ? 0u : qs + (op == 2u ? 1u : len));
          my_rows[2 * (n_staged + lane) + 1] = make_uint4(rec, 0xFFFFFFFFu, (op == 2u ? (uint32_t)VSV_M_DEL : 0u) | (hp2 ? (uint32_t)VSV_M_HP2 : 0u), hd_tid);
        }
        __builtin_amdgcn_wave_barrier();
        n_staged += nemit;
        ord += nemit;
      }
    }
    // checkpoint for the record that is open at the end of this chunk (dense chunks then never walk)
    const uint32_t last = min(cb + 256u, oe_rel) - 1u;
    const uint32_t cnt = lookup(last);
    if (bad) return;
    const uint32_t rec = tbase + wbase + cnt - 1u;
    const uint32_t s_r = rdlane(s_rel, cnt - 1u);
    if (s_r >= cb) {
      uint32_t b_r, b_q;
      prefix_at(d, cb, s_r, b_r, b_q);
      ck_rec = rec; ck_chunk = cb + 256u; ck_r = d.tot_r - b_r; ck_q = d.tot_q - b_q;
    } else if (ck_rec == rec && ck_chunk == cb) {
      ck_r += d.tot_r; ck_q += d.tot_q; ck_chunk = cb + 256u;
    }
  };

  // ---- streaming path: per op one shift against the (bit-reversed, duplicated) op mask, a sign test and a compare
  // of the packed word against min_svlen<<4. The test is a superset of the emit predicate; `slow` is exact.
  auto process_chunk = [&](const uint4& wcur, const uint32_t cb) {
    if (cb >= oe_rel || bad) return;             // ring slots past the end of the part
    if (GATE) {
      const uint32_t ci = cb >> 8;
      if (ci < (uint32_t)K1G_CH) {
        const uint32_t ww[4] = {wcur.x, wcur.y, wcur.z, wcur.w};
        uint64_t mk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) mk[k] = __ballot((ww[k] & 15u) == 0u);            // M ops (var_dist's list, C:66-68): one mask per op slot
        // ins_pct's denominator (C:57-61) is zero although the record has M ops only if every M and I op of it is EMPTY: such ops
        // (packed words 0 and 1) do not occur in real alignments; a part that holds one takes the separate gate pass instead
        uint32_t lo4 = min(min(ww[0], ww[1]), min(ww[2], ww[3]));
        if (cb + 256u > oe_rel || cb < ob_rel) {                                       // ops outside the part read as 0 / belong to the neighbour
          lo4 = 0xFFFFFFFFu;
#pragma unroll
          for (int k = 0; k < 4; ++k) { const uint32_t x = cb + 4u * (uint32_t)lane + (uint32_t)k; if (x >= ob_rel && x < oe_rel) lo4 = min(lo4, ww[k]); }
        }
        if (__ballot(lo4 < 2u)) gate_over = true;
        gate_run_g += (uint32_t)(__popcll(mk[0]) + __popcll(mk[1]) + __popcll(mk[2]) + __popcll(mk[3]));
        if (lane == 0) {
#pragma unroll
          for (int k = 0; k < 4; ++k) my_gm[ci][k] = mk[k];
          my_cp[ci + 1] = gate_run_g;
        }
      } else gate_over = true;
    }
    bool cand = ((int32_t)(EMIT_R << (wcur.x & 31u)) < 0 && wcur.x >= thr16) |
                ((int32_t)(EMIT_R << (wcur.y & 31u)) < 0 && wcur.y >= thr16) |
                ((int32_t)(EMIT_R << (wcur.z & 31u)) < 0 && wcur.z >= thr16) |
                ((int32_t)(EMIT_R << (wcur.w & 31u)) < 0 && wcur.w >= thr16);
    if (T::BAD != 0) {
      const uint32_t t = (BAD_R << (wcur.x & 31u)) | (BAD_R << (wcur.y & 31u)) | (BAD_R << (wcur.z & 31u)) | (BAD_R << (wcur.w & 31u));
      cand |= (int32_t)t < 0;
    }
    if (__ballot(cand) == 0ull || (ablate & 1)) return;
    slow(wcur, cb);
  };

  // three chunks in flight per wave; the ring is unrolled so no in-flight register is ever copied
  if (DEPTH == 4) {
    for (uint32_t cb = 0; cb < oe_rel && !bad; cb += 1024) {
      process_chunk(wa, cb);
      wa = load_chunk(cb + 1024);
      process_chunk(wb, cb + 256);
      wb = load_chunk(cb + 1280);
      process_chunk(wc, cb + 512);
      wc = load_chunk(cb + 1536);
      process_chunk(wd, cb + 768);
      wd = load_chunk(cb + 1792);
    }
  } else {
    for (uint32_t cb = 0; cb < oe_rel && !bad; cb += 768) {
      process_chunk(wa, cb);
      wa = load_chunk(cb + 768);
      process_chunk(wb, cb + 256);
      wb = load_chunk(cb + 1024);
      process_chunk(wc, cb + 512);
      wc = load_chunk(cb + 1280);
    }
  }
  // every record start of the part goes through stage() once: an empty CIGAR must raise (H:63 IndexError)
  while (!bad && tbase + n_tab < r1) stage(tbase + n_tab);
  if (GATE && !bad) {
    if (gate_over) { if (lane == 0) atomicOr(&ec.ctr->err, ERRB_CLR_FALLBACK); }
    else {
      __builtin_amdgcn_wave_barrier();
      const uint32_t n_ch = (oe_rel + 255u) >> 8;
      auto m_ops_before = [&](uint32_t x) -> uint32_t {                 // M ops of the part in front of op x
        const uint32_t ci = x >> 8;
        if (ci >= n_ch) return my_cp[n_ch];
        const uint32_t l = (x & 255u) >> 2, kk = x & 3u;
        uint32_t c = my_cp[ci];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
          const uint32_t nl = l + (k < kk ? 1u : 0u);                   // lanes of slot k in front of x
          c += (uint32_t)__popcll(my_gm[ci][k] & (nl >= 64u ? ~0ull : (1ull << nl) - 1ull));
        }
        return c;
      };
      for (uint32_t first = r0; first < r1; first += (uint32_t)K1_RMAX) {
        const uint32_t nrec = min((uint32_t)K1_RMAX, r1 - first);
        for (uint32_t i0 = 0; i0 < nrec; i0 += 64) {                     // whole waves: the on-demand gates below are wave-cooperative
          const uint32_t i = i0 + (uint32_t)lane;
          const bool live = i < nrec;
          uint32_t s_r = 0, s_e = 0, fl = 0, mq = 0;
          if (live) {
            s_r = off_lo[2 * (size_t)(first + i)] - cb0_lo;
            s_e = off_lo[2 * (size_t)(first + i) + 2] - cb0_lo;
            fl = rv.flag[first + i]; mq = rv.mapq[first + i];
          }
          const bool tagged = live && (fl & (VSV_F_HP1 | VSV_F_HP2));
          // no M op at all: var_dist (and ins_pct, unless the record has I ops: then var_dist) divides by zero — C:61 / C:70,
          // whatever the record's mapq
          if (tagged && m_ops_before(s_e) == m_ops_before(s_r)) atomicOr(&ec.ctr->err, ERRB_ZERODIV);
          // a tagged record with enough mapq and a SEQ of another length: asserted only if its gate lets it be walked (C:427-431)
          uint64_t need = __ballot(tagged && (fl & VSV_F_SEQ_MISMATCH) && mq >= (uint32_t)min_mapq);
          while (need) {
            const int src = __builtin_ctzll(need);
            need &= need - 1;
            if (gate_of(rdlane(s_r, (uint32_t)src), rdlane(s_e, (uint32_t)src)) && lane == 0) atomicOr(&ec.ctr->err, ERRB_SEQLEN);
          }
        }
      }
    }
  }
  flush(true);
  if (lane == 0) part_count[part] = ord;
}

// ---- K1L: the scan for LONG records (contig alignments: 10^3-10^6 ops per CIGAR) ---------------------------------
// cigar_scan_emit above is built for read-shaped input: parts are cut at record boundaries (a record is never shared by two
// waves) and the streaming path keeps no prefix, because a 256-op chunk holds several records. With Mb contigs both choices turn
// around: a record-aligned part is one whole CIGAR per wave (a 10^6-op alignment = 4 MB streamed by ONE wave while the chip
// idles), and a chunk almost always lies inside one record, so the lazily walked prefix re-reads and re-decodes what was just
// streamed and every signature pays the scalar lookup / checkpoint bookkeeping (0.35 of the roofline on the dense contig shape).
// Here instead
//  * parts are fixed 8192-op slices of the flat op stream, whatever the records do: perfect balance, a record may span parts;
//  * every chunk is decoded (two bit-field extracts against duplicated 32-bit op masks and two ANDs per op) and ONE pair of DPP
//    prefix sums gives both the chunk totals (lane 63) and, when the chunk holds candidates, every op's prefix: the wave carries
//    (run_r, run_q) = advance since the current record's start (or since the part's start while the record that was open there
//    continues);
//  * a chunk without a record start lies in one record: all its candidate ops build their rows at once (ballot ordinals,
//    ONE slot allocation, vector stores), with the record's header cached in SGPRs; chunks that hold record starts are cut into
//    segments at the starts and run the same emission per segment;
//  * rows of the record that was open at the part's start are relative to the part's start and flagged; every part
//    publishes (has_start, advance since its last start); a segmented scan over the parts (k1l_scan_*) turns that into the
//    carry each part's head rows still miss, and place_raw adds it while it puts the rows in order. No spinning, no inter-wave
//    communication inside the kernel, one extra 12-byte record per 32 KiB of CIGAR.
constexpr int K1L_PART = 8192;            // ops per part (32 KiB), a multiple of the 256-op chunk
struct PartAgg { uint32_t has_start, run_r, run_q; };

constexpr uint32_t dup16(uint32_t t) { return t | (t << 16); }

// ---- decoupled look-back: an exclusive scan inside one launch ------------------------------------------------------------------
// A part's rows belong at raw[rows of the parts in front + ordinal], and the rows of the record that was open at its start still miss
// that record's advance in the parts in front (the carry). Both are exclusive prefixes over the parts: x[p] = (has_start, run_r,
// run_q, count) with (f1, v1) (+) (f2, v2) = (f1 | f2, f2 ? v2 : v1 + v2) for the runs and a plain sum for the counts. The placement
// kernel (k1l_place) computes them for blocks of parts: every block publishes its aggregate, looks back over its predecessors' words
// until it meets one that already holds an inclusive prefix, and publishes its own inclusive prefix.
// Three 64-bit words per block, each valid by itself: [63:40] epoch of the run (the host counts runs; words of earlier runs are
// simply "not yet"), [39:38] kind (aggregate / inclusive prefix), [32] has_start (word 0), [31:0] value. They are written and read
// with relaxed agent-scope atomics and nothing else depends on their order, so no fence is needed (a fence writes back and
// invalidates L2 for every engine on the chip); a reader that meets words of different kinds (the writer is between its stores) reads
// again. Workgroups are dispatched in index order and a block waits for lower blocks only, so the lowest unfinished block is always
// resident or next to be dispatched: the waits end. They are bounded all the same (ERRB_LOOKBACK instead of a hung GPU).
constexpr uint64_t LB_AGG = 1ull, LB_PRE = 2ull;
struct LbItem { uint32_t f, r, q, cnt; };
__device__ __forceinline__ LbItem lb_combine(const LbItem& a, const LbItem& b) {          // a: the earlier parts, b: the later ones
  LbItem c;
  c.f = a.f | b.f; c.r = b.f ? b.r : a.r + b.r; c.q = b.f ? b.q : a.q + b.q; c.cnt = a.cnt + b.cnt;
  return c;
}
__device__ __forceinline__ uint64_t lb_pack(uint32_t epoch, uint64_t kind, uint32_t f, uint32_t v) {
  return ((uint64_t)epoch << 40) | (kind << 38) | ((uint64_t)(f & 1u) << 32) | (uint64_t)v;
}
__device__ __forceinline__ uint32_t lb_kind(uint64_t w, uint32_t epoch) { return (uint32_t)(w >> 40) == epoch ? (uint32_t)(w >> 38) & 3u : 0u; }
__device__ __forceinline__ uint64_t lb_ld(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lb_st(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) { return rdlane(wave_incl_scan(v), 63); }
constexpr uint32_t LB_SPIN_LIMIT = 1u << 20;      // polls of ~1 us each

// Called by all 64 lanes of one wave of the block that owns item `part` (wave-uniform arguments). lb_publish: the item's aggregate.
// lb_resolve: the exclusive prefix over the items in front of it (waits until all of them have published), and the item's
// inclusive prefix for its successors.
__device__ __forceinline__ void lb_publish(uint64_t* __restrict__ lbw, uint32_t epoch, uint32_t part, const LbItem& own, uint64_t kind, int lane) {
  uint64_t* w = lbw + 3 * (size_t)part;            // (three stores by three lanes: a select over the lane becomes an indexed read of a stack copy of `own`)
  if (lane == 0) lb_st(w, lb_pack(epoch, kind, own.f, own.cnt));
  if (lane == 1) lb_st(w + 1, lb_pack(epoch, kind, 0u, own.r));
  if (lane == 2) lb_st(w + 2, lb_pack(epoch, kind, 0u, own.q));
}
__device__ __forceinline__ LbItem lb_resolve(uint64_t* __restrict__ lbw, uint32_t epoch, uint32_t part, const LbItem& own, int lane, uint32_t* err) {
  LbItem acc{0, 0, 0, 0};
  if (part != 0) {
    int64_t j = (int64_t)part - 1;                 // the nearest part not yet accounted for
    uint32_t spins = 0;
    for (;;) {
      const int64_t pj = j - lane;
      uint64_t w0 = 0, w1 = 0, w2 = 0;
      uint32_t state = 3u;                         // in front of part 0: an empty inclusive prefix
      if (pj >= 0) {
        const uint64_t* q = lbw + 3 * (size_t)pj;
        w0 = lb_ld(q); w1 = lb_ld(q + 1); w2 = lb_ld(q + 2);
        const uint32_t k0 = lb_kind(w0, epoch), k1 = lb_kind(w1, epoch), k2 = lb_kind(w2, epoch);
        state = (k0 == k1 && k1 == k2) ? k0 : 0u;
      }
      const uint64_t m_inv = __ballot(state == 0u), m_pre = __ballot(state >= 2u);
      const uint32_t lp = m_pre ? (uint32_t)__builtin_ctzll(m_pre) : 64u, li = m_inv ? (uint32_t)__builtin_ctzll(m_inv) : 64u;
      if (li < lp) {                               // a part between this one and the nearest prefix has not published yet
        if (++spins > LB_SPIN_LIMIT) { if (lane == 0) atomicOr(err, ERRB_LOOKBACK); break; }
        __builtin_amdgcn_s_sleep(16);
        continue;
      }
      const bool take = (uint32_t)lane <= lp;      // lanes 0 .. lp: aggregates, closed by the prefix of lane lp (if any)
      const uint32_t f = take ? (uint32_t)(w0 >> 32) & 1u : 0u, cnt = take ? (uint32_t)w0 : 0u;
      const uint64_t mf = __ballot(f != 0u);
      const uint32_t lf = mf ? (uint32_t)__builtin_ctzll(mf) : 64u;          // the nearest part that holds a record start
      const bool in_run = take && (uint32_t)lane <= lf;
      LbItem t;
      t.f = mf ? 1u : 0u; t.cnt = wave_sum(cnt);
      t.r = wave_sum(in_run ? (uint32_t)w1 : 0u); t.q = wave_sum(in_run ? (uint32_t)w2 : 0u);
      acc = lb_combine(t, acc);
      if (lp < 64u) break;
      j -= 64;
    }
  }
  lb_publish(lbw, epoch, part, lb_combine(acc, own), LB_PRE, lane);
  return acc;
}
// ---- K1L: the scan for LONG records, second form ---------------------------------------------------------------------------------
// Same streaming blocks as cigar_scan_long_pool above; what changed is everything behind the candidate test:
//  * a candidate leaves a 16-byte DESCRIPTOR in the wave's LDS stage — {pos, q_start (both still without the part's carry where the
//    record was open at the part's start), len << 4 | flags, record} — instead of a finished 32-byte row and its 8-byte pool key;
//  * the stage leaves as ONE batch per part (one slot allocation, coalesced 16-byte stores), and the part files where it lies;
//  * k1l_place then scans the part counts and carries by a look-back among its own (short, uniform) blocks and builds the rows where
//    they belong, coalesced: one launch for round 3's three scan launches + place_raw.
// The scan itself never waits. (The rows straight from the scan, by a look-back over the parts INSIDE it, were built and measured
// first: bit-exact, but a part waits a median 22-32 us for the slowest of the ~6000 lower parts in flight to end its stream —
// per-part stream times spread p50 63 / p99 139 us at 16384 ops, evenly over XCDs and CUs — and this kernel needs every wave slot
// streaming to saturate HBM (4 KiB in flight per wave, ~4 us of loaded latency): 4.66 ms at 8192-op parts, 2.59 ms at 65536;
// resolving a part behind the stream of the wave's next one, parts handed out by a counter, 3.0 ms; the pool form 2.58.)
constexpr int K1L_DSTAGE = 192;
constexpr uint32_t KD_DEL = 1u, KD_CARRY = 2u, KD_HP2 = 4u;
constexpr int K1L_RMAX = 64;               // record starts staged per wave (a part of a contig-shaped input holds a handful; more restage)
__device__ __forceinline__ void k1l_row(const uint4& d, uint32_t cr, uint32_t cq, uint32_t tid, bool no_qend, uint4& lo4, uint4& hi4) {
  const uint32_t fl = d.z & 15u, len = d.z >> 4;
  const uint32_t pos = d.x + ((fl & KD_CARRY) ? cr : 0u), qs = d.y + ((fl & KD_CARRY) ? cq : 0u);
  lo4 = make_uint4(pos, len, qs, no_qend ? 0u : qs + ((fl & KD_DEL) ? 1u : len));
  hi4 = make_uint4(d.w, 0xFFFFFFFFu, ((fl & KD_DEL) ? (uint32_t)VSV_M_DEL : 0u) | ((fl & KD_HP2) ? (uint32_t)VSV_M_HP2 : 0u), tid);
}
struct K1LArgs {
  RecView rv; const uint32_t* rb; Counters* ctr;
  uint4* dpool; uint32_t* shard_cnt; uint32_t shard_cap;      // descriptor pool: K1_SHARDS shards of shard_cap descriptors, cursor s at shard_cnt[16 s]
  uint4* batches; uint32_t batch_cap;                         // overflow list {part, first ordinal, location, descriptors}, cursor at shard_cnt[8]
  uint2* prec; uint32_t* part_count; PartAgg* agg;            // per part: {location, descriptors} of its last batch, rows, (has_start, advance since its last start)
  int n_parts, min_svlen, min_mapq; uint32_t part_ops; int ablate;
};
template <int CLS>
__global__ __launch_bounds__(256) void cigar_scan_long(K1LArgs A) {
  using T = OpTab<CLS>;
  __shared__ uint32_t sh_off[K1_WAVES][K1L_RMAX + 1];
  __shared__ uint4 sh_desc[K1_WAVES][K1L_DSTAGE];              // staged descriptors, wave-private
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int part = blockIdx.x * K1_WAVES + wv;
  const RecView& rv = A.rv;
  const uint32_t* __restrict__ rb = A.rb;
  Counters* ctr = A.ctr;
  const int n_parts = A.n_parts, min_svlen = A.min_svlen, min_mapq = A.min_mapq, ablate = A.ablate;
  const uint32_t part_ops = A.part_ops;
  if (part >= n_parts) return;
  const uint64_t n_rec = (uint64_t)rv.n_records;
  uint64_t end_all = rv.cigar_off[n_rec];                       // ops behind the last record belong to nobody
  if (end_all > (uint64_t)rv.n_ops) end_all = (uint64_t)rv.n_ops;
  const uint64_t e0 = (uint64_t)part * part_ops;
  uint32_t ord = 0;                                             // rows of this part so far (staged and spilled)
  uint32_t r0 = __builtin_amdgcn_readfirstlane(rb[part]), r1 = __builtin_amdgcn_readfirstlane(rb[part + 1]);
  if (r1 < r0) r1 = r0;                                          // offsets that do not ascend: reported below, never followed
  const uint32_t part_len = __builtin_amdgcn_readfirstlane(e0 >= end_all ? 0u : (uint32_t)((end_all - e0) < (uint64_t)part_ops ? (end_all - e0) : (uint64_t)part_ops));

  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const uint32_t* part_base = rv.cigar + e0;
  const uint32_t base_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)part_base);
  const uint32_t base_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)part_base >> 32));
  // ring loads issued and awaited by hand, raw buffer descriptor clipped to the part: see cigar_scan_long_pool
  u32x4 rsrc;
  rsrc.x = base_lo; rsrc.y = base_hi & 0xFFFFu; rsrc.z = __builtin_amdgcn_readfirstlane(part_len * 4u); rsrc.w = 0x00020000u;
  const uint32_t lane16 = 16u * (uint32_t)lane;
  auto issue = [&](u32x4& dst, uint32_t cb) {
    const uint32_t voff = cb * 4u + lane16;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rsrc) : "memory");
  };
  u32x4 wa, wb, wc, wd;
  issue(wa, 0); issue(wb, 256); issue(wc, 512); issue(wd, 768);
#define K1L_ARRIVE(reg) asm volatile("s_waitcnt vmcnt(3)" : "+v"(reg) : : "memory")

  // ---- record starts of this part, relative to e0, staged K1_RMAX at a time; every record is checked by the part that holds
  // its start (an empty CIGAR must raise: H:63 IndexError) -------------------------------------------------------------------
  const uint32_t e0_lo = (uint32_t)e0;
  uint32_t* my_off = sh_off[wv];
  uint32_t tbase = r0, n_tab = 0, ti = 0;       // staged records [tbase, tbase + n_tab), next unconsumed entry ti
  bool bad = false;
  auto stage = [&](uint32_t first) {
    tbase = first; ti = 0;
    n_tab = min((uint32_t)K1L_RMAX, r1 - first);
    bool ok = true;
    for (uint32_t i = lane; i < n_tab; i += 64) {
      const uint64_t o = rv.cigar_off[first + i], nx = rv.cigar_off[(uint64_t)first + i + 1];
      my_off[i] = (uint32_t)o - e0_lo;
      ok = ok && nx > o && o >= e0 && o - e0 < (uint64_t)part_len;      // ascending, inside this part
      if (CLS == 0 || CLS == 1) {               // a walked record whose SEQ length differs from its CIGAR's: H:397-398, RS:123-124
        const uint32_t fl = rv.flag[first + i];
        if ((fl & VSV_F_SEQ_MISMATCH) && rv.mapq[first + i] >= (uint32_t)min_mapq && (CLS == 1 || (fl & (VSV_F_HP1 | VSV_F_HP2))))
          atomicOr(&ctr->err, ERRB_SEQLEN);
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (__ballot(!ok)) { if (lane == 0) atomicOr(&ctr->err, ERRB_EMPTY_CIGAR); bad = true; }
  };
  if (r1 > r0) stage(r0);
  auto next_start = [&]() -> uint32_t {         // relative op index of the next record start of the part (wave-uniform)
    if (ti >= n_tab) {
      if (tbase + n_tab >= r1 || bad) return 0xFFFFFFFFu;
      stage(tbase + n_tab);
      if (bad || n_tab == 0) return 0xFFFFFFFFu;
    }
    return __builtin_amdgcn_readfirstlane(my_off[ti]);
  };
  uint32_t nxt = (r1 > r0 && !bad) ? next_start() : 0xFFFFFFFFu;
  uint32_t cur_rec = (r0 > 0 && !(nxt == 0u)) ? r0 - 1u : 0xFFFFFFFFu;    // the record that is open at the part's start
  bool in_head = true;                          // no record start seen yet: rows miss the carry of the part
  uint32_t has_start = 0;
  uint32_t run_r = 0, run_q = 0;                // advance since the current record's start (or since e0 while in_head)

  // ---- descriptor stage: it leaves as ONE batch at the end of the part (a part of the pile holds ~16 candidates, the stage 192), in a
  // pool sharded over 256 cursors; a part that fills its stage earlier files those batches in the overflow list ----------------------
  const uint32_t shard = (uint32_t)part % K1_SHARDS;
  uint4* my_desc = sh_desc[wv];
  uint32_t n_staged = 0;
  auto flush = [&](bool last) {
    uint32_t loc = 0, n = 0;
    if (n_staged != 0) {
      uint32_t b = 0;
      if (lane == 0) b = atomicAdd(&A.shard_cnt[shard * 16], n_staged);
      const uint32_t slot = __builtin_amdgcn_readfirstlane(b);
      if (slot + n_staged <= A.shard_cap) {                     // (else: the placement pass reports the shard's use, the rows of this batch stay unwritten)
        loc = shard * A.shard_cap + slot; n = n_staged;
        for (uint32_t i = lane; i < n_staged; i += 64) A.dpool[loc + i] = my_desc[i];
      }
      if (!last && n != 0 && lane == 0) {
        const uint32_t e = atomicAdd(&A.shard_cnt[8], 1u);
        if (e < A.batch_cap) A.batches[e] = make_uint4((uint32_t)part, ord - n_staged, loc, n);
        else atomicOr(&ctr->err, ERRB_CAPACITY);
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (last && lane == 0) A.prec[part] = make_uint2(loc, n);
    n_staged = 0;
  };

  const uint32_t thr = (uint32_t)min_svlen;
  const uint32_t thr16 = thr >= (1u << 28) ? 0xFFFFFFFFu : thr << 4;
  constexpr uint32_t EMIT_R = rev32(EMIT_MASK | (EMIT_MASK << 16));
  constexpr uint32_t BAD_R = rev32(T::BAD | (T::BAD << 16));
  constexpr uint32_t REF2 = dup16(T::REF), QRY2 = dup16(T::QRY);   // bit (w & 31) of these = bit (op) of the table
  uint32_t hd_rec = 0xFFFFFFFFu, hd_fl = 0, hd_mq = 0, hd_tid = 0, hd_first = 0, hd_pos = 0;   // cached record header (SGPRs)
  uint32_t hd_hap = 0, hd_hc = 0;                                                               // ... and what the emissions derive from it

  // ---- the chunk handed from the streaming blocks to the (single) emission block -----------------------------------------
  uint32_t pw[4] = {0, 0, 0, 0}, p_pr0 = 0, p_pq0 = 0;          // its ops and each lane's exclusive in-chunk prefix
  bool pe0 = false, pe1 = false, pe2 = false, pe3 = false;      // ... and which of them are candidates (lane masks: they stay in scalar registers)
  bool p_badany = false;
  uint32_t p_cb = 0, p_tot_r = 0, p_tot_q = 0;
  bool pending = false;

  auto stream = [&](const u32x4& wcur, const uint32_t cb) {
    if (cb >= part_len || bad) return;
    const uint32_t w[4] = {wcur.x, wcur.y, wcur.z, wcur.w};
    const bool e0 = (int32_t)(EMIT_R << (w[0] & 31u)) < 0 && w[0] >= thr16, e1 = (int32_t)(EMIT_R << (w[1] & 31u)) < 0 && w[1] >= thr16;
    const bool e2 = (int32_t)(EMIT_R << (w[2] & 31u)) < 0 && w[2] >= thr16, e3 = (int32_t)(EMIT_R << (w[3] & 31u)) < 0 && w[3] >= thr16;
    bool bd = false;
    if (T::BAD != 0) {
      const uint32_t t = (BAD_R << (w[0] & 31u)) | (BAD_R << (w[1] & 31u)) | (BAD_R << (w[2] & 31u)) | (BAD_R << (w[3] & 31u));
      bd = (int32_t)t < 0;
    }
    uint32_t ar[4], aq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t len = w[k] >> 4;
      ar[k] = len & (uint32_t)__builtin_amdgcn_sbfe(REF2, w[k], 1);
      aq[k] = len & (uint32_t)__builtin_amdgcn_sbfe(QRY2, w[k], 1);
    }
    const uint32_t r1 = ar[0], r2 = r1 + ar[1], r3 = r2 + ar[2], sum_r = r3 + ar[3];      // (the partial sums: what a candidate in sub-slot k has in front of it in its lane)
    const uint32_t q1 = aq[0], q2 = q1 + aq[1], q3 = q2 + aq[2], sum_q = q3 + aq[3];
    const uint32_t incl_r = wave_incl_scan(sum_r), incl_q = wave_incl_scan(sum_q);
    const uint32_t tot_r = rdlane(incl_r, 63), tot_q = rdlane(incl_q, 63);
    const uint64_t m_bd = __ballot(bd);
    const uint64_t m0 = __ballot(e0), m1 = __ballot(e1), m2 = __ballot(e2), m3 = __ballot(e3);
    const uint64_t m01 = m0 | m1, m012 = m01 | m2, many = m012 | m3;
    const bool any_cand = (many | m_bd) != 0ull && !(ablate & 1);
    if (!any_cand && !(nxt < cb + 256u)) { run_r += tot_r; run_q += tot_q; return; }
    // The usual chunk of a contig pile — every second one holds a candidate, nearly always exactly one: no record starts inside, nothing the
    // table forbids, the record's header is cached and carries one haplotype, at most one candidate per lane, room in the stage. Its
    // descriptors are built without a branch (the sub-slot's values by selects) and without the segment loop of the general block
    // below, which cost ~100 scalar + ~50 vector instructions per such chunk.
    if (!(nxt < cb + 256u) && m_bd == 0ull && cur_rec == hd_rec && (hd_hap == 1u || hd_hap == 2u) && n_staged + 64u <= (uint32_t)K1L_DSTAGE &&
        ((m0 & m1) | (m01 & m2) | (m012 & m3)) == 0ull && !(ablate & 2)) {
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(many >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)many, 0u));
      uint32_t wk = e2 ? w[2] : w[3], fr = e2 ? r2 : r3, fq = e2 ? q2 : q3;
      wk = e1 ? w[1] : wk; fr = e1 ? r1 : fr; fq = e1 ? q1 : fq;
      wk = e0 ? w[0] : wk; fr = e0 ? 0u : fr; fq = e0 ? 0u : fq;
      const uint32_t fbits = (in_head ? KD_CARRY : 0u) | ((CLS == 0 && hd_hap == 2u) ? KD_HP2 : 0u);
      const uint4 d = make_uint4(hd_pos + run_r + (incl_r - sum_r) + fr, run_q + hd_hc + (incl_q - sum_q) + fq, (wk & ~15u) | ((wk >> 1) & 1u) | fbits, hd_rec);
      if (e0 | e1 | e2 | e3) my_desc[n_staged + rank] = d;
      const uint32_t n = (uint32_t)__popcll(many);
      n_staged += n; ord += n;
      run_r += tot_r; run_q += tot_q;
      return;
    }
    pw[0] = w[0]; pw[1] = w[1]; pw[2] = w[2]; pw[3] = w[3];
    pe0 = e0 && !(ablate & 1); pe1 = e1 && !(ablate & 1); pe2 = e2 && !(ablate & 1); pe3 = e3 && !(ablate & 1);
    p_pr0 = incl_r - sum_r; p_pq0 = incl_q - sum_q;
    p_badany = any_cand && m_bd != 0ull;
    p_cb = cb; p_tot_r = tot_r; p_tot_q = tot_q;
    pending = true;
  };

  // emission block: the pending chunk is cut into segments at the record starts it holds (usually none: one segment); the
  // candidate ops of a segment belong to one record and leave their descriptors together.
  auto emit_pending = [&]() {
    const uint32_t cb = p_cb;
    const uint32_t x = cb + 4u * (uint32_t)lane;
    uint32_t base_r = run_r, base_q = run_q, lo = cb, hi_cap = 0xFFFFFFFFu;
    for (;;) {
      uint32_t hi = nxt < cb + 256u ? nxt : cb + 256u;
      if (hi_cap < hi) hi = hi_cap;
      if (hi > lo && cur_rec != 0xFFFFFFFFu) {
        bool e[4] = {pe0, pe1, pe2, pe3};                        // (the streaming block's tests: not evaluated again)
        const bool whole = lo == cb && hi == cb + 256u;
        if (!whole) {
#pragma unroll
          for (int k = 0; k < 4; ++k) e[k] = e[k] && x + k >= lo && x + k < hi;
        }
        const uint32_t c = (e[0] ? 1u : 0u) + (e[1] ? 1u : 0u) + (e[2] ? 1u : 0u) + (e[3] ? 1u : 0u);
        const uint64_t many = __ballot(c != 0u);
        bool any_b = false;
        if (T::BAD != 0 && p_badany) {
          bool b = false;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            uint32_t wk = pw[k];
            asm volatile("" : "+v"(wk));       // keep the test behind its guard (the optimizer hoists loop-invariant arithmetic)
            b = b || ((int32_t)(BAD_R << (wk & 31u)) < 0 && wk >= 16u && x + k >= lo && x + k < hi);
          }
          any_b = __ballot(b) != 0ull;
        }
        if (many != 0ull || any_b) {
          const uint32_t rec = __builtin_amdgcn_readfirstlane(cur_rec);
          if (rec != hd_rec) {                                   // header through the scalar cache, once per record and part
            const uint64_t* po = rv.cigar_off + rec;
            uint64_t fo;
            asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(fo) : "s"(po) : "memory");
            if (fo >= (uint64_t)rv.n_ops) fo = 0;                // garbage offsets are reported by the part that stages them
            sload5(align4(rv.flag + rec), align4(rv.mapq + rec), rv.tid + rec, rv.cigar + fo, rv.pos + rec, hd_fl, hd_mq, hd_tid, hd_first, hd_pos);
            hd_rec = rec;
            const uint32_t fl = byte_of(hd_fl, rv.flag + rec), mq = byte_of(hd_mq, rv.mapq + rec);
            if (CLS == 0) hd_hap = (mq >= (uint32_t)min_mapq) ? ((fl >> 2) & 3u) : 0u;               // H:392-394
            else if (CLS == 1) hd_hap = (mq >= (uint32_t)min_mapq) ? 1u : 0u;                        // RS:120
            else if (CLS == 3) hd_hap = (!(fl & VSV_F_SKIP) && mq >= (uint32_t)min_mapq) ? 1u : 0u; // SE:439, 446
            else hd_hap = (!(fl & (VSV_F_UNMAPPED | VSV_F_SECONDARY)) && mq >= (uint32_t)min_mapq) ? 1u : 0u;
            hd_hc = (T::HC && (hd_first & 15u) == 5u) ? (hd_first >> 4) : 0u;                       // H:63-65
          }
          const uint32_t hapbits = hd_hap;
          if (hapbits && any_b && lane == 0) atomicOr(&ctr->err, ERRB_REFEND);   // N/=/X on the contig table: H:396 assert
          if (hapbits && many != 0ull) {
            const uint32_t sh = (hapbits == 3u) ? 1u : 0u;       // two rows per signature when the name carries both tags
            uint32_t n, rank;
            if (__ballot(c > 1u) == 0ull) {                      // (nearly always: at most one candidate per lane)
              n = (uint32_t)__popcll(many);
              rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(many >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)many, 0u));
            } else {
              const uint32_t incl = wave_incl_scan(c);
              n = rdlane(incl, 63); rank = incl - c;
            }
            const uint32_t nrows = n << sh;
            if (nrows > (uint32_t)K1L_DSTAGE) { hi_cap = lo + (uint32_t)K1L_DSTAGE / 2u; continue; }   // cut the segment: 96 ops hold at most 192 rows
            if (n_staged + nrows > (uint32_t)K1L_DSTAGE) flush(false);
            const uint32_t pos0 = hd_pos + base_r, q0 = base_q + hd_hc;
            const uint32_t fbits = (in_head ? KD_CARRY : 0u) | ((CLS == 0 && hapbits == 2u) ? KD_HP2 : 0u);
            if (c != 0u) {
              uint32_t slot = n_staged + (rank << sh);
              uint32_t pr = p_pr0, pq = p_pq0;                   // prefix in front of sub-slot k, advanced as k goes up
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                uint32_t wk = pw[k];
                asm volatile("" : "+v"(wk));   // descriptors are built only by the lanes that hold a candidate: keep the arithmetic here
                const uint32_t len = wk >> 4;
                if (e[k]) {
                  const uint4 d = make_uint4(pos0 + pr, q0 + pq, (wk & ~15u) | ((wk >> 1) & 1u) | fbits, rec);     // (op 1 = I, 2 = D: bit 1 of the op is "DEL")
                  my_desc[slot] = d;
                  if (sh) my_desc[slot + 1u] = make_uint4(d.x, d.y, d.z | KD_HP2, d.w);
                  slot += 1u << sh;
                }
                if (k < 3) {
                  pr += len & (uint32_t)__builtin_amdgcn_sbfe(REF2, wk, 1);
                  pq += len & (uint32_t)__builtin_amdgcn_sbfe(QRY2, wk, 1);
                }
              }
            }
            __builtin_amdgcn_wave_barrier();
            n_staged += nrows; ord += nrows;
          }
        }
      }
      lo = hi; hi_cap = 0xFFFFFFFFu;
      if (lo >= cb + 256u) break;
      if (lo != nxt) continue;                                   // a cut segment: same record, next slice
      // ---- a record starts at nxt: ops from there on lie px - P(start) behind their record's start ----
      {
        const uint32_t ls = (nxt - cb) >> 2, ss = (nxt - cb) & 3u;
        uint32_t pr = p_pr0, pq = p_pq0;
        for (uint32_t k = 0; k < ss; ++k) {                      // prefix in front of sub-slot ss of lane ls (wave-uniform trip count)
          uint32_t wk = k == 0 ? pw[0] : k == 1 ? pw[1] : pw[2];
          asm volatile("" : "+v"(wk));
          const uint32_t len = wk >> 4;
          pr += len & (uint32_t)__builtin_amdgcn_sbfe(REF2, wk, 1);
          pq += len & (uint32_t)__builtin_amdgcn_sbfe(QRY2, wk, 1);
        }
        base_r = 0u - rdlane(pr, ls); base_q = 0u - rdlane(pq, ls);
      }
      cur_rec = tbase + ti; in_head = false; has_start = 1;
      ++ti;
      nxt = next_start();
      if (bad) return;
    }
    run_r = base_r + p_tot_r; run_q = base_q + p_tot_q;
  };

  uint32_t cb = 0, slot = 0;
  while (cb < part_len && !bad) {
    if (slot == 0 && !pending) { K1L_ARRIVE(wa); stream(wa, cb); issue(wa, cb + 1024); cb += 256; slot = 1; }
    if (slot == 1 && !pending) { K1L_ARRIVE(wb); stream(wb, cb); issue(wb, cb + 1024); cb += 256; slot = 2; }
    if (slot == 2 && !pending) { K1L_ARRIVE(wc); stream(wc, cb); issue(wc, cb + 1024); cb += 256; slot = 3; }
    if (slot == 3 && !pending) { K1L_ARRIVE(wd); stream(wd, cb); issue(wd, cb + 1024); cb += 256; slot = 0; }
    if (pending) { emit_pending(); pending = false; }
  }
  // the last prefetches are still in flight and will write their (clipped, zero) data into the ring registers: the registers stay
  // reserved until everything has landed — the compiler does not know about loads issued by inline asm
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(wa), "+v"(wb), "+v"(wc), "+v"(wd) : : "memory");
  if (pending && !bad) emit_pending();
#undef K1L_ARRIVE
  // records whose starts were never reached (offsets beyond the part, part cut short by end_all): still validate them
  while (!bad && tbase + n_tab < r1) stage(tbase + n_tab);
  flush(true);
  if (lane == 0) { A.part_count[part] = ord; PartAgg a; a.has_start = has_start; a.run_r = run_r; a.run_q = run_q; A.agg[part] = a; }
}

// ---- carry of the long-record scan: segmented exclusive scan over the parts --------------------------------------------
// x[p] = (has_start, run): S = x[0] (+) ... (+) x[p-1] with (f1,v1) (+) (f2,v2) = (f1|f2, f2 ? v2 : v1+v2); carry[p] = S.v.
// Same three-kernel shape as the count scan below; the counts' exclusive sum (part_off) rides along.
struct CarryItem { uint32_t f, r, q, cnt; };
__device__ __forceinline__ CarryItem carry_op(const CarryItem& a, const CarryItem& b) {
  CarryItem c;
  c.f = a.f | b.f; c.r = b.f ? b.r : a.r + b.r; c.q = b.f ? b.q : a.q + b.q; c.cnt = a.cnt + b.cnt;
  return c;
}
// block-wide inclusive scan of one item per thread (Hillis-Steele in LDS, 256 threads)
__device__ __forceinline__ CarryItem k1l_block_scan(CarryItem v, CarryItem* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    CarryItem t = sh[threadIdx.x];
    if ((int)threadIdx.x >= d) t = carry_op(sh[threadIdx.x - d], t);
    __syncthreads();
    sh[threadIdx.x] = t;
    __syncthreads();
  }
  return sh[threadIdx.x];
}
// ---- placement of the long scan's descriptors: part scan by look-back + rows, one launch ------------------------------------------
// A block owns 256 consecutive parts: their (has_start, advance, rows) are scanned in LDS, the block's aggregate goes through the
// look-back over the BLOCKS in front (lb_publish / lb_resolve: every block reaches it a few microseconds after it starts and all
// blocks are alike, so nobody waits long), and then the block's rows — consecutive in T_RAW — are built from the descriptors of its
// parts' batches and stored coalesced. part_off / carries go to memory for the batches of the overflow list (k1l_place_batches).
struct PlaceArgs {
  const uint4* dpool; const uint32_t* shard_cnt; uint32_t shard_cap;
  const uint2* prec; const uint32_t* part_count; const PartAgg* agg; int n_parts;
  uint64_t* lbw; uint32_t epoch;
  const int32_t* tid; vsv_sig* raw; uint32_t cap; Counters* ctr;
  uint32_t* part_off; uint32_t* carry_r; uint32_t* carry_q; int no_qend;
  SlimOut so;                                       // the rows' 16-byte elements next to them (base == nullptr: rows only)
  int pool_rows;                                    // the batches hold finished 32-byte rows (the read-shaped scan: parts end at record ends, no carry), not descriptors
};
__device__ __forceinline__ void k1l_store(vsv_sig* __restrict__ raw, uint32_t row, const uint4& lo4, const uint4& hi4, const SlimOut& so) {
  uint4* dst = reinterpret_cast<uint4*>(raw + row);
  dst[0] = lo4; dst[1] = hi4;
  if (so.base) {
    vsv_sig v;
    v.pos = (int32_t)lo4.x; v.svlen = (int32_t)lo4.y; v.q_start = (int32_t)lo4.z; v.q_end = (int32_t)lo4.w;
    v.rec = hi4.x; v.rec2 = hi4.y; v.meta = hi4.z; v.tid = (int32_t)hi4.w;
    vsv_slim_emit(so, row, v);
  }
}
constexpr int PL_PARTS = 256;
__global__ __launch_bounds__(256) void k1l_place(PlaceArgs A) {
  __shared__ CarryItem sh[256];
  __shared__ uint32_t s_off[PL_PARTS + 1], s_cr[PL_PARTS], s_cq[PL_PARTS], s_loc[PL_PARTS], s_first[PL_PARTS];
  __shared__ uint32_t s_ex[4];
  const int t = threadIdx.x, lane = t & 63;
  const int part = blockIdx.x * PL_PARTS + t;
  CarryItem it{0, 0, 0, 0};
  uint2 pr = make_uint2(0u, 0u);
  if (part < A.n_parts) {
    if (A.agg) { const PartAgg a = A.agg[part]; it.f = a.has_start; it.r = a.run_r; it.q = a.run_q; }
    it.cnt = A.part_count[part]; pr = A.prec[part];
  }
  k1l_block_scan(it, sh);                           // sh[t] = parts [block start, t] combined
  if (t < 64) {                                     // (one whole wave: the look-back is wave-cooperative)
    const CarryItem tot = sh[255];
    LbItem own; own.f = tot.f; own.r = tot.r; own.q = tot.q; own.cnt = tot.cnt;
    lb_publish(A.lbw, A.epoch, blockIdx.x, own, LB_AGG, lane);
    const LbItem ex = lb_resolve(A.lbw, A.epoch, blockIdx.x, own, lane, &A.ctr->err);
    if (lane == 0) { s_ex[0] = ex.f; s_ex[1] = ex.r; s_ex[2] = ex.q; s_ex[3] = ex.cnt; }
  }
  __syncthreads();
  {
    CarryItem ex{s_ex[0], s_ex[1], s_ex[2], s_ex[3]};
    if (t > 0) ex = carry_op(ex, sh[t - 1]);
    s_off[t] = ex.cnt; s_cr[t] = ex.r; s_cq[t] = ex.q; s_loc[t] = pr.x; s_first[t] = it.cnt - pr.y;
    if (t == 255) s_off[PL_PARTS] = ex.cnt + it.cnt;
    if (part < A.n_parts) { A.part_off[part] = ex.cnt; if (A.agg) { A.carry_r[part] = ex.r; A.carry_q[part] = ex.q; } }
    if (part == A.n_parts - 1) {                    // the table's size, and what a retry would have to reserve
      const uint64_t total = (uint64_t)ex.cnt + it.cnt;
      A.ctr->n_raw = total < (uint64_t)A.cap ? (uint32_t)total : A.cap;
      atomicMax(&A.ctr->n_pool, total > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)total);
      if (total > (uint64_t)A.cap) atomicOr(&A.ctr->err, ERRB_CAPACITY);
    }
  }
  if (blockIdx.x == 0 && t == 0) A.ctr->pad[0] = A.shard_cnt[8];      // overflow batches of this run: the host skips their launch while there are none
  if (blockIdx.x == 0) {                            // a shard that overflowed: rows are missing, and a retry needs shards of that size
    const uint32_t used = A.shard_cnt[t * 16];      // (K1_SHARDS == 256 threads)
    if (used > A.shard_cap) {
      atomicOr(&A.ctr->err, ERRB_CAPACITY);
      const uint64_t need = A.pool_rows ? (uint64_t)used * K1_SHARDS : ((uint64_t)used * K1_SHARDS + 1) / 2;     // (a row of capacity holds two descriptors)
      atomicMax(&A.ctr->n_pool, need > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)need);
    }
  }
  __syncthreads();
  const uint32_t off0 = s_off[0], rows = s_off[PL_PARTS] - off0;
  for (uint32_t i = t; i < rows; i += 256) {
    const uint32_t row = off0 + i;
    uint32_t lo = 0, hi = PL_PARTS;                 // the part p with s_off[p] <= row < s_off[p + 1]
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_off[mid] <= row) lo = mid; else hi = mid; }
    const uint32_t o = row - s_off[lo];
    if (o < s_first[lo] || row >= A.cap) continue;  // (a row of an earlier batch of the part: k1l_place_batches)
    const uint32_t src = s_loc[lo] + (o - s_first[lo]);
    uint4 lo4, hi4;
    if (A.pool_rows) { lo4 = A.dpool[2 * (size_t)src]; hi4 = A.dpool[2 * (size_t)src + 1]; }
    else { const uint4 d = A.dpool[src]; k1l_row(d, s_cr[lo], s_cq[lo], (uint32_t)A.tid[d.w], A.no_qend != 0, lo4, hi4); }
    k1l_store(A.raw, row, lo4, hi4, A.so);
  }
}
// the batches a part filed before its last one (a part with more candidates than its stage holds): a wave per batch
__global__ __launch_bounds__(256) void k1l_place_batches(const uint4* __restrict__ batches, const uint32_t* __restrict__ n_batches, uint32_t batch_cap,
                                                         const uint4* __restrict__ dpool, const uint32_t* __restrict__ part_off, const uint32_t* __restrict__ carry_r,
                                                         const uint32_t* __restrict__ carry_q, const int32_t* __restrict__ tid, vsv_sig* __restrict__ raw, uint32_t cap, int no_qend,
                                                         SlimOut so, int pool_rows) {
  const uint32_t nb = min(*n_batches, batch_cap);
  const int lane = threadIdx.x & 63;
  for (uint32_t e = blockIdx.x * 4 + (threadIdx.x >> 6); e < nb; e += gridDim.x * 4) {
    const uint4 b = batches[e];                     // {part, first ordinal, location, descriptors}
    const uint32_t row0 = part_off[b.x] + b.y, cr = pool_rows ? 0u : carry_r[b.x], cq = pool_rows ? 0u : carry_q[b.x];
    for (uint32_t i = lane; i < b.w; i += 64) {
      if (row0 + i >= cap) break;
      uint4 lo4, hi4;
      if (pool_rows) { lo4 = dpool[2 * (size_t)(b.z + i)]; hi4 = dpool[2 * (size_t)(b.z + i) + 1]; }
      else { const uint4 d = dpool[b.z + i]; k1l_row(d, cr, cq, (uint32_t)tid[d.w], no_qend != 0, lo4, hi4); }
      k1l_store(raw, row0 + i, lo4, hi4, so);
    }
  }
}

// ---- exclusive scan of part_count (3 tiny kernels) -------------------------------------------------
constexpr int SCAN_TILE = 2048;  // 256 threads x 8
__global__ __launch_bounds__(256) void scan_tile_sums(const uint32_t* __restrict__ in, int n, uint32_t* __restrict__ sums) {
  __shared__ uint32_t sh[256];
  const int base = blockIdx.x * SCAN_TILE;
  uint32_t s = 0;
  for (int k = 0; k < 8; ++k) { int i = base + k * 256 + threadIdx.x; if (i < n) s += in[i]; }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) { if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d]; __syncthreads(); }
  if (threadIdx.x == 0) sums[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(1024) void scan_sums_inplace(uint32_t* __restrict__ sums, int n) {
  // single block: serial over chunks of 1024 with a Hillis-Steele scan in LDS
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    int i = base + threadIdx.x;
    uint32_t v = i < n ? sums[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      uint32_t t = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    uint32_t incl = sh[threadIdx.x], c = carry;
    if (i < n) sums[i] = c + incl - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry = c + incl;
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void scan_tile_apply(const uint32_t* __restrict__ in, int n, const uint32_t* __restrict__ sums,
                                                       uint32_t* __restrict__ out) {
  // thread t owns 8 consecutive items of the tile
  __shared__ uint32_t sh[256];
  const int base = blockIdx.x * SCAN_TILE + threadIdx.x * 8;
  uint32_t v[8], s = 0;
  for (int k = 0; k < 8; ++k) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    uint32_t t = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = sums[blockIdx.x] + sh[threadIdx.x] - s;
  for (int k = 0; k < 8; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
}

}  // namespace

// one block for small arrays (n <= 8192: the per-block candidate counts of a 10 M-record shard): thread t owns 8 consecutive items
__global__ __launch_bounds__(1024) void scan_small(const uint32_t* __restrict__ in, int n, uint32_t* __restrict__ out) {
  __shared__ uint32_t sh[1024];
  const int base = threadIdx.x * 8;
  uint32_t v[8], s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const uint32_t t = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = sh[threadIdx.x] - s;
#pragma unroll
  for (int k = 0; k < 8; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
}

void vsv_scan_u32_exclusive(hipStream_t st, const uint32_t* in, int n, uint32_t* out, uint32_t* tmp) {
  if (n <= 0) return;
  if (n <= 8192) { scan_small<<<1, 1024, 0, st>>>(in, n, out); return; }
  const int tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  scan_tile_sums<<<tiles, 256, 0, st>>>(in, n, tmp);
  scan_sums_inplace<<<1, 1024, 0, st>>>(tmp, tiles);
  scan_tile_apply<<<tiles, 256, 0, st>>>(in, n, tmp, out);
}

int vsv_cigar_parts(int64_t n_ops, int ops_per_part) { return (int)((n_ops + ops_per_part - 1) / ops_per_part); }

bool vsv_scan_is_long(const RecView& rv, const vsv_params& p) {
  static const char* force = vsv_dbg_env("VSV_K1_MODE");              // timing experiments: "long" / "short"
  if (force) return force[0] == 'l';
  if (p.scan_layout == VSV_SCAN_READS) return false;
  if (p.scan_layout == VSV_SCAN_CONTIGS) return true;
  return rv.n_records > 0 && rv.n_ops / rv.n_records >= 512;    // Mb contigs: 10^3-10^6 ops per record; reads: tens to hundreds
}
int vsv_cigar_parts_long(int64_t n_ops) { return (int)((n_ops + K1L_PART - 1) / K1L_PART); }

// the scan's batches -> rows in T_RAW order (and their elements): behind the scan, and again into the raw table when a caller asks
// for VSV_T_RAW of a run whose rows went straight into the stage-1 table (capi.hip). n_parts: the scan's (vsv_scan_parts).
void vsv_launch_place(hipStream_t st, const RecView& rv, const vsv_params& p, int n_parts, vsv_sig* pool, uint64_t* pool_key, uint32_t cap, uint32_t* part_count,
                      uint32_t* part_off, vsv_sig* rows, Counters* ctr, uint32_t* shard_cnt, const LongScanBufs& lb, uint32_t epoch, const SlimOut& so,
                      bool with_batches) {
  if (n_parts <= 0) return;
  const bool long_mode = vsv_scan_is_long(rv, p);
  const int no_qend = (p.dtype == VSV_DTYPE_READS || p.dtype == VSV_DTYPE_CUTESV) ? 1 : 0;
  const int pool_rows = long_mode ? 0 : 1;
  const uint32_t shard_cap = (uint32_t)(((uint64_t)cap * (long_mode ? 2 : 1)) / K1_SHARDS);      // (a row of the pool holds two 16-byte descriptors)
  k1l_place<<<(n_parts + PL_PARTS - 1) / PL_PARTS, 256, 0, st>>>(PlaceArgs{(const uint4*)pool, shard_cnt, shard_cap, (const uint2*)lb.prec, part_count,
                                                                 long_mode ? (const PartAgg*)lb.agg : nullptr, n_parts, lb.lbw, epoch, rv.tid, rows, cap, ctr, part_off,
                                                                 lb.carry_r, lb.carry_q, no_qend, so, pool_rows});
  if (with_batches) k1l_place_batches<<<64, 256, 0, st>>>((const uint4*)pool_key, shard_cnt + 8, cap / 2, (const uint4*)pool, part_off, lb.carry_r, lb.carry_q, rv.tid, rows, cap, no_qend, so, pool_rows);
}
// ops per part and parts of the scan this call takes (the long scan's parts are never shorter than the read-shaped ones: the part
// tables are sized for those)
static int scan_part_ops(const RecView& rv, const vsv_params& p, int ops_per_part) {
  if (!vsv_scan_is_long(rv, p)) return ops_per_part;
  static const int forced = vsv_dbg_env("VSV_K1L_PART") ? atoi(vsv_dbg_env("VSV_K1L_PART")) : 0;      // timing experiments / tests
  int64_t po = K1L_PART;
  if (forced >= 256 && forced % 256 == 0) po = forced;
  if (po < ops_per_part) po = ((ops_per_part + 255) / 256) * 256;
  return (int)po;
}
int vsv_scan_parts(const RecView& rv, const vsv_params& p, int ops_per_part) { return vsv_cigar_parts(rv.n_ops, scan_part_ops(rv, p, ops_per_part)); }

// K0 + K1 + placement: part boundaries, the scan (read-shaped or long-record layout), rows in T_RAW order in `raw` — or, with
// lb.fused_rows, in the stage-1 input table next to their elements.
void vsv_launch_cigar_scan(hipStream_t st, const RecView& rv, const vsv_params& p, uint32_t* part_rb, int n_parts,
                           int ops_per_part, vsv_sig* pool, uint64_t* pool_key, uint32_t cap, uint32_t* part_count,
                           uint32_t* part_off, uint32_t* scan_tmp, vsv_sig* raw, Counters* ctr, uint32_t* shard_cnt,
                           hipEvent_t ev0, hipEvent_t ev1, const LongScanBufs& lb) {
  if (n_parts <= 0) return;
  (void)scan_tmp;
  const bool long_mode = vsv_scan_is_long(rv, p);
  ops_per_part = scan_part_ops(rv, p, ops_per_part);
  n_parts = vsv_cigar_parts(rv.n_ops, ops_per_part);
  if (!lb.arena_zeroed) (void)hipMemsetAsync(shard_cnt, 0, K1_SHARDS * 16 * sizeof(uint32_t), st);
  if (n_parts < (1 << 20)) partition_search<<<(n_parts + 1 + 255) / 256, 256, 0, st>>>(rv.cigar_off, rv.n_records, part_rb, n_parts, ops_per_part, rv.n_ops, 1);
  else {
    const int coarse = n_parts / 64 + 1;
    partition_search<<<(coarse + 255) / 256, 256, 0, st>>>(rv.cigar_off, rv.n_records, part_rb, n_parts, ops_per_part, rv.n_ops, 64);
    partition_fill<<<(n_parts + 1 + 63) / 64, 64, 0, st>>>(rv.cigar_off, rv.n_records, part_rb, n_parts, ops_per_part, rv.n_ops);
  }
  const EmitCtx ec{pool, shard_cnt, cap / K1_SHARDS, ctr, (uint2*)lb.prec, (uint4*)pool_key, cap / 2};
  const uint32_t dshard_cap = (uint32_t)(((uint64_t)cap * 2) / K1_SHARDS);      // the row pool holds two 16-byte descriptors per row
  const int waves_per_block = 4;
  const int grid = (n_parts + waves_per_block - 1) / waves_per_block;
  if (ev0) (void)hipEventRecord(ev0, st);
  static const int ablate = vsv_dbg_env("VSV_K1_ABLATE") ? atoi(vsv_dbg_env("VSV_K1_ABLATE")) : 0;  // timing experiments only
  static const int depth = vsv_dbg_env("VSV_K1_DEPTH") ? atoi(vsv_dbg_env("VSV_K1_DEPTH")) : 4;     // chunks in flight per wave (3 or 4)
#define K1_LAUNCH(CLS)                                                                                                          \
  do {                                                                                                                          \
    if (long_mode) cigar_scan_long<CLS><<<grid, 256, 0, st>>>(K1LArgs{rv, part_rb, ctr, (uint4*)pool, shard_cnt, dshard_cap, (uint4*)pool_key, cap / 2, \
                                                                      (uint2*)lb.prec, part_count, (PartAgg*)lb.agg, n_parts, p.min_svlen, p.min_cigar_mapq, (uint32_t)ops_per_part, ablate}); \
    else if (CLS == 0 && lb.clr_fused) cigar_scan_emit<0, 4, true><<<grid, 256, 0, st>>>(rv, part_rb, n_parts, p.min_svlen, p.min_cigar_mapq, ec, part_count, ablate); \
    else if (depth == 4) cigar_scan_emit<CLS, 4><<<grid, 256, 0, st>>>(rv, part_rb, n_parts, p.min_svlen, p.min_cigar_mapq, ec, part_count, ablate); \
    else cigar_scan_emit<CLS, 3><<<grid, 256, 0, st>>>(rv, part_rb, n_parts, p.min_svlen, p.min_cigar_mapq, ec, part_count, ablate);            \
  } while (0)
  if (p.dtype == VSV_DTYPE_READS) K1_LAUNCH(1);
  else if (p.dtype == VSV_DTYPE_SVIM) K1_LAUNCH(2);
  else if (p.dtype == VSV_DTYPE_CUTESV) K1_LAUNCH(3);
  else K1_LAUNCH(0);
#undef K1_LAUNCH
  if (ev1) (void)hipEventRecord(ev1, st);
  vsv_launch_place(st, rv, p, n_parts, pool, pool_key, cap, part_count, part_off, lb.fused_rows ? (vsv_sig*)lb.fused_rows : raw, ctr, shard_cnt, lb, lb.epoch,
                   lb.fused_rows ? lb.so : SlimOut{nullptr, 0, 0, 0, nullptr}, !lb.skip_batches);
}
size_t vsv_lookback_bytes(int64_t n_ops, int ops_per_part) {
  const size_t a = (size_t)vsv_cigar_parts_long(n_ops) + 16, b = (size_t)vsv_cigar_parts(n_ops, ops_per_part) + 16;
  return (a > b ? a : b) * 3 * sizeof(uint64_t);
}
size_t vsv_long_scan_bytes(int64_t n_ops, int which, int ops_per_part) {   // 0: PartAgg[], 1: carry (u32 per part), 3: last-batch records (uint2 per part)
  const size_t pa = (size_t)vsv_cigar_parts_long(n_ops), pb = (size_t)vsv_cigar_parts(n_ops, ops_per_part);      // (the long scan's parts are never shorter than the read-shaped ones)
  const size_t n_parts = (pa > pb ? pa : pb) + 16;
  return which == 0 ? n_parts * sizeof(PartAgg) : which == 1 ? n_parts * sizeof(uint32_t) : which == 3 ? n_parts * sizeof(uint2) : 256;
}

// ---- streaming ceiling of this part, measured with the library's own kernels -----------------------------------
// read: every thread XORs its 16-byte loads (one word per block could leave the kernel, so nothing is elided);
// copy: 16-byte load + store. The scan is a read stream with ~2 % of writes, so `read` is its ceiling; `copy` is the usual figure.
// Same access shape as the scan: a wave owns a contiguous 16 KiB part and sweeps it with 1 KiB (64 x 16 B) loads, 4 in flight (16 in flight, 4096-65536-block grid-stride sweeps and strided unrolls measured lower).
constexpr size_t STREAM_PART16 = 1024;      // 16-byte words per wave
__global__ __launch_bounds__(256) void stream_read_kernel(const uint4* __restrict__ src, size_t n16, uint32_t* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const size_t part = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), base = part * STREAM_PART16;
  uint32_t acc = 0;
  if (base + STREAM_PART16 <= n16) {
#pragma unroll
    for (int k = 0; k < 16; k += 4) {
      const uint4 v0 = src[base + (k + 0) * 64 + lane], v1 = src[base + (k + 1) * 64 + lane];
      const uint4 v2 = src[base + (k + 2) * 64 + lane], v3 = src[base + (k + 3) * 64 + lane];
      acc ^= v0.x ^ v0.y ^ v0.z ^ v0.w ^ v1.x ^ v1.y ^ v1.z ^ v1.w ^ v2.x ^ v2.y ^ v2.z ^ v2.w ^ v3.x ^ v3.y ^ v3.z ^ v3.w;
    }
  } else {
    for (size_t i = base + lane; i < n16; i += 64) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  }
  if (acc == 0x9E3779B9u) sink[blockIdx.x & 255] = acc;     // practically never taken; keeps the loads alive
}
__global__ __launch_bounds__(256) void stream_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  const int lane = threadIdx.x & 63;
  const size_t part = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), base = part * STREAM_PART16;
  if (base + STREAM_PART16 <= n16) {
#pragma unroll
    for (int k = 0; k < 16; k += 4) {
      const uint4 v0 = src[base + (k + 0) * 64 + lane], v1 = src[base + (k + 1) * 64 + lane];
      const uint4 v2 = src[base + (k + 2) * 64 + lane], v3 = src[base + (k + 3) * 64 + lane];
      dst[base + (k + 0) * 64 + lane] = v0; dst[base + (k + 1) * 64 + lane] = v1;
      dst[base + (k + 2) * 64 + lane] = v2; dst[base + (k + 3) * 64 + lane] = v3;
    }
  } else {
    for (size_t i = base + lane; i < n16; i += 64) dst[i] = src[i];
  }
}
static int stream_grid(size_t bytes) { return (int)((bytes / 16 + 4 * STREAM_PART16 - 1) / (4 * STREAM_PART16)); }
void vsv_launch_stream_read(hipStream_t st, const void* src, size_t bytes, uint32_t* sink) {
  stream_read_kernel<<<stream_grid(bytes), 256, 0, st>>>((const uint4*)src, bytes / 16, sink);
}
void vsv_launch_stream_copy(hipStream_t st, const void* src, void* dst, size_t bytes) {
  stream_copy_kernel<<<stream_grid(bytes), 256, 0, st>>>((const uint4*)src, (uint4*)dst, bytes / 16);
}
