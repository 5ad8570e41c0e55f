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
//    wave-instruction, fully coalesced), decodes the 4 ops with bit-field extracts against constant op
//    masks, and the wave takes ONE plain (unsegmented) DPP prefix sum of the (ref,query) advances.
//    Arithmetic is mod 2^32, so differences of prefix values are exact.
//  * Record boundaries never enter the vector path. Up to 64 records of the part sit one-per-lane in
//    registers (start offset, pos); the scalar unit tracks the record that is open at the chunk edge
//    (ballot + popcount + v_readlane) and its prefix base. A signature's reference position is
//    pos[rec] + (P(op) - P(record start)).
//  * Emission (>= min_svlen I/D ops, ~1 per 100 records) is a wave-uniform slow path: ballot, then a
//    scalar loop per emitting op that finds the record (ballot over lane-held starts), applies the
//    mapq/hp filters and appends to a pool with one atomic. Each row carries (part, ordinal), and
//    place_raw scatters rows to part_off[part]+ordinal, so T_RAW is in (record, op) order whatever
//    the atomic order was.
#include "vsv_device.h"

namespace {

// ---- op tables as 16-bit masks (bit op set => op advances / emits) ---------------------------------
// contig (H:72-85): M both; S query; D ref+emit; I query+emit; everything else ignored.
// reads  (RS:66-81): M,=,X both; N ref. svim (SVIM_intra.py:13-29): M,=,X both; no H offset.
template <int CLS> struct OpTab;
template <> struct OpTab<0> { static constexpr uint32_t REF = 0x005, QRY = 0x013, BAD = 0x188; static constexpr bool HC = true; };
template <> struct OpTab<1> { static constexpr uint32_t REF = 0x18D, QRY = 0x193, BAD = 0x000; static constexpr bool HC = true; };
template <> struct OpTab<2> { static constexpr uint32_t REF = 0x185, QRY = 0x193, BAD = 0x000; static constexpr bool HC = false; };
constexpr uint32_t EMIT_MASK = 0x006;  // I(1), D(2)

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
__global__ void partition_kernel(const uint64_t* __restrict__ cigar_off, int64_t n_records, uint32_t* __restrict__ rb,
                                 int n_parts, int ops_per_part) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > n_parts) return;
  if (p == n_parts) { rb[p] = (uint32_t)n_records; return; }
  uint64_t target = (uint64_t)p * (uint64_t)ops_per_part;
  int64_t lo = 0, hi = n_records;  // first r in [0,n] with cigar_off[r] >= target
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if (cigar_off[mid] >= target) hi = mid; else lo = mid + 1;
  }
  rb[p] = (uint32_t)lo;
}

struct EmitCtx {
  vsv_sig* pool;
  uint64_t* pool_key;
  uint32_t cap;
  Counters* ctr;
};

// ---- K1 ------------------------------------------------------------------------------------------
template <int CLS>
__global__ __launch_bounds__(256) void cigar_scan_emit(RecView rv, const uint32_t* __restrict__ rb, int n_parts,
                                                        int min_svlen, int min_mapq, EmitCtx ec,
                                                        uint32_t* __restrict__ part_count) {
  using T = OpTab<CLS>;
  const int lane = threadIdx.x & 63;
  const int part = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (part >= n_parts) return;
  const uint32_t r0 = rb[part], r1 = rb[part + 1];
  uint32_t ord = 0;  // signatures emitted by this part so far (wave-uniform)
  const uint4* __restrict__ cig4 = reinterpret_cast<const uint4*>(rv.cigar);

  for (uint32_t rbase = r0; rbase < r1; rbase += 64) {
    const uint32_t nrec = min(64u, r1 - rbase);
    const bool rv_ok = (uint32_t)lane < nrec;
    const uint32_t my_r = rbase + min((uint32_t)lane, nrec - 1);
    const uint64_t so = rv.cigar_off[my_r];
    const uint32_t rpos = (uint32_t)rv.pos[my_r];
    const uint64_t ob = __builtin_amdgcn_readfirstlane((uint32_t)so) |
                        ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(so >> 32)) << 32);
    const uint64_t oe = rv.cigar_off[rbase + nrec];
    const uint64_t cb0 = ob & ~3ull;  // absolute op index of rel 0 (16-byte aligned)
    if (oe - cb0 >= 0x7FFFFF00ull) {  // relative offsets must fit 31 bits
      if (lane == 0) atomicOr(&ec.ctr->err, ERRB_RANGE);
      break;
    }
    const uint32_t s_rel = (uint32_t)(so - cb0);
    const uint32_t ob_rel = (uint32_t)(ob - cb0), oe_rel = (uint32_t)(oe - cb0);
    {  // strictly increasing starts (empty CIGAR => reference IndexError at H:63)
      uint64_t nxt = rv.cigar_off[my_r + 1];
      if (__ballot(rv_ok && nxt <= so)) {
        if (lane == 0) atomicOr(&ec.ctr->err, ERRB_EMPTY_CIGAR);
        break;
      }
    }

    uint32_t pbase_r = 0, pbase_q = 0;   // prefix value at chunk start (mod 2^32)
    uint32_t open_r = 0, open_q = 0;     // prefix value at the open record's start
    uint32_t jprev = 0;                  // records of this batch started before the current chunk

    for (uint32_t cb = 0; cb < oe_rel; cb += 256) {
      const uint32_t x = cb + 4u * (uint32_t)lane;  // rel index of this lane's first op
      uint32_t w[4] = {15u, 15u, 15u, 15u};        // op 15 / len 0: advances nothing, emits nothing
      const uint64_t xa = cb0 + x;
      if (x < oe_rel) {
        if (xa + 4 <= (uint64_t)rv.n_ops) {
          uint4 v = cig4[xa >> 2];
          w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) if (xa + k < (uint64_t)rv.n_ops) w[k] = rv.cigar[xa + k];
        }
      }
      if (cb < ob_rel || cb + 256 > oe_rel) {  // edge chunk (wave-uniform): mask ops outside [ob,oe)
#pragma unroll
        for (int k = 0; k < 4; ++k) if (x + k < ob_rel || x + k >= oe_rel) w[k] = 15u;
      }
      uint32_t ar[4], aq[4];
      uint32_t em = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t op = w[k] & 15u, len = w[k] >> 4;
        ar[k] = len & bfe_mask(T::REF, op);
        aq[k] = len & bfe_mask(T::QRY, op);
        const uint32_t e = (bfe_mask(EMIT_MASK, op) & (uint32_t)-(int)(len >= (uint32_t)min_svlen)) |
                           (bfe_mask(T::BAD, op) & (uint32_t)-(int)(len != 0));
        em |= (e & 1u) << k;
      }
      const uint32_t sum_r = ar[0] + ar[1] + ar[2] + ar[3];
      const uint32_t sum_q = aq[0] + aq[1] + aq[2] + aq[3];
      const uint32_t incl_r = wave_incl_scan(sum_r), incl_q = wave_incl_scan(sum_q);
      const uint32_t excl_r = incl_r - sum_r, excl_q = incl_q - sum_q;

      // ---- slow path: one scalar iteration per candidate op, in (lane, sub) = op order --------------
      uint64_t anym = __ballot(em != 0);
      while (anym) {
        const uint32_t l = (uint32_t)__builtin_ctzll(anym);
        anym &= anym - 1;
        const uint32_t eml = rdlane(em, l);
        const uint32_t wl[4] = {rdlane(w[0], l), rdlane(w[1], l), rdlane(w[2], l), rdlane(w[3], l)};
        const uint32_t arl[4] = {rdlane(ar[0], l), rdlane(ar[1], l), rdlane(ar[2], l), rdlane(ar[3], l)};
        const uint32_t aql[4] = {rdlane(aq[0], l), rdlane(aq[1], l), rdlane(aq[2], l), rdlane(aq[3], l)};
        uint32_t px_r = rdlane(excl_r, l), px_q = rdlane(excl_q, l);  // in-chunk exclusive prefix at (l,0)
        for (uint32_t sub = 0; sub < 4; ++sub) {
          if (eml & (1u << sub)) {
            const uint32_t xo = cb + 4u * l + sub;
            const uint32_t op = wl[sub] & 15u, len = wl[sub] >> 4;
            // record of op xo: last lane-held start <= xo (starts ascend with the lane index)
            const uint32_t rloc = (uint32_t)__popcll(__ballot(rv_ok && s_rel <= xo)) - 1u;
            const uint32_t s_r = rdlane(s_rel, rloc);
            uint32_t a_r, a_q;  // advances from the record start to xo
            if (rloc + 1u == jprev) {        // record was already open at the chunk edge
              a_r = pbase_r + px_r - open_r;
              a_q = pbase_q + px_q - open_q;
            } else {                         // record starts inside this chunk
              const uint32_t ls = (s_r - cb) >> 2, ss = (s_r - cb) & 3u;
              uint32_t b_r = rdlane(excl_r, ls), b_q = rdlane(excl_q, ls);
              if (ss > 0) { b_r += rdlane(ar[0], ls); b_q += rdlane(aq[0], ls); }
              if (ss > 1) { b_r += rdlane(ar[1], ls); b_q += rdlane(aq[1], ls); }
              if (ss > 2) { b_r += rdlane(ar[2], ls); b_q += rdlane(aq[2], ls); }
              a_r = px_r - b_r;
              a_q = px_q - b_q;
            }
            const uint32_t rec = rbase + rloc;
            const uint32_t fl = rv.flag[rec], mq = rv.mapq[rec];
            uint32_t hapbits;
            if (CLS == 0) hapbits = (mq >= (uint32_t)min_mapq) ? ((fl >> 2) & 3u) : 0u;               // H:392-394
            else if (CLS == 1) hapbits = (mq >= (uint32_t)min_mapq) ? 1u : 0u;                        // RS:120
            else hapbits = (!(fl & (VSV_F_UNMAPPED | VSV_F_SECONDARY)) && mq >= (uint32_t)min_mapq) ? 1u : 0u;
            if (hapbits) {
              if (op != 1u && op != 2u) {  // N/=/X on the contig table: assert offset_ref==reference_end (H:396)
                if (lane == 0) atomicOr(&ec.ctr->err, ERRB_REFEND);
              } else {
                uint32_t hc = 0;
                if (T::HC) {
                  const uint32_t first = rv.cigar[cb0 + s_r];
                  hc = ((first & 15u) == 5u) ? (first >> 4) : 0u;                                     // H:63-65
                }
                const uint32_t nemit = (hapbits == 3u) ? 2u : 1u;
                uint32_t slot = 0;
                if (lane == 0) slot = atomicAdd(&ec.ctr->n_pool, nemit);
                slot = __builtin_amdgcn_readfirstlane(slot);
                if (lane < (int)nemit && slot + (uint32_t)lane < ec.cap) {
                  vsv_sig s;
                  s.pos = (int32_t)(rdlane(rpos, rloc) + a_r);
                  s.svlen = (int32_t)len;
                  s.q_start = (int32_t)(a_q + hc);
                  s.q_end = (CLS == 1) ? 0 : s.q_start + (op == 2u ? 1 : (int32_t)len);
                  s.rec = rec;
                  s.rec2 = 0xFFFFFFFFu;
                  const uint32_t hp2 = (CLS == 0) ? ((hapbits == 3u) ? (uint32_t)lane : (hapbits >> 1)) : 0u;
                  s.meta = (op == 2u ? VSV_M_DEL : 0u) | (hp2 ? VSV_M_HP2 : 0u);
                  s.tid = rv.tid[rec];
                  ec.pool[slot + lane] = s;
                  ec.pool_key[slot + lane] = ((uint64_t)(uint32_t)part << 32) | (uint64_t)(ord + (uint32_t)lane);
                }
                ord += nemit;
              }
            }
          }
          px_r += arl[sub];
          px_q += aql[sub];
        }
      }

      // ---- scalar bookkeeping: which record is open at the end of this chunk ----------------------
      const uint32_t j = (uint32_t)__popcll(__ballot(rv_ok && s_rel < cb + 256u));
      if (j > jprev) {
        const uint32_t s = rdlane(s_rel, j - 1u);
        const uint32_t ls = (s - cb) >> 2, ss = (s - cb) & 3u;
        uint32_t b_r = rdlane(excl_r, ls), b_q = rdlane(excl_q, ls);
        if (ss > 0) { b_r += rdlane(ar[0], ls); b_q += rdlane(aq[0], ls); }
        if (ss > 1) { b_r += rdlane(ar[1], ls); b_q += rdlane(aq[1], ls); }
        if (ss > 2) { b_r += rdlane(ar[2], ls); b_q += rdlane(aq[2], ls); }
        open_r = pbase_r + b_r;
        open_q = pbase_q + b_q;
        jprev = j;
      }
      pbase_r += rdlane(incl_r, 63);
      pbase_q += rdlane(incl_q, 63);
    }
  }
  if (lane == 0) part_count[part] = ord;
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

// ---- ordered placement: raw[part_off[part] + ordinal] = pool[e] ----------------------------------
__global__ __launch_bounds__(256) void place_raw(const vsv_sig* __restrict__ pool, const uint64_t* __restrict__ pool_key,
                                                 const uint32_t* __restrict__ part_off, vsv_sig* __restrict__ raw,
                                                 uint32_t cap, Counters* ctr) {
  const uint32_t n = min(ctr->n_pool, cap);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    ctr->n_raw = n;
    if (ctr->n_pool > cap) atomicOr(&ctr->err, ERRB_CAPACITY);
  }
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const uint64_t k = pool_key[e];
    const uint32_t dst = part_off[(uint32_t)(k >> 32)] + (uint32_t)k;
    if (dst < cap) raw[dst] = pool[e];
  }
}

}  // namespace

void vsv_scan_u32_exclusive(hipStream_t st, const uint32_t* in, int n, uint32_t* out, uint32_t* tmp) {
  if (n <= 0) return;
  const int tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  scan_tile_sums<<<tiles, 256, 0, st>>>(in, n, tmp);
  scan_sums_inplace<<<1, 1024, 0, st>>>(tmp, tiles);
  scan_tile_apply<<<tiles, 256, 0, st>>>(in, n, tmp, out);
}

int vsv_cigar_parts(int64_t n_ops, int ops_per_part) { return (int)((n_ops + ops_per_part - 1) / ops_per_part); }

void vsv_launch_cigar_scan(hipStream_t st, const RecView& rv, const vsv_params& p, uint32_t* part_rb, int n_parts,
                           int ops_per_part, vsv_sig* pool, uint64_t* pool_key, uint32_t cap, uint32_t* part_count,
                           uint32_t* part_off, uint32_t* scan_tmp, vsv_sig* raw, Counters* ctr, hipEvent_t ev0,
                           hipEvent_t ev1) {
  if (n_parts <= 0) return;
  partition_kernel<<<(n_parts + 1 + 255) / 256, 256, 0, st>>>(rv.cigar_off, rv.n_records, part_rb, n_parts, ops_per_part);
  EmitCtx ec{pool, pool_key, cap, ctr};
  const int waves_per_block = 4;
  const int grid = (n_parts + waves_per_block - 1) / waves_per_block;
  if (ev0) hipEventRecord(ev0, st);
  if (p.dtype == VSV_DTYPE_READS)
    cigar_scan_emit<1><<<grid, 256, 0, st>>>(rv, part_rb, n_parts, p.min_svlen, p.min_cigar_mapq, ec, part_count);
  else if (p.dtype == VSV_DTYPE_SVIM)
    cigar_scan_emit<2><<<grid, 256, 0, st>>>(rv, part_rb, n_parts, p.min_svlen, p.min_cigar_mapq, ec, part_count);
  else
    cigar_scan_emit<0><<<grid, 256, 0, st>>>(rv, part_rb, n_parts, p.min_svlen, p.min_cigar_mapq, ec, part_count);
  if (ev1) hipEventRecord(ev1, st);
  vsv_scan_u32_exclusive(st, part_count, n_parts, part_off, scan_tmp);
  place_raw<<<256, 256, 0, st>>>(pool, pool_key, part_off, raw, cap, ctr);
}
