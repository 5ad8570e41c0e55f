// inflate.hip — BGZF members inflated on the GPU ("multithreaded host BGZF reader ... or GPU inflate").
//
// A BGZF file is a sequence of independent raw-deflate streams (RFC 1951) of at most 64 KiB output each, so the members are
// the parallel axis: ONE LANE decodes ONE MEMBER, front to back, exactly like a scalar inflate — bit reader over the
// compressed bytes (dword refills), stored / fixed / dynamic blocks, canonical Huffman decoding by code length (count[] /
// symbol[] tables in LDS, interleaved by lane), LZ77 copies out of the lane's own output. There is no cross-lane cooperation and no shared
// state; lanes of a wave diverge between literals and matches, and the throughput comes from having tens of thousands of
// members in flight. Every loop is bounded by the member's input and output sizes; a malformed stream sets the member's
// status and the lane stops.
#include "vsv_device.h"

namespace {

constexpr int MAXBITS = 15, MAXLCODES = 286, MAXDCODES = 30, FIXLCODES = 288;

// Per-lane Huffman tables live in LDS, interleaved by lane (entry i of lane l at [i][l]): a decode step is a chain of
// dependent table reads, and LDS answers in tens of cycles where private (scratch) memory takes hundreds.
struct HuffL {              // literal/length code (also used for the 19-symbol code-length code)
  uint16_t count[MAXBITS + 1][64];
  uint16_t symbol[FIXLCODES][64];
};
struct HuffD {
  uint16_t count[MAXBITS + 1][64];
  uint16_t symbol[MAXDCODES][64];
};
struct LaneLds {
  HuffL lc;
  HuffD dc;
  uint8_t lengths[MAXLCODES + MAXDCODES + 4][64];
};

struct Bits {
  const uint8_t* p; const uint8_t* end; uint64_t buf; int cnt; bool over;
  uint32_t ahead; bool have_ahead;          // the next aligned dword, loaded one refill early so its latency overlaps the decode
  __device__ __forceinline__ void refill() {
    // bytes until the pointer is 4-byte aligned, then whole dwords
    while (cnt <= 56 && p < end && ((uintptr_t)p & 3u) && !have_ahead) { buf |= (uint64_t)(*p++) << cnt; cnt += 8; }
    if (cnt <= 32 && (have_ahead || p + 4 <= end) && (((uintptr_t)p & 3u) == 0)) {
      const uint32_t w = have_ahead ? ahead : *reinterpret_cast<const uint32_t*>(p);
      buf |= (uint64_t)w << cnt; p += 4; cnt += 32;
      have_ahead = p + 4 <= end;
      if (have_ahead) ahead = *reinterpret_cast<const uint32_t*>(p);
    }
    while (cnt <= 56 && p < end && (p + 4 > end)) { buf |= (uint64_t)(*p++) << cnt; cnt += 8; }
  }
  __device__ __forceinline__ uint32_t get(int n) {          // n <= 16
    if (cnt < n) { refill(); if (cnt < n) { over = true; return 0; } }
    const uint32_t v = (uint32_t)(buf & ((1ull << n) - 1ull));
    buf >>= n; cnt -= n;
    return v;
  }
};

// canonical Huffman decode, one bit per step (codes are at most 15 bits): the code is in range at length `len` when
// code - count[len] < first
template <typename H>
__device__ int decode(Bits& b, const H& h, int lane) {
  if (b.cnt < MAXBITS) b.refill();
  int code = 0, first = 0, index = 0;
  for (int len = 1; len <= MAXBITS; ++len) {
    if (b.cnt < 1) { b.over = true; return -1; }
    code |= (int)(b.buf & 1ull);
    b.buf >>= 1; b.cnt -= 1;
    const int count = h.count[len][lane];
    if (code - count < first) return h.symbol[index + (code - first)][lane];
    index += count; first += count; first <<= 1; code <<= 1;
  }
  return -2;   // ran out of codes
}

// The per-length counts of a code, copied out of LDS into registers once per block: the bit-serial decode loop below is
// fully unrolled, so the 15 counts are plain VGPRs and only the final symbol lookup touches LDS.
struct Counts { uint16_t c[MAXBITS + 1]; };
template <typename H>
__device__ __forceinline__ Counts load_counts(const H& h, int lane) {
  Counts k;
#pragma unroll
  for (int len = 0; len <= MAXBITS; ++len) k.c[len] = h.count[len][lane];
  return k;
}
template <typename H>
__device__ __forceinline__ int decode_reg(Bits& b, const Counts& k, const H& h, int lane) {
  if (b.cnt < MAXBITS) b.refill();
  int code = 0, first = 0, index = 0;
#pragma unroll
  for (int len = 1; len <= MAXBITS; ++len) {
    if (b.cnt < 1) { b.over = true; return -1; }
    code |= (int)(b.buf & 1ull);
    b.buf >>= 1; b.cnt -= 1;
    const int count = k.c[len];
    if (code - count < first) return h.symbol[index + (code - first)][lane];
    index += count; first += count; first <<= 1; code <<= 1;
  }
  return -2;
}

// builds count[] / symbol[] from the lane's code lengths (LDS, starting at entry `from`); returns 0 for a complete code,
// < 0 over-subscribed, > 0 incomplete
template <typename H>
__device__ int construct(H& h, const uint8_t (*length)[64], int from, int n, int lane) {
  for (int len = 0; len <= MAXBITS; ++len) h.count[len][lane] = 0;
  for (int s = 0; s < n; ++s) h.count[length[from + s][lane]][lane]++;
  if (h.count[0][lane] == n) return 0;
  int left = 1;
  for (int len = 1; len <= MAXBITS; ++len) { left <<= 1; left -= h.count[len][lane]; if (left < 0) return left; }
  uint16_t offs[MAXBITS + 1];
  offs[1] = 0;
  for (int len = 1; len < MAXBITS; ++len) offs[len + 1] = offs[len] + h.count[len][lane];
  for (int s = 0; s < n; ++s) { const int l = length[from + s][lane]; if (l != 0) h.symbol[offs[l]++][lane] = (uint16_t)s; }
  return left;
}

__constant__ uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// literal/length + distance codes of one block -> output; returns 0 at end-of-block, < 0 on error
__device__ int codes(Bits& b, const HuffL& lc, const HuffD& dc, int lane, uint8_t* out, uint32_t& o, uint32_t cap) {
  const Counts kl = load_counts(lc, lane), kd = load_counts(dc, lane);
  for (;;) {
    int sym = decode_reg(b, kl, lc, lane);
    if (sym < 0) return sym;
    if (sym < 256) {
      if (o >= cap) return -3;
      out[o++] = (uint8_t)sym;
    } else if (sym == 256) {
      return 0;
    } else {
      sym -= 257;
      if (sym >= 29) return -4;
      const uint32_t len = LBASE[sym] + b.get(LEXT[sym]);
      const int ds = decode_reg(b, kd, dc, lane);
      if (ds < 0) return ds;
      if (ds >= 30) return -5;
      const uint32_t dist = DBASE[ds] + b.get(DEXT[ds]);
      if (b.over) return -1;
      if (dist > o) return -6;
      if (o + len > cap) return -3;
      uint32_t k = 0;
      // a copy step is a round trip to L2 (the bytes were just written, the L1 does not hold them): keep 16 or 4 loads in
      // flight per round trip when the distance leaves that many source bytes untouched by the copy itself
      if (dist >= 16)
        for (; k + 16 <= len; k += 16) {
          uint8_t c[16];
#pragma unroll
          for (int t = 0; t < 16; ++t) c[t] = out[o - dist + t];
#pragma unroll
          for (int t = 0; t < 16; ++t) out[o + t] = c[t];
          o += 16;
        }
      if (dist >= 4)
        for (; k + 4 <= len; k += 4) {
          const uint8_t c0 = out[o - dist], c1 = out[o - dist + 1], c2 = out[o - dist + 2], c3 = out[o - dist + 3];
          out[o] = c0; out[o + 1] = c1; out[o + 2] = c2; out[o + 3] = c3; o += 4;
        }
      for (; k < len; ++k) { out[o] = out[o - dist]; ++o; }
    }
  }
}

__global__ __launch_bounds__(64) void bgzf_inflate(const uint8_t* __restrict__ comp, const uint64_t* __restrict__ comp_off,
                                                   const uint64_t* __restrict__ out_off, int64_t n, uint8_t* __restrict__ out,
                                                   int32_t* __restrict__ status) {
  __shared__ LaneLds L;
  const int lane = threadIdx.x;
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n) return;
  Bits b{comp + comp_off[m], comp + comp_off[m + 1], 0, 0, false, 0u, false};
  uint8_t* dst = out + out_off[m];
  const uint32_t cap = (uint32_t)(out_off[m + 1] - out_off[m]);
  uint32_t o = 0;
  int err = 0;
  for (int guard = 0; guard < 70000 && !err; ++guard) {        // a member holds at most 64 KiB: far fewer blocks than this
    const uint32_t last = b.get(1), type = b.get(2);
    if (b.over) { err = -1; break; }
    if (type == 0) {                                            // stored
      b.buf >>= (b.cnt & 7); b.cnt -= (b.cnt & 7);
      const uint32_t len = b.get(16), nlen = b.get(16);
      if (b.over || (len ^ 0xFFFFu) != nlen) { err = -7; break; }
      if (o + len > cap) { err = -3; break; }
      for (uint32_t k = 0; k < len; ++k) { const uint32_t v = b.get(8); if (b.over) { err = -1; break; } dst[o++] = (uint8_t)v; }
    } else if (type == 1) {                                     // fixed codes
      int s = 0;
      for (; s < 144; ++s) L.lengths[s][lane] = 8;
      for (; s < 256; ++s) L.lengths[s][lane] = 9;
      for (; s < 280; ++s) L.lengths[s][lane] = 7;
      for (; s < FIXLCODES; ++s) L.lengths[s][lane] = 8;
      construct(L.lc, L.lengths, 0, FIXLCODES, lane);
      for (s = 0; s < MAXDCODES; ++s) L.lengths[s][lane] = 5;
      construct(L.dc, L.lengths, 0, MAXDCODES, lane);
      err = codes(b, L.lc, L.dc, lane, dst, o, cap);
    } else if (type == 2) {                                     // dynamic codes
      const int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
      if (b.over || nlen > MAXLCODES || ndist > MAXDCODES) { err = -8; break; }
      int idx = 0;
      for (; idx < ncode; ++idx) L.lengths[CLORDER[idx]][lane] = (uint8_t)b.get(3);
      for (; idx < 19; ++idx) L.lengths[CLORDER[idx]][lane] = 0;
      if (construct(L.lc, L.lengths, 0, 19, lane) != 0) { err = -9; break; }
      idx = 0;
      while (idx < nlen + ndist) {
        int sym = decode(b, L.lc, lane);
        if (sym < 0) { err = sym; break; }
        if (sym < 16) L.lengths[idx++][lane] = (uint8_t)sym;
        else {
          int len = 0, rep;
          if (sym == 16) { if (idx == 0) { err = -10; break; } len = L.lengths[idx - 1][lane]; rep = 3 + (int)b.get(2); }
          else if (sym == 17) rep = 3 + (int)b.get(3);
          else rep = 11 + (int)b.get(7);
          if (idx + rep > nlen + ndist) { err = -11; break; }
          while (rep--) L.lengths[idx++][lane] = (uint8_t)len;
        }
      }
      if (err) break;
      if (b.over) { err = -1; break; }
      if (L.lengths[256][lane] == 0) { err = -12; break; }
      int r = construct(L.lc, L.lengths, 0, nlen, lane);
      if (r < 0 || (r > 0 && nlen - L.lc.count[0][lane] != 1)) { err = -13; break; }
      r = construct(L.dc, L.lengths, nlen, ndist, lane);         // the distance lengths follow the literal/length lengths
      if (r < 0 || (r > 0 && ndist - L.dc.count[0][lane] != 1)) { err = -14; break; }
      err = codes(b, L.lc, L.dc, lane, dst, o, cap);
    } else { err = -15; break; }
    if (last) break;
  }
  if (!err && o != cap) err = -16;                              // ISIZE of the member must be met exactly
  status[m] = err;
}

}  // namespace

void vsv_launch_bgzf_inflate(hipStream_t st, const uint8_t* comp, const uint64_t* comp_off, const uint64_t* out_off, int64_t n, uint8_t* out,
                             int32_t* status) {
  if (n <= 0) return;
  bgzf_inflate<<<(int)((n + 63) / 64), 64, 0, st>>>(comp, comp_off, out_off, n, out, status);
}
