// inflate.hip — BGZF members inflated on the GPU ("multithreaded host BGZF reader ... or GPU inflate").
//
// A BGZF file is a sequence of independent raw-deflate streams (RFC 1951) of at most 64 KiB output each, so the members are
// the parallel axis: ONE LANE decodes ONE MEMBER, front to back, exactly like a scalar inflate — bit reader over the
// compressed bytes, stored / fixed / dynamic blocks, canonical Huffman decoding by code length (count[] / symbol[] tables in
// the lane's private memory), LZ77 copies out of the lane's own output. There is no cross-lane cooperation and no shared
// state; lanes of a wave diverge between literals and matches, and the throughput comes from having tens of thousands of
// members in flight. Every loop is bounded by the member's input and output sizes; a malformed stream sets the member's
// status and the lane stops.
#include "vsv_device.h"

namespace {

constexpr int MAXBITS = 15, MAXLCODES = 286, MAXDCODES = 30, FIXLCODES = 288;

struct Huff { uint16_t count[MAXBITS + 1]; uint16_t symbol[FIXLCODES]; };

struct Bits {
  const uint8_t* p; const uint8_t* end; uint64_t buf; int cnt; bool over;
  __device__ __forceinline__ void refill() {
    while (cnt <= 56 && p < end) { buf |= (uint64_t)(*p++) << cnt; cnt += 8; }
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
__device__ int decode(Bits& b, const Huff& h) {
  int code = 0, first = 0, index = 0;
  for (int len = 1; len <= MAXBITS; ++len) {
    code |= (int)b.get(1);
    if (b.over) return -1;
    const int count = h.count[len];
    if (code - count < first) return h.symbol[index + (code - first)];
    index += count; first += count; first <<= 1; code <<= 1;
  }
  return -2;   // ran out of codes
}

// builds count[] / symbol[] from code lengths; returns 0 for a complete code, < 0 over-subscribed, > 0 incomplete
__device__ int construct(Huff& h, const uint8_t* length, int n) {
  for (int len = 0; len <= MAXBITS; ++len) h.count[len] = 0;
  for (int s = 0; s < n; ++s) h.count[length[s]]++;
  if (h.count[0] == n) return 0;
  int left = 1;
  for (int len = 1; len <= MAXBITS; ++len) { left <<= 1; left -= h.count[len]; if (left < 0) return left; }
  uint16_t offs[MAXBITS + 1];
  offs[1] = 0;
  for (int len = 1; len < MAXBITS; ++len) offs[len + 1] = offs[len] + h.count[len];
  for (int s = 0; s < n; ++s) if (length[s] != 0) h.symbol[offs[length[s]]++] = (uint16_t)s;
  return left;
}

__constant__ uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// literal/length + distance codes of one block -> output; returns 0 at end-of-block, < 0 on error
__device__ int codes(Bits& b, const Huff& lc, const Huff& dc, uint8_t* out, uint32_t& o, uint32_t cap) {
  for (;;) {
    int sym = decode(b, lc);
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
      const int ds = decode(b, dc);
      if (ds < 0) return ds;
      if (ds >= 30) return -5;
      const uint32_t dist = DBASE[ds] + b.get(DEXT[ds]);
      if (b.over) return -1;
      if (dist > o) return -6;
      if (o + len > cap) return -3;
      for (uint32_t k = 0; k < len; ++k) { out[o] = out[o - dist]; ++o; }
    }
  }
}

__global__ __launch_bounds__(64) void bgzf_inflate(const uint8_t* __restrict__ comp, const uint64_t* __restrict__ comp_off,
                                                   const uint64_t* __restrict__ out_off, int64_t n, uint8_t* __restrict__ out,
                                                   int32_t* __restrict__ status) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n) return;
  Bits b{comp + comp_off[m], comp + comp_off[m + 1], 0, 0, false};
  uint8_t* dst = out + out_off[m];
  const uint32_t cap = (uint32_t)(out_off[m + 1] - out_off[m]);
  uint32_t o = 0;
  int err = 0;
  Huff lc, dc;
  uint8_t lengths[MAXLCODES + MAXDCODES];
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
      for (; s < 144; ++s) lengths[s] = 8;
      for (; s < 256; ++s) lengths[s] = 9;
      for (; s < 280; ++s) lengths[s] = 7;
      for (; s < FIXLCODES; ++s) lengths[s] = 8;
      construct(lc, lengths, FIXLCODES);
      for (s = 0; s < MAXDCODES; ++s) lengths[s] = 5;
      construct(dc, lengths, MAXDCODES);
      err = codes(b, lc, dc, dst, o, cap);
    } else if (type == 2) {                                     // dynamic codes
      const int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
      if (b.over || nlen > MAXLCODES || ndist > MAXDCODES) { err = -8; break; }
      int idx = 0;
      for (; idx < ncode; ++idx) lengths[CLORDER[idx]] = (uint8_t)b.get(3);
      for (; idx < 19; ++idx) lengths[CLORDER[idx]] = 0;
      if (construct(lc, lengths, 19) != 0) { err = -9; break; }
      idx = 0;
      while (idx < nlen + ndist) {
        int sym = decode(b, lc);
        if (sym < 0) { err = sym; break; }
        if (sym < 16) lengths[idx++] = (uint8_t)sym;
        else {
          int len = 0, rep;
          if (sym == 16) { if (idx == 0) { err = -10; break; } len = lengths[idx - 1]; rep = 3 + (int)b.get(2); }
          else if (sym == 17) rep = 3 + (int)b.get(3);
          else rep = 11 + (int)b.get(7);
          if (idx + rep > nlen + ndist) { err = -11; break; }
          while (rep--) lengths[idx++] = (uint8_t)len;
        }
      }
      if (err) break;
      if (b.over) { err = -1; break; }
      if (lengths[256] == 0) { err = -12; break; }
      int r = construct(lc, lengths, nlen);
      if (r < 0 || (r > 0 && nlen - lc.count[0] != 1)) { err = -13; break; }
      // the distance lengths follow the literal/length lengths: move them to the front before building the second table
      uint8_t dl[MAXDCODES];
      for (int s = 0; s < ndist; ++s) dl[s] = lengths[nlen + s];
      r = construct(dc, dl, ndist);
      if (r < 0 || (r > 0 && ndist - dc.count[0] != 1)) { err = -14; break; }
      err = codes(b, lc, dc, dst, o, cap);
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
