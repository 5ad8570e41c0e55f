// inflate.hip — BGZF members inflated on the GPU ("multithreaded host BGZF reader ... or GPU inflate").
//
// A BGZF file is a sequence of independent raw-deflate streams (RFC 1951) of at most 64 KiB output each, so the members are
// the parallel axis: ONE WAVEFRONT decodes ONE MEMBER. The decoder itself is sequential (bit reader, stored / fixed / dynamic
// blocks, canonical Huffman decoding by code length), and it is written wave-uniform: the workgroup is a single wave and the
// member index is the block index, so the compiler keeps the bit buffer, the code counts and the control flow on the scalar
// unit and fetches the compressed bytes through the scalar cache — none of it waits on the vector memory counter that the
// output stores keep busy. The 64 lanes do the byte work: literals collect one per lane and leave as one coalesced store;
// an LZ77 match is copied min(distance, 64) bytes per step. Tables (count[] / symbol[]) sit in LDS, ~1.4 KB per wave, so a
// CU holds as many members in flight as it has wave slots. (A first version with one LANE per member was bit-exact too but
// ran at 1 GB/s: 64 divergent decoders in lockstep, every refill waiting behind the latest byte store.)
// Every loop is bounded by the member's input and output sizes; a malformed stream sets the member's status and the wave stops.
#include "vsv_device.h"

namespace {

constexpr int MAXBITS = 15, MAXLCODES = 286, MAXDCODES = 30, FIXLCODES = 288;

constexpr int LFAST = 11, DFAST = 10;  // bits resolved by one table lookup; longer codes fall back to the bit-serial decode
struct Tables {
  uint16_t lcount[MAXBITS + 1], lsymbol[FIXLCODES];
  uint16_t dcount[MAXBITS + 1], dsymbol[MAXDCODES];
  uint8_t lengths[MAXLCODES + MAXDCODES + 2];
  // [next LFAST / DFAST stream bits] -> everything the body of a block needs from the symbol in one dword (lit_entry / dist_entry
  // below) | code length in bits 0-3; 0 in those bits = a longer code (bit-serial decode)
  uint16_t lfast[1 << LFAST];
  uint16_t dfast[1 << DFAST];
};

// one dword of the compressed stream through the scalar cache (lgkmcnt): it does not queue behind the output stores
__device__ __forceinline__ uint32_t sload_dword(const uint32_t* p) {
  const uint64_t a = (uint64_t)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  const uint32_t* q = (const uint32_t*)(uintptr_t)(((uint64_t)hi << 32) | lo);
  uint32_t v;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(q) : "memory");
  return v;
}

struct Bits {                       // all members wave-uniform
  const uint32_t* p; const uint32_t* end; uint64_t buf; int cnt; int skip; bool over;
  const uint32_t* w0; uint32_t end_bit;   // the member's first (aligned) dword and its end, in bits from there
  // the stream is read as aligned dwords; `skip` leading bytes of the first dword belong to the previous member
  __device__ __forceinline__ void refill() {
    if (cnt <= 32 && p != end) {
      uint64_t v = sload_dword(p); ++p;
      int nb = 32;
      if (skip) { v >>= 8 * skip; nb -= 8 * skip; skip = 0; }
      buf |= v << cnt; cnt += nb;
    }
  }
  // the same once the first dword is in (skip == 0): the block bodies
  __device__ __forceinline__ void refill_body() {
    if (cnt <= 32) {
      if (p != end) { const uint64_t v = sload_dword(p); ++p; buf |= v << cnt; cnt += 32; }
    }
  }
  __device__ __forceinline__ uint32_t get(int n) {          // n <= 16
    if (cnt < n) { refill(); if (cnt < n) { refill(); if (cnt < n) { over = true; return 0; } } }
    const uint32_t v = (uint32_t)(buf & ((1ull << n) - 1ull));
    buf >>= n; cnt -= n;
    return v;
  }
};

// canonical Huffman decode, one bit per step (codes are at most 15 bits): the code is in range at length `len` when
// code - count[len] < first
__device__ __forceinline__ int decode(Bits& b, const uint16_t* count, const uint16_t* symbol) {
  if (b.cnt < MAXBITS) { b.refill(); if (b.cnt < MAXBITS) b.refill(); }
  int code = 0, first = 0, index = 0;
  for (int len = 1; len <= MAXBITS; ++len) {
    if (b.cnt < 1) { b.over = true; return -1; }
    code |= (int)(b.buf & 1ull);
    b.buf >>= 1; b.cnt -= 1;
    const int c = __builtin_amdgcn_readfirstlane((int)count[len]);
    if (code - c < first) return __builtin_amdgcn_readfirstlane((int)symbol[index + (code - first)]);
    index += c; first += c; first <<= 1; code <<= 1;
  }
  return -2;   // ran out of codes
}

// Same decode with the 15 per-length counts held in (scalar) registers: the loop is fully unrolled, only the final symbol
// lookup reads LDS. Used for the two codes of a block's body; the 19-symbol code-length code uses the LDS form above.
struct Counts { int c[MAXBITS + 1]; };
__device__ __forceinline__ Counts load_counts(const uint16_t* count) {
  Counts k;
#pragma unroll
  for (int len = 0; len <= MAXBITS; ++len) k.c[len] = __builtin_amdgcn_readfirstlane((int)count[len]);   // LDS -> SGPR
  return k;
}
__device__ __forceinline__ int decode_reg(Bits& b, const Counts& k, const uint16_t* symbol) {
  if (b.cnt < MAXBITS) { b.refill(); if (b.cnt < MAXBITS) b.refill(); }
  int code = 0, first = 0, index = 0;
#pragma unroll
  for (int len = 1; len <= MAXBITS; ++len) {
    if (b.cnt < 1) { b.over = true; return -1; }
    code |= (int)(b.buf & 1ull);
    b.buf >>= 1; b.cnt -= 1;
    const int c = k.c[len];
    if (code - c < first) return __builtin_amdgcn_readfirstlane((int)symbol[index + (code - first)]);
    index += c; first += c; first <<= 1; code <<= 1;
  }
  return -2;
}

// builds count[] / symbol[] from code lengths; returns 0 for a complete code, < 0 over-subscribed, > 0 incomplete.
// Executed by one lane (the tables are wave-shared LDS).
__device__ __forceinline__ int construct(uint16_t* count, uint16_t* symbol, const uint8_t* length, int n) {
  for (int len = 0; len <= MAXBITS; ++len) count[len] = 0;
  for (int s = 0; s < n; ++s) count[length[s]]++;
  if (count[0] == n) return 0;
  int left = 1;
  for (int len = 1; len <= MAXBITS; ++len) { left <<= 1; left -= count[len]; if (left < 0) return left; }
  uint16_t offs[MAXBITS + 1];
  offs[1] = 0;
  for (int len = 1; len < MAXBITS; ++len) offs[len + 1] = offs[len] + count[len];
  for (int s = 0; s < n; ++s) { const int l = length[s]; if (l != 0) symbol[offs[l]++] = (uint16_t)s; }
  return left;
}
__device__ __forceinline__ int construct_by_lane0(uint16_t* count, uint16_t* symbol, const uint8_t* length, int n, int lane) {
  int r = 0;
  if (lane == 0) r = construct(count, symbol, length, n);
  __builtin_amdgcn_wave_barrier();
  return __builtin_amdgcn_readfirstlane(r);
}

// Fast table of a code whose count[] / symbol[] are built: entry [v] for every LFAST-bit value v of the stream whose low bits
// are a complete code of length <= BITS (stream bits arrive LSB first, code bits MSB first, hence the bit reversal).
// The 64 lanes share the work: lane handles the sorted symbols lane, lane + 64, ...
template <int BITS, typename ENTRY>
__device__ __forceinline__ void build_fast(const uint16_t* count, const uint16_t* symbol, uint16_t* fast, int lane, ENTRY entry) {
  for (int i = lane; i < (1 << BITS); i += 64) fast[i] = 0;
  __builtin_amdgcn_wave_barrier();
  int code = 0, index = 0;                           // canonical first code / first sorted index of the current length
  for (int len = 1; len <= BITS; ++len) {
    const int c = __builtin_amdgcn_readfirstlane((int)count[len]);
    for (int r = lane; r < c; r += 64) {
      const uint32_t rev = __builtin_bitreverse32((uint32_t)(code + r)) >> (32 - len);
      const uint16_t e = (uint16_t)(entry((int)symbol[index + r]) | (uint32_t)len);
      for (uint32_t hi = 0; hi < (1u << (BITS - len)); ++hi) fast[rev | (hi << len)] = e;
    }
    code = (code + c) << 1;
    index += c;
  }
  __builtin_amdgcn_wave_barrier();
}

// base value and extra bits of the length symbols 257..285 (sym = symbol - 257) and the distance symbols 0..29 (RFC 1951 3.2.5),
// computed: a table in constant memory is a vector load and a round trip to the cache per lookup, twice per match
__device__ __forceinline__ void length_code(int sym, uint32_t& base, int& ext) {
  if (sym < 8) { base = 3u + (uint32_t)sym; ext = 0; }
  else if (sym == 28) { base = 258u; ext = 0; }
  else { ext = (sym - 4) >> 2; base = 3u + ((4u + ((uint32_t)sym & 3u)) << ext); }
}
__device__ __forceinline__ void dist_code(int ds, uint32_t& base, int& ext) {
  if (ds < 4) { base = 1u + (uint32_t)ds; ext = 0; }
  else { ext = (ds - 2) >> 1; base = 1u + ((2u + ((uint32_t)ds & 1u)) << ext); }
}
__constant__ uint8_t CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Output side of one member: literals wait one per lane and are stored 64 at a time. Every byte also goes to a ring of the last
// RING bytes in LDS: an LZ77 match that starts inside the ring (the usual case: the previous records of a BAM block) copies from
// LDS at LDS latency — a store to global memory followed by a load of the same bytes is a round trip to L2 of a microsecond, and a
// member holds thousands of matches (10 ms per member, whatever the occupancy, when every match took that trip). Matches from
// further back read global memory as before, after a wait for the stores.
constexpr uint32_t RING = 8192;
struct Out {
  uint8_t* dst; uint32_t o, cap; uint32_t npend; uint32_t lit; int lane; uint8_t* ring;   // o counts flushed bytes; lit = this lane's pending byte
  bool full;                        // more output than the member's ISIZE: nothing is written any more, the block body ends with -3
  // the pending literals leave — unless they do not fit the member's ISIZE
  __device__ __forceinline__ void flush() {
    // (one lane predicate, no early exit: a uniform branch around the masked stores makes the compiler treat the whole bit
    // reader as divergent, i.e. moves it from the scalar to the vector unit)
    const bool fits = o + npend <= cap;
    if (!fits) full = true;
    if ((uint32_t)lane < (fits ? npend : 0u)) { dst[o + lane] = (uint8_t)lit; ring[(o + lane) & (RING - 1u)] = (uint8_t)lit; }
    o += npend; npend = 0;
  }
  __device__ __forceinline__ void literal(uint32_t sym) {   // (the ISIZE test waits for the flush: at most 64 pending bytes)
    if ((uint32_t)lane == npend) lit = sym;
    if (++npend == 64) flush();
  }
  // copy len bytes from distance dist behind the write position; sources may overlap the destination (dist < len)
  __device__ __forceinline__ bool match(uint32_t len, uint32_t dist) {
    flush();
    if (full || dist > o || o + len > cap) return false;
    __builtin_amdgcn_wave_barrier();
    if (dist + 64u <= RING) {
      // every source byte is in the ring (the ring holds the last RING bytes; a step writes at most 64 ahead). An overlapping match
      // (dist < len) repeats its first dist bytes: all lanes read from that period, no step waits for the one before.
      if (dist >= len) {
        for (uint32_t base = 0; base < len; base += 64u) {
          const uint32_t i = base + (uint32_t)lane;
          if (i < len) { const uint8_t v = ring[(o - dist + i) & (RING - 1u)]; dst[o + i] = v; ring[(o + i) & (RING - 1u)] = v; }
        }
      } else {
        uint32_t src = (uint32_t)lane % dist;                 // position of byte `i` in the period; advances by 64 mod dist per step
        const uint32_t adv = 64u % dist;
        for (uint32_t base = 0; base < len; base += 64u) {
          const uint32_t i = base + (uint32_t)lane;
          if (i < len) dst[o + i] = ring[(o - dist + src) & (RING - 1u)];
          src += adv; if (src >= dist) src -= dist;
        }
        __builtin_amdgcn_wave_barrier();                      // the ring receives the run after all reads of its period are done
        src = (uint32_t)lane % dist;
        for (uint32_t base = 0; base < len; base += 64u) {
          const uint32_t i = base + (uint32_t)lane;
          if (i < len) ring[(o + i) & (RING - 1u)] = ring[(o - dist + src) & (RING - 1u)];
          src += adv; if (src >= dist) src -= dist;
        }
      }
      __builtin_amdgcn_wave_barrier();
      o += len;
      return true;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the bytes stored so far are the bytes about to be read
    const uint32_t step = dist < 64u ? dist : 64u;                      // (dist > RING - 64 here: one step per 64 bytes)
    for (uint32_t base = 0; base < len; base += step) {
      const uint32_t nb = (len - base) < step ? (len - base) : step;
      if ((uint32_t)lane < nb) { const uint8_t v = dst[o + base + lane - dist]; dst[o + base + lane] = v; ring[(o + base + lane) & (RING - 1u)] = v; }
    }
    o += len;
    return true;
  }
};

// What the block body needs from a literal/length symbol, in one dword (bits 0-3 are the code length, added by build_fast):
//   bits 4-5  kind: 0 literal, 1 end of block, 2 length, 3 not a symbol of the alphabet (286, 287)
//   literal:  bits 8-15 the byte          length: bits 8-16 base length, bits 20-23 number of extra bits
// and from a distance symbol: bits 4-7 number of extra bits, bits 8-23 base distance, bit 24 not a distance (30, 31)
enum { K_LIT = 0, K_END = 1, K_LEN = 2, K_BAD = 3 };
__device__ __forceinline__ uint32_t lit_entry(int sym) {
  if (sym < 256) return ((uint32_t)sym << 8) | (K_LIT << 4);
  if (sym == 256) return K_END << 4;
  if (sym - 257 >= 29) return K_BAD << 4;
  uint32_t base; int ext;
  length_code(sym - 257, base, ext);
  return (base << 8) | ((uint32_t)ext << 20) | (K_LEN << 4);
}
__device__ __forceinline__ uint32_t dist_entry(int ds) {
  if (ds >= 30) return 1u << 24;
  uint32_t base; int ext;
  dist_code(ds, base, ext);
  return (base << 8) | ((uint32_t)ext << 4);
}

// The fast tables hold 16-bit entries (2048 + 1024 of them: 6 KB): bits 0-3 the code length (0 = longer than the table resolves),
// bits 4-5 the kind, bits 6-13 the literal byte or the index of the length symbol (0..28) / bits 4-8 the index of the distance symbol
// (30, 31: not a distance, bit 9). Base value and extra bits follow from the index by arithmetic (length_code / dist_code) in the
// lanes that need them.
__device__ __forceinline__ uint32_t lit16(int sym) {
  if (sym < 256) return ((uint32_t)sym << 6) | (K_LIT << 4);
  if (sym == 256) return K_END << 4;
  if (sym - 257 >= 29) return K_BAD << 4;
  return ((uint32_t)(sym - 257) << 6) | (K_LEN << 4);
}
__device__ __forceinline__ uint32_t dist16(int ds) { return ds >= 30 ? (1u << 9) : ((uint32_t)ds << 4); }

// The body of a block. A wave that decodes symbol after symbol on the scalar unit needs ~100 scalar instructions per symbol, and a
// CU issues one scalar instruction per cycle for all its waves: that, not memory, bounded this kernel (rocprofv3: 1.2e10 scalar
// instructions for 9 124 members, 0.63 per CU cycle, the vector unit idle). fast_windows decodes a WINDOW of 64 bit positions at
// once on the vector unit: lane i assumes that a literal/length symbol starts at bit pos + i and decodes it completely — table
// lookup, extra bits, and for a length symbol the distance code behind it — which gives every lane the symbol's kind, payload and
// total length in bits, i.e. where the next symbol would start. The window's first bit IS a symbol start, so a short scalar walk
// (lane -> next lane, ~6 instructions per symbol) marks the lanes that really are symbols; one prefix sum over their output lengths
// places them, the literals are stored by their lanes at once and the matches are copied in order, 64 bytes per step, from the LDS
// ring or (sources further back than the ring) from global memory. A window ends in front of anything the tables do not resolve
// in one lookup, an end-of-block, a match that overlaps its own output, or the end of the stream: codes() decodes that one
// symbol the general way and comes back.
// inclusive prefix sum over the 64 lanes in six DPP steps (row_shr 1, 2, 4, 8, row_bcast 15 / 31): shuffles through ds_bpermute cost
// an LDS round trip each, and this sum sits on the critical path of every window
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp0(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false); }
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t x, int) {
  x += dpp0<0x111>(x); x += dpp0<0x112>(x); x += dpp0<0x114>(x); x += dpp0<0x118>(x);
  x += dpp0<0x142, 0xA>(x); x += dpp0<0x143, 0xC>(x);
  return x;
}
__device__ __forceinline__ void sload_5(const uint32_t* p, uint32_t& w0, uint32_t& w1, uint32_t& w2, uint32_t& w3, uint32_t& w4) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const uint64_t a = (uint64_t)(uintptr_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  const uint32_t* q = (const uint32_t*)(uintptr_t)(((uint64_t)hi << 32) | lo);
  u32x4 v; uint32_t t;
  asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dword %1, %2, 0x10\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v), "=&s"(t) : "s"(q) : "memory");
  w0 = v.x; w1 = v.y; w2 = v.z; w3 = v.w; w4 = t;
}
enum { C_LIT = 0, C_END = 1, C_MATCH = 2, C_STOP = 3 };
__device__ __forceinline__ void fast_windows(Bits& b, const Tables& T, Out& out) {
  out.flush();
  if (out.full || b.skip) return;
  const int lane = out.lane;
  uint8_t* const dst = out.dst; uint8_t* const ring = out.ring;
  const uint32_t cap = out.cap;
  uint32_t o = out.o;
  const uint32_t nwords = (uint32_t)(b.end - b.w0);
  uint32_t bp = (uint32_t)(b.p - b.w0) * 32u - (uint32_t)b.cnt;      // bits of the stream consumed so far
  for (int guard = 0; guard < 70000; ++guard) {
    const uint32_t wi = bp >> 5, r = bp & 31u;
    if (wi + 5u > nwords || bp + 160u > b.end_bit) break;               // the last few dozen bytes of a member: the general path
    uint32_t W0, W1, W2, W3, W4;
    sload_5(b.w0 + wi, W0, W1, W2, W3, W4);
    // 64 stream bits from bit position bp + lane
    const uint32_t t = r + (uint32_t)lane, d = t >> 5, sh = t & 31u;
    const uint32_t xl = d == 0u ? W0 : d == 1u ? W1 : W2, xm = d == 0u ? W1 : d == 1u ? W2 : W3, xh = d == 0u ? W2 : d == 1u ? W3 : W4;
    uint64_t x = (((uint64_t)xm << 32) | xl) >> sh;
    if (sh) x |= (uint64_t)xh << (64u - sh);
    // the symbol that would start here
    const uint32_t e = T.lfast[(uint32_t)x & ((1u << LFAST) - 1u)];
    const uint32_t len = e & 15u, kind = (e >> 4) & 3u;
    uint32_t cls = len == 0u ? (uint32_t)C_STOP : kind == K_LIT ? (uint32_t)C_LIT : kind == K_END ? (uint32_t)C_END : kind == K_LEN ? (uint32_t)C_MATCH : (uint32_t)C_STOP;
    uint32_t tb = len, mlen = 1u, dist = 0u;
    const uint32_t byte = (e >> 6) & 255u;
    if (cls == C_MATCH) {
      uint32_t lbase, dbase; int lext, dext;
      length_code((int)byte, lbase, lext);                              // (byte = the length symbol's index here)
      const uint64_t x1 = x >> len;
      mlen = lbase + ((uint32_t)x1 & ((1u << lext) - 1u));
      const uint64_t x2 = x1 >> lext;
      const uint32_t de = T.dfast[(uint32_t)x2 & ((1u << DFAST) - 1u)];
      const uint32_t dl = de & 15u;
      dist_code((int)((de >> 4) & 31u), dbase, dext);
      dist = dbase + ((uint32_t)(x2 >> dl) & ((1u << dext) - 1u));
      tb = len + (uint32_t)lext + dl + (uint32_t)dext;                   // <= 15 + 5 + 15 + 13 = 48 of the 64 bits at hand
      if (dl == 0u || (de >> 9)) cls = C_STOP;
    }
    // ---- which lanes are symbols: the walk from lane 0 -------------------------------------------------------------
    const uint32_t next = (uint32_t)lane + tb;
    uint64_t valid = 0;
    uint32_t i = 0;
    bool stop = false;
    while (i < 64u) {
      const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cls, (int)i);
      if (c == C_STOP || c == C_END) { stop = true; break; }
      valid |= 1ull << i;
      i = (uint32_t)__builtin_amdgcn_readlane((int)next, (int)i);
    }
    uint32_t consumed = i;                                               // bits (the stopping symbol itself is not consumed)
    const bool mine = (valid >> lane) & 1ull;
    // ---- where the symbols' bytes go ----------------------------------------------------------------------------------
    const uint32_t outlen = mine ? mlen : 0u;                            // (literals: mlen = 1)
    const uint32_t incl = wave_incl_sum(outlen, lane);
    const uint32_t at = o + (incl - outlen);
    // a symbol that cannot be committed here: output beyond ISIZE, a match that reaches in front of the member's output or into its
    // own (dist < mlen: the general path's repeating copy). The window ends in front of the first one.
    const bool bad = mine && (at + outlen > cap || (cls == C_MATCH && (dist > at || dist < mlen)));
    const uint64_t bm = __ballot(bad);
    uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (bm) {
      const int j = __builtin_ctzll(bm);
      valid &= (1ull << j) - 1ull;
      consumed = (uint32_t)j;
      total = (uint32_t)__builtin_amdgcn_readlane((int)(incl - outlen), j);
      stop = true;
    }
    const bool on = (valid >> lane) & 1ull;
    // ---- literals: every lane its own byte; matches: in stream order, the whole wave on each ----------------------------
    if (on && cls == C_LIT) { dst[at] = (uint8_t)byte; ring[at & (RING - 1u)] = (uint8_t)byte; }
    __builtin_amdgcn_wave_barrier();
    uint64_t mm = __ballot(on && cls == C_MATCH);
    while (mm) {
      const int l = __builtin_ctzll(mm);
      mm &= mm - 1ull;
      const uint32_t ml = (uint32_t)__builtin_amdgcn_readlane((int)mlen, l), di = (uint32_t)__builtin_amdgcn_readlane((int)dist, l);
      const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)at, l);
      // the ring already holds this window's literals, up to o + total: a source is read from it only if none of them (nor the
      // 64 bytes a step writes ahead) can have overwritten it
      if (di + (o + total - a) + 64u <= RING) {
        for (uint32_t base = 0; base < ml; base += 64u) {
          const uint32_t k = base + (uint32_t)lane;
          if (k < ml) { const uint8_t v = ring[(a - di + k) & (RING - 1u)]; dst[a + k] = v; ring[(a + k) & (RING - 1u)] = v; }
        }
      } else {                                                           // the source left the ring: global memory, once the stores have landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (uint32_t base = 0; base < ml; base += 64u) {
          const uint32_t k = base + (uint32_t)lane;
          if (k < ml) { const uint8_t v = dst[a - di + k]; dst[a + k] = v; ring[(a + k) & (RING - 1u)] = v; }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    o += total; bp += consumed;
    if (stop || consumed == 0u) break;
  }
  // back to the scalar reader at bit position bp
  out.o = o;
  const uint32_t wi = bp >> 5, r = bp & 31u;
  if (wi < nwords) {
    const uint64_t v = sload_dword(b.w0 + wi);
    b.buf = v >> r; b.cnt = 32 - (int)r; b.p = b.w0 + wi + 1;
  } else { b.buf = 0; b.cnt = 0; b.p = b.end; }
}

__device__ __forceinline__ int codes(Bits& b, Tables& T, Out& out) {
  build_fast<LFAST>(T.lcount, T.lsymbol, T.lfast, out.lane, [](int s_) { return lit16(s_); });
  build_fast<DFAST>(T.dcount, T.dsymbol, T.dfast, out.lane, [](int s_) { return dist16(s_); });
  const Counts kl = load_counts(T.lcount), kd = load_counts(T.dcount);
  for (;;) {
    fast_windows(b, T, out);
    // one symbol the general way: long codes, the end of the stream or of the block, far or overlapping matches, errors
    uint32_t e;
    {
      const int sym = decode_reg(b, kl, T.lsymbol);              // (bit-serial canonical decode: a few percent of the symbols come here)
      if (sym < 0) return sym;
      e = lit_entry(sym);
    }
    const uint32_t kind = (e >> 4) & 3u;
    if (kind == K_LIT) { out.literal((e >> 8) & 255u); continue; }
    if (kind == K_END) { out.flush(); return out.full ? -3 : 0; }
    if (kind == K_BAD) return -4;
    const int lext = (int)((e >> 20) & 15u);
    b.refill();
    if (lext > b.cnt) { b.over = true; return -1; }
    const uint32_t mlen = ((e >> 8) & 0x1FFu) + ((uint32_t)b.buf & ((1u << lext) - 1u));
    b.buf >>= lext; b.cnt -= lext;
    uint32_t d;
    {
      const int ds = decode_reg(b, kd, T.dsymbol);
      if (ds < 0) return ds;
      d = dist_entry(ds);
    }
    if (d >> 24) return -5;
    const int dext = (int)((d >> 4) & 15u);
    b.refill();
    if (dext > b.cnt) { b.over = true; return -1; }
    const uint32_t dist = ((d >> 8) & 0xFFFFu) + ((uint32_t)b.buf & ((1u << dext) - 1u));
    b.buf >>= dext; b.cnt -= dext;
    if (dist > out.o + out.npend) return -6;
    if (!out.match(mlen, dist)) return -3;
  }
}

__global__ __launch_bounds__(64) void bgzf_inflate(const uint8_t* __restrict__ comp, const uint64_t* __restrict__ comp_off,
                                                   const uint64_t* __restrict__ out_off, int64_t n, uint8_t* __restrict__ outp,
                                                   int32_t* __restrict__ status) {
  __shared__ Tables T;
  __shared__ uint8_t ring[RING];
  const int lane = threadIdx.x;
  const int64_t m = blockIdx.x;                                 // one wave = one block = one member: everything below is uniform
  if (m >= n) return;
  const uint64_t c0 = comp_off[m], c1 = comp_off[m + 1];
  const uint64_t a0 = c0 & ~3ull;                               // comp is 256-byte aligned (hipMalloc), so this is a dword boundary
  const uint32_t* w0 = reinterpret_cast<const uint32_t*>(comp + a0);
  Bits b{w0, w0 + (uint32_t)((c1 - a0 + 3) / 4), 0ull, 0, (int)(c0 - a0), false, w0, (uint32_t)((c1 - a0) * 8)};
  // bits beyond the member's last byte (the tail of its last dword) are never consumed by a valid stream: it ends first
  Out out{outp + out_off[m], 0u, (uint32_t)(out_off[m + 1] - out_off[m]), 0u, 0u, lane, ring, false};
  int err = 0;
  for (int guard = 0; guard < 70000 && !err; ++guard) {        // a member holds at most 64 KiB: far fewer blocks than this
    const uint32_t last = b.get(1), type = b.get(2);
    if (b.over) { err = -1; break; }
    if (type == 0) {                                            // stored
      b.buf >>= (b.cnt & 7); b.cnt -= (b.cnt & 7);
      const uint32_t len = b.get(16), nlen = b.get(16);
      if (b.over || (len ^ 0xFFFFu) != nlen) { err = -7; break; }
      if (out.o + out.npend + len > out.cap) { err = -3; break; }
      for (uint32_t k = 0; k < len; ++k) { const uint32_t v = b.get(8); if (b.over) { err = -1; break; } out.literal(v); if (out.full) { err = -3; break; } }
    } else if (type == 1) {                                     // fixed codes
      if (lane == 0) {
        int s = 0;
        for (; s < 144; ++s) T.lengths[s] = 8;
        for (; s < 256; ++s) T.lengths[s] = 9;
        for (; s < 280; ++s) T.lengths[s] = 7;
        for (; s < FIXLCODES; ++s) T.lengths[s] = 8;
      }
      __builtin_amdgcn_wave_barrier();
      construct_by_lane0(T.lcount, T.lsymbol, T.lengths, FIXLCODES, lane);
      if (lane == 0) for (int s = 0; s < MAXDCODES; ++s) T.lengths[s] = 5;
      __builtin_amdgcn_wave_barrier();
      construct_by_lane0(T.dcount, T.dsymbol, T.lengths, MAXDCODES, lane);
      err = codes(b, T, out);
    } else if (type == 2) {                                     // dynamic codes
      const int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
      if (b.over || nlen > MAXLCODES || ndist > MAXDCODES) { err = -8; break; }
      int idx = 0;
      for (; idx < ncode; ++idx) { const uint32_t v = b.get(3); if (lane == 0) T.lengths[CLORDER[idx]] = (uint8_t)v; }
      for (; idx < 19; ++idx) if (lane == 0) T.lengths[CLORDER[idx]] = 0;
      __builtin_amdgcn_wave_barrier();
      if (construct_by_lane0(T.lcount, T.lsymbol, T.lengths, 19, lane) != 0) { err = -9; break; }
      idx = 0;
      while (idx < nlen + ndist) {
        int sym = decode(b, T.lcount, T.lsymbol);
        if (sym < 0) { err = sym; break; }
        if (sym < 16) { if (lane == 0) T.lengths[idx] = (uint8_t)sym; ++idx; }
        else {
          int len = 0, rep;
          if (sym == 16) {
            if (idx == 0) { err = -10; break; }
            __builtin_amdgcn_wave_barrier();
            len = __builtin_amdgcn_readfirstlane((int)T.lengths[idx - 1]); rep = 3 + (int)b.get(2);
          } else if (sym == 17) rep = 3 + (int)b.get(3);
          else rep = 11 + (int)b.get(7);
          if (idx + rep > nlen + ndist) { err = -11; break; }
          while (rep--) { if (lane == 0) T.lengths[idx] = (uint8_t)len; ++idx; }
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (err) break;
      if (b.over) { err = -1; break; }
      __builtin_amdgcn_wave_barrier();
      if (T.lengths[256] == 0) { err = -12; break; }
      // the 19-symbol code used T.lcount / T.lsymbol; the real tables are built now (distance lengths follow the literal ones)
      int r = construct_by_lane0(T.dcount, T.dsymbol, T.lengths + nlen, ndist, lane);
      if (r < 0 || (r > 0 && ndist - (int)T.dcount[0] != 1)) { err = -14; break; }
      r = construct_by_lane0(T.lcount, T.lsymbol, T.lengths, nlen, lane);
      if (r < 0 || (r > 0 && nlen - (int)T.lcount[0] != 1)) { err = -13; break; }
      err = codes(b, T, out);
    } else { err = -15; break; }
    if (last) break;
  }
  out.flush();
  if (out.full && !err) err = -3;
  if (!err && out.o != out.cap) err = -16;                      // ISIZE of the member must be met exactly
  if (lane == 0) status[m] = err;
}

}  // namespace

// ---- CRC-32 of the inflated members (the gzip trailer's check, as htslib's bgzf.c verifies it) --------------------------------
// One wave per member. The member is cut into 64 equal slices (the message is thought of as zero-padded at the FRONT to 64 x S
// bytes: leading zeros leave a zero state untouched); lane i runs the byte-wise table CRC over its slice from state 0, moves
// the result across the (63 - i) x S bytes that follow it with the precomputed "2^k zero bytes" operators, the lanes' states
// are XORed, and the initial state 0xFFFFFFFF is moved across the n real bytes the same way:
//   state(I, D1 || D2) = shift_|D2|(state(I, D1)) ^ state(0, D2),   crc = ~state(0xFFFFFFFF, M).
// tab[0..255] = the reflected CRC-32 table (polynomial 0xEDB88320), zop[k][b] = column b of the operator for 2^k zero bytes.
__device__ __forceinline__ uint32_t crc_shift(uint32_t v, uint32_t nbytes, const uint32_t (*zop)[32]) {
  for (int k = 0; nbytes; ++k, nbytes >>= 1) {
    if (nbytes & 1u) {
      uint32_t r = 0;
      for (int b = 0; b < 32; ++b) r ^= ((v >> b) & 1u) ? zop[k][b] : 0u;
      v = r;
    }
  }
  return v;
}
__global__ __launch_bounds__(64) void bgzf_crc32(const uint8_t* __restrict__ data, const uint64_t* __restrict__ out_off, int64_t n,
                                                 const uint32_t* __restrict__ tables, uint32_t* __restrict__ crc) {
  __shared__ uint32_t tab[256];
  __shared__ uint32_t zop[17][32];
  const int lane = threadIdx.x;
  for (int i = lane; i < 256; i += 64) tab[i] = tables[i];
  for (int i = lane; i < 17 * 32; i += 64) zop[i / 32][i % 32] = tables[256 + i];
  __syncthreads();
  const int64_t m = blockIdx.x;
  if (m >= n) return;
  const uint8_t* p = data + out_off[m];
  const uint32_t len = (uint32_t)(out_off[m + 1] - out_off[m]);
  const uint32_t S = (len + 63u) / 64u, pad = 64u * S - len;          // slice length, zero bytes in front
  uint32_t c = 0;
  const uint32_t lo = (uint32_t)lane * S, hi = lo + S;                // slice in padded coordinates
  {   // this lane's bytes: single bytes up to a 16-byte boundary, then 16 at a time (one load per 16 table steps), then the tail
    const uint8_t* q = p + ((lo < pad ? pad : lo) - pad);
    const uint8_t* e = p + (hi > pad ? hi - pad : 0u);
    if (q > e) q = e;
    while (q < e && ((uintptr_t)q & 15u)) { c = tab[(c ^ *q) & 0xFFu] ^ (c >> 8); ++q; }
    for (; q + 16 <= e; q += 16) {
      const uint4 w = *reinterpret_cast<const uint4*>(q);
      const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        uint32_t v = ww[k];
#pragma unroll
        for (int b = 0; b < 4; ++b) { c = tab[(c ^ v) & 0xFFu] ^ (c >> 8); v >>= 8; }
      }
    }
    while (q < e) { c = tab[(c ^ *q) & 0xFFu] ^ (c >> 8); ++q; }
  }
  c = crc_shift(c, (63u - (uint32_t)lane) * S, zop);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) c ^= (uint32_t)__shfl_xor((int)c, d, 64);
  if (lane == 0) crc[m] = ~(c ^ crc_shift(0xFFFFFFFFu, len, zop));
}

void vsv_launch_bgzf_inflate(hipStream_t st, const uint8_t* comp, const uint64_t* comp_off, const uint64_t* out_off, int64_t n, uint8_t* out,
                             int32_t* status) {
  if (n <= 0) return;
  bgzf_inflate<<<(int)n, 64, 0, st>>>(comp, comp_off, out_off, n, out, status);
}

// tables[0..255] = CRC-32 table, tables[256 + 32 k + b] = column b of the operator "2^k zero bytes" (k = 0..16)
void vsv_crc32_tables(uint32_t* t) {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
    t[i] = c;
  }
  uint32_t* z = t + 256;
  for (int b = 0; b < 32; ++b) {                    // one zero byte: state -> tab[state & 0xFF] ^ (state >> 8), column by column
    const uint32_t v = 1u << b;
    z[b] = t[v & 0xFFu] ^ (v >> 8);
  }
  for (int k = 1; k < 17; ++k)                      // square: apply the previous operator to each of its own columns
    for (int b = 0; b < 32; ++b) {
      const uint32_t v = z[32 * (k - 1) + b];
      uint32_t r = 0;
      for (int j = 0; j < 32; ++j) if ((v >> j) & 1u) r ^= z[32 * (k - 1) + j];
      z[32 * k + b] = r;
    }
}
void vsv_launch_bgzf_crc32(hipStream_t st, const uint8_t* data, const uint64_t* out_off, int64_t n, const uint32_t* tables, uint32_t* crc) {
  if (n > 0) bgzf_crc32<<<(int)n, 64, 0, st>>>(data, out_off, n, tables, crc);
}
