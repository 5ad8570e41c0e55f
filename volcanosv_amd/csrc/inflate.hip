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

constexpr int LFAST = 9, DFAST = 8;    // bits resolved by one table lookup; longer codes fall back to the bit-serial decode
struct Tables {
  uint16_t lcount[MAXBITS + 1], lsymbol[FIXLCODES];
  uint16_t dcount[MAXBITS + 1], dsymbol[MAXDCODES];
  uint8_t lengths[MAXLCODES + MAXDCODES + 2];
  uint16_t lfast[1 << LFAST];          // [next LFAST stream bits] -> symbol << 4 | code length, 0 = longer code
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
  const uint32_t* w; uint32_t nwords; uint32_t wi; uint64_t buf; int cnt; int skip; bool over;
  // the stream is read as aligned dwords; `skip` leading bytes of the first dword belong to the previous member
  __device__ __forceinline__ void refill() {
    if (cnt <= 32 && wi < nwords) {
      uint64_t v = sload_dword(w + wi); ++wi;
      int nb = 32;
      if (skip) { v >>= 8 * skip; nb -= 8 * skip; skip = 0; }
      buf |= v << cnt; cnt += nb;
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
template <int BITS>
__device__ __forceinline__ void build_fast(const uint16_t* count, const uint16_t* symbol, uint16_t* fast, int lane) {
  for (int i = lane; i < (1 << BITS); i += 64) fast[i] = 0;
  __builtin_amdgcn_wave_barrier();
  int code = 0, index = 0;                           // canonical first code / first sorted index of the current length
  for (int len = 1; len <= BITS; ++len) {
    const int c = __builtin_amdgcn_readfirstlane((int)count[len]);
    for (int r = lane; r < c; r += 64) {
      const uint32_t rev = __builtin_bitreverse32((uint32_t)(code + r)) >> (32 - len);
      const uint16_t e = (uint16_t)((symbol[index + r] << 4) | len);
      for (uint32_t hi = 0; hi < (1u << (BITS - len)); ++hi) fast[rev | (hi << len)] = e;
    }
    code = (code + c) << 1;
    index += c;
  }
  __builtin_amdgcn_wave_barrier();
}

template <int BITS>
__device__ __forceinline__ int decode_fast(Bits& b, const uint16_t* fast, const Counts& k, const uint16_t* symbol) {
  if (b.cnt < MAXBITS) { b.refill(); if (b.cnt < MAXBITS) b.refill(); }
  const int e = __builtin_amdgcn_readfirstlane((int)fast[(uint32_t)b.buf & ((1u << BITS) - 1u)]);
  const int len = e & 15;
  if (e != 0 && len <= b.cnt) { b.buf >>= len; b.cnt -= len; return e >> 4; }
  return decode_reg(b, k, symbol);                   // code longer than BITS bits, or the stream is about to end
}

__constant__ uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t CLORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// Output side of one member: literals wait one per lane and are stored 64 at a time.
struct Out {
  uint8_t* dst; uint32_t o, cap; uint32_t npend; uint32_t lit; int lane;      // o counts flushed bytes; lit = this lane's pending byte
  __device__ __forceinline__ void flush() {
    if ((uint32_t)lane < npend) dst[o + lane] = (uint8_t)lit;
    o += npend; npend = 0;
  }
  __device__ __forceinline__ bool literal(uint32_t sym) {
    if (o + npend >= cap) return false;
    if ((uint32_t)lane == npend) lit = sym;
    if (++npend == 64) flush();
    return true;
  }
  // copy len bytes from distance dist behind the write position; sources may overlap the destination (dist < len)
  __device__ __forceinline__ bool match(uint32_t len, uint32_t dist) {
    flush();
    if (dist > o || o + len > cap) return false;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the bytes just stored are the bytes about to be read
    const uint32_t step = dist < 64u ? dist : 64u;
    for (uint32_t base = 0; base < len; base += step) {
      const uint32_t nb = (len - base) < step ? (len - base) : step;
      if ((uint32_t)lane < nb) dst[o + base + lane] = dst[o + base + lane - dist];
      if (base + step < len && dist < len) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next step reads what this one wrote
    }
    o += len;
    return true;
  }
};

__device__ __forceinline__ int codes(Bits& b, Tables& T, Out& out) {
  build_fast<LFAST>(T.lcount, T.lsymbol, T.lfast, out.lane);
  build_fast<DFAST>(T.dcount, T.dsymbol, T.dfast, out.lane);
  const Counts kl = load_counts(T.lcount), kd = load_counts(T.dcount);
  for (;;) {
    int sym = decode_fast<LFAST>(b, T.lfast, kl, T.lsymbol);
    if (sym < 0) return sym;
    if (sym < 256) {
      if (!out.literal((uint32_t)sym)) return -3;
    } else if (sym == 256) {
      return 0;
    } else {
      sym -= 257;
      if (sym >= 29) return -4;
      const uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)LBASE[sym]) + b.get(__builtin_amdgcn_readfirstlane((int)LEXT[sym]));
      const int ds = decode_fast<DFAST>(b, T.dfast, kd, T.dsymbol);
      if (ds < 0) return ds;
      if (ds >= 30) return -5;
      const uint32_t dist = (uint32_t)__builtin_amdgcn_readfirstlane((int)DBASE[ds]) + b.get(__builtin_amdgcn_readfirstlane((int)DEXT[ds]));
      if (b.over) return -1;
      if (dist > out.o + out.npend) return -6;
      if (!out.match(len, dist)) return -3;
    }
  }
}

__global__ __launch_bounds__(64) void bgzf_inflate(const uint8_t* __restrict__ comp, const uint64_t* __restrict__ comp_off,
                                                   const uint64_t* __restrict__ out_off, int64_t n, uint8_t* __restrict__ outp,
                                                   int32_t* __restrict__ status) {
  __shared__ Tables T;
  const int lane = threadIdx.x;
  const int64_t m = blockIdx.x;                                 // one wave = one block = one member: everything below is uniform
  if (m >= n) return;
  const uint64_t c0 = comp_off[m], c1 = comp_off[m + 1];
  const uint64_t a0 = c0 & ~3ull;                               // comp is 256-byte aligned (hipMalloc), so this is a dword boundary
  Bits b{reinterpret_cast<const uint32_t*>(comp + a0), (uint32_t)((c1 - a0 + 3) / 4), 0u, 0ull, 0, (int)(c0 - a0), false};
  // bits beyond the member's last byte (the tail of its last dword) are never consumed by a valid stream: it ends first
  Out out{outp + out_off[m], 0u, (uint32_t)(out_off[m + 1] - out_off[m]), 0u, 0u, lane};
  int err = 0;
  for (int guard = 0; guard < 70000 && !err; ++guard) {        // a member holds at most 64 KiB: far fewer blocks than this
    const uint32_t last = b.get(1), type = b.get(2);
    if (b.over) { err = -1; break; }
    if (type == 0) {                                            // stored
      b.buf >>= (b.cnt & 7); b.cnt -= (b.cnt & 7);
      const uint32_t len = b.get(16), nlen = b.get(16);
      if (b.over || (len ^ 0xFFFFu) != nlen) { err = -7; break; }
      if (out.o + out.npend + len > out.cap) { err = -3; break; }
      for (uint32_t k = 0; k < len; ++k) { const uint32_t v = b.get(8); if (b.over) { err = -1; break; } out.literal(v); }
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
