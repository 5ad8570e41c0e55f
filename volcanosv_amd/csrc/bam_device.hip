// bam_device.hip — BAM records parsed on the GPU: inflated BGZF bytes (device) -> record SoA (device).
// Replaces pysam's record decode (htslib sam.c bam_read1) for the fields the path reads; SURVEY §8f-1 "GPU inflate producing
// the §8b SoA directly".
//
// The only sequential thing in a BAM stream is the chain of record starts (each record's block_size leads to the next).
// It is broken per BGZF member: every member speculates where its first record starts (strict header plausibility test
// over candidate offsets), walks its own records, and reports where it lands in a later member; the host checks that every
// landing equals the speculated start there, starting from member 0 whose start (end of the BAM header) is known exactly —
// a chain that verifies is the true chain, a mismatch is patched with the proven landing and the walk repeated.
// After that everything is data-parallel: one lane per record for the fixed fields, name hash and hp flags, a scan for the
// CIGAR offsets, one wave per record for the CIGAR copy (CG:B,I long CIGARs honoured), and a stable sort of the name hashes
// for the dense first-appearance query ids.
#include "vsv_device.h"

namespace {

constexpr uint64_t NONE64 = 0xFFFFFFFFFFFFFFFFull;

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

// plausibility of a record header at absolute offset x (needs x + 36 <= total); *size = 4 + block_size
__device__ bool plausible(const uint8_t* s, uint64_t x, uint64_t total, int32_t n_ref, uint64_t* size) {
  if (x + 36 > total) return false;
  const uint8_t* r = s + x;
  const uint32_t bs = ld32(r);
  if (bs < 32 || bs > (1u << 29)) return false;
  const int32_t ref = (int32_t)ld32(r + 4), pos = (int32_t)ld32(r + 8);
  if (ref < -1 || ref >= n_ref || pos < -1) return false;
  const uint32_t l_name = r[12], n_cig = ld16(r + 16);
  const int32_t l_seq = (int32_t)ld32(r + 20), nref = (int32_t)ld32(r + 24), npos = (int32_t)ld32(r + 28);
  if (l_name < 2 || l_seq < 0 || nref < -1 || nref >= n_ref || npos < -1) return false;
  const uint64_t need = 32ull + l_name + 4ull * n_cig + (uint64_t)((l_seq + 1) / 2) + (uint64_t)l_seq;
  if (need > bs) return false;
  if (x + 36 + l_name > total) return true;            // name not available: accept on the numeric fields
  for (uint32_t k = 0; k + 1 < l_name; ++k) { const uint8_t c = r[36 + k]; if (c < 33 || c > 126) return false; }
  if (r[36 + l_name - 1] != 0) return false;
  *size = 4ull + bs;
  return true;
}

// spec[m] = absolute offset of the first plausible record start inside member m (two chained records when the second one is
// visible), NONE if the member holds none. Member 0..first_member-1 are header; member first_member starts at `first`.
__global__ __launch_bounds__(64) void rec_speculate(const uint8_t* __restrict__ s, const uint64_t* __restrict__ moff, int64_t n_members,
                                                    uint64_t first, int32_t n_ref, uint64_t* __restrict__ spec) {
  const int64_t m = blockIdx.x;
  const int lane = threadIdx.x;
  if (m >= n_members) return;
  const uint64_t total = moff[n_members], lo = moff[m], hi = moff[m + 1];
  if (first >= lo && first < hi) { if (lane == 0) spec[m] = first; return; }     // the exact start
  if (hi <= first) { if (lane == 0) spec[m] = NONE64; return; }                   // header members
  uint64_t found = NONE64;
  for (uint64_t base = lo; base < hi && found == NONE64; base += 64) {
    const uint64_t x = base + lane;
    bool ok = false;
    uint64_t sz = 0;
    if (x < hi && plausible(s, x, total, n_ref, &sz)) {
      uint64_t sz2;
      ok = (x + sz + 36 > total) ? (x + sz == total || x + sz + 36 > total) : plausible(s, x + sz, total, n_ref, &sz2);
    }
    const uint64_t b = __ballot(ok);
    if (b) found = base + (uint64_t)__builtin_ctzll(b);
  }
  if (lane == 0) spec[m] = found;
}

// walks the records of member m from spec[m]: count[m] = records starting in the member, land[m] = absolute offset of the
// first record start at or beyond the member's end — or, when a record is not complete inside the inflated window, the start
// of that record (< member end): the caller resumes there with the next window. NONE on a malformed hop.
// WRITE: also stores the record offsets at rec_off[base[m] + k].
template <bool WRITE>
__global__ __launch_bounds__(256) void rec_chain(const uint8_t* __restrict__ s, const uint64_t* __restrict__ moff, int64_t n_members,
                                                 const uint64_t* __restrict__ spec, uint32_t* __restrict__ count, uint64_t* __restrict__ land,
                                                 const uint64_t* __restrict__ base, uint64_t* __restrict__ rec_off) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n_members) return;
  const uint64_t total = moff[n_members], hi = moff[m + 1];
  uint64_t x = spec[m];
  if (x == NONE64) { if (!WRITE) { count[m] = 0; land[m] = NONE64; } return; }
  uint32_t c = 0;
  const uint64_t b0 = WRITE ? base[m] : 0;
  while (x < hi) {
    if (x + 4 > total) break;                        // the size field itself is cut off: the record belongs to the next window
    const uint32_t bs = ld32(s + x);
    if (bs < 32) { x = NONE64; break; }
    if (x + 4ull + bs > total) break;                // record not complete inside this window: land on its start
    if (WRITE) rec_off[b0 + c] = x;
    ++c;
    x += 4ull + bs;
  }
  if (!WRITE) { count[m] = c; land[m] = x; }
}

// one lane per record: fixed fields, flag byte, name hash (FNV-1a), output CIGAR length (CG:B,I aware), keep flag (tid filter)
__global__ __launch_bounds__(256) void rec_fields(const uint8_t* __restrict__ s, const uint64_t* __restrict__ rec_off, int64_t n, int32_t want_tid,
                                                  int32_t* __restrict__ pos, int32_t* __restrict__ tid, uint8_t* __restrict__ mapq,
                                                  uint8_t* __restrict__ flag, uint32_t* __restrict__ l_seq, uint32_t* __restrict__ sam_flag,
                                                  uint32_t* __restrict__ n_cig_out, uint64_t* __restrict__ cg_src, uint64_t* __restrict__ hash,
                                                  uint32_t* __restrict__ keep, uint32_t* __restrict__ err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* r = s + rec_off[i];
  const uint32_t bs = ld32(r);
  const int32_t ref = (int32_t)ld32(r + 4);
  const uint32_t l_name = r[12], nc = ld16(r + 16), fl = ld16(r + 18);
  const int32_t ls = (int32_t)ld32(r + 20);
  pos[i] = (int32_t)ld32(r + 8); tid[i] = ref; mapq[i] = r[13]; l_seq[i] = (uint32_t)ls; sam_flag[i] = fl;
  // the variable-length fields must fit the record: every later access (name, CIGAR, tags) relies on it
  // l_read_name counts the terminating NUL, so 0 is malformed (htslib's bam_read1 rejects it too); accepting it would make
  // name_copy's separator store land at d[-1]
  if (l_name == 0 || ls < 0 || 32ull + l_name + 4ull * nc + (uint64_t)((ls + 1ll) / 2) + (uint64_t)ls > (uint64_t)bs) {
    atomicOr(err, 4u);
    flag[i] = 0; hash[i] = 0; n_cig_out[i] = 0; cg_src[i] = 0; keep[i] = 0;
    return;
  }
  const uint8_t* name = r + 36;
  const uint32_t nlen = l_name ? l_name - 1u : 0u;
  uint64_t h = 1469598103934665603ull;
  uint8_t f8 = 0;
  for (uint32_t k = 0; k < nlen; ++k) {
    h ^= name[k]; h *= 1099511628211ull;
    if (k + 3 <= nlen && name[k] == 'h' && name[k + 1] == 'p') { if (name[k + 2] == '1') f8 |= VSV_F_HP1; else if (name[k + 2] == '2') f8 |= VSV_F_HP2; }
  }
  if (fl & 0x10) f8 |= VSV_F_REVERSE;
  if (fl & 0x800) f8 |= VSV_F_SUPP;
  if (fl & 0x100) f8 |= VSV_F_SECONDARY;
  if (fl & 0x4) f8 |= VSV_F_UNMAPPED;
  flag[i] = f8;
  hash[i] = h;
  // CIGAR source: the record's own ops, or the CG:B,I tag when the placeholder kSmN is present (htslib convention)
  uint64_t src = (uint64_t)(name + l_name - s);
  uint32_t n_out = nc;
  if (nc == 2) {
    uint64_t off = 36ull + l_name + 8ull + (uint64_t)((ls + 1) / 2) + (uint64_t)ls;     // first tag
    const uint64_t end = 4ull + bs;
    while (off + 3 <= end) {
      const uint8_t t0 = r[off], t1 = r[off + 1], ty = r[off + 2];
      off += 3;
      uint64_t len = 0;
      if (ty == 'A' || ty == 'c' || ty == 'C') len = 1;
      else if (ty == 's' || ty == 'S') len = 2;
      else if (ty == 'i' || ty == 'I' || ty == 'f') len = 4;
      else if (ty == 'Z' || ty == 'H') { uint64_t e = off; while (e < end && r[e]) ++e; len = e - off + 1; }
      else if (ty == 'B') {
        if (off + 5 > end) { atomicOr(err, 4u); break; }
        const uint8_t sub = r[off]; const uint32_t cnt = ld32(r + off + 1);
        const uint64_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
        len = 5 + es * cnt;
        if (len > end - off) { atomicOr(err, 4u); break; }                        // the array must end inside the record
        if (t0 == 'C' && t1 == 'G' && sub == 'I') { src = (uint64_t)(r + off + 5 - s); n_out = cnt; }
      } else { atomicOr(err, 1u); break; }
      off += len;
    }
  }
  n_cig_out[i] = n_out;
  cg_src[i] = src;
  keep[i] = (ref >= 0 && (want_tid < 0 || ref == want_tid)) ? 1u : 0u;
}

// compaction of the kept records + CIGAR copy: G lanes per input record — a whole wave for contig alignments (10^3-10^6 ops), eight
// lanes for reads (tens of ops: a wave per record left most lanes idle and took 1.5 ms for 3 M records, 0.5 TB/s)
template <int G>
__global__ __launch_bounds__(256) void rec_emit(const uint8_t* __restrict__ s, int64_t n, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ kidx,
                                                const int32_t* __restrict__ pos, const int32_t* __restrict__ tid, const uint8_t* __restrict__ mapq,
                                                const uint8_t* __restrict__ flag, const uint32_t* __restrict__ l_seq, const uint32_t* __restrict__ sam_flag,
                                                const uint32_t* __restrict__ n_cig_out, const uint64_t* __restrict__ cg_src, const uint64_t* __restrict__ hash,
                                                const uint64_t* __restrict__ cig_off_in, const uint64_t* __restrict__ rec_off,
                                                int32_t* __restrict__ o_pos, int32_t* __restrict__ o_tid, uint8_t* __restrict__ o_mapq,
                                                uint8_t* __restrict__ o_flag, uint32_t* __restrict__ o_l_seq, uint32_t* __restrict__ o_sam_flag,
                                                uint64_t* __restrict__ o_cig_off, uint32_t* __restrict__ o_cigar, uint64_t* __restrict__ o_hash,
                                                uint64_t* __restrict__ o_rec_off, uint64_t k0, uint64_t c0) {
  const int lane = threadIdx.x & (G - 1);
  const int64_t ng = (int64_t)gridDim.x * (blockDim.x / G);
  const int64_t n_pad = (n + (64 / G) - 1) / (64 / G) * (64 / G);          // whole waves iterate together: the group sums shuffle
  for (int64_t i0 = (int64_t)blockIdx.x * (blockDim.x / G) + (threadIdx.x / G); i0 < n_pad; i0 += ng) {
    const bool live = i0 < n && keep[i0];
    const int64_t i = i0 < n ? i0 : n - 1;
    const uint32_t kl = kidx[i];                     // index among the kept records of this window
    const uint64_t k = k0 + kl;                       // ... and of the whole file
    const uint64_t co = c0 + cig_off_in[i];
    if (live && lane == 0) {
      o_pos[k] = pos[i]; o_tid[k] = tid[i]; o_mapq[k] = mapq[i]; o_flag[k] = flag[i]; o_l_seq[k] = l_seq[i]; o_sam_flag[k] = sam_flag[i];
      o_cig_off[k] = co; o_hash[k] = hash[i]; o_rec_off[kl] = rec_off[i];
    }
    const uint8_t* src = s + cg_src[i];
    const uint32_t nc = live ? n_cig_out[i] : 0u;
    uint64_t ql = 0;                                   // query length of the CIGAR (M,I,S,=,X), summed while the ops are copied
    for (uint32_t c = lane; c < nc; c += G) {
      const uint32_t w = ld32(src + 4ull * c), op = w & 15u;
      o_cigar[co + c] = w;
      if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) ql += w >> 4;
    }
#pragma unroll
    for (int d = G / 2; d > 0; d >>= 1) ql += __shfl_xor(ql, d, 64);
    // a stored SEQ of another length is what the extractors assert on (H:397-398): VSV_F_SEQ_MISMATCH
    if (live && lane == 0 && l_seq[i] != 0 && ql != (uint64_t)l_seq[i]) o_flag[k] = flag[i] | (uint8_t)VSV_F_SEQ_MISMATCH;
  }
}

// u64 exclusive scan of (keep ? n_cig : 0), single pass per block + block sums on the host side is avoided by a two-level scheme:
// block sums -> serial scan by one block -> apply. n up to 2^31.
constexpr int SC64_TILE = 2048;
__global__ __launch_bounds__(256) void scan64_sums(const uint32_t* __restrict__ v, const uint32_t* __restrict__ keep, int64_t n, uint64_t* __restrict__ sums) {
  __shared__ uint64_t sh[256];
  const int64_t base = (int64_t)blockIdx.x * SC64_TILE + threadIdx.x * 8;
  uint64_t t = 0;
  for (int k = 0; k < 8; ++k) if (base + k < n && keep[base + k]) t += v[base + k];
  sh[threadIdx.x] = t;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) { if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d]; __syncthreads(); }
  if (threadIdx.x == 0) sums[blockIdx.x] = sh[0];
}
__global__ void scan64_serial(uint64_t* __restrict__ sums, int64_t nb, uint64_t* __restrict__ total) {
  uint64_t run = 0;
  for (int64_t b = 0; b < nb; ++b) { const uint64_t t = sums[b]; sums[b] = run; run += t; }
  *total = run;
}
__global__ __launch_bounds__(256) void scan64_apply(const uint32_t* __restrict__ v, const uint32_t* __restrict__ keep, int64_t n, const uint64_t* __restrict__ sums,
                                                    uint64_t* __restrict__ out) {
  __shared__ uint64_t sh[256];
  const int64_t base = (int64_t)blockIdx.x * SC64_TILE + threadIdx.x * 8;
  uint64_t loc[8], t = 0;
  for (int k = 0; k < 8; ++k) { loc[k] = t; if (base + k < n && keep[base + k]) t += v[base + k]; }
  sh[threadIdx.x] = t;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const uint64_t u = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += u;
    __syncthreads();
  }
  const uint64_t excl = sums[blockIdx.x] + sh[threadIdx.x] - t;
  for (int k = 0; k < 8; ++k) if (base + k < n) out[base + k] = excl + loc[k];
}

// ---- dense first-appearance query ids from the sorted (hash, record) pairs --------------------------------------------
// sorted by hash (stable, so records of one name ascend); head j = first slot of its hash group.
__global__ __launch_bounds__(256) void qid_mark_first(const uint64_t* __restrict__ skey, const uint32_t* __restrict__ sval, int64_t n,
                                                      uint32_t* __restrict__ is_first) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x)
    if (j == 0 || skey[j] != skey[j - 1]) is_first[sval[j]] = 1u;      // the group's lowest record index
}
// a group head hands its (scanned) id to every member; names (compact store, l_read_name bytes each) are compared byte by
// byte (a 64-bit hash collision between different names raises the error flag and the caller falls back to the host reader)
__global__ __launch_bounds__(256) void qid_assign(const uint64_t* __restrict__ skey, const uint32_t* __restrict__ sval, int64_t n,
                                                  const uint32_t* __restrict__ first_rank, const uint8_t* __restrict__ names,
                                                  const uint64_t* __restrict__ nm_off, const uint32_t* __restrict__ nm_len, uint32_t* __restrict__ qid,
                                                  uint32_t* __restrict__ err) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    if (j != 0 && skey[j] == skey[j - 1]) continue;
    const uint32_t head = sval[j];
    const uint32_t id = first_rank[head];
    const uint8_t* hn = names + nm_off[head];
    const uint32_t hl = nm_len[head];
    for (int64_t k = j; k < n && skey[k] == skey[j]; ++k) {
      const uint32_t r = sval[k];
      qid[r] = id;
      if (k != j) {
        const uint8_t* rn = names + nm_off[r];
        bool same = nm_len[r] == hl;
        for (uint32_t c = 0; same && c < hl; ++c) same = rn[c] == hn[c];
        if (!same) atomicOr(err, 2u);
      }
    }
  }
}
// names of a window's kept records into the compact store: length pass, then copy at store_base + scanned offset
__global__ __launch_bounds__(256) void win_name_lens(const uint8_t* __restrict__ s, const uint64_t* __restrict__ w_rec_off, int64_t nk, uint32_t* __restrict__ len) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nk; k += (int64_t)gridDim.x * blockDim.x) len[k] = s[w_rec_off[k] + 12];
}
__global__ __launch_bounds__(256) void win_name_store(const uint8_t* __restrict__ s, const uint64_t* __restrict__ w_rec_off, const uint32_t* __restrict__ loff,
                                                      int64_t nk, uint64_t k0, uint64_t n0, uint8_t* __restrict__ names, uint64_t* __restrict__ nm_off,
                                                      uint32_t* __restrict__ nm_len) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nk; k += (int64_t)gridDim.x * blockDim.x) {
    const uint8_t* nm = s + w_rec_off[k] + 36;
    const uint32_t l = s[w_rec_off[k] + 12];
    uint8_t* d = names + n0 + loff[k];
    for (uint32_t c = 0; c < l; ++c) d[c] = nm[c];
    nm_off[k0 + k] = n0 + loff[k];
    nm_len[k0 + k] = l;
  }
}
// SA:Z tag of every kept record of the window (SVIM_COLLECT.py:12, SE:479): one lane per record walks the aux fields (whose extent
// rec_fields has validated) and reports the text's stream offset and length (0: no SA tag); sa_store copies the texts, each followed
// by '\n', into the compact store in record order.
__global__ __launch_bounds__(256) void win_sa_find(const uint8_t* __restrict__ s, const uint64_t* __restrict__ w_rec_off, int64_t nk,
                                                   uint64_t* __restrict__ sa_off, uint32_t* __restrict__ sa_len) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nk; k += (int64_t)gridDim.x * blockDim.x) {
    const uint8_t* r = s + w_rec_off[k];
    const uint32_t bs = ld32(r), l_name = r[12], nc = ld16(r + 16);
    const int32_t ls = (int32_t)ld32(r + 20);
    uint64_t off = 36ull + l_name + 4ull * nc + (uint64_t)((ls + 1) / 2) + (uint64_t)ls;
    const uint64_t end = 4ull + bs;
    uint64_t so = 0; uint32_t sl = 0;
    while (off + 3 <= end) {
      const uint8_t t0 = r[off], t1 = r[off + 1], ty = r[off + 2];
      off += 3;
      uint64_t len = 0;
      if (ty == 'A' || ty == 'c' || ty == 'C') len = 1;
      else if (ty == 's' || ty == 'S') len = 2;
      else if (ty == 'i' || ty == 'I' || ty == 'f') len = 4;
      else if (ty == 'Z' || ty == 'H') {
        uint64_t e = off; while (e < end && r[e]) ++e;
        if (t0 == 'S' && t1 == 'A' && ty == 'Z') { so = (uint64_t)(r + off - s); sl = (uint32_t)(e - off); }
        len = e - off + 1;
      } else if (ty == 'B') {
        if (off + 5 > end) break;
        const uint8_t sub = r[off]; const uint32_t cnt = ld32(r + off + 1);
        len = 5 + ((sub == 'c' || sub == 'C') ? 1ull : (sub == 's' || sub == 'S') ? 2ull : 4ull) * cnt;
      } else break;
      if (len > end - off) break;
      off += len;
    }
    sa_off[k] = so; sa_len[k] = sl + 1u;          // + the separator
  }
}
__global__ __launch_bounds__(256) void win_sa_store(const uint8_t* __restrict__ s, const uint64_t* __restrict__ sa_off, const uint32_t* __restrict__ sa_len,
                                                    const uint32_t* __restrict__ loff, int64_t nk, uint64_t s0, uint8_t* __restrict__ blob) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nk; k += (int64_t)gridDim.x * blockDim.x) {
    uint8_t* d = blob + s0 + loff[k];
    const uint32_t l = sa_len[k] - 1u;
    const uint8_t* p = s + sa_off[k];
    for (uint32_t c = 0; c < l; ++c) d[c] = p[c];
    d[l] = '\n';
  }
}
// Packed SEQ of every kept record of the window (two bases per byte, BAM nibble alphabet): one lane per record reports the field's
// stream offset and byte count, seq_store copies the bytes into the compact store (one wave per record: reads are kilobases long) and
// records where every record's SEQ starts. The inserted sequences of sig_extract (SE:468-469) are slices of it (seq_slices below).
__global__ __launch_bounds__(256) void win_seq_find(const uint8_t* __restrict__ s, const uint64_t* __restrict__ w_rec_off, int64_t nk,
                                                    uint64_t* __restrict__ seq_off, uint32_t* __restrict__ seq_len) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nk; k += (int64_t)gridDim.x * blockDim.x) {
    const uint8_t* r = s + w_rec_off[k];
    const uint32_t l_name = r[12], nc = ld16(r + 16);
    const int32_t ls = (int32_t)ld32(r + 20);
    seq_off[k] = w_rec_off[k] + 36ull + l_name + 4ull * nc;             // (rec_fields has checked that the field ends inside the record)
    seq_len[k] = ls > 0 ? (uint32_t)((ls + 1) / 2) : 0u;
  }
}
__global__ __launch_bounds__(256) void win_seq_store(const uint8_t* __restrict__ s, const uint64_t* __restrict__ seq_off, const uint32_t* __restrict__ seq_len,
                                                     const uint32_t* __restrict__ loff, int64_t nk, uint64_t k0, uint64_t q0, uint8_t* __restrict__ blob,
                                                     uint64_t* __restrict__ rec_seq_off) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t k = wave; k < nk; k += nwaves) {
    const uint8_t* p = s + seq_off[k];
    uint8_t* d = blob + q0 + loff[k];
    const uint32_t l = seq_len[k];
    for (uint32_t c = (uint32_t)lane; c < l; c += 64u) d[c] = p[c];
    if (lane == 0) rec_seq_off[k0 + (uint64_t)k] = q0 + loff[k];
  }
}
// ASCII slices of the stored sequences: slice i = bases [start, start + len) of record rec[i], read from the REVERSED read when
// rev[i] (sig_extract's split INS, SE:215: reversed, not complemented). One wave per slice; the caller guarantees start + len <= l_seq.
__global__ __launch_bounds__(256) void seq_slices(const uint8_t* __restrict__ blob, const uint64_t* __restrict__ rec_seq_off, const uint32_t* __restrict__ l_seq,
                                                  const uint32_t* __restrict__ rec, const uint32_t* __restrict__ start, const uint32_t* __restrict__ len,
                                                  const uint8_t* __restrict__ rev, int64_t n, const uint64_t* __restrict__ out_off, uint8_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t i = wave; i < n; i += nwaves) {
    const uint32_t r = rec[i], a = start[i], l = len[i], ls = l_seq[r];
    const uint8_t* p = blob + rec_seq_off[r];
    uint8_t* d = out + out_off[i];
    const bool rv = rev[i] != 0;
    for (uint32_t c = (uint32_t)lane; c < l; c += 64u) {
      const uint32_t idx = rv ? ls - 1u - (a + c) : a + c;
      const uint32_t b = p[idx >> 1];
      d[c] = (uint8_t)"=ACMGRSVTWYHKDBN"[(idx & 1u) ? (b & 15u) : (b >> 4)];
    }
  }
}
// name table for the host: the first occurrences in id order, '\n'-separated (the stored NUL becomes the separator)
__global__ __launch_bounds__(256) void name_lens(const uint32_t* __restrict__ nm_len, const uint32_t* __restrict__ is_first, int64_t n, uint32_t* __restrict__ len) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) len[i] = is_first[i] ? nm_len[i] : 0u;
}
__global__ __launch_bounds__(256) void name_copy(const uint8_t* __restrict__ names, const uint64_t* __restrict__ nm_off, const uint32_t* __restrict__ nm_len,
                                                 const uint32_t* __restrict__ is_first, const uint32_t* __restrict__ noff, int64_t n, uint8_t* __restrict__ blob) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (!is_first[i]) continue;
    const uint8_t* nm = names + nm_off[i];
    const uint32_t l = nm_len[i];
    if (l == 0) continue;                      // rec_fields rejects empty names; never store at d[-1]
    uint8_t* d = blob + noff[i];
    for (uint32_t c = 0; c + 1 < l; ++c) d[c] = nm[c];
    d[l - 1] = '\n';
  }
}
__global__ void iota_u32(uint32_t* p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (uint32_t)i;
}

}  // namespace

// ---- launchers (orchestrated by capi.hip) --------------------------------------------------------------------------------------
void vsv_bamdev_speculate(hipStream_t st, const uint8_t* s, const uint64_t* moff, int64_t n_members, uint64_t first, int32_t n_ref, uint64_t* spec) {
  rec_speculate<<<(int)n_members, 64, 0, st>>>(s, moff, n_members, first, n_ref, spec);
}
void vsv_bamdev_chain(hipStream_t st, bool write, const uint8_t* s, const uint64_t* moff, int64_t n_members, const uint64_t* spec, uint32_t* count,
                      uint64_t* land, const uint64_t* base, uint64_t* rec_off) {
  const int grid = (int)((n_members + 255) / 256);
  if (write) rec_chain<true><<<grid, 256, 0, st>>>(s, moff, n_members, spec, count, land, base, rec_off);
  else rec_chain<false><<<grid, 256, 0, st>>>(s, moff, n_members, spec, count, land, base, rec_off);
}
void vsv_bamdev_fields(hipStream_t st, const uint8_t* s, const uint64_t* rec_off, int64_t n, int32_t want_tid, int32_t* pos, int32_t* tid, uint8_t* mapq,
                       uint8_t* flag, uint32_t* l_seq, uint32_t* sam_flag, uint32_t* n_cig_out, uint64_t* cg_src, uint64_t* hash, uint32_t* keep, uint32_t* err) {
  if (n <= 0) return;
  rec_fields<<<(int)((n + 255) / 256), 256, 0, st>>>(s, rec_off, n, want_tid, pos, tid, mapq, flag, l_seq, sam_flag, n_cig_out, cg_src, hash, keep, err);
}
void vsv_bamdev_scan64(hipStream_t st, const uint32_t* v, const uint32_t* keep, int64_t n, uint64_t* sums, uint64_t* out, uint64_t* total) {
  if (n <= 0) return;
  const int64_t nb = (n + SC64_TILE - 1) / SC64_TILE;
  scan64_sums<<<(int)nb, 256, 0, st>>>(v, keep, n, sums);
  scan64_serial<<<1, 1, 0, st>>>(sums, nb, total);
  scan64_apply<<<(int)nb, 256, 0, st>>>(v, keep, n, sums, out);
}
void vsv_bamdev_emit(hipStream_t st, const uint8_t* s, int64_t n, const uint32_t* keep, const uint32_t* kidx, const int32_t* pos, const int32_t* tid,
                     const uint8_t* mapq, const uint8_t* flag, const uint32_t* l_seq, const uint32_t* sam_flag, const uint32_t* n_cig_out,
                     const uint64_t* cg_src, const uint64_t* hash, const uint64_t* cig_off_in, const uint64_t* rec_off, int32_t* o_pos, int32_t* o_tid,
                     uint8_t* o_mapq, uint8_t* o_flag, uint32_t* o_l_seq, uint32_t* o_sam_flag, uint64_t* o_cig_off, uint32_t* o_cigar, uint64_t* o_hash,
                     uint64_t* w_rec_off, uint64_t k0, uint64_t c0, uint64_t ops_hint) {
  if (n <= 0) return;
  if (ops_hint / (uint64_t)n >= 256)
    rec_emit<64><<<2048, 256, 0, st>>>(s, n, keep, kidx, pos, tid, mapq, flag, l_seq, sam_flag, n_cig_out, cg_src, hash, cig_off_in, rec_off, o_pos, o_tid,
                                       o_mapq, o_flag, o_l_seq, o_sam_flag, o_cig_off, o_cigar, o_hash, w_rec_off, k0, c0);
  else
    rec_emit<8><<<4096, 256, 0, st>>>(s, n, keep, kidx, pos, tid, mapq, flag, l_seq, sam_flag, n_cig_out, cg_src, hash, cig_off_in, rec_off, o_pos, o_tid,
                                      o_mapq, o_flag, o_l_seq, o_sam_flag, o_cig_off, o_cigar, o_hash, w_rec_off, k0, c0);
}
void vsv_bamdev_iota(hipStream_t st, uint32_t* p, int64_t n) { if (n > 0) iota_u32<<<1024, 256, 0, st>>>(p, n); }
void vsv_bamdev_mark_first(hipStream_t st, const uint64_t* skey, const uint32_t* sval, int64_t n, uint32_t* is_first) {
  if (n > 0) qid_mark_first<<<1024, 256, 0, st>>>(skey, sval, n, is_first);
}
void vsv_bamdev_assign(hipStream_t st, const uint64_t* skey, const uint32_t* sval, int64_t n, const uint32_t* first_rank, const uint8_t* names,
                       const uint64_t* nm_off, const uint32_t* nm_len, uint32_t* qid, uint32_t* err) {
  if (n > 0) qid_assign<<<1024, 256, 0, st>>>(skey, sval, n, first_rank, names, nm_off, nm_len, qid, err);
}
void vsv_bamdev_win_name_lens(hipStream_t st, const uint8_t* s, const uint64_t* w_rec_off, int64_t nk, uint32_t* len) {
  if (nk > 0) win_name_lens<<<1024, 256, 0, st>>>(s, w_rec_off, nk, len);
}
void vsv_bamdev_win_name_store(hipStream_t st, const uint8_t* s, const uint64_t* w_rec_off, const uint32_t* loff, int64_t nk, uint64_t k0, uint64_t n0,
                               uint8_t* names, uint64_t* nm_off, uint32_t* nm_len) {
  if (nk > 0) win_name_store<<<1024, 256, 0, st>>>(s, w_rec_off, loff, nk, k0, n0, names, nm_off, nm_len);
}
void vsv_bamdev_name_lens(hipStream_t st, const uint32_t* nm_len, const uint32_t* is_first, int64_t n, uint32_t* len) {
  if (n > 0) name_lens<<<1024, 256, 0, st>>>(nm_len, is_first, n, len);
}
void vsv_bamdev_name_copy(hipStream_t st, const uint8_t* names, const uint64_t* nm_off, const uint32_t* nm_len, const uint32_t* is_first, const uint32_t* noff,
                          int64_t n, uint8_t* blob) {
  if (n > 0) name_copy<<<1024, 256, 0, st>>>(names, nm_off, nm_len, is_first, noff, n, blob);
}

void vsv_bamdev_win_seq_find(hipStream_t st, const uint8_t* s, const uint64_t* w_rec_off, int64_t nk, uint64_t* seq_off, uint32_t* seq_len) {
  if (nk > 0) win_seq_find<<<1024, 256, 0, st>>>(s, w_rec_off, nk, seq_off, seq_len);
}
void vsv_bamdev_win_seq_store(hipStream_t st, const uint8_t* s, const uint64_t* seq_off, const uint32_t* seq_len, const uint32_t* loff, int64_t nk,
                              uint64_t k0, uint64_t q0, uint8_t* blob, uint64_t* rec_seq_off) {
  if (nk > 0) win_seq_store<<<2048, 256, 0, st>>>(s, seq_off, seq_len, loff, nk, k0, q0, blob, rec_seq_off);
}
void vsv_bamdev_seq_slices(hipStream_t st, const uint8_t* blob, const uint64_t* rec_seq_off, const uint32_t* l_seq, const uint32_t* rec,
                           const uint32_t* start, const uint32_t* len, const uint8_t* rev, int64_t n, const uint64_t* out_off, uint8_t* out) {
  if (n > 0) seq_slices<<<(int)((n + 3) / 4 < 2048 ? (n + 3) / 4 : 2048), 256, 0, st>>>(blob, rec_seq_off, l_seq, rec, start, len, rev, n, out_off, out);
}
void vsv_bamdev_win_sa_find(hipStream_t st, const uint8_t* s, const uint64_t* w_rec_off, int64_t nk, uint64_t* sa_off, uint32_t* sa_len) {
  if (nk > 0) win_sa_find<<<1024, 256, 0, st>>>(s, w_rec_off, nk, sa_off, sa_len);
}
void vsv_bamdev_win_sa_store(hipStream_t st, const uint8_t* s, const uint64_t* sa_off, const uint32_t* sa_len, const uint32_t* loff, int64_t nk,
                             uint64_t s0, uint8_t* blob) {
  if (nk > 0) win_sa_store<<<1024, 256, 0, st>>>(s, sa_off, sa_len, loff, nk, s0, blob);
}
