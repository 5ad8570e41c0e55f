// bnd.hip — svim-asm breakend branch on the GPU (Complex_SV path of VolcanoSV).
//   segment_bnd   analyze_read_segments BND cases + CandidateBreakend canonical form
//                 (svim_asm/SVIM_inter.py:62-258, svim_asm/SVCandidate.py:350-373)
//   bnd_pair      form_partitions + pair_haplotypes_breakends + BND part of pair_candidates
//                 (svim_asm/SVIM_COMBINE.py:15-32, 105-117, 143-161, 334-365)
// Tiny data (<= a few 10^6 segments): one lane per split contig / per partition, no atomics, deterministic slots.
#include "vsv_device.h"

namespace {

constexpr int BND_MAX_SEG = 64;    // segments per split contig handled in the lane-local sort
constexpr int BND_MAX_PART = 16;   // >= max_partition

struct Seg { int32_t q_start, q_end, ref_id, ref_start, ref_end, rev; };

__device__ __forceinline__ int32_t clampi(int64_t v, int64_t hi) { if (v < 0) v = 0; if (v > hi) v = hi; return (int32_t)v; }

__device__ __forceinline__ vsv_bnd make_bnd(const vsv_segments& s, int t1, int64_t p1, int d1, int t2, int64_t p2, int d2,
                                            uint32_t read, int hap) {
  vsv_bnd b;
  const bool keep = (s.contig_rank[t1] < s.contig_rank[t2]) || (t1 == t2 && p1 < p2);   // SVCandidate.py:352
  if (keep) {
    b.src_tid = t1; b.src_pos = clampi(p1, s.contig_len[t1]); b.dst_tid = t2; b.dst_pos = clampi(p2, s.contig_len[t2]);
    b.meta = (d1 ? VSV_B_SRC_FWD : 0u) | (d2 ? VSV_B_DST_FWD : 0u);
  } else {
    b.src_tid = t2; b.src_pos = clampi(p2, s.contig_len[t2]); b.dst_tid = t1; b.dst_pos = clampi(p1, s.contig_len[t1]);
    b.meta = (d2 ? 0u : VSV_B_SRC_FWD) | (d1 ? 0u : VSV_B_DST_FWD);
  }
  if (hap == 2) b.meta |= VSV_B_HAP2;
  b.read = read; b.read2 = 0xFFFFFFFFu; b.pad = 0;
  return b;
}

__device__ bool bnd_of_pair(const vsv_segments& s, const vsv_bnd_params& p, const Seg& cur, const Seg& nxt, uint32_t read, int hap,
                            vsv_bnd& out) {
  const int64_t QOT = p.query_overlap_tolerance, QGT = p.query_gap_tolerance, ROT = p.reference_overlap_tolerance;
  const int64_t MINSV = p.min_sv_size, MAXSV = p.max_sv_size;
  const int64_t dor = (int64_t)nxt.q_start - cur.q_end;
  const int c1 = cur.ref_id, c2 = nxt.ref_id;
  if (c1 == c2) {
    if (cur.rev == nxt.rev) {
      const int64_t dref = cur.rev ? (int64_t)cur.ref_start - nxt.ref_end : (int64_t)nxt.ref_start - cur.ref_end;
      if (dor >= -QOT) {
        if (dref >= -ROT) {
          const int64_t dev = dor - dref;
          if (dev >= MINSV) return false;
          if (-MAXSV <= dev && dev <= -MINSV) return false;
          if (dev < -MAXSV && dor <= QGT) {                                                     // SVIM_inter.py:131-139
            out = cur.rev ? make_bnd(s, c1, cur.ref_start, 0, c1, (int64_t)nxt.ref_end - 1, 0, read, hap)
                          : make_bnd(s, c1, (int64_t)cur.ref_end - 1, 1, c1, nxt.ref_start, 1, read, hap);
            return true;
          }
        } else if (dor <= QGT) {                                                                // :141-168
          const int64_t dev = dor - dref;
          if (dev >= MINSV) {
            if (!cur.rev) {
              if (nxt.ref_end > cur.ref_start) return false;
              if (dref >= -MAXSV) return false;
              out = make_bnd(s, c1, (int64_t)cur.ref_end - 1, 1, c1, nxt.ref_start, 1, read, hap); return true;
            }
            if (nxt.ref_start < cur.ref_end) return false;
            if (dref >= -MAXSV) return false;
            out = make_bnd(s, c1, cur.ref_start, 0, c1, (int64_t)nxt.ref_end - 1, 0, read, hap); return true;
          }
        }
      }
      return false;
    }
    if (!cur.rev && nxt.rev) {                                                                  // :171-192
      const int64_t dref = (int64_t)nxt.ref_end - cur.ref_end, dev = dor - dref;
      if (-QOT <= dor && dor <= QGT) {
        if ((int64_t)nxt.ref_start - cur.ref_end >= -ROT) {
          if (MINSV <= -dev && -dev <= MAXSV) return false;
          out = make_bnd(s, c1, (int64_t)cur.ref_end - 1, 1, c1, (int64_t)nxt.ref_end - 1, 0, read, hap); return true;
        } else if ((int64_t)cur.ref_start - nxt.ref_end >= -ROT) {
          if (MINSV <= dev && dev <= MAXSV) return false;
          out = make_bnd(s, c1, (int64_t)cur.ref_end - 1, 1, c1, (int64_t)nxt.ref_end - 1, 0, read, hap); return true;
        }
      }
      return false;
    }
    const int64_t dref = (int64_t)nxt.ref_start - cur.ref_start, dev = dor - dref;               // :197-219
    if (-QOT <= dor && dor <= QGT) {
      if ((int64_t)nxt.ref_start - cur.ref_end >= -ROT) {
        if (MINSV <= -dev && -dev <= MAXSV) return false;
        out = make_bnd(s, c1, cur.ref_start, 0, c1, nxt.ref_start, 1, read, hap); return true;
      } else if ((int64_t)cur.ref_start - nxt.ref_end >= -ROT) {
        if (MINSV <= dev && dev <= MAXSV) return false;
        out = make_bnd(s, c1, cur.ref_start, 0, c1, nxt.ref_start, 1, read, hap); return true;
      }
    }
    return false;
  }
  if (dor >= -QOT && dor <= QGT) {                                                              // :224-258
    if (cur.rev == nxt.rev)
      out = cur.rev ? make_bnd(s, c1, cur.ref_start, 0, c2, (int64_t)nxt.ref_end - 1, 0, read, hap)
                    : make_bnd(s, c1, (int64_t)cur.ref_end - 1, 1, c2, nxt.ref_start, 1, read, hap);
    else
      out = cur.rev ? make_bnd(s, c1, cur.ref_start, 0, c2, nxt.ref_start, 1, read, hap)
                    : make_bnd(s, c1, (int64_t)cur.ref_end - 1, 1, c2, (int64_t)nxt.ref_end - 1, 0, read, hap);
    return true;
  }
  return false;
}

__device__ __forceinline__ vsv_bnd dead_bnd() {
  vsv_bnd b; b.src_tid = 0; b.src_pos = 0; b.dst_tid = 0; b.dst_pos = 0; b.read = 0; b.read2 = 0; b.meta = VSV_B_DEAD; b.pad = 0;
  return b;
}

// one lane per split contig; slot of pair k of read r = seg_off[r] - r + k
__global__ __launch_bounds__(64) void segment_bnd(vsv_segments s, vsv_bnd_params p, vsv_bnd* __restrict__ out, uint32_t cap,
                                                  Counters* ctr) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= s.n_reads) return;
  const int64_t a = (int64_t)s.seg_off[r], b = (int64_t)s.seg_off[r + 1];
  const int n = (int)(b - a);
  if (n < 1 || n > BND_MAX_SEG) { atomicOr(&ctr->err, n < 1 ? ERRB_EMPTY_CIGAR : ERRB_RANGE); return; }
  const int64_t slot0 = a - r;
  if ((uint64_t)(slot0 + n - 1) > cap) { atomicOr(&ctr->err, ERRB_CAPACITY); return; }
  Seg sg[BND_MAX_SEG];
  for (int k = 0; k < n; ++k) {   // stable insertion sort by (q_start, q_end) (sorted() of SVIM_inter.py:83)
    Seg x;
    x.q_start = s.q_start[a + k]; x.q_end = s.q_end[a + k]; x.ref_id = s.ref_id[a + k]; x.ref_start = s.ref_start[a + k];
    x.ref_end = s.ref_end[a + k]; x.rev = s.is_reverse[a + k] ? 1 : 0;
    int j = k;
    while (j > 0 && (sg[j - 1].q_start > x.q_start || (sg[j - 1].q_start == x.q_start && sg[j - 1].q_end > x.q_end))) { sg[j] = sg[j - 1]; --j; }
    sg[j] = x;
  }
  const int hap = s.hap[r];
  for (int k = 0; k + 1 < n; ++k) {
    vsv_bnd x;
    out[slot0 + k] = bnd_of_pair(s, p, sg[k], sg[k + 1], (uint32_t)r, hap, x) ? x : dead_bnd();
  }
}
__global__ void bnd_set_count(vsv_segments s, uint32_t cap, Counters* ctr) {
  int64_t n = s.n_segs - s.n_reads;
  ctr->n_s1 = (uint32_t)(n < 0 ? 0 : (n > cap ? cap : n));
}

// key = [contig_rank | src_pos | hap]: stable sort == sorted(cands1 + cands2, key=(contig, pos)) of form_partitions
__global__ __launch_bounds__(256) void bnd_keys(const vsv_bnd* __restrict__ c, const int32_t* __restrict__ contig_rank,
                                                const uint32_t* __restrict__ d_n, uint64_t* __restrict__ key, uint32_t* __restrict__ idx) {
  const uint32_t n = *d_n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const vsv_bnd v = c[i];
    key[i] = (v.meta & VSV_B_DEAD) ? VSV_KEY_DEAD
                                   : ((uint64_t)(uint32_t)contig_rank[v.src_tid] << 33) | ((uint64_t)(uint32_t)v.src_pos << 1) | ((v.meta & VSV_B_HAP2) ? 1u : 0u);
    idx[i] = i;
  }
}
__global__ __launch_bounds__(256) void bnd_gather(const vsv_bnd* __restrict__ in, const uint32_t* __restrict__ idx,
                                                  const uint32_t* __restrict__ d_n, vsv_bnd* __restrict__ out) {
  const uint32_t n = *d_n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[idx[i]];
}
__global__ __launch_bounds__(256) void bnd_count_alive(const uint64_t* __restrict__ key, const uint32_t* __restrict__ d_n,
                                                       uint32_t* __restrict__ d_alive) {
  const uint32_t n = *d_n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if (key[i] != VSV_KEY_DEAD && (i + 1 == n || key[i + 1] == VSV_KEY_DEAD)) *d_alive = i + 1;
}

// one lane per partition (run of sorted candidates with the same contig and consecutive |d src_pos| <= max distance)
__global__ __launch_bounds__(256) void bnd_pair(const vsv_bnd* __restrict__ c, const uint32_t* __restrict__ d_n, vsv_bnd_params p,
                                                vsv_bnd* __restrict__ out) {
  const uint32_t n = *d_n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const vsv_bnd me = c[i];
    if (i > 0) {
      const vsv_bnd pv = c[i - 1];
      int64_t d = (int64_t)pv.src_pos - me.src_pos; if (d < 0) d = -d;
      if (pv.src_tid == me.src_tid && d <= p.partition_max_distance) continue;   // not a partition head (COMBINE.py:24-26)
    }
    uint32_t e = i + 1;
    while (e < n) {
      int64_t d = (int64_t)c[e - 1].src_pos - c[e].src_pos; if (d < 0) d = -d;
      if (c[e - 1].src_tid != c[e].src_tid || d > p.partition_max_distance) break;
      ++e;
    }
    const uint32_t sz = e - i;
    if (sz > (uint32_t)p.max_partition || sz > BND_MAX_PART) {                   // ignored (COMBINE.py:151-152)
      for (uint32_t k = i; k < e; ++k) out[k] = dead_bnd();
      continue;
    }
    int mate[BND_MAX_PART];
    for (uint32_t k = 0; k < sz; ++k) mate[k] = -1;
    for (;;) {   // complete linkage at 0.3 == greedy matching of hp1/hp2 pairs in ascending (|d1|+|d2|) <= 900
      int64_t best = -1; int bi = -1, bj = -1;
      for (uint32_t a = 0; a < sz; ++a)
        for (uint32_t b = a + 1; b < sz; ++b) {
          if (mate[a] >= 0 || mate[b] >= 0) continue;
          const vsv_bnd x = c[i + a], y = c[i + b];
          if (((x.meta ^ y.meta) & VSV_B_HAP2) == 0) continue;
          if ((x.meta ^ y.meta) & (VSV_B_SRC_FWD | VSV_B_DST_FWD)) continue;
          int64_t d1 = (int64_t)x.src_pos - y.src_pos, d2 = (int64_t)x.dst_pos - y.dst_pos;
          if (d1 < 0) d1 = -d1;
          if (d2 < 0) d2 = -d2;
          if (d1 + d2 > p.pair_distance) continue;
          if (best < 0 || d1 + d2 < best) { best = d1 + d2; bi = (int)a; bj = (int)b; }
        }
      if (bi < 0) break;
      mate[bi] = bj; mate[bj] = bi;
    }
    for (uint32_t k = 0; k < sz; ++k) {
      if (mate[k] >= 0 && mate[k] < (int)k) { out[i + k] = dead_bnd(); continue; }
      vsv_bnd v = c[i + k];
      const uint32_t gt = mate[k] >= 0 ? 3u : ((v.meta & VSV_B_HAP2) ? 2u : 1u);   // COMBINE.py:339-353
      if (mate[k] >= 0) v.read2 = c[i + mate[k]].read;
      v.meta = (v.meta & ~(3u << VSV_B_GT_SHIFT)) | (gt << VSV_B_GT_SHIFT);
      out[i + k] = v;
    }
  }
}

}  // namespace

void vsv_launch_bnd_segments(hipStream_t st, const vsv_segments& s, const vsv_bnd_params& p, vsv_bnd* cand, uint32_t cap, Counters* ctr) {
  if (s.n_reads > 0) segment_bnd<<<(unsigned)((s.n_reads + 63) / 64), 64, 0, st>>>(s, p, cand, cap, ctr);
  bnd_set_count<<<1, 1, 0, st>>>(s, cap, ctr);
}

void vsv_launch_bnd_pair(hipStream_t st, const vsv_bnd* cand, const int32_t* contig_rank, int rank_bits, const vsv_bnd_params& p,
                         vsv_bnd* sorted, vsv_bnd* calls, Counters* ctr, const StageBufs& b, const SortWork& sw, int64_t cap) {
  bnd_keys<<<128, 256, 0, st>>>(cand, contig_rank, &ctr->n_s1, b.key, b.idx);
  const SortResult r = vsv_radix_sort_pairs(st, b.key, b.idx, sw.key_alt, sw.val_alt, &ctr->n_s1, cap, 33 + rank_bits + 1, sw);
  bnd_gather<<<128, 256, 0, st>>>(cand, r.val, &ctr->n_s1, sorted);
  bnd_count_alive<<<128, 256, 0, st>>>(r.key, &ctr->n_s1, &ctr->n_alive1);
  bnd_pair<<<128, 256, 0, st>>>(sorted, &ctr->n_alive1, p, calls);
}
