// cutesv.hip — split-read branch of Large_INDEL/sig_extract.py: analysis_split_read (SE:193-319), INS/DEL candidates.
//
// One lane per primary read with an SA tag: the read's segments (primary first, then the SA entries in tag order,
// organize_split_signal SE:341-371) are ordered by read start with a stable insertion sort (sorted(), SE:198), then the
// two-segment rule (SE:206-236) or the triple scan (SE:242-296) and the INS-in-translocation rule (SE:298-319) run exactly
// as written; the TRA/BND candidates (analysis_bnd) are not produced — the collector keeps only `grep -w INS|DEL` lines
// (SE:637-638). Latency-bound, a few dozen integers per read. Count pass -> exclusive scan -> write pass, so the table is dense
// and in (read, emission) order whatever the lane order.
#include "vsv_device.h"

namespace {

struct CSeg { int64_t qs, qe, rs, re; int32_t chr; int32_t rev; };

__device__ __forceinline__ CSeg cflip(const CSeg& x, int64_t rl) { CSeg y = x; y.qs = rl - x.qe; y.qe = rl - x.qs; return y; }   // SE:213-214

struct CEmit {
  vsv_sig* out; uint32_t base, n, cap_local; uint32_t rec;   // out == nullptr: counting pass
  __device__ void put(bool del, int64_t pos, int64_t len, int64_t s0, int64_t s1, int32_t chr, int qrev) {
    if (!out) { ++n; return; }
    if (n >= cap_local) return;
    vsv_sig s;
    s.pos = (int32_t)pos; s.svlen = (int32_t)len; s.q_start = (int32_t)s0; s.q_end = (int32_t)s1;
    s.rec = rec; s.rec2 = 0xFFFFFFFFu;
    s.meta = (del ? VSV_M_DEL : 0u) | VSV_M_SPLIT | (qrev ? VSV_M_QREV : 0u);
    s.tid = chr;
    out[base + n++] = s;
  }
};

// the INS / DEL test between two consecutive segments (SE:216-235, 258-276, 281-296); e3 = the following segment of a
// triple (its start must not precede e2's end, SE:261, 270) or nullptr
__device__ void pair_rule(const CSeg& e1, const CSeg& e2, const CSeg* e3, int64_t sv, int64_t mx, int qrev, CEmit& em) {
  if (e1.re - e2.rs >= sv) return;
  const int64_t ins_len = e2.qs + e1.re - e2.rs - e1.qe;
  if (ins_len >= sv && e2.rs - e1.re <= 100 && (ins_len <= mx || mx == -1) && (!e3 || e3->rs >= e2.re)) {
    const int64_t half = (e2.rs - e1.re) / 2;                       // int(x / 2): truncation toward zero
    em.put(false, (e2.rs + e1.re) / 2, ins_len, e1.qe + half, e2.qs - half, e2.chr, qrev);
  }
  const int64_t del_len = e2.rs - e2.qs + e1.qe - e1.re;
  if (del_len >= sv && e2.qs - e1.qe <= 100 && (del_len <= mx || mx == -1) && (!e3 || e3->rs >= e2.re))
    em.put(true, e1.re, del_len, 0, 0, e2.chr, 0);
}

constexpr int CS_MAX_SEG = 64;

// Two passes: WRITE = false counts the rows of every read (cnt[r]); after an exclusive scan WRITE = true stores them at off[r],
// so the table is dense and in (read, emission) order.
template <bool WRITE>
__global__ __launch_bounds__(64) void cutesv_split(vsv_segments sg, const int32_t* __restrict__ read_len, const uint32_t* __restrict__ read_rec,
                                                   int sv_size, int max_size, int max_parts, vsv_sig* __restrict__ out, uint32_t cap,
                                                   uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off, Counters* ctr,
                                                   uint8_t* __restrict__ tra) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= sg.n_reads) return;
  // tra[r]: the read yields a translocation candidate (analysis_bnd reached with the two segments <= 100 read bases apart,
  // SE:109): not an INS/DEL row, but it makes the read's task non-empty (SE:533-535: its reads go to reads.sigs)
  uint8_t has_tra = 0;
  const uint64_t a = sg.seg_off[r], b = sg.seg_off[r + 1];
  const uint32_t n = (uint32_t)(b - a);
  if (!WRITE) { cnt[r] = 0; tra[r] = 0; }
  if (!(n <= (uint32_t)max_parts || max_parts == -1)) return;         // SE:370
  if (n > CS_MAX_SEG) { atomicOr(&ctr->err, ERRB_RANGE); return; }
  const uint64_t base64 = WRITE ? off[r] : 0, slots = WRITE ? cnt[r] : 0xFFFFFFFFull;
  if (WRITE && base64 + slots > cap) { atomicOr(&ctr->err, ERRB_CAPACITY); return; }
  uint8_t idx[CS_MAX_SEG];
  for (uint32_t k = 0; k < n; ++k) {                                   // stable insertion sort by read start (SE:198)
    const int32_t key = sg.q_start[a + k];
    uint32_t j = k;
    while (j > 0 && sg.q_start[a + idx[j - 1]] > key) { idx[j] = idx[j - 1]; --j; }
    idx[j] = (uint8_t)k;
  }
  auto seg = [&](uint32_t k) {
    const uint64_t i = a + idx[k];
    CSeg s; s.qs = sg.q_start[i]; s.qe = sg.q_end[i]; s.rs = sg.ref_start[i]; s.re = sg.ref_end[i]; s.chr = sg.ref_id[i]; s.rev = sg.is_reverse[i];
    return s;
  };
  const int64_t rl = read_len[r], sv = sv_size, mx = max_size;
  CEmit em{WRITE ? out : nullptr, (uint32_t)base64, 0u, (uint32_t)slots, read_rec[r]};
  int trigger = 0, qrev = 0;
  if (n == 2) {
    CSeg e1 = seg(0), e2 = seg(1);
    if (e1.chr == e2.chr) {
      if (e1.rev == e2.rev) {
        if (e1.rev) { const CSeg t1 = cflip(seg(1), rl), t2 = cflip(seg(0), rl); e1 = t1; e2 = t2; qrev ^= 1; }   // SE:212-215
        pair_rule(e1, e2, nullptr, sv, mx, qrev, em);
      }
    } else { trigger = 1; if (e2.qs - e1.qe <= 100) has_tra = 1; }                                                  // SE:233-235: analysis_bnd
  } else if (n >= 3) {
    for (uint32_t k = 0; k + 2 < n; ++k) {
      CSeg e1 = seg(k), e2 = seg(k + 1), e3 = seg(k + 2);
      if (e1.chr == e2.chr) {
        if (e2.chr == e3.chr && e1.rev == e3.rev && e1.rev == e2.rev) {
          if (e1.rev) { e1 = cflip(seg(k + 2), rl); e2 = cflip(seg(k + 1), rl); e3 = cflip(seg(k), rl); qrev ^= 1; }  // SE:250-254
          pair_rule(e1, e2, &e3, sv, mx, qrev, em);
          if (n - 3 == k) pair_rule(e2, e3, nullptr, sv, mx, qrev, em);                                              // SE:277-296
        }
      } else {                                                                                                        // SE:298-302: analysis_bnd
        trigger = 1;
        if (e2.qs - e1.qe <= 100) has_tra = 1;
        if (n - 3 == k && e2.chr != e3.chr && e3.qs - e2.qe <= 100) has_tra = 1;
      }
    }
    if (trigger) {                                                                                                    // SE:305-319
      const CSeg f = seg(0), l = seg(n - 1);
      if (f.chr == l.chr && f.rev == l.rev) {
        CSeg e1 = f, e2 = l;
        if (f.rev) { e1 = cflip(l, rl); e2 = cflip(f, rl); qrev ^= 1; }
        const int64_t dis_ref = e2.rs - e1.re, dis_read = e2.qs - e1.qe;
        if (dis_ref < 100 && dis_read - dis_ref >= sv && (dis_read - dis_ref <= mx || mx == -1))
          em.put(false, e2.rs < e1.re ? e2.rs : e1.re, dis_read - dis_ref, e1.qe + dis_ref / 2, e2.qs - dis_ref / 2, e2.chr, qrev);
      }
    }
  }
  if (!WRITE) { cnt[r] = em.n; tra[r] = has_tra; }
}

}  // namespace

__global__ void cutesv_set_rows(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ off, int64_t n_reads, Counters* ctr) {
  ctr->n_alive2 = off[n_reads - 1] + cnt[n_reads - 1];
}

void vsv_launch_cutesv_split(hipStream_t st, const vsv_segments& sg, const int32_t* read_len, const uint32_t* read_rec, int sv_size,
                             int max_size, int max_parts, vsv_sig* out, uint32_t cap, uint32_t* cnt, uint32_t* off, uint32_t* scan_tmp,
                             Counters* ctr, uint8_t* tra) {
  if (sg.n_reads <= 0) return;
  const int grid = (int)((sg.n_reads + 63) / 64);
  cutesv_split<false><<<grid, 64, 0, st>>>(sg, read_len, read_rec, sv_size, max_size, max_parts, out, cap, cnt, off, ctr, tra);
  vsv_scan_u32_exclusive(st, cnt, (int)sg.n_reads, off, scan_tmp);
  cutesv_split<true><<<grid, 64, 0, st>>>(sg, read_len, read_rec, sv_size, max_size, max_parts, out, cap, cnt, off, ctr, tra);
  cutesv_set_rows<<<1, 1, 0, st>>>(cnt, off, sg.n_reads, ctr);
}
