// redundancy.hip — remove_redundancy.py (Large_INDEL): candidate pairs inside the position window and the pair predicates
// (RR:92-134, 162-181). DEL pairs are decided by arithmetic; INS pairs need the global unit-cost edit distance of the two ALT
// strings (edlib.align, RR:75-81), computed here with Myers' bit-parallel recurrence, one wavefront per pair:
//   lane l owns 64 rows of the DP matrix (a 64-bit block of the shorter string); lanes run skewed by one column, so the
//   horizontal delta a block needs from the block above arrives by a one-lane shuffle from the previous step; the text
//   character travels down the lanes the same way. Patterns longer than 64 x 64 = 4096 symbols are processed in
//   super-blocks of 64 lanes; the deltas leaving the bottom of a super-block are parked in a byte array for the next one.
// Work per pair: (n + 63) steps of ~30 instructions per super-block — latency/ALU-bound, no HBM traffic to speak of.
#include "vsv_device.h"

namespace {

__device__ __forceinline__ bool size_ok(int64_t l1, int64_t l2, double thr) {
  const int64_t mn = l1 < l2 ? l1 : l2, mx = l1 < l2 ? l2 : l1;
  return (double)mn / (double)mx >= thr;                                   // get_size_sim (RR:86-90)
}
__device__ __forceinline__ bool del_match(int64_t p1, int64_t l1, int64_t p2, int64_t l2, double overlap_thr) {
  const int64_t e1 = p1 + l1, e2 = p2 + l2, mx = l1 < l2 ? l2 : l1;
  const int64_t ov = (e1 < e2 ? e1 : e2) - (p1 > p2 ? p1 : p2);
  return (double)ov / (double)mx >= overlap_thr;                           // get_reciprocal_overlap (RR:103-112)
}

// pass 0: count, pass 1: write (i, j) at off[i]. DEL pairs are final; INS pairs are the candidates of the edit-distance kernel.
template <bool WRITE>
__global__ __launch_bounds__(256) void rr_pairs(const int32_t* __restrict__ pos, const int32_t* __restrict__ svlen, int64_t n, int is_del,
                                                int64_t dist, double size_thr, double overlap_thr, uint32_t* __restrict__ cnt,
                                                const uint32_t* __restrict__ off, uint32_t* __restrict__ pairs, uint32_t cap) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p1 = pos[i], l1 = svlen[i];
    uint32_t c = 0, o = WRITE ? off[i] : 0;
    for (int64_t j = i + 1; j < n && (int64_t)pos[j] <= p1 + dist; ++j) {      // RR:118-126
      const int64_t l2 = svlen[j];
      if (!size_ok(l1, l2, size_thr)) continue;
      if (is_del && !del_match(p1, l1, pos[j], l2, overlap_thr)) continue;
      if (WRITE) { if (o + c < cap) { pairs[2 * (size_t)(o + c)] = (uint32_t)i; pairs[2 * (size_t)(o + c) + 1] = (uint32_t)j; } }
      ++c;
    }
    if (!WRITE) cnt[i] = c;
  }
}

constexpr int ED_WAVES = 4, ED_SYMS = 16;

// one wave per candidate pair: flag[p] = 1 if (la + lb - editDistance) / (la + lb) >= seq_sim_thr
__global__ __launch_bounds__(64 * ED_WAVES) void rr_edit_sim(const uint32_t* __restrict__ pairs, uint32_t n_pairs, const uint8_t* __restrict__ seq,
                                                             const uint64_t* __restrict__ seq_off, const uint64_t* __restrict__ hoff,
                                                             int32_t* __restrict__ hbuf, double seq_sim_thr, uint8_t* __restrict__ flag,
                                                             uint32_t* __restrict__ dist_out) {
  __shared__ uint64_t peq[ED_WAVES][ED_SYMS][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t p = blockIdx.x * ED_WAVES + wv;
  if (p >= n_pairs) return;
  const uint32_t i = pairs[2 * (size_t)p], j = pairs[2 * (size_t)p + 1];
  const uint8_t* a = seq + seq_off[i]; int64_t la = (int64_t)(seq_off[i + 1] - seq_off[i]);
  const uint8_t* b = seq + seq_off[j]; int64_t lb = (int64_t)(seq_off[j + 1] - seq_off[j]);
  const uint8_t* pat = la <= lb ? a : b; const int64_t m = la <= lb ? la : lb;      // pattern = the shorter string
  const uint8_t* txt = la <= lb ? b : a; const int64_t n = la <= lb ? lb : la;
  int64_t score = m;
  if (m > 0) {
    const int64_t W = (m + 63) / 64;
    int32_t* hb0 = hbuf + hoff[p];        // only pairs with W > 64 own a slice (two arrays of n deltas)
    int32_t* hb1 = hb0 + n;
    for (int64_t sb = 0; sb * 64 < W; ++sb) {
      const int64_t blk = sb * 64 + lane;
      const bool active = blk < W;
      const int64_t nb = (W - sb * 64) < 64 ? (W - sb * 64) : 64;
      for (int c = 0; c < ED_SYMS; ++c) peq[wv][c][lane] = 0;
      if (active)
        for (int k = 0; k < 64; ++k) {
          const int64_t idx = blk * 64 + k;
          if (idx < m) peq[wv][pat[idx] & (ED_SYMS - 1)][lane] |= 1ull << k;
        }
      const int bit = (blk == W - 1) ? (int)((m - 1) & 63) : 63;
      const bool is_last = blk == W - 1, is_bottom = lane == nb - 1 && !is_last;
      const int32_t* hin_buf = (sb & 1) ? hb1 : hb0;     // written by the previous super-block
      int32_t* hout_buf = (sb & 1) ? hb0 : hb1;
      uint64_t Pv = ~0ull, Mv = 0;
      int carry = 0;                                     // packed (char | (hout + 1) << 8) handed down the lanes
      for (int64_t t = 0; t < n + nb - 1; ++t) {
        const int up = __shfl_up(carry, 1, 64);
        int ch, hin;
        if (lane == 0) {
          ch = t < n ? txt[t] & (ED_SYMS - 1) : 0;
          hin = sb == 0 ? 1 : (t < n ? __hip_atomic_load(&hin_buf[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0);          // D[0][j] = j: the top row rises by one per column
        } else { ch = up & 0xFF; hin = ((up >> 8) & 3) - 1; }
        int hout = 0;
        const int64_t col = t - lane;
        if (active && col >= 0 && col < n) {
          uint64_t Eq = peq[wv][ch][lane];
          const uint64_t Xv = Eq | Mv;
          if (hin < 0) Eq |= 1ull;
          const uint64_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
          uint64_t Ph = Mv | ~(Xh | Pv), Mh = Pv & Xh;
          hout = ((Ph >> bit) & 1ull) ? 1 : (((Mh >> bit) & 1ull) ? -1 : 0);
          Ph <<= 1; Mh <<= 1;
          if (hin < 0) Mh |= 1ull; else if (hin > 0) Ph |= 1ull;
          Pv = Mh | ~(Xv | Ph);
          Mv = Ph & Xv;
          if (is_last) score += hout;
          if (is_bottom) __hip_atomic_store(&hout_buf[col], hout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // L2, not the (stale-able) L1
        }
        carry = ch | ((hout + 1) << 8);
      }
      if (W > 64) __threadfence();                        // the parked deltas are visible before the next super-block reads them
    }
    // the lane that owns the last block holds the score
    const int owner = (int)((W - 1) & 63);
    score = __shfl(score, owner, 64);
  } else score = n;
  if (lane == 0) {
    const double tot = (double)(la + lb);
    flag[p] = ((tot - (double)score) / tot >= seq_sim_thr) ? 1 : 0;        // edit_sim (RR:75-81) >= seq_sim_thresh (RR:98-100)
    dist_out[p] = (uint32_t)score;
  }
}

}  // namespace

void vsv_launch_rr_pairs(hipStream_t st, bool write, const int32_t* pos, const int32_t* svlen, int64_t n, int is_del, int64_t dist,
                         double size_thr, double overlap_thr, uint32_t* cnt, const uint32_t* off, uint32_t* pairs, uint32_t cap) {
  if (n <= 0) return;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  if (write) rr_pairs<true><<<grid, 256, 0, st>>>(pos, svlen, n, is_del, dist, size_thr, overlap_thr, cnt, off, pairs, cap);
  else rr_pairs<false><<<grid, 256, 0, st>>>(pos, svlen, n, is_del, dist, size_thr, overlap_thr, cnt, off, pairs, cap);
}

void vsv_launch_rr_edit_sim(hipStream_t st, const uint32_t* pairs, uint32_t n_pairs, const uint8_t* seq, const uint64_t* seq_off,
                            const uint64_t* hoff, int32_t* hbuf, double seq_sim_thr, uint8_t* flag, uint32_t* dist_out) {
  if (!n_pairs) return;
  rr_edit_sim<<<(n_pairs + ED_WAVES - 1) / ED_WAVES, 64 * ED_WAVES, 0, st>>>(pairs, n_pairs, seq, seq_off, hoff, hbuf, seq_sim_thr, flag, dist_out);
}
