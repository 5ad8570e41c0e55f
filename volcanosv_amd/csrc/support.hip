// support.hip — read-signature support of the calls: FP_filter_v1.eval_sig / compare_sigs
// (Large_INDEL/FP_filter_v1.py:87-123) as a sorted window join.
//
// The reference scans the whole read-signature list for every call and leaves at the first signature more than
// max_dist to the right; on the position-sorted list (RS:281-286) that is the window [pos-max_dist, pos+max_dist].
// One wave64 per call: two wave-uniform binary searches bound the window, the lanes test 64 signatures at a time and
// the count is a popcount of the ballot. Latency/L2-bound: 8 B per call + 8 B per window signature.
#include "vsv_device.h"

namespace {

__device__ __forceinline__ int64_t lower_bound_pos(const int32_t* __restrict__ a, int64_t n, int64_t t) {   // first a[j] >= t
  int64_t lo = 0, hi = n;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)a[mid] >= t) hi = mid; else lo = mid + 1; }
  return lo;
}

__global__ __launch_bounds__(256) void support_join(const int32_t* __restrict__ cpos, const int32_t* __restrict__ clen, int64_t nc,
                                                    const int32_t* __restrict__ spos, const int32_t* __restrict__ slen, int64_t ns,
                                                    vsv_support_params p, uint32_t* __restrict__ support, uint32_t* __restrict__ err) {
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
  // the early `break` of FP:118-119 equals the window only on an ascending list
  bool bad = false;
  for (int64_t j = gtid; j + 1 < ns; j += gsz) bad |= spos[j] > spos[j + 1];
  if (bad) atomicOr(err, ERRB_UNSORTED);

  const int lane = threadIdx.x & 63;
  const int64_t nwaves = gsz >> 6;
  for (int64_t i = gtid >> 6; i < nc; i += nwaves) {
    const int32_t len1 = clen[i];
    if (len1 > p.max_comp_svlen) { if (lane == 0) support[i] = 60u; continue; }      // FP:110-111
    const int64_t pos1 = cpos[i];
    const int64_t lo = lower_bound_pos(spos, ns, pos1 - (int64_t)p.max_dist);          // shift < -max_dist: continue
    const int64_t hi = lower_bound_pos(spos, ns, pos1 + (int64_t)p.max_dist + 1);      // shift >  max_dist: break
    uint32_t cnt = 0;
    for (int64_t j0 = lo; j0 < hi; j0 += 64) {
      const int64_t j = j0 + lane;
      bool ok = false;
      if (j < hi) {
        int64_t shift = (int64_t)spos[j] - pos1;
        if (shift < 0) shift = -shift;
        const int32_t len2 = slen[j];
        const int32_t mn = len1 < len2 ? len1 : len2, mx = len1 < len2 ? len2 : len1;
        const double sim = mx == 0 ? 0.0 : (double)mn / (double)mx;                    // FP:93-96 (ZeroDivisionError -> 0)
        ok = shift <= (int64_t)p.max_shift && sim >= p.min_size_sim;                  // FP:98
      }
      cnt += (uint32_t)__popcll(__ballot(ok));
    }
    if (lane == 0) support[i] = cnt;
  }
}

}  // namespace

void vsv_launch_support_join(hipStream_t st, const int32_t* call_pos, const int32_t* call_len, int64_t n_calls, const int32_t* sig_pos,
                             const int32_t* sig_len, int64_t n_sigs, const vsv_support_params& p, uint32_t* support, uint32_t* err) {
  if (n_calls <= 0) return;
  int64_t blocks = (n_calls + 3) / 4;          // 4 waves per block, one call per wave
  if (blocks > 8192) blocks = 8192;
  support_join<<<(int)blocks, 256, 0, st>>>(call_pos, call_len, n_calls, sig_pos, sig_len, n_sigs, p, support, err);
}
