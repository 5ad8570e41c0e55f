// support.hip — read-signature support of the calls: FP_filter_v1.eval_sig / compare_sigs
// (Large_INDEL/FP_filter_v1.py:87-123) as a sorted window join.
//
// The reference scans the whole read-signature list for every call and leaves at the first signature more than
// max_dist to the right; on the position-sorted list (RS:281-286) that is the window [pos-max_dist, pos+max_dist].
// One wave64 per call: two wave-uniform binary searches bound the window, the lanes test 64 signatures at a time and
// the count is a popcount of the ballot. Latency/L2-bound: 8 B per call + 8 B per window signature.
#include "vsv_device.h"

namespace {

// Wave-cooperative lower bound (first j with a[j] >= t): every round the 64 lanes probe 64 evenly spaced elements of the live
// range, so a 2 M-element list needs 4 dependent loads instead of 21 — the joins are bound by exactly that latency chain.
__device__ __forceinline__ int64_t wave_lower_bound(const int32_t* __restrict__ a, int64_t n, int64_t t, int lane) {
  int64_t lo = 0, hi = n;                                   // answer in [lo, hi]
  while (hi - lo > 64) {
    const int64_t step = (hi - lo + 63) / 64;
    const int64_t p = lo + (int64_t)lane * step;
    const bool ge = p >= hi || (int64_t)a[p] >= t;
    const uint64_t m = __ballot(ge);
    if (m & 1ull) return lo;                                // a[lo] >= t
    const int f = m ? __builtin_ctzll(m) : 64;              // first probe that is >= t (64: none)
    const int64_t new_lo = lo + (int64_t)(f - 1) * step + 1;
    if (f < 64) { const int64_t ph = lo + (int64_t)f * step; hi = ph < hi ? ph : hi; }
    lo = new_lo;
  }
  const int64_t p = lo + lane;
  const bool lt = p < hi && (int64_t)a[p] < t;
  return lo + (int64_t)__popcll(__ballot(lt));
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
    const int64_t lo = wave_lower_bound(spos, ns, pos1 - (int64_t)p.max_dist, lane);          // shift < -max_dist: continue
    const int64_t hi = wave_lower_bound(spos, ns, pos1 + (int64_t)p.max_dist + 1, lane);      // shift >  max_dist: break
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

// ---- calculate_signature_support.py ----------------------------------------------------------------------------------
__device__ __forceinline__ int64_t wave_sum_i64(int64_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// calc_ins_call_cov (CS:81-125): the reference adds the summed length of every distinct signature position to every call
// bin [pos-flanking, pos+flanking] that contains it; per call that is the window sum below.
__global__ __launch_bounds__(256) void cov_ins(const int32_t* __restrict__ cpos, int64_t nc, const int32_t* __restrict__ spos,
                                               const int32_t* __restrict__ slen, int64_t ns, int32_t flanking,
                                               int64_t* __restrict__ cov, uint32_t* __restrict__ err) {
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (int64_t j = gtid; j + 1 < ns; j += gsz) bad |= spos[j] > spos[j + 1];
  if (bad) atomicOr(err, ERRB_UNSORTED);
  const int lane = threadIdx.x & 63;
  for (int64_t i = gtid >> 6; i < nc; i += gsz >> 6) {
    const int64_t pos = cpos[i];
    const int64_t lo = wave_lower_bound(spos, ns, pos - (int64_t)flanking, lane);        // lb > pos: break (CS:105-106)
    const int64_t hi = wave_lower_bound(spos, ns, pos + (int64_t)flanking + 1, lane);    // pos <= rb (CS:107)
    int64_t sum = 0;
    for (int64_t j = lo + lane; j < hi; j += 64) sum += (int64_t)slen[j];
    sum = wave_sum_i64(sum);
    if (lane == 0) cov[i] = sum;
  }
}

__global__ __launch_bounds__(256) void cov_del_span(const int32_t* __restrict__ sstart, const int32_t* __restrict__ send, int64_t ns,
                                                    uint32_t* __restrict__ err) {
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  uint32_t span = 0;
  for (int64_t j = gtid; j < ns; j += gsz) {
    if (j + 1 < ns) bad |= sstart[j] > sstart[j + 1];
    const int64_t d = (int64_t)send[j] - (int64_t)sstart[j];
    span = max(span, (uint32_t)(d < 0 ? 0 : (d > 0x7FFFFFFF ? 0x7FFFFFFF : d)));
  }
  if (bad) atomicOr(&err[0], ERRB_UNSORTED);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) span = max(span, (uint32_t)__shfl_xor((int)span, d, 64));
  if ((threadIdx.x & 63) == 0 && span) atomicMax(&err[1], span);
}

// calc_del_call_cov (CS:138-280): signature j supports call region [lb, rb] when its start or its end lies inside the
// region (CS:171-200) or when lb or rb lies inside [start, end] (CS:204-241) — i.e. the closed intervals meet — and is
// counted once (set(), CS:247). Signatures ascend by start, so candidates are those with start in [lb - max_span, rb].
__global__ __launch_bounds__(256) void cov_del(const int32_t* __restrict__ cstart, const int32_t* __restrict__ cend, int64_t nc,
                                               const int32_t* __restrict__ sstart, const int32_t* __restrict__ send,
                                               const int32_t* __restrict__ svlen, int64_t ns, int32_t flanking,
                                               int64_t* __restrict__ cov, const uint32_t* __restrict__ err) {
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (int64_t)gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  const int64_t max_span = err[1];
  for (int64_t i = gtid >> 6; i < nc; i += gsz >> 6) {
    const int64_t lb = (int64_t)cstart[i] - flanking, rb = (int64_t)cend[i] + flanking;   // CS:154
    const int64_t lo = wave_lower_bound(sstart, ns, lb - max_span, lane);
    const int64_t hi = wave_lower_bound(sstart, ns, rb + 1, lane);
    int64_t sum = 0;
    for (int64_t j = lo + lane; j < hi; j += 64)
      if ((int64_t)send[j] >= lb) sum += (int64_t)svlen[j];
    sum = wave_sum_i64(sum);
    if (lane == 0) cov[i] = sum;
  }
}

}  // namespace

static int cov_blocks(int64_t n_calls) {
  int64_t blocks = (n_calls + 3) / 4;
  return (int)(blocks > 8192 ? 8192 : (blocks < 1 ? 1 : blocks));
}
void vsv_launch_cov_ins(hipStream_t st, const int32_t* call_pos, int64_t n_calls, const int32_t* sig_pos, const int32_t* sig_len,
                        int64_t n_sigs, int32_t flanking, int64_t* cov, uint32_t* err) {
  if (n_calls <= 0) return;
  cov_ins<<<cov_blocks(n_calls), 256, 0, st>>>(call_pos, n_calls, sig_pos, sig_len, n_sigs, flanking, cov, err);
}
void vsv_launch_cov_del(hipStream_t st, const int32_t* call_start, const int32_t* call_end, int64_t n_calls, const int32_t* sig_start,
                        const int32_t* sig_end, const int32_t* sig_svlen, int64_t n_sigs, int32_t flanking, int64_t* cov, uint32_t* err) {
  if (n_calls <= 0) return;
  if (n_sigs > 0) cov_del_span<<<(int)((n_sigs + 255) / 256 > 2048 ? 2048 : (n_sigs + 255) / 256), 256, 0, st>>>(sig_start, sig_end, n_sigs, err);
  cov_del<<<cov_blocks(n_calls), 256, 0, st>>>(call_start, call_end, n_calls, sig_start, sig_end, sig_svlen, n_sigs, flanking, cov, err);
}

void vsv_launch_support_join(hipStream_t st, const int32_t* call_pos, const int32_t* call_len, int64_t n_calls, const int32_t* sig_pos,
                             const int32_t* sig_len, int64_t n_sigs, const vsv_support_params& p, uint32_t* support, uint32_t* err) {
  if (n_calls <= 0) return;
  int64_t blocks = (n_calls + 3) / 4;          // 4 waves per block, one call per wave
  if (blocks > 8192) blocks = 8192;
  support_join<<<(int)blocks, 256, 0, st>>>(call_pos, call_len, n_calls, sig_pos, sig_len, n_sigs, p, support, err);
}
