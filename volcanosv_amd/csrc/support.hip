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

// ---- correct_gt_{del,ins}_real_data.py ------------------------------------------------------------------------------------------
// Variant-signature support (match_varlist_siglist DG:92-137 / extract_sig_support IG:105-156): the distinct read signatures
// (chrom, pos, svlen, count) are listed chromosome block by chromosome block with ascending positions; a variant's window is
// [pos - max_shift, pos + max_shift] with max_shift = max(svlen * ratio, 500), all in floating point like the reference.
// One wave per variant: bounds by wave-cooperative search inside the variant's block, then the size-filtered count sum.
// The reference's "resume at the last match" bookkeeping (it makes the element at the previous window's first index count
// twice) needs the windows of all variants in order and is applied by the caller from lo / hi.
namespace {

__device__ __forceinline__ int64_t wave_first_ge_f(const int32_t* __restrict__ a, int64_t lo, int64_t hi, double t, bool strict, int lane) {
  // first j in [lo, hi) with a[j] >= t (strict: a[j] > t); hi if none. Same 64-ary scheme as wave_lower_bound.
  while (hi - lo > 64) {
    const int64_t step = (hi - lo + 63) / 64;
    const int64_t p = lo + (int64_t)lane * step;
    const bool ge = p >= hi || (strict ? (double)a[p] > t : (double)a[p] >= t);
    const uint64_t m = __ballot(ge);
    if (m & 1ull) return lo;
    const int f = m ? __builtin_ctzll(m) : 64;
    const int64_t new_lo = lo + (int64_t)(f - 1) * step + 1;
    if (f < 64) { const int64_t ph = lo + (int64_t)f * step; hi = ph < hi ? ph : hi; }
    lo = new_lo;
  }
  const int64_t p = lo + lane;
  const bool lt = p < hi && (strict ? !((double)a[p] > t) : !((double)a[p] >= t));
  return lo + (int64_t)__popcll(__ballot(lt));
}

__global__ __launch_bounds__(256) void gt_support(const int32_t* __restrict__ vpos, const int32_t* __restrict__ vlen, const int32_t* __restrict__ blo,
                                                  const int32_t* __restrict__ bhi, int64_t nv, const int32_t* __restrict__ spos,
                                                  const int32_t* __restrict__ slen, const int32_t* __restrict__ scnt, double shift_ratio,
                                                  double size_sim, int64_t* __restrict__ sum, int32_t* __restrict__ lo_out, int32_t* __restrict__ hi_out) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t i = gw; i < nv; i += nw) {
    const double svlen = (double)vlen[i], pos = (double)vpos[i];
    const double prod = svlen * shift_ratio;
    const double max_shift = prod > 500.0 ? prod : 500.0;                 // max(svlen * ratio, 500)
    const double min_pos = pos - max_shift, max_pos = pos + max_shift;
    const double min_size = svlen * size_sim, max_size = svlen / size_sim;
    const int64_t lo = wave_first_ge_f(spos, blo[i], bhi[i], min_pos, false, lane);
    const int64_t hi = wave_first_ge_f(spos, lo, bhi[i], max_pos, true, lane);
    int64_t t = 0;
    for (int64_t j = lo + lane; j < hi; j += 64) {
      const double l = (double)slen[j];
      if (min_size <= l && l <= max_size) t += scnt[j];
    }
    t = wave_sum_i64(t);
    if (lane == 0) { sum[i] = t; lo_out[i] = (int32_t)lo; hi_out[i] = (int32_t)hi; }
  }
}

// reference end (pos + M/D/N/=/X lengths, pysam reference_end) of every record: one lane per record
__global__ __launch_bounds__(256) void ref_end_kernel(RecView rv, int32_t* __restrict__ rend, uint32_t* __restrict__ max_span) {
  uint32_t ms = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rv.n_records; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t span = 0;
    uint64_t a, b;
    vsv_op_range(rv, i, a, b);
    for (uint64_t k = a; k < b; ++k) {
      const uint32_t w = rv.cigar[k], op = w & 15u;
      if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += w >> 4;
    }
    rend[i] = (int32_t)(rv.pos[i] + span);
    ms = max(ms, (uint32_t)(span > 0x7FFFFFFF ? 0x7FFFFFFF : span));
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) ms = max(ms, (uint32_t)__shfl_xor((int)ms, d, 64));
  if ((threadIdx.x & 63) == 0 && ms) atomicMax(max_span, ms);
}

// first record index with (tid, pos) >= (qt, qp), records ascending by (tid, pos)
__device__ __forceinline__ int64_t wave_lower_bound_tp(const int32_t* __restrict__ tid, const int32_t* __restrict__ pos, int64_t n, int32_t qt, int64_t qp, int lane) {
  int64_t lo = 0, hi = n;
  auto ge = [&](int64_t p) { const int32_t t = tid[p]; return t > qt || (t == qt && (int64_t)pos[p] >= qp); };
  while (hi - lo > 64) {
    const int64_t step = (hi - lo + 63) / 64;
    const int64_t p = lo + (int64_t)lane * step;
    const uint64_t m = __ballot(p >= hi || ge(p));
    if (m & 1ull) return lo;
    const int f = m ? __builtin_ctzll(m) : 64;
    const int64_t new_lo = lo + (int64_t)(f - 1) * step + 1;
    if (f < 64) { const int64_t ph = lo + (int64_t)f * step; hi = ph < hi ? ph : hi; }
    lo = new_lo;
  }
  const int64_t p = lo + lane;
  return lo + (int64_t)__popcll(__ballot(p < hi && !ge(p)));
}

// count_reads_span_region (DG:140-147) / check_full_cover_reads (IG:178-186): reads of chromosome qt with
// reference_start < a and reference_end > b. Candidates start in [a - longest span, a).
__global__ __launch_bounds__(256) void span_count(RecView rv, const int32_t* __restrict__ rend, const uint32_t* __restrict__ max_span,
                                                  const int32_t* __restrict__ qt, const int32_t* __restrict__ qa, const int32_t* __restrict__ qb,
                                                  int64_t nq, uint32_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const int64_t ms = *max_span;
  for (int64_t i = gw; i < nq; i += nw) {
    const int64_t a = qa[i], b = qb[i];
    const int64_t lo = wave_lower_bound_tp(rv.tid, rv.pos, rv.n_records, qt[i], a - ms, lane);
    const int64_t hi = wave_lower_bound_tp(rv.tid, rv.pos, rv.n_records, qt[i], a, lane);        // start < a
    uint32_t c = 0;
    for (int64_t j = lo + lane; j < hi; j += 64) c += ((int64_t)rend[j] > b && !(rv.flag[j] & VSV_F_UNMAPPED)) ? 1u : 0u;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d, 64);
    if (lane == 0) out[i] = c;
  }
}

}  // namespace

void vsv_launch_gt_support(hipStream_t st, const int32_t* vpos, const int32_t* vlen, const int32_t* blo, const int32_t* bhi, int64_t nv, const int32_t* spos,
                           const int32_t* slen, const int32_t* scnt, double shift_ratio, double size_sim, int64_t* sum, int32_t* lo, int32_t* hi) {
  if (nv <= 0) return;
  gt_support<<<cov_blocks(nv), 256, 0, st>>>(vpos, vlen, blo, bhi, nv, spos, slen, scnt, shift_ratio, size_sim, sum, lo, hi);
}
void vsv_launch_span_count(hipStream_t st, const RecView& rv, int32_t* rend, uint32_t* max_span, const int32_t* qt, const int32_t* qa, const int32_t* qb,
                           int64_t nq, uint32_t* out) {
  if (rv.n_records > 0) ref_end_kernel<<<2048, 256, 0, st>>>(rv, rend, max_span);
  if (nq > 0) span_count<<<cov_blocks(nq), 256, 0, st>>>(rv, rend, max_span, qt, qa, qb, nq, out);
}
