// sig_stages.hip — everything downstream of the CIGAR scan, on <= a few % of the input bytes:
//   fold_kernel        cluster_ins_one_read / cluster_del_one_read   (Hifi.py:91-161)
//   clr_gate_records   ins_pct / var_dist gate                        (CLR.py:53-70, 425-427)
//   split_*            extract_sig_from_split_reads + extract_sig_from_split
//                      (Hifi.py:307-371, 421-457; ONT.py:307-382; CLR.py:328-402; reads.py:147-237)
//   cluster_kernel     cluster_del / cluster_ins seeded greedy        (Hifi.py:196-288)
//   pair_kernel        pair_sig                                        (Hifi.py:548-592)
// All paths are relative to bin/VolcanoSV-vc/Large_INDEL/extract_contig_signature_<X>.py /
// extract_reads_signature.py in the reference.
//
// No stage compacts: rows that a stage drops are marked VSV_M_DEAD, get the all-ones sort key and
// fall to the end of the next stable radix sort; `count_alive` then publishes the live row count on
// the device. Nothing here synchronises with the host.
#include "vsv_device.h"
#include "vsv_env.h"

namespace {

__device__ __forceinline__ vsv_sig dead_sig() {
  vsv_sig s;
  s.pos = 0; s.svlen = 0; s.q_start = 0; s.q_end = 0; s.rec = 0; s.rec2 = 0; s.meta = VSV_M_DEAD; s.tid = 0;
  return s;
}

__device__ __forceinline__ int64_t wave_sum64(int64_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// ---- intra-read fold ------------------------------------------------------------------------------
// T_RAW order is (record, op, hap). The reference folds a record's signatures left to right per hap pass and type
// (cluster_ins_one_read / cluster_del_one_read, Hifi.py:91-161): the running "last kept" signature s1 absorbs s2 when both are
// long and close. Positions never decrease along a record, and s1 lies at or before the previous signature of its list, so a
// signature at least T (150 DEL / 380 INS: the widest test) behind its list predecessor can merge with nothing: the lists are cut
// into independent CHAINS at such gaps and one thread folds one chain, whatever the record's length (a Mb contig carries
// thousands of signatures; one thread per RECORD made this the slowest kernel of the contig-like shape). Out of place: every
// row is written once, by the thread that owns its chain.
__device__ __forceinline__ int fold_slot(uint32_t meta) { return ((meta & VSV_M_HP2) ? 2 : 0) + ((meta & VSV_M_DEL) ? 1 : 0); }
__global__ __launch_bounds__(256) void fold_kernel(const vsv_sig* __restrict__ in, vsv_sig* __restrict__ out_rows, const Counters* ctr, SlimOut so) {
  const uint32_t n = ctr->n_raw;
  // every row is written once (+ its element, when the run works on elements)
#define FOLD_PUT(IDX, ROW) do { const vsv_sig r_ = (ROW); const uint32_t x_ = (IDX); out_rows[x_] = r_; vsv_slim_emit(so, x_, r_); } while (0)
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const vsv_sig me = in[i];
    if (me.meta & VSV_M_DEAD) { FOLD_PUT(i, me); continue; }
    const int slot = fold_slot(me.meta);
    const int64_t T = (me.meta & VSV_M_DEL) ? 150 : 380;
    bool head = true;
    for (uint32_t j = i; j-- > 0;) {                       // nearest earlier row of the same list within T
      const vsv_sig p = in[j];
      if (p.rec != me.rec || (int64_t)me.pos - p.pos >= T) break;
      if (!(p.meta & VSV_M_DEAD) && fold_slot(p.meta) == slot) { head = false; break; }
    }
    if (!head) continue;
    vsv_sig s1 = me;
    uint32_t last = i;
    int32_t prev_pos = me.pos;                             // position of the list's previous row
    for (uint32_t k = i + 1; k < n; ++k) {
      const vsv_sig s2 = in[k];
      if (s2.rec != me.rec || (int64_t)s2.pos - prev_pos >= T) break;   // whatever follows in this list starts a new chain
      if ((s2.meta & VSV_M_DEAD) || fold_slot(s2.meta) != slot) continue;
      prev_pos = s2.pos;
      int64_t d = (int64_t)s2.pos - s1.pos; if (d < 0) d = -d;
      bool merged = false;
      if (s2.meta & VSV_M_DEL) {
        if (s1.svlen > 150 && s2.svlen > 150 && d < 150) {              // Hifi.py:148-150
          s1.svlen = s2.pos + s2.svlen - s1.pos;                         // Hifi.py:104
          s1.q_end = s1.q_start + 1;
          merged = true;
        }
      } else {
        if ((s1.svlen > 100 && s2.svlen > 100 && d < 250) ||             // Hifi.py:115-117 (subset), 126-128
            (s1.svlen > 320 && s2.svlen > 320 && d < 380)) {             // Hifi.py:120-122
          s1.q_end = s2.q_end;                                            // Hifi.py:94
          s1.svlen = s1.q_end - s1.q_start;                               // Hifi.py:96
          merged = true;
        }
      }
      if (merged) { vsv_sig dd = s2; dd.meta |= VSV_M_DEAD; FOLD_PUT(k, dd); }
      else { FOLD_PUT(last, s1); last = k; s1 = s2; }
    }
    FOLD_PUT(last, s1);
#undef FOLD_PUT
  }
}

// The same fold IN PLACE, for runs whose scan placed rows and elements straight into the stage-1 input table (cigar_scan.hip,
// k1l_place): on a contig pile one row in ~700 has a neighbour it could fold with, and the out-of-place pass above moves 500 MB
// (rows in, rows + elements out) to find that out. Here every thread reads its row's 16-byte ELEMENT and those around it: a row
// without another element of its list (haplotype, type) within T on either side — scanned over whatever lies between, as the rule
// does, and across record boundaries, which the elements do not show: a superset — is a chain of one and stays as it is. The few
// others run the rule above on the rows. In place is safe because every decision reads fields no fold changes — record, position,
// list — (the DEAD bit is ignored: the scan never emits a dead row, and a neighbour's owner may be setting it right now), a chain is
// written by its head's thread only, and an element somebody already killed just counts as "something is near".
__global__ __launch_bounds__(256) void fold_elems(vsv_sig* __restrict__ rows, uint4* __restrict__ elems, const Counters* ctr, SlimOut so) {
  const uint32_t n = ctr->n_raw;
  const int pb = so.pb;
  const uint64_t pmask = pb >= 64 ? ~0ull : ((1ull << pb) - 1ull);
  // the position range of the table (sort 1 of the element path cuts it into buckets): every element passes through here anyway
  __shared__ uint32_t s_hi, s_lo;
  if (threadIdx.x == 0) { s_hi = 0; s_lo = 0; }
  __syncthreads();
  uint32_t p_hi = 0, p_lo = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint4 e = elems[i];
    const uint64_t key = (uint64_t)e.x | ((uint64_t)e.y << 32);
    const int64_t pos = (int64_t)(key & pmask);
    p_hi = max(p_hi, (uint32_t)pos); p_lo = max(p_lo, ~(uint32_t)pos);
    const uint32_t slot = (uint32_t)(key >> (pb + 1)) & 3u;          // hap << 1 | del
    const uint64_t tidk = key >> (pb + 3);
    const int64_t T = (slot & 1u) ? 150 : 380;
    // (positions never decrease along a record: where they do between two neighbours, a record ends — the rule stops there, and so
    // does the scan. Without that test the first and last element of every contig walked through its neighbours' whole records,
    // whose positions lie megabases on the far side: 742 us on the pile)
    bool near = false;
    int64_t edge = pos;
    for (uint32_t j = i; j-- > 0;) {
      const uint4 x = elems[j];
      const uint64_t kj = (uint64_t)x.x | ((uint64_t)x.y << 32);
      if (kj == VSV_KEY_DEAD) { near = true; break; }
      const int64_t pj = (int64_t)(kj & pmask);
      if ((kj >> (pb + 3)) != tidk || pj > edge || pos - pj >= T) break;
      if (((uint32_t)(kj >> (pb + 1)) & 3u) == slot) { near = true; break; }
      edge = pj;
    }
    edge = pos;
    for (uint32_t k = i + 1; !near && k < n; ++k) {
      const uint4 x = elems[k];
      const uint64_t kk = (uint64_t)x.x | ((uint64_t)x.y << 32);
      if (kk == VSV_KEY_DEAD) { near = true; break; }
      const int64_t pk = (int64_t)(kk & pmask);
      if ((kk >> (pb + 3)) != tidk || pk < edge || pk - pos >= T) break;
      if (((uint32_t)(kk >> (pb + 1)) & 3u) == slot) { near = true; break; }
      edge = pk;
    }
    if (!near) continue;
    // ---- the rule itself (fold_kernel), on the rows ----
    const vsv_sig me = rows[i];
    const int fslot = fold_slot(me.meta);
    bool head = true;
    for (uint32_t j = i; j-- > 0;) {
      const vsv_sig p = rows[j];
      if (p.rec != me.rec || (int64_t)me.pos - p.pos >= T) break;
      if (fold_slot(p.meta) == fslot) { head = false; break; }
    }
    if (!head) continue;
    vsv_sig s1 = me;
    uint32_t last = i;
    int32_t prev_pos = me.pos;
    bool dirty = false;
    auto put = [&]() {                                   // s1 changed: its row, and the length in its element
      rows[last] = s1;
      reinterpret_cast<uint32_t*>(elems + last)[2] = (uint32_t)s1.svlen;
      if ((uint32_t)s1.svlen >= (1u << 30)) atomicOr(so.err, ERRB_SLIM_FALLBACK);
    };
    for (uint32_t k = i + 1; k < n; ++k) {
      const vsv_sig s2 = rows[k];
      if (s2.rec != me.rec || (int64_t)s2.pos - prev_pos >= T) break;
      if (fold_slot(s2.meta) != fslot) continue;
      prev_pos = s2.pos;
      int64_t d = (int64_t)s2.pos - s1.pos; if (d < 0) d = -d;
      // (the merged length and query end as values of their own, assigned to s1 in one place: with the assignments inside the two
      // branches hipcc 7.2 dropped the copy of s2.q_end into s1.q_end on the INS side — q_end 0 in every folded insertion)
      const bool is_del = (s2.meta & VSV_M_DEL) != 0;
      const bool merged = is_del ? (s1.svlen > 150 && s2.svlen > 150 && d < 150)                                   // Hifi.py:148-150
                                 : ((s1.svlen > 100 && s2.svlen > 100 && d < 250) ||                               // Hifi.py:115-117 (subset), 126-128
                                    (s1.svlen > 320 && s2.svlen > 320 && d < 380));                                // Hifi.py:120-122
      const int32_t new_qe = is_del ? s1.q_start + 1 : s2.q_end;                                                    // Hifi.py:94
      const int32_t new_len = is_del ? s2.pos + s2.svlen - s1.pos : s2.q_end - s1.q_start;                          // Hifi.py:104 / 96
      if (merged) { s1.q_end = new_qe; s1.svlen = new_len; }
      if (merged) {
        rows[k].meta = s2.meta | VSV_M_DEAD;
        elems[k] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
        dirty = true;
      } else {
        if (dirty) put();
        last = k; s1 = s2; dirty = false;
      }
    }
    if (dirty) put();
  }
  if (so.mm && pb < 32) {            // (block-uniform; every thread gets here)
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { p_hi = max(p_hi, (uint32_t)__shfl_xor((int)p_hi, d, 64)); p_lo = max(p_lo, (uint32_t)__shfl_xor((int)p_lo, d, 64)); }
    if ((threadIdx.x & 63) == 0) { atomicMax(&s_hi, p_hi); atomicMax(&s_lo, p_lo); }
    __syncthreads();
    // (one slot per residue of the block index: thousands of atomics on ONE address take their turns at the L2, ~10 ns each, and in a
    // position-sorted table every block brings a new maximum; MsdDigit::prep reduces the slots)
    if (threadIdx.x == 0) { if (s_hi) atomicMax(&so.mm[blockIdx.x & 63], s_hi); if (s_lo) atomicMax(&so.mm[64 + (blockIdx.x & 63)], s_lo); }
  }
}

// ---- sig_extract.py generate_combine_sigs (SE:373-435): signals of one read merged by distance ---------------------------
// One thread per record's group of raw rows, in op order, INS and DEL independently. INS: a signal at most
// merge_ins_threshold after the LAST merged signal's position joins the current one (lengths add; the host concatenates the
// sequences of the q_end pieces starting at raw row rec2). DEL: the comparison value is pos+len of the last merged signal —
// except that a group opened after a flush starts from its own pos (SE:427-428 appends i[0], not sum(i)); kept as is.
__global__ __launch_bounds__(256) void combine_kernel(vsv_sig* __restrict__ s, const Counters* ctr, int merge_ins, int merge_del) {
  const uint32_t n = ctr->n_raw;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t rec = s[i].rec;
    if (i > 0 && s[i - 1].rec == rec) continue;  // not a group head
    int64_t ins_first = -1, del_first = -1, ins_last = 0, del_cmp = 0;
    for (uint32_t k = i; k < n && s[k].rec == rec; ++k) {
      const vsv_sig s2 = s[k];
      if (s2.meta & VSV_M_DEL) {
        if (del_first >= 0 && (int64_t)s2.pos - del_cmp <= (int64_t)merge_del) {     // SE:417-419
          s[del_first].svlen += s2.svlen;
          s[del_first].q_end += 1;
          del_cmp = (int64_t)s2.pos + s2.svlen;
          s[k].meta = s2.meta | VSV_M_DEAD;
        } else {
          del_cmp = del_first < 0 ? (int64_t)s2.pos + s2.svlen : (int64_t)s2.pos;    // SE:414 vs SE:427-428
          del_first = k;
          s[k].q_end = 1; s[k].rec2 = k;
        }
      } else {
        if (ins_first >= 0 && (int64_t)s2.pos - ins_last <= (int64_t)merge_ins) {     // SE:395-398
          s[ins_first].svlen += s2.svlen;
          s[ins_first].q_end += 1;
          ins_last = s2.pos;
          s[k].meta = s2.meta | VSV_M_DEAD;
        } else {
          ins_first = k; ins_last = s2.pos;                                           // SE:393, 406-407
          s[k].q_end = 1; s[k].rec2 = k;
        }
      }
    }
  }
}

// ---- CLR gate: every haplotype-tagged record, BEFORE the scan -----------------------------------------
// C:422-427 computes ins_pct / var_dist for every record whose name carries the haplotype tag (whatever its mapq: a record
// without M ops raises ZeroDivisionError there) and only walks the CIGAR - and asserts reference_end - when the gate passes.
// The scan therefore reads a gated copy of the flag bytes: haplotype bits cleared where the gate fails. The split stage
// (C:453-457) is not gated and keeps the caller's flags.
// G lanes per record: 64 for contig-like records (thousands of ops), 8 for read-like ones (tens of ops, eight records per wave).
template <int G>
__global__ __launch_bounds__(256) void clr_gate_records(RecView rv, uint8_t* __restrict__ gflag, Counters* ctr) {
  const int lane = threadIdx.x & 63, sub = lane & (G - 1);
  constexpr int PER_WAVE = 64 / G;
  const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x >> 6) * PER_WAVE;
  const int64_t g0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * PER_WAVE + lane / G;
  const int64_t rounds = (rv.n_records + ngroups - 1) / ngroups;          // every lane of a wave takes the same number of rounds
  for (int64_t r = 0; r < rounds; ++r) {
    const int64_t rec = g0 + r * ngroups;
    const bool live = rec < rv.n_records;
    const uint8_t fl = live ? rv.flag[rec] : 0;
    uint8_t out = fl;
    const bool tagged = live && (fl & (VSV_F_HP1 | VSV_F_HP2));
    uint64_t a = 0, b = 0;
    if (tagged) vsv_op_range(rv, rec, a, b);
    int64_t ins = 0, m = 0, nm = 0;
    for (uint64_t k = a + sub; k < b; k += G) {
      const uint32_t w = rv.cigar[k], op = w & 15u;
      if (op == 0) { m += w >> 4; nm++; } else if (op == 1) ins += w >> 4;
    }
#pragma unroll
    for (int d = G >> 1; d > 0; d >>= 1) { ins += __shfl_xor(ins, d, 64); m += __shfl_xor(m, d, 64); nm += __shfl_xor(nm, d, 64); }
    if (tagged && b > a) {                                                // an empty CIGAR is the scan's error to report
      bool pass;
      if (m + ins == 0 || nm == 0) { if (sub == 0) atomicOr(&ctr->err, ERRB_ZERODIV); pass = false; }
      else pass = (100 * ins <= 13 * (m + ins)) || (m >= 200 * nm);       // CLR.py:61, 70, 427 in exact integers
      if (!pass) out = fl & (uint8_t)~(VSV_F_HP1 | VSV_F_HP2);
    }
    if (live && sub == 0) gflag[rec] = out;
  }
}

// ---- split stage ----------------------------------------------------------------------------------
struct SplitCfg {
  int contig;        // 1: hp/mapq eligibility, two hap passes; 0: reads path, every record, one pass
  int min_mapq;
  int qid_bits;
  int tid_shift;     // ckey = (tid - tid_lo) << tid_shift | hap << qid_bits | qid
  int tid_lo, tid_bits;
};

// ---- names that occur more than once, without a counting table ---------------------------------------
// A record whose qid is <= the running maximum of all earlier qids MAY be a repeated name; every true repeat is such a
// record (its name appeared before, so the maximum is >= its qid). With qids dense in first-appearance order (the
// ingest's numbering) the test is exact and only the ~1 % repeat records touch memory: one atomicOr into a
// 1-bit-per-name table that stays L2-resident. Any other numbering only adds false candidates, which drop out later
// because a candidate group needs two eligible members with equal (tid, hap, qid).
constexpr int QM_TILE = 2048;   // 256 threads x 8 consecutive records
// 8 consecutive qids (+1; 0 = no record) of a thread: two 16-byte loads when the array is 16-byte aligned
__device__ __forceinline__ void load_qid8(const uint32_t* __restrict__ qid, int64_t base, int64_t n, bool vec, uint32_t (&q)[8]) {
  if (vec && base + 8 <= n) {
    const uint4 a = *reinterpret_cast<const uint4*>(qid + base), b = *reinterpret_cast<const uint4*>(qid + base + 4);
    q[0] = a.x + 1u; q[1] = a.y + 1u; q[2] = a.z + 1u; q[3] = a.w + 1u; q[4] = b.x + 1u; q[5] = b.y + 1u; q[6] = b.z + 1u; q[7] = b.w + 1u;
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = (base + k < n) ? qid[base + k] + 1u : 0u;
  }
}
// Shards of up to 8192 tiles: two launches, every marking block reduces the raw maxima of the tiles in front of it itself.
// (+ every block clears its slice of the name table's bits for qid_mark_dups: no fill launch)
__global__ __launch_bounds__(256) void qid_tile_max(const uint32_t* __restrict__ qid, int64_t n, uint32_t* __restrict__ tile_max, bool vec,
                                                    uint32_t* __restrict__ tab, uint32_t tab_words) {
  __shared__ uint32_t sh[4];
  {
    const uint32_t per = (tab_words + gridDim.x - 1) / gridDim.x;
    const uint32_t a = min(tab_words, blockIdx.x * per), e = min(tab_words, a + per);
    for (uint32_t i = a + threadIdx.x; i < e; i += 256) tab[i] = 0;
  }
  uint32_t q[8];
  load_qid8(qid, (int64_t)blockIdx.x * QM_TILE + threadIdx.x * 8, n, vec, q);
  uint32_t m = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) m = max(m, q[k]);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) tile_max[blockIdx.x] = max(max(sh[0], sh[1]), max(sh[2], sh[3]));
}
// raw_tiles: tile_excl holds the tiles' own maxima
__global__ __launch_bounds__(256) void qid_mark_dups(const uint32_t* __restrict__ qid, int64_t n, const uint32_t* __restrict__ tile_excl,
                                                     uint32_t* __restrict__ dupbits, bool vec, uint32_t n_qids, uint32_t* __restrict__ err,
                                                     bool raw_tiles) {
  __shared__ uint32_t sh[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t q[8], m = 0;
  load_qid8(qid, (int64_t)blockIdx.x * QM_TILE + threadIdx.x * 8, n, vec, q);
#pragma unroll
  for (int k = 0; k < 8; ++k) m = max(m, q[k]);
  uint32_t incl = m;                       // inclusive prefix maximum over the wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl = max(incl, o); }
  if (lane == 63) sh[wv] = incl;
  uint32_t excl = (uint32_t)__shfl_up((int)incl, 1, 64);
  if (lane == 0) excl = 0;
  __syncthreads();
  // the largest id of the tiles in front of this one: every block reduces the raw tile maxima itself (a few thousand L2-hot words)
  // instead of waiting for a single-block prefix-max launch
  __shared__ uint32_t sh_before[4];
  uint32_t before = 0;
  if (raw_tiles) {
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) before = max(before, tile_excl[j]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) before = max(before, (uint32_t)__shfl_xor((int)before, d, 64));
  } else before = tile_excl[blockIdx.x];
  if (lane == 0) sh_before[wv] = before;
  __syncthreads();
  uint32_t run = max(max(max(sh_before[0], sh_before[1]), max(sh_before[2], sh_before[3])), excl);
  for (int w = 0; w < wv; ++w) run = max(run, sh[w]);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (q[k] != 0 && q[k] - 1u >= n_qids) atomicOr(err, ERRB_RANGE);                 // qid outside [0, n_qids): the table has no bit for it
    else if (q[k] != 0 && q[k] <= run) atomicOr(&dupbits[(q[k] - 1u) >> 5], 1u << ((q[k] - 1u) & 31u));
    run = max(run, q[k]);
  }
}

// One launch: the tile's largest id, its exclusive prefix maximum over the tiles in front by look-back (vsv_lb1_exclusive), the marks.
// For shards of MORE than 8192 tiles (config 3: 24 k), where the two-launch form above needed a single-block scan launch in between
// (45 us) — on smaller shards the look-back loses: the blocks wait for the blocks in front while another engine's scan holds the
// chip's slots (config 2, four engines, same box: 0.433 against 0.417 ms per step). The table this run marks in was cleared by the run in front (or when it was
// allocated); every block clears its slice of the OTHER table, which the next run marks in: a launch cannot clear the table it marks.
constexpr int QD_SUB = 4;                    // sub-tiles of QM_TILE records per block: one look-back per 8192 records (a block of 2048 spent
                                             // more time waiting for the blocks in front than working: 33 us against 24 for the two launches)
__global__ __launch_bounds__(256) void qid_dups(const uint32_t* __restrict__ qid, int64_t n, uint32_t* __restrict__ dupbits, bool vec, uint32_t n_qids,
                                                uint32_t* __restrict__ err, uint64_t* __restrict__ lbw, uint32_t epoch, uint32_t* __restrict__ tab_next, uint32_t clear_words) {
  __shared__ uint32_t sh[QD_SUB][4];
  __shared__ uint32_t sh_before;
  {
    const uint32_t per = (clear_words + gridDim.x - 1) / gridDim.x;
    const uint32_t a = min(clear_words, blockIdx.x * per), e = min(clear_words, a + per);
    for (uint32_t i = a + threadIdx.x; i < e; i += 256) tab_next[i] = 0;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t q[QD_SUB][8], excl[QD_SUB];
#pragma unroll
  for (int s = 0; s < QD_SUB; ++s) {
    load_qid8(qid, ((int64_t)blockIdx.x * QD_SUB + s) * QM_TILE + threadIdx.x * 8, n, vec, q[s]);
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) m = max(m, q[s][k]);
    uint32_t incl = m;                     // inclusive prefix maximum over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl = max(incl, o); }
    if (lane == 63) sh[s][wv] = incl;
    excl[s] = (uint32_t)__shfl_up((int)incl, 1, 64);
    if (lane == 0) excl[s] = 0;
  }
  __syncthreads();
  if (wv == 0) {                           // the largest id of the blocks in front of this one
    uint32_t tile = 0;
#pragma unroll
    for (int s = 0; s < QD_SUB; ++s) tile = max(tile, max(max(sh[s][0], sh[s][1]), max(sh[s][2], sh[s][3])));
    const uint32_t before = vsv_lb1_exclusive<true>(lbw, epoch, blockIdx.x, tile, lane, err);
    if (lane == 0) sh_before = before;
  }
  __syncthreads();
  uint32_t front = sh_before;              // everything in front of the sub-tile
#pragma unroll
  for (int s = 0; s < QD_SUB; ++s) {
    uint32_t run = max(front, excl[s]);
    for (int w = 0; w < wv; ++w) run = max(run, sh[s][w]);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (q[s][k] != 0 && q[s][k] - 1u >= n_qids) atomicOr(err, ERRB_RANGE);                 // qid outside [0, n_qids): the table has no bit for it
      else if (q[s][k] != 0 && q[s][k] <= run) atomicOr(&dupbits[(q[s][k] - 1u) >> 5], 1u << ((q[s][k] - 1u) & 31u));
      run = max(run, q[s][k]);
    }
    front = max(front, max(max(sh[s][0], sh[s][1]), max(sh[s][2], sh[s][3])));
  }
}

// ordered compaction of candidate (record, hap) items. A thread owns 4 consecutive records per round (flag / mapq as one
// dword each, qid as one 16-byte load when the arrays are aligned), i.e. 8 items in (record, hap) order as an 8-bit mask.
constexpr int SC_ROUNDS = 8;     // (one look-back per 8192 records: see qid_dups)
constexpr int SC_TILE_REC = 256 * 4 * SC_ROUNDS;    // records per block
// One launch: the block's item masks and count, its offset by look-back over the blocks in front (vsv_lb1_exclusive), the writes.
// (Round 3: a count launch that left the masks in memory and a fill launch that read them back, with a scan between them beyond
// 8192 blocks.)
__global__ __launch_bounds__(256) void split_cand(RecView rv, SplitCfg c, const uint32_t* __restrict__ tab, uint64_t* __restrict__ ckey,
                                                  uint32_t* __restrict__ crec, uint32_t cap, bool vec, Counters* __restrict__ ctr,
                                                  uint64_t* __restrict__ lbw, uint32_t epoch) {
  __shared__ uint32_t cnt[SC_ROUNDS][4];
  __shared__ uint32_t sh_off;
  const uint64_t n = (uint64_t)rv.n_records;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t m[SC_ROUNDS], below[SC_ROUNDS], q[SC_ROUNDS][4];
#pragma unroll
  for (int k = 0; k < SC_ROUNDS; ++k) {
    const uint64_t r0 = (uint64_t)blockIdx.x * SC_TILE_REC + (uint64_t)k * 1024 + (uint64_t)threadIdx.x * 4;
    m[k] = 0;
    uint32_t fl4 = 0, mq4 = 0;
    if (vec && r0 + 4 <= n) {
      const uint4 qq = *reinterpret_cast<const uint4*>(rv.qid + r0);
      q[k][0] = qq.x; q[k][1] = qq.y; q[k][2] = qq.z; q[k][3] = qq.w;
      if (c.contig) { fl4 = *reinterpret_cast<const uint32_t*>(rv.flag + r0); mq4 = *reinterpret_cast<const uint32_t*>(rv.mapq + r0); }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        q[k][j] = 0;
        if (r0 + j < n) {
          q[k][j] = rv.qid[r0 + j];
          if (c.contig) { fl4 |= (uint32_t)rv.flag[r0 + j] << (8 * j); mq4 |= (uint32_t)rv.mapq[r0 + j] << (8 * j); }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (r0 + j >= n) continue;
      uint32_t e;                                                     // bit 0: hap 0 item, bit 1: hap 1 item
      if (!c.contig) e = 1u;
      else {
        const uint32_t fl = (fl4 >> (8 * j)) & 0xFFu, mq = (mq4 >> (8 * j)) & 0xFFu;
        e = (mq >= (uint32_t)c.min_mapq) ? (((fl & VSV_F_HP1) ? 1u : 0u) | ((fl & VSV_F_HP2) ? 2u : 0u)) : 0u;   // Hifi.py:425-427
      }
      if (e && q[k][j] < (uint32_t)rv.n_qids && ((tab[q[k][j] >> 5] >> (q[k][j] & 31u)) & 1u)) m[k] |= e << (2 * j);
    }
    const uint32_t cc = (uint32_t)__popc(m[k]);
    uint32_t incl = cc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
    below[k] = incl - cc;
    if (lane == 63) cnt[k][wv] = incl;
  }
  __syncthreads();
  if (wv == 0) {
    uint32_t t = 0;
    for (int k = 0; k < SC_ROUNDS; ++k) for (int w = 0; w < 4; ++w) t += cnt[k][w];
    const uint32_t before = vsv_lb1_exclusive<false>(lbw, epoch, blockIdx.x, t, lane, &ctr->err);
    if (lane == 0) {
      sh_off = before;
      if (blockIdx.x == gridDim.x - 1) {                    // the last block knows the total: publish the candidate count
        const uint64_t tot = (uint64_t)before + t;          // (n_s1 = n_raw + n_cand is split_eval's: this kernel may run next to the scan)
        ctr->n_cand = tot < cap ? (uint32_t)tot : cap;
        if (tot > cap) atomicOr(&ctr->err, ERRB_CAPACITY);
      }
    }
  }
  __syncthreads();
  uint32_t off = sh_off;
#pragma unroll
  for (int k = 0; k < SC_ROUNDS; ++k) {
    uint32_t dst = off + below[k];
    for (int w = 0; w < wv; ++w) dst += cnt[k][w];
    const uint64_t r0 = (uint64_t)blockIdx.x * SC_TILE_REC + (uint64_t)k * 1024 + (uint64_t)threadIdx.x * 4;
    uint32_t mm = m[k];
    while (mm) {
      const int it = __builtin_ctz(mm);
      mm &= mm - 1;
      const uint32_t j = (uint32_t)it >> 1, hap = (uint32_t)it & 1u;
      if (dst < cap) {
        const uint32_t qj = rv.qid[r0 + j];
        const uint32_t trel = (uint32_t)(rv.tid[r0 + j] - c.tid_lo);
        if (trel >> c.tid_bits) atomicOr(&ctr->err, ERRB_RANGE);     // tid outside [tid_lo, n_tids)
        ckey[dst] = ((uint64_t)trel << c.tid_shift) | ((uint64_t)hap << c.qid_bits) | qj;
        crec[dst] = (uint32_t)(r0 + j);
      }
      ++dst;
    }
    off += cnt[k][0] + cnt[k][1] + cnt[k][2] + cnt[k][3];
  }
}

constexpr int SC2_ROUNDS = 2;
constexpr int SC2_TILE_REC = 256 * 4 * SC2_ROUNDS;    // records per block
// The two-launch form (count, then fill from the masks the count pass left): shards of up to 8192 blocks of 2048 records, where every
// fill block adds up the raw counts of the blocks in front of it itself. WRITE with raw_counts: blk still holds the blocks' own counts
template <bool WRITE>
__global__ __launch_bounds__(256) void split_cand_2pass(RecView rv, SplitCfg c, const uint32_t* __restrict__ tab,
                                                  uint32_t* __restrict__ blk, uint64_t* __restrict__ ckey,
                                                  uint32_t* __restrict__ crec, uint32_t cap, bool vec, Counters* __restrict__ ctr,
                                                  uint8_t* __restrict__ masks, bool raw_counts = false) {
  // masks[block][round][thread]: the 8-bit item mask of a thread's 4 records. The count pass writes it, the write pass reads it
  // back instead of streaming qid / flag / mapq and probing the name table a second time.
  __shared__ uint32_t cnt[SC2_ROUNDS][4];
  const uint64_t n = (uint64_t)rv.n_records;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t m[SC2_ROUNDS], below[SC2_ROUNDS], q[SC2_ROUNDS][4];
#pragma unroll
  for (int k = 0; k < SC2_ROUNDS; ++k) {
    const uint64_t r0 = (uint64_t)blockIdx.x * SC2_TILE_REC + (uint64_t)k * 1024 + (uint64_t)threadIdx.x * 4;
    m[k] = 0;
    uint32_t fl4 = 0, mq4 = 0;
    const size_t mi = ((size_t)blockIdx.x * SC2_ROUNDS + k) * 256 + threadIdx.x;
    if (WRITE) {
      m[k] = masks[mi];
    } else if (vec && r0 + 4 <= n) {
      const uint4 qq = *reinterpret_cast<const uint4*>(rv.qid + r0);
      q[k][0] = qq.x; q[k][1] = qq.y; q[k][2] = qq.z; q[k][3] = qq.w;
      if (c.contig) { fl4 = *reinterpret_cast<const uint32_t*>(rv.flag + r0); mq4 = *reinterpret_cast<const uint32_t*>(rv.mapq + r0); }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        q[k][j] = 0;
        if (r0 + j < n) {
          q[k][j] = rv.qid[r0 + j];
          if (c.contig) { fl4 |= (uint32_t)rv.flag[r0 + j] << (8 * j); mq4 |= (uint32_t)rv.mapq[r0 + j] << (8 * j); }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (WRITE || r0 + j >= n) continue;
      uint32_t e;                                                     // bit 0: hap 0 item, bit 1: hap 1 item
      if (!c.contig) e = 1u;
      else {
        const uint32_t fl = (fl4 >> (8 * j)) & 0xFFu, mq = (mq4 >> (8 * j)) & 0xFFu;
        e = (mq >= (uint32_t)c.min_mapq) ? (((fl & VSV_F_HP1) ? 1u : 0u) | ((fl & VSV_F_HP2) ? 2u : 0u)) : 0u;   // Hifi.py:425-427
      }
      if (e && q[k][j] < (uint32_t)rv.n_qids && ((tab[q[k][j] >> 5] >> (q[k][j] & 31u)) & 1u)) m[k] |= e << (2 * j);
    }
    if (!WRITE) masks[mi] = (uint8_t)m[k];
    const uint32_t cc = (uint32_t)__popc(m[k]);
    uint32_t incl = cc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
    below[k] = incl - cc;
    if (lane == 63) cnt[k][wv] = incl;
  }
  __syncthreads();
  if (!WRITE) {
    if (threadIdx.x == 0) {
      uint32_t t = 0;
      for (int k = 0; k < SC2_ROUNDS; ++k) for (int w = 0; w < 4; ++w) t += cnt[k][w];
      blk[blockIdx.x] = t;
    }
    return;
  }
  // exclusive block offset = the counts of the blocks in front of this one, summed here (no scan launch in between)
  __shared__ uint32_t sh_off[4];
  uint32_t off = 0;
  if (raw_counts) {
    for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) off += blk[j];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) off += (uint32_t)__shfl_xor((int)off, d, 64);
  } else if (threadIdx.x == 0) off = blk[blockIdx.x];
  if ((threadIdx.x & 63) == 0) sh_off[threadIdx.x >> 6] = off;
  __syncthreads();
  off = sh_off[0] + sh_off[1] + sh_off[2] + sh_off[3];
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {   // the last block knows the total: publish the candidate count
    uint32_t t = off;                                       // (n_s1 = n_raw + n_cand is split_eval's: this kernel may run next to the scan)
    for (int k = 0; k < SC2_ROUNDS; ++k) for (int w = 0; w < 4; ++w) t += cnt[k][w];
    ctr->n_cand = t < cap ? t : cap;
    if (t > cap) atomicOr(&ctr->err, ERRB_CAPACITY);
  }
#pragma unroll
  for (int k = 0; k < SC2_ROUNDS; ++k) {
    uint32_t dst = off + below[k];
    for (int w = 0; w < wv; ++w) dst += cnt[k][w];
    const uint64_t r0 = (uint64_t)blockIdx.x * SC2_TILE_REC + (uint64_t)k * 1024 + (uint64_t)threadIdx.x * 4;
    uint32_t mm = m[k];
    while (mm) {
      const int it = __builtin_ctz(mm);
      mm &= mm - 1;
      const uint32_t j = (uint32_t)it >> 1, hap = (uint32_t)it & 1u;
      if (dst < cap) {
        const uint32_t qj = rv.qid[r0 + j];
        const uint32_t trel = (uint32_t)(rv.tid[r0 + j] - c.tid_lo);
        if (trel >> c.tid_bits) atomicOr(&ctr->err, ERRB_RANGE);     // tid outside [tid_lo, n_tids)
        ckey[dst] = ((uint64_t)trel << c.tid_shift) | ((uint64_t)hap << c.qid_bits) | qj;
        crec[dst] = (uint32_t)(r0 + j);
      }
      ++dst;
    }
    off += cnt[k][0] + cnt[k][1] + cnt[k][2] + cnt[k][3];
  }
}

// slot j of the sorted candidate list starts a pair iff slot j+1 has the same (tid,hap,qid) key.
// okey orders pairs like the reference: per (tid, hap pass) by first appearance of the name
// (Counter insertion order, Hifi.py:428-432,446), then by position in the name's record list.
__global__ __launch_bounds__(256) void split_mark_pairs(const uint64_t* __restrict__ ckey, const uint32_t* __restrict__ crec,
                                                        SplitCfg c, int rec_bits, uint64_t* __restrict__ okey,
                                                        uint32_t* __restrict__ oval, const Counters* ctr) {
  const uint32_t n = ctr->n_cand;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const uint64_t k = ckey[j];
    uint64_t ok = VSV_KEY_DEAD;
    if (j + 1 < n && ckey[j + 1] == k) {
      uint32_t g = j;
      while (g > 0 && ckey[g - 1] == k) --g;
      const uint64_t tidhap = k >> c.qid_bits;  // (tid, hap)
      ok = (tidhap << rec_bits) | crec[g];
    }
    okey[j] = ok;
    oval[j] = j;
  }
}

constexpr uint32_t SE_LONG_OPS = 1024;   // a pair whose two CIGARs hold more ops than this is summed by the whole wave
template <int SE_GROUP>
__device__ __forceinline__ int64_t group_sum64(int64_t v) {
#pragma unroll
  for (int d = SE_GROUP / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
// reference span (pysam reference_end - pos) and read length (get_readlen, Hifi.py:290-305 / reads.py) of the ops [a, b): lane `l` of
// `width` lanes takes every width-th group of four ops (one 16-byte load); the caller reduces. The kernel is bound by the instructions
// it issues, not by the CIGAR bytes (PMC, config 3: 317 M vector + 351 M scalar wave instructions per launch in the compare-and-select
// form — every `op == x || ...` is a compare plus a scalar OR): an op costs a bit-field extract of its bit in the op set (sign-extended
// to a mask), an AND and an add per sum, in 32 bits; eight ops (< 2^31: a length has 28 bits) go to the 64-bit sums at a time.
constexpr uint32_t SE_REF_OPS = 0x18D;                  // M D N = X consume the reference (pysam reference_end)
constexpr uint32_t SE_LEN_OPS = 0x033;                  // M I S H count for get_readlen (Hifi.py:290-305) ...
constexpr uint32_t SE_LEN_OPS_READS = 0x1B3;            // ... and = X too in reads.py
__device__ __forceinline__ uint32_t se_mask(uint32_t table, uint32_t op) { return (uint32_t)__builtin_amdgcn_sbfe(table, op, 1); }
template <bool WANT_REF>
__device__ __forceinline__ void op_sums(const uint32_t* __restrict__ cigar, uint64_t a, uint64_t b, uint32_t len_ops, uint32_t l, uint32_t width,
                                        int64_t& rf, int64_t& rl) {
  if (b <= a) return;
  uint32_t sr = 0, sq = 0;
  auto add = [&](uint32_t w) {
    const uint32_t op = w & 15u, len = w >> 4;
    if (WANT_REF) sr += len & se_mask(SE_REF_OPS, op);
    sq += len & se_mask(len_ops, op);
  };
  // 16-byte groups of the ARRAY (whatever the pointer's own alignment): [a4, b4); up to three ops in front and behind, one lane each
  const uint64_t mis = ((uintptr_t)cigar >> 2) & 3u;
  const uint64_t a4 = ((a + mis + 3ull) & ~3ull) - mis, b4 = ((b + mis) & ~3ull) - mis;
  const uint64_t he = a4 < b ? a4 : b;                  // head = [a, he)
  if (a + l < he) add(cigar[a + l]);
  if (b4 >= a4 && b4 + l < b) add(cigar[b4 + l]);       // tail = [b4, b) (b4 < a4: the head was everything)
  const uint64_t nq = b4 > a4 ? (b4 - a4) >> 2 : 0;
  const uint4* __restrict__ qp = reinterpret_cast<const uint4*>(cigar + a4);
  uint64_t q = l;
  for (; q + width < nq; q += 2ull * width) {
    const uint4 v0 = qp[q], v1 = qp[q + width];
    add(v0.x); add(v0.y); add(v0.z); add(v0.w);
    if (WANT_REF) { rf += sr; sr = 0; }
    rl += sq; sq = 0;
    add(v1.x); add(v1.y); add(v1.z); add(v1.w);
    if (WANT_REF) { rf += sr; sr = 0; }
    rl += sq; sq = 0;
  }
  if (q < nq) { const uint4 v0 = qp[q]; add(v0.x); add(v0.y); add(v0.z); add(v0.w); }
  if (WANT_REF) rf += sr;
  rl += sq;
}

// the pair rules of one candidate pair (`need` = it reached the length test): Hifi.py:331-371, ONT.py:348-373, CLR.py:369-377, reads.py:179-196
__device__ __forceinline__ vsv_sig split_rules(bool need, uint32_t i1, uint32_t i2, int32_t pos1, int32_t pos2, int32_t tid1, uint32_t hap, uint32_t last1,
                                               uint32_t first2, int64_t rf1, int64_t rl1, int64_t rl2, int dtype, int max_svlen, Counters* ctr) {
  vsv_sig out = dead_sig();
  if (need) {
    if (rl1 != rl2) atomicOr(&ctr->err, ERRB_READLEN);                                          // Hifi.py:331
    else {
      const int64_t Ref1e = (int64_t)pos1 + rf1, Ref2s = pos2;
      const int64_t Read1e = rl1 - (int64_t)(last1 >> 4), Read2s = first2 >> 4;
      const int64_t Diffdis = (Ref2s - Ref1e) - (Read2s - Read1e);
      const int64_t absd = Diffdis < 0 ? -Diffdis : Diffdis;
      if (absd <= max_svlen) {                                                                 // Hifi.py:354
        vsv_sig s = dead_sig();
        s.rec = i1; s.rec2 = i2; s.tid = tid1;
        uint32_t meta = VSV_M_SPLIT | (hap ? VSV_M_HP2 : 0u);
        bool emit = false;
        if (dtype == VSV_DTYPE_HIFI) {
          if (Diffdis >= 30) {
            const int64_t Diffolp = Read1e - Read2s, ao = Diffolp < 0 ? -Diffolp : Diffolp;
            if (ao <= 3000) {                                                                 // Hifi.py:357
              const int64_t h = Diffolp / 2;                                                  // int(Diffolp/2)
              s.pos = (int32_t)(Ref1e - h); s.svlen = (int32_t)Diffdis;
              s.q_start = (int32_t)(Read1e - h); s.q_end = s.q_start + 1; meta |= VSV_M_DEL; emit = true;
            }
          } else if (Diffdis <= -30) {
            const int64_t Diffolp = Ref1e - Ref2s, ao = Diffolp < 0 ? -Diffolp : Diffolp;
            if (Diffolp < 3000) {                                                             // Hifi.py:362
              int64_t sv = Read2s - Read1e + Diffolp; if (sv < 0) sv = -sv;
              s.pos = (int32_t)(ao > 400 ? (Ref1e + Ref2s) / 2 : Ref2s);
              s.svlen = (int32_t)sv; s.q_start = (int32_t)(Read1e - Diffolp); s.q_end = (int32_t)Read2s; emit = true;
            }
          }
        } else if (dtype == VSV_DTYPE_ONT || dtype == VSV_DTYPE_CLR) {
          // fp64 products exactly as CPython evaluates them (ONT.py:348-373, CLR.py:369-377); this
          // file is compiled with -ffp-contract=off.
          const double r_ = dtype == VSV_DTYPE_ONT ? 0.5 : 0.3;
          const double lo_f = dtype == VSV_DTYPE_ONT ? 0.8 : 0.3;
          if (Diffdis >= 30) {
            const int64_t Diffolp = Read1e - Read2s;
            const double dr = (double)Diffdis * r_;
            if (-dr <= (double)Diffolp && (double)Diffolp <= dr) {
              s.pos = (int32_t)Ref1e; s.svlen = (int32_t)Diffdis; s.q_start = (int32_t)Read1e; s.q_end = (int32_t)Read2s;
              meta |= VSV_M_DEL; emit = true;
            }
          } else {
            const int64_t Diffolp = Ref1e - Ref2s, ao = Diffolp < 0 ? -Diffolp : Diffolp;
            const double lo = (double)Diffdis * lo_f, hi = (double)absd * r_;
            if (lo <= (double)Diffolp && (double)Diffolp <= hi && Diffdis <= -30) {
              int64_t sv = Read2s - Read1e + Diffolp; if (sv < 0) sv = -sv;
              s.pos = (int32_t)(ao > 400 ? (Ref1e + Ref2s) / 2 : Ref2s);
              s.svlen = (int32_t)sv; s.q_start = (int32_t)(Read1e - Diffolp); s.q_end = (int32_t)Read2s; emit = true;
            }
          }
        } else {  // READS: reads.py:179-196
          const int64_t Diffolp = Ref1e - Ref2s;
          if (Diffolp < 30 && Diffdis >= 30) {
            s.pos = (int32_t)Ref1e; s.svlen = (int32_t)Diffdis; s.q_start = (int32_t)Read1e; s.q_end = (int32_t)Read2s;
            meta |= VSV_M_DEL; emit = true;
          } else if (Diffolp < 30 && Diffdis <= -30) {
            s.pos = (int32_t)((Ref1e + Ref2s) / 2); s.svlen = (int32_t)absd;
            s.q_start = (int32_t)Read1e; s.q_end = (int32_t)Read2s; emit = true;
          }
        }
        if (emit) { s.meta = meta; out = s; }
      }
    }
  }
  return out;
}

// SE_GROUP lanes per pair slot (in okey order); writes one signature row (possibly dead) per slot. Control flow is uniform
// per group up to the sums (the group shuffles need every lane of the group). SE_GROUP = 8 for reads (tens to hundreds of ops per
// CIGAR; a stray pair of long records is summed by all 64 lanes of the wave, one such pair after the other), SE_GROUP = 64 for
// contig alignments (10^4-10^6 ops per CIGAR): one pair per wave, so the pairs of a chromosome spread over the chip.
template <int SE_GROUP>
__global__ __launch_bounds__(256) void split_eval(RecView rv, const uint64_t* __restrict__ okey, const uint32_t* __restrict__ oval,
                                                  const uint64_t* __restrict__ ckey, const uint32_t* __restrict__ crec,
                                                  SplitCfg c, int dtype, int max_svlen, vsv_sig* __restrict__ s1in,
                                                  uint32_t cap, Counters* ctr, SlimOut sl, bool live_only) {
  const uint32_t n = live_only ? ctr->n_pairs : ctr->n_cand, n_raw = ctr->n_raw;     // (live_only: the dead slots were dropped by the sort)
  if (blockIdx.x == 0 && threadIdx.x == 0) { const uint32_t s1 = n_raw + n; ctr->n_s1 = s1 < cap ? s1 : cap; }
  const int lane = threadIdx.x & (SE_GROUP - 1), wlane = threadIdx.x & 63;
  const uint32_t ngroups = gridDim.x * (blockDim.x / SE_GROUP);
  const uint32_t len_ops = dtype == VSV_DTYPE_READS ? SE_LEN_OPS_READS : SE_LEN_OPS;
  // whole waves iterate together (uniform trip count) so that every shuffle sees all its lanes active
  const uint32_t n_pad = (n + (64 / SE_GROUP) - 1) / (64 / SE_GROUP) * (64 / SE_GROUP);
  for (uint32_t q0 = blockIdx.x * (blockDim.x / SE_GROUP) + (threadIdx.x / SE_GROUP); q0 < n_pad; q0 += ngroups) {
    const bool live = q0 < n;
    const uint32_t q = live ? q0 : n - 1;   // padding groups recompute the last slot and do not store
    const bool room = n_raw + q < cap;
    if (!room && lane == 0) atomicOr(&ctr->err, ERRB_CAPACITY);
    // ---- phase 1: which pair, and does it reach the length test (Hifi.py:315-324)? ------------------------------------
    bool need = false;
    uint32_t i1 = 0, i2 = 0, hap = 0, last1 = 0, first2 = 0;
    uint64_t a1 = 0, b1 = 0, a2 = 0, b2 = 0;
    if (room && okey[q] != VSV_KEY_DEAD) {
      const uint32_t j = oval[q];
      i1 = crec[j]; i2 = crec[j + 1];
      hap = (uint32_t)(ckey[j] >> c.qid_bits) & 1u;
      if (rv.pos[i1] > rv.pos[i2]) { if (lane == 0) atomicOr(&ctr->err, ERRB_UNSORTED); }   // Hifi.py:315
      else {
        const int minq = c.min_mapq;
        const uint32_t f1 = rv.flag[i1], f2 = rv.flag[i2];
        vsv_op_range(rv, i1, a1, b1);
        vsv_op_range(rv, i2, a2, b2);
        last1 = b1 > a1 ? rv.cigar[b1 - 1] : 0u; first2 = b2 > a2 ? rv.cigar[a2] : 0u;
        const uint32_t lop = last1 & 15u, fop = first2 & 15u;
        need = ((f1 ^ f2) & VSV_F_REVERSE) == 0 && rv.mapq[i1] >= minq && rv.mapq[i2] >= minq &&
               (lop == 4 || lop == 5) && (fop == 4 || fop == 5);                                  // Hifi.py:323-324
      }
    }
    // ---- phase 2: reference span of record 1, read lengths of both --------------------------------------------------
    const bool is_long = SE_GROUP < 64 && need && (b1 - a1) + (b2 - a2) > SE_LONG_OPS;
    int64_t rf1 = 0, rl1 = 0, rf2 = 0, rl2 = 0;
    if (need && !is_long) { op_sums<true>(rv.cigar, a1, b1, len_ops, lane, SE_GROUP, rf1, rl1); op_sums<false>(rv.cigar, a2, b2, len_ops, lane, SE_GROUP, rf2, rl2); }
    rf1 = group_sum64<SE_GROUP>(rf1); rl1 = group_sum64<SE_GROUP>(rl1); rl2 = group_sum64<SE_GROUP>(rl2);
    uint64_t lm = __ballot(is_long && lane == 0);
    while (lm) {                                       // wave-uniform: one long pair at a time, 64 lanes on its two CIGARs
      const int src = __builtin_ctzll(lm);
      lm &= lm - 1;
      const uint64_t xa1 = __shfl(a1, src, 64), xb1 = __shfl(b1, src, 64), xa2 = __shfl(a2, src, 64), xb2 = __shfl(b2, src, 64);
      int64_t f1 = 0, l1 = 0, f2 = 0, l2 = 0;
      op_sums<true>(rv.cigar, xa1, xb1, len_ops, (uint32_t)wlane, 64, f1, l1);
      op_sums<false>(rv.cigar, xa2, xb2, len_ops, (uint32_t)wlane, 64, f2, l2);
      f1 = wave_sum64(f1); l1 = wave_sum64(l1); l2 = wave_sum64(l2);
      if ((wlane & ~(SE_GROUP - 1)) == src) { rf1 = f1; rl1 = l1; rl2 = l2; }
    }
    // ---- phase 3: the pair rules, one lane per slot -------------------------------------------------------------------
    if (lane != 0 || !live || !room) continue;
    int32_t pos1 = 0, pos2 = 0, tid1 = 0;
    if (need) { pos1 = rv.pos[i1]; pos2 = rv.pos[i2]; tid1 = rv.tid[i1]; }
    const vsv_sig out = split_rules(need, i1, i2, pos1, pos2, tid1, hap, last1, first2, rf1, rl1, rl2, dtype, max_svlen, ctr);
    s1in[n_raw + q] = out;
    vsv_slim_emit(sl, n_raw + q, out);
  }
}

// ---- large candidate tables of read-shaped input (config 3: 7.2 M candidates, 1.5 M live pair slots of ~180-op ONT reads) ------------
// split_eval<8> above spends its time on scattered loads: a slot follows five dependent levels of them (slot -> candidate -> record
// fields of both records -> CIGAR edges -> the CIGAR sums), some twelve sectors per pair in front of the sums, a wave holds 8 slots and
// only lane 0 of a group evaluates the rules (ablation on config 3: 0.28 ms slot skeleton, 0.30 chain, 0.53 sums, 0.20 rules of 1.29 ms).
// Here the record fields a pair needs are collected once per CANDIDATE, in record order (cand_info: coalesced candidates, ascending
// gathers, 32 bytes out) as part of the candidate half of the stage, which a fused run enqueues next to the CIGAR scan; the sorts carry
// the candidates' ordinals (slim_path.hip SrcPairs / SrcPairSlots); and split_eval_info takes 64 live slots per wave: ONE lane per
// slot for the key, the two ordinals, the two entries, the rules and the row store, and the CIGAR sums (the bulk: 2.4 GB on config 3,
// 0.4 ms at HBM speed) in eight rounds of 8-lane groups — round r sums the slots of lanes 8r..8r+7 with every load of both CIGARs of a
// round, the two edge ops included, in flight before the first add.
struct __align__(16) CandInfo {
  uint32_t rec; int32_t pos; uint32_t fm, nops;        // record, POS, flag | mapq << 8, CIGAR ops
  uint64_t a; int32_t tid; uint32_t pad;               // first op of the CIGAR, tid
};
static_assert(sizeof(CandInfo) == 32, "CandInfo layout");
__global__ __launch_bounds__(256) void cand_info(RecView rv, const uint64_t* __restrict__ ckey, const uint32_t* __restrict__ crec, SplitCfg c,
                                                 CandInfo* __restrict__ out, const Counters* __restrict__ ctr) {
  const uint32_t n = ctr->n_cand;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    CandInfo ci;
    uint64_t a, b;
    ci.rec = crec[i];
    ci.tid = (int32_t)(ckey[i] >> c.tid_shift) + c.tid_lo;
    vsv_op_range(rv, ci.rec, a, b);
    ci.a = a; ci.nops = (uint32_t)(b - a < 0xFFFFFFFFull ? b - a : 0xFFFFFFFFull);      // (a CIGAR holds < 2^32 ops: cigar_off spans are checked by the scan)
    ci.pos = rv.pos[ci.rec];
    ci.fm = (uint32_t)rv.flag[ci.rec] | ((uint32_t)rv.mapq[ci.rec] << 8);
    ci.pad = 0;
    out[i] = ci;
  }
}
constexpr int SEW_DEPTH = 8;
// (same sums as op_sums<WANT_REF>(..., l, 8, ...); up to 8 x 16 bytes per lane in flight: 256 ops per trip; b > a is the caller's)
template <bool WANT_REF>
__device__ __forceinline__ void op_sums_burst(const uint32_t* __restrict__ cigar, uint64_t a, uint64_t b, uint32_t len_ops, uint32_t l, int64_t& rf, int64_t& rl) {
  uint32_t sr = 0, sq = 0;
  auto add = [&](uint32_t w) {
    const uint32_t op = w & 15u, len = w >> 4;
    if (WANT_REF) sr += len & se_mask(SE_REF_OPS, op);
    sq += len & se_mask(len_ops, op);
  };
  const uint64_t mis = ((uintptr_t)cigar >> 2) & 3u;
  const uint64_t a4 = ((a + mis + 3ull) & ~3ull) - mis, b4 = ((b + mis) & ~3ull) - mis;
  const uint64_t he = a4 < b ? a4 : b;
  uint32_t wh = 0, wt = 0;                               // (a packed word 0 = an empty M op: adds nothing)
  if (a + l < he) wh = cigar[a + l];
  if (b4 >= a4 && b4 + l < b) wt = cigar[b4 + l];
  const uint64_t nq = b4 > a4 ? (b4 - a4) >> 2 : 0;
  const uint4* __restrict__ qp = reinterpret_cast<const uint4*>(cigar + a4);
  for (uint64_t q0 = 0; q0 < nq; q0 += 8ull * SEW_DEPTH) {
    uint4 v[SEW_DEPTH];
#pragma unroll
    for (int u = 0; u < SEW_DEPTH; ++u) {
      const uint64_t q = q0 + (uint64_t)u * 8u + l;
      v[u] = q < nq ? qp[q] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < SEW_DEPTH; ++u) {
      add(v[u].x); add(v[u].y); add(v[u].z); add(v[u].w);
      if (u & 1) { if (WANT_REF) { rf += sr; sr = 0; } rl += sq; sq = 0; }     // (eight ops per flush: < 2^31)
    }
  }
  add(wh); add(wt);
  if (WANT_REF) rf += sr;
  rl += sq;
}
__device__ __forceinline__ bool is_clip(uint32_t w) { const uint32_t op = w & 15u; return op == 4u || op == 5u; }
__global__ __launch_bounds__(256) void split_eval_info(const uint32_t* __restrict__ cigar, const uint64_t* __restrict__ okey, const uint32_t* __restrict__ oc1,
                                                       const uint32_t* __restrict__ oc2, const CandInfo* __restrict__ cinfo, int min_mapq, int rec_bits,
                                                       int dtype, int max_svlen, vsv_sig* __restrict__ s1in, uint32_t cap, Counters* ctr, SlimOut sl) {
  const uint32_t n = ctr->n_pairs, n_raw = ctr->n_raw;           // (the live slots: the sort dropped the others)
  if (blockIdx.x == 0 && threadIdx.x == 0) { const uint32_t s1 = n_raw + n; ctr->n_s1 = s1 < cap ? s1 : cap; }
  const uint32_t lane = threadIdx.x & 63u, g = lane >> 3, gl = lane & 7u;
  const uint32_t nwaves = gridDim.x * (blockDim.x / 64), wave = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
  const uint32_t len_ops = dtype == VSV_DTYPE_READS ? SE_LEN_OPS_READS : SE_LEN_OPS;
  for (uint64_t q64 = (uint64_t)wave * 64u; q64 < n; q64 += (uint64_t)nwaves * 64u) {     // wave-uniform trip count
    const uint32_t q = (uint32_t)q64 + lane;
    const bool room = q < n && n_raw + q < cap;
    if (q < n && !room) atomicOr(&ctr->err, ERRB_CAPACITY);
    // ---- the pair, and what of the length test (Hifi.py:315-324) does not need the CIGARs ----
    bool pre = false;
    uint32_t i1 = 0, i2 = 0, hap = 0, n1 = 0, n2 = 0;
    int32_t pos1 = 0, pos2 = 0, tid1 = 0;
    uint64_t a1 = 0, a2 = 0;
    if (room) {
      const uint64_t ok = okey[q];
      if (ok != VSV_KEY_DEAD) {
        const CandInfo A = cinfo[oc1[q]], B = cinfo[oc2[q]];
        i1 = A.rec; i2 = B.rec; pos1 = A.pos; pos2 = B.pos; tid1 = A.tid; a1 = A.a; n1 = A.nops; a2 = B.a; n2 = B.nops;
        hap = (uint32_t)(ok >> rec_bits) & 1u;
        if (pos1 > pos2) atomicOr(&ctr->err, ERRB_UNSORTED);                                   // Hifi.py:315
        else pre = (((A.fm ^ B.fm) & VSV_F_REVERSE) == 0) && (int)(A.fm >> 8) >= min_mapq && (int)(B.fm >> 8) >= min_mapq && n1 > 0 && n2 > 0;
      }
    }
    // ---- reference span of record 1, read lengths of both, the last op of record 1 and the first of record 2 ----
    const bool is_long = pre && (uint64_t)n1 + n2 > SE_LONG_OPS;
    int64_t rf1 = 0, rl1 = 0, rl2 = 0;
    uint32_t last1 = 0, first2 = 0;
    const uint64_t todo = __ballot(pre && !is_long);
#pragma unroll 1
    for (uint32_t r = 0; r < 8; ++r) {
      if (((todo >> (8u * r)) & 0xFFull) == 0) continue;                                     // (wave-uniform)
      const int src = (int)(8u * r + g);
      const bool on = (todo >> src) & 1ull;
      const uint64_t xa1 = __shfl(a1, src, 64), xa2 = __shfl(a2, src, 64);
      const uint32_t xn1 = (uint32_t)__shfl((int)n1, src, 64), xn2 = (uint32_t)__shfl((int)n2, src, 64);
      int64_t f1 = 0, l1 = 0, f2 = 0, l2 = 0;
      uint32_t e = 0;
      if (on) {
        if (gl == 0) e = cigar[xa1 + xn1 - 1u];                                              // (the lines of the sums: no extra traffic)
        if (gl == 1) e = cigar[xa2];
        op_sums_burst<true>(cigar, xa1, xa1 + xn1, len_ops, gl, f1, l1);
        op_sums_burst<false>(cigar, xa2, xa2 + xn2, len_ops, gl, f2, l2);
      }
      f1 = group_sum64<8>(f1); l1 = group_sum64<8>(l1); l2 = group_sum64<8>(l2);
      // the owner of slot 8r + k is lane 8r + k; its sums sit in (every lane of) group k, its edge ops in lanes 0 and 1 of that group
      const int from = (int)((lane & 7u) * 8u);
      const int64_t of1 = __shfl(f1, from, 64), ol1 = __shfl(l1, from, 64), ol2 = __shfl(l2, from, 64);
      const uint32_t oe1 = (uint32_t)__shfl((int)e, from, 64), oe2 = (uint32_t)__shfl((int)e, from + 1, 64);
      if (g == r) { rf1 = of1; rl1 = ol1; rl2 = ol2; last1 = oe1; first2 = oe2; }
    }
    uint64_t lm = __ballot(is_long);
    while (lm) {                                       // wave-uniform: one long pair at a time, 64 lanes on its two CIGARs
      const int src = __builtin_ctzll(lm);
      lm &= lm - 1;
      const uint64_t xa1 = __shfl(a1, src, 64), xa2 = __shfl(a2, src, 64);
      const uint32_t xn1 = (uint32_t)__shfl((int)n1, src, 64), xn2 = (uint32_t)__shfl((int)n2, src, 64);
      int64_t f1 = 0, l1 = 0, f2 = 0, l2 = 0;
      op_sums<true>(cigar, xa1, xa1 + xn1, len_ops, lane, 64, f1, l1);
      op_sums<false>(cigar, xa2, xa2 + xn2, len_ops, lane, 64, f2, l2);
      f1 = wave_sum64(f1); l1 = wave_sum64(l1); l2 = wave_sum64(l2);
      if ((int)lane == src) { rf1 = f1; rl1 = l1; rl2 = l2; last1 = cigar[xa1 + xn1 - 1u]; first2 = cigar[xa2]; }
    }
    // ---- the rest of the length test, the pair rules and the row, one lane per slot ----
    if (!room) continue;
    const bool need = pre && is_clip(last1) && is_clip(first2);                                // Hifi.py:323-324
    const vsv_sig out = split_rules(need, i1, i2, pos1, pos2, tid1, hap, last1, first2, rf1, rl1, rl2, dtype, max_svlen, ctr);
    s1in[n_raw + q] = out;
    vsv_slim_emit(sl, n_raw + q, out);
  }
}

__global__ void set_n_s1(Counters* ctr, uint32_t cap) {
  uint32_t t = ctr->n_raw + ctr->n_cand;
  ctr->n_s1 = t < cap ? t : cap;
}

// ---- keys / gather / alive count --------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_keys(const vsv_sig* __restrict__ s, const uint32_t* __restrict__ d_n, int stage, int pb, int tid_lo,
                                                  int tid_bits, uint64_t* __restrict__ key, uint32_t* __restrict__ idx, Counters* ctr) {
  const uint32_t n = *d_n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const vsv_sig v = s[i];
    uint64_t k;
    if (stage == 5) {  // READS final order (reads.py:281-286): (tid,pos) then source, DEL before INS, list order
      k = (v.meta & VSV_M_DEAD) ? VSV_KEY_DEAD
          : ((uint64_t)(uint32_t)(v.tid - tid_lo) << (pb + 2)) | (vsv_kpos(v.pos) << 2) | ((v.meta & VSV_M_SPLIT) ? 2u : 0u) | ((v.meta & VSV_M_DEL) ? 0u : 1u);
    } else k = vsv_key_stage(v, stage, pb, tid_lo);
    if (!(v.meta & VSV_M_DEAD) && ((pb < 32 && (vsv_kpos(v.pos) >> pb) != 0) || ((uint32_t)(v.tid - tid_lo) >> tid_bits) != 0))
      atomicOr(&ctr->err, ERRB_RANGE);  // max_pos hint too small / tid outside [tid_lo, n_tids)
    key[i] = k;
    idx[i] = i;
  }
}
__global__ __launch_bounds__(256) void build_call_keys(const vsv_call* __restrict__ c, const uint32_t* __restrict__ d_n, int pb, int tid_lo,
                                                       uint64_t* __restrict__ key, uint32_t* __restrict__ idx) {
  const uint32_t n = *d_n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    key[i] = vsv_key_stage(c[i].sig, 4, pb, tid_lo);
    idx[i] = i;
  }
}
// payload gather after a sort; also publishes the alive row count (keys ascend with dead = all ones last: the count is
// the index of the first dead key) and clears the long-run queue for the cluster / pair kernel that follows
template <typename T>
__global__ __launch_bounds__(256) void gather_rows(const T* __restrict__ in, const uint32_t* __restrict__ idx,
                                                   const uint64_t* __restrict__ key, const uint32_t* __restrict__ d_n,
                                                   T* __restrict__ out, uint32_t* __restrict__ d_alive, uint32_t* __restrict__ n_long,
                                                   int32_t* __restrict__ fill) {
  const uint32_t n = *d_n;
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_long = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    out[i] = in[idx[i]];
    if (fill) fill[i] = -1;                // pairing state of the stage that follows (pair_kernel: -1 = unpaired)
    if (key[i] != VSV_KEY_DEAD && (i + 1 == n || key[i + 1] == VSV_KEY_DEAD)) *d_alive = i + 1;
  }
}
// ---- seeded greedy clustering ------------------------------------------------------------------------
// A run = maximal stretch of one list whose consecutive positions differ by <= max_shift; no match can
// cross a run boundary (every match needs shift <= max_shift), so runs are independent and the
// reference's full scans (Hifi.py:207-226) reduce to a scan inside the run. One lane owns one run and
// performs the sequential greedy exactly: seed = first unassigned, members = unassigned rows matching
// the SEED, representative = first longest member (Hifi.py:236-247). Output row i = representative
// if i is a seed, dead otherwise, so seed order is preserved.
constexpr uint32_t LONG_RUN = 48;   // runs / stretches longer than this go to the wave-cooperative kernels

__device__ __forceinline__ int32_t ld_i32(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_i32(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One wave on a long run: seeds stay sequential (the greedy is order dependent), the scan of a seed's window and the
// search for the next seed are 64-wide. cl[] is accessed with agent-scope relaxed atomics (L2), so the wave sees its
// own earlier stores whatever the L1 holds. Called by ALL 64 lanes with wave-uniform (i, e).
__device__ __forceinline__ void cluster_long_run(const vsv_sig* __restrict__ s, int max_shift, int32_t* __restrict__ cl,
                                                 vsv_sig* __restrict__ out, const uint32_t i, const uint32_t e, const int lane) {
  for (uint32_t k = i + lane; k < e; k += 64) st_i32(&cl[k], -1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the wave's (L2, sc1) stores are acknowledged before its next loads
  uint32_t a = i;
  while (a < e) {
    const vsv_sig s1 = s[a];
    if (lane == 0) st_i32(&cl[a], (int32_t)a);
    uint64_t best = ((uint64_t)(uint32_t)s1.svlen << 32) | (0xFFFFFFFFu - a);   // max length, then lowest index
    for (uint32_t b0 = a + 1; b0 < e; b0 += 64) {
      const uint32_t b = b0 + lane;
      const bool valid = b < e;
      vsv_sig s2 = s1;
      if (valid) s2 = s[b];
      const bool inwin = valid && (int64_t)s2.pos - s1.pos <= max_shift;
      if (__ballot(inwin) == 0) break;          // positions ascend: the whole tile is beyond the window
      if (inwin && ld_i32(&cl[b]) == -1 && vsv_match(s1, s2, max_shift)) {
        st_i32(&cl[b], (int32_t)a);
        const uint64_t c = ((uint64_t)(uint32_t)s2.svlen << 32) | (0xFFFFFFFFu - b);
        if (c > best) best = c;
      }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const uint64_t o = __shfl_xor(best, d, 64); if (o > best) best = o; }
    if (lane == 0) out[a] = s[0xFFFFFFFFu - (uint32_t)best];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // next seed: first unassigned row after a
    uint32_t nxt = e;
    for (uint32_t b0 = a + 1; b0 < e; b0 += 64) {
      const uint32_t b = b0 + lane;
      const uint64_t free_m = __ballot(b < e && ld_i32(&cl[b]) == -1);
      if (free_m) { nxt = b0 + (uint32_t)__builtin_ctzll(free_m); break; }
    }
    a = nxt;
  }
  for (uint32_t k = i + lane; k < e; k += 64) if (ld_i32(&cl[k]) != (int32_t)k) out[k] = dead_sig();
}

// A lane owns the run that starts at its row (short runs: the sequential greedy, exactly); runs of more than LONG_RUN rows are
// taken over by the lane's whole wave right away, one after the other (no queue, no second launch: the waves that find dense
// regions are spread over the grid like the regions over the table).
__global__ __launch_bounds__(256) void cluster_kernel(const vsv_sig* __restrict__ s, const uint64_t* __restrict__ key,
                                                      const uint32_t* __restrict__ d_n, int max_shift, int pb,
                                                      int32_t* __restrict__ cl, vsv_sig* __restrict__ out) {
  const uint32_t n = *d_n;
  const int lane = threadIdx.x & 63;
  for (uint32_t i0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += gridDim.x * blockDim.x) {   // wave-uniform trip count
    const uint32_t i = i0 + (uint32_t)lane;
    bool is_long = false;
    uint32_t e = 0;
    if (i < n) {
      const uint64_t lk = key[i] >> pb;
      const bool head = !(i > 0 && (key[i - 1] >> pb) == lk && (int64_t)s[i].pos - s[i - 1].pos <= max_shift);
      if (head) {
        e = i + 1;
        while (e < n && e - i <= LONG_RUN && (key[e] >> pb) == lk && (int64_t)s[e].pos - s[e - 1].pos <= max_shift) ++e;
        if (e - i > LONG_RUN) {
          is_long = true;
          while (e < n && (key[e] >> pb) == lk && (int64_t)s[e].pos - s[e - 1].pos <= max_shift) ++e;
        } else {
          for (uint32_t k = i; k < e; ++k) cl[k] = -1;
          for (uint32_t a = i; a < e; ++a) {
            if (cl[a] != -1) { out[a] = dead_sig(); continue; }
            cl[a] = (int32_t)a;
            const vsv_sig s1 = s[a];
            uint32_t best = a;
            int32_t best_len = s1.svlen;
            for (uint32_t b = a + 1; b < e; ++b) {
              const vsv_sig s2 = s[b];
              if ((int64_t)s2.pos - s1.pos > max_shift) break;
              if (cl[b] != -1) continue;
              if (vsv_match(s1, s2, max_shift)) {
                cl[b] = (int32_t)a;
                if (s2.svlen > best_len) { best = b; best_len = s2.svlen; }
              }
            }
            out[a] = s[best];
          }
        }
      }
    }
    uint64_t lm = __ballot(is_long);
    while (lm) {
      const int src = __builtin_ctzll(lm);
      lm &= lm - 1;
      cluster_long_run(s, max_shift, cl, out, (uint32_t)__shfl((int)i, src, 64), (uint32_t)__shfl((int)e, src, 64), lane);
    }
  }
}

// ---- haplotype pairing --------------------------------------------------------------------------------
// merged table m sorted by (tid, hap, pos), keys = stage-3 keys. One lane owns a stretch of hp1 rows whose
// consecutive positions differ by <= 2*pair_shift (candidate windows of different stretches are disjoint,
// so stretches are independent); inside it the greedy of Hifi.py:552-569 runs in hp1 order: first unpaired
// hp2 row of the same type within pair_shift that matches. The reference scans from j=0 and breaks at
// pos2 - pos1 > pair_window (H:557-559), so a mate lies at most `right` = min(pair_shift, pair_window) to the right and at
// most pair_shift to the left: the same candidates in the same order.
__device__ __forceinline__ uint32_t lower_bound_key(const uint64_t* __restrict__ key, uint32_t n, uint64_t target) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (key[mid] >= target) hi = mid; else lo = mid + 1; }
  return lo;
}
// One wave on a long hp1 stretch: hp1 rows stay sequential (first come, first served), the candidate window of a row is
// scanned 64 hp2 rows at a time and the first match is the lowest set ballot bit. Called by all 64 lanes, (i, jlo) wave-uniform.
__device__ __forceinline__ void pair_long_stretch(const vsv_sig* __restrict__ m, const uint64_t* __restrict__ key, const uint32_t n,
                                                  int pair_shift, int right, int sh_hap, int32_t* __restrict__ st2,
                                                  vsv_call* __restrict__ out, const uint32_t i, uint32_t jlo, const int lane,
                                                  uint32_t* __restrict__ max_stretch) {
  const uint64_t hp1_prefix = key[i] >> sh_hap, hp2_prefix = hp1_prefix | 1ull;
  uint32_t a = i;
  for (; a < n; ++a) {
    const vsv_sig s1 = m[a];
    if ((key[a] >> sh_hap) != hp1_prefix) break;
    if (a > i && (int64_t)s1.pos - m[a - 1].pos > 2 * (int64_t)pair_shift) break;
    while (jlo < n && (key[jlo] >> sh_hap) == hp2_prefix && (int64_t)s1.pos - m[jlo].pos > pair_shift) ++jlo;
    int32_t mate = -1;
    for (uint32_t j0 = jlo; j0 < n; j0 += 64) {
      const uint32_t j = j0 + lane;
      const bool valid = j < n && (key[j] >> sh_hap) == hp2_prefix;
      vsv_sig s2 = s1;
      if (valid) s2 = m[j];
      const bool inwin = valid && (int64_t)s2.pos - s1.pos <= right;
      if (__ballot(inwin) == 0) break;
      const bool ok = inwin && ((s1.meta ^ s2.meta) & VSV_M_DEL) == 0 && ld_i32(&st2[j]) == -1 && vsv_match(s1, s2, pair_shift);
      const uint64_t bal = __ballot(ok);
      if (bal) { mate = (int32_t)(j0 + (uint32_t)__builtin_ctzll(bal)); break; }
    }
    if (lane == 0) {
      vsv_call c;
      c.a = (int32_t)a; c.pad = 0;
      if (mate < 0) { c.sig = s1; c.b = -1; c.gt = 1; }
      else { st_i32(&st2[mate], (int32_t)a); const vsv_sig s2 = m[mate]; c.sig = (s1.svlen > s2.svlen) ? s1 : s2; c.b = mate; c.gt = 2; }
      out[a] = c;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (lane == 0) atomicMax(max_stretch, a - i);     // a table with very long stretches is paired in rounds from the next run on
}

// ---- pairing in rounds: dense tables ------------------------------------------------------------------------------------
// When signatures lie closer than 2 * pair_shift for whole chromosomes (hundreds of thousands of contigs piled on one
// reference), a "stretch" is the entire haplotype list and the first-come-first-served walk above is one wave's job. The
// same result in parallel ("deterministic reservations"): every undecided hp1 row reserves ALL its free candidates with its
// index as priority (atomicMin; smaller index = earlier in the reference's order), then takes its FIRST free candidate if it
// holds the reservation there — no earlier undecided row can then ever want that hp2 row, and the rows already decided are
// final, so this is exactly what the sequential walk would give it; a row without a free candidate is unpaired for good (free
// candidates only disappear). Rows that lose wait for the next round; what is left after the rounds goes through the
// sequential rule in chains of neighbouring undecided rows (pair_leftover). Reservations carry the round in their high word,
// so they need no clearing between rounds.
__device__ __forceinline__ bool pair_candidate(const vsv_sig& s1, const vsv_sig& s2, int pair_shift) {
  return ((s1.meta ^ s2.meta) & VSV_M_DEL) == 0 && vsv_match(s1, s2, pair_shift);
}
__global__ __launch_bounds__(256) void pair_rounds_prep(const vsv_sig* __restrict__ m, const uint64_t* __restrict__ key, const uint32_t* __restrict__ d_n,
                                                        int pair_shift, int pb, uint32_t* __restrict__ jlo, uint32_t* __restrict__ done1,
                                                        uint64_t* __restrict__ res) {
  const uint32_t n = *d_n;
  const int sh_hap = pb + 2, sh_tid = pb + 3;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const vsv_sig me = m[i];
    done1[i] = 0;
    res[i] = ~0ull;
    if (me.meta & VSV_M_HP2) { jlo[i] = 0; continue; }
    const uint64_t tk = key[i] >> sh_tid;
    jlo[i] = lower_bound_key(key, n, (tk << sh_tid) | (1ull << sh_hap) | vsv_kpos((int32_t)max((int64_t)me.pos - pair_shift, (int64_t)-VSV_POS_BIAS)));
  }
}
template <bool COMMIT>
__global__ __launch_bounds__(256) void pair_round(const vsv_sig* __restrict__ m, const uint64_t* __restrict__ key, const uint32_t* __restrict__ d_n,
                                                  int pair_shift, int right, int pb, int32_t* __restrict__ st2, vsv_call* __restrict__ out,
                                                  const uint32_t* __restrict__ jlo, uint32_t* __restrict__ done1, uint64_t* __restrict__ res,
                                                  uint32_t round) {
  const uint32_t n = *d_n;
  const int sh_hap = pb + 2;
  const uint64_t stamp = (uint64_t)(~round) << 32;          // newer rounds compare smaller: old reservations lose by themselves
  for (uint32_t a = blockIdx.x * blockDim.x + threadIdx.x; a < n; a += gridDim.x * blockDim.x) {
    const vsv_sig s1 = m[a];
    if ((s1.meta & VSV_M_HP2) || done1[a]) continue;
    const uint64_t hp2_prefix = (key[a] >> sh_hap) | 1ull;
    int32_t first = -1;
    for (uint32_t j = jlo[a]; j < n && (key[j] >> sh_hap) == hp2_prefix; ++j) {
      const vsv_sig s2 = m[j];
      if ((int64_t)s2.pos - s1.pos > right) break;
      if (ld_i32(&st2[j]) != -1 || !pair_candidate(s1, s2, pair_shift)) continue;
      if (!COMMIT) atomicMin((unsigned long long*)&res[j], (unsigned long long)(stamp | a));
      else { first = (int32_t)j; break; }
    }
    if (!COMMIT) continue;
    vsv_call c;
    c.a = (int32_t)a; c.pad = 0;
    if (first < 0) { c.sig = s1; c.b = -1; c.gt = 1; out[a] = c; done1[a] = 1; }                       // Hifi.py:575-576
    else if (res[first] == (stamp | a)) {
      st_i32(&st2[first], (int32_t)a);
      const vsv_sig s2 = m[first];
      c.sig = (s1.svlen > s2.svlen) ? s1 : s2; c.b = first; c.gt = 2;                                   // Hifi.py:583-586
      out[a] = c; done1[a] = 1;
    }
  }
}
// what the rounds left undecided: chains of undecided hp1 rows whose neighbours lie within 2 * pair_shift (rows further apart
// share no candidate), one lane per chain, the sequential rule inside
__global__ __launch_bounds__(256) void pair_leftover(const vsv_sig* __restrict__ m, const uint64_t* __restrict__ key, const uint32_t* __restrict__ d_n,
                                                     int pair_shift, int right, int pb, int32_t* __restrict__ st2, vsv_call* __restrict__ out,
                                                     const uint32_t* __restrict__ jlo, const uint32_t* __restrict__ done1) {
  const uint32_t n = *d_n;
  const int sh_hap = pb + 2;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const vsv_sig me = m[i];
    if ((me.meta & VSV_M_HP2) || done1[i]) continue;
    const uint64_t hp1_prefix = key[i] >> sh_hap, hp2_prefix = hp1_prefix | 1ull;
    bool head = true;
    for (uint32_t k = i; k-- > 0;) {                        // an undecided hp1 row of this list within 2 * pair_shift in front?
      if ((key[k] >> sh_hap) != hp1_prefix || (int64_t)me.pos - m[k].pos > 2 * (int64_t)pair_shift) break;
      if (!done1[k]) { head = false; break; }
    }
    if (!head) continue;
    int32_t last_pos = me.pos;
    for (uint32_t a = i; a < n && (key[a] >> sh_hap) == hp1_prefix; ++a) {
      const vsv_sig s1 = m[a];
      if ((int64_t)s1.pos - last_pos > 2 * (int64_t)pair_shift) break;      // the next undecided row, if any, leads its own chain
      if (done1[a]) continue;
      last_pos = s1.pos;
      int32_t mate = -1;
      for (uint32_t j = jlo[a]; j < n && (key[j] >> sh_hap) == hp2_prefix; ++j) {
        const vsv_sig s2 = m[j];
        if ((int64_t)s2.pos - s1.pos > right) break;
        if (ld_i32(&st2[j]) == -1 && pair_candidate(s1, s2, pair_shift)) { mate = (int32_t)j; st_i32(&st2[j], (int32_t)a); break; }
      }
      vsv_call c;
      c.a = (int32_t)a; c.pad = 0;
      if (mate < 0) { c.sig = s1; c.b = -1; c.gt = 1; }
      else { const vsv_sig s2 = m[mate]; c.sig = (s1.svlen > s2.svlen) ? s1 : s2; c.b = mate; c.gt = 2; }
      out[a] = c;                          // (done1 stays as the rounds left it: the other lanes' head tests read it meanwhile)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
}

// A lane owns the stretch its hp1 row leads; stretches of more than LONG_RUN rows are taken over by the lane's whole wave.
__global__ __launch_bounds__(256) void pair_kernel(const vsv_sig* __restrict__ m, const uint64_t* __restrict__ key,
                                                   const uint32_t* __restrict__ d_n, int pair_shift, int right, int pb,
                                                   int32_t* __restrict__ st2, vsv_call* __restrict__ out, uint32_t* __restrict__ max_stretch) {
  const uint32_t n = *d_n;
  const int lane = threadIdx.x & 63;
  const int sh_hap = pb + 2, sh_tid = pb + 3;
  for (uint32_t i0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); i0 < n; i0 += gridDim.x * blockDim.x) {   // wave-uniform trip count
    const uint32_t i = i0 + (uint32_t)lane;
    bool is_long = false;
    uint32_t jlo = 0;
    if (i < n) {
      const vsv_sig me = m[i];
      // stretches are led by hp1 rows
      const bool head = !(me.meta & VSV_M_HP2) &&
                        !(i > 0 && (key[i - 1] >> sh_hap) == (key[i] >> sh_hap) && (int64_t)me.pos - m[i - 1].pos <= 2 * (int64_t)pair_shift);
      if (head) {
        const uint64_t tk = key[i] >> sh_tid;
        // first hp2 row of this tid with pos >= pos_i - pair_shift (rows further left can never match); the hp2 block ends where
        // the (tid, hap) prefix of the key changes, so no search for its bounds is needed
        const uint64_t hp2_prefix = (tk << 1) | 1ull;
        jlo = lower_bound_key(key, n, (tk << sh_tid) | (1ull << sh_hap) | vsv_kpos((int32_t)max((int64_t)me.pos - pair_shift, (int64_t)-VSV_POS_BIAS)));
        uint32_t e = i + 1;
        while (e < n && e - i <= LONG_RUN && (key[e] >> sh_hap) == (key[i] >> sh_hap) && (int64_t)m[e].pos - m[e - 1].pos <= 2 * (int64_t)pair_shift) ++e;
        if (e - i > LONG_RUN) is_long = true;
        else {
          for (uint32_t a = i; a < e; ++a) {
            const vsv_sig s1 = m[a];
            while (jlo < n && (key[jlo] >> sh_hap) == hp2_prefix && (int64_t)s1.pos - m[jlo].pos > pair_shift) ++jlo;
            int32_t mate = -1;
            for (uint32_t j = jlo; j < n && (key[j] >> sh_hap) == hp2_prefix; ++j) {
              const vsv_sig s2 = m[j];
              if ((int64_t)s2.pos - s1.pos > right) break;
              if (((s1.meta ^ s2.meta) & VSV_M_DEL) == 0 && st2[j] == -1 && vsv_match(s1, s2, pair_shift)) {
                mate = (int32_t)j; st2[j] = (int32_t)a; break;                 // Hifi.py:560-569
              }
            }
            vsv_call c;
            c.a = (int32_t)a; c.pad = 0;
            if (mate < 0) { c.sig = s1; c.b = -1; c.gt = 1; }                  // Hifi.py:575-576
            else { const vsv_sig s2 = m[mate]; c.sig = (s1.svlen > s2.svlen) ? s1 : s2; c.b = mate; c.gt = 2; }  // Hifi.py:583-586
            out[a] = c;
          }
        }
      }
    }
    uint64_t lm = __ballot(is_long);
    while (lm) {
      const int src = __builtin_ctzll(lm);
      lm &= lm - 1;
      pair_long_stretch(m, key, n, pair_shift, right, sh_hap, st2, out, (uint32_t)__shfl((int)i, src, 64), (uint32_t)__shfl((int)jlo, src, 64), lane, max_stretch);
    }
  }
}

// hp2 rows: unpaired -> 0/1 call (Hifi.py:588-592), paired -> dead slot; and the final sort's (tid, pos) key of every call
__global__ __launch_bounds__(256) void pair_finish(const vsv_sig* __restrict__ m, const uint32_t* __restrict__ d_n,
                                                   const int32_t* __restrict__ st2, vsv_call* __restrict__ out, int pb, int tid_lo,
                                                   uint64_t* __restrict__ key, uint32_t* __restrict__ idx) {
  const uint32_t n = *d_n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const vsv_sig me = m[i];
    vsv_sig kept;
    if (me.meta & VSV_M_HP2) {
      vsv_call c;
      c.pad = 0; c.a = -1; c.b = (int32_t)i; c.gt = 1;
      c.sig = (st2[i] == -1) ? me : dead_sig();
      out[i] = c;
      kept = c.sig;
    } else {
      kept = out[i].sig;                   // written by pair_kernel / pair_long_kernel
    }
    key[i] = vsv_key_stage(kept, 4, pb, tid_lo);
    idx[i] = i;
  }
}
__global__ void fill_i32(int32_t* p, int32_t v, const uint32_t* d_n) {
  const uint32_t n = *d_n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}
template <typename T>
__global__ void copy_rows(const T* __restrict__ in, const uint32_t* __restrict__ d_n, T* __restrict__ out) {
  const uint32_t n = *d_n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[i];
}

constexpr int EW_GRID = 128;

}  // namespace

// ======================================= host-side launchers ==========================================
void vsv_launch_clr_gate(hipStream_t st, const RecView& rv, uint8_t* gflag, Counters* ctr) {
  if (rv.n_records <= 0) return;
  if (rv.n_ops / rv.n_records >= 256) clr_gate_records<64><<<2048, 256, 0, st>>>(rv, gflag, ctr);
  else clr_gate_records<8><<<2048, 256, 0, st>>>(rv, gflag, ctr);
}

// raw (T_RAW, written by place_raw) -> s1in: folded on the contig path, combined for sig_extract, a plain copy otherwise
void vsv_launch_fold(hipStream_t st, const vsv_sig* raw, vsv_sig* s1in, const RecView& rv, const vsv_params& p, Counters* ctr, int grid, const SlimOut& so) {
  const int dtype = p.dtype;
  if (dtype == VSV_DTYPE_HIFI || dtype == VSV_DTYPE_ONT || dtype == VSV_DTYPE_CLR) { fold_kernel<<<grid, 256, 0, st>>>(raw, s1in, ctr, so); return; }
  copy_rows<vsv_sig><<<grid, 256, 0, st>>>(raw, &ctr->n_raw, s1in);
  if (dtype == VSV_DTYPE_CUTESV) combine_kernel<<<EW_GRID, 256, 0, st>>>(s1in, ctr, p.merge_ins_threshold, p.merge_del_threshold);
}

void vsv_launch_fold_elems(hipStream_t st, vsv_sig* rows, Counters* ctr, int grid, const SlimOut& so) {
  fold_elems<<<grid, 256, 0, st>>>(rows, (uint4*)so.base, ctr, so);
}

static int bits_for(uint64_t n) { int b = 1; while ((1ull << b) < n && b < 63) ++b; return b; }

static SplitCfg split_cfg(const RecView& rv, const vsv_params& p, int n_tids) {
  SplitCfg c;
  c.contig = p.dtype != VSV_DTYPE_READS;
  c.min_mapq = p.min_split_mapq;
  c.qid_bits = bits_for((uint64_t)(rv.n_qids > 0 ? rv.n_qids : rv.n_records) + 1);
  c.tid_shift = c.qid_bits + 1;
  c.tid_lo = rv.tid_lo;
  c.tid_bits = bits_for((uint64_t)(n_tids > 0 ? n_tids - rv.tid_lo : 65536) + 1);
  return c;
}

// The split stage in two halves. The first — which query names occur more than once, the candidate records, their two orderings
// (by name, then by record) — reads nothing but the record arrays: the fused run enqueues it on the handle's auxiliary stream,
// next to the CIGAR scan (capi.hip enq_scan). The second, split_eval, appends its rows behind the scan's (n_raw) and joins.
SplitSorted vsv_launch_split_candidates(hipStream_t st, const RecView& rv, const vsv_params& p, int n_tids, uint32_t* tab, uint32_t tab_size,
                                        uint32_t* blk_cnt, uint32_t* blk_off, uint32_t* scan_tmp, uint64_t* ckey, uint32_t* crec,
                                        uint64_t* okey, uint32_t* oval, uint64_t* key2, uint32_t* idx2, const SortWork& sw,
                                        uint32_t cap, Counters* ctr, uint8_t* cmask, int grid, const SlimWork* slim, const CandBufs& cb, int phase,
                                        const CandLb& lb) {
  (void)blk_off; (void)scan_tmp;
  SplitSorted out{nullptr, nullptr, nullptr, nullptr};
  if (rv.n_records <= 0) return out;
  const SplitCfg c = split_cfg(rv, p, n_tids);
  const int rec_bits = bits_for((uint64_t)rv.n_records + 1);
  const uint64_t nq = rv.n_qids > 0 ? (uint64_t)rv.n_qids : (uint64_t)tab_size * 32;
  const int qtiles = (int)((rv.n_records + QM_TILE - 1) / QM_TILE);
  // wide loads need 16-byte (qid) / 4-byte (flag, mapq) aligned arrays; anything else takes the element-wise path
  const bool vec = ((uintptr_t)rv.qid & 15u) == 0 && ((uintptr_t)rv.flag & 3u) == 0 && ((uintptr_t)rv.mapq & 3u) == 0;
  // phase 1 = the candidates only (name repeats, count, fill: the same kernels whatever the tables' sizes: a cold handle runs them
  // beside its scan and takes the candidate count from them), phase 2 = their two sorts only, 0 = both
  if (phase != 2) {
    if (lb.qwords) {              // large shards: one launch each, scans by look-back
      qid_dups<<<(qtiles + QD_SUB - 1) / QD_SUB, 256, 0, st>>>(rv.qid, rv.n_records, tab, vec, (uint32_t)nq, &ctr->err, lb.qwords, lb.epoch, lb.tab_next, lb.clear_words);
      const int nblk = (int)(((uint64_t)rv.n_records + SC_TILE_REC - 1) / SC_TILE_REC);
      split_cand<<<nblk, 256, 0, st>>>(rv, c, tab, ckey, crec, cap, vec, ctr, lb.cwords, lb.epoch);
    } else {                      // up to 8192 tiles: the consumers reduce the per-tile values in front of them themselves
      qid_tile_max<<<qtiles, 256, 0, st>>>(rv.qid, rv.n_records, blk_cnt, vec, tab, (uint32_t)((nq + 31) / 32 + 1));
      qid_mark_dups<<<qtiles, 256, 0, st>>>(rv.qid, rv.n_records, blk_cnt, tab, vec, (uint32_t)nq, &ctr->err, true);
      const int nblk = (int)(((uint64_t)rv.n_records + SC2_TILE_REC - 1) / SC2_TILE_REC);
      split_cand_2pass<false><<<nblk, 256, 0, st>>>(rv, c, tab, blk_cnt, ckey, crec, cap, vec, ctr, cmask);
      split_cand_2pass<true><<<nblk, 256, 0, st>>>(rv, c, tab, blk_cnt, ckey, crec, cap, vec, ctr, cmask, true);
    }
  }
  if (phase == 1) return out;
  if (slim) {
    // large inputs (config 3: ~10^7 candidates): both sorts through the 8-bit passes over 16-byte elements (slim_path.hip; the
    // 9-11 bit passes over separate key / value arrays write 16- and 8-byte pieces: 0.3-0.4 ms per pass there). Dead pair slots
    // are dropped by the second sort's first pass: the live ones come first, as before.
    // read-shaped input: what a pair needs of its two records is computed per candidate, here (cand_info), and the sorts carry the
    // candidates' ordinals; contig alignments (10^4-10^6 ops per CIGAR, few candidates) keep the per-pair sums of split_eval<64>
    static const char* info_mode = vsv_dbg_env("VSV_SPLIT_INFO");          // timing experiments: "0" = per-pair gathers (split_eval<8>)
    const bool info = cb.cinfo && !vsv_scan_is_long(rv, p) && !(info_mode && info_mode[0] == '0');
    if (info) {
      cand_info<<<grid, 256, 0, st>>>(rv, ckey, crec, c, (CandInfo*)cb.cinfo, ctr);
    }
    const SortResult r1 = vsv_slim_sort_pairs(st, ckey, crec, &ctr->n_cand, c.tid_shift + c.tid_bits, sw.key_alt, sw.val_alt, &ctr->n_pairs, *slim,
                                              info ? cb.cord : nullptr);
    const SortResult r2 = vsv_slim_sort_pair_slots(st, r1.key, r1.val, c.qid_bits, rec_bits, &ctr->n_cand, rec_bits + 1 + c.tid_bits, okey, oval, &ctr->n_pairs, *slim,
                                                   info ? cb.cord : nullptr, info ? cb.oc1 : nullptr);
    out.ckey = r1.key; out.crec = r1.val; out.okey = r2.key; out.oval = r2.val;
    out.live_only = true;                              // ctr->n_pairs = the live slots, in front
    if (info) { out.oc1 = cb.oc1; out.cinfo = cb.cinfo; }
    return out;
  }
  // the first result may live in the shared scratch pair, so the second sort gets a scratch pair of its own
  const SortResult r1 = vsv_radix_sort_pairs(st, ckey, crec, sw.key_alt, sw.val_alt, &ctr->n_cand, cap, c.tid_shift + c.tid_bits, sw,
                                             1ull << (c.tid_shift + c.tid_bits));     // no dead keys in this table: the whole range is alive
  // pair slots by record: the bucket sort takes their keys straight from the sorted candidates; the LSD passes want them written out
  SortResult r2 = vsv_bucket_sort_pair_slots(st, r1.key, r1.val, c.qid_bits, rec_bits, &ctr->n_cand, okey, oval, key2, idx2, cap, rec_bits + 1 + c.tid_bits + 1, sw);
  if (!r2.key) {
    split_mark_pairs<<<grid, 256, 0, st>>>(r1.key, r1.val, c, rec_bits, okey, oval, ctr);
    r2 = vsv_radix_sort_pairs(st, okey, oval, key2, idx2, &ctr->n_cand, cap, rec_bits + 1 + c.tid_bits + 1, sw);
  }
  out.ckey = r1.key; out.crec = r1.val; out.okey = r2.key; out.oval = r2.val;
  return out;
}

void vsv_launch_split_eval(hipStream_t st, const RecView& rv, const vsv_params& p, int n_tids, const SplitSorted& so, vsv_sig* s1in,
                           uint32_t cap, Counters* ctr, int grid, const SlimOut& sl) {
  if (rv.n_records <= 0 || !so.okey) { set_n_s1<<<1, 1, 0, st>>>(ctr, cap); return; }
  const SplitCfg c = split_cfg(rv, p, n_tids);
  const int eg = grid * 8 < 1024 ? 1024 : (grid * 8 > 8192 ? 8192 : grid * 8);
  if (so.cinfo) {
    const int rec_bits = bits_for((uint64_t)rv.n_records + 1);
    split_eval_info<<<eg, 256, 0, st>>>(rv.cigar, so.okey, so.oc1, so.oval, (const CandInfo*)so.cinfo, c.min_mapq, rec_bits, p.dtype, p.max_split_svlen, s1in, cap, ctr, sl);
  } else if (vsv_scan_is_long(rv, p)) split_eval<64><<<eg, 256, 0, st>>>(rv, so.okey, so.oval, so.ckey, so.crec, c, p.dtype, p.max_split_svlen, s1in, cap, ctr, sl, so.live_only);
  else split_eval<8><<<eg, 256, 0, st>>>(rv, so.okey, so.oval, so.ckey, so.crec, c, p.dtype, p.max_split_svlen, s1in, cap, ctr, sl, so.live_only);
}

// sort rows `in[0,n)` by the stage key into `sorted`, publish the alive count
const uint64_t* vsv_launch_sort_stage(hipStream_t st, const vsv_sig* in, const uint32_t* d_n, int stage, int pb, int nbits,
                                      vsv_sig* sorted, uint32_t* d_alive, const StageBufs& b, const SortWork& sw, int64_t cap,
                                      Counters* ctr, int32_t* fill_minus1) {
  const uint64_t* fast = vsv_bucket_sort_sigs(st, in, d_n, stage, pb, b.tid_lo, b.tid_bits, nbits, stage == 5 ? 0 : b.kmax, sorted, b.key, d_alive,
                                              &ctr->n_long, fill_minus1, sw, cap);
  if (fast) return fast;
  build_keys<<<b.grid, 256, 0, st>>>(in, d_n, stage, pb, b.tid_lo, b.tid_bits, b.key, b.idx, ctr);
  const SortResult r = vsv_radix_sort_pairs(st, b.key, b.idx, sw.key_alt, sw.val_alt, d_n, cap, nbits, sw, stage == 5 ? 0 : b.kmax);
  gather_rows<vsv_sig><<<b.grid, 256, 0, st>>>(in, r.val, r.key, d_n, sorted, d_alive, &ctr->n_long, fill_minus1);
  return r.key;
}

void vsv_launch_cluster(hipStream_t st, const vsv_sig* sorted, const uint64_t* sorted_key, const uint32_t* d_alive, int max_shift,
                        int pb, vsv_sig* out, const StageBufs& b, uint64_t* long_list, Counters* ctr) {
  (void)long_list; (void)ctr;
  cluster_kernel<<<b.grid, 256, 0, st>>>(sorted, sorted_key, d_alive, max_shift, pb, b.cl, out);
}

constexpr int PAIR_ROUNDS = 10;
void vsv_launch_pair(hipStream_t st, const vsv_sig* merged, const uint64_t* merged_key, const uint32_t* d_alive3, int pair_shift, int pair_window,
                     vsv_call* calls_tmp, vsv_call* calls, uint32_t* d_ncalls, const StageBufs& b, uint64_t* key2, uint32_t* idx2,
                     const SortWork& sw, int pb, int nbits, int64_t cap, Counters* ctr, bool dense) {
  // (the stage-3 gather set the pairing state b.cl to -1)
  const int right = pair_shift < pair_window ? pair_shift : pair_window;
  if (dense) {
    // the previous run of this handle walked a stretch of thousands of rows with one wave: pair in rounds instead (scratch: the
    // call-key arrays, free until pair_finish, and the stage sort's index array)
    uint32_t* jlo = idx2; uint32_t* done1 = b.idx; uint64_t* res = key2;
    pair_rounds_prep<<<b.grid, 256, 0, st>>>(merged, merged_key, d_alive3, pair_shift, pb, jlo, done1, res);
    for (uint32_t r = 0; r < (uint32_t)PAIR_ROUNDS; ++r) {
      pair_round<false><<<b.grid, 256, 0, st>>>(merged, merged_key, d_alive3, pair_shift, right, pb, b.cl, calls_tmp, jlo, done1, res, r);
      pair_round<true><<<b.grid, 256, 0, st>>>(merged, merged_key, d_alive3, pair_shift, right, pb, b.cl, calls_tmp, jlo, done1, res, r);
    }
    pair_leftover<<<b.grid, 256, 0, st>>>(merged, merged_key, d_alive3, pair_shift, right, pb, b.cl, calls_tmp, jlo, done1);
  } else {
    pair_kernel<<<b.grid, 256, 0, st>>>(merged, merged_key, d_alive3, pair_shift, right, pb, b.cl, calls_tmp, &ctr->max_stretch);
  }
  // bucket sort: the hp2 rows' calls and every key are derived inside the sort's kernels (CallKey / RowIO<vsv_call>::fetch)
  if (vsv_bucket_sort_calls(st, calls_tmp, d_alive3, pb, b.tid_lo, nbits, b.kmax, calls, key2, d_ncalls, &ctr->n_long, sw, cap, merged, b.cl)) return;
  pair_finish<<<b.grid, 256, 0, st>>>(merged, d_alive3, b.cl, calls_tmp, pb, b.tid_lo, key2, idx2);
  const SortResult r = vsv_radix_sort_pairs(st, key2, idx2, sw.key_alt, sw.val_alt, d_alive3, cap, nbits, sw, b.kmax);
  gather_rows<vsv_call><<<b.grid, 256, 0, st>>>(calls_tmp, r.val, r.key, d_alive3, calls, d_ncalls, &ctr->n_long, nullptr);
}
