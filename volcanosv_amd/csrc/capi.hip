// capi.hip — extern "C" entry points of include/volcanosv.h: handle, workspace, stage orchestration.
// All device work is enqueued on the handle's stream without host synchronisation; counts and error
// bits live in a device `Counters` block that vsv_finish() reads back once.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "vsv_device.h"
#include <chrono>

namespace {

// ops per K1 part (one wave each); VSV_OPS_PER_PART overrides it for timing experiments
static const int OPS_PER_PART = vsv_dbg_env("VSV_OPS_PER_PART") && atoi(vsv_dbg_env("VSV_OPS_PER_PART")) >= 1024 ? atoi(vsv_dbg_env("VSV_OPS_PER_PART")) : 4096;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

}  // namespace

struct vsv_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  int64_t last_count = 0;
  // capacities
  int64_t cap_sigs = 0, cap_records = 0, cap_ops = 0;
  // record upload buffers (host-pointer callers)
  DevBuf r_pos, r_tid, r_qid, r_off, r_mapq, r_flag, r_cigar;
  DevBuf g_off, g_qs, g_qe, g_rid, g_rs, g_re, g_rev, g_hap, g_len, g_rank;   // segment uploads (BND branch)
  vsv_segments segs{};
  int bnd_stage = 0;
  int64_t cutesv_rows = -1;   // rows of the last vsv_cutesv_split (c2 buffer)
  // workspace
  DevBuf part_rb, part_count, part_off, scan_tmp;
  DevBuf l_agg, l_carry_r, l_carry_q, l_tiles;   // long-record scan: per-part aggregates, carries, tile sums
  uint32_t cand_epoch = 0;                       // ... of the candidate kernels' scans
  DevBuf lbw; uint32_t lb_epoch = 0;             // look-back words of the placement's scan (valid by epoch: zeroed when allocated, never per run)
  DevBuf l_prec;                                 // long-record scan: where every part's last descriptor batch lies
  DevBuf z_crctab, z_crc;  // CRC-32 tables (uploaded once) and per-member results of the device inflate
  const uint32_t* expect_crc = nullptr;   // vsv_bgzf_set_expected_crc: trailer CRCs of the members of the next inflate / parse
  int64_t expect_crc_n = 0;
  DevBuf cmask;           // split stage: item masks of the candidate count pass (1 byte per 4 records)
  DevBuf gflag;           // CLR: flag bytes with the haplotype bits cleared where the gate fails (input of the scan)
  DevBuf pool, pool_key, raw0, s1in, s1s, c1, s2s, c2, merged, calls_tmp, calls, reads;
  DevBuf tab, tab2, blk_cnt, blk_off, ckey, crec, okey, oval;   // tab / tab2: the name table ("occurs more than once") of this run and of the next
  DevBuf qlbw, clbw;               // look-back words of the two scans inside the candidate kernels (one per 2048 records each)
  uint32_t tab_dirty[2] = {0, 0};  // words a run may have left set in each (cleared by the run in front of the one that uses it)
  DevBuf cinfo, cord, oc1;             // large read-shaped inputs: per-candidate record info, candidate ordinals by name, a slot's first candidate
  DevBuf key, idx, cl, key2, idx2, key_alt, val_alt, hist;
  DevBuf ctr, shard_cnt, totals;   // views into `arena`
  DevBuf arena;                    // Counters | emit-pool cursors | part-count tile sums | sort digit totals: zeroed by ONE fill per run
  uint32_t* tile_cnt = nullptr;
  DevBuf j_cpos, j_clen, j_spos, j_slen, j_send, j_out, j_err;
  DevBuf cs_tra; int64_t cs_reads = -1;   // vsv_cutesv_split: per-read translocation flags of the last call
  DevBuf z_comp, z_coff, z_ooff, z_out, z_stat;          // BGZF inflate
  DevBuf p_spec, p_cnt, p_land, p_base, p_recoff, p_pos, p_tid, p_mapq, p_flag, p_lseq, p_sflag, p_ncig, p_cgsrc, p_hash, p_keep, p_kidx,
      p_cigoff, p_sums, p_tot, p_err;                     // device BAM parse: per input record
  DevBuf o_pos, o_tid, o_qid, o_cigoff, o_mapq, o_flag, o_cigar, o_lseq, o_sflag, o_hash, o_recoff, o_first, o_rank, o_nlen, o_noff, o_blob, o_n, o_names, o_nmoff, o_nmlen;
  std::string sa_text;                 // SA tag texts of the last device parse (host copy)
  char* names_pin = nullptr; size_t names_cap = 0, names_len = 0;   // name table of the last device parse: page-locked host copy (a
                                       // std::string would be zero-filled by resize and filled through the runtime's staging: 46 MB = 5 ms)
  bool want_sa = false;                // vsv_bam_set_want_sa: the device parse also collects the SA:Z texts
  DevBuf o_saoff, o_salen, o_saloc, o_sa;
  bool want_seq = false;               // vsv_bam_device_want_seq: the device parse also keeps the packed SEQ fields (device-resident)
  DevBuf o_sqoff, o_sqlen, o_sqloc, o_seq, o_recseq, q_rec, q_start, q_len, q_rev, q_ooff, q_out;
  int64_t seq_records = -1;            // records whose SEQ the store holds (-1: none)
  void* pin_buf = nullptr; size_t pin_bytes = 0;   // page-locked staging of a BAM file's compressed bytes (vsv_internal_pinned)
  int pass_cursor = 0;
  int group_cursor = 0;
  uint32_t* groups = nullptr;      // view into `arena`
  const uint64_t* sorted_key = nullptr;   // sorted keys of the stage just sorted (cluster / pair read them)
  bool small_sort_tiles = true;    // radix tile size, re-decided after every run from its row counts
  int lsd_runs = 0;                // > 0: the bucket sort overflowed recently, the next runs use the LSD passes
  uint64_t sort_hint_rows = 0;     // the row count the bucket sort of the run in flight was sized for
  bool clr_unfused = false;        // the fused CLR scan met a part too long for its gate state: separate gate pass from now on
  bool dense_pairing = false;      // the last run walked a pairing stretch of thousands of rows with one wave: pair in rounds
  bool in_rerun = false;
  int row_runs = 0;                // > 0: an input met the 32-bit limits of the element path recently: the next runs stay on rows
  int lsd_sort1_runs = 0;          // > 0: a bucket of the element path's first sort did not fit in LDS recently (ERRB_BUCKET1_SLOW): the next runs sort 1 with the LSD passes
  int64_t sort1_slow_runs = 0;     // runs whose first sort met a bucket too large for LDS (vsv_path_counts)
  int lsd_slim_runs = 0;           // > 0: a rank-and-merge sort of the element path gave up recently (ERRB_MERGE_FALLBACK): the next runs sort with the LSD passes
  const void* sl_sorted1 = nullptr; // element path: the sorted stage-1 table (anchors of the sort behind its clusters)
  void* sl_ctl2 = nullptr;          // ... and that sort's control slot (its tile counts come from the stage-1 cluster kernel)
  int sl_shift1 = 0;                // ... and the cluster_shift those clusters were cut with (the staged entry points may change it in between)
  int64_t reruns = 0;             // whole-run repetitions taken by finish() so far (vsv_rerun_count)
  // large tables (slim_path.hip): element buffers, pairing scratch, what the run in flight left where
  DevBuf sl[6], sl_hj, sl_done;
  bool big_run = false;            // the stages of this run work on 16-byte elements; rows are gathered on request
  bool sl_mm_ready = false;        // ... and the in-place fold has left their position range behind the counters (fold_mm)
  bool sl_prebuilt = false;        // ... and fold / split_eval have written the stage-1 elements next to their rows
  void* sl_e2 = nullptr;           // stage-1 cluster output (n_alive1 slots)
  void* sl_m = nullptr;            // merged elements (n_alive3)
  bool c1_stale = false, merged_stale = false;      // VSV_T_CLUSTER1 / VSV_T_MERGED rows not gathered yet
  int raw_state = 0;               // VSV_T_RAW: 0 = in raw0; 1 = the scan placed its rows straight into the stage-1 table, raw0 is filled from the
                                   // scan's descriptors when somebody asks; 2 = those are gone (buffers re-allocated since)
  int raw_parts = 0;               // ... parts of that scan
  bool have_history = false;       // a signature run of this handle has finished: host_ctr describes real tables (else: a COLD handle)
  bool skip_batches = false;       // the handle's last run filed no overflow batch (parts with more rows than their stage): the next one has no launch for them
  bool batches_skipped = false;    // ... and the run in flight went without
  int batch_runs = 0;              // > 0: such a run met one and was repeated: the next runs keep the launch
  bool cold_run = false;           // the run in flight is such a handle's first one: its row count comes from a wait for its own scan
  bool ctr_of_run = false;         // host_ctr holds the counters of the run in progress (a staged call's finish()), not of the previous one
  int64_t element_runs = 0, cold_syncs = 0;          // vsv_path_counts
  vsv_bnd_params bnd_prm{};
  Counters host_ctr;
  Counters* pinned = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool have_scan_ev = false;
  // the fused run builds the split candidates (record arrays only) on `aux` while the scan streams the CIGARs on `stream`
  hipStream_t aux = nullptr, aux2 = nullptr;   // aux2: second lane of the sliced BGZF inflate
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool fork_split = false;         // this run is the fused one: enq_scan may start the candidates early
  bool split_cands_done = false;   // the candidates of this run are enqueued (enq_split only evaluates the pairs)
  bool aux_pending = false;        // work on `aux` that `stream` has not been made to wait for yet
  SplitSorted split_sorted{nullptr, nullptr, nullptr, nullptr};
  DevBuf scan_tmp2;                // block sums of the candidate-count scan (scan_tmp belongs to the CIGAR scan)
  // state
  RecView rv{};
  int n_tids = 0;
  int64_t max_pos = 0;
  vsv_params prm{};
  int stage_done = 0;     // 0 none, 1 scan, 2 split, 3 sort_cluster, 4 merge, 5 pair
  bool pending = false;   // work enqueued, counters not read back
  uint32_t tab_size = 0;
};

namespace {

int fail(vsv_handle* h, int status, const std::string& msg) {
  if (h) h->err = msg;
  return status;
}

#define HIPCHK(h, expr)                                                                      \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess) return fail(h, VSV_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

int ensure(vsv_handle* h, DevBuf& b, size_t bytes) {
  if (bytes < 256) bytes = 256;
  if (b.bytes >= bytes) return 0;
  if (b.p) { HIPCHK(h, hipFree(b.p)); b.p = nullptr; b.bytes = 0; }
  HIPCHK(h, hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  return 0;
}

uint32_t pow2_at_least(uint64_t n) { uint32_t p = 1024; while (p < n && p < (1u << 30)) p <<= 1; return p; }
int bits_for(uint64_t n) { int b = 1; while ((1ull << b) < n && b < 63) ++b; return b; }

int reserve(vsv_handle* h, int64_t max_records, int64_t max_ops, int64_t max_sigs) {
  if (max_sigs < 1024) max_sigs = 1024;
  if ((max_sigs > h->cap_sigs || max_ops > h->cap_ops || max_records > h->cap_records) && h->raw_state == 1) h->raw_state = 2;   // the descriptors move
  if (max_records < 1) max_records = 1;
  if (max_ops < 1) max_ops = 1;
  if (max_sigs > h->cap_sigs) {
    const size_t n = (size_t)max_sigs;
    h->stage_done = 0; h->bnd_stage = 0; h->cutesv_rows = -1;      // the row buffers move: the tables of the last run are gone
    h->c1_stale = h->merged_stale = false; h->sl_e2 = nullptr; h->sl_m = nullptr;
    DevBuf* sigbufs[] = {&h->pool, &h->raw0, &h->s1in, &h->s1s, &h->c1, &h->s2s, &h->c2, &h->merged, &h->reads};
    for (DevBuf* b : sigbufs) { int st = ensure(h, *b, n * sizeof(vsv_sig)); if (st) return st; }
    DevBuf* callbufs[] = {&h->calls_tmp, &h->calls};
    for (DevBuf* b : callbufs) { int st = ensure(h, *b, n * sizeof(vsv_call)); if (st) return st; }
    DevBuf* u64bufs[] = {&h->pool_key, &h->ckey, &h->okey, &h->key, &h->key2, &h->key_alt};
    for (DevBuf* b : u64bufs) { int st = ensure(h, *b, n * sizeof(uint64_t)); if (st) return st; }
    DevBuf* u32bufs[] = {&h->crec, &h->oval, &h->idx, &h->cl, &h->idx2, &h->val_alt};
    for (DevBuf* b : u32bufs) { int st = ensure(h, *b, n * sizeof(uint32_t)); if (st) return st; }
    int st = ensure(h, h->hist, (size_t)vsv_radix_hist_entries(max_sigs) * sizeof(uint32_t));
    if (st) return st;
    if ((st = ensure(h, h->shard_cnt, 256 * 16 * sizeof(uint32_t)))) return st;
    if ((st = ensure(h, h->totals, (size_t)64 * 2048 * sizeof(uint32_t)))) return st;
    h->cap_sigs = max_sigs;
  }
  if (max_ops > h->cap_ops || max_records > h->cap_records) {
    const int64_t ops = max_ops > h->cap_ops ? max_ops : h->cap_ops;
    const int64_t recs = max_records > h->cap_records ? max_records : h->cap_records;
    const size_t n_parts = (size_t)vsv_cigar_parts(ops, OPS_PER_PART) + 16;
    const size_t nblk = (size_t)((recs * 2 + 2047) / 2048) + 2;
    int st;
    if ((st = ensure(h, h->part_rb, (n_parts + 1) * 4))) return st;
    if ((st = ensure(h, h->part_count, n_parts * 4))) return st;
    if ((st = ensure(h, h->part_off, n_parts * 4))) return st;
    if ((st = ensure(h, h->l_agg, vsv_long_scan_bytes(ops, 0, OPS_PER_PART)))) return st;
    if ((st = ensure(h, h->l_carry_r, vsv_long_scan_bytes(ops, 1, OPS_PER_PART)))) return st;
    if ((st = ensure(h, h->l_carry_q, vsv_long_scan_bytes(ops, 1, OPS_PER_PART)))) return st;
    if ((st = ensure(h, h->l_tiles, vsv_long_scan_bytes(ops, 2, OPS_PER_PART)))) return st;
    if ((st = ensure(h, h->l_prec, vsv_long_scan_bytes(ops, 3, OPS_PER_PART)))) return st;
    {
      const void* before = h->lbw.p;
      if ((st = ensure(h, h->lbw, vsv_lookback_bytes(ops, OPS_PER_PART)))) return st;
      if (h->lbw.p != before) { HIPCHK(h, hipMemsetAsync(h->lbw.p, 0, h->lbw.bytes, h->stream)); h->lb_epoch = 0; }
    }
    if ((st = ensure(h, h->blk_cnt, nblk * 4))) return st;
    if ((st = ensure(h, h->blk_off, nblk * 4))) return st;
    if ((st = ensure(h, h->cmask, nblk * 512 + 1024))) return st;      // SC_ROUNDS (2) x 256 bytes per block of 2048 records
    const size_t m = n_parts > nblk ? n_parts : nblk;
    if ((st = ensure(h, h->scan_tmp, (m / 2048 + 2) * 4))) return st;
    if ((st = ensure(h, h->scan_tmp2, (nblk / 2048 + 2) * 4))) return st;
    h->cap_ops = ops;
    h->cap_records = recs;
  }
  return 0;
}

int upload(vsv_handle* h, DevBuf& b, const void* src, size_t bytes) {
  int st = ensure(h, b, bytes + 16);
  if (st) return st;
  if (bytes) HIPCHK(h, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, h->stream));
  return 0;
}

int bind_records(vsv_handle* h, const vsv_records* r) {
  if (!r) return fail(h, VSV_E_INVALID, "records is NULL");
  if (r->n_records < 0 || r->n_ops < 0 || r->n_records > 0xFFFFFFF0ll) return fail(h, VSV_E_INVALID, "bad record counts");
  if (r->n_records > 0 && (!r->pos || !r->tid || !r->qid || !r->cigar_off || !r->mapq || !r->flag || (r->n_ops > 0 && !r->cigar)))
    return fail(h, VSV_E_INVALID, "NULL record array");
  RecView v{};
  v.n_records = r->n_records; v.n_ops = r->n_ops; v.n_qids = r->n_qids;
  if (r->on_device) {
    if (((uintptr_t)r->cigar & 15) != 0) return fail(h, VSV_E_INVALID, "cigar must be 16-byte aligned");
    v.pos = r->pos; v.tid = r->tid; v.qid = r->qid; v.cigar_off = r->cigar_off; v.mapq = r->mapq; v.flag = r->flag; v.cigar = r->cigar;
  } else {
    const size_t n = (size_t)r->n_records;
    int st;
    if ((st = upload(h, h->r_pos, r->pos, n * 4))) return st;
    if ((st = upload(h, h->r_tid, r->tid, n * 4))) return st;
    if ((st = upload(h, h->r_qid, r->qid, n * 4))) return st;
    if ((st = upload(h, h->r_off, r->cigar_off, (n + 1) * 8))) return st;
    if ((st = upload(h, h->r_mapq, r->mapq, n))) return st;
    if ((st = upload(h, h->r_flag, r->flag, n))) return st;
    if ((st = upload(h, h->r_cigar, r->cigar, (size_t)r->n_ops * 4))) return st;
    v.pos = (const int32_t*)h->r_pos.p; v.tid = (const int32_t*)h->r_tid.p; v.qid = (const uint32_t*)h->r_qid.p;
    v.cigar_off = (const uint64_t*)h->r_off.p; v.mapq = (const uint8_t*)h->r_mapq.p; v.flag = (const uint8_t*)h->r_flag.p;
    v.cigar = (const uint32_t*)h->r_cigar.p;
  }
  if (r->tid_lo < 0 || (r->n_tids > 0 && r->tid_lo >= r->n_tids)) return fail(h, VSV_E_INVALID, "tid_lo outside [0, n_tids)");
  v.tid_lo = r->tid_lo;
  h->rv = v;
  h->n_tids = r->n_tids;
  h->max_pos = r->max_pos > 0 ? r->max_pos : 0;
  int st = reserve(h, r->n_records, r->n_ops, h->cap_sigs > 0 ? h->cap_sigs : (1 << 22));
  if (st) return st;
  // 1 bit per query name ("name occurs more than once"); n_qids = max qid + 1 is part of the contract
  if (r->n_qids <= 0 && r->n_records > 0) return fail(h, VSV_E_INVALID, "n_qids (max qid + 1) is required");
  const uint32_t words = (uint32_t)(((uint64_t)r->n_qids + 31) / 32 + 2);
  for (DevBuf* t : {&h->tab, &h->tab2}) {
    const void* before = t->p;
    if ((st = ensure(h, *t, (size_t)words * 4))) return st;
    if (t->p != before) {            // a new table: clean, once (afterwards every run clears the table of the run behind it)
      HIPCHK(h, hipMemsetAsync(t->p, 0, t->bytes, h->stream));
      h->tab_dirty[t == &h->tab ? 0 : 1] = 0;
    }
  }
  h->tab_size = words;
  {
    const size_t nw = (size_t)(r->n_records / 2048 + 4) * sizeof(uint64_t);
    for (DevBuf* t : {&h->qlbw, &h->clbw}) {
      const void* before = t->p;
      if ((st = ensure(h, *t, nw))) return st;
      if (t->p != before) HIPCHK(h, hipMemsetAsync(t->p, 0, t->bytes, h->stream));
    }
  }
  return 0;
}

constexpr int MAX_SORT_PASSES = 64;
constexpr int MAX_GROUP_SLOTS = 8;      // bucket sorts of one run that may skip their scan launch (a contig run has 6-7)
// table_rows >= 0: a sort of a table of about that many rows, whatever the handle's largest table is (the few thousand split candidates
// of a contig pile whose signature tables hold millions)
SortWork sort_work(vsv_handle* h, int64_t table_rows = -1) {
  SortWork w;
  w.key_alt = (uint64_t*)h->key_alt.p; w.val_alt = (uint32_t*)h->val_alt.p; w.hist = (uint32_t*)h->hist.p; w.max_items = h->cap_sigs;
  w.totals = (uint32_t*)h->totals.p; w.pass_cursor = &h->pass_cursor; w.max_passes = MAX_SORT_PASSES; w.small_tiles = h->small_sort_tiles;
  static const char* force = vsv_dbg_env("VSV_SORT_TILE");   // timing experiments: "big" / "small"
  if (force) w.small_tiles = force[0] == 's';
  // bucket sort: the fewest buckets (256..2048) that keep the largest table of the previous run at <= 640 rows per bucket, i.e.
  // ~1300 per occupied one with half of the key range populated (a workgroup sorts up to 4096); tables beyond ~1.3 M rows, a
  // handle whose last bucket sort overflowed, and VSV_SORT=lsd take the LSD passes
  static const char* mode = vsv_dbg_env("VSV_SORT");
  const Counters& c = h->host_ctr;
  const uint64_t rows = table_rows >= 0 ? (uint64_t)table_rows : (c.n_s1 > c.n_cand ? c.n_s1 : c.n_cand);
  if (table_rows >= 0 && !force) w.small_tiles = rows <= 128u * 4096u;
  static const int per_bucket = vsv_dbg_env("VSV_BK_ROWS") ? atoi(vsv_dbg_env("VSV_BK_ROWS")) : 640;     // timing experiments
  int bb = 8;
  while (bb < 11 && (rows >> bb) > (uint64_t)per_bucket) ++bb;
  // (a handle's first run knows no row count: LSD passes rather than a guess that overflows and repeats the run)
  w.bucket_bits = (h->lsd_runs > 0 || rows == 0 || (mode && mode[0] == 'l') || (rows >> 11) > 640) ? 0 : bb;
  if (table_rows < 0) h->sort_hint_rows = rows;
  w.hint_rows = rows;
  w.shared_gpu = h->prm.split_overlap == VSV_OVERLAP_OFF;
  static const int bk_threads = vsv_dbg_env("VSV_BK_THREADS") ? atoi(vsv_dbg_env("VSV_BK_THREADS")) : 0;     // timing experiments: 256 | 512
  if (bk_threads) w.shared_gpu = bk_threads == 256;
  w.groups = h->groups; w.group_cursor = &h->group_cursor; w.max_group_slots = MAX_GROUP_SLOTS;
  w.err = h->ctr.p ? &((Counters*)h->ctr.p)->err : nullptr;
  w.slots_ok = true;      // (the signature stages read nothing but the key of a dead row)
  return w;
}
// ... for callers that gather by the values of dead rows too (breakend candidates, name hashes): the three-launch forms
SortWork sort_work_plain(vsv_handle* h) { SortWork w = sort_work(h); w.slots_ok = false; return w; }
// zero the device counters and the per-pass sort totals: start of every run
constexpr size_t ARENA_CTR = 1024, ARENA_SHARD = 256 * 16 * sizeof(uint32_t), ARENA_TILES = 4096 * sizeof(uint32_t),
                 ARENA_TOTALS = (size_t)MAX_SORT_PASSES * 2048 * sizeof(uint32_t),
                 ARENA_GROUPS = (size_t)MAX_GROUP_SLOTS * VSV_RS_MAX_GROUPS * 2048 * sizeof(uint32_t),     // 4 MB: ~1 us more of the fill
                 ARENA_BYTES = ARENA_CTR + ARENA_SHARD + ARENA_TILES + ARENA_TOTALS + ARENA_GROUPS;
// the handle's auxiliary stream(s), created when first needed; false = not available, everything stays on `stream`
bool have_aux(vsv_handle* h, bool both) {
  if (!h->aux && hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking) != hipSuccess) { h->aux = nullptr; (void)hipGetLastError(); }
  if (both && !h->aux2 && hipStreamCreateWithFlags(&h->aux2, hipStreamNonBlocking) != hipSuccess) { h->aux2 = nullptr; (void)hipGetLastError(); }
  return h->aux && (!both || h->aux2);
}
int join_aux(vsv_handle* h) {
  if (h->aux_pending) { HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join, 0)); h->aux_pending = false; }
  return 0;
}
int reset_run_state(vsv_handle* h) {
  { int js = join_aux(h); if (js) return js; }    // an abandoned run's candidates may still be counting into the arena
  h->split_cands_done = false;
  HIPCHK(h, hipMemsetAsync(h->arena.p, 0, ARENA_BYTES, h->stream));
  h->pass_cursor = 0;
  h->group_cursor = 0;
  return 0;
}
// a staged stage called AGAIN on the same run (e.g. another cluster_shift): the sort scratch of the run starts over — per-pass digit
// totals, tile counts, group sums; the counters and the scan's cursors (VSV_T_RAW may still be placed from them) stay
int reenter_reset(vsv_handle* h) {
  HIPCHK(h, hipMemsetAsync((char*)h->arena.p + ARENA_CTR + ARENA_SHARD, 0, ARENA_BYTES - ARENA_CTR - ARENA_SHARD, h->stream));
  h->pass_cursor = 0;
  h->group_cursor = 0;
  return 0;
}
int tid_bits(vsv_handle* h);
LongScanBufs long_bufs(vsv_handle* h, bool clr_fused, void* fused_rows, const SlimOut& so, bool skip_batches = false) {
  return LongScanBufs{h->l_agg.p, (uint32_t*)h->l_carry_r.p, (uint32_t*)h->l_carry_q.p, h->l_tiles.p, h->tile_cnt, 4096, true, clr_fused,
                      (uint64_t*)h->lbw.p, h->lb_epoch, h->l_prec.p, fused_rows, so, skip_batches};
}
bool want_big(vsv_handle* h);
bool big_allowed(vsv_handle* h);
int slim_work(vsv_handle* h, SlimWork& w);
int slim_alloc(vsv_handle* h, bool all);
int sort_bits(vsv_handle* h);
// blocks of the row-parallel kernels: one row per thread for the largest table of the handle's previous run (+25 %), between 128
// and 4096 blocks; a first run sizes for the row capacity. Every such kernel grid-strides, so this is speed only.
int ew_grid(vsv_handle* h) {
  const Counters& c = h->host_ctr;
  uint64_t rows = c.n_s1 > c.n_cand ? c.n_s1 : c.n_cand;
  rows = rows ? rows + rows / 4 : (uint64_t)h->cap_sigs;
  const uint64_t g = (rows + 255) / 256;
  return (int)(g < 128 ? 128 : g > 4096 ? 4096 : g);
}
int pos_bits(vsv_handle* h);
StageBufs stage_bufs(vsv_handle* h) {
  const uint64_t kmax = h->n_tids > h->rv.tid_lo ? (uint64_t)(h->n_tids - h->rv.tid_lo) << (pos_bits(h) + 3) : 0;
  return StageBufs{(uint64_t*)h->key.p, (uint32_t*)h->idx.p, (int32_t*)h->cl.p, h->rv.tid_lo, tid_bits(h), kmax, ew_grid(h)};
}
Counters* dctr(vsv_handle* h) { return (Counters*)h->ctr.p; }
// 2 x 64 words behind the counters (zeroed with them): the position range of the stage-1 cigar elements, left by the in-place fold
uint32_t* fold_mm(vsv_handle* h) { static_assert(sizeof(Counters) <= 256, "counters"); return (uint32_t*)h->ctr.p + 64; }
int pos_bits(vsv_handle* h) { return h->max_pos > 0 ? bits_for((uint64_t)h->max_pos + VSV_POS_BIAS + 2) : 32; }
int tid_bits(vsv_handle* h) { return bits_for((uint64_t)(h->n_tids > 0 ? h->n_tids - h->rv.tid_lo : 65536) + 1); }
int key_bits(vsv_handle* h) { return pos_bits(h) + 3 + tid_bits(h) + 1; }   // +1: dead keys (all ones) sort last

bool is_contig(int dtype) { return dtype == VSV_DTYPE_HIFI || dtype == VSV_DTYPE_ONT || dtype == VSV_DTYPE_CLR; }

// ---- stage enqueue functions ------------------------------------------------------------------------
int cand_alloc(vsv_handle* h) {
  const size_t n = (size_t)h->cap_sigs;
  int ws;
  if ((ws = ensure(h, h->cinfo, n * 32 + 64)) || (ws = ensure(h, h->cord, n * 4 + 64)) || (ws = ensure(h, h->oc1, n * 4 + 64))) return ws;
  return 0;
}
int enq_split_candidates(vsv_handle* h, hipStream_t st, int phase = 0) {
  SlimWork w;
  // a large-table run: the candidate tables are usually large too (config 3: 10^7) — unless the handle knows better: the contig pile
  // has a few thousand split candidates, which the bucket sort of small tables orders in 6 launches where the element passes take 20
  const uint64_t ncand = h->host_ctr.n_cand;
  const bool few = h->big_run && ncand != 0 && ((ncand + ncand / 4) >> 11) <= 640;
  const bool slim = h->big_run && is_contig(h->prm.dtype) && !few;
  CandBufs cb{nullptr, nullptr, nullptr};
  if (slim) {
    int ws = slim_work(h, w); if (ws) return ws;
    if (!vsv_scan_is_long(h->rv, h->prm)) {           // per-candidate record info for split_eval_info (sig_stages.hip)
      if ((ws = cand_alloc(h))) return ws;            // (allocated in front of the run's first launch: this finds them in place)
      cb = CandBufs{h->cinfo.p, (uint32_t*)h->cord.p, (uint32_t*)h->oc1.p};
    }
  }
  // Shards of more than 8192 tiles of 2048 records take the one-launch candidate kernels (scans by look-back, words valid by epoch): they
  // mark in one name table, which must be clean, and clear the other for the run behind them. Smaller shards take the two-launch
  // kernels, which clear table 0 themselves and leave their marks there.
  const bool cand_lb = h->rv.n_records > (int64_t)8192 * 2048;
  int cur = 0;
  CandLb clb{nullptr, nullptr, 0, nullptr, 0};
  if (cand_lb) {
    cur = h->tab_dirty[0] == 0 ? 0 : 1;
    if (phase != 2) {
      if (h->tab_dirty[cur] != 0) { HIPCHK(h, hipMemsetAsync(h->tab2.p, 0, h->tab2.bytes, st)); h->tab_dirty[1] = 0; }     // (both hold marks: a switch of forms)
      if (++h->cand_epoch >= (1u << 24)) {
        HIPCHK(h, hipMemsetAsync(h->qlbw.p, 0, h->qlbw.bytes, st));
        HIPCHK(h, hipMemsetAsync(h->clbw.p, 0, h->clbw.bytes, st));
        h->cand_epoch = 1;
      }
    }
    clb = CandLb{(uint64_t*)h->qlbw.p, (uint64_t*)h->clbw.p, h->cand_epoch, (uint32_t*)(cur ? h->tab.p : h->tab2.p), h->tab_dirty[1 - cur]};
    if (phase != 2) { h->tab_dirty[1 - cur] = 0; h->tab_dirty[cur] = h->tab_size; }
  } else if (phase != 2) h->tab_dirty[0] = h->tab_size;
  DevBuf& tcur = cur ? h->tab2 : h->tab;
  h->split_sorted = vsv_launch_split_candidates(st, h->rv, h->prm, h->n_tids, (uint32_t*)tcur.p, h->tab_size, (uint32_t*)h->blk_cnt.p,
                                                (uint32_t*)h->blk_off.p, (uint32_t*)h->scan_tmp2.p, (uint64_t*)h->ckey.p, (uint32_t*)h->crec.p,
                                                (uint64_t*)h->okey.p, (uint32_t*)h->oval.p, (uint64_t*)h->key2.p, (uint32_t*)h->idx2.p,
                                                few ? sort_work(h, (int64_t)(ncand + ncand / 4)) : sort_work(h),
                                                (uint32_t)h->cap_sigs, dctr(h), (uint8_t*)h->cmask.p, ew_grid(h), slim ? &w : nullptr, cb, phase, clb);
  HIPCHK(h, hipGetLastError());
  h->split_cands_done = phase != 1;
  return 0;
}
int enq_scan(vsv_handle* h) {
  hipStream_t st = h->stream;
  { int rs = reset_run_state(h); if (rs) return rs; }
  if (h->lsd_runs > 0 && !h->in_rerun) --h->lsd_runs;
  if (h->row_runs > 0 && !h->in_rerun) --h->row_runs;
  if (h->lsd_slim_runs > 0 && !h->in_rerun) --h->lsd_slim_runs;
  if (h->lsd_sort1_runs > 0 && !h->in_rerun) --h->lsd_sort1_runs;
  if (h->batch_runs > 0 && !h->in_rerun) --h->batch_runs;
  h->cutesv_rows = -1;   // the split-candidate table shares a buffer with the merge stage
  h->ctr_of_run = false;
  if (is_contig(h->prm.dtype)) {   // the element buffers this run may need exist before its first launch (a large-table handle: all of them)
    const bool all = want_big(h) || !h->have_history;
    int as = slim_alloc(h, all);
    if (as) return as;
    if (all && !vsv_scan_is_long(h->rv, h->prm) && (as = cand_alloc(h))) return as;
  }
  const int n_parts = vsv_cigar_parts(h->rv.n_ops, OPS_PER_PART);
  RecView srv = h->rv;
  // CLR: the read-shaped scan carries the gate itself (cigar_scan_emit<0, 4, true>); the long-record scan, and a handle whose fused
  // scan met a part too long for its gate state, see haplotype tags only where a separate pass over the CIGARs lets the gate pass
  static const char* clr_mode = vsv_dbg_env("VSV_CLR");          // tests: "separate"
  const bool clr_fused = h->prm.dtype == VSV_DTYPE_CLR && srv.n_records > 0 && !vsv_scan_is_long(srv, h->prm) && !h->clr_unfused &&
                         !(clr_mode && clr_mode[0] == 's');
  if (h->prm.dtype == VSV_DTYPE_CLR && srv.n_records > 0 && !clr_fused) {
    int gs = ensure(h, h->gflag, (size_t)srv.n_records + 16);
    if (gs) return gs;
    vsv_launch_clr_gate(st, h->rv, (uint8_t*)h->gflag.p, dctr(h));
    srv.flag = (const uint8_t*)h->gflag.p;
  }
  // fused run: the split candidates need the record arrays only — they go to the auxiliary stream, enqueued BEHIND the scan (the
  // host spends ~50 us enqueuing them: the scan is already streaming by then) but ordered after this run's reset only
  static const char* where = vsv_dbg_env("VSV_SPLIT_STREAM");   // timing experiments: "main" keeps everything on the handle's stream
  const bool early_cands = h->fork_split && h->prm.enable_split && h->prm.dtype != VSV_DTYPE_SVIM && h->prm.dtype != VSV_DTYPE_CUTESV && h->rv.n_records > 0;
  const bool fork = early_cands && h->prm.split_overlap != VSV_OVERLAP_OFF && !(where && where[0] == 'm') && have_aux(h, false);
  if (fork) HIPCHK(h, hipEventRecord(h->ev_fork, st));
  if (++h->lb_epoch >= (1u << 24)) {            // the epoch field of the look-back words is about to wrap: start over on zeroed words
    HIPCHK(h, hipMemsetAsync(h->lbw.p, 0, h->lbw.bytes, st));
    h->lb_epoch = 1;
  }
  // the contig path: the placement writes the rows straight into the stage-1 input table, next to their elements, and the fold runs
  // there in place (VSV_T_RAW is re-placed from the scan's batches when somebody asks)
  const int long_parts = vsv_scan_parts(srv, h->prm, OPS_PER_PART);
  const bool fused = is_contig(h->prm.dtype) && long_parts > 0;
  SlimOut so_f{nullptr, 0, 0, 0, nullptr};
  // (the position range of the elements, for the first sort of a run on elements: a handle that knows its tables are small skips it)
  const bool want_mm = fused && pos_bits(h) < 32 && (!h->have_history || (want_big(h) && big_allowed(h)));
  if (fused) so_f = SlimOut{h->sl[0].p, pos_bits(h), h->rv.tid_lo, tid_bits(h), &dctr(h)->err, want_mm ? fold_mm(h) : nullptr};
  h->sl_mm_ready = want_mm;
  vsv_launch_cigar_scan(st, srv, h->prm, (uint32_t*)h->part_rb.p, n_parts, OPS_PER_PART, (vsv_sig*)h->pool.p,
                        (uint64_t*)h->pool_key.p, (uint32_t)h->cap_sigs, (uint32_t*)h->part_count.p, (uint32_t*)h->part_off.p,
                        (uint32_t*)h->scan_tmp.p, (vsv_sig*)h->raw0.p, dctr(h), (uint32_t*)h->shard_cnt.p, h->ev0, h->ev1,
                        long_bufs(h, clr_fused, fused ? h->s1in.p : nullptr, so_f, h->skip_batches));
  h->batches_skipped = h->skip_batches;
  h->have_scan_ev = n_parts > 0;
  h->raw_state = fused ? 1 : 0;
  h->raw_parts = long_parts;
  h->sl_prebuilt = false;
  if (fused) { vsv_launch_fold_elems(st, (vsv_sig*)h->s1in.p, dctr(h), ew_grid(h), so_f); h->sl_prebuilt = true; }
  // A COLD handle knows nothing about its tables — and every invocation of the drop-in CLI is one (one process per chromosome,
  // Raw_variant_call.py:65-73): the fused run waits for the scan once and takes the row count from it, so that the path, the sort
  // form and the grids of the stages behind it are those a warm handle would use.
  const bool cold = h->fork_split && !h->have_history && !h->in_rerun && is_contig(h->prm.dtype) && h->rv.n_records > 0;
  if (cold) {
    if (early_cands) {            // the candidates themselves (not their sorts: those depend on the path) beside the scan, as on a warm handle
      if (fork) HIPCHK(h, hipStreamWaitEvent(h->aux, h->ev_fork, 0));
      int cs = enq_split_candidates(h, fork ? h->aux : st, 1);
      if (cs) return cs;
      if (fork) HIPCHK(h, hipStreamSynchronize(h->aux));
    }
    HIPCHK(h, hipMemcpyAsync(h->pinned, h->ctr.p, sizeof(Counters), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    const Counters c = *h->pinned;
    if (!(c.err & ERRB_CAPACITY)) {
      h->host_ctr.n_raw = c.n_raw; h->host_ctr.n_s1 = c.n_raw; h->host_ctr.n_cand = early_cands ? c.n_cand : 0;
      const uint32_t big = c.n_raw > h->host_ctr.n_cand ? c.n_raw : h->host_ctr.n_cand;
      h->small_sort_tiles = big <= 128u * 4096u;
    }
    ++h->cold_syncs;
    h->cold_run = true;
  }
  // the path of the stages behind the split stage is decided here, so that the fold (and split_eval) can write the elements of a
  // large-table run next to their rows
  h->big_run = want_big(h) && big_allowed(h) && 4 * vsv_slim_sort_passes(sort_bits(h)) + 16 <= MAX_SORT_PASSES;
  if (!fused) {
    SlimOut so{nullptr, 0, 0, 0, nullptr};
    if (h->big_run) {
      so = SlimOut{h->sl[0].p, pos_bits(h), h->rv.tid_lo, tid_bits(h), &dctr(h)->err};
      h->sl_prebuilt = true;
    }
    vsv_launch_fold(st, (const vsv_sig*)h->raw0.p, (vsv_sig*)h->s1in.p, h->rv, h->prm, dctr(h), ew_grid(h), so);
  }
  HIPCHK(h, hipGetLastError());
  if (early_cands) {
    if (fork && !cold) HIPCHK(h, hipStreamWaitEvent(h->aux, h->ev_fork, 0));
    int cs = enq_split_candidates(h, fork ? h->aux : st, cold ? 2 : 0);
    if (cs) return cs;
    if (fork) { HIPCHK(h, hipEventRecord(h->ev_join, h->aux)); h->aux_pending = true; }
  }
  h->stage_done = 1;
  return 0;
}

int enq_split(vsv_handle* h) {
  hipStream_t st = h->stream;
  vsv_params p = h->prm;
  RecView rv = h->rv;
  if (!p.enable_split || p.dtype == VSV_DTYPE_SVIM) rv.n_records = 0;  // n_s1 = n_raw
  if (rv.n_records > 0 && !h->split_cands_done) { int cs = enq_split_candidates(h, st); if (cs) return cs; }
  { int js = join_aux(h); if (js) return js; }
  SlimOut so{nullptr, 0, 0, 0, nullptr};
  if (h->sl_prebuilt) so = SlimOut{h->sl[0].p, pos_bits(h), h->rv.tid_lo, tid_bits(h), &dctr(h)->err};     // (the cigar rows have their elements: the split rows too)
  vsv_launch_split_eval(st, rv, p, h->n_tids, rv.n_records > 0 ? h->split_sorted : SplitSorted{nullptr, nullptr, nullptr, nullptr},
                        (vsv_sig*)h->s1in.p, (uint32_t)h->cap_sigs, dctr(h), ew_grid(h), so);
  if (p.dtype == VSV_DTYPE_READS) {
    // reads.py:281-286 merge_all: one stable sort of [del_cigar, ins_cigar, del_split, ins_split] by pos
    const int nbits = pos_bits(h) + 2 + tid_bits(h) + 1;
    vsv_launch_sort_stage(st, (vsv_sig*)h->s1in.p, &dctr(h)->n_s1, 5, pos_bits(h), nbits, (vsv_sig*)h->reads.p, &dctr(h)->n_reads,
                          stage_bufs(h), sort_work(h), h->cap_sigs, dctr(h));
  }
  HIPCHK(h, hipGetLastError());
  h->stage_done = 2;
  return 0;
}

// Tables beyond what the bucket sort takes (~1.3 M rows: the contig pile of SURVEY row 2c, config 3's ONT tables) run their sort /
// cluster / merge / pair stages on 16-byte elements (slim_path.hip). Decided per run from the row counts of the handle's previous
// run; VSV_BIG=1 / 0 (under VSV_DEBUG) forces either path for the tests. Results are identical.
bool want_big(vsv_handle* h) {
  if (!is_contig(h->prm.dtype)) return false;
  static const char* force = vsv_dbg_env("VSV_BIG");
  if (force) return force[0] == '1';
  const Counters& c = h->host_ctr;
  const uint64_t rows = c.n_s1 > c.n_cand ? c.n_s1 : c.n_cand;
  return (rows >> 11) > 640;
}
// (the element kernels compute positions and lengths in 32 bits: an input that does not fit raises ERRB_SLIM_FALLBACK and the handle stays
// on rows for a while; element indices carry two flag bits)
bool big_allowed(vsv_handle* h) { return h->row_runs == 0 && h->cap_sigs < (1ll << 30); }
// the element buffers: allocated in front of a run's first launch (enq_scan), never in the middle of one
int slim_alloc(vsv_handle* h, bool all) {
  const size_t n = (size_t)h->cap_sigs;
  for (int k = 0; k < (all ? 6 : 1); ++k) {
    const void* before = h->sl[k].p;
    int st = ensure(h, h->sl[k], n * 16 + 64); if (st) return st;
    if (h->sl[k].p != before) { h->sl_e2 = nullptr; h->sl_m = nullptr; h->c1_stale = h->merged_stale = false; }     // (tables gathered from the old buffers are gone)
  }
  if (!all) return 0;
  int st;
  if ((st = ensure(h, h->sl_hj, n * 4 + 64))) return st;
  if ((st = ensure(h, h->sl_done, n * 4 + 64))) return st;
  return 0;
}
int slim_work(vsv_handle* h, SlimWork& w) {
  { int st = slim_alloc(h, true); if (st) return st; }
  for (int k = 0; k < 6; ++k) w.buf[k] = h->sl[k].p;
  { const Counters& c = h->host_ctr; w.rows_hint = c.n_s1 ? (int64_t)c.n_s1 + c.n_s1 / 4 : h->cap_sigs; w.cand_hint = c.n_cand ? (int64_t)c.n_cand + c.n_cand / 4 : h->cap_sigs; }
  w.cap = h->cap_sigs; w.hist = (uint32_t*)h->hist.p; w.totals = (uint32_t*)h->totals.p; w.pass_cursor = &h->pass_cursor;
  w.grid = ew_grid(h); w.cl = (int32_t*)h->cl.p; w.hj = (uint32_t*)h->sl_hj.p; w.done1 = (uint32_t*)h->sl_done.p;
  static const char* sort_mode = vsv_dbg_env("VSV_SLIM_SORT");      // tests / timing: "lsd" = the LSD passes for every sort of the element path
  w.merge_sorts = h->lsd_slim_runs == 0 && !(sort_mode && sort_mode[0] == 'l');
  w.bucket_sort1 = h->lsd_sort1_runs == 0 && !(sort_mode && sort_mode[0] == 'l');
  {  // share of split-list elements: pair slots of the previous run, or — a cold handle knows its candidates only — half of those (a slot takes two)
    const Counters& c = h->host_ctr;
    const double pairs = c.n_pairs ? (double)c.n_pairs : 0.5 * (double)c.n_cand, rows = (double)c.n_raw + pairs;
    w.split_share = rows > 0 ? pairs / rows : 0.05;
  }
  w.mm = h->sl_mm_ready && h->sl_prebuilt ? fold_mm(h) : nullptr;
  w.err = &dctr(h)->err;
  return 0;
}
int sort_bits(vsv_handle* h) { return pos_bits(h) + 3 + tid_bits(h); }
// (tid - tid_lo) takes bits_for(n_tids - tid_lo) bits: none for a single-chromosome shard (tid_bits() keeps one more for the range check)
int slim_tid_bits(vsv_handle* h) { return h->n_tids > h->rv.tid_lo ? bits_for((uint64_t)(h->n_tids - h->rv.tid_lo)) - (h->n_tids - h->rv.tid_lo == 1 ? 1 : 0) : tid_bits(h); }

int enq_stage1(vsv_handle* h) {
  hipStream_t st = h->stream;
  Counters* c = dctr(h);
  if (h->stage_done >= 3) { int rs = reenter_reset(h); if (rs) return rs; }
  // (decided in enq_scan; a staged call knows this run's own tables by now: vsv_finish() has read their counters)
  if (h->ctr_of_run && !h->in_rerun) h->big_run = want_big(h) && big_allowed(h);
  h->big_run = h->big_run && h->pass_cursor + 4 * vsv_slim_sort_passes(sort_bits(h)) <= MAX_SORT_PASSES;
  if (h->big_run) ++h->element_runs;
  h->c1_stale = h->merged_stale = false;
  if (h->big_run) {
    SlimWork w;
    { int ws = slim_work(h, w); if (ws) return ws; }
    h->sl_e2 = vsv_slim_stage1(st, (const vsv_sig*)h->s1in.p, &c->n_s1, &c->n_alive1, pos_bits(h), h->rv.tid_lo, slim_tid_bits(h), h->prm.cluster_shift, w, c, h->sl_prebuilt, &h->sl_sorted1, &h->sl_ctl2);
    h->sl_prebuilt = false;          // (the sort used the element buffer as scratch: a staged call that comes here again builds the elements from the rows)
    h->sl_shift1 = h->prm.cluster_shift;
    h->c1_stale = true;
    HIPCHK(h, hipGetLastError());
    h->stage_done = 3;
    return 0;
  }
  h->sorted_key = vsv_launch_sort_stage(st, (vsv_sig*)h->s1in.p, &c->n_s1, 1, pos_bits(h), key_bits(h), (vsv_sig*)h->s1s.p, &c->n_alive1, stage_bufs(h),
                        sort_work(h), h->cap_sigs, dctr(h));
  vsv_launch_cluster(st, (vsv_sig*)h->s1s.p, h->sorted_key, &c->n_alive1, h->prm.cluster_shift, pos_bits(h), (vsv_sig*)h->c1.p, stage_bufs(h), (uint64_t*)h->key2.p, dctr(h));
  HIPCHK(h, hipGetLastError());
  h->stage_done = 3;
  return 0;
}

int enq_merge(vsv_handle* h) {
  hipStream_t st = h->stream;
  Counters* c = dctr(h);
  if (h->big_run) {
    SlimWork w;
    { int ws = slim_work(h, w); if (ws) return ws; }
    h->sl_m = vsv_slim_merge(st, h->sl_e2, h->sl_sorted1, h->sl_ctl2, &c->n_alive1, &c->n_alive2, &c->n_alive3, pos_bits(h), slim_tid_bits(h), h->sl_shift1, h->prm.cluster_shift, w);
    h->merged_stale = true;
    HIPCHK(h, hipGetLastError());
    h->stage_done = 4;
    return 0;
  }
  h->sorted_key = vsv_launch_sort_stage(st, (vsv_sig*)h->c1.p, &c->n_alive1, 2, pos_bits(h), key_bits(h), (vsv_sig*)h->s2s.p, &c->n_alive2, stage_bufs(h),
                        sort_work(h), h->cap_sigs, dctr(h));
  vsv_launch_cluster(st, (vsv_sig*)h->s2s.p, h->sorted_key, &c->n_alive2, h->prm.cluster_shift, pos_bits(h), (vsv_sig*)h->c2.p, stage_bufs(h), (uint64_t*)h->key2.p, dctr(h));
  h->sorted_key = vsv_launch_sort_stage(st, (vsv_sig*)h->c2.p, &c->n_alive2, 3, pos_bits(h), key_bits(h), (vsv_sig*)h->merged.p, &c->n_alive3, stage_bufs(h),
                        sort_work(h), h->cap_sigs, dctr(h), (int32_t*)h->cl.p);     // + pairing state of vsv_launch_pair = -1
  HIPCHK(h, hipGetLastError());
  h->stage_done = 4;
  return 0;
}

int enq_pair(vsv_handle* h) {
  hipStream_t st = h->stream;
  Counters* c = dctr(h);
  if (h->big_run) {
    SlimWork w;
    { int ws = slim_work(h, w); if (ws) return ws; }
    vsv_slim_pair(st, h->sl_m, &c->n_alive3, &c->n_calls, pos_bits(h), slim_tid_bits(h), h->prm.pair_shift, h->prm.pair_window, (const vsv_sig*)h->s1in.p,
                  (vsv_call*)h->calls.p, h->dense_pairing, w, c);
    HIPCHK(h, hipGetLastError());
    h->stage_done = 5;
    return 0;
  }
  vsv_launch_pair(st, (vsv_sig*)h->merged.p, h->sorted_key, &c->n_alive3, h->prm.pair_shift, h->prm.pair_window, (vsv_call*)h->calls_tmp.p, (vsv_call*)h->calls.p,
                  &c->n_calls, stage_bufs(h), (uint64_t*)h->key2.p, (uint32_t*)h->idx2.p, sort_work(h), pos_bits(h), key_bits(h), h->cap_sigs, dctr(h),
                  h->dense_pairing);
  HIPCHK(h, hipGetLastError());
  h->stage_done = 5;
  return 0;
}

int rerun(vsv_handle* h);
int finish(vsv_handle* h) {
  HIPCHK(h, hipSetDevice(h->device));   // the current device is per host thread
  { int js = join_aux(h); if (js) return js; }
  HIPCHK(h, hipMemcpyAsync(h->pinned, h->ctr.p, sizeof(Counters), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->host_ctr = *h->pinned;
  h->pending = false;
  if (h->stage_done >= 1) {
    h->have_history = true; h->ctr_of_run = true;
    if (h->batches_skipped && h->host_ctr.pad[0] != 0) h->host_ctr.err |= ERRB_BATCH_FALLBACK;    // rows of those batches were never placed: again, with the launch
    h->skip_batches = h->host_ctr.pad[0] == 0 && h->batch_runs == 0;
  }
  h->cold_run = false;
  if (h->stage_done >= 5) {   // pairing of the NEXT run: in rounds once a stretch of thousands of rows was met, back to the plain
                              // kernel when the merged table gets small again (the rounds cost ~20 launches)
    static const char* force = vsv_dbg_env("VSV_PAIR");          // tests: "rounds" / "walk"
    if (h->host_ctr.max_stretch > 2048) h->dense_pairing = true;
    else if (h->host_ctr.n_alive3 < 100000 || h->big_run) h->dense_pairing = false;     // (a run on elements measures its stretches whichever way it pairs)
    if (force) h->dense_pairing = force[0] == 'r';
  }
  {  // sort tile size of the NEXT run: small tiles while the largest table stays within ~128 tiles of 4096 rows
    const uint32_t big = h->host_ctr.n_s1 > h->host_ctr.n_cand ? h->host_ctr.n_s1 : h->host_ctr.n_cand;
    h->small_sort_tiles = big <= 128u * 4096u;
  }
  // (the rows of a contig run always get their elements; their 32-bit limits matter only to a run whose stages worked on them)
  if (!h->big_run) h->host_ctr.err &= ~(uint32_t)ERRB_SLIM_FALLBACK;
  // a bucket of sort 1 went through global memory (correct, slow: a pile far from uniform, or a stale size hint): the passes for a while
  if (h->host_ctr.err & ERRB_BUCKET1_SLOW) { h->lsd_sort1_runs = 16; ++h->sort1_slow_runs; h->host_ctr.err &= ~(uint32_t)ERRB_BUCKET1_SLOW; }
  const uint32_t e = h->host_ctr.err;
  h->last_count = h->host_ctr.n_pool;
  static const char* trace = vsv_dbg_env("VSV_TRACE_COUNTERS");
  if (trace) {
    const Counters& c = h->host_ctr;
    fprintf(stderr, "[vsv] n_pool %u n_raw %u n_cand %u n_s1 %u alive %u %u %u calls %u max_stretch %u err %#x pad %u %u %u\n", c.n_pool, c.n_raw, c.n_cand, c.n_s1,
            c.n_alive1, c.n_alive2, c.n_alive3, c.n_calls, c.max_stretch, c.err, c.pad[0], c.pad[1], c.pad[2]);
  }
  if (e & (ERRB_CLR_FALLBACK | ERRB_SORT_FALLBACK | ERRB_SLIM_FALLBACK | ERRB_MERGE_FALLBACK | ERRB_BATCH_FALLBACK)) {
    // ERRB_CLR_FALLBACK: a part of the read-shaped CLR scan held more chunks than its gate state — nothing that run decided can be
    // trusted; same input again with the gate as a separate pass (this handle keeps that form). ERRB_SORT_FALLBACK: a bucket of
    // the bucket sort did not fit in LDS (tables far from uniform, or much larger than the previous run's) — the stages behind it
    // ran on a table that was not written completely; same input again through the LSD passes (real errors show up there; the
    // handle keeps them for a while, or goes back to buckets of the right size at once when the size hint was simply stale).
    // The fallbacks are resolved in a loop, one at a time — the rerun of a CLR fallback may still overflow a bucket — and neither
    // bit ever reaches the success path.
    if (h->in_rerun) return 1;                 // the outermost finish() decides (1 = "again", never leaves this file)
    h->in_rerun = true;
    int st = 1;
    bool stale_hint = false;
    for (int attempt = 0; attempt < 4 && st == 1; ++attempt) {
      const uint32_t ee = h->host_ctr.err;
      if (ee & ERRB_BATCH_FALLBACK) { h->batch_runs = 16; h->skip_batches = false; }
      else if (ee & ERRB_CLR_FALLBACK) h->clr_unfused = true;
      else if (ee & ERRB_SLIM_FALLBACK) h->row_runs = 16;       // a length outside [0, 2^30): the same input on rows (64-bit predicates)
      else if (ee & ERRB_MERGE_FALLBACK) h->lsd_slim_runs = 16; // slots too dense for the windows of the rank-and-merge sorts: the same input through the LSD passes
      else {
        const uint64_t rows_now = h->host_ctr.n_s1 > h->host_ctr.n_cand ? h->host_ctr.n_s1 : h->host_ctr.n_cand;
        stale_hint = rows_now > h->sort_hint_rows + h->sort_hint_rows / 2;
        h->lsd_runs = 16;
      }
      ++h->reruns;
      st = rerun(h);
    }
    h->in_rerun = false;
    if (st == 1) return fail(h, VSV_E_HIP, "bucket-sort / fused-CLR-gate fallback did not converge");
    if (stale_hint) h->lsd_runs = 0;
    return st;
  }
  if (e & ERRB_CAPACITY) {
    char b[160];
    snprintf(b, sizeof b, "signature capacity %lld exceeded (cigar signatures emitted: %u, split candidates: %u)",
             (long long)h->cap_sigs, h->host_ctr.n_pool, h->host_ctr.n_cand);
    int64_t need = (int64_t)h->host_ctr.n_pool + h->host_ctr.n_cand;
    h->last_count = need;
    return fail(h, VSV_E_CAPACITY, b);
  }
  if (e & ERRB_LOOKBACK) return fail(h, VSV_E_HIP, "the scan's look-back over the part counts timed out (internal error)");
  if (e & ERRB_EMPTY_CIGAR) return fail(h, VSV_E_EMPTY_CIGAR, "record with no CIGAR ops / cigar_off not increasing");
  if (e & ERRB_RANGE) return fail(h, VSV_E_INVALID, "a part spans >= 2^30 CIGAR ops, a position exceeds the max_pos hint, a tid lies outside [tid_lo, n_tids), or a qid is >= n_qids");
  if (e & ERRB_REFEND) return fail(h, VSV_E_REFEND, "N/=/X op in an eligible record on the contig path (offset_ref != reference_end)");
  if (e & ERRB_SEQLEN) return fail(h, VSV_E_SEQLEN, "a walked record stores a SEQ whose length differs from its CIGAR's query length (VSV_F_SEQ_MISMATCH)");
  if (e & ERRB_ZERODIV) return fail(h, VSV_E_ZERODIV, "CLR gate on a record without M ops");
  if (e & ERRB_UNSORTED) return fail(h, VSV_E_UNSORTED, "split pair with pos1 > pos2");
  if (e & ERRB_READLEN) return fail(h, VSV_E_READLEN, "split pair with unequal read lengths");
  return 0;
}

// the stages of the last call again (bucket-sort fallback): the records are still bound, every stage recomputes its tables
int rerun(vsv_handle* h) {
  int st;
  if (h->stage_done == 0 && h->bnd_stage == 2) {
    HIPCHK(h, hipMemsetAsync(&dctr(h)->err, 0, sizeof(uint32_t), h->stream));     // the other counters describe the candidates
    if (h->totals.p) HIPCHK(h, hipMemsetAsync(h->totals.p, 0, (size_t)MAX_SORT_PASSES * 2048 * sizeof(uint32_t), h->stream));
    h->pass_cursor = 0;
    h->group_cursor = MAX_GROUP_SLOTS;      // (the group sums are not zeroed here: these sorts keep their scan launch)
    vsv_launch_bnd_pair(h->stream, (const vsv_bnd*)h->s1in.p, h->segs.contig_rank, bits_for((uint64_t)h->segs.n_tids + 1), h->bnd_prm,
                        (vsv_bnd*)h->s1s.p, (vsv_bnd*)h->c1.p, dctr(h), stage_bufs(h), sort_work_plain(h), h->cap_sigs);
    HIPCHK(h, hipGetLastError());
    return finish(h);
  }
  const int upto = h->stage_done;
  if (upto < 1) return fail(h, VSV_E_HIP, "bucket sort overflow outside a signature run");
  h->fork_split = upto >= 2;
  st = enq_scan(h);
  h->fork_split = false;
  if (st) return st;
  if (upto >= 2 && (st = enq_split(h))) return st;
  if (upto >= 3 && (st = enq_stage1(h))) return st;
  if (upto >= 4 && (st = enq_merge(h))) return st;
  if (upto >= 5 && (st = enq_pair(h))) return st;
  return finish(h);
}

int start(vsv_handle* h, const vsv_records* recs, const vsv_params* p) {
  if (!h || !p) return VSV_E_INVALID;
  if (p->dtype < 0 || p->dtype > VSV_DTYPE_CUTESV) return fail(h, VSV_E_INVALID, "bad dtype");
  if (p->scan_layout < VSV_SCAN_AUTO || p->scan_layout > VSV_SCAN_CONTIGS) return fail(h, VSV_E_INVALID, "bad scan_layout");
  HIPCHK(h, hipSetDevice(h->device));
  h->prm = *p;
  return bind_records(h, recs);
}

}  // namespace

// ============================================ C ABI ==================================================
extern "C" {

int vsv_abi_version(void) { return VSV_ABI_VERSION; }

const char* vsv_status_string(int s) {
  switch (s) {
    case VSV_OK: return "VSV_OK";
    case VSV_E_INVALID: return "VSV_E_INVALID";
    case VSV_E_HIP: return "VSV_E_HIP";
    case VSV_E_CAPACITY: return "VSV_E_CAPACITY";
    case VSV_E_EMPTY_CIGAR: return "VSV_E_EMPTY_CIGAR";
    case VSV_E_REFEND: return "VSV_E_REFEND";
    case VSV_E_READLEN: return "VSV_E_READLEN";
    case VSV_E_UNSORTED: return "VSV_E_UNSORTED";
    case VSV_E_ZERODIV: return "VSV_E_ZERODIV";
    case VSV_E_NO_DEVICE: return "VSV_E_NO_DEVICE";
    case VSV_E_SEQLEN: return "VSV_E_SEQLEN";
  }
  return "VSV_E_?";
}

int vsv_default_params(int dtype, vsv_params* p) {
  if (!p || dtype < 0 || dtype > VSV_DTYPE_CUTESV) return VSV_E_INVALID;
  memset(p, 0, sizeof *p);
  p->dtype = dtype;
  p->min_svlen = dtype == VSV_DTYPE_SVIM ? 40 : 30;
  p->min_cigar_mapq = dtype == VSV_DTYPE_SVIM ? 20 : 50;
  p->min_split_mapq = dtype == VSV_DTYPE_READS ? 0 : 50;
  p->max_split_svlen = 50000;
  p->cluster_shift = 100;
  p->pair_shift = 200;
  p->pair_window = 1000;
  p->enable_split = dtype == VSV_DTYPE_SVIM ? 0 : 1;
  if (dtype == VSV_DTYPE_CUTESV) {   // sig_extract.py defaults (SE:703-747): -sl 10, -q 20, -mi 100, -md 0; CIGAR stage only
    p->min_svlen = 10; p->min_cigar_mapq = 20; p->enable_split = 0;
    p->merge_ins_threshold = 100; p->merge_del_threshold = 0;
  }
  return 0;
}

int vsv_create(int device_id, void* hip_stream, vsv_handle** out) {
  if (!out) return VSV_E_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) return VSV_E_NO_DEVICE;
  vsv_handle* h = new vsv_handle();
  h->device = device_id;
  h->stream = (hipStream_t)hip_stream;
  if (hipSetDevice(device_id) != hipSuccess) { delete h; return VSV_E_NO_DEVICE; }
  if (hipMalloc(&h->arena.p, ARENA_BYTES) != hipSuccess) { delete h; return VSV_E_HIP; }
  h->arena.bytes = ARENA_BYTES;
  char* ar = (char*)h->arena.p;
  h->ctr.p = ar; h->ctr.bytes = ARENA_CTR;
  h->shard_cnt.p = ar + ARENA_CTR; h->shard_cnt.bytes = ARENA_SHARD;
  h->tile_cnt = (uint32_t*)(ar + ARENA_CTR + ARENA_SHARD);
  h->totals.p = ar + ARENA_CTR + ARENA_SHARD + ARENA_TILES; h->totals.bytes = ARENA_TOTALS;
  h->groups = (uint32_t*)(ar + ARENA_CTR + ARENA_SHARD + ARENA_TILES + ARENA_TOTALS);
  if (hipHostMalloc((void**)&h->pinned, sizeof(Counters)) != hipSuccess) { hipFree(h->arena.p); delete h; return VSV_E_HIP; }
  hipEventCreate(&h->ev0);
  hipEventCreate(&h->ev1);
  // (the auxiliary streams are created on first use: a process that keeps several handles busy maps every stream it creates onto a
  // handful of hardware queues, and streams nobody uses still shift that mapping — three engines measured 0.55 instead of 0.44 ms
  // per step with two idle streams per handle)
  hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
  hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming);
  memset(&h->host_ctr, 0, sizeof h->host_ctr);
  { const char* pm = vsv_dbg_env("VSV_PAIR"); h->dense_pairing = pm && pm[0] == 'r'; }     // tests: the round-based pairing from the first run
  *out = h;
  return 0;
}

void vsv_destroy(vsv_handle* h) {
  if (!h) return;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  if (h->aux) { hipStreamSynchronize(h->aux); hipStreamDestroy(h->aux); }
  if (h->aux2) { hipStreamSynchronize(h->aux2); hipStreamDestroy(h->aux2); }
  if (h->ev_fork) hipEventDestroy(h->ev_fork);
  if (h->ev_join) hipEventDestroy(h->ev_join);
  DevBuf* bufs[] = {&h->scan_tmp2, &h->r_pos, &h->r_tid, &h->r_qid, &h->r_off, &h->r_mapq, &h->r_flag, &h->r_cigar, &h->part_rb, &h->part_count,
                    &h->part_off, &h->scan_tmp, &h->lbw, &h->l_prec, &h->l_agg, &h->l_carry_r, &h->l_carry_q, &h->l_tiles, &h->z_crctab, &h->z_crc, &h->pool, &h->pool_key, &h->raw0, &h->s1in, &h->s1s, &h->c1, &h->s2s, &h->c2,
                    &h->merged, &h->calls_tmp, &h->calls, &h->reads, &h->tab, &h->tab2, &h->qlbw, &h->clbw, &h->blk_cnt, &h->blk_off, &h->ckey, &h->crec,
                    &h->okey, &h->oval, &h->key, &h->idx, &h->cl, &h->key2, &h->idx2, &h->key_alt, &h->val_alt, &h->hist, &h->arena, &h->g_off, &h->g_qs, &h->g_qe, &h->g_rid, &h->g_rs,
                    &h->g_re, &h->g_rev, &h->g_hap, &h->g_len, &h->g_rank, &h->j_cpos, &h->j_clen, &h->j_spos, &h->j_slen, &h->j_send, &h->j_out, &h->j_err, &h->z_comp, &h->z_coff, &h->z_ooff, &h->z_out, &h->z_stat, &h->p_spec, &h->p_cnt, &h->p_land, &h->p_base, &h->p_recoff, &h->p_pos, &h->p_tid,
                    &h->p_mapq, &h->p_flag, &h->p_lseq, &h->p_sflag, &h->p_ncig, &h->p_cgsrc, &h->p_hash, &h->p_keep, &h->p_kidx, &h->p_cigoff, &h->p_sums,
                    &h->p_tot, &h->p_err, &h->o_pos, &h->o_tid, &h->o_qid, &h->o_cigoff, &h->o_mapq, &h->o_flag, &h->o_cigar, &h->o_lseq, &h->o_sflag,
                    &h->o_hash, &h->o_recoff, &h->o_first, &h->o_rank, &h->o_nlen, &h->o_noff, &h->o_blob, &h->o_n, &h->o_names, &h->o_nmoff, &h->o_nmlen, &h->gflag, &h->cmask, &h->cs_tra, &h->o_saoff, &h->o_salen, &h->o_saloc, &h->o_sa,
                    &h->o_sqoff, &h->o_sqlen, &h->o_sqloc, &h->o_seq, &h->o_recseq, &h->q_rec, &h->q_start, &h->q_len, &h->q_rev, &h->q_ooff, &h->q_out};
  for (DevBuf* b : bufs) if (b->p) hipFree(b->p);
  for (DevBuf& b : h->sl) if (b.p) hipFree(b.p);
  if (h->sl_hj.p) hipFree(h->sl_hj.p);
  for (DevBuf* b : {&h->cinfo, &h->cord, &h->oc1}) if (b->p) hipFree(b->p);
  if (h->sl_done.p) hipFree(h->sl_done.p);
  if (h->pinned) hipHostFree(h->pinned);
  if (h->pin_buf) hipHostFree(h->pin_buf);
  if (h->names_pin) hipHostFree(h->names_pin);
  if (h->ev0) hipEventDestroy(h->ev0);
  if (h->ev1) hipEventDestroy(h->ev1);
  delete h;
}

const char* vsv_last_error(vsv_handle* h) { return h ? h->err.c_str() : "null handle"; }
int64_t vsv_last_count(vsv_handle* h) { return h ? h->last_count : 0; }
int64_t vsv_rerun_count(vsv_handle* h) { return h ? h->reruns : 0; }
int vsv_path_counts(vsv_handle* h, int64_t* element_runs, int64_t* cold_syncs) {
  if (!h) return VSV_E_INVALID;
  if (element_runs) *element_runs = h->element_runs;
  if (cold_syncs) *cold_syncs = h->cold_syncs;
  return 0;
}
int64_t vsv_sort1_slow_count(vsv_handle* h) { return h ? h->sort1_slow_runs : 0; }

int vsv_wait_for_stream(vsv_handle* h, void* producer_hip_stream) {
  if (!h) return VSV_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  if ((hipStream_t)producer_hip_stream == h->stream) return 0;          // same stream: already ordered
  hipEvent_t ev;
  HIPCHK(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t e = hipEventRecord(ev, (hipStream_t)producer_hip_stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(h->stream, ev, 0);
  (void)hipEventDestroy(ev);                                              // released once the wait has been satisfied
  if (e != hipSuccess) return fail(h, VSV_E_HIP, std::string("vsv_wait_for_stream: ") + hipGetErrorString(e));
  return 0;
}

int vsv_reserve(vsv_handle* h, int64_t max_records, int64_t max_ops, int64_t max_sigs) {
  if (!h) return VSV_E_INVALID;
  if (hipSetDevice(h->device) != hipSuccess) return VSV_E_HIP;
  if (max_sigs > 0x7FFFFF00ll) return fail(h, VSV_E_INVALID, "max_sigs too large");
  if (max_sigs != h->cap_sigs && max_sigs > 0 && max_sigs < h->cap_sigs) {
    // shrinking the row capacity only lowers the logical limit (used by capacity tests)
    h->cap_sigs = max_sigs;
    return 0;
  }
  return reserve(h, max_records, max_ops, max_sigs);
}

int vsv_reserve_large_tables(vsv_handle* h) {
  if (!h) return VSV_E_INVALID;
  if (hipSetDevice(h->device) != hipSuccess) return VSV_E_HIP;
  if (h->cap_sigs <= 0) { int st = reserve(h, 1, 1, 1 << 22); if (st) return st; }
  return slim_alloc(h, true);
}

int vsv_cigar_scan(vsv_handle* h, const vsv_records* recs, const vsv_params* p) {
  int st = start(h, recs, p);
  if (st) return st;
  if ((st = enq_scan(h))) return st;
  return finish(h);
}

int vsv_split_pairs(vsv_handle* h, const vsv_records* recs, const vsv_params* p) {
  if (!h || h->stage_done < 1) return fail(h, VSV_E_INVALID, "vsv_cigar_scan must run first");
  (void)recs;
  if (p) { h->prm.enable_split = p->enable_split; h->prm.min_split_mapq = p->min_split_mapq; h->prm.max_split_svlen = p->max_split_svlen; }
  h->split_cands_done = false;          // the candidates depend on min_split_mapq
  int st = enq_split(h);
  if (st) return st;
  return finish(h);
}

int vsv_sort_cluster(vsv_handle* h, const vsv_params* p) {
  if (!h || h->stage_done < 2) return fail(h, VSV_E_INVALID, "vsv_split_pairs must run first");
  if (!is_contig(h->prm.dtype)) return fail(h, VSV_E_INVALID, "clustering exists only on the contig path");
  if (p) h->prm.cluster_shift = p->cluster_shift;
  int st = enq_stage1(h);
  if (st) return st;
  return finish(h);
}

int vsv_merge_sources(vsv_handle* h, const vsv_params* p) {
  if (!h || h->stage_done < 3) return fail(h, VSV_E_INVALID, "vsv_sort_cluster must run first");
  int st;
  // (an element run consumed the stage-1 tables when it merged them: called again, the stage in front is enqueued again, with ITS shift)
  if (h->big_run && h->stage_done >= 4) { const int keep = h->prm.cluster_shift; h->prm.cluster_shift = h->sl_shift1; st = enq_stage1(h); h->prm.cluster_shift = keep; if (st) return st; }
  if (p) h->prm.cluster_shift = p->cluster_shift;
  st = enq_merge(h);
  if (st) return st;
  return finish(h);
}

int vsv_pair_haplotypes(vsv_handle* h, const vsv_params* p) {
  if (!h || h->stage_done < 4) return fail(h, VSV_E_INVALID, "vsv_merge_sources must run first");
  if (p) { h->prm.pair_shift = p->pair_shift; h->prm.pair_window = p->pair_window; }
  int st;
  if (h->big_run && h->stage_done >= 5) {     // called again on an element run: its sorts take their scratch slots from the run's arena — start over from stage 1
    const int keep = h->prm.cluster_shift;
    h->prm.cluster_shift = h->sl_shift1;
    st = enq_stage1(h);
    h->prm.cluster_shift = keep;
    if (st || (st = enq_merge(h))) return st;
  } else if (h->stage_done >= 5) {            // ... on a row run: the pairing state (one word per merged row, -1 = free) was left by the sort in front
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipMemsetAsync(h->cl.p, 0xFF, (size_t)h->cap_sigs * sizeof(int32_t), h->stream));
  }
  st = enq_pair(h);
  if (st) return st;
  return finish(h);
}

int vsv_run_chromosome_async(vsv_handle* h, const vsv_records* recs, const vsv_params* p) {
  int st = start(h, recs, p);
  if (st) return st;
  h->fork_split = true;
  st = enq_scan(h);
  h->fork_split = false;
  if (st) return st;
  if (p->dtype != VSV_DTYPE_SVIM && p->dtype != VSV_DTYPE_CUTESV) { if ((st = enq_split(h))) return st; }
  if (is_contig(p->dtype)) {
    if ((st = enq_stage1(h))) return st;
    if ((st = enq_merge(h))) return st;
    if ((st = enq_pair(h))) return st;
  }
  h->pending = true;
  return 0;
}

int vsv_stream_ceiling(vsv_handle* h, const void* dev_buf, int64_t bytes, int32_t reps, double* read_gbs, double* copy_gbs) {
  if (!h || !dev_buf || bytes < (1 << 20) || reps < 1 || !read_gbs || !copy_gbs) return VSV_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t n = (size_t)bytes & ~(size_t)15;
  void* dst = nullptr; uint32_t* sink = nullptr;
  HIPCHK(h, hipMalloc(&dst, n));
  if (hipMalloc((void**)&sink, 1024) != hipSuccess) { (void)hipFree(dst); return fail(h, VSV_E_HIP, "hipMalloc failed"); }
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  float best_r = 1e30f, best_c = 1e30f;
  for (int r = 0; r < reps + 1; ++r) {                       // first round warms up
    float ms = 0;
    (void)hipEventRecord(a, h->stream); vsv_launch_stream_read(h->stream, dev_buf, n, sink); (void)hipEventRecord(b, h->stream);
    (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b);
    if (r && ms < best_r) best_r = ms;
    (void)hipEventRecord(a, h->stream); vsv_launch_stream_copy(h->stream, dev_buf, dst, n); (void)hipEventRecord(b, h->stream);
    (void)hipEventSynchronize(b); (void)hipEventElapsedTime(&ms, a, b);
    if (r && ms < best_c) best_c = ms;
  }
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  (void)hipFree(dst); (void)hipFree(sink);
  HIPCHK(h, hipGetLastError());
  *read_gbs = (double)n / (best_r * 1e-3) / 1e9;
  *copy_gbs = 2.0 * (double)n / (best_c * 1e-3) / 1e9;      // bytes read + bytes written
  return 0;
}

int vsv_finish(vsv_handle* h) {
  if (!h) return VSV_E_INVALID;
  return finish(h);
}

int vsv_run_chromosome(vsv_handle* h, const vsv_records* recs, const vsv_params* p) {
  int st = vsv_run_chromosome_async(h, recs, p);
  if (st) return st;
  return finish(h);
}

int vsv_default_bnd_params(vsv_bnd_params* p) {
  if (!p) return VSV_E_INVALID;
  memset(p, 0, sizeof *p);
  p->min_sv_size = 40; p->max_sv_size = 100000; p->query_gap_tolerance = 50; p->query_overlap_tolerance = 50;
  p->reference_gap_tolerance = 50; p->reference_overlap_tolerance = 50; p->partition_max_distance = 1000;
  p->pair_distance = 900; p->max_partition = 10;
  return 0;
}

// host-side segment tables: the offsets must tile [0, n_segs) and every reference id must index the contig tables
static int validate_segments(vsv_handle* h, const vsv_segments* sg, bool check_tids) {
  if (sg->on_device || sg->n_reads == 0) return 0;
  if (!sg->seg_off || (sg->n_segs > 0 && (!sg->q_start || !sg->q_end || !sg->ref_id || !sg->ref_start || !sg->ref_end || !sg->is_reverse)))
    return fail(h, VSV_E_INVALID, "segment arrays are NULL");
  if (sg->seg_off[0] != 0 || sg->seg_off[sg->n_reads] != (uint64_t)sg->n_segs) return fail(h, VSV_E_INVALID, "seg_off does not span [0, n_segs)");
  for (int64_t r = 0; r < sg->n_reads; ++r)
    if (sg->seg_off[r + 1] < sg->seg_off[r]) return fail(h, VSV_E_INVALID, "seg_off does not ascend");
  if (check_tids)
    for (int64_t k = 0; k < sg->n_segs; ++k)
      if (sg->ref_id[k] < 0 || sg->ref_id[k] >= sg->n_tids) return fail(h, VSV_E_INVALID, "segment reference id outside [0, n_tids)");
  return 0;
}

int vsv_bnd_segments(vsv_handle* h, const vsv_segments* sg, const vsv_bnd_params* p) {
  if (!h || !sg || !p) return VSV_E_INVALID;
  if (sg->n_reads < 0 || sg->n_segs < sg->n_reads || sg->n_tids <= 0) return fail(h, VSV_E_INVALID, "bad segment counts");
  { int vs = validate_segments(h, sg, true); if (vs) return vs; }
  HIPCHK(h, hipSetDevice(h->device));
  int st = reserve(h, 1, 1, h->cap_sigs > 0 ? h->cap_sigs : (1 << 22));
  if (st) return st;
  if (sg->n_segs - sg->n_reads > h->cap_sigs) { h->last_count = sg->n_segs - sg->n_reads; return fail(h, VSV_E_CAPACITY, "more segment pairs than row capacity"); }
  vsv_segments d = *sg;
  if (!sg->on_device) {
    const size_t nr = (size_t)sg->n_reads, ns = (size_t)sg->n_segs, nt = (size_t)sg->n_tids;
    if ((st = upload(h, h->g_off, sg->seg_off, (nr + 1) * 8))) return st;
    if ((st = upload(h, h->g_qs, sg->q_start, ns * 4))) return st;
    if ((st = upload(h, h->g_qe, sg->q_end, ns * 4))) return st;
    if ((st = upload(h, h->g_rid, sg->ref_id, ns * 4))) return st;
    if ((st = upload(h, h->g_rs, sg->ref_start, ns * 4))) return st;
    if ((st = upload(h, h->g_re, sg->ref_end, ns * 4))) return st;
    if ((st = upload(h, h->g_rev, sg->is_reverse, ns))) return st;
    if ((st = upload(h, h->g_hap, sg->hap, nr))) return st;
    if ((st = upload(h, h->g_len, sg->contig_len, nt * 4))) return st;
    if ((st = upload(h, h->g_rank, sg->contig_rank, nt * 4))) return st;
    d.seg_off = (const uint64_t*)h->g_off.p; d.q_start = (const int32_t*)h->g_qs.p; d.q_end = (const int32_t*)h->g_qe.p;
    d.ref_id = (const int32_t*)h->g_rid.p; d.ref_start = (const int32_t*)h->g_rs.p; d.ref_end = (const int32_t*)h->g_re.p;
    d.is_reverse = (const uint8_t*)h->g_rev.p; d.hap = (const uint8_t*)h->g_hap.p; d.contig_len = (const int32_t*)h->g_len.p;
    d.contig_rank = (const int32_t*)h->g_rank.p;
  }
  h->segs = d;
  { int rs = reset_run_state(h); if (rs) return rs; }
  vsv_launch_bnd_segments(h->stream, d, *p, (vsv_bnd*)h->s1in.p, (uint32_t)h->cap_sigs, dctr(h));
  HIPCHK(h, hipGetLastError());
  h->bnd_stage = 1;
  h->stage_done = 0;   // the signature tables share buffers with the BND tables
  return finish(h);
}

int vsv_cutesv_split(vsv_handle* h, const vsv_segments* sg, const int32_t* read_len, const uint32_t* read_rec, int32_t sv_size,
                     int32_t max_size, int32_t max_split_parts) {
  if (!h || !sg) return VSV_E_INVALID;
  if (sg->n_reads < 0 || sg->n_segs < 0) return fail(h, VSV_E_INVALID, "bad segment counts");
  if (sg->n_reads > 0 && (!read_len || !read_rec || !sg->seg_off)) return fail(h, VSV_E_INVALID, "read arrays are NULL");
  { int vs = validate_segments(h, sg, false); if (vs) return vs; }
  HIPCHK(h, hipSetDevice(h->device));
  const int64_t rows = 2 * sg->n_segs + sg->n_reads;    // upper bound of the rows a read can produce, summed
  int st = reserve(h, 1, 1, rows > h->cap_sigs ? rows : (h->cap_sigs > 0 ? h->cap_sigs : (1 << 22)));
  if (st) return st;
  vsv_segments d = *sg;
  const int32_t* rl = read_len; const uint32_t* rr = read_rec;
  if (!sg->on_device) {
    const size_t nr = (size_t)sg->n_reads, ns = (size_t)sg->n_segs;
    if ((st = upload(h, h->g_off, sg->seg_off, (nr + 1) * 8))) return st;
    if ((st = upload(h, h->g_qs, sg->q_start, ns * 4))) return st;
    if ((st = upload(h, h->g_qe, sg->q_end, ns * 4))) return st;
    if ((st = upload(h, h->g_rid, sg->ref_id, ns * 4))) return st;
    if ((st = upload(h, h->g_rs, sg->ref_start, ns * 4))) return st;
    if ((st = upload(h, h->g_re, sg->ref_end, ns * 4))) return st;
    if ((st = upload(h, h->g_rev, sg->is_reverse, ns))) return st;
    if ((st = upload(h, h->g_len, read_len, nr * 4))) return st;
    if ((st = upload(h, h->g_rank, read_rec, nr * 4))) return st;
    d.seg_off = (const uint64_t*)h->g_off.p; d.q_start = (const int32_t*)h->g_qs.p; d.q_end = (const int32_t*)h->g_qe.p;
    d.ref_id = (const int32_t*)h->g_rid.p; d.ref_start = (const int32_t*)h->g_rs.p; d.ref_end = (const int32_t*)h->g_re.p;
    d.is_reverse = (const uint8_t*)h->g_rev.p;
    rl = (const int32_t*)h->g_len.p; rr = (const uint32_t*)h->g_rank.p;
  }
  { int rs = reset_run_state(h); if (rs) return rs; }
  if ((st = ensure(h, h->j_spos, (size_t)sg->n_reads * 4 + 16))) return st;      // rows per read
  if ((st = ensure(h, h->j_slen, (size_t)sg->n_reads * 4 + 16))) return st;      // their exclusive scan
  if ((st = ensure(h, h->j_err, (size_t)(sg->n_reads / 2048 + 4) * 4 + 256))) return st;
  if ((st = ensure(h, h->cs_tra, (size_t)sg->n_reads + 16))) return st;
  h->cs_reads = sg->n_reads;
  if (sg->n_reads > 0)
    vsv_launch_cutesv_split(h->stream, d, rl, rr, sv_size, max_size, max_split_parts, (vsv_sig*)h->c2.p, (uint32_t)h->cap_sigs,
                            (uint32_t*)h->j_spos.p, (uint32_t*)h->j_slen.p, (uint32_t*)h->j_err.p, dctr(h), (uint8_t*)h->cs_tra.p);
  HIPCHK(h, hipGetLastError());
  st = finish(h);
  h->cutesv_rows = sg->n_reads > 0 ? (int64_t)h->host_ctr.n_alive2 : 0;
  return st;
}

int vsv_cutesv_split_tra(vsv_handle* h, uint8_t* has_tra, int64_t n_reads) {
  if (!h || n_reads < 0 || (n_reads > 0 && !has_tra)) return VSV_E_INVALID;
  if (h->cs_reads != n_reads) return fail(h, VSV_E_INVALID, "vsv_cutesv_split_tra: n_reads differs from the last vsv_cutesv_split");
  if (n_reads == 0) return 0;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(has_tra, h->cs_tra.p, (size_t)n_reads, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int vsv_bnd_set_candidates(vsv_handle* h, const vsv_bnd* rows, int64_t n, const int32_t* contig_rank, int32_t n_tids, int on_device) {
  if (!h || n < 0 || n_tids <= 0 || !contig_rank || (n > 0 && !rows)) return VSV_E_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  int st = reserve(h, 1, 1, h->cap_sigs > 0 ? h->cap_sigs : (1 << 22));
  if (st) return st;
  if (n > h->cap_sigs) { h->last_count = n; return fail(h, VSV_E_CAPACITY, "more candidate rows than row capacity"); }
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (n) HIPCHK(h, hipMemcpyAsync(h->s1in.p, rows, (size_t)n * sizeof(vsv_bnd), kind, h->stream));
  if ((st = ensure(h, h->g_rank, (size_t)n_tids * 4 + 16))) return st;
  HIPCHK(h, hipMemcpyAsync(h->g_rank.p, contig_rank, (size_t)n_tids * 4, kind, h->stream));
  if (h->totals.p) HIPCHK(h, hipMemsetAsync(h->totals.p, 0, (size_t)MAX_SORT_PASSES * 2048 * sizeof(uint32_t), h->stream));
  h->pass_cursor = 0;
  h->group_cursor = MAX_GROUP_SLOTS;        // (the group sums are not zeroed here: these sorts keep their scan launch)
  Counters c; memset(&c, 0, sizeof c);
  c.n_s1 = (uint32_t)n;
  *h->pinned = c;
  HIPCHK(h, hipMemcpyAsync(h->ctr.p, h->pinned, sizeof(Counters), hipMemcpyHostToDevice, h->stream));
  memset(&h->segs, 0, sizeof h->segs);
  h->segs.contig_rank = (const int32_t*)h->g_rank.p;
  h->segs.n_tids = n_tids;
  h->bnd_stage = 1;
  h->stage_done = 0;
  return finish(h);
}

int vsv_bnd_pair(vsv_handle* h, const vsv_bnd_params* p) {
  if (!h || !p || h->bnd_stage < 1) return fail(h, VSV_E_INVALID, "vsv_bnd_segments must run first");
  if (p->max_partition < 0 || p->max_partition > 16) return fail(h, VSV_E_INVALID, "max_partition must lie in [0, 16] (the pairing kernel holds a partition in registers)");
  HIPCHK(h, hipSetDevice(h->device));
  h->bnd_prm = *p;
  vsv_launch_bnd_pair(h->stream, (const vsv_bnd*)h->s1in.p, h->segs.contig_rank, bits_for((uint64_t)h->segs.n_tids + 1), *p,
                      (vsv_bnd*)h->s1s.p, (vsv_bnd*)h->c1.p, dctr(h), stage_bufs(h), sort_work_plain(h), h->cap_sigs);
  HIPCHK(h, hipGetLastError());
  h->bnd_stage = 2;
  return finish(h);
}

int vsv_default_support_params(vsv_support_params* p) {
  if (!p) return VSV_E_INVALID;
  p->max_comp_svlen = 250; p->max_dist = 1000; p->max_shift = 500; p->pad = 0; p->min_size_sim = 0.5;   // FP:8-11
  return 0;
}

int vsv_support_join(vsv_handle* h, const int32_t* call_pos, const int32_t* call_len, int64_t n_calls, const int32_t* sig_pos,
                     const int32_t* sig_len, int64_t n_sigs, const vsv_support_params* p, int on_device, uint32_t* support) {
  if (!h || !p) return VSV_E_INVALID;
  if (n_calls < 0 || n_sigs < 0) return fail(h, VSV_E_INVALID, "negative row count");
  if (n_calls > 0 && (!call_pos || !call_len || !support)) return fail(h, VSV_E_INVALID, "call arrays are NULL");
  if (n_sigs > 0 && (!sig_pos || !sig_len)) return fail(h, VSV_E_INVALID, "signature arrays are NULL");
  if (p->max_dist < 0) return fail(h, VSV_E_INVALID, "max_dist < 0");
  if (n_calls == 0) return 0;
  HIPCHK(h, hipSetDevice(h->device));
  int st;
  if ((st = ensure(h, h->j_err, 256))) return st;
  HIPCHK(h, hipMemsetAsync(h->j_err.p, 0, 4, h->stream));
  const int32_t *cp = call_pos, *cl = call_len, *sp = sig_pos, *sl = sig_len;
  uint32_t* out = support;
  if (!on_device) {
    if ((st = upload(h, h->j_cpos, call_pos, (size_t)n_calls * 4))) return st;
    if ((st = upload(h, h->j_clen, call_len, (size_t)n_calls * 4))) return st;
    if ((st = upload(h, h->j_spos, sig_pos, (size_t)n_sigs * 4))) return st;
    if ((st = upload(h, h->j_slen, sig_len, (size_t)n_sigs * 4))) return st;
    if ((st = ensure(h, h->j_out, (size_t)n_calls * 4))) return st;
    cp = (const int32_t*)h->j_cpos.p; cl = (const int32_t*)h->j_clen.p; sp = (const int32_t*)h->j_spos.p; sl = (const int32_t*)h->j_slen.p;
    out = (uint32_t*)h->j_out.p;
  }
  vsv_launch_support_join(h->stream, cp, cl, n_calls, sp, sl, n_sigs, *p, out, (uint32_t*)h->j_err.p);
  HIPCHK(h, hipGetLastError());
  uint32_t e = 0;
  if (!on_device) HIPCHK(h, hipMemcpyAsync(support, out, (size_t)n_calls * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(&e, h->j_err.p, 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (e & ERRB_UNSORTED) return fail(h, VSV_E_UNSORTED, "read signatures are not sorted by position");
  return 0;
}

// shared body of the two coverage entry points: up to three call/signature arrays each, int64 result
static int cov_common(vsv_handle* h, const int32_t* const* call_arr, int n_call_arr, int64_t n_calls, const int32_t* const* sig_arr,
                      int n_sig_arr, int64_t n_sigs, int32_t flanking, int on_device, int64_t* cov, bool is_del) {
  if (!h) return VSV_E_INVALID;
  if (n_calls < 0 || n_sigs < 0 || flanking < 0) return fail(h, VSV_E_INVALID, "negative count or flanking");
  for (int k = 0; k < n_call_arr; ++k) if (n_calls > 0 && !call_arr[k]) return fail(h, VSV_E_INVALID, "call arrays are NULL");
  for (int k = 0; k < n_sig_arr; ++k) if (n_sigs > 0 && !sig_arr[k]) return fail(h, VSV_E_INVALID, "signature arrays are NULL");
  if (n_calls == 0) return 0;
  if (!cov) return fail(h, VSV_E_INVALID, "cov is NULL");
  HIPCHK(h, hipSetDevice(h->device));
  int st;
  if ((st = ensure(h, h->j_err, 256))) return st;
  HIPCHK(h, hipMemsetAsync(h->j_err.p, 0, 8, h->stream));
  const int32_t* c[2] = {call_arr[0], n_call_arr > 1 ? call_arr[1] : nullptr};
  const int32_t* g[3] = {sig_arr[0], sig_arr[1], n_sig_arr > 2 ? sig_arr[2] : nullptr};
  int64_t* out = cov;
  if (!on_device) {
    DevBuf* cb[2] = {&h->j_cpos, &h->j_clen};
    DevBuf* sb[3] = {&h->j_spos, &h->j_slen, &h->j_send};
    for (int k = 0; k < n_call_arr; ++k) { if ((st = upload(h, *cb[k], call_arr[k], (size_t)n_calls * 4))) return st; c[k] = (const int32_t*)cb[k]->p; }
    for (int k = 0; k < n_sig_arr; ++k) { if ((st = upload(h, *sb[k], sig_arr[k], (size_t)n_sigs * 4))) return st; g[k] = (const int32_t*)sb[k]->p; }
    if ((st = ensure(h, h->j_out, (size_t)n_calls * 8))) return st;
    out = (int64_t*)h->j_out.p;
  }
  if (is_del) vsv_launch_cov_del(h->stream, c[0], c[1], n_calls, g[0], g[1], g[2], n_sigs, flanking, out, (uint32_t*)h->j_err.p);
  else vsv_launch_cov_ins(h->stream, c[0], n_calls, g[0], g[1], n_sigs, flanking, out, (uint32_t*)h->j_err.p);
  HIPCHK(h, hipGetLastError());
  uint32_t e = 0;
  if (!on_device) HIPCHK(h, hipMemcpyAsync(cov, out, (size_t)n_calls * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(&e, h->j_err.p, 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (e & ERRB_UNSORTED) return fail(h, VSV_E_UNSORTED, "signatures are not sorted by position");
  return 0;
}

int vsv_support_cov_ins(vsv_handle* h, const int32_t* call_pos, int64_t n_calls, const int32_t* sig_pos, const int32_t* sig_len,
                        int64_t n_sigs, int32_t flanking, int on_device, int64_t* cov) {
  const int32_t* ca[1] = {call_pos};
  const int32_t* sa[2] = {sig_pos, sig_len};
  return cov_common(h, ca, 1, n_calls, sa, 2, n_sigs, flanking, on_device, cov, false);
}

int vsv_support_cov_del(vsv_handle* h, const int32_t* call_start, const int32_t* call_end, int64_t n_calls, const int32_t* sig_start,
                        const int32_t* sig_end, const int32_t* sig_svlen, int64_t n_sigs, int32_t flanking, int on_device, int64_t* cov) {
  const int32_t* ca[2] = {call_start, call_end};
  const int32_t* sa[3] = {sig_start, sig_end, sig_svlen};
  return cov_common(h, ca, 2, n_calls, sa, 3, n_sigs, flanking, on_device, cov, true);
}

// inflate into h->z_out and leave it there (shared by vsv_bgzf_inflate's device half and the device parse)
static int inflate_to_device(vsv_handle* h, const uint8_t* comp, const uint64_t* comp_off, const uint32_t* isize, int64_t n, std::vector<uint64_t>& ooff,
                             const uint32_t* expect_crc = nullptr) {
  ooff.assign((size_t)n + 1, 0);
  for (int64_t i = 0; i < n; ++i) {
    if (comp_off[i + 1] < comp_off[i] || isize[i] > 65536u) return fail(h, VSV_E_INVALID, "bad BGZF member table");
    ooff[i + 1] = ooff[i] + isize[i];
  }
  const size_t cbytes = (size_t)(comp_off[n] - comp_off[0]), obytes = (size_t)ooff[n];
  int st;
  std::vector<uint64_t> coff((size_t)n + 1);
  for (int64_t i = 0; i <= n; ++i) coff[i] = comp_off[i] - comp_off[0];
  if ((st = upload(h, h->z_coff, coff.data(), (size_t)(n + 1) * 8))) return st;
  if ((st = upload(h, h->z_ooff, ooff.data(), (size_t)(n + 1) * 8))) return st;
  if ((st = ensure(h, h->z_out, obytes + 64))) return st;
  if ((st = ensure(h, h->z_stat, (size_t)n * 4))) return st;
  if ((st = ensure(h, h->z_comp, cbytes + 16))) return st;
  if (expect_crc) {        // the gzip trailer's CRC-32 of every member, computed where the bytes are
    if (!h->z_crctab.p) {
      std::vector<uint32_t> t(256 + 17 * 32);
      vsv_crc32_tables(t.data());
      if ((st = upload(h, h->z_crctab, t.data(), t.size() * 4))) return st;
    }
    if ((st = ensure(h, h->z_crc, (size_t)n * 4))) return st;
  }
  // The members go up in a few slices and every slice is decoded as soon as it has arrived, on the handle's auxiliary streams:
  // the copy of slice k + 1 (pageable memory: the host thread is inside the copy) runs under the decode of slice k. A slice's
  // launch lasts as long as its slowest member (milliseconds of serial decode): on ONE stream those latencies add up (39 -> 71 ms
  // on a 9 k-member file), on two streams they overlap. What is left is the decode itself: 28.7 ms of kernel time for 9 124
  // members whatever the slicing.
  static const int slices_env = vsv_dbg_env("VSV_INFLATE_SLICES") ? atoi(vsv_dbg_env("VSV_INFLATE_SLICES")) : 0;      // timing experiments
  int n_slices = slices_env > 0 ? slices_env : (int)(n / 2048);       // 9 k members: 1 slice 38.6 ms, 2: 33.7, 4: 31.3, 8: 34.5
  if (n_slices < 1 || (n_slices > 1 && !have_aux(h, true))) n_slices = 1;
  if (n_slices > (slices_env > 0 ? 8 : 4)) n_slices = slices_env > 0 ? 8 : 4;
  std::vector<hipEvent_t> evs;
  auto drop_events = [&]() { for (hipEvent_t e : evs) (void)hipEventDestroy(e); evs.clear(); };
  hipStream_t lanes[2] = {n_slices > 1 ? h->aux : h->stream, n_slices > 1 ? h->aux2 : h->stream};
  hipError_t herr = hipSuccess;
  for (int c = 0; c < n_slices && herr == hipSuccess; ++c) {
    const int64_t m0 = n * c / n_slices, m1 = n * (c + 1) / n_slices;
    if (m1 <= m0) continue;
    const size_t b0 = (size_t)coff[m0], b1 = (size_t)coff[m1];
    if (b1 > b0) herr = hipMemcpyAsync((uint8_t*)h->z_comp.p + b0, comp + comp_off[0] + b0, b1 - b0, hipMemcpyHostToDevice, h->stream);
    hipStream_t ls = lanes[c & 1];
    if (herr == hipSuccess && ls != h->stream) {
      hipEvent_t e;
      herr = hipEventCreateWithFlags(&e, hipEventDisableTiming);
      if (herr == hipSuccess) { evs.push_back(e); herr = hipEventRecord(e, h->stream); }
      if (herr == hipSuccess) herr = hipStreamWaitEvent(ls, e, 0);
    }
    if (herr != hipSuccess) break;
    vsv_launch_bgzf_inflate(ls, (const uint8_t*)h->z_comp.p, (const uint64_t*)h->z_coff.p + m0, (const uint64_t*)h->z_ooff.p + m0, m1 - m0,
                            (uint8_t*)h->z_out.p, (int32_t*)h->z_stat.p + m0);
    if (expect_crc)
      vsv_launch_bgzf_crc32(ls, (const uint8_t*)h->z_out.p, (const uint64_t*)h->z_ooff.p + m0, m1 - m0, (const uint32_t*)h->z_crctab.p, (uint32_t*)h->z_crc.p + m0);
    herr = hipGetLastError();
  }
  for (int k = 0; k < 2 && herr == hipSuccess && n_slices > 1; ++k) {       // the handle's stream continues when both lanes are done
    hipEvent_t e;
    herr = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (herr == hipSuccess) { evs.push_back(e); herr = hipEventRecord(e, lanes[k]); }
    if (herr == hipSuccess) herr = hipStreamWaitEvent(h->stream, e, 0);
  }
  if (herr != hipSuccess) {
    (void)hipDeviceSynchronize();
    drop_events();
    return fail(h, VSV_E_HIP, std::string("bgzf inflate: ") + hipGetErrorString(herr));
  }
  std::vector<uint32_t> crc;
  if (expect_crc) {
    crc.resize((size_t)n);
    hipError_t ce = hipMemcpyAsync(crc.data(), h->z_crc.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream);
    if (ce != hipSuccess) { (void)hipDeviceSynchronize(); drop_events(); return fail(h, VSV_E_HIP, std::string("crc copy: ") + hipGetErrorString(ce)); }
  }
  std::vector<int32_t> stat((size_t)n);
  {
    hipError_t se = hipMemcpyAsync(stat.data(), h->z_stat.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream);
    if (se == hipSuccess) se = hipStreamSynchronize(h->stream);
    drop_events();
    if (se != hipSuccess) return fail(h, VSV_E_HIP, std::string("bgzf inflate: ") + hipGetErrorString(se));
  }
  for (int64_t i = 0; i < n; ++i)
    if (stat[i]) { h->last_count = i; char m[96]; snprintf(m, sizeof m, "BGZF member %lld is not a valid deflate stream (code %d)", (long long)i, stat[i]); return fail(h, VSV_E_INVALID, m); }
  if (expect_crc)
    for (int64_t i = 0; i < n; ++i)
      if (crc[i] != expect_crc[i]) { h->last_count = i; char m[96]; snprintf(m, sizeof m, "BGZF member %lld fails its CRC-32", (long long)i); return fail(h, VSV_E_INVALID, m); }
  return 0;
}

// Page-locked host buffer of at least `bytes` (kept by the handle, grown geometrically): the device reader stages a file's compressed
// bytes there, so that their upload is an asynchronous DMA at PCIe speed instead of a copy through the runtime's own staging.
// nullptr if it cannot be had (the caller then hands over pageable memory).
void* vsv_internal_pinned(vsv_handle* h, uint64_t bytes) {
  if (!h || bytes == 0) return nullptr;
  if (h->pin_bytes >= bytes) return h->pin_buf;
  if (hipSetDevice(h->device) != hipSuccess) return nullptr;
  if (h->pin_buf) { (void)hipHostFree(h->pin_buf); h->pin_buf = nullptr; h->pin_bytes = 0; }
  size_t nb = (size_t)bytes + (size_t)bytes / 4 + 4096;
  if (hipHostMalloc(&h->pin_buf, nb) != hipSuccess) { (void)hipGetLastError(); h->pin_buf = nullptr; return nullptr; }
  h->pin_bytes = nb;
  return h->pin_buf;
}
int vsv_bam_device_want_seq(vsv_handle* h, int want) {
  if (!h) return VSV_E_INVALID;
  h->want_seq = want != 0;
  return 0;
}
int vsv_bam_device_seq_slices(vsv_handle* h, const uint32_t* rec, const uint32_t* start, const uint32_t* len, const uint8_t* rev, int64_t n,
                              const uint64_t* out_off, uint8_t* out, int64_t out_bytes) {
  if (!h || n < 0 || out_bytes < 0 || (n > 0 && (!rec || !start || !len || !rev || !out_off || !out))) return VSV_E_INVALID;
  if (h->seq_records < 0) return fail(h, VSV_E_INVALID, "no sequences on the device: vsv_bam_device_want_seq before the load");
  if (n == 0) return 0;
  HIPCHK(h, hipSetDevice(h->device));
  // the slices must lie inside their records and inside the output: checked here, on the host copies of the (small) request arrays
  std::vector<uint32_t> lseq((size_t)h->seq_records);
  HIPCHK(h, hipMemcpyAsync(lseq.data(), h->o_lseq.p, lseq.size() * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  for (int64_t i = 0; i < n; ++i) {
    if ((int64_t)rec[i] >= h->seq_records) return fail(h, VSV_E_INVALID, "sequence slice of a record that was not loaded");
    if ((uint64_t)start[i] + len[i] > lseq[rec[i]]) return fail(h, VSV_E_INVALID, "sequence slice outside its record");
    if (out_off[i] + len[i] > (uint64_t)out_bytes) return fail(h, VSV_E_INVALID, "sequence slice outside the output buffer");
  }
  int st;
  if ((st = upload(h, h->q_rec, rec, (size_t)n * 4))) return st;
  if ((st = upload(h, h->q_start, start, (size_t)n * 4))) return st;
  if ((st = upload(h, h->q_len, len, (size_t)n * 4))) return st;
  if ((st = upload(h, h->q_rev, rev, (size_t)n))) return st;
  if ((st = upload(h, h->q_ooff, out_off, (size_t)n * 8))) return st;
  if ((st = ensure(h, h->q_out, (size_t)out_bytes + 16))) return st;
  vsv_bamdev_seq_slices(h->stream, (const uint8_t*)h->o_seq.p, (const uint64_t*)h->o_recseq.p, (const uint32_t*)h->o_lseq.p, (const uint32_t*)h->q_rec.p,
                        (const uint32_t*)h->q_start.p, (const uint32_t*)h->q_len.p, (const uint8_t*)h->q_rev.p, n, (const uint64_t*)h->q_ooff.p,
                        (uint8_t*)h->q_out.p);
  HIPCHK(h, hipGetLastError());
  if (out_bytes) HIPCHK(h, hipMemcpyAsync(out, h->q_out.p, (size_t)out_bytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}
int vsv_bam_device_want_sa(vsv_handle* h, int want) {
  if (!h) return VSV_E_INVALID;
  h->want_sa = want != 0;
  return 0;
}
const char* vsv_bam_device_sa_tags(vsv_handle* h, int64_t* len) {
  if (len) *len = h ? (int64_t)h->sa_text.size() : 0;
  return h ? h->sa_text.c_str() : "";
}

int vsv_bgzf_set_expected_crc(vsv_handle* h, const uint32_t* crc, int64_t n_members) {
  if (!h || n_members < 0 || (n_members > 0 && !crc)) return VSV_E_INVALID;
  h->expect_crc = n_members ? crc : nullptr;
  h->expect_crc_n = n_members;
  return 0;
}

int vsv_bgzf_inflate(vsv_handle* h, const uint8_t* comp, const uint64_t* comp_off, const uint32_t* isize, int64_t n, uint8_t* out) {
  if (!h) return VSV_E_INVALID;
  if (n < 0) return fail(h, VSV_E_INVALID, "negative member count");
  if (n == 0) return 0;
  if (!comp || !comp_off || !isize || !out) return fail(h, VSV_E_INVALID, "member arrays are NULL");
  HIPCHK(h, hipSetDevice(h->device));
  std::vector<uint64_t> ooff;
  const int rc = inflate_to_device(h, comp, comp_off, isize, n, ooff, (h->expect_crc && h->expect_crc_n == n) ? h->expect_crc : nullptr);
  if (rc) return rc;
  if (ooff[n]) HIPCHK(h, hipMemcpyAsync(out, h->z_out.p, (size_t)ooff[n], hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}

// grow a device buffer, keeping its first `used` bytes
static int ensure_keep(vsv_handle* h, DevBuf& b, size_t need, size_t used) {
  if (b.bytes >= need) return 0;
  size_t nb = b.bytes * 2 > need ? b.bytes * 2 : need;
  if (nb < 4096) nb = 4096;
  void* p = nullptr;
  HIPCHK(h, hipMalloc(&p, nb));
  if (used && b.p) HIPCHK(h, hipMemcpyAsync(p, b.p, used, hipMemcpyDeviceToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (b.p) HIPCHK(h, hipFree(b.p));
  b.p = p; b.bytes = nb;
  return 0;
}

int vsv_bam_parse_device(vsv_handle* h, const uint8_t* comp, const uint64_t* comp_off, const uint32_t* isize, int64_t n_members, uint64_t first_record,
                         int32_t n_ref, int32_t tid, vsv_records* out, const char** names, int64_t* names_len, const uint32_t** l_seq_dev,
                         const uint32_t** sam_flags_dev) {
  if (!h || !out) return VSV_E_INVALID;
  if (n_members <= 0 || !comp || !comp_off || !isize) return fail(h, VSV_E_INVALID, "no BGZF members");
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = h->stream;
  // The file is processed in windows of members (the whole file when it is small): a window is inflated, its records are
  // chained / parsed / appended to the output arrays, and the next window starts at the member that holds the first record
  // the previous one could not complete. VSV_BAM_WINDOW overrides the window size (tests use tiny windows).
  const int64_t window_members = vsv_dbg_env("VSV_BAM_WINDOW") && atoll(vsv_dbg_env("VSV_BAM_WINDOW")) > 0 ? atoll(vsv_dbg_env("VSV_BAM_WINDOW")) : 32768;
  std::vector<uint64_t> ginf((size_t)n_members + 1, 0);
  for (int64_t i = 0; i < n_members; ++i) ginf[i + 1] = ginf[i] + isize[i];
  const uint64_t stream_total = ginf[n_members];
  if (first_record > stream_total) return fail(h, VSV_E_INVALID, "header longer than the stream");
  memset(out, 0, sizeof *out);
  out->on_device = 1;
  h->names_len = 0;
  if (names) *names = "";
  if (names_len) *names_len = 0;
  int rc;
#define DEVMEM(expr) do { if ((rc = (expr))) return rc == VSV_E_HIP ? fail(h, VSV_E_CAPACITY, "device memory exhausted in the device reader: use the host reader") : rc; } while (0)
  DEVMEM(ensure(h, h->p_err, 256)); DEVMEM(ensure(h, h->p_tot, 256)); DEVMEM(ensure(h, h->o_n, 256));
  HIPCHK(h, hipMemsetAsync(h->p_err.p, 0, 4, st));
  uint64_t carry = first_record, K0 = 0, C0 = 0, N0 = 0, S0 = 0, Q0 = 0;
  h->sa_text.clear();
  h->seq_records = -1;
  const bool timing = vsv_dbg_env("VSV_BAM_TIMING") != nullptr;
  double t_inf = 0, t_chain = 0, t_rec = 0, t_names = 0, t_qid = 0;
  int n_windows = 0, n_rounds = 0;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tq = now();
  auto lap = [&](double& acc) { const double t = now(); acc += t - tq; tq = t; };
  std::vector<uint64_t> moff, spec, land, base;
  std::vector<uint32_t> cnt;
  std::vector<uint8_t> active;
  while (carry < stream_total) {
    const int64_t ma = (int64_t)(std::upper_bound(ginf.begin(), ginf.end(), carry) - ginf.begin()) - 1;
    const int64_t mb = ma + window_members < n_members ? ma + window_members : n_members;
    const bool last_window = mb == n_members;
    rc = inflate_to_device(h, comp, comp_off + ma, isize + ma, mb - ma, moff, (h->expect_crc && h->expect_crc_n == n_members) ? h->expect_crc + ma : nullptr);
    if (rc == VSV_E_HIP) return fail(h, VSV_E_CAPACITY, "device memory exhausted in the device reader: use the host reader");
    if (rc) return rc;
    lap(t_inf); ++n_windows;
    const uint64_t total = moff[mb - ma], first_rel = carry - ginf[ma];
    const uint8_t* s = (const uint8_t*)h->z_out.p;
    const uint64_t* d_moff = (const uint64_t*)h->z_ooff.p;
    const size_t nm = (size_t)(mb - ma);
    DEVMEM(ensure(h, h->p_spec, nm * 8)); DEVMEM(ensure(h, h->p_cnt, nm * 4)); DEVMEM(ensure(h, h->p_land, nm * 8)); DEVMEM(ensure(h, h->p_base, nm * 8));
    // ---- record-start chain: speculate per member, walk, verify on the host from the exactly known first record -----------
    vsv_bamdev_speculate(st, s, d_moff, (int64_t)nm, first_rel, n_ref, (uint64_t*)h->p_spec.p);
    spec.assign(nm, 0); land.assign(nm, 0); base.assign(nm, 0); cnt.assign(nm, 0); active.assign(nm, 0);
    HIPCHK(h, hipMemcpyAsync(spec.data(), h->p_spec.p, nm * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    uint64_t n_w = 0, next_rel = first_rel;
    for (int round = 0;; ++round) {
      if (round > 1024) return fail(h, VSV_E_INVALID, "record chain does not settle (not a BAM stream?): use the host reader");
      HIPCHK(h, hipMemcpyAsync(h->p_spec.p, spec.data(), nm * 8, hipMemcpyHostToDevice, st));
      vsv_bamdev_chain(st, false, s, d_moff, (int64_t)nm, (const uint64_t*)h->p_spec.p, (uint32_t*)h->p_cnt.p, (uint64_t*)h->p_land.p, nullptr, nullptr);
      HIPCHK(h, hipMemcpyAsync(cnt.data(), h->p_cnt.p, nm * 4, hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipMemcpyAsync(land.data(), h->p_land.p, nm * 8, hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
      std::fill(active.begin(), active.end(), 0);
      bool patched = false;
      uint64_t expect = first_rel;
      n_w = 0;
      while (expect < total) {
        const size_t m = (size_t)(std::upper_bound(moff.begin(), moff.end(), expect) - moff.begin()) - 1;   // member holding `expect`
        if (spec[m] != expect) { spec[m] = expect; patched = true; break; }                                  // proven start: patch and rewalk
        if (land[m] == 0xFFFFFFFFFFFFFFFFull) return fail(h, VSV_E_INVALID, "malformed BAM record chain");
        active[m] = 1; base[m] = n_w; n_w += cnt[m];
        const bool stopped = land[m] < moff[m + 1];            // a record that is not complete inside this window starts at land[m]
        expect = land[m];
        if (stopped) break;
      }
      ++n_rounds;
      if (!patched) { next_rel = expect; break; }
    }
    lap(t_chain);
    if (last_window && next_rel != total) return fail(h, VSV_E_INVALID, "BAM stream ends inside a record");
    if (n_w == 0 && !last_window) return fail(h, VSV_E_CAPACITY, "a record is larger than the device reader's window: use the host reader");
    if (K0 + n_w > 0x7FFFFFF0ull) return fail(h, VSV_E_CAPACITY, "too many records for the device reader: use the host reader");
    if (n_w > 0) {
      for (size_t m = 0; m < nm; ++m) if (!active[m]) spec[m] = 0xFFFFFFFFFFFFFFFFull;      // speculated starts the chain never reached
      const int64_t n = (int64_t)n_w;
      const size_t N = (size_t)n;
      HIPCHK(h, hipMemcpyAsync(h->p_spec.p, spec.data(), nm * 8, hipMemcpyHostToDevice, st));
      HIPCHK(h, hipMemcpyAsync(h->p_base.p, base.data(), nm * 8, hipMemcpyHostToDevice, st));
      DEVMEM(ensure(h, h->p_recoff, N * 8)); DEVMEM(ensure(h, h->p_pos, N * 4)); DEVMEM(ensure(h, h->p_tid, N * 4)); DEVMEM(ensure(h, h->p_mapq, N));
      DEVMEM(ensure(h, h->p_flag, N)); DEVMEM(ensure(h, h->p_lseq, N * 4)); DEVMEM(ensure(h, h->p_sflag, N * 4)); DEVMEM(ensure(h, h->p_ncig, N * 4));
      DEVMEM(ensure(h, h->p_cgsrc, N * 8)); DEVMEM(ensure(h, h->p_hash, N * 8)); DEVMEM(ensure(h, h->p_keep, N * 4 + 16)); DEVMEM(ensure(h, h->p_kidx, N * 4 + 16));
      DEVMEM(ensure(h, h->p_cigoff, N * 8)); DEVMEM(ensure(h, h->p_sums, (N / 2048 + 4) * 8)); DEVMEM(ensure(h, h->o_recoff, N * 8));
      DEVMEM(ensure(h, h->o_nlen, N * 4 + 16)); DEVMEM(ensure(h, h->o_noff, N * 4 + 16));
      vsv_bamdev_chain(st, true, s, d_moff, (int64_t)nm, (const uint64_t*)h->p_spec.p, nullptr, nullptr, (const uint64_t*)h->p_base.p, (uint64_t*)h->p_recoff.p);
      vsv_bamdev_fields(st, s, (const uint64_t*)h->p_recoff.p, n, tid, (int32_t*)h->p_pos.p, (int32_t*)h->p_tid.p, (uint8_t*)h->p_mapq.p, (uint8_t*)h->p_flag.p,
                        (uint32_t*)h->p_lseq.p, (uint32_t*)h->p_sflag.p, (uint32_t*)h->p_ncig.p, (uint64_t*)h->p_cgsrc.p, (uint64_t*)h->p_hash.p,
                        (uint32_t*)h->p_keep.p, (uint32_t*)h->p_err.p);
      vsv_scan_u32_exclusive(st, (const uint32_t*)h->p_keep.p, (int)n, (uint32_t*)h->p_kidx.p, (uint32_t*)h->p_sums.p);
      vsv_bamdev_scan64(st, (const uint32_t*)h->p_ncig.p, (const uint32_t*)h->p_keep.p, n, (uint64_t*)h->p_sums.p, (uint64_t*)h->p_cigoff.p, (uint64_t*)h->p_tot.p);
      uint32_t last_k = 0, last_keep = 0, err = 0;
      uint64_t ops_w = 0;
      HIPCHK(h, hipMemcpyAsync(&last_k, (uint32_t*)h->p_kidx.p + (n - 1), 4, hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipMemcpyAsync(&last_keep, (uint32_t*)h->p_keep.p + (n - 1), 4, hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipMemcpyAsync(&ops_w, h->p_tot.p, 8, hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipMemcpyAsync(&err, h->p_err.p, 4, hipMemcpyDeviceToHost, st));
      HIPCHK(h, hipStreamSynchronize(st));
      if (err & 1u) return fail(h, VSV_E_INVALID, "unknown BAM tag type");
      if (err & 4u) return fail(h, VSV_E_INVALID, "malformed BAM record (a field runs past its record): use the host reader");
      const uint64_t nk = (uint64_t)last_k + last_keep;
      if (nk > 0) {
        const size_t K = (size_t)(K0 + nk);
        DEVMEM(ensure_keep(h, h->o_pos, K * 4, (size_t)K0 * 4)); DEVMEM(ensure_keep(h, h->o_tid, K * 4, (size_t)K0 * 4));
        DEVMEM(ensure_keep(h, h->o_cigoff, (K + 1) * 8, (size_t)K0 * 8)); DEVMEM(ensure_keep(h, h->o_mapq, K, (size_t)K0)); DEVMEM(ensure_keep(h, h->o_flag, K, (size_t)K0));
        DEVMEM(ensure_keep(h, h->o_cigar, (size_t)(C0 + ops_w) * 4 + 16, (size_t)C0 * 4)); DEVMEM(ensure_keep(h, h->o_lseq, K * 4, (size_t)K0 * 4));
        DEVMEM(ensure_keep(h, h->o_sflag, K * 4, (size_t)K0 * 4)); DEVMEM(ensure_keep(h, h->o_hash, K * 8, (size_t)K0 * 8));
        DEVMEM(ensure_keep(h, h->o_nmoff, K * 8, (size_t)K0 * 8)); DEVMEM(ensure_keep(h, h->o_nmlen, K * 4, (size_t)K0 * 4));
        vsv_bamdev_emit(st, s, n, (const uint32_t*)h->p_keep.p, (const uint32_t*)h->p_kidx.p, (const int32_t*)h->p_pos.p, (const int32_t*)h->p_tid.p,
                        (const uint8_t*)h->p_mapq.p, (const uint8_t*)h->p_flag.p, (const uint32_t*)h->p_lseq.p, (const uint32_t*)h->p_sflag.p,
                        (const uint32_t*)h->p_ncig.p, (const uint64_t*)h->p_cgsrc.p, (const uint64_t*)h->p_hash.p, (const uint64_t*)h->p_cigoff.p,
                        (const uint64_t*)h->p_recoff.p, (int32_t*)h->o_pos.p, (int32_t*)h->o_tid.p, (uint8_t*)h->o_mapq.p, (uint8_t*)h->o_flag.p,
                        (uint32_t*)h->o_lseq.p, (uint32_t*)h->o_sflag.p, (uint64_t*)h->o_cigoff.p, (uint32_t*)h->o_cigar.p, (uint64_t*)h->o_hash.p,
                        (uint64_t*)h->o_recoff.p, K0, C0, ops_w);
        lap(t_rec);
        // names of the kept records into the compact store (the inflated window is gone when the query ids are assigned)
        vsv_bamdev_win_name_lens(st, s, (const uint64_t*)h->o_recoff.p, (int64_t)nk, (uint32_t*)h->o_nlen.p);
        vsv_scan_u32_exclusive(st, (const uint32_t*)h->o_nlen.p, (int)nk, (uint32_t*)h->o_noff.p, (uint32_t*)h->p_sums.p);
        uint32_t lo_ = 0, ll_ = 0;
        HIPCHK(h, hipMemcpyAsync(&lo_, (uint32_t*)h->o_noff.p + (nk - 1), 4, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipMemcpyAsync(&ll_, (uint32_t*)h->o_nlen.p + (nk - 1), 4, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
        const uint64_t nb_w = (uint64_t)lo_ + ll_;
        DEVMEM(ensure_keep(h, h->o_names, (size_t)(N0 + nb_w) + 16, (size_t)N0));
        vsv_bamdev_win_name_store(st, s, (const uint64_t*)h->o_recoff.p, (const uint32_t*)h->o_noff.p, (int64_t)nk, K0, N0, (uint8_t*)h->o_names.p,
                                  (uint64_t*)h->o_nmoff.p, (uint32_t*)h->o_nmlen.p);
        HIPCHK(h, hipStreamSynchronize(st));
        if (h->want_sa) {   // SA:Z texts of the kept records, '\n' after each, appended in record order
          DEVMEM(ensure(h, h->o_saoff, (size_t)nk * 8)); DEVMEM(ensure(h, h->o_salen, (size_t)nk * 4 + 16)); DEVMEM(ensure(h, h->o_saloc, (size_t)nk * 4 + 16));
          vsv_bamdev_win_sa_find(st, s, (const uint64_t*)h->o_recoff.p, (int64_t)nk, (uint64_t*)h->o_saoff.p, (uint32_t*)h->o_salen.p);
          vsv_scan_u32_exclusive(st, (const uint32_t*)h->o_salen.p, (int)nk, (uint32_t*)h->o_saloc.p, (uint32_t*)h->p_sums.p);
          uint32_t so_ = 0, sl_ = 0;
          HIPCHK(h, hipMemcpyAsync(&so_, (uint32_t*)h->o_saloc.p + (nk - 1), 4, hipMemcpyDeviceToHost, st));
          HIPCHK(h, hipMemcpyAsync(&sl_, (uint32_t*)h->o_salen.p + (nk - 1), 4, hipMemcpyDeviceToHost, st));
          HIPCHK(h, hipStreamSynchronize(st));
          const uint64_t sb_w = (uint64_t)so_ + sl_;
          DEVMEM(ensure_keep(h, h->o_sa, (size_t)(S0 + sb_w) + 16, (size_t)S0));
          vsv_bamdev_win_sa_store(st, s, (const uint64_t*)h->o_saoff.p, (const uint32_t*)h->o_salen.p, (const uint32_t*)h->o_saloc.p, (int64_t)nk, S0,
                                  (uint8_t*)h->o_sa.p);
          HIPCHK(h, hipStreamSynchronize(st));
          S0 += sb_w;
        }
        if (h->want_seq) {  // packed SEQ of the kept records, appended in record order; stays on the device
          DEVMEM(ensure(h, h->o_sqoff, (size_t)nk * 8)); DEVMEM(ensure(h, h->o_sqlen, (size_t)nk * 4 + 16)); DEVMEM(ensure(h, h->o_sqloc, (size_t)nk * 4 + 16));
          DEVMEM(ensure_keep(h, h->o_recseq, K * 8, (size_t)K0 * 8));
          vsv_bamdev_win_seq_find(st, s, (const uint64_t*)h->o_recoff.p, (int64_t)nk, (uint64_t*)h->o_sqoff.p, (uint32_t*)h->o_sqlen.p);
          vsv_scan_u32_exclusive(st, (const uint32_t*)h->o_sqlen.p, (int)nk, (uint32_t*)h->o_sqloc.p, (uint32_t*)h->p_sums.p);
          uint32_t qo_ = 0, ql_ = 0;
          HIPCHK(h, hipMemcpyAsync(&qo_, (uint32_t*)h->o_sqloc.p + (nk - 1), 4, hipMemcpyDeviceToHost, st));
          HIPCHK(h, hipMemcpyAsync(&ql_, (uint32_t*)h->o_sqlen.p + (nk - 1), 4, hipMemcpyDeviceToHost, st));
          HIPCHK(h, hipStreamSynchronize(st));
          const uint64_t qb_w = (uint64_t)qo_ + ql_;
          DEVMEM(ensure_keep(h, h->o_seq, (size_t)(Q0 + qb_w) + 16, (size_t)Q0));
          vsv_bamdev_win_seq_store(st, s, (const uint64_t*)h->o_sqoff.p, (const uint32_t*)h->o_sqlen.p, (const uint32_t*)h->o_sqloc.p, (int64_t)nk, K0, Q0,
                                   (uint8_t*)h->o_seq.p, (uint64_t*)h->o_recseq.p);
          HIPCHK(h, hipStreamSynchronize(st));
          Q0 += qb_w;
        }
        K0 += nk; C0 += ops_w; N0 += nb_w;
        lap(t_names);
      }
    }
    if (next_rel == first_rel && n_w == 0 && last_window) break;
    carry = ginf[ma] + next_rel;
  }
  const int64_t nk = (int64_t)K0;
  out->n_records = nk; out->n_ops = (int64_t)C0; out->n_tids = n_ref;
  if (h->want_seq) h->seq_records = nk;
  if (nk == 0) return 0;
  // the name table's offsets are 32-bit scans (per window the inflated bytes bound them; over the whole file N0 does)
  if (N0 > 0xFFFFFFF0ull) return fail(h, VSV_E_CAPACITY, "more than 4 GiB of query names for the device reader: use the host reader");
  if (h->want_sa && S0 > 0) {      // one text per record, '\n' between them (the last separator is dropped)
    h->sa_text.resize((size_t)S0);
    HIPCHK(h, hipMemcpyAsync(&h->sa_text[0], h->o_sa.p, (size_t)S0, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    h->sa_text.pop_back();
  }
  const size_t K = (size_t)nk;
  HIPCHK(h, hipMemcpyAsync((uint64_t*)h->o_cigoff.p + K, &C0, 8, hipMemcpyHostToDevice, st));
  // ---- dense first-appearance query ids: stable sort of (name hash, record), group heads, ranks of the first occurrences ---
  DEVMEM(ensure(h, h->o_qid, K * 4)); DEVMEM(ensure(h, h->o_first, K * 4 + 16)); DEVMEM(ensure(h, h->o_rank, K * 4 + 16));
  DEVMEM(ensure(h, h->o_nlen, K * 4 + 16)); DEVMEM(ensure(h, h->o_noff, K * 4 + 16)); DEVMEM(ensure(h, h->p_sums, (K / 2048 + 4) * 8));
  {  // sort scratch (key / idx / alt pair / histograms) for nk rows; never below the default signature capacity of a run
    const int64_t floor_cap = h->cap_sigs > 0 ? h->cap_sigs : (1 << 22);
    DEVMEM(reserve(h, 1, 1, nk > floor_cap ? nk : floor_cap));
  }
  { int rs = reset_run_state(h); if (rs) return rs; }
  const uint32_t nk32 = (uint32_t)nk;
  HIPCHK(h, hipMemcpyAsync(h->o_n.p, &nk32, 4, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(h->key.p, h->o_hash.p, K * 8, hipMemcpyDeviceToDevice, st));
  vsv_bamdev_iota(st, (uint32_t*)h->idx.p, nk);
  SortWork sw = sort_work_plain(h);
  sw.small_tiles = nk <= 128 * 4096;
  const SortResult sr = vsv_radix_sort_pairs(st, (uint64_t*)h->key.p, (uint32_t*)h->idx.p, sw.key_alt, sw.val_alt, (const uint32_t*)h->o_n.p, nk, 64, sw);
  HIPCHK(h, hipMemsetAsync(h->o_first.p, 0, K * 4, st));
  vsv_bamdev_mark_first(st, sr.key, sr.val, nk, (uint32_t*)h->o_first.p);
  vsv_scan_u32_exclusive(st, (const uint32_t*)h->o_first.p, (int)nk, (uint32_t*)h->o_rank.p, (uint32_t*)h->p_sums.p);
  vsv_bamdev_assign(st, sr.key, sr.val, nk, (const uint32_t*)h->o_rank.p, (const uint8_t*)h->o_names.p, (const uint64_t*)h->o_nmoff.p,
                    (const uint32_t*)h->o_nmlen.p, (uint32_t*)h->o_qid.p, (uint32_t*)h->p_err.p);
  vsv_bamdev_name_lens(st, (const uint32_t*)h->o_nmlen.p, (const uint32_t*)h->o_first.p, nk, (uint32_t*)h->o_nlen.p);
  vsv_scan_u32_exclusive(st, (const uint32_t*)h->o_nlen.p, (int)nk, (uint32_t*)h->o_noff.p, (uint32_t*)h->p_sums.p);
  uint32_t lr = 0, lf = 0, lo = 0, ll = 0, err = 0;
  HIPCHK(h, hipMemcpyAsync(&lr, (uint32_t*)h->o_rank.p + (nk - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(&lf, (uint32_t*)h->o_first.p + (nk - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(&lo, (uint32_t*)h->o_noff.p + (nk - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(&ll, (uint32_t*)h->o_nlen.p + (nk - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(&err, h->p_err.p, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  if (err & 2u) return fail(h, VSV_E_INVALID, "64-bit name hash collision: use the host reader");
  const size_t blob_bytes = (size_t)lo + ll;
  DEVMEM(ensure(h, h->o_blob, blob_bytes + 16));
  vsv_bamdev_name_copy(st, (const uint8_t*)h->o_names.p, (const uint64_t*)h->o_nmoff.p, (const uint32_t*)h->o_nmlen.p, (const uint32_t*)h->o_first.p,
                       (const uint32_t*)h->o_noff.p, nk, (uint8_t*)h->o_blob.p);
  if (h->names_cap < blob_bytes + 1) {
    if (h->names_pin) { (void)hipHostFree(h->names_pin); h->names_pin = nullptr; h->names_cap = 0; }
    const size_t nb = blob_bytes + blob_bytes / 4 + 4096;
    HIPCHK(h, hipHostMalloc((void**)&h->names_pin, nb));
    h->names_cap = nb;
  }
  if (blob_bytes) HIPCHK(h, hipMemcpyAsync(h->names_pin, h->o_blob.p, blob_bytes, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  h->names_len = blob_bytes ? blob_bytes - 1 : 0;                     // the separator after the last name is dropped
  h->names_pin[h->names_len] = 0;
  lap(t_qid);
  if (timing)
    fprintf(stderr, "[vsv_bam_parse_device] %d windows: upload+inflate %.1f ms, record chain %.1f ms (%d rounds), fields+emit %.1f ms, names %.1f ms, query ids + name table %.1f ms\n",
            n_windows, t_inf * 1e3, t_chain * 1e3, n_rounds, t_rec * 1e3, t_names * 1e3, t_qid * 1e3);
#undef DEVMEM
  out->pos = (const int32_t*)h->o_pos.p; out->tid = (const int32_t*)h->o_tid.p; out->qid = (const uint32_t*)h->o_qid.p;
  out->cigar_off = (const uint64_t*)h->o_cigoff.p; out->mapq = (const uint8_t*)h->o_mapq.p; out->flag = (const uint8_t*)h->o_flag.p;
  out->cigar = (const uint32_t*)h->o_cigar.p;
  out->n_qids = (int32_t)(lr + lf);
  if (names) *names = h->names_pin;
  if (names_len) *names_len = (int64_t)h->names_len;
  if (l_seq_dev) *l_seq_dev = (const uint32_t*)h->o_lseq.p;
  if (sam_flags_dev) *sam_flags_dev = (const uint32_t*)h->o_sflag.p;
  h->stage_done = 0;
  return 0;
}

int vsv_gt_support(vsv_handle* h, const int32_t* var_pos, const int32_t* var_svlen, const int32_t* blk_lo, const int32_t* blk_hi, int64_t nv,
                   const int32_t* sig_pos, const int32_t* sig_svlen, const int32_t* sig_cnt, int64_t ns, double max_shift_ratio, double min_size_sim,
                   int64_t* sum, int32_t* lo, int32_t* hi) {
  if (!h) return VSV_E_INVALID;
  if (nv < 0 || ns < 0) return fail(h, VSV_E_INVALID, "negative count");
  if (nv == 0) return 0;
  if (!var_pos || !var_svlen || !blk_lo || !blk_hi || !sum || !lo || !hi || (ns > 0 && (!sig_pos || !sig_svlen || !sig_cnt))) return fail(h, VSV_E_INVALID, "arrays are NULL");
  for (int64_t i = 0; i < nv; ++i) {
    if (blk_lo[i] < 0 || blk_hi[i] < blk_lo[i] || blk_hi[i] > ns) return fail(h, VSV_E_INVALID, "bad chromosome block");
  }
  HIPCHK(h, hipSetDevice(h->device));
  int st;
  if ((st = upload(h, h->j_cpos, var_pos, (size_t)nv * 4)) || (st = upload(h, h->j_clen, var_svlen, (size_t)nv * 4)) ||
      (st = upload(h, h->g_qs, blk_lo, (size_t)nv * 4)) || (st = upload(h, h->g_qe, blk_hi, (size_t)nv * 4)) ||
      (st = upload(h, h->j_spos, sig_pos, (size_t)ns * 4)) || (st = upload(h, h->j_slen, sig_svlen, (size_t)ns * 4)) ||
      (st = upload(h, h->j_send, sig_cnt, (size_t)ns * 4)) || (st = ensure(h, h->j_out, (size_t)nv * 8)) || (st = ensure(h, h->g_rs, (size_t)nv * 4)) ||
      (st = ensure(h, h->g_re, (size_t)nv * 4))) return st;
  vsv_launch_gt_support(h->stream, (const int32_t*)h->j_cpos.p, (const int32_t*)h->j_clen.p, (const int32_t*)h->g_qs.p, (const int32_t*)h->g_qe.p, nv,
                        (const int32_t*)h->j_spos.p, (const int32_t*)h->j_slen.p, (const int32_t*)h->j_send.p, max_shift_ratio, min_size_sim,
                        (int64_t*)h->j_out.p, (int32_t*)h->g_rs.p, (int32_t*)h->g_re.p);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(sum, h->j_out.p, (size_t)nv * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(lo, h->g_rs.p, (size_t)nv * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(hi, h->g_re.p, (size_t)nv * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int vsv_span_count(vsv_handle* h, const vsv_records* recs, const int32_t* q_tid, const int32_t* q_a, const int32_t* q_b, int64_t nq, uint32_t* out) {
  if (!h || !recs) return VSV_E_INVALID;
  if (nq < 0) return fail(h, VSV_E_INVALID, "negative count");
  if (nq == 0) return 0;
  if (!q_tid || !q_a || !q_b || !out) return fail(h, VSV_E_INVALID, "query arrays are NULL");
  HIPCHK(h, hipSetDevice(h->device));
  int st = bind_records(h, recs);
  if (st) return st;
  if ((st = upload(h, h->j_cpos, q_tid, (size_t)nq * 4)) || (st = upload(h, h->j_clen, q_a, (size_t)nq * 4)) || (st = upload(h, h->j_spos, q_b, (size_t)nq * 4)) ||
      (st = ensure(h, h->j_out, (size_t)nq * 4)) || (st = ensure(h, h->j_slen, (size_t)(h->rv.n_records + 1) * 4)) || (st = ensure(h, h->j_err, 256))) return st;
  HIPCHK(h, hipMemsetAsync(h->j_err.p, 0, 8, h->stream));
  vsv_launch_span_count(h->stream, h->rv, (int32_t*)h->j_slen.p, (uint32_t*)h->j_err.p, (const int32_t*)h->j_cpos.p, (const int32_t*)h->j_clen.p,
                        (const int32_t*)h->j_spos.p, nq, (uint32_t*)h->j_out.p);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(out, h->j_out.p, (size_t)nq * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int vsv_copy_to_host(vsv_handle* h, void* dst, const void* src_device, int64_t bytes) {
  if (!h || bytes < 0 || (bytes > 0 && (!dst || !src_device))) return VSV_E_INVALID;
  if (bytes == 0) return 0;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(dst, src_device, (size_t)bytes, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return 0;
}

int vsv_default_redundancy_params(vsv_redundancy_params* p) {
  if (!p) return VSV_E_INVALID;
  p->dist_thresh = 500; p->dist_thresh_del = 3000; p->overlap_thresh = 0.0; p->size_sim_thresh = 0.5;    // RR:9-14
  p->size_sim_thresh_del = 0.1; p->seq_sim_thresh = 0.5;
  return 0;
}

int vsv_redundancy_pairs(vsv_handle* h, int is_del, const int32_t* pos, const int32_t* svlen, const uint8_t* seq, const uint64_t* seq_off,
                         int64_t n, const vsv_redundancy_params* p, uint32_t* pairs, int64_t cap, int64_t* n_pairs) {
  if (!h || !p || !n_pairs) return VSV_E_INVALID;
  *n_pairs = 0;
  if (n < 0 || n > 0x7FFFFFF0ll) return fail(h, VSV_E_INVALID, "bad call count");
  if (n == 0) return 0;
  if (!pos || !svlen || (!is_del && (!seq || !seq_off))) return fail(h, VSV_E_INVALID, "call arrays are NULL");
  for (int64_t i = 0; i < n; ++i) {
    if (i + 1 < n && pos[i] > pos[i + 1]) return fail(h, VSV_E_UNSORTED, "calls are not sorted by position");
    if (svlen[i] <= 0) return fail(h, VSV_E_ZERODIV, "call with |len(REF)-len(ALT)| == 0 (the reference divides by it, RR:88-90)");
  }
  if (!is_del) {
    if (seq_off[0] != 0) return fail(h, VSV_E_INVALID, "seq_off must start at 0");
    for (int64_t i = 0; i < n; ++i) if (seq_off[i + 1] < seq_off[i]) return fail(h, VSV_E_INVALID, "seq_off does not ascend");
    for (uint64_t k = 0; k < seq_off[n]; ++k) if (seq[k] > 15) return fail(h, VSV_E_INVALID, "sequence symbols must be coded 0..15");
  }
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = h->stream;
  int rc;
  if ((rc = upload(h, h->j_cpos, pos, (size_t)n * 4))) return rc;
  if ((rc = upload(h, h->j_clen, svlen, (size_t)n * 4))) return rc;
  if ((rc = ensure(h, h->j_spos, (size_t)n * 4 + 16))) return rc;          // counts
  if ((rc = ensure(h, h->j_slen, (size_t)n * 4 + 16))) return rc;          // offsets
  if ((rc = ensure(h, h->j_err, (size_t)(n / 2048 + 4) * 4 + 256))) return rc;   // scan scratch
  uint32_t *cnt = (uint32_t*)h->j_spos.p, *off = (uint32_t*)h->j_slen.p;
  const int64_t dist = is_del ? p->dist_thresh_del : p->dist_thresh;
  const double size_thr = is_del ? p->size_sim_thresh_del : p->size_sim_thresh;
  vsv_launch_rr_pairs(st, false, (const int32_t*)h->j_cpos.p, (const int32_t*)h->j_clen.p, n, is_del, dist, size_thr, p->overlap_thresh, cnt, off,
                      nullptr, 0);
  vsv_scan_u32_exclusive(st, cnt, (int)n, off, (uint32_t*)h->j_err.p);
  uint32_t last_off = 0, last_cnt = 0;
  HIPCHK(h, hipMemcpyAsync(&last_off, off + (n - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(&last_cnt, cnt + (n - 1), 4, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  const int64_t n_cand = (int64_t)last_off + last_cnt;
  if (n_cand == 0) return 0;
  if ((rc = ensure(h, h->j_out, (size_t)n_cand * 8))) return rc;
  vsv_launch_rr_pairs(st, true, (const int32_t*)h->j_cpos.p, (const int32_t*)h->j_clen.p, n, is_del, dist, size_thr, p->overlap_thresh, cnt, off,
                      (uint32_t*)h->j_out.p, (uint32_t)n_cand);
  std::vector<uint32_t> cand((size_t)n_cand * 2);
  HIPCHK(h, hipMemcpyAsync(cand.data(), h->j_out.p, (size_t)n_cand * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  std::vector<uint8_t> flag((size_t)n_cand, 1);
  if (!is_del) {
    // per-pair slice of the delta scratch: only pairs whose shorter string needs more than one super-block (> 4096 symbols)
    std::vector<uint64_t> hoff((size_t)n_cand, 0);
    uint64_t hwords = 1;
    for (int64_t k = 0; k < n_cand; ++k) {
      const uint64_t la = seq_off[cand[2 * k] + 1] - seq_off[cand[2 * k]], lb = seq_off[cand[2 * k + 1] + 1] - seq_off[cand[2 * k + 1]];
      const uint64_t m = la < lb ? la : lb, nn = la < lb ? lb : la;
      if (m > 4096) { hoff[k] = hwords; hwords += 2 * nn; }
    }
    if ((rc = upload(h, h->g_qs, seq, (size_t)seq_off[n]))) return rc;
    if ((rc = upload(h, h->g_off, seq_off, (size_t)(n + 1) * 8))) return rc;
    if ((rc = upload(h, h->g_qe, hoff.data(), (size_t)n_cand * 8))) return rc;
    if ((rc = ensure(h, h->g_rs, (size_t)hwords * 4))) return rc;
    if ((rc = ensure(h, h->g_rev, (size_t)n_cand))) return rc;
    if ((rc = ensure(h, h->g_re, (size_t)n_cand * 4))) return rc;
    vsv_launch_rr_edit_sim(st, (const uint32_t*)h->j_out.p, (uint32_t)n_cand, (const uint8_t*)h->g_qs.p, (const uint64_t*)h->g_off.p,
                           (const uint64_t*)h->g_qe.p, (int32_t*)h->g_rs.p, p->seq_sim_thresh, (uint8_t*)h->g_rev.p, (uint32_t*)h->g_re.p);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(flag.data(), h->g_rev.p, (size_t)n_cand, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
  }
  int64_t m = 0;
  for (int64_t k = 0; k < n_cand; ++k) m += flag[k] ? 1 : 0;
  h->last_count = m;
  *n_pairs = m;
  if (m > cap) return fail(h, VSV_E_CAPACITY, "pair buffer too small");
  if (m && !pairs) return fail(h, VSV_E_INVALID, "pairs is NULL");
  int64_t w = 0;
  for (int64_t k = 0; k < n_cand; ++k) if (flag[k]) { pairs[2 * w] = cand[2 * k]; pairs[2 * w + 1] = cand[2 * k + 1]; ++w; }
  return 0;
}

int vsv_last_scan_ms(vsv_handle* h, float* ms) {
  if (!h || !ms || !h->have_scan_ev) return VSV_E_INVALID;
  HIPCHK(h, hipEventSynchronize(h->ev1));
  HIPCHK(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
  return 0;
}

static int table_src(vsv_handle* h, int table, const void** src, int64_t* n_rows, size_t* row, bool* filter) {
  const Counters& c = h->host_ctr;
  // a run on elements (slim_path.hip) gathers the rows of these two tables when somebody asks for them
  if (h->big_run && ((table == VSV_T_CLUSTER1 && h->stage_done >= 3 && h->c1_stale) || (table == VSV_T_MERGED && h->stage_done >= 4 && h->merged_stale))) {
    if (hipSetDevice(h->device) != hipSuccess) return VSV_E_HIP;
    if (table == VSV_T_CLUSTER1) { vsv_slim_rows(h->stream, h->sl_e2, c.n_alive1, (const vsv_sig*)h->s1in.p, (vsv_sig*)h->c1.p); h->c1_stale = false; }
    else { vsv_slim_rows(h->stream, h->sl_m, c.n_alive3, (const vsv_sig*)h->s1in.p, (vsv_sig*)h->merged.p); h->merged_stale = false; }
    if (hipStreamSynchronize(h->stream) != hipSuccess) return VSV_E_HIP;
  }
  if (table == VSV_T_RAW && h->stage_done >= 1 && h->raw_state != 0) {
    // the scan placed its rows straight into the stage-1 input table (where the fold has run since): the rows as the scan emitted
    // them are built again from its descriptors, which nothing else writes
    if (h->raw_state == 2) return VSV_E_INVALID;
    if (hipSetDevice(h->device) != hipSuccess) return VSV_E_HIP;
    if (++h->lb_epoch >= (1u << 24)) { if (hipMemsetAsync(h->lbw.p, 0, h->lbw.bytes, h->stream) != hipSuccess) return VSV_E_HIP; h->lb_epoch = 1; }
    vsv_launch_place(h->stream, h->rv, h->prm, h->raw_parts, (vsv_sig*)h->pool.p, (uint64_t*)h->pool_key.p, (uint32_t)h->cap_sigs, (uint32_t*)h->part_count.p,
                          (uint32_t*)h->part_off.p, (vsv_sig*)h->raw0.p, dctr(h), (uint32_t*)h->shard_cnt.p, long_bufs(h, false, nullptr, SlimOut{nullptr, 0, 0, 0, nullptr}),
                          h->lb_epoch, SlimOut{nullptr, 0, 0, 0, nullptr});
    if (hipStreamSynchronize(h->stream) != hipSuccess) return VSV_E_HIP;
    h->raw_state = 0;
  }
  *filter = true;
  *row = sizeof(vsv_sig);
  switch (table) {
    case VSV_T_RAW: if (h->stage_done < 1) return VSV_E_INVALID; *src = h->raw0.p; *n_rows = c.n_raw; return 0;
    case VSV_T_CIGAR: if (h->stage_done < 1) return VSV_E_INVALID; *src = h->s1in.p; *n_rows = c.n_raw; return 0;
    case VSV_T_SPLIT: if (h->stage_done < 2) return VSV_E_INVALID; *src = (const vsv_sig*)h->s1in.p + c.n_raw; *n_rows = c.n_s1 - c.n_raw; return 0;
    case VSV_T_CLUSTER1: if (h->stage_done < 3) return VSV_E_INVALID; *src = h->c1.p; *n_rows = c.n_alive1; return 0;
    case VSV_T_MERGED: if (h->stage_done < 4) return VSV_E_INVALID; *src = h->merged.p; *n_rows = c.n_alive3; *filter = false; return 0;
    case VSV_T_CALLS: if (h->stage_done < 5) return VSV_E_INVALID; *src = h->calls.p; *n_rows = c.n_calls; *row = sizeof(vsv_call); *filter = false; return 0;
    case VSV_T_BND_CAND: if (h->bnd_stage < 1) return VSV_E_INVALID; *src = h->s1in.p; *n_rows = c.n_s1; return 0;
    case VSV_T_BND_CALLS: if (h->bnd_stage < 2) return VSV_E_INVALID; *src = h->c1.p; *n_rows = c.n_alive1; return 0;
    case VSV_T_BND_SLOTS: if (h->bnd_stage < 1) return VSV_E_INVALID; *src = h->s1in.p; *n_rows = c.n_s1; *filter = false; return 0;
    case VSV_T_BND_CALL_SLOTS: if (h->bnd_stage < 2) return VSV_E_INVALID; *src = h->c1.p; *n_rows = c.n_alive1; *filter = false; return 0;
    case VSV_T_CUTESV_SPLIT: if (h->cutesv_rows < 0) return VSV_E_INVALID; *src = h->c2.p; *n_rows = h->cutesv_rows; *filter = false; return 0;
    case VSV_T_READS: if (h->stage_done < 2 || h->prm.dtype != VSV_DTYPE_READS) return VSV_E_INVALID; *src = h->reads.p; *n_rows = c.n_reads; *filter = false; return 0;
  }
  return VSV_E_INVALID;
}

static int table_read(vsv_handle* h, int table, std::vector<char>& host, int64_t* n_out, size_t* row_out) {
  if (!h) return VSV_E_INVALID;
  if (h->pending) { int st = finish(h); if (st) return st; }
  const void* src; int64_t n; size_t row; bool filter;
  int st = table_src(h, table, &src, &n, &row, &filter);
  if (st) return fail(h, st, "table not available at this stage");
  host.resize((size_t)n * row + 1);
  if (n) HIPCHK(h, hipMemcpy(host.data(), src, (size_t)n * row, hipMemcpyDeviceToHost));
  int64_t m = n;
  if (filter) {  // drop rows a stage marked dead
    m = 0;
    vsv_sig* s = (vsv_sig*)host.data();   // vsv_bnd has the same size and its meta word at the same offset
    const uint32_t dead = (table == VSV_T_BND_CAND || table == VSV_T_BND_CALLS) ? (uint32_t)VSV_B_DEAD : (uint32_t)VSV_M_DEAD;
    for (int64_t i = 0; i < n; ++i) if (!(s[i].meta & dead)) s[m++] = s[i];
  }
  *n_out = m; *row_out = row;
  return 0;
}

int vsv_table_count(vsv_handle* h, int table, int64_t* n_rows) {
  if (!n_rows || !h) return VSV_E_INVALID;
  if (h->pending) { int st = finish(h); if (st) return st; }
  const void* src; int64_t n; size_t row; bool filter;
  int st = table_src(h, table, &src, &n, &row, &filter);
  if (st) return fail(h, st, "table not available at this stage");
  if (!filter) { *n_rows = n; return 0; }      // live tables: the device counter is the row count
  std::vector<char> host;
  return table_read(h, table, host, n_rows, &row);
}

int vsv_table_fill(vsv_handle* h, int table, void* dst, int64_t cap_rows, int dst_on_device) {
  if (!h || (!dst && cap_rows > 0)) return VSV_E_INVALID;
  if (h->pending) { int st = finish(h); if (st) return st; }
  const void* src; int64_t n; size_t row; bool filter;
  int st = table_src(h, table, &src, &n, &row, &filter);
  if (st) return fail(h, st, "table not available at this stage");
  if (!filter) {                                 // straight copy, no staging
    if (n > cap_rows) { h->last_count = n; return fail(h, VSV_E_CAPACITY, "destination too small"); }
    // on the handle's stream, then a wait on THAT stream only: a device-wide synchronisation here would stall the other engines of
    // a rank that keeps several chromosomes in flight
    if (n) {
      HIPCHK(h, hipMemcpyAsync(dst, src, (size_t)n * row, dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return 0;
  }
  std::vector<char> host;
  st = table_read(h, table, host, &n, &row);
  if (st) return st;
  if (n > cap_rows) { h->last_count = n; return fail(h, VSV_E_CAPACITY, "destination too small"); }
  if (n == 0) return 0;
  if (dst_on_device) HIPCHK(h, hipMemcpy(dst, host.data(), (size_t)n * row, hipMemcpyHostToDevice));
  else memcpy(dst, host.data(), (size_t)n * row);
  return 0;
}

}  // extern "C"
