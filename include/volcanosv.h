/*
 * volcanosv.h — C-ABI of the MI355X-native SV-signature hot path.
 *
 * The reference (maiziezhoulab/VolcanoSV) has no FFI for this path: the boundary is a
 * process + file contract (Raw_variant_call.py:65-73 spawns extract_contig_signature_<dtype>.py,
 * Raw_variant_call.py:83-88 spawns extract_reads_signature.py). This header is what a binding for
 * the *functions behind those scripts* would bind; each entry point cites the reference function it
 * replaces. Paths are relative to bin/VolcanoSV-vc/ in the reference:
 *   H  = Large_INDEL/extract_contig_signature_Hifi.py     O = ..._ONT.py     C = ..._CLR.py
 *   RS = Large_INDEL/extract_reads_signature.py
 *   SV = Complex_SV/svim-asm-1.0.2/src/svim_asm/
 *
 * Conventions: plain C, no C++ types or exceptions across the boundary; every call returns an int
 * status (0 = ok, <0 = vsv_status); no global mutable state; one handle per device/stream; calls on
 * one handle are not thread-safe, different handles are independent. Strings (qname, REF/ALT) never
 * cross the boundary — records are referenced by index into the caller's SoA.
 */
#ifndef VOLCANOSV_H
#define VOLCANOSV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSV_ABI_VERSION 2

/* ---- status codes ------------------------------------------------------------------------ */
typedef enum vsv_status {
  VSV_OK = 0,
  VSV_E_INVALID = -1,      /* bad argument (null pointer, misaligned cigar, n < 0 ...)            */
  VSV_E_HIP = -2,          /* HIP runtime error, text in vsv_last_error                            */
  VSV_E_CAPACITY = -3,     /* signature capacity exceeded; required count in vsv_last_count         */
  VSV_E_EMPTY_CIGAR = -4,  /* record with 0 CIGAR ops (reference: IndexError at H:63 cigar[0])      */
  VSV_E_REFEND = -5,       /* reference `assert offset_ref==read.reference_end` (H:396, RS:123)     */
  VSV_E_READLEN = -6,      /* reference `assert rl1==rl2` (H:331, RS:172)                           */
  VSV_E_UNSORTED = -7,     /* reference `assert read1.pos<=read2.pos` (H:315) / cigar_off not mono. */
  VSV_E_ZERODIV = -8,      /* CLR gate ZeroDivisionError (C:61, C:70)                               */
  VSV_E_NO_DEVICE = -9,    /* no HIP device / extension built without a GPU present                */
  VSV_E_SEQLEN = -10       /* reference `assert len(read.seq)==offset_contig` (H:397-398, RS:123-124)       */
} vsv_status;

/* ---- data types (dtype of the extractor) -------------------------------------------------- */
enum {
  VSV_DTYPE_HIFI = 0,  /* H  */
  VSV_DTYPE_ONT = 1,   /* O  */
  VSV_DTYPE_CLR = 2,   /* C  */
  VSV_DTYPE_READS = 3, /* RS: M-like ops {0,7,8}, N advances ref, no fold / cluster / pair          */
  VSV_DTYPE_SVIM = 4,  /* SV/SVIM_intra.py:8-30 op table (CIGAR stage only)                         */
  VSV_DTYPE_CUTESV = 5 /* Large_INDEL/sig_extract.py parse_read (SE:438-493): M,D,=,X advance the reference, every op
                        * but D the read offset; CIGAR stage + in-read merging (generate_combine_sigs, SE:373-435)   */
};

/* ---- record flag bits (u8 per record) ------------------------------------------------------ */
enum {
  VSV_F_REVERSE = 1,   /* read.is_reverse                                                        */
  VSV_F_SUPP = 2,      /* supplementary (only used by the svim table)                             */
  VSV_F_HP1 = 4,       /* 'hp1' in qname (H:392)                                                  */
  VSV_F_HP2 = 8,       /* 'hp2' in qname                                                          */
  VSV_F_SECONDARY = 16,
  VSV_F_UNMAPPED = 32,
  VSV_F_SKIP = 64,     /* record excluded by the host adaptor (sig_extract: query_length < min_read_len, SE:439; BED) */
  VSV_F_SEQ_MISMATCH = 128 /* the record stores a SEQ whose length differs from the query length of its CIGAR (M,I,S,=,X). Set by the
                              BAM readers; a record the extractors walk with this bit raises VSV_E_SEQLEN like the reference's
                              `if read.seq: assert len(read.seq)==offset_contig` (H:397-398, O:408-409, C:430-431, RS:123-124)   */
};

/* ---- caller-owned SoA of alignment records (BAM order: tid, pos ascending) ------------------ */
typedef struct vsv_records {
  int64_t n_records;
  int64_t n_ops;               /* == cigar_off[n_records]                                         */
  const int32_t* pos;          /* [n]   0-based leftmost reference position (read.pos)             */
  const int32_t* tid;          /* [n]   reference id                                               */
  const uint32_t* qid;         /* [n]   dense query-name id (same qname <=> same qid)              */
  const uint64_t* cigar_off;   /* [n+1] exclusive prefix of per-record op counts                   */
  const uint8_t* mapq;         /* [n]                                                              */
  const uint8_t* flag;         /* [n]   VSV_F_* bits                                               */
  const uint32_t* cigar;       /* [n_ops] BAM packing: len<<4 | op ; 16-byte aligned               */
  int32_t on_device;           /* 0: host pointers (library uploads), 1: device pointers — their contents must be complete
                                  before the call, or ordered before the handle's stream with vsv_wait_for_stream: the
                                  library reads them on the handle's stream and does not know the stream that produced them */
  int32_t n_qids;              /* max qid + 1 (required when n_records > 0)                            */
  int32_t n_tids;              /* max tid + 1; 0 = unknown (16 key bits are reserved for it)       */
  int32_t max_pos;             /* upper bound of pos (e.g. contig length); 0 = unknown. Speed only: it trims sort passes and balances the
                                * sorters' buckets - without it a handle's bucket sorts overflow on a real chromosome and vsv_finish repeats
                                * the run on the radix passes now and then (vsv_rerun_count). */
  int32_t tid_lo;              /* lowest tid of this run (0 = unknown): sort keys carry tid - tid_lo, so a single-
                                  chromosome shard (tid_lo = tid, n_tids = tid + 1) sorts on one tid bit        */
  int32_t reserved0;
} vsv_records;

/* ---- parameters; defaults equal the hard-coded reference values ----------------------------- */
typedef struct vsv_params {
  int32_t dtype;            /* VSV_DTYPE_*                                                       */
  int32_t min_svlen;        /* 30  (H:395 extract_sig_from_cigar(read,min_svlen=30))              */
  int32_t min_cigar_mapq;   /* 50  (H:24)                                                         */
  int32_t min_split_mapq;   /* 50  (H:41 assigns min_cigar_mapq); RS: 0 (RS:235)                  */
  int32_t max_split_svlen;  /* 50000 (H:455)                                                      */
  int32_t cluster_shift;    /* 100 (H:407-415)                                                    */
  int32_t pair_shift;       /* 200 (H:562,564)                                                    */
  int32_t pair_window;      /* 1000 (H:768 max_compare_dist)                                      */
  int32_t enable_split;     /* 1: run the split-alignment stage                                   */
  int32_t merge_ins_threshold; /* CUTESV only: 100 (SE -mi): INS signals of one read at most this far apart are merged */
  int32_t merge_del_threshold; /* CUTESV only: 0   (SE -md)                                           */
  int32_t scan_layout;      /* VSV_SCAN_*: work mapping of the CIGAR scan; results are identical, only the speed differs  */
  int32_t split_overlap;    /* VSV_OVERLAP_*: where vsv_run_chromosome builds the split candidates; results are identical        */
  int32_t reserved[3];
} vsv_params;
enum {
  VSV_OVERLAP_AUTO = 0, /* on a second stream of the handle, next to the CIGAR scan: shortest latency of one handle             */
  VSV_OVERLAP_OFF = 1   /* on the handle's stream: callers that keep three or more handles busy on one GPU (the engines'
                           kernels overlap each other already; the extra stream only adds events). The sorts then also run
                           in workgroups of one wave per SIMD, which find room beside another handle's scan                     */
};
enum {
  VSV_SCAN_AUTO = 0,    /* by the mean CIGAR length of the call's records                                             */
  VSV_SCAN_READS = 1,   /* record-aligned parts, lazily evaluated offsets: reads, tens to hundreds of ops per record  */
  VSV_SCAN_CONTIGS = 2  /* fixed 8192-op parts, a record may span parts: contig alignments, 10^3-10^6 ops per record  */
};

/* ---- signature row (32 bytes) ---------------------------------------------------------------
 * Mirrors the reference 10-field list (H:80,84): chrom->tid, type/source/hap in meta, pos, svlen,
 * qname/strand/mapq via rec (and rec2 for split signatures, whose mapq is "m1-m2", H:358).      */
typedef struct vsv_sig {
  int32_t pos;      /* sig[2]                                                                     */
  int32_t svlen;    /* sig[3]                                                                     */
  int32_t q_start;  /* sig[5]                                                                     */
  int32_t q_end;    /* sig[6]  (READS cigar signatures have no q_end: 0)                          */
  uint32_t rec;     /* record index of read / read1                                               */
  uint32_t rec2;    /* record index of read2 for split signatures, 0xFFFFFFFF otherwise           */
  uint32_t meta;    /* VSV_M_* bits                                                               */
  int32_t tid;      /* sig[0]                                                                     */
} vsv_sig;

enum {
  VSV_M_DEL = 1,    /* bit0: 0 = INS, 1 = DEL                                                     */
  VSV_M_SPLIT = 2,  /* bit1: 0 = 'cigar', 1 = 'split-alignment'                                   */
  VSV_M_HP2 = 4,    /* bit2: 0 = hp1 pass, 1 = hp2 pass (always 0 for READS/SVIM)                 */
  VSV_M_QREV = 16,  /* CUTESV split INS only: the sequence slice [q_start, q_end) is taken from the REVERSED read (SE:215) */
  VSV_M_DEAD = 8    /* internal: folded away by the intra-read merge                              */
};

/* ---- call row (48 bytes): pair_sig output (H:571-592) ---------------------------------------- */
typedef struct vsv_call {
  vsv_sig sig;      /* the kept signature (sig1 if l1>l2 else sig2, H:583-586)                    */
  int32_t a;        /* index into VSV_T_MERGED of the hp1 member, -1 if none                      */
  int32_t b;        /* index into VSV_T_MERGED of the hp2 member, -1 if none                      */
  int32_t gt;       /* 1 = '0/1', 2 = '1/1'                                                       */
  int32_t pad;
} vsv_call;

/* ---- stage tables a caller can read back ----------------------------------------------------- */
enum {
  VSV_T_RAW = 0,      /* vsv_sig: CIGAR emit stream, (rec, op) order, before the intra-read fold  */
  VSV_T_CIGAR = 1,    /* vsv_sig: after the fold (H:108-161), same order, dead rows removed       */
  VSV_T_SPLIT = 2,    /* vsv_sig: split signatures in (hap, name first-appearance, pair) order    */
  VSV_T_CLUSTER1 = 3, /* vsv_sig: representatives of the per-source clustering (H:407-415,465-473) */
  VSV_T_MERGED = 4,   /* vsv_sig: merge_all output per (tid, hap), sorted by pos (H:492-499)      */
  VSV_T_CALLS = 5,    /* vsv_call: pair_sig output, sorted by (tid,pos) (H:548-603)               */
  VSV_T_READS = 6     /* vsv_sig: RS merge_all order: sort(del_cigar+ins_cigar+del_split+ins_split) */
};

typedef struct vsv_handle vsv_handle;

int vsv_abi_version(void);
const char* vsv_status_string(int status);

/* create/destroy a per-device context. `hip_stream` is a hipStream_t (NULL = default stream).      */
int vsv_create(int device_id, void* hip_stream, vsv_handle** out);
void vsv_destroy(vsv_handle* h);
const char* vsv_last_error(vsv_handle* h);
int64_t vsv_last_count(vsv_handle* h);

/* params with the reference's hard-coded values for `dtype` */
int vsv_default_params(int dtype, vsv_params* p);

/* Device-resident inputs are read on the HANDLE's stream. If another stream produced them and may still be running, either
 * synchronise that stream on the host before the call, or call this first: the handle's stream will wait (on the device, no host
 * synchronisation) for everything enqueued on `producer_hip_stream` (a hipStream_t; NULL = the default stream) so far. */
int vsv_wait_for_stream(vsv_handle* h, void* producer_hip_stream);

/* capacity (rows) of the signature tables; default 1<<22. Re-allocates the workspace: growing the row capacity drops the tables
 * of the handle's last run (they live in the buffers that move). */
int vsv_reserve(vsv_handle* h, int64_t max_records, int64_t max_ops, int64_t max_sigs);
/* Optional: allocate, for the current row capacity, the element buffers of the large-table form of the stages behind the split
 * stage (96 bytes per row of capacity) ahead of the first run. Without it the first contig run of a handle allocates them before its
 * first launch (never in the middle of a run). */
int vsv_reserve_large_tables(vsv_handle* h);

/* Stage entry points. Each consumes the handle state left by the previous one.
 * vsv_cigar_scan      replaces extract_sig_from_cigar + the loop of extract_signature_from_cigar
 *                     (H:53-166, 386-400; RS:47-83, 107-125; SV/SVIM_intra.py:8-30)
 * vsv_split_pairs     replaces extract_sig_from_split_reads + extract_sig_from_split
 *                     (H:307-371, 421-457; O:307-382; C:328-402; RS:147-248)
 * vsv_sort_cluster    replaces sort_sig + cluster_del/cluster_ins on the per-source lists
 *                     (H:170-179, 196-288, 402-415, 459-473)
 * vsv_merge_sources   replaces merge_all (H:478-499)
 * vsv_pair_haplotypes replaces pair_sig (H:515-603)
 * vsv_run_chromosome  = all of the above for every tid present, no host synchronisation between
 *                     stages (replaces the per-chromosome body of H:742-772 minus VCF text).     */
int vsv_cigar_scan(vsv_handle* h, const vsv_records* recs, const vsv_params* p);
int vsv_split_pairs(vsv_handle* h, const vsv_records* recs, const vsv_params* p);
int vsv_sort_cluster(vsv_handle* h, const vsv_params* p);
int vsv_merge_sources(vsv_handle* h, const vsv_params* p);
int vsv_pair_haplotypes(vsv_handle* h, const vsv_params* p);
int vsv_run_chromosome(vsv_handle* h, const vsv_records* recs, const vsv_params* p);

/* Asynchronous variant for benchmarking: enqueues the whole path on the handle's stream and does
 * not synchronise; errors and counts are picked up by vsv_finish().                              */
int vsv_run_chromosome_async(vsv_handle* h, const vsv_records* recs, const vsv_params* p);
/* Measurement aid: best of `reps` sweeps of the library's own 16-byte read-stream and copy kernels over a device
 * buffer of the caller (>= 1 MiB), in GB/s; the copy figure counts bytes read + bytes written. */
int vsv_stream_ceiling(vsv_handle* h, const void* dev_buf, int64_t bytes, int32_t reps, double* read_gbs, double* copy_gbs);
int vsv_finish(vsv_handle* h);
/* Measurement aid: how many times vsv_finish() had to repeat a whole run on this handle so far — a bucket of the bucket sort that
 * did not fit in LDS (the run is repeated through the LSD radix passes) or a part too long for the gate state of the fused CLR
 * scan (repeated with the separate gate pass). Results are identical either way; a bench line reports the count so that a hidden
 * repetition inside a timed region is visible. */
int64_t vsv_rerun_count(vsv_handle* h);
/* Measurement aid: runs of this handle whose first sort on 16-byte elements (one counting pass into position buckets + an LDS sort per
 * bucket) met a bucket too large for LDS — a pile far from uniform over the chromosome, or a size hint that was stale. Such a bucket
 * is sorted in global memory by the same launch (slower, same result: no repetition) and the handle's next runs take the LSD radix
 * passes for that sort. */
int64_t vsv_sort1_slow_count(vsv_handle* h);
/* Measurement aid: which form of the stages behind the split stage the handle's runs took, and whether a cold handle had to wait.
 * element_runs: runs (or staged vsv_sort_cluster calls) whose sort / cluster / merge / pair stages worked on 16-byte elements — the
 * form for tables beyond ~1.3 M rows (10^6-10^7: contig alignments piled on one chromosome, ONT-scale read sets); the others took
 * the row form. cold_syncs: fused runs of a handle without history (its first one — every invocation of the drop-in CLI, which runs
 * one chromosome per process like Raw_variant_call.py:65-73) that waited for the scan once to take the row count from it, so that
 * the first run of a handle takes the same path as a later one. Either pointer may be NULL. Results never depend on the path. */
int vsv_path_counts(vsv_handle* h, int64_t* element_runs, int64_t* cold_syncs);

/* two-phase readback: count, then fill a caller buffer (host or device) */
int vsv_table_count(vsv_handle* h, int table, int64_t* n_rows);
int vsv_table_fill(vsv_handle* h, int table, void* dst, int64_t cap_rows, int dst_on_device);

/* timing of the dominant kernel (cigar_scan_emit) of the last vsv_finish()/sync call, measured
 * with HIP events on the handle's stream. Returns milliseconds in *ms.                            */
int vsv_last_scan_ms(vsv_handle* h, float* ms);

/* ---- Complex_SV / svim-asm breakend (BND) branch ---------------------------------------------------
 * Input: split contigs as segment lists — for every primary alignment that has SA-tag alignments passing
 * min_mapq, the primary + those alignments (SV/SVIM_COLLECT.py:8-54, 67-76) reduced to the six numbers
 * analyze_read_segments reads (SV/SVIM_inter.py:62-81): q_start/q_end already flipped for reverse strands.
 * vsv_bnd_segments  replaces the BND branches of analyze_read_segments (SV/SVIM_inter.py:83-258) and the
 *                   canonical orientation / clamping of CandidateBreakend (SV/SVCandidate.py:350-373);
 * vsv_bnd_pair      replaces form_partitions + pair_haplotypes_breakends + the BND part of pair_candidates
 *                   (SV/SVIM_COMBINE.py:15-32, 105-117, 143-161, 334-365). */
typedef struct vsv_segments {
  int64_t n_reads;              /* primary alignments with >= 1 good supplementary alignment            */
  int64_t n_segs;
  const uint64_t* seg_off;      /* [n_reads+1]; segment 0 of a read is the primary alignment            */
  const int32_t* q_start;       /* [n_segs] read coordinates, reverse strands already flipped (:68-73)  */
  const int32_t* q_end;
  const int32_t* ref_id;
  const int32_t* ref_start;
  const int32_t* ref_end;
  const uint8_t* is_reverse;
  const uint8_t* hap;           /* [n_reads] 1 = first BAM (hp1), 2 = second BAM (hp2)                  */
  const int32_t* contig_len;    /* [n_tids] bam.get_reference_length                                    */
  const int32_t* contig_rank;   /* [n_tids] rank of the contig NAME in Python string order (:352, 'chr10'<'chr2') */
  int32_t n_tids;
  int32_t on_device;
} vsv_segments;

typedef struct vsv_bnd_params {
  int32_t min_sv_size;                  /* 40      (SV/SVIM_input_parsing.py:162-164) */
  int32_t max_sv_size;                  /* 100000  */
  int32_t query_gap_tolerance;          /* 50 */
  int32_t query_overlap_tolerance;      /* 50 */
  int32_t reference_gap_tolerance;      /* 50 */
  int32_t reference_overlap_tolerance;  /* 50 */
  int32_t partition_max_distance;       /* 1000 (:219-221) */
  int32_t pair_distance;                /* 900: (|d1|+|d2|)/3000 <= 0.3 (SV/SVIM_COMBINE.py:105-117, 143) */
  int32_t max_partition;                /* 10: larger partitions are ignored (:151-152); at most 16 */
  int32_t reserved[7];
} vsv_bnd_params;

typedef struct vsv_bnd {       /* 32 bytes */
  int32_t src_tid, src_pos;    /* CandidateBreakend.source_contig / source_start (0-based) */
  int32_t dst_tid, dst_pos;
  uint32_t read;               /* read row (primary alignment) in vsv_segments that produced it        */
  uint32_t read2;              /* second read of a 1/1 pair, else 0xFFFFFFFF                          */
  uint32_t meta;               /* VSV_B_* */
  uint32_t pad;
} vsv_bnd;
enum {
  VSV_B_SRC_FWD = 1, VSV_B_DST_FWD = 2,
  VSV_B_HAP2 = 4,              /* candidate came from the second BAM                                  */
  VSV_B_GT_SHIFT = 4,          /* bits 4-5: 1 = "1/0", 2 = "0/1", 3 = "1/1" (calls only)              */
  VSV_B_DEAD = 64
};
enum { VSV_T_BND_CAND = 7, VSV_T_BND_CALLS = 8,
       /* the same two tables unfiltered, i.e. every slot (one candidate slot per adjacent segment pair, VSV_B_DEAD where the pair
        * yields no breakend; call slots with the second members of 1/1 pairs and the dropped partitions VSV_B_DEAD): plain
        * device-to-device copies for callers that stay on the GPU (multi-GPU exchange) */
       VSV_T_BND_SLOTS = 11, VSV_T_BND_CALL_SLOTS = 12 };

int vsv_default_bnd_params(vsv_bnd_params* p);
int vsv_bnd_segments(vsv_handle* h, const vsv_segments* segs, const vsv_bnd_params* p);
int vsv_bnd_pair(vsv_handle* h, const vsv_bnd_params* p);
/* multi-GPU join: install candidate rows received from other ranks (collection order: hp1 rows, then hp2 rows)
 * as the input of vsv_bnd_pair. */
int vsv_bnd_set_candidates(vsv_handle* h, const vsv_bnd* rows, int64_t n, const int32_t* contig_rank, int32_t n_tids, int on_device);

/* ---- genotype correction: correct_gt_del_real_data.py / correct_gt_ins_real_data.py (filter_GT_correction.py:150-170) ----------
 * vsv_gt_support  replaces the window sums of match_varlist_siglist (DG:92-137) / extract_sig_support (IG:105-156): for variant i
 *   sum[i] = sum of sig_cnt over the signatures j in [blk_lo[i], blk_hi[i]) (the variant's chromosome block, positions ascending)
 *   with pos - max_shift <= sig_pos[j] <= pos + max_shift, max_shift = max(svlen * max_shift_ratio, 500), and
 *   svlen * min_size_sim <= sig_svlen[j] <= svlen / min_size_sim (fp64 like the reference); lo[i] / hi[i] = the window's index
 *   range, from which the caller replays the reference's resume-index bookkeeping. Host arrays.
 * vsv_span_count  replaces count_reads_span_region (DG:140-147) / check_full_cover_reads (IG:178-186): out[i] = records of
 *   reference q_tid[i] with reference_start < q_a[i] and reference_end > q_b[i]; `recs` ascending by (tid, pos), host or device. */
int vsv_gt_support(vsv_handle* h, const int32_t* var_pos, const int32_t* var_svlen, const int32_t* blk_lo, const int32_t* blk_hi, int64_t n_var,
                   const int32_t* sig_pos, const int32_t* sig_svlen, const int32_t* sig_cnt, int64_t n_sig, double max_shift_ratio,
                   double min_size_sim, int64_t* sum, int32_t* lo, int32_t* hi);
int vsv_span_count(vsv_handle* h, const vsv_records* recs, const int32_t* q_tid, const int32_t* q_a, const int32_t* q_b, int64_t n_q, uint32_t* out);

/* ---- sig_extract.py split-read branch ------------------------------------------------------------------------------
 * vsv_cutesv_split replaces analysis_split_read (SE:193-319), INS/DEL candidates only (the TRA candidates of analysis_bnd
 * never reach INS.sigs / DEL.sigs, SE:637-638). Input: for every flag-0/16 read with an SA tag, the segment list
 * organize_split_signal builds (SE:341-371): the primary [clip_left, query_length - clip_right, pos, reference_end] when its
 * mapq passes, then one entry per SA alignment; read_len = read.query_length, read_rec = the record the rows refer to.
 * Rows (table VSV_T_CUTESV_SPLIT, vsv_sig): pos, svlen, tid = chromosome of the second segment; INS rows carry the
 * sequence slice [q_start, q_end) (Python slice semantics) and VSV_M_QREV when it indexes the reversed read. */
enum { VSV_T_CUTESV_SPLIT = 10 };
int vsv_cutesv_split(vsv_handle* h, const vsv_segments* segs, const int32_t* read_len, const uint32_t* read_rec,
                     int32_t sv_size, int32_t max_size, int32_t max_split_parts);
/* has_tra[r] = 1 when read r of the last vsv_cutesv_split yields a translocation candidate (analysis_bnd, SE:100-191, reached with
 * its two segments at most 100 read bases apart): such a candidate never reaches INS.sigs / DEL.sigs, but it keeps the read's 10 Mb
 * task from being skipped as empty (SE:533-535), i.e. the task's reads are written to reads.sigs. Host array of n_reads bytes. */
int vsv_cutesv_split_tra(vsv_handle* h, uint8_t* has_tra, int64_t n_reads);

/* ---- post-filter: read-signature support of the calls (the step right of the path) -------------------
 * Replaces FP_filter_v1.eval_sig + compare_sigs (Large_INDEL/FP_filter_v1.py:87-123), run by Raw_variant_call.py:91-96
 * on the raw variant VCF against <chr>_reads_sig.txt. A call is (pos, |len(ALT)-len(REF)|) (FP:41-57), a read signature
 * is (pos, svlen) (FP:77-86); the SV type is NOT part of the predicate. */
typedef struct vsv_support_params {
  int32_t max_comp_svlen;      /* 250   calls longer than this are not checked: support 60 (FP:110-111)  */
  int32_t max_dist;            /* 1000  scan window around the call position (FP:116-119)                 */
  int32_t max_shift;           /* 500   |pos_sig - pos_call| <= max_shift (FP:98-99)                      */
  int32_t pad;
  double min_size_sim;         /* 0.5   min(len)/max(len) >= min_size_sim; 0/0 counts as 0 (FP:93-96)     */
} vsv_support_params;
int vsv_default_support_params(vsv_support_params* p);
/* support[i] = 60 if call_len[i] > max_comp_svlen, else the number of read signatures inside the window that pass
 * compare_sigs (FP:106-123). sig_pos must ascend (the reference file is sorted by merge_all, RS:281-286; its early `break` relies on it), else
 * VSV_E_UNSORTED. on_device: all five arrays are device pointers (and the call synchronises the handle's stream). */
int vsv_support_join(vsv_handle* h, const int32_t* call_pos, const int32_t* call_len, int64_t n_calls,
                     const int32_t* sig_pos, const int32_t* sig_len, int64_t n_sigs,
                     const vsv_support_params* p, int on_device, uint32_t* support);

/* Signature coverage of the calls: calculate_signature_support.py (Large_INDEL), the first step of the GT-correction filter
 * chain (filter_GT_correction.py:134-137).
 *  vsv_support_cov_ins  replaces calc_ins_call_cov (CS:81-125): cov[i] = sum of sig_len over the INS signatures with
 *                       |sig_pos - call_pos[i]| <= flanking (the reference's weighted bins keyed by call position).
 *  vsv_support_cov_del  replaces calc_del_call_cov (CS:138-280): cov[i] = sum of sig_svlen over the DISTINCT DEL signatures
 *                       whose closed interval [sig_start, sig_end] meets [call_start[i]-flanking, call_end[i]+flanking]
 *                       (the union of the reference's four boundary scans, de-duplicated as by its set()).
 * Signatures must ascend by position / start (VSV_E_UNSORTED otherwise; the .sigs files are sorted, SE:637-638). The reference
 * orders the call regions with an unstable argsort and mixes sorted and original call indices (CS:171, 213-241); the result
 * here is the one it produces on a position-sorted VCF with ties kept in file order. */
int vsv_support_cov_ins(vsv_handle* h, const int32_t* call_pos, int64_t n_calls, const int32_t* sig_pos, const int32_t* sig_len,
                        int64_t n_sigs, int32_t flanking, int on_device, int64_t* cov);
int vsv_support_cov_del(vsv_handle* h, const int32_t* call_start, const int32_t* call_end, int64_t n_calls,
                        const int32_t* sig_start, const int32_t* sig_end, const int32_t* sig_svlen, int64_t n_sigs,
                        int32_t flanking, int on_device, int64_t* cov);

/* ---- remove_redundancy.py: redundant-call matching (last step of Raw_variant_call.py, RV:99-104) ------------------------
 * vsv_redundancy_pairs replaces match_del_chr / match_ins_chr with their pair predicates (RR:92-134, 162-181) for the calls of
 * one chromosome in ascending position order: DEL = size similarity + reciprocal overlap, INS = size similarity + sequence
 * similarity (totlen - editDistance)/totlen of the ALT strings, where edlib's global unit-cost edit distance (RR:75-81) is
 * computed by a bit-parallel (Myers) kernel. Returns the matching pairs i < j in (i, j) order (the reference's link list
 * holds both directions); connected components and the VCF text stay on the host. */
typedef struct vsv_redundancy_params {
  int32_t dist_thresh;          /* 500   INS window and distance (RR:9)  */
  int32_t dist_thresh_del;      /* 3000  DEL (RR:10)                      */
  double overlap_thresh;        /* 0     DEL reciprocal overlap (RR:11)   */
  double size_sim_thresh;       /* 0.5   INS min/max length (RR:12)       */
  double size_sim_thresh_del;   /* 0.1   DEL (RR:13)                      */
  double seq_sim_thresh;        /* 0.5   INS sequence similarity (RR:14)  */
} vsv_redundancy_params;
int vsv_default_redundancy_params(vsv_redundancy_params* p);
/* pos ascending (VSV_E_UNSORTED otherwise); svlen = |len(REF)-len(ALT)| > 0 (VSV_E_ZERODIV otherwise, the reference divides);
 * INS only: seq = concatenated ALT strings (symbols coded 0..15 by the caller), seq_off[n+1]. Host pointers.
 * pairs: room for cap pairs (2 x u32 each); VSV_E_CAPACITY reports the needed count through vsv_last_count. */
int vsv_redundancy_pairs(vsv_handle* h, int is_del, const int32_t* pos, const int32_t* svlen, const uint8_t* seq,
                         const uint64_t* seq_off, int64_t n, const vsv_redundancy_params* p, uint32_t* pairs, int64_t cap,
                         int64_t* n_pairs);

/* ---- BGZF inflate on the GPU ------------------------------------------------------------------------------------------
 * The members of a BGZF file (htslib bgzf.c: independent raw-deflate streams, <= 64 KiB of output each) are decoded one lane
 * per member. comp = the members' deflate payloads back to back (host), comp_off[n+1] their byte offsets, isize[n] the
 * uncompressed sizes from the member trailers; out (host) receives sum(isize) bytes in member order. A member that is not a
 * valid deflate stream of exactly isize bytes fails the call with VSV_E_INVALID (vsv_last_count = its index). */
/* The device parse also collects the SA:Z tag text of every kept record (what svim-asm's retrieve_other_alignments and
 * sig_extract's split-read branch read, SVIM_COLLECT.py:12, SE:479): '\n'-joined in record order, empty for records without the
 * tag; valid until the handle's next device parse. */
int vsv_bam_device_want_sa(vsv_handle* h, int want);
/* The device reader also keeps the packed SEQ fields (4 bits per base, device-resident, in record order): pysam's query_sequence
 * of sig_extract.py (SE:468-469 slices it per INS piece, SE:215 reads the reversed read). vsv_bam_device_seq_slices decodes n slices
 * to ASCII: slice i = bases [start[i], start[i] + len[i]) of record rec[i] (of the REVERSED read where rev[i] != 0), written to
 * out + out_off[i]. Host arrays in, host bytes out; a slice outside its record or outside out_bytes is VSV_E_INVALID. */
int vsv_bam_device_want_seq(vsv_handle* h, int want);
int vsv_bam_device_seq_slices(vsv_handle* h, const uint32_t* rec, const uint32_t* start, const uint32_t* len, const uint8_t* rev, int64_t n,
                              const uint64_t* out_off, uint8_t* out, int64_t out_bytes);
const char* vsv_bam_device_sa_tags(vsv_handle* h, int64_t* len);
/* CRC-32 values of the members' gzip trailers (host array of n_members words, caller-owned, NULL / 0 clears): the next
 * vsv_bgzf_inflate / vsv_bam_parse_device over exactly n_members members computes each member's CRC-32 on the GPU and fails
 * on a mismatch, as htslib's bgzf.c does on the host. */
int vsv_bgzf_set_expected_crc(vsv_handle* h, const uint32_t* crc, int64_t n_members);
int vsv_bgzf_inflate(vsv_handle* h, const uint8_t* comp, const uint64_t* comp_off, const uint32_t* isize, int64_t n_members, uint8_t* out);

/* BAM members -> device-resident record SoA, inflated AND parsed on the GPU (no byte of the records returns to the host except
 * the query names). comp / comp_off / isize describe all BGZF members of the file as for vsv_bgzf_inflate; first_record = offset
 * of the first alignment record in the inflated stream (= length of the BAM header block), n_ref = number of reference
 * sequences; tid < 0 keeps every placed record. On success `out` holds DEVICE pointers (on_device = 1) owned by the handle,
 * valid until the next call; qids are dense in first-appearance order like vsv_bam_load's; *names / *names_len receive a
 * library-owned '\n'-joined name blob in qid order. VSV_E_INVALID: the stream is not a well-formed BAM (or a 64-bit name-hash
 * collision was detected) — use the host reader. SA tags and SEQ are not extracted by this path. */
/* plain device -> host copy on the handle's stream (for callers without a HIP runtime binding of their own) */
int vsv_copy_to_host(vsv_handle* h, void* dst, const void* src_device, int64_t bytes);
int vsv_bam_parse_device(vsv_handle* h, const uint8_t* comp, const uint64_t* comp_off, const uint32_t* isize, int64_t n_members,
                         uint64_t first_record, int32_t n_ref, int32_t tid, vsv_records* out, const char** names, int64_t* names_len,
                         const uint32_t** l_seq_dev, const uint32_t** sam_flags_dev);

/* ---- host-side ingest: BAM/BGZF -> record SoA ----------------------------------------------------
 * Replaces pysam.AlignmentFile(bam).fetch(chr) (H:387-391, RS:108-113). Arrays returned through `out` are owned
 * by the vsv_bam object and stay valid until the next vsv_bam_load / vsv_bam_close. No index is used. */
typedef struct vsv_bam vsv_bam;
int vsv_bam_open(const char* path, vsv_bam** out);
void vsv_bam_close(vsv_bam* b);
const char* vsv_bam_error(vsv_bam* b);
void vsv_bam_set_threads(vsv_bam* b, int n);                  /* BGZF inflate workers, 0 = all (<= 16) */
void vsv_bam_set_inflate_device(vsv_bam* b, vsv_handle* h);   /* inflate the windows of vsv_bam_load with vsv_bgzf_inflate (NULL: host zlib) */
int vsv_bam_n_refs(vsv_bam* b);
const char* vsv_bam_ref_name(vsv_bam* b, int i);
int64_t vsv_bam_ref_len(vsv_bam* b, int i);
int vsv_bam_load(vsv_bam* b, int tid, vsv_records* out);       /* tid < 0: every placed record */
int vsv_bam_load_device(vsv_bam* b, vsv_handle* h, int tid, vsv_records* out);   /* same, through vsv_bam_parse_device: device pointers */
const char* vsv_bam_qnames(vsv_bam* b, int64_t* len);           /* '\n'-joined names, qid order (after vsv_bam_load_device: owned by
                                                                    the GPU handle, valid until its next device parse)          */
const char* vsv_bam_sa_tags(vsv_bam* b, int64_t* len);          /* '\n'-joined SA tags, record order */
void vsv_bam_set_keep_seq(vsv_bam* b, int keep);              /* keep SEQ of the records loaded next (sig_extract INS text) */
const uint8_t* vsv_bam_seq(vsv_bam* b, int64_t* len);          /* packed 4-bit SEQ (BAM nibbles '=ACMGRSVTWYHKDBN'), record i: (l_seq[i]+1)/2 bytes */
const uint32_t* vsv_bam_l_seq(vsv_bam* b);
const uint32_t* vsv_bam_sam_flags(vsv_bam* b);
const uint32_t* vsv_bam_l_seq_device(vsv_bam* b);             /* device arrays of the last vsv_bam_load_device */
const uint32_t* vsv_bam_sam_flags_device(vsv_bam* b);

#ifdef __cplusplus
}
#endif
#endif /* VOLCANOSV_H */
