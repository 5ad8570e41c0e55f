/*
 * vsv_oracle.c — CPU restatement (plain C) of VolcanoSV's SV-signature hot path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product path (volcanosv_amd/) may import, link or call
 * this file; it is the checker for tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Parity pinning: every function below is checked against the JSON files in tests/golden/, which were generated
 * by tests/golden/make_golden.py from the reference's own functions (AST-extracted FunctionDefs of
 * the files cited below, run in the build container), plus the reference's own known-answer vectors
 * (svim-asm tests/test_intra.py:8-22, tests/test_inter.py:8-11).
 *
 * Citations are relative to /root/reference/bin/VolcanoSV-vc/ :
 *   H  = Large_INDEL/extract_contig_signature_Hifi.py   O = ..._ONT.py   C = ..._CLR.py
 *   RS = Large_INDEL/extract_reads_signature.py         SV = Complex_SV/svim-asm-1.0.2/src/svim_asm/
 *
 * Canonical tie rule (SURVEY.md §7 hard part 1): sort_sig (H:170-179) is restated as a STABLE sort
 * by pos; ties keep list order. The reference uses numpy's default (unstable) argsort, so tie-rich
 * goldens were generated with argsort(kind='stable') injected; tie-free goldens with the unmodified
 * function.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#include "../include/volcanosv.h"

typedef struct {
  vsv_sig* v;
  int64_t n, cap;
} sigvec;

static void sv_push(sigvec* s, const vsv_sig* x) {
  if (s->n == s->cap) {
    s->cap = s->cap ? s->cap * 2 : 1024;
    s->v = (vsv_sig*)realloc(s->v, (size_t)s->cap * sizeof(vsv_sig));
  }
  s->v[s->n++] = *x;
}

typedef struct orc_out {
  vsv_sig* raw;      int64_t n_raw;
  vsv_sig* cigar;    int64_t n_cigar;
  vsv_sig* split;    int64_t n_split;
  vsv_sig* cluster1; int64_t n_cluster1;
  vsv_sig* merged;   int64_t n_merged;
  vsv_call* calls;   int64_t n_calls;
  vsv_sig* reads;    int64_t n_reads;
  int32_t status;
} orc_out;

/* ------------------------------------------------------------------------------------------------
 * op tables. contig (H:72-85): M(0) advances both, S(4) advances contig, D(2)/I(1) emit+advance,
 * everything else ignored. reads (RS:66-81): {0,7,8} both, N(3) advances ref. svim
 * (SV/SVIM_intra.py:13-29): {0,7,8} both, no hard-clip offset, H ignored.
 * ------------------------------------------------------------------------------------------------ */
static inline int ref_adv(int dtype, int op) {
  if (dtype == VSV_DTYPE_CUTESV) return op == 0 || op == 7 || op == 8 || op == 2;       /* SE:450-457, 463-464 */
  if (dtype == VSV_DTYPE_READS) return op == 0 || op == 7 || op == 8 || op == 2 || op == 3;
  if (dtype == VSV_DTYPE_SVIM) return op == 0 || op == 7 || op == 8 || op == 2;
  return op == 0 || op == 2;
}
static inline int qry_adv(int dtype, int op) {
  if (dtype == VSV_DTYPE_CUTESV) return op != 2;                                        /* SE:460-461 shift_ins_read */
  if (dtype == VSV_DTYPE_READS || dtype == VSV_DTYPE_SVIM) return op == 0 || op == 7 || op == 8 || op == 1 || op == 4;
  return op == 0 || op == 1 || op == 4;
}

/* pysam reference_end = pos + sum of M,D,N,=,X */
static int64_t ref_end_pysam(const vsv_records* r, int64_t i) {
  int64_t e = r->pos[i];
  for (uint64_t k = r->cigar_off[i]; k < r->cigar_off[i + 1]; ++k) {
    int op = r->cigar[k] & 15; int64_t len = r->cigar[k] >> 4;
    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) e += len;
  }
  return e;
}

/* get_readlen: H:308-313 {0,1,4,5}; RS:148-153 {0,7,8,1,4,5} */
static int64_t read_len(const vsv_records* r, int64_t i, int dtype) {
  int64_t rl = 0;
  for (uint64_t k = r->cigar_off[i]; k < r->cigar_off[i + 1]; ++k) {
    int op = r->cigar[k] & 15; int64_t len = r->cigar[k] >> 4;
    if (op == 0 || op == 1 || op == 4 || op == 5) rl += len;
    else if (dtype == VSV_DTYPE_READS && (op == 7 || op == 8)) rl += len;
  }
  return rl;
}

/* CLR gate C:53-70, 425-427: ins_pct <= 0.13 or var_dist >= 200, in exact integer form
 * (100*ins <= 13*(M+ins); sumM >= 200*nM). Returns 1 pass, 0 fail, <0 error. */
static int clr_gate(const vsv_records* r, int64_t i) {
  int64_t ins = 0, m = 0, nm = 0;
  for (uint64_t k = r->cigar_off[i]; k < r->cigar_off[i + 1]; ++k) {
    int op = r->cigar[k] & 15; int64_t len = r->cigar[k] >> 4;
    if (op == 0) { m += len; nm++; } else if (op == 1) ins += len;
  }
  if (m + ins == 0 || nm == 0) return VSV_E_ZERODIV;
  return (100 * ins <= 13 * (m + ins)) || (m >= 200 * nm);
}

/* extract_sig_from_cigar (H:53-85 / RS:47-83 / SV/SVIM_intra.py:8-30) for one record and one hap
 * pass; appends the raw (pre-fold) signatures in CIGAR order. */
/* hapmask: bit0 = emit an hp1-pass row, bit1 = emit an hp2-pass row (a qname holding both 'hp1' and 'hp2' is
 * walked by both passes of H:760-766; T_RAW order is (record, op, hap)). */
static int walk_record(const vsv_records* r, int64_t i, const vsv_params* p, uint32_t hapmask, sigvec* out) {
  uint64_t a = r->cigar_off[i], b = r->cigar_off[i + 1];
  if (b <= a) return VSV_E_EMPTY_CIGAR;
  int dtype = p->dtype;
  int64_t off_ref = r->pos[i], off_q = 0, hc = 0;
  if (dtype != VSV_DTYPE_SVIM && dtype != VSV_DTYPE_CUTESV && (r->cigar[a] & 15) == 5) hc = r->cigar[a] >> 4; /* H:63-65 */
  for (uint64_t k = a; k < b; ++k) {
    int op = r->cigar[k] & 15; int64_t len = r->cigar[k] >> 4;
    if (op == 2 || op == 1) {
      if (len >= p->min_svlen) {
        vsv_sig s; memset(&s, 0, sizeof s);
        s.pos = (int32_t)off_ref; s.svlen = (int32_t)len; s.q_start = (int32_t)(off_q + hc);
        s.rec = (uint32_t)i; s.rec2 = 0xFFFFFFFFu; s.tid = r->tid[i];
        s.meta = (op == 2 ? VSV_M_DEL : 0);
        if (dtype == VSV_DTYPE_READS || dtype == VSV_DTYPE_CUTESV) s.q_end = 0;   /* RS:72,76: 8-field sig, no q_end */
        else s.q_end = s.q_start + (op == 2 ? 1 : (int32_t)len); /* H:80, H:84 */
        if (hapmask & 1) sv_push(out, &s);
        if (hapmask & 2) { s.meta |= VSV_M_HP2; sv_push(out, &s); }
      }
    }
    if (ref_adv(dtype, op)) off_ref += len;
    if (qry_adv(dtype, op)) off_q += len;
  }
  /* assert offset_ref==read.reference_end (H:396, RS:123) */
  if (dtype != VSV_DTYPE_SVIM && dtype != VSV_DTYPE_CUTESV && off_ref != ref_end_pysam(r, i)) return VSV_E_REFEND;
  return 0;
}

/* cluster_ins_one_read / cluster_del_one_read (H:91-161): left fold over one record's signatures
 * of one type, in place on raw[lo,hi) (all same rec & hap). Dead rows get VSV_M_DEAD. */
static void fold_record_hap(vsv_sig* s, int64_t lo, int64_t hi, uint32_t hap) {
  int64_t last_ins = -1, last_del = -1;
  for (int64_t k = lo; k < hi; ++k) {
    if ((s[k].meta & VSV_M_HP2) != hap) continue;
    if (s[k].meta & VSV_M_DEL) {
      if (last_del < 0) { last_del = k; continue; }
      vsv_sig* s1 = &s[last_del]; vsv_sig* s2 = &s[k];
      int64_t d = (int64_t)s2->pos - s1->pos; if (d < 0) d = -d;
      if (s1->svlen > 150 && s2->svlen > 150 && d < 150) {     /* H:148-150 */
        s1->svlen = s2->pos + s2->svlen - s1->pos;              /* H:104 */
        s1->q_end = s1->q_start + 1;                            /* H:101-102 */
        s2->meta |= VSV_M_DEAD;
      } else last_del = k;
    } else {
      if (last_ins < 0) { last_ins = k; continue; }
      vsv_sig* s1 = &s[last_ins]; vsv_sig* s2 = &s[k];
      int64_t d = (int64_t)s2->pos - s1->pos; if (d < 0) d = -d;
      int m = (s1->svlen > 250 && s2->svlen > 250 && d < 250) ||  /* H:115-117 */
              (s1->svlen > 320 && s2->svlen > 320 && d < 380) ||  /* H:120-122 */
              (s1->svlen > 100 && s2->svlen > 100 && d < 250);    /* H:126-128 */
      if (m) {
        s1->q_end = s2->q_end;                                   /* H:94 read_end = sig2[6] */
        s1->svlen = s1->q_end - s1->q_start;                     /* H:96 */
        s2->meta |= VSV_M_DEAD;
      } else last_ins = k;
    }
  }
}

static void fold_record(vsv_sig* s, int64_t lo, int64_t hi) {
  fold_record_hap(s, lo, hi, 0);
  fold_record_hap(s, lo, hi, VSV_M_HP2);
}

/* ------------------------------------------------------------------------------------------------
 * split-alignment pair rule (H:307-371, O:307-382, C:328-402, RS:147-197)
 * ------------------------------------------------------------------------------------------------ */
static int split_pair(const vsv_records* r, int64_t i1, int64_t i2, const vsv_params* p, uint32_t hapbit,
                      int min_mapq, sigvec* out) {
  int dtype = p->dtype;
  if (r->pos[i1] > r->pos[i2]) return VSV_E_UNSORTED;            /* H:315 */
  int rev1 = r->flag[i1] & VSV_F_REVERSE, rev2 = r->flag[i2] & VSV_F_REVERSE;
  uint32_t last1 = r->cigar[r->cigar_off[i1 + 1] - 1], first2 = r->cigar[r->cigar_off[i2]];
  int lop = last1 & 15, fop = first2 & 15;
  if (!((rev1 == rev2) && r->mapq[i1] >= min_mapq && r->mapq[i2] >= min_mapq &&
        (lop == 4 || lop == 5) && (fop == 4 || fop == 5)))
    return 0;                                                    /* H:323-324 */
  int64_t rl1 = read_len(r, i1, dtype), rl2 = read_len(r, i2, dtype);
  if (rl1 != rl2) return VSV_E_READLEN;                          /* H:331 */
  int64_t Ref1e = ref_end_pysam(r, i1), Ref2s = r->pos[i2];
  int64_t Read1e = rl1 - (last1 >> 4), Read2s = first2 >> 4;
  int64_t Diffdis = (Ref2s - Ref1e) - (Read2s - Read1e);
  int64_t absd = Diffdis < 0 ? -Diffdis : Diffdis;
  if (absd > p->max_split_svlen) return 0;                       /* H:354 */
  vsv_sig s; memset(&s, 0, sizeof s);
  s.rec = (uint32_t)i1; s.rec2 = (uint32_t)i2; s.tid = r->tid[i1];
  s.meta = hapbit | VSV_M_SPLIT;
  if (dtype == VSV_DTYPE_HIFI) {
    if (Diffdis >= 30) {
      int64_t Diffolp = Read1e - Read2s, ao = Diffolp < 0 ? -Diffolp : Diffolp;
      if (ao <= 3000) {                                          /* H:357 */
        int64_t h = Diffolp / 2;                                 /* int(Diffolp/2): trunc toward 0 */
        s.pos = (int32_t)(Ref1e - h); s.svlen = (int32_t)Diffdis;
        s.q_start = (int32_t)(Read1e - h); s.q_end = s.q_start + 1; s.meta |= VSV_M_DEL;
        sv_push(out, &s);
      }
    } else if (Diffdis <= -30) {
      int64_t Diffolp = Ref1e - Ref2s, ao = Diffolp < 0 ? -Diffolp : Diffolp;
      if (Diffolp < 3000) {                                      /* H:362 */
        int64_t sv = Read2s - Read1e + Diffolp; if (sv < 0) sv = -sv;
        s.pos = (int32_t)(ao > 400 ? (Ref1e + Ref2s) / 2 : Ref2s);
        s.svlen = (int32_t)sv; s.q_start = (int32_t)(Read1e - Diffolp); s.q_end = (int32_t)Read2s;
        sv_push(out, &s);
      }
    }
  } else if (dtype == VSV_DTYPE_ONT || dtype == VSV_DTYPE_CLR) {
    double r_ = dtype == VSV_DTYPE_ONT ? 0.5 : 0.3;              /* O:348, C:369 */
    double lo_f = dtype == VSV_DTYPE_ONT ? 0.8 : 0.3;            /* O:373 Diffdis*0.8 ; C:377 Diffdis*r */
    if (Diffdis >= 30) {
      int64_t Diffolp = Read1e - Read2s;
      double dr = (double)Diffdis * r_;
      if (-dr <= (double)Diffolp && (double)Diffolp <= dr) {     /* O:354 */
        s.pos = (int32_t)Ref1e; s.svlen = (int32_t)Diffdis;
        s.q_start = (int32_t)Read1e; s.q_end = (int32_t)Read2s; s.meta |= VSV_M_DEL;
        sv_push(out, &s);
      }
    } else {
      int64_t Diffolp = Ref1e - Ref2s, ao = Diffolp < 0 ? -Diffolp : Diffolp;
      double lo = (double)Diffdis * lo_f, hi = (double)absd * r_;
      if (lo <= (double)Diffolp && (double)Diffolp <= hi && Diffdis <= -30) { /* O:373 */
        int64_t sv = Read2s - Read1e + Diffolp; if (sv < 0) sv = -sv;
        s.pos = (int32_t)(ao > 400 ? (Ref1e + Ref2s) / 2 : Ref2s);
        s.svlen = (int32_t)sv; s.q_start = (int32_t)(Read1e - Diffolp); s.q_end = (int32_t)Read2s;
        sv_push(out, &s);
      }
    }
  } else { /* READS RS:179-196 */
    int64_t Diffolp = Ref1e - Ref2s;
    if (Diffolp < 30 && Diffdis >= 30) {
      s.pos = (int32_t)Ref1e; s.svlen = (int32_t)Diffdis; s.q_start = (int32_t)Read1e; s.q_end = (int32_t)Read2s;
      s.meta |= VSV_M_DEL; sv_push(out, &s);
    } else if (Diffolp < 30 && Diffdis <= -30) {
      s.pos = (int32_t)((Ref1e + Ref2s) / 2); s.svlen = (int32_t)absd;
      s.q_start = (int32_t)Read1e; s.q_end = (int32_t)Read2s; sv_push(out, &s);
    }
  }
  return 0;
}

/* extract_sig_from_split_reads (H:421-457, RS:199-237) for one (tid range, hap pass).
 * Names counted among eligible records; groups with count>1 in first-appearance order; consecutive
 * pairs in BAM order. */
static int split_pass(const vsv_records* r, int64_t lo, int64_t hi, const vsv_params* p, uint32_t hapbit,
                      uint32_t hapflag, int min_mapq, sigvec* out) {
  int nq = r->n_qids;
  int32_t* cnt = (int32_t*)calloc((size_t)nq + 1, sizeof(int32_t));
  int64_t* head = (int64_t*)malloc(((size_t)nq + 1) * sizeof(int64_t));
  int64_t* tail = (int64_t*)malloc(((size_t)nq + 1) * sizeof(int64_t));
  int64_t* next = (int64_t*)malloc((size_t)(hi - lo + 1) * sizeof(int64_t));
  int64_t* order = (int64_t*)malloc(((size_t)nq + 1) * sizeof(int64_t));
  int64_t n_order = 0;
  int st = 0;
  for (int64_t i = lo; i < hi; ++i) {
    int elig = (hapflag == 0) ? 1 : ((r->flag[i] & hapflag) && r->mapq[i] >= min_mapq);
    if (!elig) continue;
    uint32_t q = r->qid[i];
    if (cnt[q] == 0) { head[q] = i; order[n_order++] = q; } else next[tail[q] - lo] = i;
    tail[q] = i; next[i - lo] = -1; cnt[q]++;
  }
  for (int64_t g = 0; g < n_order && st == 0; ++g) {
    uint32_t q = (uint32_t)order[g];
    if (cnt[q] < 2) continue;
    for (int64_t i = head[q]; next[i - lo] >= 0 && st == 0; i = next[i - lo])
      st = split_pair(r, i, next[i - lo], p, hapbit, min_mapq, out);
  }
  free(cnt); free(head); free(tail); free(next); free(order);
  return st;
}

/* ------------------------------------------------------------------------------------------------
 * stable sort of sig rows by a 64-bit key (sort_sig H:170-179 with the canonical stable tie rule)
 * ------------------------------------------------------------------------------------------------ */
typedef uint64_t (*keyfn)(const vsv_sig*);
static uint64_t upos(int32_t pos) { return (uint32_t)(pos ^ 0x80000000); }
/* list = (tid, hap, type, src): per-source lists of H:402-415 / H:459-473 */
static uint64_t key_stage1(const vsv_sig* s) {
  uint64_t hap = (s->meta & VSV_M_HP2) ? 1 : 0, del = (s->meta & VSV_M_DEL) ? 1 : 0, sp = (s->meta & VSV_M_SPLIT) ? 1 : 0;
  return ((uint64_t)(uint32_t)s->tid << 35) | (hap << 34) | (del << 33) | (sp << 32) | upos(s->pos);
}
/* list = (tid, hap, type): merge_sig_ins / merge_sig_del H:478-490 */
static uint64_t key_stage2(const vsv_sig* s) {
  uint64_t hap = (s->meta & VSV_M_HP2) ? 1 : 0, del = (s->meta & VSV_M_DEL) ? 1 : 0;
  return ((uint64_t)(uint32_t)s->tid << 35) | (hap << 34) | (del << 33) | upos(s->pos);
}
/* list = (tid, hap): final sort of merge_all H:495 */
static uint64_t key_stage3(const vsv_sig* s) {
  uint64_t hap = (s->meta & VSV_M_HP2) ? 1 : 0;
  return ((uint64_t)(uint32_t)s->tid << 35) | (hap << 34) | upos(s->pos);
}
static uint64_t key_tidpos(const vsv_sig* s) { return ((uint64_t)(uint32_t)s->tid << 35) | upos(s->pos); }

static void stable_sort_idx(const uint64_t* key, int64_t* idx, int64_t n) {
  int64_t* tmp = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
  for (int64_t w = 1; w < n; w *= 2) {
    for (int64_t lo = 0; lo < n; lo += 2 * w) {
      int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
      int64_t a = lo, b = mid, k = lo;
      while (a < mid && b < hi) tmp[k++] = (key[idx[b]] < key[idx[a]]) ? idx[b++] : idx[a++];
      while (a < mid) tmp[k++] = idx[a++];
      while (b < hi) tmp[k++] = idx[b++];
    }
    memcpy(idx, tmp, (size_t)n * sizeof(int64_t));
  }
  free(tmp);
}

static void sort_sigs(vsv_sig* s, int64_t n, keyfn kf) {
  if (n <= 1) return;
  uint64_t* key = (uint64_t*)malloc((size_t)n * sizeof(uint64_t));
  int64_t* idx = (int64_t*)malloc((size_t)n * sizeof(int64_t));
  vsv_sig* t = (vsv_sig*)malloc((size_t)n * sizeof(vsv_sig));
  for (int64_t i = 0; i < n; ++i) { key[i] = kf(&s[i]); idx[i] = i; }
  stable_sort_idx(key, idx, n);
  for (int64_t i = 0; i < n; ++i) t[i] = s[idx[i]];
  memcpy(s, t, (size_t)n * sizeof(vsv_sig));
  free(key); free(idx); free(t);
}

/* ------------------------------------------------------------------------------------------------
 * cluster_del / cluster_ins (H:196-288) on one sorted list s[lo,hi). Predicates in exact integer
 * form (SURVEY §7 hard part 3). literal=1 keeps the reference's full inner scan; literal=0 starts at
 * i+1 and stops once pos_j - pos_i > max_shift (identical result on a list sorted by pos).
 * ------------------------------------------------------------------------------------------------ */
static inline int match_sig(const vsv_sig* a, const vsv_sig* b, int max_shift) {
  int64_t shift = (int64_t)a->pos - b->pos; if (shift < 0) shift = -shift;
  if (shift > max_shift) return 0;
  int64_t l1 = a->svlen, l2 = b->svlen, mn = l1 < l2 ? l1 : l2, mx = l1 < l2 ? l2 : l1;
  if (2 * mn < mx) return 0;                                   /* size_similarity >= 0.5 */
  if (a->meta & VSV_M_DEL) {
    int64_t s1 = a->pos, e1 = s1 + l1, s2 = b->pos, e2 = s2 + l2;
    int64_t ov = (e1 < e2 ? e1 : e2) - (s1 > s2 ? s1 : s2);
    if (2 * ov < mn) return 0;                                 /* overlap_ratio >= 0.5 (H:200-204) */
  }
  return 1;
}

static void cluster_list(const vsv_sig* s, int64_t lo, int64_t hi, int max_shift, int literal, sigvec* out) {
  int64_t n = hi - lo;
  if (n <= 0) return;
  int64_t* cl = (int64_t*)malloc((size_t)n * sizeof(int64_t));
  for (int64_t i = 0; i < n; ++i) cl[i] = -1;
  for (int64_t i = 0; i < n; ++i) {
    if (cl[i] != -1) continue;
    cl[i] = i;                                                  /* H:209-210 */
    const vsv_sig* s1 = &s[lo + i];
    int64_t best = i;
    for (int64_t j = literal ? 0 : i + 1; j < n; ++j) {
      if (!literal && (int64_t)s[lo + j].pos - s1->pos > max_shift) break;
      if (cl[j] != -1) continue;
      if (match_sig(s1, &s[lo + j], max_shift)) {
        cl[j] = i;
        if (s[lo + j].svlen > s[lo + best].svlen) best = j;     /* H:239-246: first longest, index order */
      }
    }
    if (literal) { /* representative scan in index order over members, as H:236-247 */
      best = -1;
      for (int64_t j = 0; j < n; ++j) if (cl[j] == i && (best < 0 || s[lo + j].svlen > s[lo + best].svlen)) best = j;
    }
    sv_push(out, &s[lo + best]);
  }
  free(cl);
}

static void cluster_all(const vsv_sig* s, int64_t n, keyfn listkey, int max_shift, int literal, sigvec* out) {
  int64_t lo = 0;
  while (lo < n) {
    int64_t hi = lo + 1;
    uint64_t k = listkey(&s[lo]) >> 32;
    while (hi < n && (listkey(&s[hi]) >> 32) == k) hi++;
    cluster_list(s, lo, hi, max_shift, literal, out);
    lo = hi;
  }
}

/* ------------------------------------------------------------------------------------------------
 * pair_sig (H:548-603) for one tid: hp1 list a[0,na), hp2 list b[0,nb), both sorted by pos.
 * ------------------------------------------------------------------------------------------------ */
typedef struct { vsv_call* v; int64_t n, cap; } callvec;
static void cv_push(callvec* c, const vsv_call* x) {
  if (c->n == c->cap) { c->cap = c->cap ? c->cap * 2 : 1024; c->v = (vsv_call*)realloc(c->v, (size_t)c->cap * sizeof(vsv_call)); }
  c->v[c->n++] = *x;
}

static void pair_tid(const vsv_sig* m, int64_t a0, int64_t na, int64_t b0, int64_t nb, const vsv_params* p,
                     int literal, callvec* out) {
  int64_t* st1 = (int64_t*)malloc((size_t)(na + 1) * sizeof(int64_t));
  int64_t* st2 = (int64_t*)malloc((size_t)(nb + 1) * sizeof(int64_t));
  for (int64_t i = 0; i < na; ++i) st1[i] = -1;
  for (int64_t j = 0; j < nb; ++j) st2[j] = -1;
  int64_t jlo = 0;
  for (int64_t i = 0; i < na; ++i) {
    const vsv_sig* s1 = &m[a0 + i];
    if (!literal) while (jlo < nb && (int64_t)s1->pos - m[b0 + jlo].pos > p->pair_shift) jlo++;
    for (int64_t j = literal ? 0 : jlo; j < nb; ++j) {
      const vsv_sig* s2 = &m[b0 + j];
      int64_t dist = (int64_t)s2->pos - s1->pos;
      if (dist > p->pair_window) break;                         /* H:557-559 */
      if (((s1->meta ^ s2->meta) & VSV_M_DEL) == 0 && st2[j] == -1) { /* H:560 */
        if (match_sig(s1, s2, p->pair_shift)) { st1[i] = j; st2[j] = i; break; } /* H:561-569 */
      }
    }
  }
  for (int64_t i = 0; i < na; ++i) {                             /* H:571-586 */
    vsv_call c; memset(&c, 0, sizeof c);
    const vsv_sig* s1 = &m[a0 + i];
    if (st1[i] == -1) { c.sig = *s1; c.a = (int32_t)(a0 + i); c.b = -1; c.gt = 1; }
    else {
      const vsv_sig* s2 = &m[b0 + st1[i]];
      c.sig = (s1->svlen > s2->svlen) ? *s1 : *s2;
      c.a = (int32_t)(a0 + i); c.b = (int32_t)(b0 + st1[i]); c.gt = 2;
    }
    cv_push(out, &c);
  }
  for (int64_t j = 0; j < nb; ++j)                               /* H:588-592 */
    if (st2[j] == -1) {
      vsv_call c; memset(&c, 0, sizeof c);
      c.sig = m[b0 + j]; c.a = -1; c.b = (int32_t)(b0 + j); c.gt = 1;
      cv_push(out, &c);
    }
  free(st1); free(st2);
}

static void sort_calls(vsv_call* c, int64_t n) {
  if (n <= 1) return;
  uint64_t* key = (uint64_t*)malloc((size_t)n * sizeof(uint64_t));
  int64_t* idx = (int64_t*)malloc((size_t)n * sizeof(int64_t));
  vsv_call* t = (vsv_call*)malloc((size_t)n * sizeof(vsv_call));
  for (int64_t i = 0; i < n; ++i) { key[i] = key_tidpos(&c[i].sig); idx[i] = i; }
  stable_sort_idx(key, idx, n);
  for (int64_t i = 0; i < n; ++i) t[i] = c[idx[i]];
  memcpy(c, t, (size_t)n * sizeof(vsv_call));
  free(key); free(idx); free(t);
}

/* ------------------------------------------------------------------------------------------------
 * whole path. For contig dtypes: per-chromosome body of H:742-772 without VCF text. For READS:
 * RS:268-286. For SVIM: CIGAR stage only (SV/SVIM_intra.py:8-30).
 * ------------------------------------------------------------------------------------------------ */
static vsv_sig* sv_dup(const sigvec* s) {
  vsv_sig* v = (vsv_sig*)malloc((size_t)(s->n > 0 ? s->n : 1) * sizeof(vsv_sig));
  if (s->n) memcpy(v, s->v, (size_t)s->n * sizeof(vsv_sig));
  return v;
}

/* generate_combine_sigs (SE:373-435), per read and per type over the raw rows in CIGAR order. The merged signal lives in
 * the row of its first piece: svlen = summed length, q_end = number of pieces, rec2 = raw index of the first piece. */
static void combine_in_read(vsv_sig* s, int64_t n, int merge_ins, int merge_del) {
  int64_t lo = 0;
  while (lo < n) {
    int64_t hi = lo + 1;
    while (hi < n && s[hi].rec == s[lo].rec) hi++;
    for (int type = 0; type < 2; ++type) {                     /* SE:476-477: INS list, then DEL list */
      int64_t temp = -1, cmp = 0; int first_group = 1;
      for (int64_t k = lo; k < hi; ++k) {
        if (((s[k].meta & VSV_M_DEL) != 0) != (type == 1)) continue;
        if (temp >= 0 && (int64_t)s[k].pos - cmp <= (type ? merge_del : merge_ins)) {   /* SE:395, 417 */
          s[temp].svlen += s[k].svlen; s[temp].q_end += 1;                              /* SE:396, 418 */
          cmp = type ? (int64_t)s[k].pos + s[k].svlen : (int64_t)s[k].pos;             /* SE:398, 419 */
          s[k].meta |= VSV_M_DEAD;
        } else {
          /* new temp_sig: SE:392-393 / 414 for the first, SE:406-407 / 427-428 (append(i[0])) afterwards */
          cmp = (type && first_group) ? (int64_t)s[k].pos + s[k].svlen : (int64_t)s[k].pos;
          first_group = 0;
          temp = k; s[k].q_end = 1; s[k].rec2 = (uint32_t)k;
        }
      }
    }
    lo = hi;
  }
}

int orc_run(const vsv_records* r, const vsv_params* p, int literal, orc_out* o) {
  memset(o, 0, sizeof *o);
  int dtype = p->dtype;
  int contig = dtype == VSV_DTYPE_HIFI || dtype == VSV_DTYPE_ONT || dtype == VSV_DTYPE_CLR;
  sigvec raw = {0}, cig = {0}, spl = {0}, c1 = {0}, mg = {0};
  callvec calls = {0};
  int st = 0;

  /* ---- CIGAR stage (H:386-400 / C:422-433 / RS:107-125) ---- */
  for (int64_t i = 0; i < r->n_records && st == 0; ++i) {
    if (r->cigar_off[i + 1] <= r->cigar_off[i]) { st = VSV_E_EMPTY_CIGAR; break; }
    if (contig) {
      if (!(r->flag[i] & (VSV_F_HP1 | VSV_F_HP2))) continue;
      int gate = 1;
      if (dtype == VSV_DTYPE_CLR) { gate = clr_gate(r, i); if (gate < 0) { st = gate; break; } }
      if (r->mapq[i] < p->min_cigar_mapq || !gate) continue;
      st = walk_record(r, i, p, (r->flag[i] >> 2) & 3u, &raw);
      if (st == 0 && (r->flag[i] & VSV_F_SEQ_MISMATCH)) st = VSV_E_SEQLEN;      /* H:397-398: after the reference_end assert */
    } else if (dtype == VSV_DTYPE_READS) {
      if (r->mapq[i] < p->min_cigar_mapq) continue;            /* RS:120 */
      st = walk_record(r, i, p, 1, &raw);
      if (st == 0 && (r->flag[i] & VSV_F_SEQ_MISMATCH)) st = VSV_E_SEQLEN;      /* RS:123-124 */
    } else if (dtype == VSV_DTYPE_CUTESV) {
      if ((r->flag[i] & VSV_F_SKIP) || r->mapq[i] < p->min_cigar_mapq) continue;   /* SE:439, 446 */
      st = walk_record(r, i, p, 1, &raw);
    } else {
      if ((r->flag[i] & (VSV_F_UNMAPPED | VSV_F_SECONDARY)) || r->mapq[i] < p->min_cigar_mapq) continue; /* SV/SVIM_COLLECT.py:67 */
      st = walk_record(r, i, p, 1, &raw);
    }
  }
  o->raw = sv_dup(&raw); o->n_raw = raw.n;

  /* ---- intra-read fold (H:108-161): contig dtypes only ---- */
  if (contig) {
    int64_t lo = 0;
    while (lo < raw.n) {
      int64_t hi = lo + 1;
      while (hi < raw.n && raw.v[hi].rec == raw.v[lo].rec) hi++;
      fold_record(raw.v, lo, hi);
      lo = hi;
    }
  }
  if (dtype == VSV_DTYPE_CUTESV) combine_in_read(raw.v, raw.n, p->merge_ins_threshold, p->merge_del_threshold);
  for (int64_t k = 0; k < raw.n; ++k) if (!(raw.v[k].meta & VSV_M_DEAD)) sv_push(&cig, &raw.v[k]);
  o->cigar = sv_dup(&cig); o->n_cigar = cig.n;

  /* ---- split stage, per tid (fetch(chr) H:424) and hap pass ---- */
  if (st == 0 && p->enable_split && dtype != VSV_DTYPE_SVIM) {
    int64_t lo = 0;
    while (lo < r->n_records && st == 0) {
      int64_t hi = lo + 1;
      while (hi < r->n_records && r->tid[hi] == r->tid[lo]) hi++;
      if (contig) {
        st = split_pass(r, lo, hi, p, 0, VSV_F_HP1, p->min_split_mapq, &spl);
        if (st == 0) st = split_pass(r, lo, hi, p, VSV_M_HP2, VSV_F_HP2, p->min_split_mapq, &spl);
      } else st = split_pass(r, lo, hi, p, 0, 0, p->min_split_mapq, &spl);   /* RS:199-223 collects every record; RS:235 min_mapq (0) only enters the pair test */
      lo = hi;
    }
  }
  o->split = sv_dup(&spl); o->n_split = spl.n;

  if (dtype == VSV_DTYPE_READS) {
    /* RS:281-286: del_cigar_sorted + ins_cigar_sorted + del_split_sorted + ins_split_sorted, then
     * sort by pos. With a stable sort this is: key (tid, pos), ties by (DEL<INS within source,
     * cigar<split, list order). */
    sigvec all = {0};
    for (int pass = 0; pass < 4; ++pass) {
      const sigvec* src = pass < 2 ? &cig : &spl;
      int want_del = (pass % 2) == 0;
      sigvec tmp = {0};
      for (int64_t k = 0; k < src->n; ++k) if (((src->v[k].meta & VSV_M_DEL) != 0) == want_del) sv_push(&tmp, &src->v[k]);
      sort_sigs(tmp.v, tmp.n, key_tidpos);
      for (int64_t k = 0; k < tmp.n; ++k) sv_push(&all, &tmp.v[k]);
      free(tmp.v);
    }
    /* per-tid final sort: different tids are different files in the reference; tid-major here */
    sort_sigs(all.v, all.n, key_tidpos);
    o->reads = sv_dup(&all); o->n_reads = all.n; free(all.v);
  }

  if (contig && st == 0) {
    /* ---- stage 1: per-source sort + cluster (H:402-415, 459-473) ---- */
    sigvec s1 = {0};
    for (int64_t k = 0; k < cig.n; ++k) sv_push(&s1, &cig.v[k]);
    for (int64_t k = 0; k < spl.n; ++k) sv_push(&s1, &spl.v[k]);
    sort_sigs(s1.v, s1.n, key_stage1);
    cluster_all(s1.v, s1.n, key_stage1, p->cluster_shift, literal, &c1);
    o->cluster1 = sv_dup(&c1); o->n_cluster1 = c1.n;
    /* ---- stage 2: merge_sig_ins / merge_sig_del (H:478-490): cigar list + split list, sort, cluster ---- */
    sigvec s2 = {0}, c2 = {0};
    for (int64_t k = 0; k < c1.n; ++k) sv_push(&s2, &c1.v[k]);
    sort_sigs(s2.v, s2.n, key_stage2);
    cluster_all(s2.v, s2.n, key_stage2, p->cluster_shift, literal, &c2);
    /* ---- stage 3: sort_sig(ins_final + del_final) per hap (H:495) ---- */
    sort_sigs(c2.v, c2.n, key_stage3);
    for (int64_t k = 0; k < c2.n; ++k) sv_push(&mg, &c2.v[k]);
    o->merged = sv_dup(&mg); o->n_merged = mg.n;
    /* ---- pair_sig per tid (H:768) ---- */
    int64_t lo = 0;
    while (lo < mg.n) {
      int64_t hi = lo + 1;
      while (hi < mg.n && mg.v[hi].tid == mg.v[lo].tid) hi++;
      int64_t mid = lo;
      while (mid < hi && !(mg.v[mid].meta & VSV_M_HP2)) mid++;
      int64_t c0 = calls.n;
      pair_tid(mg.v, lo, mid - lo, mid, hi - mid, p, literal, &calls);
      sort_calls(calls.v + c0, calls.n - c0);                   /* H:594 */
      lo = hi;
    }
    o->calls = calls.v; o->n_calls = calls.n; calls.v = NULL;
    free(s1.v); free(s2.v); free(c2.v);
  }
  if (!o->cluster1) o->cluster1 = (vsv_sig*)malloc(sizeof(vsv_sig));
  if (!o->merged) o->merged = (vsv_sig*)malloc(sizeof(vsv_sig));
  if (!o->calls) o->calls = (vsv_call*)malloc(sizeof(vsv_call));
  if (!o->reads) o->reads = (vsv_sig*)malloc(sizeof(vsv_sig));
  free(raw.v); free(cig.v); free(spl.v); free(c1.v); free(mg.v); free(calls.v);
  o->status = st;
  return st;
}

void orc_free(orc_out* o) {
  free(o->raw); free(o->cigar); free(o->split); free(o->cluster1); free(o->merged); free(o->calls); free(o->reads);
  memset(o, 0, sizeof *o);
}

/* default parameters = hard-coded reference values (same table as vsv_default_params) */
int orc_default_params(int dtype, vsv_params* p) {
  memset(p, 0, sizeof *p);
  p->dtype = dtype;
  p->min_svlen = dtype == VSV_DTYPE_SVIM ? 40 : 30;   /* SV/SVIM_input_parsing.py min_sv_size 40 */
  p->min_cigar_mapq = dtype == VSV_DTYPE_SVIM ? 20 : 50;
  p->min_split_mapq = dtype == VSV_DTYPE_READS ? 0 : 50;
  p->max_split_svlen = 50000;
  p->cluster_shift = 100;
  p->pair_shift = 200;
  p->pair_window = 1000;
  p->enable_split = dtype == VSV_DTYPE_SVIM ? 0 : 1;
  if (dtype == VSV_DTYPE_CUTESV) {   /* SE:703-747 */
    p->min_svlen = 10; p->min_cigar_mapq = 20; p->enable_split = 0; p->merge_ins_threshold = 100; p->merge_del_threshold = 0;
  }
  return 0;
}

/* ================================================================================================
 * svim-asm breakend (BND) branch: analyze_read_segments BND cases (SV/SVIM_inter.py:62-258),
 * CandidateBreakend canonical form (SV/SVCandidate.py:350-373), form_partitions +
 * pair_haplotypes_breakends + BND part of pair_candidates (SV/SVIM_COMBINE.py:15-32,105-117,143-161,334-365).
 * Complete-linkage clustering at 0.3 with a 99999 distance for same-haplotype / different-direction pairs can only
 * form clusters of one hp1 + one hp2 member, merged in ascending distance order: restated as a greedy matching on
 * the integer distance |d1|+|d2| <= 900 (ties: lowest member indices first).
 * ================================================================================================ */
typedef struct { int32_t q_start, q_end, ref_id, ref_start, ref_end, rev, ord; } orc_seg;

static int seg_cmp(const void* a, const void* b) {
  const orc_seg* x = (const orc_seg*)a; const orc_seg* y = (const orc_seg*)b;
  if (x->q_start != y->q_start) return x->q_start < y->q_start ? -1 : 1;
  if (x->q_end != y->q_end) return x->q_end < y->q_end ? -1 : 1;
  return x->ord - y->ord;   /* Python sorted() is stable */
}

static int clampi(int64_t v, int64_t hi) { if (v < 0) v = 0; if (v > hi) v = hi; return (int)v; }

/* CandidateBreakend.__init__ (SV/SVCandidate.py:351-373) */
static vsv_bnd make_bnd(const vsv_segments* s, int t1, int64_t p1, int d1, int t2, int64_t p2, int d2, uint32_t read, int hap) {
  vsv_bnd b; memset(&b, 0, sizeof b);
  int keep = (s->contig_rank[t1] < s->contig_rank[t2]) || (t1 == t2 && p1 < p2);
  if (keep) {
    b.src_tid = t1; b.src_pos = clampi(p1, s->contig_len[t1]); b.dst_tid = t2; b.dst_pos = clampi(p2, s->contig_len[t2]);
    b.meta = (d1 ? VSV_B_SRC_FWD : 0) | (d2 ? VSV_B_DST_FWD : 0);
  } else {
    b.src_tid = t2; b.src_pos = clampi(p2, s->contig_len[t2]); b.dst_tid = t1; b.dst_pos = clampi(p1, s->contig_len[t1]);
    b.meta = (d2 ? 0 : VSV_B_SRC_FWD) | (d1 ? 0 : VSV_B_DST_FWD);   /* both directions flipped */
  }
  if (hap == 2) b.meta |= VSV_B_HAP2;
  b.read = read; b.read2 = 0xFFFFFFFFu;
  return b;
}

/* one adjacent pair of the (q_start,q_end)-sorted segment list; returns 1 and fills *out for a BND */
static int bnd_of_pair(const vsv_segments* s, const vsv_bnd_params* p, const orc_seg* cur, const orc_seg* nxt, uint32_t read, int hap, vsv_bnd* out) {
  const int64_t QOT = p->query_overlap_tolerance, QGT = p->query_gap_tolerance, ROT = p->reference_overlap_tolerance;
  const int64_t MINSV = p->min_sv_size, MAXSV = p->max_sv_size;
  const int64_t dor = (int64_t)nxt->q_start - cur->q_end;
  const int c1 = cur->ref_id, c2 = nxt->ref_id;
  if (c1 == c2) {
    if (cur->rev == nxt->rev) {
      const int64_t dref = cur->rev ? (int64_t)cur->ref_start - nxt->ref_end : (int64_t)nxt->ref_start - cur->ref_end;
      if (dor >= -QOT) {
        if (dref >= -ROT) {
          const int64_t dev = dor - dref;
          if (dev >= MINSV) return 0;                               /* INS candidate */
          if (-MAXSV <= dev && dev <= -MINSV) return 0;             /* DEL candidate */
          if (dev < -MAXSV && dor <= QGT) {                         /* :131-139 */
            *out = cur->rev ? make_bnd(s, c1, cur->ref_start, 0, c1, (int64_t)nxt->ref_end - 1, 0, read, hap)
                            : make_bnd(s, c1, (int64_t)cur->ref_end - 1, 1, c1, nxt->ref_start, 1, read, hap);
            return 1;
          }
        } else if (dor <= QGT) {                                    /* overlap on reference, :141-168 */
          const int64_t dev = dor - dref;
          if (dev >= MINSV) {
            if (!cur->rev) {
              if (nxt->ref_end > cur->ref_start) return 0;          /* tandem duplication */
              if (dref >= -MAXSV) return 0;
              *out = make_bnd(s, c1, (int64_t)cur->ref_end - 1, 1, c1, nxt->ref_start, 1, read, hap); return 1;
            } else {
              if (nxt->ref_start < cur->ref_end) return 0;
              if (dref >= -MAXSV) return 0;
              *out = make_bnd(s, c1, cur->ref_start, 0, c1, (int64_t)nxt->ref_end - 1, 0, read, hap); return 1;
            }
          }
        }
      }
      return 0;
    }
    if (!cur->rev && nxt->rev) {                                    /* :171-192 */
      const int64_t dref = (int64_t)nxt->ref_end - cur->ref_end, dev = dor - dref;
      if (-QOT <= dor && dor <= QGT) {
        if ((int64_t)nxt->ref_start - cur->ref_end >= -ROT) {
          if (MINSV <= -dev && -dev <= MAXSV) return 0;              /* inversion */
          *out = make_bnd(s, c1, (int64_t)cur->ref_end - 1, 1, c1, (int64_t)nxt->ref_end - 1, 0, read, hap); return 1;
        } else if ((int64_t)cur->ref_start - nxt->ref_end >= -ROT) {
          if (MINSV <= dev && dev <= MAXSV) return 0;
          *out = make_bnd(s, c1, (int64_t)cur->ref_end - 1, 1, c1, (int64_t)nxt->ref_end - 1, 0, read, hap); return 1;
        }
      }
      return 0;
    }
    {                                                               /* reverse to normal, :197-219 */
      const int64_t dref = (int64_t)nxt->ref_start - cur->ref_start, dev = dor - dref;
      if (-QOT <= dor && dor <= QGT) {
        if ((int64_t)nxt->ref_start - cur->ref_end >= -ROT) {
          if (MINSV <= -dev && -dev <= MAXSV) return 0;
          *out = make_bnd(s, c1, cur->ref_start, 0, c1, nxt->ref_start, 1, read, hap); return 1;
        } else if ((int64_t)cur->ref_start - nxt->ref_end >= -ROT) {
          if (MINSV <= dev && dev <= MAXSV) return 0;
          *out = make_bnd(s, c1, cur->ref_start, 0, c1, nxt->ref_start, 1, read, hap); return 1;
        }
      }
      return 0;
    }
  }
  if (dor >= -QOT && dor <= QGT) {                                  /* different chromosomes, :224-258 */
    if (cur->rev == nxt->rev)
      *out = cur->rev ? make_bnd(s, c1, cur->ref_start, 0, c2, (int64_t)nxt->ref_end - 1, 0, read, hap)
                      : make_bnd(s, c1, (int64_t)cur->ref_end - 1, 1, c2, nxt->ref_start, 1, read, hap);
    else
      *out = cur->rev ? make_bnd(s, c1, cur->ref_start, 0, c2, nxt->ref_start, 1, read, hap)
                      : make_bnd(s, c1, (int64_t)cur->ref_end - 1, 1, c2, (int64_t)nxt->ref_end - 1, 0, read, hap);
    return 1;
  }
  return 0;
}

typedef struct { vsv_bnd* v; int64_t n, cap; } bndvec;
static void bv_push(bndvec* b, const vsv_bnd* x) {
  if (b->n == b->cap) { b->cap = b->cap ? b->cap * 2 : 256; b->v = (vsv_bnd*)realloc(b->v, (size_t)b->cap * sizeof(vsv_bnd)); }
  b->v[b->n++] = *x;
}

int orc_default_bnd_params(vsv_bnd_params* p) {
  memset(p, 0, sizeof *p);
  p->min_sv_size = 40; p->max_sv_size = 100000; p->query_gap_tolerance = 50; p->query_overlap_tolerance = 50;
  p->reference_gap_tolerance = 50; p->reference_overlap_tolerance = 50; p->partition_max_distance = 1000;
  p->pair_distance = 900; p->max_partition = 10;
  return 0;
}

/* form_partitions + pair_haplotypes_breakends + pair_candidates (BND part) on a candidate table whose order is the
 * reference's collection order (hp1 candidates, then hp2 candidates). */
int orc_bnd_pair(const vsv_bnd* cv, int64_t n, const int32_t* contig_rank, const vsv_bnd_params* p, vsv_bnd** calls_out, int64_t* n_calls) {
  bndvec calls = {0};
  struct { const vsv_bnd* v; } cand = { cv };
  /* form_partitions: stable sort of hp1 candidates + hp2 candidates by (contig name, source_start) */
  int64_t* idx = (int64_t*)malloc((size_t)(n + 1) * sizeof(int64_t));
  uint64_t* key = (uint64_t*)malloc((size_t)(n + 1) * sizeof(uint64_t));
  int64_t m = 0;
  for (int hap = 0; hap < 2; ++hap)
    for (int64_t i = 0; i < n; ++i) if (((cand.v[i].meta & VSV_B_HAP2) != 0) == hap) idx[m++] = i;
  int64_t* order = (int64_t*)malloc((size_t)(n + 1) * sizeof(int64_t));
  for (int64_t i = 0; i < n; ++i) { const vsv_bnd* c = &cand.v[idx[i]]; key[i] = ((uint64_t)(uint32_t)contig_rank[c->src_tid] << 32) | (uint32_t)c->src_pos; order[i] = i; }
  stable_sort_idx(key, order, n);
  int64_t lo = 0;
  while (lo < n) {
    int64_t hi = lo + 1;
    while (hi < n) {
      const vsv_bnd* x = &cand.v[idx[order[hi - 1]]]; const vsv_bnd* y = &cand.v[idx[order[hi]]];
      int64_t d = (int64_t)x->src_pos - y->src_pos; if (d < 0) d = -d;
      if (x->src_tid != y->src_tid || d > p->partition_max_distance) break;
      hi++;
    }
    int64_t sz = hi - lo;
    if (sz <= p->max_partition && sz <= 64) {
      int used[64]; int mate[64];
      for (int64_t k = 0; k < sz; ++k) { used[k] = 0; mate[k] = -1; }
      for (;;) {                                   /* greedy matching in ascending distance */
        int64_t best = -1; int bi = -1, bj = -1;
        for (int i = 0; i < sz; ++i) for (int j = i + 1; j < sz; ++j) {
          if (used[i] || used[j]) continue;
          const vsv_bnd* x = &cand.v[idx[order[lo + i]]]; const vsv_bnd* y = &cand.v[idx[order[lo + j]]];
          if (((x->meta ^ y->meta) & VSV_B_HAP2) == 0) continue;                       /* same haplotype: 99999 */
          if ((x->meta ^ y->meta) & (VSV_B_SRC_FWD | VSV_B_DST_FWD)) continue;         /* different directions: 99999 */
          int64_t d1 = (int64_t)x->src_pos - y->src_pos, d2 = (int64_t)x->dst_pos - y->dst_pos;
          if (d1 < 0) d1 = -d1; if (d2 < 0) d2 = -d2;
          if (d1 + d2 > p->pair_distance) continue;
          if (best < 0 || d1 + d2 < best) { best = d1 + d2; bi = i; bj = j; }
        }
        if (bi < 0) break;
        used[bi] = used[bj] = 1; mate[bi] = bj; mate[bj] = bi;
      }
      for (int i = 0; i < sz; ++i) {
        if (mate[i] >= 0 && mate[i] < i) continue;                                     /* second member of a pair */
        vsv_bnd c = cand.v[idx[order[lo + i]]];
        uint32_t gt = mate[i] >= 0 ? 3u : ((c.meta & VSV_B_HAP2) ? 2u : 1u);
        if (mate[i] >= 0) c.read2 = cand.v[idx[order[lo + mate[i]]]].read;
        c.meta = (c.meta & ~(3u << VSV_B_GT_SHIFT)) | (gt << VSV_B_GT_SHIFT);
        bv_push(&calls, &c);
      }
    }
    lo = hi;
  }
  free(idx); free(key); free(order);
  if (!calls.v) calls.v = (vsv_bnd*)malloc(sizeof(vsv_bnd));
  *calls_out = calls.v; *n_calls = calls.n;
  return 0;
}

/* candidates in (read, pair) order; calls in (partition, first member) order */
int orc_bnd(const vsv_segments* s, const vsv_bnd_params* p, vsv_bnd** cand_out, int64_t* n_cand, vsv_bnd** calls_out, int64_t* n_calls) {
  bndvec cand = {0};
  for (int64_t r = 0; r < s->n_reads; ++r) {
    int64_t a = (int64_t)s->seg_off[r], b = (int64_t)s->seg_off[r + 1], n = b - a;
    if (n < 2) continue;
    orc_seg* sg = (orc_seg*)malloc((size_t)n * sizeof(orc_seg));
    for (int64_t k = 0; k < n; ++k) {
      sg[k].q_start = s->q_start[a + k]; sg[k].q_end = s->q_end[a + k]; sg[k].ref_id = s->ref_id[a + k];
      sg[k].ref_start = s->ref_start[a + k]; sg[k].ref_end = s->ref_end[a + k]; sg[k].rev = s->is_reverse[a + k] ? 1 : 0; sg[k].ord = (int32_t)k;
    }
    qsort(sg, (size_t)n, sizeof(orc_seg), seg_cmp);
    for (int64_t k = 0; k + 1 < n; ++k) {
      vsv_bnd x;
      if (bnd_of_pair(s, p, &sg[k], &sg[k + 1], (uint32_t)r, s->hap[r], &x)) bv_push(&cand, &x);
    }
    free(sg);
  }
  int rc = orc_bnd_pair(cand.v, cand.n, s->contig_rank, p, calls_out, n_calls);
  if (!cand.v) cand.v = (vsv_bnd*)malloc(sizeof(vsv_bnd));
  *cand_out = cand.v; *n_cand = cand.n;
  return rc;
}
void orc_bnd_free(vsv_bnd* a, vsv_bnd* b) { free(a); free(b); }

/* ================================================================================================
 * Post-filter: read-signature support of the calls — FP_filter_v1.eval_sig / compare_sigs
 * (Large_INDEL/FP_filter_v1.py:87-123), the literal double loop with its `continue` / `break`.
 * Returns VSV_E_UNSORTED if sig_pos does not ascend (the product path requires the sorted file the reference writes,
 * RS:281-286); the counts are the literal ones either way.
 * ================================================================================================ */
int orc_default_support_params(vsv_support_params* p) {
  p->max_comp_svlen = 250; p->max_dist = 1000; p->max_shift = 500; p->pad = 0; p->min_size_sim = 0.5;   /* FP:8-11 */
  return 0;
}

int orc_support(const int32_t* call_pos, const int32_t* call_len, int64_t n_calls, const int32_t* sig_pos, const int32_t* sig_len,
                int64_t n_sigs, const vsv_support_params* p, uint32_t* support) {
  int sorted = 1;
  for (int64_t j = 0; j + 1 < n_sigs; ++j) if (sig_pos[j] > sig_pos[j + 1]) sorted = 0;
  for (int64_t i = 0; i < n_calls; ++i) {
    if (call_len[i] > p->max_comp_svlen) { support[i] = 60; continue; }                 /* FP:110-111 */
    uint32_t s = 0;
    for (int64_t j = 0; j < n_sigs; ++j) {
      const int64_t shift = (int64_t)sig_pos[j] - call_pos[i];                           /* FP:115 */
      if (shift < -(int64_t)p->max_dist) continue;                                       /* FP:116-117 */
      if (shift > (int64_t)p->max_dist) break;                                           /* FP:118-119 */
      const int64_t a = shift < 0 ? -shift : shift;
      const int32_t l1 = call_len[i], l2 = sig_len[j];
      const int32_t mn = l1 < l2 ? l1 : l2, mx = l1 < l2 ? l2 : l1;
      const double sim = mx == 0 ? 0.0 : (double)mn / (double)mx;                        /* FP:93-96 */
      if (a <= p->max_shift && sim >= p->min_size_sim) ++s;                             /* FP:98-101 */
    }
    support[i] = s;
  }
  return sorted ? 0 : VSV_E_UNSORTED;
}

/* ================================================================================================
 * calculate_signature_support.py (Large_INDEL): signature coverage of the calls, restated with the reference's own
 * loop structure (sorted unique positions, moving start index, four boundary scans + set()).
 * Canonical call order: stable by start (the reference's np.argsort is unstable and it then mixes sorted and original
 * call indices, CS:171, 213-241; on a position-sorted VCF without ties both coincide with this).
 * ================================================================================================ */
typedef struct { int64_t key; int64_t val; } orc_kv;
static int orc_kv_cmp(const void* a, const void* b) {
  const orc_kv *x = (const orc_kv*)a, *y = (const orc_kv*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->val < y->val ? -1 : (x->val > y->val ? 1 : 0);
}

/* calc_ins_call_cov (CS:81-125): cov per call = its bin's weight */
int orc_cov_ins(const int32_t* call_pos, int64_t n_calls, const int32_t* sig_pos, const int32_t* sig_len, int64_t n_sigs,
                int32_t flanking, int64_t* cov) {
  /* uniq_pos + weights_list (CS:85-90) */
  orc_kv* sg = (orc_kv*)malloc(sizeof(orc_kv) * (size_t)(n_sigs + 1));
  for (int64_t j = 0; j < n_sigs; ++j) { sg[j].key = sig_pos[j]; sg[j].val = sig_len[j]; }
  qsort(sg, (size_t)n_sigs, sizeof(orc_kv), orc_kv_cmp);
  int64_t nu = 0;
  for (int64_t j = 0; j < n_sigs; ++j) {
    if (nu && sg[nu - 1].key == sg[j].key) sg[nu - 1].val += sg[j].val; else sg[nu++] = sg[j];
  }
  /* sorted_call_pos + bins_list (CS:92-93) */
  orc_kv* bins = (orc_kv*)malloc(sizeof(orc_kv) * (size_t)(n_calls + 1));
  for (int64_t i = 0; i < n_calls; ++i) { bins[i].key = call_pos[i]; bins[i].val = 0; }
  qsort(bins, (size_t)n_calls, sizeof(orc_kv), orc_kv_cmp);
  int64_t nb = 0;
  for (int64_t i = 0; i < n_calls; ++i) if (!nb || bins[nb - 1].key != bins[i].key) bins[nb++] = bins[i];
  int64_t start_i = 0;
  for (int64_t j = 0; j < nu; ++j) {                                             /* CS:96-111 */
    const int64_t pos = sg[j].key, w = sg[j].val;
    int64_t cnt = 0, real_i = 0;
    for (int64_t i = start_i; i < nb; ++i) {
      const int64_t lb = bins[i].key - flanking, rb = bins[i].key + flanking;
      if (lb > pos) break;
      if (pos <= rb) { if (++cnt == 1) real_i = i; bins[i].val += w; }
    }
    if (cnt) start_i = real_i;
  }
  for (int64_t i = 0; i < n_calls; ++i) {                                         /* dict lookup by call start (CS:340-344) */
    int64_t lo = 0, hi = nb;
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (bins[m].key >= call_pos[i]) hi = m; else lo = m + 1; }
    cov[i] = bins[lo].val;
  }
  free(sg); free(bins);
  return 0;
}

/* calc_del_call_cov (CS:138-280). has_entry[i] = 1 when the call has a dc_svlen entry (at least one supporting signature). */
int orc_cov_del(const int32_t* call_start, const int32_t* call_end, int64_t n_calls, const int32_t* sig_start, const int32_t* sig_end,
                const int32_t* sig_svlen, int64_t n_sigs, int32_t flanking, int64_t* cov, uint8_t* has_entry) {
  int sorted = 1;
  for (int64_t j = 0; j + 1 < n_sigs; ++j) if (sig_start[j] > sig_start[j + 1]) sorted = 0;
  /* (call, sig) pairs of the four scans, then set() per call */
  int64_t cap = 1024, np = 0;
  orc_kv* pr = (orc_kv*)malloc(sizeof(orc_kv) * (size_t)cap);
#define ORC_PUSH(c, s) do { if (np == cap) { cap *= 2; pr = (orc_kv*)realloc(pr, sizeof(orc_kv) * (size_t)cap); } pr[np].key = (c); pr[np].val = (s); ++np; } while (0)
  for (int64_t i = 0; i < n_calls; ++i) {
    const int64_t lb = (int64_t)call_start[i] - flanking, rb = (int64_t)call_end[i] + flanking;   /* CS:154 */
    for (int64_t j = 0; j < n_sigs; ++j) {
      const int64_t s = sig_start[j], e = sig_end[j];
      if (s >= lb && s <= rb) ORC_PUSH(i, j);          /* signature start inside the call region (CS:171-185) */
      if (e >= lb && e <= rb) ORC_PUSH(i, j);          /* signature end inside the call region   (CS:188-200) */
      if (lb >= s && lb <= e) ORC_PUSH(i, j);          /* region start inside the signature      (CS:204-220) */
      if (rb >= s && rb <= e) ORC_PUSH(i, j);          /* region end inside the signature        (CS:222-241) */
    }
  }
#undef ORC_PUSH
  qsort(pr, (size_t)np, sizeof(orc_kv), orc_kv_cmp);
  for (int64_t i = 0; i < n_calls; ++i) { cov[i] = 0; if (has_entry) has_entry[i] = 0; }
  for (int64_t k = 0; k < np; ++k) {
    if (k && pr[k].key == pr[k - 1].key && pr[k].val == pr[k - 1].val) continue;   /* list(set(...)) (CS:247) */
    cov[pr[k].key] += sig_svlen[pr[k].val];                                        /* CS:250 */
    if (has_entry) has_entry[pr[k].key] = 1;
  }
  free(pr);
  return sorted ? 0 : VSV_E_UNSORTED;
}

/* ================================================================================================
 * sig_extract.py analysis_split_read (SE:193-319), INS/DEL candidates (the TRA candidates of analysis_bnd are dropped by the
 * `grep -w INS|DEL` of SE:637-638 and are not produced). Segments per read as organize_split_signal builds them.
 * Rows in (read, emission) order; INS rows carry the sequence slice [q_start, q_end) and VSV_M_QREV when `query` is the
 * reversed read at that point (SE:215, 254, 312 reverse it cumulatively).
 * ================================================================================================ */
typedef struct { int64_t qs, qe, rs, re; int32_t chr, rev; } orc_cseg;
static orc_cseg orc_cflip(orc_cseg x, int64_t rl) { orc_cseg y = x; y.qs = rl - x.qe; y.qe = rl - x.qs; return y; }
static void orc_cput(sigvec* out, int del, int64_t pos, int64_t len, int64_t s0, int64_t s1, int32_t chr, int qrev, uint32_t rec) {
  vsv_sig s; memset(&s, 0, sizeof s);
  s.pos = (int32_t)pos; s.svlen = (int32_t)len; s.q_start = (int32_t)s0; s.q_end = (int32_t)s1; s.rec = rec; s.rec2 = 0xFFFFFFFFu;
  s.meta = (del ? VSV_M_DEL : 0) | VSV_M_SPLIT | (qrev ? VSV_M_QREV : 0); s.tid = chr;
  sv_push(out, &s);
}
static void orc_pair_rule(orc_cseg e1, orc_cseg e2, const orc_cseg* e3, int64_t sv, int64_t mx, int qrev, uint32_t rec, sigvec* out) {
  if (e1.re - e2.rs < sv) {                                                              /* SE:216 / 258 */
    int64_t ins = e2.qs + e1.re - e2.rs - e1.qe;
    if (ins >= sv)                                                                       /* SE:217 */
      if (e2.rs - e1.re <= 100 && (ins <= mx || mx == -1))                               /* SE:218 */
        if (!e3 || e3->rs >= e2.re) {                                                    /* SE:261 */
          int64_t half = (e2.rs - e1.re) / 2;                                            /* int((..)/2) */
          orc_cput(out, 0, (e2.rs + e1.re) / 2, ins, e1.qe + half, e2.qs - half, e2.chr, qrev, rec);
        }
    int64_t del = e2.rs - e2.qs + e1.qe - e1.re;
    if (del >= sv)                                                                       /* SE:225 */
      if (e2.qs - e1.qe <= 100 && (del <= mx || mx == -1))                               /* SE:226 */
        if (!e3 || e3->rs >= e2.re) orc_cput(out, 1, e1.re, del, 0, 0, e2.chr, 0, rec);  /* SE:270 */
  }
}
/* tra (optional, one byte per read): the read yields a translocation candidate — analysis_bnd (SE:100-191) is reached and its two
 * segments lie at most 100 read bases apart (SE:109). Those candidates never reach INS.sigs / DEL.sigs, but a task that holds one
 * is not "empty" (SE:533-535), so its reads go to reads.sigs. */
int orc_cutesv_split(const vsv_segments* sg, const int32_t* read_len, const uint32_t* read_rec, int32_t sv_size, int32_t max_size,
                     int32_t max_parts, vsv_sig** rows_out, int64_t* n_rows, uint8_t* tra) {
  sigvec out = {0};
  for (int64_t r = 0; r < sg->n_reads; ++r) {
    int64_t a = (int64_t)sg->seg_off[r], n = (int64_t)sg->seg_off[r + 1] - a;
    if (tra) tra[r] = 0;
    if (!(n <= max_parts || max_parts == -1)) continue;                                   /* SE:370 */
    orc_cseg* S = (orc_cseg*)malloc(sizeof(orc_cseg) * (size_t)(n + 1));
    for (int64_t k = 0; k < n; ++k) {                                                     /* sorted(key=x[0]), stable (SE:198) */
      orc_cseg s = { sg->q_start[a + k], sg->q_end[a + k], sg->ref_start[a + k], sg->ref_end[a + k], sg->ref_id[a + k], sg->is_reverse[a + k] };
      int64_t j = k;
      while (j > 0 && S[j - 1].qs > s.qs) { S[j] = S[j - 1]; --j; }
      S[j] = s;
    }
    int64_t rl = read_len[r], sv = sv_size, mx = max_size; uint32_t rec = read_rec[r];
    int trigger = 0, qrev = 0;
    if (n == 2) {
      orc_cseg e1 = S[0], e2 = S[1];
      if (e1.chr == e2.chr) {
        if (e1.rev == e2.rev) {
          if (e1.rev) { e1 = orc_cflip(S[1], rl); e2 = orc_cflip(S[0], rl); qrev ^= 1; }   /* SE:212-215 */
          orc_pair_rule(e1, e2, NULL, sv, mx, qrev, rec, &out);
        }
      } else { trigger = 1; if (tra && e2.qs - e1.qe <= 100) tra[r] = 1; }                /* SE:237-239: analysis_bnd */
    } else {
      for (int64_t k = 0; k + 2 < n; ++k) {                                               /* SE:243 */
        orc_cseg e1 = S[k], e2 = S[k + 1], e3 = S[k + 2];
        if (e1.chr == e2.chr) {
          if (e2.chr == e3.chr) {
            if (e1.rev == e3.rev && e1.rev == e2.rev) {                                   /* SE:251 */
              if (e1.rev) { e1 = orc_cflip(S[k + 2], rl); e2 = orc_cflip(S[k + 1], rl); e3 = orc_cflip(S[k], rl); qrev ^= 1; }
              orc_pair_rule(e1, e2, &e3, sv, mx, qrev, rec, &out);
              if (n - 3 == k) orc_pair_rule(e2, e3, NULL, sv, mx, qrev, rec, &out);        /* SE:277-296 */
            }
          }
        } else {                                                                          /* SE:298-302: analysis_bnd */
          trigger = 1;
          if (tra && e2.qs - e1.qe <= 100) tra[r] = 1;
          if (tra && n - 3 == k && e2.chr != e3.chr && e3.qs - e2.qe <= 100) tra[r] = 1;
        }
      }
      if (n >= 3 && trigger) {                                                            /* SE:305-319 */
        orc_cseg f = S[0], l = S[n - 1];
        if (f.chr == l.chr && f.rev == l.rev) {
          orc_cseg e1 = f, e2 = l;
          if (f.rev) { e1 = orc_cflip(l, rl); e2 = orc_cflip(f, rl); qrev ^= 1; }
          int64_t dis_ref = e2.rs - e1.re, dis_read = e2.qs - e1.qe;
          if (dis_ref < 100 && dis_read - dis_ref >= sv && (dis_read - dis_ref <= mx || mx == -1))
            orc_cput(&out, 0, e2.rs < e1.re ? e2.rs : e1.re, dis_read - dis_ref, e1.qe + dis_ref / 2, e2.qs - dis_ref / 2, e2.chr, qrev, rec);
        }
      }
    }
    free(S);
  }
  *rows_out = sv_dup(&out); *n_rows = out.n; free(out.v);
  return 0;
}

/* ================================================================================================
 * remove_redundancy.py (Large_INDEL): redundant-call matching. match_del_chr / match_ins_chr (RR:115-134, 162-181) with
 * their pair predicates (RR:92-114); edlib.align(...)["editDistance"] (RR:75-81; edlib 1.3.9, default NW mode, unit costs;
 * the package is absent from this image) restated as the Levenshtein DP. Calls of one chromosome, ascending by position
 * (sort_sig_per_chr, canonical stable order). Output: matching pairs i < j in (i, j) order; the reference's link list holds
 * each of them in both directions.
 * ================================================================================================ */
int64_t orc_levenshtein(const uint8_t* a, int64_t la, const uint8_t* b, int64_t lb) {
  int64_t* row = (int64_t*)malloc(sizeof(int64_t) * (size_t)(lb + 1));
  for (int64_t j = 0; j <= lb; ++j) row[j] = j;
  for (int64_t i = 1; i <= la; ++i) {
    int64_t diag = row[0];
    row[0] = i;
    for (int64_t j = 1; j <= lb; ++j) {
      int64_t sub = diag + (a[i - 1] != b[j - 1]);
      diag = row[j];
      int64_t v = row[j] + 1;
      if (row[j - 1] + 1 < v) v = row[j - 1] + 1;
      if (sub < v) v = sub;
      row[j] = v;
    }
  }
  int64_t d = row[lb];
  free(row);
  return d;
}

int orc_default_redundancy_params(vsv_redundancy_params* p) {
  p->dist_thresh = 500; p->dist_thresh_del = 3000; p->overlap_thresh = 0.0; p->size_sim_thresh = 0.5;     /* RR:9-14 */
  p->size_sim_thresh_del = 0.1; p->seq_sim_thresh = 0.5;
  return 0;
}

/* is_del: match_del_one_pair (RR:104-114) else match_ins_one_pair (RR:92-102); seq/seq_off = upper-cased ALT strings (RR:62) */
int orc_redundancy_pairs(int is_del, const int32_t* pos, const int32_t* svlen, const uint8_t* seq, const uint64_t* seq_off, int64_t n,
                         const vsv_redundancy_params* p, uint32_t** pairs_out, int64_t* n_pairs) {
  int64_t cap = 1024, np = 0;
  uint32_t* pr = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)cap);
  const int64_t dist = is_del ? p->dist_thresh_del : p->dist_thresh;
  const double size_thr = is_del ? p->size_sim_thresh_del : p->size_sim_thresh;
  for (int64_t i = 0; i < n; ++i) {
    for (int64_t j = i + 1; j < n; ++j) {
      if ((int64_t)pos[j] > (int64_t)pos[i] + dist) break;                      /* RR:123-124 */
      const int64_t l1 = svlen[i], l2 = svlen[j];
      const int64_t mn = l1 < l2 ? l1 : l2, mx = l1 < l2 ? l2 : l1;
      if (mx == 0) { free(pr); return VSV_E_ZERODIV; }                          /* get_size_sim divides (RR:88-90) */
      const double size_sim = (double)mn / (double)mx;
      if (!(size_sim >= size_thr)) continue;                                    /* dist_ref <= dist holds inside the window */
      int match;
      if (is_del) {
        const int64_t e1 = (int64_t)pos[i] + l1, e2 = (int64_t)pos[j] + l2;
        const int64_t ov = (e1 < e2 ? e1 : e2) - (pos[i] > pos[j] ? pos[i] : pos[j]);
        match = (double)ov / (double)mx >= p->overlap_thresh;                   /* RR:103-114 */
      } else {
        const int64_t la = (int64_t)(seq_off[i + 1] - seq_off[i]), lb = (int64_t)(seq_off[j + 1] - seq_off[j]);
        const int64_t ed = orc_levenshtein(seq + seq_off[i], la, seq + seq_off[j], lb);
        match = (double)(la + lb - ed) / (double)(la + lb) >= p->seq_sim_thresh;   /* RR:75-81, 98-100 */
      }
      if (match) {
        if (np == cap) { cap *= 2; pr = (uint32_t*)realloc(pr, sizeof(uint32_t) * 2 * (size_t)cap); }
        pr[2 * np] = (uint32_t)i; pr[2 * np + 1] = (uint32_t)j; ++np;
      }
    }
  }
  *pairs_out = pr; *n_pairs = np;
  return 0;
}
void orc_free_u32(uint32_t* p) { free(p); }

/* ================================================================================================
 * Genotype correction: correct_gt_del_real_data.py (DG) / correct_gt_ins_real_data.py (IG).
 * orc_gt_support: match_varlist_siglist (DG:92-137) / extract_sig_support (IG:105-156), the literal forward and backward
 * scans from the resume index, including their shared visit of that index. Chromosomes are ids; sig arrays in list order.
 * orc_span_count: count_reads_span_region (DG:140-147) / check_full_cover_reads (IG:178-186).
 * ================================================================================================ */
int orc_gt_support(const int32_t* var_chrom, const int32_t* var_pos, const int32_t* var_svlen, int64_t nv, const int32_t* sig_chrom,
                   const int32_t* sig_pos, const int32_t* sig_svlen, const int32_t* sig_cnt, int64_t ns, double max_shift_ratio,
                   double min_size_sim, int64_t* cnt_out, int64_t* match_out) {
  int64_t last = 0;
  for (int64_t v = 0; v < nv; ++v) {
    const double svlen = (double)var_svlen[v];
    const double prod = svlen * max_shift_ratio;
    const double max_shift = prod > 500.0 ? prod : 500.0;                          /* DG:107, IG:117 */
    const double min_pos = (double)var_pos[v] - max_shift, max_pos = (double)var_pos[v] + max_shift;
    const double min_size = svlen * min_size_sim, max_size = svlen / min_size_sim; /* DG:105-106 */
    int64_t support = 0, min_match = -1;
    match_out[v] = last;                                                            /* IG:125 */
    for (int64_t i = last; i < ns; ++i) {                                           /* DG:114-122 */
      if (sig_chrom[i] != var_chrom[v]) continue;
      const double p = (double)sig_pos[i];
      if (min_pos <= p && p <= max_pos) {
        if (min_match < 0 || i < min_match) min_match = i;
        const double l = (double)sig_svlen[i];
        if (min_size <= l && l <= max_size) support += sig_cnt[i];
      } else if (p > max_pos) break;
    }
    for (int64_t i = (ns > 0 ? last : -1); i >= 0; --i) {                           /* DG:124-132 */
      if (sig_chrom[i] != var_chrom[v]) continue;
      const double p = (double)sig_pos[i];
      if (min_pos <= p && p <= max_pos) {
        if (min_match < 0 || i < min_match) min_match = i;
        const double l = (double)sig_svlen[i];
        if (min_size <= l && l <= max_size) support += sig_cnt[i];
      } else if (p < min_pos) break;
    }
    if (min_match >= 0) last = min_match;                                           /* DG:134-135 */
    cnt_out[v] = support;
  }
  return 0;
}

int orc_span_count(const int32_t* read_tid, const int32_t* read_start, const int32_t* read_end, int64_t nr, const int32_t* q_tid,
                   const int32_t* q_a, const int32_t* q_b, int64_t nq, uint32_t* out) {
  for (int64_t i = 0; i < nq; ++i) {
    uint32_t c = 0;
    for (int64_t r = 0; r < nr; ++r)
      if (read_tid[r] == q_tid[i] && read_start[r] < q_b[i] && read_end[r] > q_a[i])   /* fetch(chrom, a, b): overlaps the region */
        if (read_start[r] < q_a[i] && read_end[r] > q_b[i]) ++c;                          /* DG:144 */
    out[i] = c;
  }
  return 0;
}
