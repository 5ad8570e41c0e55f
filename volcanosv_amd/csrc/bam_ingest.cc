// bam_ingest.cc — host-side BAM/BGZF -> record SoA adaptor (the step immediately left of the hot path).
//
// Replaces `pysam.AlignmentFile(bam).fetch(chr)` as used by the reference
// (Large_INDEL/extract_contig_signature_Hifi.py:387-391, extract_reads_signature.py:108-113): records of one
// reference id, in file (coordinate) order, reduced to the fields the path reads: pos, mapq, flag bits, qname,
// packed CIGAR (BAM packing is kept verbatim, long CIGARs are taken from the CG:B,I tag), plus the SA tag text for
// the svim segment analysis. Plain zlib inflate of BGZF members; no index is needed (whole-file scan).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/volcanosv.h"

struct vsv_bam {
  FILE* f = nullptr;
  std::string err;
  std::vector<std::string> ref_names;
  std::vector<int64_t> ref_lens;
  std::string header_text;
  // BGZF streaming state
  std::vector<uint8_t> buf;   // decompressed bytes not yet consumed
  size_t rd = 0;
  bool eof = false;
  long data_start_block = 0;  // file offset of the first BGZF block holding alignment data (unused: we rescan)
  // loaded records (library-owned, valid until the next load / close)
  std::vector<int32_t> pos, tid;
  std::vector<uint32_t> qid, cigar, l_seq, sam_flag;
  std::vector<uint64_t> cigar_off;
  std::vector<uint8_t> mapq, flag;
  std::vector<std::string> qnames;       // qid -> name
  std::string qname_blob;                // '\n'-joined, for the binding
  std::vector<std::string> sa;           // per record SA tag ("" if none)
  std::string sa_blob;
};

namespace {

bool fill(vsv_bam* b, size_t need) {
  // make at least `need` unread bytes available in b->buf
  while (b->buf.size() - b->rd < need && !b->eof) {
    uint8_t hdr[18];
    size_t got = fread(hdr, 1, 18, b->f);
    if (got == 0) { b->eof = true; break; }
    if (got != 18 || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) { b->err = "not a BGZF block"; return false; }
    uint16_t xlen = hdr[10] | (hdr[11] << 8);
    // the BC subfield is first in every htslib-written file; handle the general case anyway
    std::vector<uint8_t> extra(xlen);
    memcpy(extra.data(), hdr + 12, xlen < 6 ? xlen : 6);
    if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, b->f) != (size_t)(xlen - 6)) { b->err = "truncated BGZF extra field"; return false; }
    int bsize = -1;
    for (size_t o = 0; o + 4 <= extra.size();) {
      uint16_t slen = extra[o + 2] | (extra[o + 3] << 8);
      if (extra[o] == 'B' && extra[o + 1] == 'C' && slen == 2) bsize = extra[o + 4] | (extra[o + 5] << 8);
      o += 4 + slen;
    }
    if (bsize < 0) { b->err = "BGZF block without BC field"; return false; }
    size_t clen = (size_t)bsize + 1 - 12 - xlen - 8;
    std::vector<uint8_t> comp(clen + 8);
    if (fread(comp.data(), 1, clen + 8, b->f) != clen + 8) { b->err = "truncated BGZF block"; return false; }
    uint32_t isize = comp[clen + 4] | (comp[clen + 5] << 8) | (comp[clen + 6] << 16) | ((uint32_t)comp[clen + 7] << 24);
    if (b->rd > (1u << 20)) { b->buf.erase(b->buf.begin(), b->buf.begin() + b->rd); b->rd = 0; }
    size_t old = b->buf.size();
    b->buf.resize(old + isize);
    if (isize) {
      z_stream zs;
      memset(&zs, 0, sizeof zs);
      if (inflateInit2(&zs, -15) != Z_OK) { b->err = "inflateInit2 failed"; return false; }
      zs.next_in = comp.data(); zs.avail_in = (uInt)clen;
      zs.next_out = b->buf.data() + old; zs.avail_out = isize;
      int rc = inflate(&zs, Z_FINISH);
      inflateEnd(&zs);
      if (rc != Z_STREAM_END || zs.avail_out != 0) { b->err = "inflate failed"; return false; }
    }
  }
  return b->buf.size() - b->rd >= need;
}

bool rd_bytes(vsv_bam* b, void* dst, size_t n) {
  if (!fill(b, n)) { if (b->err.empty()) b->err = "unexpected end of BAM"; return false; }
  memcpy(dst, b->buf.data() + b->rd, n);
  b->rd += n;
  return true;
}

bool read_header(vsv_bam* b) {
  char magic[4];
  if (!rd_bytes(b, magic, 4) || memcmp(magic, "BAM\1", 4) != 0) { b->err = "bad BAM magic"; return false; }
  int32_t l_text;
  if (!rd_bytes(b, &l_text, 4)) return false;
  b->header_text.resize(l_text);
  if (l_text && !rd_bytes(b, &b->header_text[0], l_text)) return false;
  int32_t n_ref;
  if (!rd_bytes(b, &n_ref, 4)) return false;
  for (int i = 0; i < n_ref; ++i) {
    int32_t l_name, l_ref;
    if (!rd_bytes(b, &l_name, 4)) return false;
    std::string name(l_name, '\0');
    if (!rd_bytes(b, &name[0], l_name)) return false;
    if (!name.empty() && name.back() == '\0') name.pop_back();
    if (!rd_bytes(b, &l_ref, 4)) return false;
    b->ref_names.push_back(name);
    b->ref_lens.push_back(l_ref);
  }
  return true;
}

}  // namespace

extern "C" {

int vsv_bam_open(const char* path, vsv_bam** out) {
  if (!path || !out) return VSV_E_INVALID;
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) return VSV_E_INVALID;
  vsv_bam* b = new vsv_bam();
  b->f = f;
  if (!read_header(b)) { fclose(f); delete b; return VSV_E_INVALID; }
  *out = b;
  return 0;
}

void vsv_bam_close(vsv_bam* b) {
  if (!b) return;
  if (b->f) fclose(b->f);
  delete b;
}

const char* vsv_bam_error(vsv_bam* b) { return b ? b->err.c_str() : "null"; }
int vsv_bam_n_refs(vsv_bam* b) { return b ? (int)b->ref_names.size() : 0; }
const char* vsv_bam_ref_name(vsv_bam* b, int i) { return (b && i >= 0 && i < (int)b->ref_names.size()) ? b->ref_names[i].c_str() : ""; }
int64_t vsv_bam_ref_len(vsv_bam* b, int i) { return (b && i >= 0 && i < (int)b->ref_lens.size()) ? b->ref_lens[i] : -1; }

/* Loads every record with refID == tid (tid < 0: all mapped-or-placed records) in file order into library-owned
 * arrays and fills `out` with host pointers to them. qids are dense in first-appearance order; the hp flag bits come
 * from the substring test of H:392 ('hp1' in qname / 'hp2' in qname). */
int vsv_bam_load(vsv_bam* b, int tid, vsv_records* out) {
  if (!b || !out) return VSV_E_INVALID;
  // rewind and skip the header again (simple and index-free)
  fseek(b->f, 0, SEEK_SET);
  b->buf.clear(); b->rd = 0; b->eof = false;
  b->ref_names.clear(); b->ref_lens.clear();
  if (!read_header(b)) return VSV_E_INVALID;
  b->pos.clear(); b->tid.clear(); b->qid.clear(); b->cigar.clear(); b->cigar_off.assign(1, 0); b->mapq.clear(); b->flag.clear();
  b->qnames.clear(); b->sa.clear(); b->l_seq.clear(); b->sam_flag.clear();
  std::unordered_map<std::string, uint32_t> ids;
  std::vector<uint8_t> rec;
  for (;;) {
    int32_t block_size;
    if (!fill(b, 4)) { if (!b->err.empty()) return VSV_E_INVALID; break; }
    rd_bytes(b, &block_size, 4);
    if (block_size < 32) { b->err = "bad BAM record size"; return VSV_E_INVALID; }
    rec.resize(block_size);
    if (!rd_bytes(b, rec.data(), block_size)) return VSV_E_INVALID;
    int32_t refID, pos; memcpy(&refID, &rec[0], 4); memcpy(&pos, &rec[4], 4);
    if (refID < 0 || (tid >= 0 && refID != tid)) continue;
    const uint8_t l_read_name = rec[8], mq = rec[9];
    uint16_t n_cig, fl; memcpy(&n_cig, &rec[12], 2); memcpy(&fl, &rec[14], 2);
    int32_t l_seq; memcpy(&l_seq, &rec[16], 4);
    const char* name = (const char*)&rec[32];
    std::string qn(name, l_read_name ? l_read_name - 1 : 0);
    const uint8_t* cg = &rec[32 + l_read_name];
    size_t off = 32 + (size_t)l_read_name + 4u * n_cig + (size_t)((l_seq + 1) / 2) + (size_t)l_seq;
    // tags: SA:Z and CG:B,I
    std::string sa;
    const uint8_t* cg_long = nullptr; uint32_t n_long = 0;
    while (off + 3 <= (size_t)block_size) {
      const char t0 = rec[off], t1 = rec[off + 1], ty = rec[off + 2];
      off += 3;
      size_t len = 0;
      switch (ty) {
        case 'A': case 'c': case 'C': len = 1; break;
        case 's': case 'S': len = 2; break;
        case 'i': case 'I': case 'f': len = 4; break;
        case 'Z': case 'H': { size_t e = off; while (e < (size_t)block_size && rec[e]) ++e; if (t0 == 'S' && t1 == 'A' && ty == 'Z') sa.assign((const char*)&rec[off], e - off); len = e - off + 1; break; }
        case 'B': {
          const char sub = rec[off]; uint32_t cnt; memcpy(&cnt, &rec[off + 1], 4);
          size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
          if (t0 == 'C' && t1 == 'G' && sub == 'I') { cg_long = &rec[off + 5]; n_long = cnt; }
          len = 5 + es * cnt; break;
        }
        default: b->err = "unknown BAM tag type"; return VSV_E_INVALID;
      }
      off += len;
    }
    uint32_t q;
    auto it = ids.find(qn);
    if (it == ids.end()) { q = (uint32_t)b->qnames.size(); ids.emplace(qn, q); b->qnames.push_back(qn); } else q = it->second;
    uint8_t f8 = 0;
    if (fl & 0x10) f8 |= VSV_F_REVERSE;
    if (fl & 0x800) f8 |= VSV_F_SUPP;
    if (fl & 0x100) f8 |= VSV_F_SECONDARY;
    if (fl & 0x4) f8 |= VSV_F_UNMAPPED;
    if (qn.find("hp1") != std::string::npos) f8 |= VSV_F_HP1;
    if (qn.find("hp2") != std::string::npos) f8 |= VSV_F_HP2;
    b->pos.push_back(pos); b->tid.push_back(refID); b->qid.push_back(q); b->mapq.push_back(mq); b->flag.push_back(f8);
    b->l_seq.push_back((uint32_t)l_seq); b->sam_flag.push_back(fl);
    b->sa.push_back(sa);
    if (cg_long && n_cig == 2) {  // real CIGAR lives in the CG tag (htslib convention for > 65535 ops)
      size_t o = b->cigar.size(); b->cigar.resize(o + n_long); memcpy(&b->cigar[o], cg_long, 4u * n_long);
    } else {
      size_t o = b->cigar.size(); b->cigar.resize(o + n_cig); if (n_cig) memcpy(&b->cigar[o], cg, 4u * n_cig);
    }
    b->cigar_off.push_back(b->cigar.size());
  }
  b->qname_blob.clear();
  for (size_t i = 0; i < b->qnames.size(); ++i) { if (i) b->qname_blob.push_back('\n'); b->qname_blob += b->qnames[i]; }
  b->sa_blob.clear();
  for (size_t i = 0; i < b->sa.size(); ++i) { if (i) b->sa_blob.push_back('\n'); b->sa_blob += b->sa[i]; }
  memset(out, 0, sizeof *out);
  out->n_records = (int64_t)b->pos.size();
  out->n_ops = (int64_t)b->cigar.size();
  out->pos = b->pos.data(); out->tid = b->tid.data(); out->qid = b->qid.data(); out->cigar_off = b->cigar_off.data();
  out->mapq = b->mapq.data(); out->flag = b->flag.data(); out->cigar = b->cigar.data();
  out->on_device = 0;
  out->n_qids = (int32_t)b->qnames.size();
  out->n_tids = (int32_t)b->ref_names.size();
  return 0;
}

/* '\n'-joined query names in qid order / SA tags in record order of the last vsv_bam_load; *len receives the length */
const char* vsv_bam_qnames(vsv_bam* b, int64_t* len) { if (len) *len = b ? (int64_t)b->qname_blob.size() : 0; return b ? b->qname_blob.data() : ""; }
const char* vsv_bam_sa_tags(vsv_bam* b, int64_t* len) { if (len) *len = b ? (int64_t)b->sa_blob.size() : 0; return b ? b->sa_blob.data() : ""; }
const uint32_t* vsv_bam_l_seq(vsv_bam* b) { return b ? b->l_seq.data() : nullptr; }
const uint32_t* vsv_bam_sam_flags(vsv_bam* b) { return b ? b->sam_flag.data() : nullptr; }

}  // extern "C"
