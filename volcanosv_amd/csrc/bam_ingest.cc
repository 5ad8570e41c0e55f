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
#include <sys/mman.h>
#include <unistd.h>
#include <zlib.h>

#include "vsv_env.h"

#include <atomic>
#include <exception>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <chrono>

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
  uint64_t inflated_total = 0;   // bytes appended to buf since the last rewind (for the header length)
  const uint32_t* dev_l_seq = nullptr; const uint32_t* dev_sam_flag = nullptr;   // device arrays of the last vsv_bam_load_device
  int n_threads = 0;          // inflate workers for vsv_bam_load (0 = hardware_concurrency, capped at 16)
  vsv_handle* gpu = nullptr;  // vsv_bam_set_inflate_device: windows are inflated by vsv_bgzf_inflate instead of zlib
  // loaded records (library-owned, valid until the next load / close)
  std::vector<int32_t> pos, tid;
  std::vector<uint32_t> qid, cigar, l_seq, sam_flag;
  std::vector<uint64_t> cigar_off;
  std::vector<uint8_t> mapq, flag;
  std::vector<std::string> qnames;       // qid -> name
  std::string qname_blob;                // '\n'-joined, for the binding
  bool keep_seq = false;                 // vsv_bam_set_keep_seq: store the packed 4-bit SEQ of every loaded record
  std::vector<uint8_t> seq;              // concatenated, record i occupies (l_seq[i]+1)/2 bytes
  std::vector<std::string> sa;           // per record SA tag ("" if none)
  std::string sa_blob;
  const char* dev_names = nullptr;     // name table of the last vsv_bam_load_device: owned by the GPU handle, not copied
  int64_t dev_names_len = 0;
};

namespace {

// BGZF members are independent deflate streams: a window of up to WINDOW_BLOCKS members is read, inflated by a small
// thread pool (each worker takes the next member off an atomic counter) and appended to the byte queue the record
// parser consumes. Peak memory = one window (<= 64 KiB x WINDOW_BLOCKS decompressed).
constexpr int WINDOW_BLOCKS = 2048;
constexpr int WINDOW_BLOCKS_GPU = 16384;

struct Member { std::vector<uint8_t> comp; uint32_t isize; size_t out_off; };

bool read_member(vsv_bam* b, Member& m, bool& got) {
  got = false;
  uint8_t hdr[18];
  size_t n = fread(hdr, 1, 18, b->f);
  if (n == 0) { b->eof = true; return true; }
  if (n != 18 || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) { b->err = "not a BGZF block"; return false; }
  uint16_t xlen = hdr[10] | (hdr[11] << 8);
  std::vector<uint8_t> extra(xlen);
  memcpy(extra.data(), hdr + 12, xlen < 6 ? xlen : 6);
  if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, b->f) != (size_t)(xlen - 6)) { b->err = "truncated BGZF extra field"; return false; }
  int bsize = -1;
  for (size_t o = 0; o + 4 <= extra.size();) {
    uint16_t slen = extra[o + 2] | (extra[o + 3] << 8);
    if (o + 4 + slen > extra.size()) break;                      // subfield runs past the extra field
    if (extra[o] == 'B' && extra[o + 1] == 'C' && slen == 2) bsize = extra[o + 4] | (extra[o + 5] << 8);
    o += 4 + slen;
  }
  if (bsize < 0) { b->err = "BGZF block without BC field"; return false; }
  if ((size_t)bsize + 1 < 12u + xlen + 8u) { b->err = "BGZF block smaller than its own header"; return false; }
  const size_t clen = (size_t)bsize + 1 - 12 - xlen - 8;
  m.comp.resize(clen + 8);
  if (fread(m.comp.data(), 1, clen + 8, b->f) != clen + 8) { b->err = "truncated BGZF block"; return false; }
  m.comp.resize(clen + 8);
  m.isize = m.comp[clen + 4] | (m.comp[clen + 5] << 8) | (m.comp[clen + 6] << 16) | ((uint32_t)m.comp[clen + 7] << 24);
  if (m.isize > 65536u) { b->err = "BGZF block announces more than 64 KiB"; return false; }
  got = true;
  return true;
}

bool inflate_member(const Member& m, uint8_t* dst) {
  if (!m.isize) return true;
  z_stream zs;
  memset(&zs, 0, sizeof zs);
  if (inflateInit2(&zs, -15) != Z_OK) return false;
  zs.next_in = const_cast<uint8_t*>(m.comp.data()); zs.avail_in = (uInt)(m.comp.size() - 8);
  zs.next_out = dst; zs.avail_out = m.isize;
  const int rc = inflate(&zs, Z_FINISH);
  inflateEnd(&zs);
  if (!(rc == Z_STREAM_END && zs.avail_out == 0)) return false;
  // the member's CRC-32 (htslib bgzf.c checks it too): a flipped bit that still inflates must not pass as data
  const uint8_t* t = m.comp.data() + (m.comp.size() - 8);
  const uint32_t want = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
  return (uint32_t)crc32(0L, dst, m.isize) == want;
}

bool fill(vsv_bam* b, size_t need) {
  // make at least `need` unread bytes available in b->buf
  while (b->buf.size() - b->rd < need && !b->eof) {
    std::vector<Member> win;
    size_t total = 0;
    const int window_blocks = b->gpu ? WINDOW_BLOCKS_GPU : WINDOW_BLOCKS;   // the GPU wants tens of thousands of members in flight
    while ((int)win.size() < window_blocks && !b->eof) {
      Member m; bool got;
      if (!read_member(b, m, got)) return false;
      if (!got) break;
      m.out_off = total; total += m.isize;
      win.push_back(std::move(m));
      if (b->n_threads == 1 && total >= need) break;   // single-threaded callers (header parse) stay incremental
    }
    if (win.empty()) break;
    if (b->rd > 0) { b->buf.erase(b->buf.begin(), b->buf.begin() + b->rd); b->rd = 0; }
    const size_t old = b->buf.size();
    b->buf.resize(old + total);
    b->inflated_total += total;
    uint8_t* base = b->buf.data() + old;
    if (b->gpu && win.size() >= 64) {                    // whole window on the GPU, one lane per member
      std::vector<uint64_t> coff(win.size() + 1, 0);
      std::vector<uint32_t> isz(win.size());
      for (size_t i = 0; i < win.size(); ++i) { coff[i + 1] = coff[i] + (win[i].comp.size() - 8); isz[i] = win[i].isize; }
      std::vector<uint8_t> comp((size_t)coff.back() + 1);
      for (size_t i = 0; i < win.size(); ++i) memcpy(comp.data() + coff[i], win[i].comp.data(), win[i].comp.size() - 8);
      std::vector<uint32_t> want(win.size());
      for (size_t i = 0; i < win.size(); ++i) { const uint8_t* t = win[i].comp.data() + (win[i].comp.size() - 8); want[i] = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24); }
      vsv_bgzf_set_expected_crc(b->gpu, want.data(), (int64_t)want.size());
      const int gs = vsv_bgzf_inflate(b->gpu, comp.data(), coff.data(), isz.data(), (int64_t)win.size(), base);
      vsv_bgzf_set_expected_crc(b->gpu, nullptr, 0);
      if (gs != 0) { b->err = std::string("GPU inflate failed: ") + vsv_last_error(b->gpu); return false; }
      continue;
    }
    int nt = b->n_threads > 0 ? b->n_threads : (int)std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt < 1) nt = 1;
    if ((size_t)nt > win.size()) nt = (int)win.size();
    std::atomic<size_t> next{0};
    std::atomic<bool> ok{true};
    auto work = [&]() {
      for (;;) {
        const size_t i = next.fetch_add(1);
        if (i >= win.size()) break;
        if (!inflate_member(win[i], base + win[i].out_off)) ok = false;
      }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    if (!ok) { b->err = "inflate failed"; return false; }
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
  if (l_text < 0 || l_text > (1 << 30)) { b->err = "bad BAM header length"; return false; }
  b->header_text.resize(l_text);
  if (l_text && !rd_bytes(b, &b->header_text[0], l_text)) return false;
  int32_t n_ref;
  if (!rd_bytes(b, &n_ref, 4)) return false;
  if (n_ref < 0 || n_ref > (1 << 24)) { b->err = "bad BAM reference count"; return false; }
  for (int i = 0; i < n_ref; ++i) {
    int32_t l_name, l_ref;
    if (!rd_bytes(b, &l_name, 4)) return false;
    if (l_name < 1 || l_name > (1 << 16)) { b->err = "bad BAM reference name length"; return false; }
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

extern "C" void* vsv_internal_pinned(vsv_handle* h, uint64_t bytes);   // capi.hip

extern "C" {

int vsv_bam_open(const char* path, vsv_bam** out) {
  if (!path || !out) return VSV_E_INVALID;
  *out = nullptr;
  FILE* f = fopen(path, "rb");
  if (!f) return VSV_E_INVALID;
  vsv_bam* b = new vsv_bam();
  b->f = f;
  b->n_threads = 1;                      // header only: inflate the first members, not a whole window
  bool hdr_ok = false;
  try { hdr_ok = read_header(b); } catch (const std::exception&) { hdr_ok = false; }
  if (!hdr_ok) { fclose(f); delete b; return VSV_E_INVALID; }
  b->n_threads = 0;
  *out = b;
  return 0;
}

void vsv_bam_close(vsv_bam* b) {
  if (!b) return;
  if (b->f) fclose(b->f);
  delete b;
}

const char* vsv_bam_error(vsv_bam* b) { return b ? b->err.c_str() : "null"; }
/* number of inflate worker threads used by vsv_bam_load (0 = all hardware threads, at most 16) */
void vsv_bam_set_threads(vsv_bam* b, int n) { if (b) b->n_threads = n < 0 ? 0 : n; }
void vsv_bam_set_inflate_device(vsv_bam* b, vsv_handle* h) { if (b) b->gpu = h; }
int vsv_bam_n_refs(vsv_bam* b) { return b ? (int)b->ref_names.size() : 0; }
const char* vsv_bam_ref_name(vsv_bam* b, int i) { return (b && i >= 0 && i < (int)b->ref_names.size()) ? b->ref_names[i].c_str() : ""; }
int64_t vsv_bam_ref_len(vsv_bam* b, int i) { return (b && i >= 0 && i < (int)b->ref_lens.size()) ? b->ref_lens[i] : -1; }

/* Loads every record with refID == tid (tid < 0: all mapped-or-placed records) in file order into library-owned
 * arrays and fills `out` with host pointers to them. qids are dense in first-appearance order; the hp flag bits come
 * from the substring test of H:392 ('hp1' in qname / 'hp2' in qname).
 * Per window of inflated bytes: (1) a sequential hop over the 4-byte block_size fields finds the record starts and sizes
 * the output arrays, (2) the worker threads fill the fixed fields, copy CIGAR / SEQ, walk the tags (SA:Z, CG:B,I) and hash
 * the names, (3) a sequential pass interns the names (first-appearance ids need file order) and appends the SA text. */
namespace {
struct RecRef { size_t off; uint32_t size; uint32_t n_cig_out; uint64_t cig_off; uint64_t seq_off; };
struct RecAux { uint64_t name_hash; const char* sa; uint32_t sa_len; const uint8_t* cg_long; };

// walks the tag area of one record; returns false on an unknown tag type or a tag that does not end inside the record
bool walk_tags(const uint8_t* rec, size_t off, size_t block_size, const char** sa_p, uint32_t* sa_len, const uint8_t** cg_long, uint32_t* n_long) {
  while (off + 3 <= block_size) {
    const char t0 = rec[off], t1 = rec[off + 1], ty = rec[off + 2];
    off += 3;
    size_t len = 0;
    switch (ty) {
      case 'A': case 'c': case 'C': len = 1; break;
      case 's': case 'S': len = 2; break;
      case 'i': case 'I': case 'f': len = 4; break;
      case 'Z': case 'H': { size_t e = off; while (e < block_size && rec[e]) ++e; if (e >= block_size) return false; if (t0 == 'S' && t1 == 'A' && ty == 'Z') { *sa_p = (const char*)&rec[off]; *sa_len = (uint32_t)(e - off); } len = e - off + 1; break; }
      case 'B': {
        if (off + 5 > block_size) return false;
        const char sub = rec[off]; uint32_t cnt; memcpy(&cnt, &rec[off + 1], 4);
        const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : 4;
        len = 5 + es * (size_t)cnt;
        if (len > block_size - off) return false;          // the array must end inside the record
        if (t0 == 'C' && t1 == 'G' && sub == 'I') { *cg_long = &rec[off + 5]; *n_long = cnt; }
        break;
      }
      default: return false;
    }
    if (len > block_size - off) return false;              // a value that runs past the record
    off += len;
  }
  return off == block_size;                                  // 1-2 stray bytes after the last tag are malformed too
}
}  // namespace

static int bam_load_impl(vsv_bam* b, int tid, vsv_records* out);
// no exception crosses the C boundary: a corrupt file that asks for an absurd allocation ends as an error like any other
int vsv_bam_load(vsv_bam* b, int tid, vsv_records* out) {
  if (!b || !out) return VSV_E_INVALID;
  try { return bam_load_impl(b, tid, out); }
  catch (const std::exception& e) { b->err = std::string("BAM load failed: ") + e.what(); return VSV_E_INVALID; }
}
static int bam_load_impl(vsv_bam* b, int tid, vsv_records* out) {
  // rewind and skip the header again (simple and index-free)
  fseek(b->f, 0, SEEK_SET);
  b->buf.clear(); b->rd = 0; b->eof = false; b->err.clear();
  b->ref_names.clear(); b->ref_lens.clear();
  const int user_threads = b->n_threads;
  b->n_threads = 1;                      // the header is parsed incrementally
  const bool hdr_ok = read_header(b);
  b->n_threads = user_threads;
  if (!hdr_ok) return VSV_E_INVALID;
  b->pos.clear(); b->tid.clear(); b->qid.clear(); b->cigar.clear(); b->cigar_off.assign(1, 0); b->mapq.clear(); b->flag.clear();
  b->qnames.clear(); b->sa.clear(); b->l_seq.clear(); b->sam_flag.clear();
  b->qname_blob.clear(); b->sa_blob.clear(); b->seq.clear(); b->dev_names = nullptr; b->dev_names_len = 0;
  // open-addressing name table: slot -> (hash, offset into qname_blob, length, id)
  struct Slot { uint64_t h; uint32_t off, len, id; };
  std::vector<Slot> table(1u << 16, Slot{0, 0, 0, 0xFFFFFFFFu});
  size_t n_names = 0;
  auto grow = [&]() {
    std::vector<Slot> t2(table.size() * 2, Slot{0, 0, 0, 0xFFFFFFFFu});
    for (const Slot& s : table) if (s.id != 0xFFFFFFFFu) { size_t i = s.h & (t2.size() - 1); while (t2[i].id != 0xFFFFFFFFu) i = (i + 1) & (t2.size() - 1); t2[i] = s; }
    table.swap(t2);
  };
  int nt = b->n_threads > 0 ? b->n_threads : (int)std::thread::hardware_concurrency();
  if (nt > 16) nt = 16;
  if (nt < 1) nt = 1;
  std::vector<RecRef> refs;
  std::vector<RecAux> aux;
  bool first_rec = true;
  static const bool timing = vsv_dbg_env("VSV_BAM_TIMING") != nullptr;
  double t_fill = 0, t_hop = 0, t_par = 0, t_seq = 0;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  for (;;) {
    double tq = now();
    // ---- make the next window available (at least one whole record, or EOF) -----------------------------------
    if (b->buf.size() - b->rd < 4) { if (!fill(b, 4)) { if (!b->err.empty()) return VSV_E_INVALID; break; } }
    {
      int32_t bs; memcpy(&bs, b->buf.data() + b->rd, 4);
      if (bs < 32) { b->err = "bad BAM record size"; return VSV_E_INVALID; }
      if (!fill(b, 4 + (size_t)bs)) { if (b->err.empty()) b->err = "unexpected end of BAM"; return VSV_E_INVALID; }
    }
    t_fill += now() - tq; tq = now();
    // ---- (1) hop scan ---------------------------------------------------------------------------------------
    refs.clear();
    const uint8_t* base = b->buf.data();
    const size_t end = b->buf.size();
    size_t rd = b->rd;
    uint64_t cig_total = b->cigar.size(), seq_total = b->seq.size();
    while (rd + 4 <= end) {
      int32_t bs; memcpy(&bs, base + rd, 4);
      if (bs < 32) { b->err = "bad BAM record size"; return VSV_E_INVALID; }
      if (rd + 4 + (size_t)bs > end) break;                       // partial record: next window
      const uint8_t* rec = base + rd + 4;
      int32_t refID; memcpy(&refID, &rec[0], 4);
      {   // the variable-length fields must fit the record (every later access relies on it)
        uint16_t nc; memcpy(&nc, &rec[12], 2);
        int32_t ls; memcpy(&ls, &rec[16], 4);
        if (ls < 0 || 32ull + rec[8] + 4ull * nc + (uint64_t)((ls + 1ll) / 2) + (uint64_t)ls > (uint64_t)bs) { b->err = "malformed BAM record"; return VSV_E_INVALID; }
      }
      if (!(refID < 0 || (tid >= 0 && refID != tid))) {
        uint16_t n_cig; memcpy(&n_cig, &rec[12], 2);
        int32_t l_seq; memcpy(&l_seq, &rec[16], 4);
        uint32_t n_out = n_cig;
        if (n_cig == 2) {                                         // possible CG:B,I long CIGAR (htslib: kSmN placeholder)
          const uint8_t l_read_name = rec[8];
          const char* sp = nullptr; uint32_t sl = 0; const uint8_t* cgl = nullptr; uint32_t nl = 0;
          const size_t toff = 32 + (size_t)l_read_name + 8u + (size_t)((l_seq + 1) / 2) + (size_t)l_seq;
          if (!walk_tags(rec, toff, (size_t)bs, &sp, &sl, &cgl, &nl)) { b->err = "unknown or malformed BAM tag"; return VSV_E_INVALID; }
          if (cgl) n_out = nl;
        }
        refs.push_back(RecRef{rd + 4, (uint32_t)bs, n_out, cig_total, seq_total});
        cig_total += n_out;
        if (b->keep_seq && l_seq > 0) seq_total += (uint64_t)((l_seq + 1) / 2);
      }
      rd += 4 + (size_t)bs;
    }
    b->rd = rd;
    const size_t n0 = b->pos.size(), nw = refs.size();
    t_hop += now() - tq; tq = now();
    if (nw) {
      b->pos.resize(n0 + nw); b->tid.resize(n0 + nw); b->qid.resize(n0 + nw); b->mapq.resize(n0 + nw); b->flag.resize(n0 + nw);
      b->l_seq.resize(n0 + nw); b->sam_flag.resize(n0 + nw); b->cigar_off.resize(n0 + nw + 1);
      b->cigar.resize(cig_total); b->seq.resize(seq_total);
      aux.resize(nw);
      // ---- (2) parallel field parse ---------------------------------------------------------------------------
      std::atomic<size_t> next{0};
      std::atomic<bool> ok{true};
      auto work = [&]() {
        for (;;) {
          const size_t c0 = next.fetch_add(4096);
          if (c0 >= nw) break;
          const size_t c1 = c0 + 4096 < nw ? c0 + 4096 : nw;
          for (size_t k = c0; k < c1; ++k) {
            const RecRef& r = refs[k];
            const uint8_t* rec = base + r.off;
            const size_t i = n0 + k;
            int32_t refID, pos; memcpy(&refID, &rec[0], 4); memcpy(&pos, &rec[4], 4);
            const uint8_t l_read_name = rec[8], mq = rec[9];
            uint16_t n_cig, fl; memcpy(&n_cig, &rec[12], 2); memcpy(&fl, &rec[14], 2);
            int32_t l_seq; memcpy(&l_seq, &rec[16], 4);
            const char* name = (const char*)&rec[32];
            const uint32_t nlen = l_read_name ? l_read_name - 1u : 0u;
            const uint8_t* cg = &rec[32 + l_read_name];
            const uint8_t* sq = cg + 4u * n_cig;
            RecAux a{1469598103934665603ull, nullptr, 0, nullptr};
            uint32_t n_long = 0;
            if (!walk_tags(rec, 32 + (size_t)l_read_name + 4u * n_cig + (size_t)((l_seq + 1) / 2) + (size_t)l_seq, r.size, &a.sa, &a.sa_len,
                           &a.cg_long, &n_long)) { ok = false; continue; }
            for (uint32_t c = 0; c < nlen; ++c) { a.name_hash ^= (uint8_t)name[c]; a.name_hash *= 1099511628211ull; }   // FNV-1a
            uint8_t f8 = 0;
            if (fl & 0x10) f8 |= VSV_F_REVERSE;
            if (fl & 0x800) f8 |= VSV_F_SUPP;
            if (fl & 0x100) f8 |= VSV_F_SECONDARY;
            if (fl & 0x4) f8 |= VSV_F_UNMAPPED;
            for (uint32_t c = 0; c + 3 <= nlen; ++c)   // 'hp1' in qname / 'hp2' in qname (H:392)
              if (name[c] == 'h' && name[c + 1] == 'p') { if (name[c + 2] == '1') f8 |= VSV_F_HP1; else if (name[c + 2] == '2') f8 |= VSV_F_HP2; }
            if (a.cg_long && n_cig == 2) memcpy(&b->cigar[r.cig_off], a.cg_long, 4u * (size_t)r.n_cig_out);   // real CIGAR lives in CG:B,I
            else if (n_cig) memcpy(&b->cigar[r.cig_off], cg, 4u * (size_t)n_cig);
            if (l_seq > 0) {   // a stored SEQ must be as long as the CIGAR's query (M,I,S,=,X): what the extractors assert (H:397-398)
              uint64_t ql = 0;
              for (uint32_t c = 0; c < r.n_cig_out; ++c) {
                const uint32_t w = b->cigar[r.cig_off + c], op = w & 15u;
                if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) ql += w >> 4;
              }
              if (ql != (uint64_t)l_seq) f8 |= VSV_F_SEQ_MISMATCH;
            }
            b->pos[i] = pos; b->tid[i] = refID; b->mapq[i] = mq; b->flag[i] = f8; b->l_seq[i] = (uint32_t)l_seq; b->sam_flag[i] = fl;
            b->cigar_off[i + 1] = r.cig_off + r.n_cig_out;
            if (b->keep_seq && l_seq > 0) memcpy(&b->seq[r.seq_off], sq, (size_t)((l_seq + 1) / 2));
            aux[k] = a;
          }
        }
      };
      std::vector<std::thread> pool;
      const int use = nw < 8192 ? 1 : nt;
      for (int t = 1; t < use; ++t) pool.emplace_back(work);
      work();
      for (auto& th : pool) th.join();
      if (!ok) { b->err = "unknown or malformed BAM tag"; return VSV_E_INVALID; }
      t_par += now() - tq; tq = now();
      // ---- (3) sequential: name interning in file order, SA text -----------------------------------------------
      for (size_t k = 0; k < nw; ++k) {
        if (k + 16 < nw) __builtin_prefetch(&table[aux[k + 16].name_hash & (table.size() - 1)]);   // the probe is a cache miss per name
        const uint8_t* rec = base + refs[k].off;
        const char* name = (const char*)&rec[32];
        const uint32_t nlen = rec[8] ? rec[8] - 1u : 0u;
        const uint64_t h = aux[k].name_hash;
        size_t si = h & (table.size() - 1);
        uint32_t q = 0xFFFFFFFFu;
        for (;;) {
          Slot& s = table[si];
          if (s.id == 0xFFFFFFFFu) {
            q = (uint32_t)n_names++;
            if (q) b->qname_blob.push_back('\n');
            s = Slot{h, (uint32_t)b->qname_blob.size(), nlen, q};
            b->qname_blob.append(name, nlen);
            if (n_names * 2 > table.size()) grow();
            break;
          }
          if (s.h == h && s.len == nlen && memcmp(b->qname_blob.data() + s.off, name, nlen) == 0) { q = s.id; break; }
          si = (si + 1) & (table.size() - 1);
        }
        b->qid[n0 + k] = q;
        if (!first_rec) b->sa_blob.push_back('\n');
        first_rec = false;
        if (aux[k].sa_len) b->sa_blob.append(aux[k].sa, aux[k].sa_len);
      }
      t_seq += now() - tq;
    }
  }
  if (timing) fprintf(stderr, "vsv_bam_load: fill(read+inflate) %.3f s, hop scan %.3f s, parallel parse %.3f s, intern %.3f s\n", t_fill, t_hop, t_par, t_seq);
  memset(out, 0, sizeof *out);
  out->n_records = (int64_t)b->pos.size();
  out->n_ops = (int64_t)b->cigar.size();
  out->pos = b->pos.data(); out->tid = b->tid.data(); out->qid = b->qid.data(); out->cigar_off = b->cigar_off.data();
  out->mapq = b->mapq.data(); out->flag = b->flag.data(); out->cigar = b->cigar.data();
  out->on_device = 0;
  out->n_qids = (int32_t)n_names;
  out->n_tids = (int32_t)b->ref_names.size();
  return 0;
}

/* Same records, but inflated and parsed on the GPU (vsv_bam_parse_device): `out` holds device pointers owned by `h`. The header
 * is read on the host (it is a few KB) to learn the reference table and where the first record starts. */
static int bam_load_device_impl(vsv_bam* b, vsv_handle* h, int tid, vsv_records* out);
int vsv_bam_load_device(vsv_bam* b, vsv_handle* h, int tid, vsv_records* out) {
  if (!b || !h || !out) return VSV_E_INVALID;
  try { return bam_load_device_impl(b, h, tid, out); }
  catch (const std::exception& e) { b->err = std::string("BAM load failed: ") + e.what(); return VSV_E_INVALID; }
}
static int bam_load_device_impl(vsv_bam* b, vsv_handle* h, int tid, vsv_records* out) {
  fseek(b->f, 0, SEEK_SET);
  b->buf.clear(); b->rd = 0; b->eof = false; b->inflated_total = 0; b->err.clear(); b->dev_names = nullptr; b->dev_names_len = 0;
  b->ref_names.clear(); b->ref_lens.clear();
  const int user_threads = b->n_threads;
  vsv_handle* user_gpu = b->gpu;
  b->n_threads = 1; b->gpu = nullptr;
  const bool hdr_ok = read_header(b);
  b->n_threads = user_threads; b->gpu = user_gpu;
  if (!hdr_ok) return VSV_E_INVALID;
  const bool timing = vsv_dbg_env("VSV_BAM_TIMING") != nullptr;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  const uint64_t first_record = b->inflated_total - (b->buf.size() - b->rd);
  // the file is mapped, not copied: the member table reads one header per member and the device reader uploads each window's
  // bytes straight from the page cache; the members' deflate payloads are addressed in place (a deflate stream ends itself,
  // so the trailer and the next header that follow a payload inside [comp_off[i], comp_off[i+1]) are never consumed)
  fseek(b->f, 0, SEEK_END);
  const long fsize = ftell(b->f);
  fseek(b->f, 0, SEEK_SET);
  if (fsize <= 0) { b->err = "empty file"; return VSV_E_INVALID; }
  struct Mapping {
    void* p = MAP_FAILED; size_t n = 0;
    ~Mapping() { if (p != MAP_FAILED) munmap(p, n); }
  } map;
  map.n = (size_t)fsize;
  map.p = mmap(nullptr, map.n, PROT_READ, MAP_PRIVATE, fileno(b->f), 0);
  if (map.p == MAP_FAILED) { b->err = "cannot map the BAM file"; return VSV_E_INVALID; }
  (void)madvise(map.p, map.n, MADV_SEQUENTIAL);
  const uint8_t* comp = (const uint8_t*)map.p;
  // Files of up to 1 GiB are first read into a page-locked buffer of the handle by a few threads (the page cache delivers ~10 GB/s
  // per thread): the member table then reads hot memory instead of faulting one mapped page per member, and the device reader's
  // uploads are asynchronous DMAs at PCIe speed that run under the decode of the previous slice. Larger files stream from the
  // mapping, window by window.
  static const char* pin_env = vsv_dbg_env("VSV_BAM_PINNED");      // timing experiments: "0" keeps the mapped file
  std::vector<std::thread> pool;
  std::vector<std::atomic<int>> ready;                        // per reader thread: 0 reading, 1 done, -1 failed
  size_t per = 0;
  if ((size_t)fsize <= (1ull << 30) && !(pin_env && pin_env[0] == '0')) {
    uint8_t* pin = (uint8_t*)vsv_internal_pinned(h, (uint64_t)fsize + 64);
    if (pin) {
      const int fd = fileno(b->f);
      unsigned nt = std::thread::hardware_concurrency();
      nt = nt == 0 ? 4 : nt > 16 ? 16 : nt;
      if ((size_t)fsize < (8u << 20)) nt = 1;
      ready = std::vector<std::atomic<int>>(nt);
      for (auto& r : ready) r.store(0);
      per = (((size_t)fsize + nt - 1) / nt + 4095) & ~(size_t)4095;     // nt * per >= fsize: every byte has a reader
      for (unsigned t = 0; t < nt; ++t)
        pool.emplace_back([&ready, pin, fd, fsize, per, t]() {
          size_t o = (size_t)t * per, e = o + per < (size_t)fsize ? o + per : (size_t)fsize;
          while (o < e) {
            const ssize_t g = pread(fd, pin + o, e - o, (off_t)o);
            if (g <= 0) { ready[t].store(-1, std::memory_order_release); return; }
            o += (size_t)g;
          }
          ready[t].store(1, std::memory_order_release);
        });
      memset(pin + fsize, 0, 64);
      comp = pin;
    }
  }
  struct Joiner { std::vector<std::thread>& p; ~Joiner() { for (auto& th : p) if (th.joinable()) th.join(); } } joiner{pool};
  // bytes [0, end) of the staged file are in place? (the member table below walks the file while the readers are still at work)
  bool stage_failed = false;
  auto wait_for = [&](size_t end) {
    if (pool.empty()) return;
    const size_t last = (end == 0 ? 0 : end - 1) / per;
    for (size_t t = 0; t <= last && t < ready.size(); ++t) {
      int v;
      while ((v = ready[t].load(std::memory_order_acquire)) == 0) std::this_thread::yield();
      if (v < 0) stage_failed = true;
    }
  };
  const double t1 = now();
  std::vector<uint64_t> coff;
  std::vector<uint32_t> isz, crcs;
  for (size_t o = 0; o < (size_t)fsize;) {
    wait_for(o + 18 < (size_t)fsize ? o + 18 : (size_t)fsize);
    if (stage_failed) { b->err = "cannot read the BAM file"; return VSV_E_INVALID; }
    const uint8_t* hdr = comp + o;
    if (o + 18 > (size_t)fsize || hdr[0] != 31 || hdr[1] != 139 || hdr[2] != 8 || !(hdr[3] & 4)) { b->err = "not a BGZF block"; return VSV_E_INVALID; }
    const uint16_t xlen = hdr[10] | (hdr[11] << 8);
    if (o + 12 + (size_t)xlen + 8 > (size_t)fsize) { b->err = "truncated BGZF block"; return VSV_E_INVALID; }
    wait_for(o + 12 + (size_t)xlen);
    int bsize = -1;
    for (size_t e = 0; e + 4 <= xlen;) {
      const uint8_t* x = hdr + 12 + e;
      const uint16_t slen = x[2] | (x[3] << 8);
      if (e + 4 + slen > xlen) break;
      if (x[0] == 'B' && x[1] == 'C' && slen == 2) bsize = x[4] | (x[5] << 8);
      e += 4 + slen;
    }
    if (bsize < 0 || o + (size_t)bsize + 1 > (size_t)fsize || (size_t)bsize + 1 < 12u + xlen + 8u) { b->err = "truncated or malformed BGZF block"; return VSV_E_INVALID; }
    wait_for(o + (size_t)bsize + 1);
    if (stage_failed) { b->err = "cannot read the BAM file"; return VSV_E_INVALID; }
    const uint8_t* tr = hdr + bsize + 1 - 8;
    crcs.push_back((uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24));
    coff.push_back(o + 12 + xlen);
    isz.push_back((uint32_t)tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24));
    o += (size_t)bsize + 1;
  }
  wait_for((size_t)fsize);
  if (stage_failed) { b->err = "cannot read the BAM file"; return VSV_E_INVALID; }
  coff.push_back((uint64_t)fsize);
  const double t2 = now();
  const char* names = nullptr; int64_t names_len = 0;
  vsv_bgzf_set_expected_crc(h, crcs.data(), (int64_t)crcs.size());     // the trailers' CRC-32s are checked on the GPU
  const int st = vsv_bam_parse_device(h, comp, coff.data(), isz.data(), (int64_t)isz.size(), first_record, (int32_t)b->ref_names.size(), tid, out,
                                      &names, &names_len, &b->dev_l_seq, &b->dev_sam_flag);
  vsv_bgzf_set_expected_crc(h, nullptr, 0);
  if (st) { b->err = std::string("device BAM parse failed: ") + vsv_last_error(h); return st; }
  b->qname_blob.clear();
  b->dev_names = names; b->dev_names_len = names_len;     // valid until the handle's next device parse
  b->sa_blob.clear();
  if (timing) fprintf(stderr, "[vsv_bam_load_device] file map %.1f ms, member table %.1f ms (%zu members), device parse %.1f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, isz.size(), (now() - t2) * 1e3);
  return 0;
}
const uint32_t* vsv_bam_l_seq_device(vsv_bam* b) { return b ? b->dev_l_seq : nullptr; }
const uint32_t* vsv_bam_sam_flags_device(vsv_bam* b) { return b ? b->dev_sam_flag : nullptr; }

/* '\n'-joined query names in qid order / SA tags in record order of the last vsv_bam_load; *len receives the length */
const char* vsv_bam_qnames(vsv_bam* b, int64_t* len) {
  if (b && b->dev_names) { if (len) *len = b->dev_names_len; return b->dev_names; }
  if (len) *len = b ? (int64_t)b->qname_blob.size() : 0;
  return b ? b->qname_blob.data() : "";
}
const char* vsv_bam_sa_tags(vsv_bam* b, int64_t* len) { if (len) *len = b ? (int64_t)b->sa_blob.size() : 0; return b ? b->sa_blob.data() : ""; }
void vsv_bam_set_keep_seq(vsv_bam* b, int keep) { if (b) b->keep_seq = keep != 0; }
const uint8_t* vsv_bam_seq(vsv_bam* b, int64_t* len) { if (len) *len = b ? (int64_t)b->seq.size() : 0; return b ? b->seq.data() : nullptr; }
const uint32_t* vsv_bam_l_seq(vsv_bam* b) { return b ? b->l_seq.data() : nullptr; }
const uint32_t* vsv_bam_sam_flags(vsv_bam* b) { return b ? b->sam_flag.data() : nullptr; }

}  // extern "C"
