"""Synthetic alignment-record streams of the BASELINE.json shapes, generated with torch so
the same code fills host tensors (tests, CPU baseline sample) or HBM directly (bench, Philox on device).

Shapes
  hifi    config 2: k ~ Poisson(16), n_ops = 2k+1, M U[50,1000) alternating with I/D U[1,6); 1 % of records
          carry one >=50 bp event U[50,5000); mapq 60 w.p. 0.95; 1 % of names own a clipped split pair.
  ont     config 3: k ~ Poisson(100), M U[5,80), indel Geometric(0.5); 3 % events; 10 % split names; mapq 60 w.p. 0.8.
  contig  config 2c: k ~ Poisson(8000) (Mb-scale contigs), 1 % ... per-op event rate as hifi.
Records are (tid, pos)-sorted like a coordinate-sorted BAM; hp flag = record parity (hp1/hp2 names).
"""
import numpy as np
import torch

from .abi import F_HP1, F_HP2, F_REVERSE
from .soa import RecordSoA

SHAPES = {
    # k_mean, M lo, M hi, indel small (lo,hi) or geometric, event fraction, split fraction, p(mapq 60)
    "hifi": dict(k=16.0, m_lo=50, m_hi=1000, small=(1, 6), geometric=False, ev=0.01, split=0.01, mq=0.95),
    "ont": dict(k=100.0, m_lo=5, m_hi=80, small=None, geometric=True, ev=0.03, split=0.10, mq=0.80),
    "contig": dict(k=8000.0, m_lo=50, m_hi=1000, small=(1, 6), geometric=False, ev=0.01, split=0.01, mq=0.95),
}
CHR10_LEN = 135534747  # hg19 chr10 (reference Large_INDEL/header:14)


def _randint(gen, lo, hi, n, device):
    return torch.randint(int(lo), int(hi), (int(n),), generator=gen, device=device, dtype=torch.int64)


def _rand(gen, n, device):
    return torch.rand(int(n), generator=gen, device=device)


def generate(n_records, shape="hifi", seed=20250330, tid=0, chrom_len=CHR10_LEN, device="cpu", events_per_record=None,
             site_step=5000, max_chunk_ops=400_000_000):
    """Returns (dict of torch tensors in the C-ABI SoA layout, n_qids, n_tids). Large streams are generated in position
    slices of at most ~max_chunk_ops CIGAR ops each (the ragged-array temporaries are int64), so that config 3
    (50 M ONT-like records, 10^10 ops) fits in HBM while it is being built."""
    cfg = SHAPES[shape]
    est_ops = n_records * (2.0 * cfg["k"] + 1.0)
    n_chunks = int(min(max(1, -(-est_ops // max_chunk_ops)), max(1, n_records // 1000)))
    if n_chunks > 1 and shape != "contig":
        parts, per = [], n_records // n_chunks
        for c in range(n_chunks):
            n_c = per if c < n_chunks - 1 else n_records - per * (n_chunks - 1)
            lo, hi = chrom_len * c // n_chunks, chrom_len * (c + 1) // n_chunks
            t, nq, _ = _generate_slice(n_c, shape, seed + 7919 * c, tid, lo, hi, chrom_len, device, events_per_record, site_step)
            parts.append((t, nq))
        t, nq = concat(parts)
        return t, nq, int(tid) + 1
    if n_chunks > 1:
        # contig shape: a record spans megabases, so position slices cannot keep split mates apart. Independent batches over the
        # whole chromosome instead (the ragged-array temporaries of one batch fit in HBM), merged by one stable sort on pos.
        parts, per = [], n_records // n_chunks
        for c in range(n_chunks):
            n_c = per if c < n_chunks - 1 else n_records - per * (n_chunks - 1)
            t, nq, _ = _generate_slice(n_c, shape, seed + 7919 * c, tid, 0, chrom_len, chrom_len, device, events_per_record, site_step)
            parts.append((t, nq))
        t, nq = concat(parts)
        del parts
        return _sort_records(t), nq, int(tid) + 1
    return _generate_slice(n_records, shape, seed, tid, 0, chrom_len, chrom_len, device, events_per_record, site_step)


def _sort_records(t, block_ops=200_000_000):
    """Stable sort of a record SoA by pos (one tid): scalars are permuted, the ragged CIGAR array is gathered in blocks of output
    records so that the int64 index temporaries stay bounded; query ids are renumbered in first-appearance order (the ingest's
    numbering, which the split stage's repeat test is exact and cheapest for)."""
    dev = t["pos"].device
    n = int(t["pos"].numel())
    order = torch.sort(t["pos"], stable=True).indices
    off = t["cigar_off"]
    nops = (off[1:] - off[:-1])[order]
    new_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    new_off[1:] = torch.cumsum(nops, 0)
    src_start = off[:-1][order]
    cigar = torch.empty_like(t["cigar"])
    a = 0
    while a < n:
        # largest b with new_off[b] - new_off[a] <= block_ops (at least one record)
        b = int(torch.searchsorted(new_off, new_off[a] + block_ops, right=True)) - 1
        b = min(max(b, a + 1), n)
        lo, hi = int(new_off[a]), int(new_off[b])
        rec = torch.repeat_interleave(torch.arange(a, b, device=dev), nops[a:b])
        src = src_start[rec] + (torch.arange(lo, hi, device=dev) - new_off[rec])
        cigar[lo:hi] = t["cigar"][src]
        del rec, src
        a = b
    out = {k: t[k][order] for k in ("pos", "tid", "mapq", "flag")}
    q = t["qid"][order].to(torch.int64)
    first = torch.full((int(q.max()) + 1,), n, dtype=torch.int64, device=dev).scatter_reduce_(0, q, torch.arange(n, device=dev), "amin")
    rank = torch.empty_like(first)
    rank[torch.sort(first, stable=True).indices] = torch.arange(first.numel(), device=dev)
    out["qid"] = rank[q].to(torch.int32)
    out["cigar_off"] = new_off
    out["cigar"] = cigar
    return out


def _generate_slice(n_records, shape, seed, tid, pos_lo, pos_hi, chrom_len, device, events_per_record, site_step):
    """One position slice [pos_lo, pos_hi): records sorted by pos; split mates are only planted for records whose mate
    still falls inside the slice, so concatenated slices stay coordinate-sorted."""
    cfg = SHAPES[shape]
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    n_mates = int(round(n_records * cfg["split"] / (1.0 + cfg["split"])))
    nb = int(n_records) - n_mates  # base records
    # ---- base records -------------------------------------------------------------------------------
    k = torch.poisson(torch.full((nb,), cfg["k"], device=dev), generator=gen).to(torch.int64)
    n_ops = 2 * k + 1
    pos = torch.sort(_randint(gen, pos_lo, max(pos_lo + 1, min(pos_hi, chrom_len - 40000)), nb, dev)).values
    has_mate = torch.zeros(nb, dtype=torch.bool, device=dev)
    if n_mates > 0:
        sel = torch.randperm(nb, generator=gen, device=dev)[:n_mates]
        has_mate[sel] = True
        if pos_hi < chrom_len:   # interior slice: a mate (pos + span + <1 kb) must not spill into the next slice
            max_span = (2 * k + 1) * cfg["m_hi"] + 6000
            has_mate &= (pos + max_span) < pos_hi
    n_mates = int(has_mate.sum())
    n_ops_b = n_ops + has_mate.to(torch.int64)  # +1 tail clip
    off_b = torch.zeros(nb + 1, dtype=torch.int64, device=dev)
    off_b[1:] = torch.cumsum(n_ops_b, 0)
    tot = int(off_b[-1])
    rec_of = torch.repeat_interleave(torch.arange(nb, device=dev), n_ops_b)
    j = torch.arange(tot, device=dev) - off_b[rec_of]           # op index inside the record
    is_m = (j % 2 == 0)
    m_len = _randint(gen, cfg["m_lo"], cfg["m_hi"], tot, dev)
    if cfg["geometric"]:
        u = _rand(gen, tot, dev).clamp_(min=1e-12)
        small = (torch.floor(torch.log2(1.0 / u)).to(torch.int64) + 1).clamp_(max=60)   # Geometric(0.5) on {1,2,...}
    else:
        small = _randint(gen, cfg["small"][0], cfg["small"][1], tot, dev)
    is_del = _rand(gen, tot, dev) < 0.5
    op = torch.where(is_m, torch.zeros_like(j), torch.where(is_del, torch.full_like(j, 2), torch.ones_like(j)))
    ln = torch.where(is_m, m_len, small)
    # events: a record carries one >=50 bp I/D taken from a grid of shared "truth" sites (one every `site_step` bp), so
    # overlapping records of both haplotypes report the same SV with small position / length jitter and the
    # clustering and pairing stages have real work. The M op containing the site is cut at the site and the I/D
    # op after it becomes the event; type and base length U[50,5000) are a hash of the site index.
    ev_frac = cfg["ev"] if events_per_record is None else events_per_record
    mean_span = (cfg["k"] + 1.0) * 0.5 * (cfg["m_lo"] + cfg["m_hi"])
    # probability that an M op crossing a site carries the event: events/record ~= ev_frac (contig shape: every
    # 25th site, i.e. tens of events per Mb-scale contig)
    p_cross = 0.04 if shape == "contig" else min(1.0, ev_frac * site_step / mean_span)
    r_adv0 = torch.where((op == 0) | (op == 2), ln, torch.zeros_like(ln))
    csum = torch.cumsum(r_adv0, 0)
    rec_base = torch.zeros(nb, dtype=torch.int64, device=dev)
    rec_base[1:] = csum[off_b[1:-1] - 1]
    ref_start = pos[rec_of] + (csum - r_adv0) - rec_base[rec_of]          # reference offset before each op
    site = ref_start // site_step + 1                                      # first grid site right of the op start
    g = site * site_step
    crosses = is_m & (j + 1 < 2 * k[rec_of] + 1) & (g < ref_start + ln) & (ln >= 3)
    hit = crosses & (_rand(gen, tot, dev) < p_cross)
    m_idx = torch.nonzero(hit).flatten()
    if m_idx.numel():
        ne = m_idx.numel()
        gj = g[m_idx] + _randint(gen, -20, 21, ne, dev)
        gj = torch.minimum(torch.maximum(gj, ref_start[m_idx] + 1), ref_start[m_idx] + ln[m_idx] - 1)
        h = (site[m_idx] * 2654435761) & 0xFFFFFFFF
        base_len = 50 + (h >> 8) % 4950
        scale = torch.where(_rand(gen, ne, dev) < 0.1, 0.3, 1.0) * (0.95 + 0.1 * _rand(gen, ne, dev))
        ev_len = (base_len.to(torch.float64) * scale.to(torch.float64)).to(torch.int64).clamp_(min=50)
        ln[m_idx] = gj - ref_start[m_idx]
        op[m_idx + 1] = torch.where((h & 1) == 1, 2, 1)
        ln[m_idx + 1] = ev_len
    # per-record sums needed for the planted split pairs
    q_adv = torch.where((op == 0) | (op == 1), ln, torch.zeros_like(ln))
    r_adv = torch.where((op == 0) | (op == 2), ln, torch.zeros_like(ln))
    tail = (j == n_ops_b[rec_of] - 1) & has_mate[rec_of]
    q_adv = torch.where(tail, torch.zeros_like(ln), q_adv)
    r_adv = torch.where(tail, torch.zeros_like(ln), r_adv)
    q1 = torch.zeros(nb, dtype=torch.int64, device=dev).index_add_(0, rec_of, q_adv)
    rspan = torch.zeros(nb, dtype=torch.int64, device=dev).index_add_(0, rec_of, r_adv)
    mapq = torch.where(_rand(gen, nb, dev) < cfg["mq"], torch.full((nb,), 60, device=dev, dtype=torch.int64),
                       _randint(gen, 0, 50, nb, dev))
    rev = _rand(gen, nb, dev) < 0.5
    # ---- mates (second segment of a split alignment) --------------------------------------------------
    mi = torch.nonzero(has_mate).flatten()
    nm = mi.numel()
    G = _randint(gen, 0, 1000, nm, dev)                       # Ref2s - Ref1e
    D = _randint(gen, 50, 5000, nm, dev) * torch.where(_rand(gen, nm, dev) < 0.5, 1, -1)   # planted Diffdis
    D = torch.where(q1[mi] + G - D >= 1, D, -D.abs())
    h2 = q1[mi] + G - D                                        # head clip of the mate = Read2s
    q2 = (D - G).clamp_(min=0) + _randint(gen, 200, 2000, nm, dev)
    t1 = G - D + q2                                            # tail clip of the first segment
    clip1 = torch.where(_rand(gen, nm, dev) < 0.5, 4, 5)
    clip2 = torch.where(_rand(gen, nm, dev) < 0.5, 4, 5)
    tail_idx = off_b[mi + 1] - 1
    op[tail_idx] = clip1
    ln[tail_idx] = t1
    pos_m = pos[mi] + rspan[mi] + G
    rev_m = torch.where(_rand(gen, nm, dev) < 0.95, rev[mi], ~rev[mi])
    mapq_m = torch.where(_rand(gen, nm, dev) < cfg["mq"], torch.full((nm,), 60, device=dev, dtype=torch.int64),
                         _randint(gen, 0, 50, nm, dev))
    # ---- concatenate, sort by pos (stable), reorder the ragged CIGAR array ---------------------------
    n = nb + nm
    all_pos = torch.cat([pos, pos_m])
    all_nops = torch.cat([n_ops_b, torch.full((nm,), 2, dtype=torch.int64, device=dev)])
    all_qid = torch.cat([torch.arange(nb, device=dev), mi])
    all_mapq = torch.cat([mapq, mapq_m])
    all_rev = torch.cat([rev, rev_m])
    hp2 = (all_qid % 2 == 1)
    mate_op = torch.stack([clip2, torch.zeros_like(clip2)], 1).flatten()
    mate_ln = torch.stack([h2, q2], 1).flatten()
    op_all = torch.cat([op, mate_op])
    ln_all = torch.cat([ln, mate_ln])
    old_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    old_off[1:] = torch.cumsum(all_nops, 0)
    order = torch.sort(all_pos, stable=True).indices
    new_nops = all_nops[order]
    new_off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    new_off[1:] = torch.cumsum(new_nops, 0)
    rec_new = torch.repeat_interleave(torch.arange(n, device=dev), new_nops)
    src = old_off[order][rec_new] + (torch.arange(int(new_off[-1]), device=dev) - new_off[rec_new])
    packed = ((ln_all[src] << 4) | op_all[src]).to(torch.int64)
    flag = (all_rev[order].to(torch.int64) * F_REVERSE) | torch.where(hp2[order], F_HP2, F_HP1)
    out = {
        "pos": all_pos[order].to(torch.int32),
        "tid": torch.full((n,), int(tid), dtype=torch.int32, device=dev),
        # uint32 tensors are awkward in torch: qid / cigar keep int32 storage with the same bits
        "qid": all_qid[order].to(torch.int32),
        "cigar_off": new_off,   # int64 bits == uint64 (non-negative)
        "mapq": all_mapq[order].to(torch.uint8),
        "flag": flag.to(torch.uint8),
        "cigar": packed.to(torch.int32),
    }
    return out, nb, int(tid) + 1


def to_soa(tensors, n_qids=None):
    """torch dict -> host RecordSoA (numpy views of the same bits)."""
    t = {k: v.cpu().numpy() for k, v in tensors.items()}
    soa = RecordSoA(t["pos"], t["tid"], t["qid"].view(np.uint32), t["cigar_off"].view(np.uint64), t["mapq"], t["flag"],
                    t["cigar"].view(np.uint32))
    if n_qids:
        soa.n_qids = int(n_qids)
    return soa


def concat(parts):
    """Concatenates per-tid tensor dicts (tid-major, like a coordinate-sorted multi-chromosome BAM).
    qids of later parts are shifted so names stay distinct across chromosomes."""
    out = {}
    qshift, oshift = 0, 0
    offs = []
    for t, nq in parts:
        offs.append(t["cigar_off"][:-1] + oshift)
        oshift += int(t["cigar_off"][-1])
    last = torch.tensor([oshift], dtype=torch.int64, device=parts[0][0]["pos"].device)
    for name in ("pos", "tid", "mapq", "flag", "cigar"):
        out[name] = torch.cat([t[name] for t, _ in parts])
    qs = []
    for t, nq in parts:
        qs.append(t["qid"] + qshift)
        qshift += nq
    out["qid"] = torch.cat(qs)
    out["cigar_off"] = torch.cat(offs + [last])
    return out, qshift


# ---- BASELINE config 5: Complex_SV split-contig stream ----------------------------------------------------------------
HG19_LEN = [249250621, 243199373, 198022430, 191154276, 180915260, 171115067, 159138663, 146364022, 141213431, 135534747, 135006516,
            133851895, 115169878, 107349540, 102531392, 90354753, 81195210, 78077248, 59128983, 63025520, 48129895, 51304566]   # chr1..22 (Large_INDEL/header:5-26)


def generate_bnd(n_events, seed=20250333, n_tids=22, dense_frac=0.01):
    """Split contigs of SURVEY §8d config 5 as a bnd.SegmentSoA in collection order (hp1 reads, then hp2 reads): every event
    is a contig of 2-4 aligned segments (each junction a breakend: half of them inter-chromosomal, the others > 1 Mb apart on
    one chromosome), its hp2 copy jittered by U[-900,900] on every segment (SURVEY writes U[-300,300] "so that about half pair": with
    the reference's rule |d1| + |d2| <= 900 it takes +-900 for half of the junctions to pair), and 1 % of the events replicated 12 times within a
    few bp (partitions of more than 10 members, which the reference drops, SVIM_COMBINE.py:151-152).
    Returns (SegmentSoA, primary_tid[n_reads])."""
    from . import bnd
    rng = np.random.default_rng(seed)
    contigs = [("chr%d" % (i + 1), HG19_LEN[i]) for i in range(n_tids)]
    lens = np.array([c[1] for c in contigs], np.int64)
    n_dense = int(n_events * dense_frac)
    k = rng.integers(2, 5, n_events)                                  # segments per contig
    reps = np.ones(n_events, np.int64)
    reps[:n_dense] = 12
    # one row per (event, segment)
    ev_of = np.repeat(np.arange(n_events), k)
    j = np.arange(len(ev_of)) - np.repeat(np.cumsum(k) - k, k)          # segment index inside the event
    t0 = rng.integers(0, n_tids, n_events)
    tid = t0[ev_of].copy()
    inter = rng.random(len(ev_of)) < 0.5
    hop = rng.integers(1, n_tids, len(ev_of))
    # segment j > 0 jumps to another chromosome (inter) or stays on the previous one
    for step in range(1, 4):
        m = j == step
        prev = np.flatnonzero(m) - 1
        tid[m] = np.where(inter[m], (tid[prev] + hop[m]) % n_tids, tid[prev])
    span = 20000
    pos = (rng.random(len(ev_of)) * (lens[tid] - 4 * span - 2_000_000)).astype(np.int64) + span + 1_000_000
    rev = rng.random(len(ev_of)) < 0.5

    def reads_of(hap, jit_seed):
        r2 = np.random.default_rng(jit_seed)
        # replicate events: copy c of a dense event is shifted by c bp
        rep_ev = np.repeat(np.arange(n_events), reps)
        copy = np.arange(len(rep_ev)) - np.repeat(np.cumsum(reps) - reps, reps)
        first_seg = np.cumsum(k) - k
        seg_idx = np.repeat(first_seg[rep_ev], k[rep_ev]) + (np.arange(int(k[rep_ev].sum())) - np.repeat(np.cumsum(k[rep_ev]) - k[rep_ev], k[rep_ev]))
        shift = np.repeat(copy, k[rep_ev])
        jitter = r2.integers(-900, 901, len(seg_idx)) if hap == 2 else 0
        p = pos[seg_idx] + shift + jitter
        return dict(k=k[rep_ev], tid=tid[seg_idx], start=p, rev=rev[seg_idx], j=j[seg_idx])

    parts = [reads_of(1, seed + 1), reads_of(2, seed + 2)]
    kk = np.concatenate([p["k"] for p in parts])
    seg_off = np.zeros(len(kk) + 1, np.uint64)
    seg_off[1:] = np.cumsum(kk)
    jj = np.concatenate([p["j"] for p in parts])
    start = np.concatenate([p["start"] for p in parts])
    seg = bnd.SegmentSoA.from_arrays(
        contigs, seg_off, q_start=jj * span, q_end=(jj + 1) * span, ref_id=np.concatenate([p["tid"] for p in parts]),
        ref_start=start, ref_end=start + span, is_reverse=np.concatenate([p["rev"] for p in parts]).astype(np.uint8),
        hap=np.concatenate([np.full(len(parts[0]["k"]), 1, np.uint8), np.full(len(parts[1]["k"]), 2, np.uint8)]))
    primary_tid = seg.ref_id[seg.seg_off[:-1].astype(np.int64)]
    return seg, primary_tid
