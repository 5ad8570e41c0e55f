#!/usr/bin/env python3
"""Generates tests/golden/bnd_*.json from the reference's svim-asm functions (build container only).

SVIM_inter.py / SVCandidate.py import directly (numpy + scipy present); form_partitions, pair_haplotypes_breakends,
span_position_distance_breakends and sorted_nicely are AST-extracted from SVIM_COMBINE.py because that module imports
edlib at the top. Inputs are synthetic split contigs (segments of a primary alignment + its
SA-tag alignments); the fixture stores only inputs and the functions' outputs.
"""
import ast
import json
import os
import re
import sys
import types

import numpy as np

REF = os.environ.get("VSV_REFERENCE", "/root/reference")
SV = os.path.join(REF, "bin/VolcanoSV-vc/Complex_SV/svim-asm-1.0.2/src/svim_asm")
HERE = os.path.dirname(os.path.abspath(__file__))

CONTIGS = [("chr%d" % (i + 1), 3000000 + 100000 * i) for i in range(12)]


class FakeBam:
    def __init__(self, contigs):
        self.names = [c[0] for c in contigs]
        self.lens = {c[0]: c[1] for c in contigs}

    def get_reference_name(self, i):
        return self.names[i]

    getrname = get_reference_name

    def get_reference_length(self, name):
        return self.lens[name]


class FakeAln:
    def __init__(self, name, seg):
        self.query_name = name
        self.reference_id, self.reference_start, self.reference_end = seg[0], seg[1], seg[2]
        self.query_alignment_start, self.query_alignment_end = seg[3], seg[4]
        self._rl = seg[5]
        self.is_reverse = bool(seg[6])
        self.query_sequence = ""

    def infer_read_length(self):
        return self._rl


def make_reads(seed, n_events=160, dense=3):
    """Split contigs: each event = a chain of 2-4 segments on the forward read; hp2 carries a jittered copy of ~70 % of
    the hp1 events plus private ones; a few dense clusters (> 10 members within 1 kb) must be dropped."""
    rng = np.random.default_rng(seed)
    reads = []

    def one_read(hap, name, chain, L):
        segs = []
        for (tid, rs, re_, qs, qe, rev) in chain:
            if rev:
                segs.append([tid, rs, re_, L - qe, L - qs, L, 1])
            else:
                segs.append([tid, rs, re_, qs, qe, L, 0])
        # primary first, order of the others as in the SA tag (arbitrary): shuffle the tail
        order = [0] + list(rng.permutation(np.arange(1, len(segs))))
        reads.append({"hap": hap, "name": name, "segs": [segs[i] for i in order]})

    for e in range(n_events):
        nseg = int(rng.integers(2, 5))
        L = int(rng.integers(30000, 90000))
        cuts = sorted(rng.choice(np.arange(2000, L - 2000), nseg - 1, replace=False).tolist())
        bounds = [0] + cuts + [L]
        chain = []
        tid = int(rng.integers(0, len(CONTIGS)))
        pos = int(rng.integers(100000, CONTIGS[tid][1] - 500000))
        rev = bool(rng.random() < 0.4)
        for k in range(nseg):
            qs, qe = bounds[k], bounds[k + 1]
            if k > 0:
                qs += int(rng.choice([-80, -50, -49, -10, 0, 0, 10, 50, 51, 90]))   # overlap / gap on the read (tolerance 50)
                kind = int(rng.integers(0, 6))
                if kind == 0:      # other chromosome
                    tid = int((tid + rng.integers(1, len(CONTIGS))) % len(CONTIGS))
                    pos = int(rng.integers(100000, CONTIGS[tid][1] - 500000))
                    if rng.random() < 0.5:
                        rev = not rev
                elif kind == 1:    # same chromosome, far away (deviation < -max_sv_size)
                    pos = pos + int(rng.integers(100001, 400000)) * (1 if rng.random() < 0.7 else -1)
                    pos = max(1000, min(pos, CONTIGS[tid][1] - 200000))
                elif kind == 2:    # deletion-sized jump (not a BND)
                    pos = pos + int(rng.integers(40, 90000))
                elif kind == 3:    # orientation switch nearby / far
                    rev = not rev
                    pos = pos + int(rng.choice([0, 500, 5000, 150000, -150000]))
                    pos = max(1000, min(pos, CONTIGS[tid][1] - 200000))
                elif kind == 4:    # reference overlap (tandem-dup like) near or very far back
                    pos = pos - int(rng.choice([200, 3000, 120000]))
                    pos = max(1000, pos)
                # kind 5: contiguous
            span = qe - qs
            rs = pos
            re_ = rs + max(50, span + int(rng.integers(-30, 31)))
            chain.append((tid, rs, re_, max(0, qs), qe, rev))
            pos = re_ if not rev else max(1000, rs - span)
        one_read(1, "PS%d_hp1_c%d" % (1000 + e, e), chain, L)
        if rng.random() < 0.7:
            j = lambda: int(rng.integers(-300, 301))
            chain2 = [(t, max(0, rs + j()), 0, qs, qe, rv) for (t, rs, re_, qs, qe, rv) in chain]
            chain2 = [(t, rs, rs + (chain[i][2] - chain[i][1]) + int(rng.integers(-5, 6)), qs, qe, rv) for i, (t, rs, _, qs, qe, rv) in enumerate(chain2)]
            one_read(2, "PS%d_hp2_c%d" % (1000 + e, e), chain2, L)
    for e in range(n_events // 4):   # hp2-private events
        L = 40000
        t1, t2 = int(rng.integers(0, 12)), int(rng.integers(0, 12))
        p1, p2 = int(rng.integers(100000, 2000000)), int(rng.integers(100000, 2000000))
        one_read(2, "PS%d_hp2_p%d" % (5000 + e, e), [(t1, p1, p1 + 20000, 0, 20000, False), (t2, p2, p2 + 20000, 20000, 40000, bool(rng.random() < 0.5))], L)
    for d in range(dense):           # dense partitions: 6-8 reads per hap at one breakpoint (> 10 members -> ignored)
        t1, t2 = d, d + 5
        p1, p2 = 1500000 + d * 1000, 800000
        for hap in (1, 2):
            for k in range(int(rng.integers(6, 9))):
                a, b = p1 + int(rng.integers(0, 90)), p2 + int(rng.integers(0, 90))
                one_read(hap, "PS%d_hp%d_d%d_%d" % (9000 + d, hap, d, k), [(t1, a - 20000, a, 0, 20000, False), (t2, b, b + 20000, 20000, 40000, False)], 40000)
    return reads


def load_svim():
    sys.path.insert(0, SV)
    import SVIM_inter
    import SVCandidate
    src = open(os.path.join(SV, "SVIM_COMBINE.py")).read()
    tree = ast.parse(src)
    keep = {"form_partitions", "span_position_distance_breakends", "pair_haplotypes_breakends", "sorted_nicely"}
    fdefs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in keep]
    from scipy.cluster.hierarchy import fcluster, linkage
    ns = {"np": np, "linkage": linkage, "fcluster": fcluster, "re": re, "logging": types.SimpleNamespace(debug=lambda *a: None, error=lambda *a: None)}
    exec(compile(ast.Module(body=fdefs, type_ignores=[]), "SVIM_COMBINE.py", "exec"), ns)
    return SVIM_inter, SVCandidate, ns


def make_doc(reads, svim=None):
    """The fixture document (inputs + the reference's outputs) for one list of reads."""
    SVIM_inter, SVCandidate, ns = svim or load_svim()
    opts = types.SimpleNamespace(min_mapq=20, min_sv_size=40, max_sv_size=100000, query_gap_tolerance=50, query_overlap_tolerance=50,
                                 reference_gap_tolerance=50, reference_overlap_tolerance=50, partition_max_distance=1000)
    bam = FakeBam(CONTIGS)
    per_read, cands = [], {1: [], 2: []}
    for r in reads:
        alns = [FakeAln(r["name"], s) for s in r["segs"]]
        out = SVIM_inter.analyze_read_segments(alns[0], alns[1:], bam, opts)
        b = [c for c in out if c.type == "BND"]
        per_read.append([[c.source_contig, c.source_start, c.source_direction, c.dest_contig, c.dest_start, c.dest_direction] for c in b])
        cands[r["hap"]].extend(b)
    both = [(1, c) for c in cands[1]] + [(2, c) for c in cands[2]]
    partitions = ns["form_partitions"](both, opts.partition_max_distance)
    clusters = ns["pair_haplotypes_breakends"](partitions)
    paired = []
    for cl in clusters:
        c = cl[0][1]
        if len(cl) == 1:
            gt, rd = ("1/0" if cl[0][0] == 1 else "0/1"), c.reads
        elif len(cl) == 2:
            gt, rd = "1/1", cl[0][1].reads + cl[1][1].reads
        else:
            continue
        paired.append(SVCandidate.CandidateBreakend(c.source_contig, c.source_start, c.source_direction, c.dest_contig, c.dest_start,
                                                    c.dest_direction, rd, bam, gt))
    entries = []
    for c in paired:   # SVIM_COMBINE.py:461-464
        entries.append(((c.get_source()[0], c.get_source()[1] + 1, c.get_source()[1] + 2), c.get_vcf_entry(True), "BND"))
        entries.append(((c.get_destination()[0], c.get_destination()[1] + 1, c.get_destination()[1] + 2), c.get_vcf_entry_reverse(True), "BND"))
    lines, n = [], 0
    for source, entry, svtype in ns["sorted_nicely"](entries):
        n += 1
        lines.append(entry.replace("PLACEHOLDERFORID", "svim_asm.BND.%d" % n, 1))
    doc = {"contigs": CONTIGS, "reads": reads,
           "expected": {"per_read": per_read,
                        "paired": sorted([[c.source_contig, c.source_start, c.source_direction, c.dest_contig, c.dest_start, c.dest_direction, c.genotype, c.reads] for c in paired]),
                        "n_partitions": len(partitions), "n_dropped_partitions": sum(1 for p in partitions if len(p) > 10),
                        "vcf": lines},
           "generator": "tests/golden/make_golden_bnd.py"}
    return doc, len(both), paired


def main():
    svim = load_svim()
    for name, seed in (("bnd_a", 11), ("bnd_b", 12)):
        reads = make_reads(seed)
        doc, n_both, paired = make_doc(reads, svim)
        path = os.path.join(HERE, name + ".json")
        with open(path, "w") as f:
            json.dump(doc, f, separators=(",", ":"))
        print("wrote %s: %d reads, %d candidates, %d partitions (%d dropped), %d paired (%d 1/1), %d bytes" % (
            path, len(reads), n_both, doc["expected"]["n_partitions"], doc["expected"]["n_dropped_partitions"], len(paired),
            sum(1 for c in paired if c.genotype == "1/1"), os.path.getsize(path)))


if __name__ == "__main__":
    main()
