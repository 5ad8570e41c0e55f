"""Host mirror of Large_INDEL/remove_redundancy.py — the last step of Raw_variant_call.py (RV:99-104): calls that describe the
same event are linked (windowed pair predicates), linked groups collapse to their longest member.

The pair matching — match_del_chr / match_ins_chr with the size / overlap / sequence-similarity predicates, including the
edit distance of the inserted sequences (edlib in the reference) — runs on the GPU through vsv_redundancy_pairs. Connected
components (networkx in the reference; a union-find here, components numbered in the order networkx yields them: by first
appearance in the link list), the choice of the survivor and the VCF text stay on the host.

Canonical choices where the reference is not deterministic: calls of a chromosome are ordered by position with ties in
file order (the reference's np.argsort is unstable); among equally long members of a component the first in that order
survives (the reference takes the first of a `list(set)`, whose order depends on the interpreter's hash seed)."""
import os

import numpy as np

from .engine import Engine

ALPHABET = {ch: i for i, ch in enumerate("ACGTN")}


def sort_sig_per_chr(sig_list):
    """RR:39-49 with the stable tie rule."""
    order = np.argsort([sig[1] for sig in sig_list], kind="stable") if sig_list else []
    return [sig_list[i] for i in order]


def sort_sig(sig_list):
    """RR:27-36: chr1..chr22 only, each sorted by position; calls on other contigs are dropped, as in the reference."""
    out = []
    for i in range(1, 23):
        chr_name = 'chr' + str(i)
        out.extend(sort_sig_per_chr([sig for sig in sig_list if sig[0] == chr_name]))
    return out


def vcf_to_sig(vcf_path):
    """RR:52-73: (sorted DEL records, sorted INS records, {id: record}, header with the CollapseId INFO line)."""
    del_sig, ins_sig, dc, header = [], [], {}, []
    with open(vcf_path, 'r') as f:
        for line in f:
            if line[0] != '#':
                data = line.split()
                data[1] = int(data[1])
                data[3] = data[3].upper()
                data[4] = data[4].upper()
                dc[data[2]] = data
                if 'SVTYPE=DEL' in line:
                    del_sig.append(data)
                elif 'SVTYPE=INS' in line:
                    ins_sig.append(data)
            else:
                header.append(line)
    add_line = "##INFO=<ID=CollapseId,Number=1,Type=Integer,Description=\"collapse match ID\">\n"
    header = header[:-2] + [add_line] + header[-2:]
    return sort_sig(del_sig), sort_sig(ins_sig), dc, header


def _encode(seqs):
    """ALT strings -> (uint8 codes 0..15, uint64 offsets); symbols outside ACGTN get the next free codes."""
    table = dict(ALPHABET)
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in seqs], out=off[1:])
    blob = np.frombuffer("".join(seqs).encode("latin-1"), dtype=np.uint8)
    lut = np.full(256, 255, dtype=np.uint8)
    for ch, code in table.items():
        lut[ord(ch)] = code
    for b in np.unique(blob):
        if lut[b] == 255:
            lut[b] = len(table)
            table[chr(b)] = len(table)
    if len(table) > 16:
        raise ValueError("more than 16 distinct symbols in the ALT sequences")
    return lut[blob], off


def match_chr(sig_list, is_del, eng, params):
    """match_del_chr (RR:115-134) / match_ins_chr (RR:162-181): the link list [(id1, id2), ...] in the reference's order
    (for every call, its matching partners in position order — each matching pair appears in both directions)."""
    if len(sig_list) < 2:
        return []
    pos = [sig[1] for sig in sig_list]
    svlen = [abs(len(sig[3]) - len(sig[4])) for sig in sig_list]
    seq, seq_off = (None, None) if is_del else _encode([sig[4] for sig in sig_list])
    pairs = eng.redundancy_pairs(is_del, pos, svlen, seq, seq_off, params)
    both = np.concatenate([pairs, pairs[:, ::-1]]) if len(pairs) else pairs
    both = both[np.lexsort((both[:, 1], both[:, 0]))] if len(both) else both
    return [(sig_list[int(a)][2], sig_list[int(b)][2]) for a, b in both]


def connected_components(links):
    """nx.connected_components(G) after G.add_edges_from(links) (RR:143-146): node sets, in order of first appearance."""
    parent, order = {}, []

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    for a, b in links:
        for x in (a, b):
            if x not in parent:
                parent[x] = x
                order.append(x)
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[rb] = ra
    comps, index = [], {}
    for x in order:
        r = find(x)
        if r not in index:
            index[r] = len(comps)
            comps.append([])
        comps[index[r]].append(x)
    return comps


def match(sig_list, is_del, eng, params):
    """match_del / match_ins (RR:135-147, 184-196)."""
    links = []
    for i in range(1, 23):
        chr_name = 'chr' + str(i)
        links.extend(match_chr([sig for sig in sig_list if sig[0] == chr_name], is_del, eng, params))
    return connected_components(links)


def pick_best_sv(vcf_dc, nodes_list, rank):
    """RR:199-224: ({kept id: component}, {removed id: component}); the longest member stays (first in sorted order on ties)."""
    retain_index, remove_index = {}, {}
    for i, nodes in enumerate(nodes_list):
        members = sorted(nodes, key=lambda x: rank[x])
        ll = [abs(len(vcf_dc[x][3]) - len(vcf_dc[x][4])) for x in members]
        best = members[ll.index(max(ll))]
        retain_index[best] = i
        for x in members:
            if x != best:
                remove_index[x] = i
    return retain_index, remove_index


def run(vcf_path, output_dir, dist_thresh=500, dist_thresh_del=3000, overlap_thresh=0, size_sim_thresh=0.5, size_sim_thresh_del=0.1,
        seq_sim_thresh=0.5, device=0, engine=None):
    """The script body (RR:257-298): writes <output_dir>/volcano_variant_no_redundancy.vcf and ..._redundancy.vcf."""
    os.makedirs(output_dir, exist_ok=True)
    del_sig, ins_sig, vcf_dc, header = vcf_to_sig(vcf_path)
    rank = {sig[2]: k for k, sig in enumerate(del_sig + ins_sig)}
    eng = engine or Engine(device)
    try:
        p = eng.redundancy_params(dist_thresh=dist_thresh, dist_thresh_del=dist_thresh_del, overlap_thresh=float(overlap_thresh),
                                  size_sim_thresh=float(size_sim_thresh), size_sim_thresh_del=float(size_sim_thresh_del),
                                  seq_sim_thresh=float(seq_sim_thresh))
        nodes_del = match(del_sig, True, eng, p)
        nodes_ins = match(ins_sig, False, eng, p)
    finally:
        if engine is None:
            eng.close()
    retain_del, remove_del = pick_best_sv(vcf_dc, nodes_del, rank)
    retain_ins, remove_ins = pick_best_sv(vcf_dc, nodes_ins, rank)
    retain_sig, remove_sig = [], []
    for idx, sig in vcf_dc.items():                                          # write_vcf (RR:236-262)
        if idx in retain_del:
            sig[7] = sig[7] + ";CollapseId=DEL%d" % retain_del[idx]
            retain_sig.append(sig)
        elif idx in retain_ins:
            sig[7] = sig[7] + ";CollapseId=INS%d" % retain_ins[idx]
            retain_sig.append(sig)
        elif idx in remove_del:
            sig[7] = sig[7] + ";CollapseId=DEL%d" % remove_del[idx]
            remove_sig.append(sig)
        elif idx in remove_ins:
            sig[7] = sig[7] + ";CollapseId=INS%d" % remove_ins[idx]
            remove_sig.append(sig)
        else:
            retain_sig.append(sig)
    for sigs, name in ((sort_sig(remove_sig), '_redundancy.vcf'), (sort_sig(retain_sig), '_no_redundancy.vcf')):
        with open(output_dir + "/volcano_variant" + name, 'w') as f:
            f.writelines(header)
            for sig in sigs:
                f.write('\t'.join([sig[0], str(sig[1])] + sig[2:]) + '\n')
    print("original %d lines" % (len(retain_sig) + len(remove_sig)))
    print("new vcf %d lines" % len(retain_sig))
    print("redundancy %d lines" % len(remove_sig))
    return nodes_del, nodes_ins
