"""Interface selection for the I-RMSD evaluation of a ``.dat`` file: what
/root/reference/scripts/Results/Benchmark/EvaluateBenchmark.py:44-113 does between ``parse_output`` and the RMSD
loop, with the pieces of DockingBenchmark.py / VisualizeBenchmark.py it calls:

  get_contacts (DockingBenchmark.py:291-327)        residues of the BOUND receptor / ligand with an atom pair closer than
                                                    ``contact_dist`` (5 A); standard amino acids only, residues with an
                                                    insertion code skipped -> lists of (chain, resnum, one-letter name)
  get_chain_seq (:23-36)                            per-chain sequence + residue numbers (same filters)
  get_best_match (:88-121) + get_alignment          bound chain -> unbound chain + residue alignment
      (scripts/Dataset/global_alignment.py:35-70)
  transfer_selection (:397-420)                     bound selection -> unbound selection through that alignment;
                                                    unaligned residues are dropped from BOTH, a residue-name mismatch
                                                    raises "Residues are not matching"
  select_CA / select_residues_list                  C-alpha coordinates of a selection, in the selection's order
      (VisualizeBenchmark.py:53-86)
  get_irmsd loop (EvaluateBenchmark.py:60-113)      per conformation the minimum superposed RMSD over interface pairs

Build-defined where the reference leans on absent packages (parity unpinned, no vectors in the reference tree):
BioPython's ``pairwise2.align.globaldx`` with BLOSUM62 is replaced by a Needleman-Wunsch global alignment with identity
scoring and free gaps (the bound / unbound chains of a benchmark entry are the same protein: both give the identity
mapping with gaps at missing residues); chains are paired by sequence identity first, C-alpha distance second; the
symmetric re-labellings of homo-oligomers (``get_symmetric_selections``, needs the external AnAnaS tool) are not
enumerated -- one assembly per target."""
from collections import OrderedDict

import numpy as np
import torch

THREE_TO_ONE = {"ALA": "A", "CYS": "C", "ASP": "D", "GLU": "E", "PHE": "F", "GLY": "G", "HIS": "H", "ILE": "I", "LYS": "K",
                "LEU": "L", "MET": "M", "ASN": "N", "PRO": "P", "GLN": "Q", "ARG": "R", "SER": "S", "THR": "T", "VAL": "V",
                "TRP": "W", "TYR": "Y"}


def read_structure(filename):
    """ATOM records of the first model -> dict of arrays: xyz (n,3) f64, chain, resnum, icode, resname, atomname
    (alternate locations other than ' ' / 'A' dropped, as the docking front end does)."""
    xyz, chain, resnum, icode, resname, atomname = [], [], [], [], [], []
    with open(filename) as fin:
        for line in fin:
            if line.startswith("ENDMDL"):
                break
            if not line.startswith("ATOM") or line[16] not in (" ", "A"):
                continue
            atomname.append(line[12:16].strip())
            resname.append(line[17:20].strip())
            chain.append(line[21])
            resnum.append(int(line[22:26]))
            icode.append(line[26])
            xyz.append((float(line[30:38]), float(line[38:46]), float(line[46:54])))
    return {"path": filename, "xyz": np.asarray(xyz, dtype=np.float64).reshape(-1, 3), "chain": np.asarray(chain),
            "resnum": np.asarray(resnum, dtype=np.int64), "icode": np.asarray(icode), "resname": np.asarray(resname),
            "atomname": np.asarray(atomname)}


def _standard(s):
    return np.array([(r in THREE_TO_ONE) and (i == " ") for r, i in zip(s["resname"], s["icode"])], dtype=bool)


def chain_sequences(s):
    """chain -> (one-letter sequence, residue numbers): DockingBenchmark.get_chain_seq per chain, file order."""
    out = OrderedDict()
    ok = _standard(s)
    for i in np.nonzero(ok)[0]:
        c, n = str(s["chain"][i]), int(s["resnum"][i])
        seq, nums = out.setdefault(c, ([], []))
        if not nums or nums[-1] != n:
            seq.append(THREE_TO_ONE[str(s["resname"][i])])
            nums.append(n)
    return OrderedDict((c, ("".join(v[0]), v[1])) for c, v in out.items())


def align_global(seq1, seq2):
    """Global alignment with identity scoring (+1 match, 0 mismatch) and free gaps -- the longest common subsequence --
    -> (pairs [(i, j)] of aligned IDENTICAL positions, identity = matches / alignment columns).  Stand-in for
    global_alignment.get_alignment (BioPython absent); residues that differ between the two chains come out unaligned and
    are dropped from a transferred selection, where the reference would raise "Residues are not matching"."""
    n, m = len(seq1), len(seq2)
    a = np.frombuffer(seq1.encode(), dtype=np.uint8)
    b = np.frombuffer(seq2.encode(), dtype=np.uint8)
    S = np.zeros((n + 1, m + 1), dtype=np.int32)
    for i in range(1, n + 1):
        S[i, 1:] = np.maximum.accumulate(np.maximum(S[i - 1, :-1] + (a[i - 1] == b), S[i - 1, 1:]))
    pairs, i, j = [], n, m
    while i > 0 and j > 0:
        if a[i - 1] == b[j - 1] and S[i, j] == S[i - 1, j - 1] + 1:
            pairs.append((i - 1, j - 1))
            i, j = i - 1, j - 1
        elif S[i, j] == S[i - 1, j]:
            i -= 1
        else:
            j -= 1
    pairs.reverse()
    cols = n + m - len(pairs)                  # every unmatched residue of either chain is a column of its own
    return pairs, (len(pairs) / float(cols) if cols else 0.0)


def _ca_of(s, chain, resnum):
    sel = np.nonzero((s["chain"] == chain) & (s["resnum"] == resnum) & (s["atomname"] == "CA") & (s["icode"] == " "))[0]
    return s["xyz"][sel[0]] if len(sel) else None


def best_chain_match(s1, s2, min_identity=0.9):
    """chain of s1 -> (chain of s2, identity, {residue index in s1's chain: residue index in s2's chain}):
    DockingBenchmark.get_best_match -- the partner is the chain with the highest identity (ties: the smaller mean C-alpha
    distance over the aligned residues, the reference's criterion); below ``min_identity`` the match is rejected like
    the reference's "Alignment is bad"."""
    c1, c2 = chain_sequences(s1), chain_sequences(s2)
    match = {}
    for ch1, (seq1, nums1) in c1.items():
        best = None
        for ch2, (seq2, nums2) in c2.items():
            if not seq1 or not seq2:
                raise Exception("Chain is empty")
            pairs, ident = align_global(seq1, seq2)
            d, k = 0.0, 0
            for i, j in pairs:
                p, q = _ca_of(s1, ch1, nums1[i]), _ca_of(s2, ch2, nums2[j])
                if p is not None and q is not None:
                    d, k = d + float(np.linalg.norm(p - q)), k + 1
            key = (-ident, d / max(k, 1))
            if best is None or key < best[0]:
                best = (key, ch2, ident, dict(pairs))
        if best[2] < min_identity:
            raise Exception("Alignment is bad", ch1, best[1], best[2])
        match[ch1] = (best[1], best[2], best[3])
    return match


def get_contacts(receptor, ligand, contact_dist=5.0):
    """Residues of the (bound) receptor / ligand with any atom pair within ``contact_dist`` ->
    (receptor selection, ligand selection), each a sorted list of (chain, resnum, one-letter name)."""
    from scipy.spatial import cKDTree
    ok_r, ok_l = _standard(receptor), _standard(ligand)
    ir, il = np.nonzero(ok_r)[0], np.nonzero(ok_l)[0]
    pairs = cKDTree(ligand["xyz"][il]).query_ball_tree(cKDTree(receptor["xyz"][ir]), contact_dist)
    rec_sel, lig_sel = set(), set()
    for a, hits in enumerate(pairs):
        if not hits:
            continue
        i = il[a]
        lig_sel.add((str(ligand["chain"][i]), int(ligand["resnum"][i]), THREE_TO_ONE[str(ligand["resname"][i])]))
        for h in hits:
            j = ir[h]
            rec_sel.add((str(receptor["chain"][j]), int(receptor["resnum"][j]), THREE_TO_ONE[str(receptor["resname"][j])]))
    return sorted(rec_sel), sorted(lig_sel)


def transfer_selection(selection_src, src, dst, match):
    """DockingBenchmark.transfer_selection: the residues of ``selection_src`` (on ``src``) as residues of ``dst``;
    returns (kept source selection, destination selection), aligned entry by entry."""
    seq_src, seq_dst = chain_sequences(src), chain_sequences(dst)
    kept, out = [], []
    for chain_id, res_num, res in selection_src:
        matching_chain, _, alignment = match[chain_id]
        res_idx = seq_src[chain_id][1].index(res_num)
        if res_idx not in alignment:
            continue                                             # removed from the source selection too
        k = alignment[res_idx]
        matching_res_num, matching_res = seq_dst[matching_chain][1][k], seq_dst[matching_chain][0][k]
        if matching_res != res:
            raise Exception("Residues are not matching", chain_id, res_num, res, ":", matching_chain, matching_res_num,
                            matching_res)
        kept.append((chain_id, res_num, res))
        out.append((matching_chain, matching_res_num, matching_res))
    return kept, out


def select_ca(s, selection, shift=None):
    """C-alpha coordinates (n,3) float64 tensor of the selected residues in the selection's order (select_CA followed by
    select_residues_list); residues without a C-alpha are an error -- the two sides of an RMSD must stay aligned."""
    rows = []
    for chain, resnum, _ in selection:
        p = _ca_of(s, chain, resnum)
        if p is None:
            raise Exception("Residue has no CA atom", chain, resnum)
        rows.append(p)
    xyz = torch.from_numpy(np.asarray(rows, dtype=np.float64).reshape(-1, 3))
    return xyz if shift is None else xyz + torch.as_tensor(shift, dtype=torch.double).reshape(1, 3)


def bbox_centre(s):
    """centre of the bounding box of ALL atoms (DockerParser.load_protein's frame, DockerParser.py:70-74)"""
    return 0.5 * (s["xyz"].min(axis=0) + s["xyz"].max(axis=0))


def unbound_interfaces(bound_receptor, bound_ligand, unbound_receptor, unbound_ligand, contact_dist=5.0):
    """DockingBenchmark.get_unbound_interfaces for one assembly: [(urec_sel, ulig_sel, brec_sel, blig_sel)] with the
    four selections aligned entry by entry (receptor with receptor, ligand with ligand)."""
    brec_cont, blig_cont = get_contacts(bound_receptor, bound_ligand, contact_dist)
    rec_match = best_chain_match(bound_receptor, unbound_receptor)
    lig_match = best_chain_match(bound_ligand, unbound_ligand)
    brec_sel, urec_sel = transfer_selection(brec_cont, bound_receptor, unbound_receptor, rec_match)
    blig_sel, ulig_sel = transfer_selection(blig_cont, bound_ligand, unbound_ligand, lig_match)
    return [(urec_sel, ulig_sel, brec_sel, blig_sel)]


def evaluate_target(parser, target_name, bound_receptor_pdb, bound_ligand_pdb, unbound_receptor_pdb, unbound_ligand_pdb,
                    num_conf=None, contact_dist=5.0):
    """EvaluateBenchmark.get_irmsd for one target: the interface RMSD of every conformation of
    ``<decoys_dir>/<target_name>.dat`` (None if the file is missing).  The unbound structures are taken in the docking
    frame -- each centred on its own bounding-box centre, the ligand then placed by the pose."""
    if parser.parse_output(target_name, header_only=False) is None:
        return None
    br, bl = read_structure(bound_receptor_pdb), read_structure(bound_ligand_pdb)
    ur, ul = read_structure(unbound_receptor_pdb), read_structure(unbound_ligand_pdb)
    mobile, static = [], []
    for urec_sel, ulig_sel, brec_sel, blig_sel in unbound_interfaces(br, bl, ur, ul, contact_dist):
        mobile.append((select_ca(ur, urec_sel, -bbox_centre(ur)), select_ca(ul, ulig_sel, -bbox_centre(ul))))
        static.append(torch.cat([select_ca(br, brec_sel), select_ca(bl, blig_sel)], dim=0))
    n = len(parser.target_dict["conformations"])
    n = n if num_conf is None else min(n, int(num_conf))
    return [parser.interface_rmsd(mobile, static, i) for i in range(n)]
