"""Protein-sized synthetic PDB files for the atom-level tests (test infrastructure).

The benchmark structures cannot be shipped (no network, DockingBenchmark is not in the reference
tree), so the PDB-level path is exercised on generated files that have what real entries have and
the 9-40-atom toys of round 1 had not: thousands of heavy atoms in a compact globule, several
chains with TER records, HETATM ligands / waters, alternate locations, hydrogens, ANISOU records, a
second MODEL, insertion codes, negative residue numbers and an END record."""
import numpy as np

RES = {"GLY": ["N", "CA", "C", "O"], "ALA": ["N", "CA", "C", "O", "CB"],
       "SER": ["N", "CA", "C", "O", "CB", "OG"], "CYS": ["N", "CA", "C", "O", "CB", "SG"],
       "VAL": ["N", "CA", "C", "O", "CB", "CG1", "CG2"], "LEU": ["N", "CA", "C", "O", "CB", "CG", "CD1", "CD2"],
       "THR": ["N", "CA", "C", "O", "CB", "OG1", "CG2"], "MET": ["N", "CA", "C", "O", "CB", "CG", "SD", "CE"],
       "LYS": ["N", "CA", "C", "O", "CB", "CG", "CD", "CE", "NZ"],
       "ASP": ["N", "CA", "C", "O", "CB", "CG", "OD1", "OD2"],
       "GLU": ["N", "CA", "C", "O", "CB", "CG", "CD", "OE1", "OE2"],
       "GLN": ["N", "CA", "C", "O", "CB", "CG", "CD", "OE1", "NE2"],
       "ARG": ["N", "CA", "C", "O", "CB", "CG", "CD", "NE", "CZ", "NH1", "NH2"],
       "PHE": ["N", "CA", "C", "O", "CB", "CG", "CD1", "CD2", "CE1", "CE2", "CZ"],
       "TYR": ["N", "CA", "C", "O", "CB", "CG", "CD1", "CD2", "CE1", "CE2", "CZ", "OH"],
       "TRP": ["N", "CA", "C", "O", "CB", "CG", "CD1", "CD2", "NE1", "CE2", "CE3", "CZ2", "CZ3", "CH2"],
       "HIS": ["N", "CA", "C", "O", "CB", "CG", "ND1", "CD2", "CE1", "NE2"],
       "ASN": ["N", "CA", "C", "O", "CB", "CG", "OD1", "ND2"]}


def _atom_line(rec, serial, name, alt, resn, chain, resi, icode, xyz, elem):
    name4 = (" " + name) if len(name) < 4 else name
    return "%-6s%5d %-4s%1s%3s %1s%4d%1s   %8.3f%8.3f%8.3f  1.00 20.00          %2s" % (
        rec, serial % 100000, name4, alt, resn, chain, resi, icode, xyz[0], xyz[1], xyz[2], elem)


def write_protein_like_pdb(path, nres, seed, nchains=2, offset=(0.0, 0.0, 0.0), extras=True):
    """A compact random-coil globule of ``nres`` residues in ``nchains`` chains: CA trace = 3.8 A random
    walk reflected into a sphere sized for protein density, side-chain atoms 1.5 A steps off the CA.
    Returns the number of heavy ATOM-record atoms the reader must keep (first model, altloc ' '/'A')."""
    rng = np.random.RandomState(seed)
    names = list(RES)
    radius = 1.25 * (nres * 135.0 * 3.0 / (4.0 * np.pi)) ** (1.0 / 3.0)
    off = np.asarray(offset, dtype=np.float64)
    lines, serial, kept = ["HEADER    SYNTHETIC GLOBULE", "REMARK   2 RESOLUTION. NOT APPLICABLE."], 1, 0
    if extras:
        lines.append("MODEL        1")
    pos = rng.normal(size=3) * radius * 0.2
    per_chain = [nres // nchains + (1 if c < nres % nchains else 0) for c in range(nchains)]
    for c, nr in enumerate(per_chain):
        chain = "ABCDEFGH"[c % 8]
        resi = -3 if (extras and c == 0) else 1              # negative residue numbers occur in real entries
        for r in range(nr):
            step = rng.normal(size=3)
            step *= 3.8 / np.linalg.norm(step)
            if np.linalg.norm(pos + step) > radius:
                step = -step
            pos = pos + step
            rn = names[rng.randint(len(names))]
            p = pos.copy()
            for ai, an in enumerate(RES[rn]):
                if ai > 0:
                    d = rng.normal(size=3)
                    p = (pos if ai < 4 else p) + d * (1.5 / np.linalg.norm(d))
                alt = " "
                if extras and rng.rand() < 0.01:             # alternate locations: 'A' is kept, 'B' dropped
                    alt = "A"
                lines.append(_atom_line("ATOM", serial, an, alt, rn, chain, resi, " ", p + off, an[0]))
                serial += 1
                kept += 1
                if alt == "A":
                    lines.append(_atom_line("ATOM", serial, an, "B", rn, chain, resi, " ", p + off + 0.4, an[0]))
                    serial += 1
                if extras and rng.rand() < 0.01:
                    lines.append("ANISOU%5d %-4s %3s %1s%4d     1000   1000   1000      0      0      0" % (
                        (serial - 1) % 100000, an, rn, chain, resi))
            if extras and r % 17 == 0:                       # a hydrogen the typing must skip (it is read)
                lines.append(_atom_line("ATOM", serial, "H", " ", rn, chain, resi, " ", pos + off + 0.9, "H"))
                serial += 1
            if extras and r == nr // 2:                      # an inserted residue (insertion code)
                lines.append(_atom_line("ATOM", serial, "CA", " ", "GLY", chain, resi, "A", pos + off + 1.9, "C"))
                serial += 1
                kept += 1
            resi += 1
        lines.append("TER   %5d      %3s %1s%4d" % (serial % 100000, rn, chain, resi - 1))
        serial += 1
    if extras:
        for w in range(25):                                  # waters and a hetero group: not ATOM records
            q = rng.normal(size=3) * radius * 0.6 + off
            lines.append(_atom_line("HETATM", serial, "O", " ", "HOH", "W", 900 + w, " ", q, "O"))
            serial += 1
        lines.append(_atom_line("HETATM", serial, "ZN", " ", " ZN", "A", 800, " ", off, "ZN"))
        lines += ["ENDMDL", "MODEL        2",
                  _atom_line("ATOM", 1, "CA", " ", "ALA", "A", 1, " ", off + 50.0, "C"), "ENDMDL"]
    lines.append("END")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return kept
