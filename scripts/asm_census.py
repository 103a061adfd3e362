#!/usr/bin/env python3
"""Static instruction census of one kernel of a device assembly file (hipcc -S --cuda-device-only):
per basic block the number of VALU / packed / LDS / VMEM / SALU instructions and where it branches, so that the
instruction mix of each loop (pencil-set loops, staging, combine) can be read off without a GPU.
usage: scripts/asm_census.py file.s kernel_substring [min_instructions]"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    minn = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
    end = next(i for i in range(start + 1, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    blocks, cur = [], {"name": "entry", "ins": []}
    for l in lines[start + 1:end]:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append(cur)
            cur = {"name": m.group(1), "ins": []}
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        cur["ins"].append(s.split(";")[0].strip())
    blocks.append(cur)
    tot = {}
    print("%-12s %6s %6s %6s %6s %6s %6s %6s  branches" % ("block", "n", "valu", "pk", "lds", "vmem", "salu", "wait"))
    for b in blocks:
        c = dict(valu=0, pk=0, lds=0, vmem=0, salu=0, wait=0)
        br = []
        for i in b["ins"]:
            op = i.split()[0]
            if op.startswith("v_pk_"):
                c["pk"] += 1
                c["valu"] += 1
            elif op.startswith("v_"):
                c["valu"] += 1
            elif op.startswith("ds_"):
                c["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                c["vmem"] += 1
            elif op.startswith("s_waitcnt") or op.startswith("s_barrier"):
                c["wait"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
                if "branch" in op:
                    br.append(i.split()[-1])
        for k, v in c.items():
            tot[k] = tot.get(k, 0) + v
        n = len(b["ins"])
        if n >= minn:
            print("%-12s %6d %6d %6d %6d %6d %6d %6d  %s" % (b["name"], n, c["valu"], c["pk"], c["lds"], c["vmem"], c["salu"], c["wait"], " ".join(br)))
    print("total", tot)


if __name__ == "__main__":
    main()
