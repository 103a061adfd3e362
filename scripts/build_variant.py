#!/usr/bin/env python3
"""Build a variant library for A/B runs: scripts/build_variant.py <name> [--corr "<flags>"] [--k2 "<flags>"]
-> build_variants/libdlpd_<name>.so (default objects are reused for the translation units without extra flags;
select it on the GPU box with DLPD_LIB_PATH)."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--corr", default="")
    ap.add_argument("--k2", default="")
    ap.add_argument("--k3r", default="")
    ap.add_argument("--k2q", default="")
    ap.add_argument("--k1r", default="")
    ap.add_argument("--topk", default="")
    ap.add_argument("--conv", default="")
    ap.add_argument("--corr-src", default=None, help="alternative source file for dlpd_corr.hip")
    args = ap.parse_args()
    entry.build()
    extra = {"dlpd_corr.hip": args.corr.split(), "dlpd_k2.hip": args.k2.split(), "dlpd_k3r.hip": args.k3r.split(), "dlpd_k2q.hip": args.k2q.split(), "dlpd_k1r.hip": args.k1r.split(), "dlpd_topk.hip": args.topk.split(),
             "dlpd_conv.hip": args.conv.split()}
    out_dir = os.path.join(ROOT, "build_variants")
    os.makedirs(out_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src, flags in entry.SOURCES:
        obj = os.path.join(entry.CSRC, src.replace(".hip", ".o"))
        alt = args.corr_src if src == "dlpd_corr.hip" else None
        if extra.get(src) or alt:
            obj = os.path.join("/tmp", "dlpdv_%s_%s.o" % (args.name, src.replace(".hip", "")))
            cmd = [hipcc] + entry.COMMON_FLAGS + ["-I", entry.CSRC, "-I", os.path.join(ROOT, "include")] + flags + \
                extra[src] + ["-Rpass-analysis=kernel-resource-usage", "-c", alt or os.path.join(entry.CSRC, src), "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                sys.exit(r.stderr[-3000:])
        objs.append(obj)
    lib = os.path.join(out_dir, "libdlpd_%s.so" % args.name)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    main()
