#!/usr/bin/env python3
"""Turns rocprofv3 --pmc passes (tools/prof.sh, tools/prof_traffic.sh, tools/prof_decode.sh) into the two small JSON
files bench.py borrows figures from: profiles/traffic.json (L2 <-> fabric bytes per launch of the dominant kernel) and
profiles/issue.json (VALU instructions and cycles per coded symbol: the issue-rate roofline).  Every entry records the
source hash of the library that was profiled (redux_source_hash()); bench.py refuses an entry whose hash differs from
the library it loaded.  Run on the GPU box, right after the profiling passes, with the same library in place.

Usage: python tools/profile_json.py --encode gpurun_out/prof_<tag> [--decode gpurun_out/prof_dec_<tag>] --workload iid"""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BLOCKS, SYMBOLS = 65536, 65537  # bench.py's shape: 65,536 blocks of 64 KiB + the EOF symbol


def counters(root, sub):
    acc = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            if sub in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, v in acc.items():
        big = [x for x in v if x >= max(v) / 2]  # (a launch on a smaller input, e.g. a warm-up shape, is left out)
        out[k] = sum(big) / len(big)
    return out


def read_bytes(c):
    """Bytes the L2 read over the fabric.  With the request counters by size class at hand they are exact (32/64/128-byte
    requests); otherwise FETCH_SIZE doubled, the guide's gfx950 correction for wide coalesced reads (FETCH_SIZE prices the
    128-byte requests through TCC_BUBBLE, which gfx950 leaves at zero, so they are tallied at 64 bytes)."""
    if all(k in c for k in ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")):
        return int(32 * c["TCC_EA0_RDREQ_32B_sum"] + 64 * c["TCC_EA0_RDREQ_64B_sum"] + 128 * c["TCC_EA0_RDREQ_128B_sum"]), "size classes"
    return int(2 * c["FETCH_SIZE"] * 1024), "2 x FETCH_SIZE"


def merge(path, key, entry, header):
    doc = json.load(open(path)) if os.path.exists(path) else dict(header)
    doc.update({k: v for k, v in header.items() if k not in doc})
    ents = [e for e in doc.get("entries", []) if tuple(e.get(k) for k in key) != tuple(entry.get(k) for k in key)]
    ents.append(entry)
    doc["entries"] = ents
    json.dump(doc, open(path, "w"), indent=1)
    # gpurun only brings gpurun_out/ back from the GPU box: a copy there, to be committed as profiles/<same name>
    back = os.path.join(ROOT, "gpurun_out", "profiles_json")
    os.makedirs(back, exist_ok=True)
    json.dump(doc, open(os.path.join(back, os.path.basename(path)), "w"), indent=1)
    print("wrote", path, {k: entry[k] for k in key})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--encode")
    ap.add_argument("--decode")
    ap.add_argument("--workload", default="iid")
    ap.add_argument("--round", default="4")
    args = ap.parse_args()
    from redux_amd import _lib
    src = _lib.lib().redux_source_hash().decode()
    groups = BLOCKS // 64
    if args.encode:
        c = counters(args.encode, "k_encode_pair")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rb, how = read_bytes(c)
            merge(os.path.join(ROOT, "profiles", "traffic.json"), ("workload", "blocks", "kernel"), {
                "workload": args.workload, "blocks": BLOCKS, "kernel": "k_encode_pair<false, true>", "source_hash": src,
                "FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"],
                "rdreq": {k: c[k] for k in c if k.startswith("TCC_EA0_RDREQ")},
                "read_bytes": rb, "read_bytes_from": how,
                "hbm_bytes_per_launch": int(2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024),
                "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE GRBM_GUI_ACTIVE / --pmc TCC_EA0_RDREQ_* (separate passes) of `python3 bench.py "
                          f"--steps 2 --warmup 1 --no-cpu-baseline --no-decode --workload {args.workload}`, round {args.round}",
            }, {"correction": "gfx950: FETCH_SIZE counts 128-B line requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM). WRITE_SIZE exact."})
        if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
            valu = c["SQ_INSTS_VALU"] / (groups * SYMBOLS)
            cyc = c["GRBM_GUI_ACTIVE"] / 8 / SYMBOLS
            merge(os.path.join(ROOT, "profiles", "issue.json"), ("kernel", "workload"), {
                "kernel": "k_encode_pair<false, true>", "workload": args.workload, "blocks": BLOCKS, "source_hash": src,
                "valu_per_symbol": round(valu, 2), "lds_per_symbol": round(c.get("SQ_INSTS_LDS", 0) / (groups * SYMBOLS), 2),
                "salu_per_symbol": round(c.get("SQ_INSTS_SALU", 0) / (groups * SYMBOLS), 2),
                "cycles_per_symbol": round(cyc, 1), "valu_issue_frac": round(4 * valu / cyc, 3),
                "source": f"SQ_INSTS_* / (1024 groups x 65,537 symbols) of both waves of a group; cycles = GRBM_GUI_ACTIVE / 8 XCDs / 65,537; "
                          f"a wave64 VALU instruction holds its SIMD >= 4 cycles and a SIMD holds one model and one coder wave; round {args.round}",
            }, {"note": "issue-rate roofline of the coder kernels: instructions and cycles per coded symbol (rocprofv3 --pmc, tools/prof.sh / "
                        "tools/prof_decode.sh -> tools/profile_json.py); bench.py copies the entry of the loaded library's source hash"})
    if args.decode:
        c = counters(args.decode, "k_decode_lock")
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rb, how = read_bytes(c)
            merge(os.path.join(ROOT, "profiles", "traffic.json"), ("workload", "blocks", "kernel"), {
                "workload": args.workload, "blocks": BLOCKS, "kernel": "k_decode_lock<true>", "source_hash": src,
                "FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"],
                "rdreq": {k: c[k] for k in c if k.startswith("TCC_EA0_RDREQ")},
                "read_bytes": rb, "read_bytes_from": how,
                # per-lane 16-byte loads are not the guide's calibrated case (wide coalesced reads), so the read side is priced by
                # request size class; tools/ubench/fetchsize.hip (profiles/r04_ubench/fetchsize_gfx950.txt): EVERY read request of
                # this chip's L2 is a 128-byte one, coalesced or per lane, so this equals 2 x FETCH_SIZE here too
                "hbm_bytes_per_launch": int(rb + c["WRITE_SIZE"] * 1024),
                "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE GRBM_GUI_ACTIVE / --pmc TCC_EA0_RDREQ_* (separate passes, tools/prof_decode.sh) of "
                          f"`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --decode --workload {args.workload}`, round {args.round}",
            }, {})
        if "SQ_INSTS_VALU" in c and "SQ_WAVE_CYCLES" in c:
            valu = c["SQ_INSTS_VALU"] / (groups * SYMBOLS)
            cyc = 4 * c["SQ_WAVE_CYCLES"] / groups / SYMBOLS
            merge(os.path.join(ROOT, "profiles", "issue.json"), ("kernel", "workload"), {
                "kernel": "k_decode_lock<true>", "workload": args.workload, "blocks": BLOCKS, "source_hash": src,
                "valu_per_symbol": round(valu, 2), "lds_per_symbol": round(c.get("SQ_INSTS_LDS", 0) / (groups * SYMBOLS), 2),
                "salu_per_symbol": round(c.get("SQ_INSTS_SALU", 0) / (groups * SYMBOLS), 2),
                "cycles_per_symbol": round(cyc, 1), "valu_issue_frac": round(4 * valu / cyc, 3),
                "wait_frac": round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 3),
                "source": f"SQ_INSTS_* / (1024 waves x 65,537 steps); cycles = 4 x SQ_WAVE_CYCLES (quad-cycles) / 1024 / 65,537: one wave per SIMD; round {args.round}",
            }, {})


if __name__ == "__main__":
    main()
