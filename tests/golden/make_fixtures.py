#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference's data files and the CPU oracle.

What is stored and why:
  corpora/<corpus>/<file>   the reference's own test inputs (/root/reference/resources, the
                            data files tests/corpora.rs iterates over).  DATA, copied
                            byte-for-byte: the GPU box has no /root/reference, and config 3/4
                            of BASELINE.json need these exact inputs there.
  manifest.json             size + md5 of every corpus file.
  kat_streams.json          small full compressed streams (hex): the four hand-traced vectors
                            of SURVEY.md 8c, the src/lib.rs:23-39 doc-test input, and a few
                            short corpus prefixes, at the three tested widths.
  blocks.json               for every corpus file x (8,14,16),(8,22,24),(8,30,32): per-64-KiB
                            -block (stream size, blake2b-64 of the stream) and the same for the
                            whole-file (unblocked) stream.

The expected outputs come from oracle/libredux_oracle.so (the C restatement), NOT from the
Rust reference, which cannot be built in this image (no rustc/cargo).  They are regression
vectors for that restatement; the hand-traced vectors are the independent pins.

Run from the repo root:  python tests/golden/make_fixtures.py
"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import cbind as ox  # noqa: E402

REF = "/root/reference/resources"
CORPORA = ["artificial", "calgary", "canterbury", "large", "misc"]
WIDTHS = [(8, 14, 16), (8, 22, 24), (8, 30, 32)]
BLOCK = 65536


def h64(b):
    return hashlib.blake2b(b, digest_size=8).hexdigest()


def main():
    dst = os.path.join(HERE, "corpora")
    if os.path.isdir(REF):
        for c in CORPORA:
            os.makedirs(os.path.join(dst, c), exist_ok=True)
            for f in sorted(os.listdir(os.path.join(REF, c))):
                shutil.copyfile(os.path.join(REF, c, f), os.path.join(dst, c, f))
                os.chmod(os.path.join(dst, c, f), 0o644)

    manifest, blocks = {}, {}
    for c in CORPORA:
        for f in sorted(os.listdir(os.path.join(dst, c))):
            key = f"{c}/{f}"
            data = open(os.path.join(dst, c, f), "rb").read()
            manifest[key] = {"size": len(data), "md5": hashlib.md5(data).hexdigest()}
            entry = {}
            for w in WIDTHS:
                streams, status = ox.compress_blocks(data, BLOCK, w, ox.TREE, nthreads=8)
                assert not status.any()
                whole, counts = ox.compress(data, w, ox.TREE)
                assert counts == (len(data), len(whole))
                entry["%d_%d_%d" % w] = {
                    "block_sizes": [len(s) for s in streams],
                    "block_hashes": [h64(s) for s in streams],
                    "whole_size": len(whole),
                    "whole_hash": h64(whole),
                }
            blocks[key] = entry
            print(key, len(data), entry["8_30_32"]["whole_size"])

    kats = []
    small = {
        "empty": b"",
        "a": b"a",
        "doctest_redux": bytes([0x72, 0x65, 0x64, 0x75, 0x78]),
        "alice29_first_300": open(os.path.join(dst, "canterbury", "alice29.txt"), "rb").read()[:300],
        "geo_first_300": open(os.path.join(dst, "calgary", "geo"), "rb").read()[:300],
        "aaa_first_2000": open(os.path.join(dst, "artificial", "aaa.txt"), "rb").read()[:2000],
        "all_bytes_twice": bytes(range(256)) * 2,
    }
    for name, data in small.items():
        for w in WIDTHS:
            s, counts = ox.compress(data, w, ox.TREE)
            kats.append({"name": name, "params": list(w), "input_hex": data.hex(), "stream_hex": s.hex(),
                         "counts": list(counts)})

    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)
    json.dump(blocks, open(os.path.join(HERE, "blocks.json"), "w"), indent=0, sort_keys=True)
    json.dump(kats, open(os.path.join(HERE, "kat_streams.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
