#!/usr/bin/env python3
"""BASELINE.json configs[2]/[3] as the reference measures them (tests/corpora.rs:28-30, :62, :84):
every corpus file x frequency width {14, 22, 30} (code = width + 2), ratio and MiB/s = bytes / s / 1024 / 1024,
printed in the reference's own line formats -- for three coders side by side:

  cpu      the C restatement of the reference (oracle/, -O2), one thread, the whole file as ONE stream:
           what `cargo test --release` times (the Rust reference itself cannot be built in this image)
  lane     redux_compress / redux_decompress: the literal drop-in for redux::compress -- the whole file as one
           stream, coded by ONE GPU lane (host-pointer ABI, PCIe included).  Stream bytes == cpu's.
  blocks   the accelerated form: the file cut into independent 64 KiB blocks (container payloads), device
           resident, kernel time by HIP events.  A file of <= 64 blocks is ONE wave: latency-bound.

Usage (GPU box):  python tools/corpus_table.py [--corpora calgary,canterbury] > profiles/r02_corpus_table.txt
The oracle is used here as the baseline being timed and as the checker of the GPU streams, never as a fallback.
"""
import argparse
import ctypes as C
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import redux_amd as rx  # noqa: E402
from oracle import cbind as ox  # noqa: E402

BLOCK = 65536
GOLDEN = os.path.join(ROOT, "tests", "golden", "corpora")


def speed(nbytes, seconds):  # tests/corpora.rs:28-30
    return nbytes / seconds / 1024.0 / 1024.0


def timed(fn, reps=1):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return r, best


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota (as bench.py)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def batch(corpora, bits_list):
    """--batch: the files of the named corpora through ONE redux_encode_blocks_v / redux_decode_blocks_v call each (no
    padding: every file is cut into 64 KiB blocks on its own), next to the C restatement on one thread and on all host
    threads (one block per task).  Two GPU figures: the host-pointer call as a caller sees it (staging, PCIe, launch,
    copy back: wall clock), and the launches alone with everything resident in HBM (HIP events around the _v_dev call)."""
    from redux_amd import _lib
    L = _lib.lib()
    names, datas = [], []
    for corpus in corpora:
        for f in sorted(os.listdir(os.path.join(GOLDEN, corpus))):
            names.append(f"{corpus}/{f}")
            datas.append(open(os.path.join(GOLDEN, corpus, f), "rb").read())
    n = sum(len(d) for d in datas)
    nthreads = host_cores()
    print(f"# batch: {len(datas)} files of {'+'.join(corpora)}, {n} B, 64 KiB blocks per file (ragged tails), MiB/s = bytes / s / 2^20")
    rx.compress_blocks_v([b"warm"], BLOCK, (8, 30, 32))
    for bits in bits_list:
        P = (8, bits, bits + 2)
        # CPU: per-block streams, one thread and all threads
        (cpu1, t_c1) = timed(lambda: [ox.compress_blocks(d, BLOCK, P, nthreads=1)[0] for d in datas])
        (cpuN, t_cN) = timed(lambda: [ox.compress_blocks(d, BLOCK, P, nthreads=nthreads)[0] for d in datas])
        want = [s for per in cpu1 for s in per]
        (_, t_d1) = timed(lambda: [ox.decompress(s, P, cap=BLOCK + 16) for s in want])
        # GPU, host-pointer call (best of 3)
        (res, t_ge) = timed(lambda: rx.compress_blocks_v(datas, BLOCK, P), reps=3)
        out, offs, st, first = res
        got = [out[int(offs[i]): int(offs[i + 1])].tobytes() for i in range(len(offs) - 1)]
        assert got == want, "batch streams differ from the restatement"
        (dres, t_gd) = timed(lambda: rx.decompress_blocks_v(out, offs, [len(d) for d in datas], BLOCK, P), reps=3)
        assert [g.tobytes() for g in dres[0]] == datas
        # GPU, device resident: the _v_dev launches alone
        lens = np.array([len(d) for d in datas], dtype=np.uint64)
        doff = np.zeros(len(datas), dtype=np.uint64)
        doff[1:] = np.cumsum((lens + 15) & ~np.uint64(15))[:-1]
        total_in = int(doff[-1] + ((lens[-1] + 15) & ~np.uint64(15)))
        tab = rx.block_table_v(doff, lens, BLOCK)
        ne, nb = len(tab), len(offs) - 1   # table entries (idle lanes included), blocks
        packed = np.zeros(total_in + 16, dtype=np.uint8)
        for o, d in zip(doff, datas):
            packed[int(o): int(o) + len(d)] = np.frombuffer(d, dtype=np.uint8)
        cp = rx.Parameters(*P)._c()
        d_in = torch.from_numpy(packed).cuda()
        d_tab = torch.from_numpy(tab.view(np.uint8).copy()).cuda()
        wsb = L.redux_encode_workspace_bytes(C.byref(cp), ne * BLOCK, BLOCK)
        cap = nb * L.redux_encode_slot_bytes(C.byref(cp), BLOCK)
        ws = torch.empty(wsb + 256, dtype=torch.uint8, device="cuda")
        wsp = ws.data_ptr() + (-ws.data_ptr()) % 256
        d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        d_offs = torch.zeros(nb + 1, dtype=torch.int64, device="cuda")
        d_st = torch.zeros(nb, dtype=torch.int32, device="cuda")
        d_sum = torch.zeros(2, dtype=torch.int32, device="cuda")
        strm = C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def enc_dev():
            d_sum.zero_()
            r = L.redux_encode_blocks_v_dev(C.byref(cp), C.c_void_p(d_in.data_ptr()), total_in, C.c_void_p(d_tab.data_ptr()), ne, nb, BLOCK, 1,
                                            C.c_void_p(d_out.data_ptr()), cap, C.c_void_p(d_offs.data_ptr()), C.c_void_p(d_st.data_ptr()),
                                            C.c_void_p(d_sum.data_ptr()), C.c_void_p(wsp), wsb, strm)
            assert r == 0
        enc_dev()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); enc_dev(); e1.record()
        torch.cuda.synchronize()
        t_ke = e0.elapsed_time(e1) * 1e-3
        assert d_sum.tolist() == [0, 0] and d_offs.cpu().numpy().astype(np.uint64).tolist() == offs.tolist()
        dwsb = L.redux_decode_workspace_bytes(C.byref(cp), ne, BLOCK)
        dws = torch.empty(dwsb + 256, dtype=torch.uint8, device="cuda")
        dwsp = dws.data_ptr() + (-dws.data_ptr()) % 256
        d_back = torch.zeros(total_in + 16, dtype=torch.uint8, device="cuda")
        d_sz = torch.zeros(nb, dtype=torch.int32, device="cuda")

        def dec_dev():
            d_sum.zero_()
            r = L.redux_decode_blocks_v_dev(C.byref(cp), C.c_void_p(d_out.data_ptr()), C.c_void_p(d_offs.data_ptr()), C.c_void_p(d_tab.data_ptr()),
                                            ne, nb, BLOCK, 1, C.c_void_p(d_back.data_ptr()), total_in, C.c_void_p(d_sz.data_ptr()),
                                            C.c_void_p(d_st.data_ptr()), C.c_void_p(d_sum.data_ptr()), C.c_void_p(dwsp), dwsb, strm)
            assert r == 0
        dec_dev()
        torch.cuda.synchronize()
        e0.record(); dec_dev(); e1.record()
        torch.cuda.synchronize()
        t_kd = e0.elapsed_time(e1) * 1e-3
        assert d_sum.tolist() == [0, 0] and torch.equal(d_back[:total_in], d_in[:total_in])
        print(f"  Bits: {bits}  blocks: {nb} in {(ne + 63) // 64} waves  Ratio: {n / len(out):.3f}")
        print(f"    cpu, 1 thread      EncSpeed: {speed(n, t_c1):9.2f} MiB/s  DecSpeed: {speed(n, t_d1):9.2f} MiB/s")
        print(f"    cpu, {nthreads:2d} threads    EncSpeed: {speed(n, t_cN):9.2f} MiB/s")
        print(f"    gpu, host call     EncSpeed: {speed(n, t_ge):9.2f} MiB/s ({t_ge * 1e3:.2f} ms)  DecSpeed: {speed(n, t_gd):9.2f} MiB/s ({t_gd * 1e3:.2f} ms)")
        print(f"    gpu, HBM resident  EncSpeed: {speed(n, t_ke):9.2f} MiB/s ({t_ke * 1e3:.2f} ms)  DecSpeed: {speed(n, t_kd):9.2f} MiB/s ({t_kd * 1e3:.2f} ms)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--corpora", default="artificial,calgary,canterbury,large,misc")
    ap.add_argument("--bits", default="14,22,30")
    ap.add_argument("--batch", action="store_true", help="all files in one _v call (BASELINE.json configs[2] as ONE launch)")
    args = ap.parse_args()
    ox.lib()
    if args.batch:
        batch(args.corpora.split(","), [int(b) for b in args.bits.split(",")])
        return
    rx.compress_blocks(b"warm", BLOCK, (8, 30, 32))
    print("# columns per file: OrigSize, then for cpu / lane / blocks: CompSize, Ratio, EncSpeed, DecSpeed (MiB/s)")
    for corpus in args.corpora.split(","):
        files = sorted(os.listdir(os.path.join(GOLDEN, corpus)))
        for bits in (int(b) for b in args.bits.split(",")):
            P = (8, bits, bits + 2)
            model = rx.AdaptiveTreeModel.new(rx.Parameters.new(*P))
            tot = {k: [0, 0, 0.0, 0.0] for k in ("cpu", "lane", "blocks")}  # dlen, clen, ctime, dtime
            print(f"  Corpus: {corpus}, Model: Tree, Bits: {bits}")
            for f in files:
                data = open(os.path.join(GOLDEN, corpus, f), "rb").read()
                n = len(data)
                # cpu: whole stream, one thread
                (cstream, _), ct = timed(lambda: ox.compress(data, P))
                (cback, _), dt = timed(lambda: ox.decompress(cstream, P, cap=n + 16))
                assert cback == data
                row = {"cpu": (len(cstream), ct, dt)}
                # lane: whole stream on one GPU lane through the drop-in entry points
                o = io.BytesIO()
                (_, ct) = timed(lambda: (o.seek(0), o.truncate(), rx.compress(io.BytesIO(data), o, model))[-1])
                assert o.getvalue() == cstream, f"{f}: drop-in stream differs from the restatement"
                d = io.BytesIO()
                (_, dt) = timed(lambda: (d.seek(0), d.truncate(), rx.decompress(io.BytesIO(cstream), d, model, max_output=n + 16))[-1])
                assert d.getvalue() == data
                row["lane"] = (len(cstream), ct, dt)
                # blocks: 64 KiB blocks, device resident, HIP events around the launches
                d_in = torch.frombuffer(bytearray(data) if n else bytearray(1), dtype=torch.uint8)[:n].cuda()
                enc = rx.DeviceEncoder(P, BLOCK, max(n, 1))
                enc.encode(d_in)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out, offs, status, summary = enc.encode(d_in)
                e1.record()
                torch.cuda.synchronize()
                assert summary.tolist() == [0, 0]
                ct = e0.elapsed_time(e1) * 1e-3
                total = int(offs[-1].item())
                nb = offs.numel() - 1
                dec = rx.DeviceDecoder(P, BLOCK, nb)
                dec.decode(out[:total], offs)
                torch.cuda.synchronize()
                e0.record()
                d_out, d_sizes, d_status, d_sum = dec.decode(out[:total], offs)
                e1.record()
                torch.cuda.synchronize()
                assert d_sum.tolist() == [0, 0] and int(d_sizes.sum().item()) == n
                if n:
                    back = torch.cat([d_out[b * BLOCK: b * BLOCK + int(d_sizes[b].item())] for b in range(nb)])
                    assert torch.equal(back, d_in)
                row["blocks"] = (total + 32 + 4 * nb, ct, e0.elapsed_time(e1) * 1e-3)  # + container header and size table
                cols = []
                for k in ("cpu", "lane", "blocks"):
                    clen, ct, dt = row[k]
                    tot[k][0] += n; tot[k][1] += clen; tot[k][2] += ct; tot[k][3] += dt
                    cols.append(f"{k}: CompSize: {clen} B, Ratio: {n / clen:.3f}, EncSpeed: {speed(n, ct):.2f} MiB/s, DecSpeed: {speed(n, dt):.2f} MiB/s")
                print(f"    File: {f}\n      OrigSize: {n} B | " + " | ".join(cols))
            for k in ("cpu", "lane", "blocks"):
                dlen, clen, ct, dt = tot[k]
                print(f"  Corpus: {corpus}, Model: Tree[{k}], Bits: {bits}, AvgRatio: {dlen / clen:.3f}, "
                      f"AvgEncSpeed: {speed(dlen, ct):.2f} MiB/s, AvgDecSpeed: {speed(dlen, dt):.2f} MiB/s")
    # all corpus files as ONE batch of blocks: what the block path is for (many independent streams at once)
    print("\n# every file of every corpus as one batch of independent 64 KiB blocks (per-file ragged tails kept), bits 30")
    P = (8, 30, 32)
    blobs = []
    for corpus in args.corpora.split(","):
        for f in sorted(os.listdir(os.path.join(GOLDEN, corpus))):
            blobs.append(open(os.path.join(GOLDEN, corpus, f), "rb").read())
    padded = b"".join(b + bytes((-len(b)) % BLOCK) for b in blobs)   # block-aligned file starts (zero padding coded too)
    n = len(padded)
    d_in = torch.frombuffer(bytearray(padded), dtype=torch.uint8).cuda()
    enc = rx.DeviceEncoder(P, BLOCK, n)
    enc.encode(d_in)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out, offs, status, summary = enc.encode(d_in)
    e1.record()
    torch.cuda.synchronize()
    total = int(offs[-1].item())
    dec = rx.DeviceDecoder(P, BLOCK, offs.numel() - 1)
    dec.decode(out[:total], offs)
    torch.cuda.synchronize()
    d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    d0.record()
    d_out = dec.decode(out[:total], offs)[0]
    d1.record()
    torch.cuda.synchronize()
    assert torch.equal(d_out, d_in)
    print(f"  {len(blobs)} files, {n} B in {offs.numel() - 1} blocks: Ratio: {n / total:.3f}, EncSpeed: {speed(n, e0.elapsed_time(e1) * 1e-3):.2f} MiB/s "
          f"({e0.elapsed_time(e1):.2f} ms), DecSpeed: {speed(n, d0.elapsed_time(d1) * 1e-3):.2f} MiB/s ({d0.elapsed_time(d1):.2f} ms)")


if __name__ == "__main__":
    main()
