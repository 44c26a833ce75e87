#!/usr/bin/env python3
"""Headline benchmark: encode MB/s on BASELINE.json configs[1].

  python bench.py --gpus N --steps K --warmup W          (any N: for N > 1 and no RANK in the
                                                          environment it starts the N rank processes itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one batch: every rank codes its own 65,536 blocks of
64 KiB of iid bytes (generated in HBM, seed 0x5EED0001, rank r owns stream bytes
[r*4GiB, (r+1)*4GiB)) into a dense stream + offsets table, input and output resident in HBM.
Blocks are independent, so ranks share nothing: no collective on the data path ("weak"
scaling: per-GPU work is fixed).  value = input bytes of all ranks / max-over-ranks time.

Extra objects on the JSON line:
  roofline      dominant kernel (k_encode): algorithmic bytes = sum over blocks of
                (bytes read + bytes written), taken from the offsets table, divided by the
                kernel's mean duration measured with HIP events on the launch stream inside
                the timed steps; peak = 8.0e12 B/s (MI355X HBM3E).  `traffic` (PMC-measured HBM
                bytes per launch) and `issue` (VALU instructions and cycles per coded symbol: the
                kernel is bound by instruction issue, not by HBM) are borrowed from profiles/
                traffic.json / issue.json -- only when the profile's source hash is the loaded
                library's (roofline.source_hash); otherwise null and the reason.
  cpu_baseline  the CPU oracle (C restatement of the reference, kind "port") on a bounded
                prefix of the same workload, on this box's host cores, rank 0 at N=1 only.
  decode        (rank 0, after the timed steps, not part of `value`) one timed pass of the
                decoder over the dense output, checked equal to the input on the device; `traffic`
                borrowed from profiles/traffic.json under the same source-hash gate.
  small_launch  (rank 0 at N=1, not part of `value`) encode and decode of the first 62 blocks
                alone -- a launch where one block's serial chain is the whole time; its streams
                are checked equal to the same blocks' streams of the full launch.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BLOCK = 65536
PARAMS = (8, 30, 32)  # the reference CLI's fixed Parameters::new(8, 30, 32) (src/main.rs:108)
SEED = 0x5EED0001
HBM_PEAK = 8.0e12


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def launch_command(n, argv, port=None):
    """The rank launcher `python bench.py --gpus N` turns into (one process per GPU over RCCL)."""
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv, run=None):
    """Starts the N rank processes as children, lets rank 0's JSON line through on stdout and
    returns the launcher's exit code.  `run` is injectable for the CPU test."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: the only kind the host driver supports
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = launch_command(n, argv)
    return (run or subprocess.call)(cmd, env=env)


def file_workload(args, rx, torch, dist, rank, world, dev):
    """BASELINE.json configs[3] (SURVEY 8(d) cfg 4: "latency-bound; report time, not roofline"): one file, cut into 64 KiB
    blocks, contiguous block ranges scattered from rank 0 over RCCL (exact byte counts, grouped send / recv), coded by every
    rank with the HIP kernels, compressed ranges gathered back to rank 0; then the same way back through the decoder.  A step
    is one such round trip; the line reports the mean time of each phase over the timed steps (max over ranks of each step's
    wall time for `value`), the streams are checked against tests/golden/blocks.json and the decoded file against the input."""
    import hashlib
    from redux_amd import dist as rd
    path = args.workload[5:] if args.workload.startswith("file:") else ""
    path = path or os.path.join(ROOT, "tests", "golden", "corpora", "large", "bible.txt")
    if world == 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29641")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=0, world_size=1)
        else:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev))
    data, raw = None, b""
    if rank == 0:
        raw = open(path, "rb").read()
        data = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev) if raw else torch.empty(0, dtype=torch.uint8, device=dev)
    enc_local, dec_local = rd.hip_encode_local(PARAMS), rd.hip_decode_local(PARAMS)
    phases = {k: 0.0 for k in ("scatter_ms", "encode_ms", "gather_ms", "decode_scatter_ms", "decode_ms", "decode_gather_ms")}
    enc_wall = dec_wall = 0.0
    dense = offs = back = None
    for it in range(args.warmup + args.steps):
        timed = it >= args.warmup
        dist.barrier()
        torch.cuda.synchronize()
        m1, m2 = [], []
        t0 = time.perf_counter()
        dense, offs = rd.encode_file_sharded(data, BLOCK, enc_local, dev, marks=m1)
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        back = rd.decode_file_sharded(dense, offs, BLOCK, dec_local, dev, marks=m2)
        dist.barrier()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if timed:
            enc_wall += t1 - t0
            dec_wall += t2 - t1
            for k, (_, dt) in zip(("scatter_ms", "encode_ms", "gather_ms"), m1):
                phases[k] += dt * 1e3
            for k, (_, dt) in zip(("decode_scatter_ms", "decode_ms", "decode_gather_ms"), m2):
                phases[k] += dt * 1e3
    # max over ranks of the walls and of every phase (a phase lasts as long as its slowest rank)
    t = torch.tensor([enc_wall, dec_wall] + [phases[k] for k in sorted(phases)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    vals = t.tolist()
    enc_wall, dec_wall = vals[0], vals[1]
    phases = {k: round(v / args.steps, 3) for k, v in zip(sorted(phases), vals[2:])}
    if rank == 0:
        n = len(raw)
        nblocks = max(1, (n + BLOCK - 1) // BLOCK)
        assert back.cpu().numpy().tobytes() == raw, "decode(encode(file)) != file"
        o = offs.cpu().numpy()
        d = dense.cpu().numpy()
        golden = None
        gpath = os.path.join(ROOT, "tests", "golden", "blocks.json")
        key = os.path.relpath(os.path.abspath(path), os.path.join(ROOT, "tests", "golden", "corpora"))
        if os.path.exists(gpath) and not key.startswith(".."):
            gold = json.load(open(gpath)).get(key, {}).get("%d_%d_%d" % PARAMS)
            if gold:
                h64 = lambda b: hashlib.blake2b(b, digest_size=8).hexdigest()  # noqa: E731  (tests/golden/make_fixtures.py)
                assert [int(x) for x in (o[1:] - o[:-1])] == gold["block_sizes"], "block sizes differ from tests/golden/blocks.json"
                assert [h64(d[int(o[i]): int(o[i + 1])].tobytes()) for i in range(nblocks)] == gold["block_hashes"], \
                    "streams differ from tests/golden/blocks.json"
                golden = "every block's size and blake2b-64 equal tests/golden/blocks.json"
        per = (nblocks + world - 1) // world
        line = {
            "metric": "encode MB/s (whole node), per-block bitstream bit-exact",
            "value": round(n * args.steps / enc_wall / 1e6, 2) if enc_wall else None,
            "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(enc_wall / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32", "data": "file",
            "config": {"workload": f"file:{os.path.relpath(path, ROOT)} cut into 64 KiB blocks, contiguous block ranges scattered from rank 0, coded, "
                                   "gathered (BASELINE.json configs[3]; SURVEY 8(d) cfg 4: time, not roofline)",
                       "bytes": n, "block_size": BLOCK, "blocks": nblocks, "blocks_per_rank": per, "parameters": list(PARAMS),
                       "parallelism": f"{world} rank(s); exchange: sizes first, then exact byte counts as grouped send/recv ("
                                      + ("gloo, staged through host memory" if args.rehearse_on_one_gpu else "RCCL") + ")",
                       "compressed_over_input": round(int(o[-1]) / n, 5) if n else None},
            "phases": phases,
            "decode_ms_per_step": round(dec_wall / args.steps * 1e3, 3),
            "roundtrip_equal": True, "golden": golden,
            "roofline": None, "cpu_baseline": None,
            "note": ("latency-bound: one block's serial chain (~5 ms encode, ~21 ms decode per 64 KiB block) is the floor of a launch whatever "
                     "the number of ranks; " + ("rehearsal of the multi-rank code path on one GPU, not a scaling measurement"
                                                if args.rehearse_on_one_gpu else ("unmeasured at N > 1 on this pool's one-GPU boxes" if world == 1 else ""))),
        }
        print(json.dumps(line))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=10)  # the chip's clocks settle only after ~0.2 s of load
    ap.add_argument("--blocks", type=int, default=65536, help="blocks per GPU (default: the config's 65,536)")
    ap.add_argument("--workload", default="iid",
                    help="iid (default, BASELINE configs[1]) | zipf (configs[4] shape) | file[:<path>] -- BASELINE configs[3]: the root scatters "
                         "a file's 64 KiB blocks over the ranks, every rank codes its range, the root gathers and decodes back the same way "
                         "(default file: tests/golden/corpora/large/bible.txt); prints its own line (time per phase, not a roofline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--decode", action="store_true", help="(default) also time the decode kernel after the timed encode steps")
    ap.add_argument("--no-decode", action="store_true", help="skip the decode measurement / round-trip check")
    ap.add_argument("--cpu-sample-blocks", type=int, default=4096)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (default: usable cores)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="all ranks share cuda:0 and rendezvous over gloo: exercises the multi-rank code path on a one-GPU box "
                         "(the line says so; not a scaling measurement)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # Plain `python bench.py --gpus N`: this process becomes the launcher.  It has made no HIP
        # call (torch is not even imported yet) and starts N fresh rank processes; it never
        # re-execs itself.
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1 and args.rehearse_on_one_gpu:
        dist.init_process_group("gloo")
    elif world > 1:
        dist.init_process_group("nccl", device_id=torch.device(dev))

    import redux_amd as rx

    if args.workload == "file" or args.workload.startswith("file:"):
        return file_workload(args, rx, torch, dist, rank, world, dev)
    if args.workload not in ("iid", "zipf"):
        print(f"bench.py: unknown --workload {args.workload}", file=sys.stderr)
        sys.exit(2)

    nblocks = args.blocks
    n = nblocks * BLOCK
    seed = SEED if args.workload == "iid" else 0x5EED0005
    gen = rx.gen_iid if args.workload == "iid" else rx.gen_zipf
    d_in = gen(n, seed, rank * n, device=dev)
    enc = rx.DeviceEncoder(PARAMS, BLOCK, n, device=dev)
    torch.cuda.synchronize()

    def step(ev=None):
        if ev is not None:
            ev[0].record()
        enc.encode_slots(d_in)      # k_fill_rc (a few us) + k_encode, the dominant kernel
        if ev is not None:
            ev[1].record()
        enc.compact(n)              # k_scan_sizes + k_compact -> dense output + offsets

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(events[i])
    barrier()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # correctness gate on this rank: no block reported an error, output decodes on device
    assert enc.summary.tolist() == [0, 0], enc.summary.tolist()
    out_bytes = int(enc.offsets[nblocks].item())
    kern_ms = sum(a.elapsed_time(b) for a, b in events) / len(events)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_in = n * world
    value = total_in / elapsed * args.steps / 1e6
    algo_bytes = n + out_bytes
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9
    import ctypes as C
    from redux_amd import _lib
    cp = _lib.Params(*PARAMS)
    # the kernel the library's dispatch picked for exactly these arguments (not an assumption)
    kname = _lib.lib().redux_encode_kernel_name(C.byref(cp), C.c_void_p(d_in.data_ptr()), n, BLOCK).decode()
    dname = _lib.lib().redux_decode_kernel_name(C.byref(cp), None, BLOCK).decode()
    # `traffic` and `issue` are not measured by this run: they are PMC figures of separate rocprofv3 --pmc passes of this
    # same command (tools/prof.sh -> tools/profile_json.py), kept per workload in profiles/traffic.json and
    # profiles/issue.json together with the source hash of the library that was profiled.  A figure is only borrowed
    # when that hash equals the loaded library's: after any change to the kernels it reads null + the reason.
    lib_hash = _lib.lib().redux_source_hash().decode()

    def borrowed(fname, want):
        path = os.path.join(ROOT, "profiles", fname)
        if not os.path.exists(path):
            return None, f"no profiles/{fname}"
        try:
            ents = json.load(open(path)).get("entries", [])
        except Exception as e:  # noqa: BLE001
            return None, f"profiles/{fname} unreadable: {e}"
        for ent in ents:
            if all(ent.get(k) == v for k, v in want.items()):
                if ent.get("source_hash") != lib_hash:
                    return None, (f"profiles/{fname} was measured on source hash {ent.get('source_hash')}, the loaded library is "
                                  f"{lib_hash}: re-run tools/prof.sh + tools/profile_json.py")
                return ent, ent.get("source", f"profiles/{fname}")
        return None, f"profiles/{fname} has no entry for {want}"

    tent, traffic_src = borrowed("traffic.json", {"workload": args.workload, "blocks": nblocks, "kernel": kname.split(" (")[0]})
    traffic = tent.get("hbm_bytes_per_launch") if tent else None
    ient, issue_src = borrowed("issue.json", {"workload": args.workload, "blocks": nblocks, "kernel": kname.split(" (")[0]})
    issue = ({k: ient[k] for k in ("valu_per_symbol", "lds_per_symbol", "salu_per_symbol", "cycles_per_symbol", "valu_issue_frac")}
             if ient else None)
    if issue is not None:
        issue["source"] = issue_src
    else:
        issue = {"valu_per_symbol": None, "cycles_per_symbol": None, "valu_issue_frac": None, "source": issue_src}
    line = {
        "metric": "encode MB/s (whole node), per-block bitstream bit-exact",
        "value": round(value, 1),
        "unit": "MB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {
            "workload": (f"{nblocks} independent 64 KiB iid-byte blocks per GPU (BASELINE.json configs[1])"
                         if args.workload == "iid" else
                         f"{nblocks} independent 64 KiB Zipf(1.2) blocks per GPU (BASELINE.json configs[4] shape)"),
            "block_size": BLOCK,
            "blocks_per_gpu": nblocks,
            "parameters": list(PARAMS),
            "parallelism": f"blocks sharded over {world} GPU(s), no data-path collective",
            "compressed_over_input": round(out_bytes / n, 5),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": kname,
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK / 1e9,
            "unit": "GB/s",
            "frac": round(achieved * 1e9 / HBM_PEAK, 5),
            "traffic": traffic,
            "traffic_source": traffic_src,
            "issue": issue,
            "source_hash": lib_hash,
            "algorithmic_bytes_per_launch": algo_bytes,
            "kernel_ms": round(kern_ms, 3),
        },
    }

    if args.rehearse_on_one_gpu:
        line["rehearsal"] = f"{world} ranks shared cuda:0 over gloo: a code-path check, not a scaling measurement"
    if not args.no_decode:
        dec = rx.DeviceDecoder(PARAMS, BLOCK, nblocks, device=dev)
        offs_t = enc.offsets[: nblocks + 1]
        dense = enc.out[:out_bytes]
        dec.decode(dense, offs_t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        d_out, d_sizes, d_status, d_sum = dec.decode(dense, offs_t)
        e1.record()
        torch.cuda.synchronize()
        assert d_sum.tolist() == [0, 0] and torch.equal(d_out, d_in), "decode(encode(x)) != x"
        dms = e0.elapsed_time(e1)
        dent, dsrc = borrowed("issue.json", {"workload": args.workload, "blocks": nblocks, "kernel": dname.split(" (")[0]})
        dissue = ({k: dent[k] for k in ("valu_per_symbol", "lds_per_symbol", "salu_per_symbol", "cycles_per_symbol", "valu_issue_frac")}
                  if dent else {"valu_per_symbol": None, "cycles_per_symbol": None, "valu_issue_frac": None})
        dissue["source"] = dsrc
        dtent, dtraffic_src = borrowed("traffic.json", {"workload": args.workload, "blocks": nblocks, "kernel": dname.split(" (")[0]})
        line["decode"] = {"kernel": dname, "ms": round(dms, 3),
                          "traffic": dtent.get("hbm_bytes_per_launch") if dtent else None, "traffic_source": dtraffic_src,
                          "MBps": round(n / (dms * 1e-3) / 1e6, 1),
                          "algorithmic_GBps": round(algo_bytes / (dms * 1e-3) / 1e9, 2),
                          "frac": round(algo_bytes / (dms * 1e-3) / HBM_PEAK, 5), "issue": dissue, "roundtrip_equal": True}

    if world == 1 and not args.no_decode:
        # Small launch (rank 0, N = 1, not part of `value`): 62 blocks of the same workload -- a 4 MiB file at this block size --
        # where the serial chain of ONE block is the whole time.  The encoder is the small-grid path (redux_coop.hpp).
        sb = min(62, nblocks)
        sn = sb * BLOCK
        senc = rx.DeviceEncoder(PARAMS, BLOCK, sn, device=dev)
        sdec = rx.DeviceDecoder(PARAMS, BLOCK, sb, device=dev)
        s_in = d_in[:sn]
        for _ in range(2):
            s_out, s_offs, s_st, s_sum = senc.encode(s_in)
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        stotal = int(s_offs[sb].item())
        sdec.decode(s_out[:stotal], s_offs)
        torch.cuda.synchronize()
        e0.record()
        s_out, s_offs, s_st, s_sum = senc.encode(s_in)
        e1.record()
        sd_out, _, _, sd_sum = sdec.decode(s_out[:stotal], s_offs)
        e2.record()
        torch.cuda.synchronize()
        assert s_sum.tolist() == [0, 0] and sd_sum.tolist() == [0, 0] and torch.equal(sd_out[:sn], s_in)
        assert torch.equal(s_out[:stotal], enc.out[:stotal]), "small-grid kernels and the full-grid kernel differ on the same blocks"
        line["small_launch"] = {"blocks": sb, "encode_ms": round(e0.elapsed_time(e1), 3), "decode_ms": round(e1.elapsed_time(e2), 3),
                                "encode_kernel": _lib.lib().redux_encode_kernel_name(C.byref(cp), C.c_void_p(s_in.data_ptr()), sn, BLOCK).decode().split(" (")[0],
                                "decode_kernel": _lib.lib().redux_decode_kernel_name_n(C.byref(cp), None, BLOCK, sb).decode().split(" (")[0],
                                "equal_to_full_grid_streams": True}

    if world == 1 and not args.no_cpu_baseline:
        from oracle import cbind as ox
        sample_blocks = min(args.cpu_sample_blocks, nblocks)
        host = d_in[: sample_blocks * BLOCK].cpu().numpy()
        cores = args.cpu_threads or host_cores()
        ox.lib()
        t1 = time.perf_counter()
        o1, sizes1, status1, slot1 = ox.compress_blocks_raw(host[: 64 * BLOCK], BLOCK, PARAMS, ox.TREE, nthreads=1)
        single = 64 * BLOCK / (time.perf_counter() - t1) / 1e6
        t1 = time.perf_counter()
        o, sizes, status, slot = ox.compress_blocks_raw(host, BLOCK, PARAMS, ox.TREE, nthreads=cores)
        dt = time.perf_counter() - t1
        assert not status.any()
        # parity spot check against what the GPU produced for the same blocks
        offs = enc.offsets[: sample_blocks + 1].cpu().numpy()
        import numpy as np
        assert (np.diff(offs) == sizes).all(), "GPU block sizes differ from the CPU oracle"
        for b in (0, sample_blocks // 2, sample_blocks - 1):
            g = enc.out[int(offs[b]): int(offs[b + 1])].cpu().numpy()
            assert (g == o[b * slot: b * slot + int(sizes[b])]).all(), f"block {b} differs from the CPU oracle"
        # the LITERAL drop-in (redux::compress of one whole stream = one GPU lane, host-pointer ABI), for scale: a
        # serial adaptive stream has no GPU parallelism, the blocked API above is the accelerated path
        import io
        one = host[: 1 << 20].tobytes()
        sink = io.BytesIO()
        rx.compress(io.BytesIO(one), io.BytesIO(), rx.AdaptiveTreeModel.new(rx.Parameters.new(*PARAMS)))  # (sizes the host context)
        t1 = time.perf_counter()
        rx.compress(io.BytesIO(one), sink, rx.AdaptiveTreeModel.new(rx.Parameters.new(*PARAMS)))
        one_lane = len(one) / (time.perf_counter() - t1) / 1e6
        assert sink.getvalue() == ox.compress(one, PARAMS)[0], "whole-stream drop-in differs from the CPU oracle"
        line["cpu_baseline"] = {
            "value": round(sample_blocks * BLOCK / dt / 1e6, 1),
            "unit": "MB/s",
            "cores": cores,
            "kind": "port",
            "sample": f"first {sample_blocks} blocks ({sample_blocks * BLOCK >> 20} MiB) of the same stream, C restatement "
                      f"of the reference (-O2), {cores} threads, one block per task; single thread: {single:.1f} MB/s; "
                      "sizes of all sampled blocks and bytes of 3 blocks compared with the GPU output",
            "one_lane_drop_in_MBps": round(one_lane, 2),
            "one_lane_note": "redux_compress (the literal redux::compress drop-in: the first 1 MiB as ONE stream -- its model computed by 64 lanes "
                             "from prefix counts, its interval chain by one, redux_coop.hpp -- PCIe included, bytes equal to the CPU oracle's): "
                             "for scale against the single-thread figure above",
        }
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
