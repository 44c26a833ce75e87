#!/usr/bin/env python3
"""A/B build (CPU; measured on the GPU box with tools/zipf_variants.sh + tools/traffic_ab.sh): the pair kernel's coder wave
with LINEAR slots and a 16-byte register cell per lane -- the cheapest of the store stagings DESIGN.md costs for the Zipf
write traffic (one build-and-measure, then close the item).

What the variant changes in a COPY of redux_amd/csrc (the product sources are not touched):
  * k_encode_pair's slots are linear (a lane's dwords are contiguous; mode word = 2: k_compact, byte-reversed dwords) instead
    of row-major group areas;
  * encode_symbol_spec keeps a lane's last three completed dwords in registers (EncState::q0..q2, shifted under the store's
    exec mask) and stores 16 bytes when the fourth completes -- the hot path's instruction stream and store traffic are what
    a shipped version would have.
What it does NOT do (so its output is not valid and it is a TIMING / TRAFFIC build only: bench.py --no-decode
--no-cpu-baseline): flush a block's last partial cell, keep the careful redo path and the checked chunks consistent with the
staged cell.  If it measured faster on Zipf the full version would be worth building; see profiles/r04_zipf_staging.txt.

usage: python tools/ab/zipf_stage16.py [--no-stage]  ->  variants/zipf_stage16.so (variants/zipf_linear4.so)"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from redux_amd import build  # noqa: E402


def sub(s, old, new):
    assert old in s, old[:60]
    return s.replace(old, new, 1)


def main():
    tmp = tempfile.mkdtemp(prefix="zv")
    dst = os.path.join(tmp, "redux_amd", "csrc")
    shutil.copytree(os.path.join(ROOT, "redux_amd", "csrc"), dst)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    # ---- redux_coder.hpp: the staged store
    p = os.path.join(dst, "redux_coder.hpp")
    s = open(p).read()
    s = sub(s, "    uint64_t acc;  // newest bit at bit 0\n};", "    uint64_t acc;  // newest bit at bit 0\n    uint32_t q0, q1, q2; // A/B: the lane's last three completed dwords\n};")
    s = sub(s, "S.low = 0; S.ihigh = 0; S.pend = 0; S.nb = 0; S.off = off0; S.acc = 0;", "S.low = 0; S.ihigh = 0; S.pend = 0; S.nb = 0; S.off = off0; S.acc = 0; S.q0 = S.q1 = S.q2 = 0;")
    s = sub(s, "constexpr int kSwapped = 1 << 16;", "constexpr int kSwapped = 1 << 16;\nconstexpr int kStaged16 = 1 << 17; // A/B: 16-byte register cell per lane (linear slots)")
    s = sub(s, """        *reinterpret_cast<uint32_t *>(wbase + S.off) = stream_dword<ST>((uint32_t)(S.acc >> (nb & 63u)));
        // in place: as plain C++ the sum lands in a new register and a v_mov merges it after the region""",
            """        const uint32_t w_ = stream_dword<ST>((uint32_t)(S.acc >> (nb & 63u)));
        if (ST & kStaged16) {
            if ((S.off & 12u) == 12u)
                *reinterpret_cast<uint4 *>(wbase + S.off - 12u) = make_uint4(S.q0, S.q1, S.q2, w_);
            S.q0 = S.q1;
            S.q1 = S.q2;
            S.q2 = w_;
        } else
            *reinterpret_cast<uint32_t *>(wbase + S.off) = w_;
        // in place: as plain C++ the sum lands in a new register and a v_mov merges it after the region""")
    open(p, "w").write(s)
    # ---- redux_encode.hpp: linear slots in the pair kernel
    p = os.path.join(dst, "redux_encode.hpp")
    s = open(p).read()
    staged = "--no-stage" not in sys.argv  # (--no-stage: linear slots with the 4-byte stores, the layout's own share of the difference)
    s = sub(s, "constexpr int kPairStride = 256 | kSwapped;", "constexpr int kPairStride = 4 | kSwapped | kStaged16;" if staged else "constexpr int kPairStride = 4 | kSwapped;")
    s = sub(s, """    uint8_t       *wdst  = a.slots + (uint64_t)blockIdx.x * (64 * a.slot_bytes + 128);
    const uint32_t off0  = lane * 4u;
    const uint32_t limit = off0 + (a.slot_cap / 4u) * 256u;""",
            """    uint8_t       *wdst  = a.slots + (uint64_t)blockIdx.x * (64 * a.slot_bytes);
    const uint32_t off0  = lane * (uint32_t)a.slot_bytes;
    const uint32_t limit = off0 + a.slot_cap;""")
    open(p, "w").write(s)
    # ---- redux_coop.hpp keeps the row-major layout of its own
    p = os.path.join(dst, "redux_coop.hpp")
    s = open(p).read()
    s = s.replace("kPairStride", "(256 | kSwapped)")
    open(p, "w").write(s)
    # ---- redux_hip.hip: the pair kernel leaves linear, byte-reversed slots (the small-grid kernels keep mode 3)
    p = os.path.join(dst, "redux_hip.hip")
    s = open(p).read()
    s = sub(s, """    if (which == EncKernel::PairCb32 || which == EncKernel::Pair || which == EncKernel::CoopCb32 || which == EncKernel::Coop)
        HIP_TRY(hipMemsetAsync(ws + g.off_mode, 1 | 2, 4, s));""",
            """    if (which == EncKernel::PairCb32 || which == EncKernel::Pair)
        HIP_TRY(hipMemsetAsync(ws + g.off_mode, 2, 4, s));
    if (which == EncKernel::CoopCb32 || which == EncKernel::Coop)
        HIP_TRY(hipMemsetAsync(ws + g.off_mode, 1 | 2, 4, s));""")
    open(p, "w").write(s)
    out = os.path.join(ROOT, "variants", "zipf_stage16.so" if staged else "zipf_linear4.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = ["hipcc"] + build.FLAGS + ['-DREDUX_SOURCE_HASH="variant:zipf_stage16"', "-o", out, os.path.join(dst, "redux_hip.hip")]
    subprocess.check_call(cmd)
    shutil.rmtree(tmp)
    print(out)


if __name__ == "__main__":
    main()
