"""redux_amd -- MI355X-native block coder behind peterbudai/redux's compress/decompress.

Host-side mirror of the reference's public surface (src/lib.rs, src/model/mod.rs) on top of
the C ABI in include/redux_hip.h.  All coding runs in hand-written gfx950 kernels; nothing
here computes a bitstream on the CPU.
"""
from .api import (  # noqa: F401
    Error, Eof, InvalidInput, IoError, OutputTooSmall, Unsupported,
    Parameters, AdaptiveTreeModel,
    compress, decompress, compress_blocks, decompress_blocks, compress_blocks_v, decompress_blocks_v, block_table_v, BLOCK_DTYPE, BLOCK_IDLE,
    host_set_devices, host_chunk_plan, host_set_chunk_bytes,
    DeviceEncoder, DeviceDecoder, DeviceStaticCoder, gen_iid, gen_zipf, zipf_thresholds, version,
)
