// redux.hpp -- C++ host-side mirror of peterbudai/redux's public interface for the block-coding
// path, header-only, on top of the C ABI in include/redux_hip.h (link with libredux_hip.so).
//
// The reference is a Rust crate and this image has no Rust toolchain, so the host side that
// sits above the C ABI is written in C++ with the reference's names, argument meaning and
// error behaviour (file:line under the reference checkout):
//
//   redux::Error { Eof, InvalidInput, IoError }            src/lib.rs:57-64
//   redux::model::Parameters::make(symbol, freq, code)      src/model/mod.rs:63  (Parameters::new)
//   redux::model::Model / AdaptiveTreeModel::make(params)   src/model/mod.rs:17, adaptive_tree.rs:36
//   redux::compress(istream, ostream, model) -> (u64,u64)   src/lib.rs:102
//   redux::decompress(istream, ostream, model)              src/lib.rs:113
//   redux::hip::compress_blocks / decompress_blocks         the block API the GPU path adds
//   redux::hip::compress_blocks_v / decompress_blocks_v     many independent inputs in one launch (tests/corpora.rs:32-85)
//
// Every stream byte is produced by the gfx950 kernels; there is no CPU coder in this header.
#pragma once

#include <cstdint>
#include <istream>
#include <iterator>
#include <memory>
#include <ostream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/redux_hip.h"

namespace redux {

// src/lib.rs:57-96
class Error : public std::runtime_error {
public:
    enum Kind { Eof = REDUX_EOF, InvalidInput = REDUX_INVALID_INPUT, IoError = REDUX_IO_ERROR };
    Error(Kind k, int status, const std::string &what) : std::runtime_error(what), kind_(k), status_(status) {}
    Kind kind() const { return kind_; }
    int  status() const { return status_; } // the raw C-ABI status (4 OutputTooSmall, 5 Unsupported map to IoError)
    static Error from_status(int st)
    {
        switch (st) {
        case REDUX_EOF: return Error(Eof, st, "Unexpected end of file");                               // lib.rs:69
        case REDUX_INVALID_INPUT: return Error(InvalidInput, st, "Invalid data found while processing input"); // lib.rs:70
        case REDUX_OUTPUT_TOO_SMALL: return Error(IoError, st, "I/O error: output buffer too small");
        case REDUX_UNSUPPORTED: return Error(IoError, st, "I/O error: parameters not supported by the device path");
        default: return Error(IoError, st, "I/O error: HIP runtime failure");
        }
    }

private:
    Kind kind_;
    int  status_;
};

inline void check(int st)
{
    if (st != REDUX_OK)
        throw Error::from_status(st);
}

namespace model {

// src/model/mod.rs:32-81
struct Parameters {
    std::size_t   symbol_bits, symbol_eof, symbol_count, freq_bits;
    std::uint64_t freq_max;
    std::size_t   code_bits;
    std::uint64_t code_min, code_one_fourth, code_half, code_three_fourths, code_max;

    // Parameters::new: throws Error(InvalidInput) exactly where the reference returns Err (mod.rs:64)
    static Parameters make(std::size_t symbol, std::size_t frequency, std::size_t code)
    {
        check(redux_params_check((uint32_t)symbol, (uint32_t)frequency, (uint32_t)code));
        Parameters p;
        p.symbol_bits        = symbol;
        p.symbol_eof         = std::size_t(1) << symbol;
        p.symbol_count       = (std::size_t(1) << symbol) + 1;
        p.freq_bits          = frequency;
        p.freq_max           = (std::uint64_t(1) << frequency) - 1;
        p.code_bits          = code;
        p.code_min           = 0;
        p.code_one_fourth    = std::uint64_t(1) << (code - 2);
        p.code_half          = std::uint64_t(2) << (code - 2);
        p.code_three_fourths = std::uint64_t(3) << (code - 2);
        p.code_max           = (std::uint64_t(1) << code) - 1;
        return p;
    }
    redux_params c_abi() const { return redux_params{(uint32_t)symbol_bits, (uint32_t)freq_bits, (uint32_t)code_bits}; }
};

// src/model/mod.rs:17-29.  On this path a Model names WHICH model the device runs; the tree
// itself lives in LDS, one per block, so the trait's per-symbol methods have no host form.
class Model {
public:
    virtual ~Model() {}
    virtual const Parameters &parameters() const = 0;
    virtual bool              device_resident() const { return false; }
};

// src/model/adaptive_tree.rs:36
class AdaptiveTreeModel : public Model {
public:
    static std::unique_ptr<Model> make(const Parameters &p) { return std::unique_ptr<Model>(new AdaptiveTreeModel(p)); }
    const Parameters &parameters() const override { return params_; }
    bool              device_resident() const override { return true; }

private:
    explicit AdaptiveTreeModel(const Parameters &p) : params_(p) {}
    Parameters params_;
};

} // namespace model

namespace hip {

struct Blocks {
    std::vector<std::uint8_t>  data;    // dense concatenation of per-block streams (or decoded blocks)
    std::vector<std::uint64_t> offsets; // nblocks + 1
};

// one redux::compress per block_size bytes, all on the GPU
inline Blocks compress_blocks(const std::uint8_t *in, std::uint64_t len, std::uint32_t block_size,
                              const model::Parameters &p)
{
    const redux_params cp = p.c_abi();
    check(redux_device_supports(&cp));
    if (block_size == 0)
        throw Error::from_status(REDUX_INVALID_INPUT);
    Blocks b;
    const std::uint64_t nb = redux_block_count(len, block_size);
    b.data.resize(redux_encode_bound(&cp, len, block_size));
    b.offsets.resize(nb + 1);
    check(redux_encode_blocks(&cp, in, len, block_size, b.data.data(), b.data.size(), b.offsets.data(), nullptr));
    b.data.resize(b.offsets[nb]);
    return b;
}

// inverse: block b of the result is data[b*block_size .. b*block_size + sizes[b])
inline std::vector<std::uint8_t> decompress_blocks(const Blocks &streams, std::uint32_t block_size,
                                                   const model::Parameters &p, std::vector<std::uint32_t> *sizes = nullptr)
{
    const redux_params cp = p.c_abi();
    check(redux_device_supports(&cp));
    if (streams.offsets.empty() || streams.offsets.back() > streams.data.size())
        throw Error::from_status(REDUX_INVALID_INPUT); // (the C call reads data[offsets[b] .. offsets[b + 1]) from caller memory)
    for (std::size_t i = 1; i < streams.offsets.size(); i++)
        if (streams.offsets[i] < streams.offsets[i - 1])
            throw Error::from_status(REDUX_INVALID_INPUT);
    const std::uint64_t        nb = streams.offsets.size() - 1;
    std::vector<std::uint8_t>  out(nb * (std::uint64_t)block_size);
    std::vector<std::uint32_t> sz(nb);
    check(redux_decode_blocks(&cp, streams.data.data(), streams.offsets.data(), nb, block_size, out.data(), out.size(),
                              sz.data(), nullptr));
    if (sizes)
        *sizes = sz;
    return out;
}

// Many independent inputs in ONE launch (the reference's corpus harness codes file by file, tests/corpora.rs:32-85):
// every input is cut into blocks on its own; first[i] is the number of input i's first block (inputs.size() + 1 entries).
struct BlocksV {
    Blocks                     blocks;
    std::vector<std::uint64_t> first;
};
inline BlocksV compress_blocks_v(const std::vector<std::vector<std::uint8_t>> &inputs, std::uint32_t block_size,
                                 const model::Parameters &p)
{
    const redux_params cp = p.c_abi();
    check(redux_device_supports(&cp));
    if (block_size == 0 || inputs.empty())
        throw Error::from_status(REDUX_INVALID_INPUT);
    std::vector<std::uint64_t> off(inputs.size()), len(inputs.size());
    std::vector<std::uint8_t>  flat;
    BlocksV                    r;
    r.first.assign(inputs.size() + 1, 0);
    for (std::size_t i = 0; i < inputs.size(); i++) {
        off[i] = flat.size();
        len[i] = inputs[i].size();
        flat.insert(flat.end(), inputs[i].begin(), inputs[i].end());
        r.first[i + 1] = r.first[i] + redux_block_count(len[i], block_size);
    }
    const std::uint64_t nb = redux_block_count_v(len.data(), len.size(), block_size);
    r.blocks.data.resize(nb * redux_encode_slot_bytes(&cp, block_size));
    r.blocks.offsets.resize(nb + 1);
    check(redux_encode_blocks_v(&cp, flat.data(), off.data(), len.data(), len.size(), block_size, r.blocks.data.data(),
                                r.blocks.data.size(), r.blocks.offsets.data(), nullptr));
    r.blocks.data.resize(r.blocks.offsets[nb]);
    return r;
}
inline std::vector<std::vector<std::uint8_t>> decompress_blocks_v(const Blocks &streams, const std::vector<std::uint64_t> &lengths,
                                                                  std::uint32_t block_size, const model::Parameters &p)
{
    const redux_params cp = p.c_abi();
    check(redux_device_supports(&cp));
    if (block_size == 0 || lengths.empty() || streams.offsets.empty() ||
        redux_block_count_v(lengths.data(), lengths.size(), block_size) + 1 != streams.offsets.size())
        throw Error::from_status(REDUX_INVALID_INPUT);
    // the C call copies data[offsets.front() .. offsets.back()) from caller memory: inside `data`, in order
    if (streams.offsets.front() != 0 || streams.offsets.back() > streams.data.size())
        throw Error::from_status(REDUX_INVALID_INPUT);
    for (std::size_t i = 1; i < streams.offsets.size(); i++)
        if (streams.offsets[i] < streams.offsets[i - 1])
            throw Error::from_status(REDUX_INVALID_INPUT);
    std::vector<std::uint64_t> off(lengths.size(), 0);
    for (std::size_t i = 1; i < lengths.size(); i++)
        off[i] = off[i - 1] + lengths[i - 1];
    std::vector<std::uint8_t>  out(off.back() + lengths.back() + 1);
    std::vector<std::uint32_t> sz(streams.offsets.size() - 1);
    check(redux_decode_blocks_v(&cp, streams.data.data(), streams.offsets.data(), out.data(), off.data(), lengths.data(),
                                lengths.size(), block_size, sz.data(), nullptr));
    std::vector<std::vector<std::uint8_t>> r(lengths.size());
    for (std::size_t i = 0; i < lengths.size(); i++)
        r[i].assign(out.begin() + off[i], out.begin() + off[i] + lengths[i]);
    return r;
}

} // namespace hip

// src/lib.rs:102-109: returns (bytes in the decompressed stream, bytes in the compressed stream)
inline std::pair<std::uint64_t, std::uint64_t> compress(std::istream &istream, std::ostream &ostream,
                                                        std::unique_ptr<model::Model> model)
{
    if (!model || !model->device_resident())
        throw Error::from_status(REDUX_UNSUPPORTED);
    const redux_params        cp = model->parameters().c_abi();
    std::vector<std::uint8_t> in((std::istreambuf_iterator<char>(istream)), std::istreambuf_iterator<char>());
    std::vector<std::uint8_t> out(redux_encode_bound(&cp, in.size(), in.empty() ? 1u : (std::uint32_t)in.size()));
    std::uint64_t             bi = 0, bo = 0;
    check(redux_compress(&cp, in.data(), in.size(), out.data(), out.size(), &bi, &bo));
    ostream.write(reinterpret_cast<const char *>(out.data()), (std::streamsize)bo);
    if (!ostream)
        throw Error(Error::IoError, REDUX_IO_ERROR, "I/O error: write failed");
    return {bi, bo};
}

// src/lib.rs:113-120
inline std::pair<std::uint64_t, std::uint64_t> decompress(std::istream &istream, std::ostream &ostream,
                                                          std::unique_ptr<model::Model> model,
                                                          std::uint64_t max_output = 0)
{
    if (!model || !model->device_resident())
        throw Error::from_status(REDUX_UNSUPPORTED);
    const redux_params        cp = model->parameters().c_abi();
    std::vector<std::uint8_t> in((std::istreambuf_iterator<char>(istream)), std::istreambuf_iterator<char>());
    if (max_output == 0)
        max_output = in.size() * 64 + (1u << 20);
    std::vector<std::uint8_t> out(max_output);
    std::uint64_t             bi = 0, bo = 0;
    check(redux_decompress(&cp, in.data(), in.size(), out.data(), out.size(), &bi, &bo));
    ostream.write(reinterpret_cast<const char *>(out.data()), (std::streamsize)bo);
    if (!ostream)
        throw Error(Error::IoError, REDUX_IO_ERROR, "I/O error: write failed");
    return {bi, bo};
}

} // namespace redux
