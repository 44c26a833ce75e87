// Reads like the reference's own tests: the doc-test of src/lib.rs:23-39, the round-trip
// contract of tests/corpora.rs:32-85 (identity + returned counts == lengths), and the
// Parameters::new error case -- against the C++ mirror in redux_amd/host/redux.hpp.
// With --no-gpu only the host-side parts run (validation, geometry); the rest needs an MI355X.
#include "../../redux_amd/host/redux.hpp"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

using namespace redux;
using namespace redux::model;

#define REQUIRE(c)                                                                     \
    do {                                                                               \
        if (!(c)) {                                                                    \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);        \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

int main(int argc, char **argv)
{
    bool gpu = true;
    const char *file = nullptr;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--no-gpu")) gpu = false; else file = argv[i];
    }
    // model/mod.rs:64
    bool threw = false;
    try { Parameters::make(8, 9, 16); } catch (const Error &e) { threw = e.kind() == Error::InvalidInput; }
    REQUIRE(threw);
    Parameters p = Parameters::make(8, 14, 16);
    REQUIRE(p.symbol_eof == 256 && p.symbol_count == 257 && p.freq_max == 16383 && p.code_half == 32768);
    REQUIRE(redux_block_count(0, 65536) == 1 && redux_block_count(65537, 65536) == 2);
    if (!gpu) { std::puts("host-side checks ok"); return 0; }

    // src/lib.rs:23-39
    const std::string data("\x72\x65\x64\x75\x78", 5);
    std::istringstream cursor1(data);
    std::ostringstream compressed;
    auto c = compress(cursor1, compressed, AdaptiveTreeModel::make(Parameters::make(8, 14, 16)));
    REQUIRE(c.first == 5 && c.second == compressed.str().size());
    REQUIRE(compressed.str() == std::string("\x71\xf2\x34\x84\xc4\xc5\x10", 7)); // tests/golden/kat_streams.json
    std::istringstream cursor2(compressed.str());
    std::ostringstream decompressed;
    auto d = decompress(cursor2, decompressed, AdaptiveTreeModel::make(Parameters::make(8, 14, 16)));
    REQUIRE(decompressed.str() == data && d.first == compressed.str().size() && d.second == 5);

    // tests/corpora.rs:32-85 on one file, block API, three widths
    if (file) {
        std::ifstream f(file, std::ios::binary);
        std::vector<std::uint8_t> in((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        for (int bits : {14, 22, 30}) {
            Parameters q = Parameters::make(8, bits, bits + 2);
            hip::Blocks b = hip::compress_blocks(in.data(), in.size(), 65536, q);
            std::vector<std::uint32_t> sizes;
            std::vector<std::uint8_t> out = hip::decompress_blocks(b, 65536, q, &sizes);
            std::uint64_t total = 0;
            for (std::size_t i = 0; i < sizes.size(); i++) {
                REQUIRE(std::memcmp(out.data() + i * 65536ull, in.data() + total, sizes[i]) == 0);
                total += sizes[i];
            }
            REQUIRE(total == in.size());
            std::printf("bits %d: %zu -> %zu bytes\n", bits, in.size(), b.data.size());
        }
        // the same file three times over, cut at odd places, through the batch call: block for block what the per-input
        // call gives, and back
        Parameters q = Parameters::make(8, 30, 32);
        std::vector<std::vector<std::uint8_t>> parts;
        parts.emplace_back(in.begin(), in.begin() + in.size() / 3);
        parts.emplace_back();
        parts.emplace_back(in.begin() + in.size() / 3, in.end());
        parts.emplace_back(in.begin(), in.begin() + 1);
        hip::BlocksV v = hip::compress_blocks_v(parts, 65536, q);
        REQUIRE(v.first.back() + 1 == v.blocks.offsets.size());
        for (std::size_t i = 0; i < parts.size(); i++) {
            hip::Blocks one = hip::compress_blocks(parts[i].data(), parts[i].size(), 65536, q);
            for (std::size_t b = 0; b + 1 < one.offsets.size(); b++) {
                const std::uint64_t o0 = v.blocks.offsets[v.first[i] + b], o1 = v.blocks.offsets[v.first[i] + b + 1];
                REQUIRE(o1 - o0 == one.offsets[b + 1] - one.offsets[b]);
                REQUIRE(std::memcmp(v.blocks.data.data() + o0, one.data.data() + one.offsets[b], o1 - o0) == 0);
            }
        }
        std::vector<std::uint64_t> lens;
        for (auto &x : parts) lens.push_back(x.size());
        REQUIRE(hip::decompress_blocks_v(v.blocks, lens, 65536, q) == parts);
    }
    // a truncated stream is Error::Eof (bitio/mod.rs:107)
    threw = false;
    try {
        std::istringstream cut(compressed.str().substr(0, 1));
        std::ostringstream sink;
        decompress(cut, sink, AdaptiveTreeModel::make(Parameters::make(8, 14, 16)));
    } catch (const Error &e) { threw = e.kind() == Error::Eof; }
    REQUIRE(threw);
    std::puts("host mirror ok");
    return 0;
}
