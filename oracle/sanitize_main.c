/* AddressSanitizer / UBSan driver for the CPU oracle (test infrastructure only; see
 * redux_oracle.h).  Runs the hand-traced vectors, a corpus round trip at the three tested
 * widths with both models, the block driver with threads, and the model differential test. */
#include "redux_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) {                                                     \
            fprintf(stderr, "FAILED %s:%d %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                   \
        }                                                               \
    } while (0)

int main(int argc, char **argv)
{
    static const size_t W[3][3] = {{8, 14, 16}, {8, 22, 24}, {8, 30, 32}};
    uint8_t  out[64];
    uint64_t bi, bo;
    CHECK(ox_compress((const uint8_t *)"", 0, out, sizeof out, 8, 30, 32, OX_MODEL_TREE, &bi, &bo) == OX_OK);
    CHECK(bo == 4 && !memcmp(out, "\xff\x00\xff\x00", 4));
    CHECK(ox_compress((const uint8_t *)"a", 1, out, sizeof out, 8, 30, 32, OX_MODEL_TREE, &bi, &bo) == OX_OK);
    CHECK(bo == 5 && !memcmp(out, "\x61\x9d\x64\x97\x0e", 5));
    CHECK(ox_compress((const uint8_t *)"a", 1, out, 2, 8, 30, 32, OX_MODEL_TREE, &bi, &bo) == OX_IO_ERROR);
    if (argc > 1) {
        FILE *f = fopen(argv[1], "rb");
        CHECK(f);
        fseek(f, 0, SEEK_END);
        long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        uint8_t *in = (uint8_t *)malloc((size_t)n + 1), *cmp = (uint8_t *)malloc((size_t)n * 2 + 1024),
                *dec = (uint8_t *)malloc((size_t)n + 1);
        CHECK(fread(in, 1, (size_t)n, f) == (size_t)n);
        fclose(f);
        for (int w = 0; w < 3; w++)
            for (int model = 0; model < 2; model++) {
                size_t len = model == OX_MODEL_LINEAR && n > 60000 ? 60000 : (size_t)n;
                CHECK(ox_compress(in, len, cmp, (size_t)n * 2 + 1024, W[w][0], W[w][1], W[w][2], model, &bi, &bo) == OX_OK);
                CHECK(bi == len);
                uint64_t di, dout;
                CHECK(ox_decompress(cmp, (size_t)bo, dec, (size_t)n + 1, W[w][0], W[w][1], W[w][2], model, &di, &dout) == OX_OK);
                CHECK(dout == len && di == bo && !memcmp(dec, in, len));
                CHECK(ox_decompress(cmp, (size_t)bo / 2, dec, (size_t)n + 1, W[w][0], W[w][1], W[w][2], model, &di, &dout) == OX_EOF);
            }
        uint64_t  nb    = ((uint64_t)n + 65535) / 65536;
        uint64_t  slot  = 65536 + 65536 / 4 + 1024;
        uint8_t  *slots = (uint8_t *)malloc(nb * slot);
        uint32_t *sizes = (uint32_t *)calloc(nb, 4), *osz = (uint32_t *)calloc(nb, 4);
        int32_t  *st    = (int32_t *)calloc(nb, 4);
        uint8_t  *back  = (uint8_t *)malloc(nb * 65536);
        CHECK(ox_compress_blocks(in, (uint64_t)n, 65536, slots, slot, sizes, st, 8, 30, 32, OX_MODEL_TREE, 4) == OX_OK);
        CHECK(ox_decompress_blocks(slots, slot, sizes, nb, back, 65536, osz, st, 8, 30, 32, OX_MODEL_TREE, 4) == OX_OK);
        CHECK(!memcmp(back, in, 65536 < n ? 65536 : (size_t)n));
        free(slots); free(sizes); free(osz); free(st); free(back); free(in); free(cmp); free(dec);
    }
    CHECK(ox_selftest_models(4, 10, 16, 3000, 1, 0, 1) == -1);
    CHECK(ox_selftest_models(8, 10, 16, 3000, 2, 1, 1) == -1);
    puts("sanitize ok");
    return 0;
}
