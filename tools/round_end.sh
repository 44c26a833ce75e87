#!/bin/bash
# Round-end measurement batch (GPU box): parity tests, bench lines, side measurements, rocprofv3 summaries.
# Usage: tools/round_end.sh <tag> [a|b]   -> gpurun_out/final_<tag>/... and gpurun_out/profiles_json/{traffic,issue}.json
# (the whole batch no longer fits one 20-minute gpurun call: part a = tests, profiles, bench lines; part b = side measurements)
TAG=${1:-x}; PART=${2:-ab}; OUT=gpurun_out/final_$TAG; mkdir -p $OUT
if [[ $PART == *a* ]]; then
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/tests_gpu.log 2>&1; tail -2 $OUT/tests_gpu.log
timeout -k 10 600 bash tools/prof.sh $TAG > $OUT/prof.log 2>&1
python3 tools/pmc_summary.py gpurun_out/prof_$TAG k_encode_pair > $OUT/prof_summary.txt 2>&1
python3 tools/profile_json.py --encode gpurun_out/prof_$TAG --workload iid >> $OUT/prof.log 2>&1
timeout -k 10 400 bash tools/prof_decode.sh dec_$TAG > $OUT/prof_decode.txt 2>&1
python3 tools/profile_json.py --decode gpurun_out/prof_dec_$TAG --workload iid >> $OUT/prof.log 2>&1
WORKLOAD=zipf timeout -k 10 400 bash tools/prof_traffic.sh zipf_$TAG > $OUT/prof_traffic_zipf.txt 2>&1
cp $(ls -t gpurun_out/prof_$TAG/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
# the bench lines last: they borrow the figures the passes above just wrote (same library, same source hash)
timeout -k 10 300 python bench.py --steps 10 --warmup 10 > $OUT/bench_iid.log 2>&1 && tail -1 $OUT/bench_iid.log
timeout -k 10 300 python bench.py --steps 10 --warmup 10 --workload zipf --no-cpu-baseline > $OUT/bench_zipf.log 2>&1 && tail -1 $OUT/bench_zipf.log
timeout -k 10 300 python bench.py --workload file > $OUT/bench_file.log 2>&1 && tail -1 $OUT/bench_file.log
fi
if [[ $PART == *b* ]]; then
timeout -k 10 300 python tools/measure_extra.py > $OUT/extra.log 2>&1 && tail -1 $OUT/extra.log
# side measurements of the round: small launches, the other symbol widths, the static model, the corpus as one batch
timeout -k 10 300 python tools/small_grid.py > $OUT/small_grid.txt 2>&1; tail -3 $OUT/small_grid.txt
timeout -k 10 300 bash tools/prof_small.sh $TAG > $OUT/prof_small.txt 2>&1
for nb in 4096 16384 65536; do timeout -k 10 400 python tools/measure_gen.py $nb >> $OUT/gen.txt 2>&1; done; tail -3 $OUT/gen.txt
timeout -k 10 400 python tools/measure_gen.py 65536 1,10,16 2,10,16 4,10,16 6,26,32 7,25,32 9,23,32 10,22,32 12,14,16 >> $OUT/gen.txt 2>&1; tail -1 $OUT/gen.txt
timeout -k 10 300 python tools/measure_wave.py > $OUT/wave.txt 2>&1; tail -2 $OUT/wave.txt
timeout -k 10 300 python tools/measure_blocksize.py 2048 8,30,32 16 64 96 128 256 512 1024 2048 4096 > $OUT/blocksize.txt 2>&1; tail -1 $OUT/blocksize.txt
timeout -k 10 200 python tools/big_stream.py 200000000 > $OUT/big_stream.txt 2>&1; tail -2 $OUT/big_stream.txt
timeout -k 10 300 python tools/soak_cells.py 150 41 > $OUT/soak.txt 2>&1; tail -1 $OUT/soak.txt
timeout -k 10 300 python tools/measure_static.py >> $OUT/static.txt 2>&1; tail -1 $OUT/static.txt
timeout -k 10 500 python tools/corpus_table.py --batch > $OUT/corpus_batch.txt 2>&1
timeout -k 10 500 python tools/corpus_table.py --batch --corpora calgary,canterbury >> $OUT/corpus_batch.txt 2>&1
fi
echo done
