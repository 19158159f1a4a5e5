#!/bin/bash
# The benchmark lines on the final library (no profiler): written to gpurun_out/r03m_*.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/r03m_bench_10m_f32.json 2> $O/r03m_bench_10m_f32.err
timeout -k 10 300 python bench.py --docs 1000000 --steps 40 --warmup 10 > $O/r03m_bench_1m_f32.json 2> $O/r03m_bench_1m_f32.err
timeout -k 10 300 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/r03m_bench_10m_bf16.json 2> $O/r03m_bench_10m_bf16.err
timeout -k 10 400 python bench.py --dtype bin1 --steps 20 --warmup 5 > $O/r03m_bench_bin1.json 2> $O/r03m_bench_bin1.err
for K in 100 1000; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 --k $K --sweep 10,1 > $O/r03m_bench_k$K.json 2> $O/r03m_bench_k$K.err
done
for P in 100 10 1; do
  timeout -k 10 300 python bench.py --dtype bf16 --batch-queries 256 --steps 20 --warmup 10 --no-cpu-baseline --sweep= --pass-percent $P > $O/r03m_bench_bf16_batch256_p$P.json 2> $O/r03m_bench_bf16_batch256_p$P.err
done
timeout -k 10 300 python bench.py --batch-queries 64 --steps 10 --warmup 3 --no-cpu-baseline --sweep= > $O/r03m_bench_f32_batch64.json 2> $O/r03m_bench_f32_batch64.err
: > $O/r03m_write_path.jsonl
timeout -k 10 200 python tools/write_path_probe.py >> $O/r03m_write_path.jsonl
VS_MIRROR_WRITE_EACH=1 timeout -k 10 300 python tools/write_path_probe.py >> $O/r03m_write_path.jsonl
cd search_engine_amd
timeout -k 10 900 ./bench_hybrid 10000000 768 100 1 > ../$O/r03m_hybrid_1store.jsonl 2> ../$O/r03m_hybrid.err
timeout -k 10 900 ./bench_hybrid 10000000 768 100 3 > ../$O/r03m_hybrid_3stores.jsonl 2>> ../$O/r03m_hybrid.err
VS_MIRROR_STORE_LOOP=1 timeout -k 10 900 ./bench_hybrid 10000000 768 100 3 > ../$O/r03m_hybrid_3stores_loop.jsonl 2>> ../$O/r03m_hybrid.err
