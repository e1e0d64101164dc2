# round 4, second batch: small-call latency after the few-row GEMM / sort draw / host rows, the SA layer table, the single-product bf16
# error on hardware, the new tests (8-rank bench, singular poses), the GEMM crossover
export TMPDIR=/tmp
O=gpurun_out/r4batch2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_bench_multirank_gpu.py tests/test_predicter_gpu.py tests/test_hostprep_gpu.py tests/test_pointnet_gpu.py tests/test_fullsize_properties_gpu.py tests/test_workload_gpu.py tests/test_zz_c1_config_gpu.py -x -q > $O/tests.txt 2>&1; tail -4 $O/tests.txt
timeout 300 python scripts/time_predict_small.py > $O/predict_small.txt 2>&1
for G in 1 256; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr$G -- python scripts/prof_predict_small2.py $G device > $O/tr$G.log 2>&1
find $O/tr$G -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_$G.csv \; ; rm -rf $O/tr$G
done
timeout 300 python scripts/sa_layer_time.py $O/sa_layer.json > $O/sa_layer.txt 2>&1
timeout 300 python scripts/bf16_single_product_hw.py $O/bf16_single_product.json > $O/bf16_single_product.txt 2>&1
for m in 0 100000; do
echo "== CATGRASP_AMD_GEMM_SMALL_TILES=$m" >> $O/gemm_small.txt
CATGRASP_AMD_GEMM_SMALL_TILES=$m timeout 200 python scripts/gemm_small_time.py >> $O/gemm_small.txt 2>&1
done
