export TMPDIR=/tmp
O=gpurun_out/r4small; rm -rf $O; mkdir -p $O
for G in 1 16 256; do
python scripts/prof_predict_small2.py $G device >> $O/wall.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr$G -- python scripts/prof_predict_small2.py $G device > $O/tr$G.log 2>&1
find $O/tr$G -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_$G.csv \; ; rm -rf $O/tr$G
done
python scripts/prof_predict_small.py > $O/cprofile16.txt 2>&1
