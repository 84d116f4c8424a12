#!/bin/bash
# Everything profiles/ holds for a round, in one GPU-box call:  bash tools/round_profiles.sh r02
# (writes under gpurun_out/<tag>_*; copy what is to be kept into profiles/)
set -e
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
python3 bench.py > $O/${TAG}_bench_n65536.json 2> $O/${TAG}_bench.err
echo "bench done"
python3 tools/kernel_bench.py > $O/${TAG}_kernel_bench.jsonl 2> /dev/null
python3 tools/trsv_bench.py 1000 2300 4096 8192 16384 32768 65536 2> /dev/null | grep '^{' > $O/${TAG}_trsv_bench.jsonl
python3 tools/multi_field_bench.py 16384 8 2> /dev/null | grep -v amdgpu > $O/${TAG}_multi_field.txt
python3 tools/multi_field_bench.py 65536 8 2> /dev/null | grep -v amdgpu >> $O/${TAG}_multi_field.txt
python3 tools/world_of_one.py 65536 2 2> /dev/null | grep -v amdgpu > $O/${TAG}_world_of_one.txt
echo "tools done"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_prof_bench $O/${TAG}_prof_trsv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_bench -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-configs > $O/${TAG}_bench_n65536_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_trsv -- python3 $R/tools/trsv_bench.py 8192 65536 > /dev/null 2>&1
find $O/${TAG}_prof_bench -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_bench_n65536_kernel_stats.csv \;
find $O/${TAG}_prof_trsv -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_trsv_kernel_stats.csv \;
echo "rocprof done"
cd $R && bash tools/pmc_bench.sh > $O/${TAG}_pmc.log 2>&1
cp $O/pmc_bench_n65536.json $O/${TAG}_pmc_bench_n65536.json
echo "pmc done"
bash tools/pmc_sq.sh > $O/${TAG}_pmc_sq.txt 2>&1
cp $O/pmc_sq.json $O/${TAG}_pmc_sq.json
echo "pmc sq done"
