#!/bin/bash
# Round-3 closing evidence on the single-program-kernel build: GPU suite, bench line, rocprofv3 kernel statistics of the same command and
# of config 1, op profile.  Everything lands in gpurun_out/ev3f/ and is copied into profiles/ by hand.
cd $GRAFT_REPO_ROOT
E=gpurun_out/ev3f
mkdir -p $E
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $E/gputests.log; tail -3 $E/gputests.log
timeout 900 python bench.py > $E/r03_bench_n1.json 2> $E/r03_bench_n1.err; head -c 1200 $E/r03_bench_n1.json; echo; tail -2 $E/r03_bench_n1.err
timeout 300 python tools/op_profile2.py 256 > $E/r03_op_profile_wg0.txt 2>&1; tail -2 $E/r03_op_profile_wg0.txt
# the N > 1 code path (RCCL init, sharded_sample, all-gather inside the timed region) on this one GPU
BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu-baseline --no-other-configs > $E/r03_bench_forced_dist.json 2> $E/r03_bench_forced_dist.err; tail -c 900 $E/r03_bench_forced_dist.json; echo; tail -2 $E/r03_bench_forced_dist.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$E/stats -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-other-configs > $R/$E/stats.log 2>&1
f=$(find $R/$E/stats -name "*kernel_stats.csv" | head -1); cp "$f" $R/$E/r03_rocprofv3_kernel_stats.csv; head -4 $R/$E/r03_rocprofv3_kernel_stats.csv | cut -c1-200
rm -rf $R/$E/stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$E/stats1 -- python $R/tools/bench_configs.py cfg1 > $R/$E/stats1.log 2>&1
f=$(find $R/$E/stats1 -name "*kernel_stats.csv" | head -1); cp "$f" $R/$E/r03_cfg1_rocprofv3_kernel_stats.csv; head -3 $R/$E/r03_cfg1_rocprofv3_kernel_stats.csv | cut -c1-200
rm -rf $R/$E/stats1
