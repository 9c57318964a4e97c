REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_adam_bpr -o b -- python $REPO/bench.py --opt adam --steps 128 --warmup 64 --no-cpu-baseline > $OUT/adam_bpr.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_adam_dlrm -o d -- python $REPO/bench.py --model dlrm --opt adam --fp16-mlp --steps 60 --warmup 20 --no-cpu-baseline > $OUT/adam_dlrm.log 2>&1
cd $REPO
tail -1 $OUT/adam_bpr.log | cut -c1-250; tail -1 $OUT/adam_dlrm.log | cut -c 1-100
find $OUT/prof_adam_bpr $OUT/prof_adam_dlrm -name "*kernel_stats.csv" | head
timeout 120 python examples/bpr_synthetic.py 2>&1 | tail -4
