#!/bin/bash
# round-6 evidence on the final build: rocprofv3 stats + PMC passes + bench lines, sharded world-1 lines (uniform, Zipf with and
# without hot-item replication), the K = 20 call launch by launch, the DLRM step timeline
cd $GRAFT_REPO_ROOT
bash scripts/collect_profiles.sh r6 all > gpurun_out/r6_collect.log 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --gpus 1 --sharded --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6_bench_sharded_world1_local.json
ORX_SHARD_RCCL_SELF=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29551 bench.py --gpus 1 --sharded --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6_bench_sharded_world1_rccl.json
timeout 300 python bench.py --gpus 1 --sharded --zipf 1.05 --hot-items 16384 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6_bench_sharded_world1_zipf_hot16384.json
ORX_SHARD_RCCL_SELF=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --sharded --zipf 1.05 --hot-items 16384 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6_bench_sharded_world1_zipf_hot16384_rccl.json
timeout 300 python bench.py --gpus 1 --sharded --model dlrm --fp16-mlp --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r6_bench_dlrm_sharded_world1_local.json
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r6_tl -o p -- python bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 20 --warmup 5 > /dev/null 2>&1
f=$(find gpurun_out/r6_tl -name '*kernel_trace.csv' | head -1)
python scripts/trace_last_step.py "$f" head_bwd_kernel > gpurun_out/r6_dlrm_fp16_step_timeline.txt
rm -rf gpurun_out/r6_tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r6_k20 -o p -- python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > /dev/null 2>&1
f=$(find gpurun_out/r6_k20 -name '*kernel_trace.csv' | head -1)
python scripts/k20_timeline.py "$f" > gpurun_out/r6_c2_k20_timeline.txt
rm -rf gpurun_out/r6_k20
# the same call without a tracing tool: dispatch-attached events in launch order, the host's way through the call
ORX_PROF_TIMELINE=1 ORX_HOST_TIMING=1 timeout 200 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>&1 >/dev/null | grep "orx host\|orx timeline" > gpurun_out/r6_c2_k20_events.txt
ls gpurun_out | grep "^r6_" | head -80
for f in gpurun_out/r6_c2_bench.json gpurun_out/r6_c2_k20_bench.json gpurun_out/r6_c3_ucml128_censor_bench.json gpurun_out/r6_bpr_adagrad_bench.json gpurun_out/r6_dlrm_fp16_bench.json gpurun_out/r6_bench_sharded_world1_local.json gpurun_out/r6_bench_sharded_world1_zipf_hot16384.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print(sys.argv[1].split('/')[-1], 'ms/step', round(d['ms_per_step'],5), 'frac', round(r.get('frac',0),4), 'kernel_us', r.get('kernel_us'), 'traffic', r.get('traffic'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done
