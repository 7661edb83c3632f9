#!/bin/bash
# What the GPU box's HOST gives the CPU baseline: cgroup CPU-time grant, and oracle/cpu_bench.py at several thread counts (passive waiting).
# usage: bash tools/cpu_scaling_probe.sh TAG
TAG=${1:-cpuprobe}; O=gpurun_out/$TAG; mkdir -p $O
{ echo "nproc $(nproc)"; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>&1)"; echo "cfs_quota_us: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>&1) period $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>&1)";
  echo "cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>&1 | head -c 200)"; grep -c ^processor /proc/cpuinfo; cat /proc/loadavg; } > $O/host.txt 2>&1
cat $O/host.txt
for th in 8 16 32 64 128 256; do
  OMP_WAIT_POLICY=passive timeout 200 python oracle/cpu_bench.py --seconds 10 --threads $th < /dev/null 2> $O/err_$th.txt | tail -1 > $O/cpu_bench_$th.json
  timeout 20 python -c "import json; d=json.load(open('$O/cpu_bench_$th.json')); print('threads $th:', [(l['actions'], l['threads'], round(l['env_steps_per_s'])) for l in d['legs']], 'speedup %.1f' % d['parallel_speedup_over_one_core'])" < /dev/null
done
