#!/bin/bash
# One diagnostic run of the rocSOLVER path on the GPU box (tools/rocsolver_probe.py).  If the probe is still running after
# $LIMIT seconds its stacks are taken with rocgdb (what was asked for in round 2: the stack of the stuck dlopen) and it is killed.
#   tools/rocsolver_diag.sh OUTDIR [late]
out=${1:-gpurun_out/rocsolver}; mode=$2; LIMIT=${LIMIT:-150}
mkdir -p "$out"
GPF_USE_ROCSOLVER=1 GPF_DEBUG=1 python tools/rocsolver_probe.py $mode > "$out/probe_${mode:-early}.log" 2>&1 &
pid=$!
for i in $(seq 1 "$LIMIT"); do sleep 1; kill -0 $pid 2>/dev/null || break; done
if kill -0 $pid 2>/dev/null; then
    echo "probe still running after $LIMIT s: taking its stacks" >> "$out/probe_${mode:-early}.log"
    timeout -k 5 60 /opt/rocm/bin/rocgdb -batch -p $pid -ex "thread apply all bt 30" > "$out/stuck_stacks_${mode:-early}.txt" 2>&1
    grep -i "rocblas\|rocsolver\|amdhip64\|rocroller" /proc/$pid/maps | awk '{print $6}' | sort -u > "$out/stuck_images_${mode:-early}.txt"
    kill -9 $pid
    wait $pid 2>/dev/null
    echo "killed" >> "$out/probe_${mode:-early}.log"
    exit 1
fi
wait $pid
rc=$?
echo "exit status $rc" >> "$out/probe_${mode:-early}.log"
exit $rc
