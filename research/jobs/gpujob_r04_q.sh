#!/bin/bash
# stream priorities: the frame chain high, the encoder-prefetch / hoist streams low
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r04q; mkdir -p $O
python -c "import torch; print('priority range (least, greatest):', torch.cuda.Stream.priority_range())" 2>/dev/null
for rep in 1 2; do
  for cfg in "default::" "main_hi:-1:" "side_lo::1" "main_hi_side_lo:-1:1" "main_hi_side_0:-1:0"; do
    name=${cfg%%:*}; rest=${cfg#*:}; mp=${rest%%:*}; sp=${rest#*:}
    echo -n "$name: "
    env ${mp:+RMEM_MAIN_PRIORITY=$mp} ${sp:+RMEM_SIDE_PRIORITY=$sp} timeout 600 python bench.py --no-cpu-baseline --no-dropin 2>> $O/err.log | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['mean_us'],1))"
  done
done | tee $O/r04q_stream_priorities.txt
tail -3 $O/err.log
