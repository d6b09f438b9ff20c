#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04cpu; mkdir -p $O
cat /proc/sys/kernel/yama/ptrace_scope 2>/dev/null
rocgdb -batch -ex "set pagination off" -ex "handle SIGINT stop nopass" -ex "run" -ex "info threads" -ex "thread apply all bt 14" -ex "kill" --args python bench.py --steps 12000 > $O/gdb_run.txt 2>&1 &
GDB=$!
sleep 40
CH=$(ps -o pid= --ppid $GDB | head -1)
echo "gdb $GDB child $CH"
for t in /proc/$CH/task/*; do echo "$(basename $t) $(awk '{print $14+$15}' $t/stat)"; done | sort -k2 -n -r | head -4 | tee $O/gdb_hot_threads.txt
kill -INT $CH
sleep 25
kill $GDB 2>/dev/null
wc -l $O/gdb_run.txt
