#!/bin/bash
# run_checked.sh <log prefix> <command...>: runs the command with stdout -> <prefix>.out, stderr -> <prefix>.err; exits non-zero when the command
# does OR when the GPU runtime reported a memory access fault (which some runs survive with exit code 0) -- so that `&&` chains stop at the first fault
p=$1; shift
"$@" > "$p.out" 2> "$p.err"
rc=$?
if grep -q "Memory access fault" "$p.err" "$p.out" 2>/dev/null; then echo "[run_checked] GPU memory access fault in: $*"; tail -3 "$p.err"; exit 99; fi
if [ $rc -ne 0 ]; then echo "[run_checked] rc=$rc: $*"; tail -5 "$p.err"; fi
exit $rc
