#!/bin/bash
# where does a bench command hang?  runs it under faulthandler, sends SIGUSR1 after $1 seconds (python traceback to stderr), kills it 10 s later
T=$1; shift
python - "$@" <<'PY' &
import faulthandler, signal, sys, runpy
faulthandler.register(signal.SIGUSR1, all_threads=True)
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
PY
PID=$!
( sleep $T; kill -USR1 $PID 2>/dev/null; sleep 10; kill -9 $PID 2>/dev/null ) &
wait $PID
echo "exit code $?"
