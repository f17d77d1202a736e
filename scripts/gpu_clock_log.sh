# Log of the visible GPU's clocks and socket power (rocm-smi, as fast as it answers: ~3-5 Hz) while a command runs
# (VERDICT r4 item 4a).  usage: bash scripts/gpu_clock_log.sh <logfile> <command...>
LOG=$1; shift
( while true; do
    printf "%s %s\n" "$(date +%s.%N)" "$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power \(W\)' | sed 's/.*: *//' | tr '\n' ' ')"
  done ) > $LOG 2>&1 &
LP=$!
"$@"
kill $LP
