#!/bin/bash
# Samples the GPU's power draw and clocks (rocm-smi) while a command runs: `bash tools/power_probe.sh <logfile> <command...>`.
# Evidence for DESIGN.md's "the conv main loop is power-bound": socket power against the cap, sclk while conv / attention launches run.
LOG=$1; shift
( while true; do rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n'; echo; sleep 0.5; done ) > "$LOG" &
SAMPLER=$!
"$@"
RC=$?
kill $SAMPLER 2>/dev/null
wait $SAMPLER 2>/dev/null
exit $RC
