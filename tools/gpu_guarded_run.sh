#!/bin/bash
# The way a changed or experimental kernel goes to the GPU: first the GPU-marked extraction and
# matching tests under the kernel host simulation with guard pages behind and in front of every
# device allocation and of dynamic LDS (memory pre-filled with 0xFF), built from the sources as
# they are in the working tree; only when both runs are clean is the command handed to gpurun, and
# the whole exchange (command, verdict, output tail) is kept as gpurun_out/<tag>.gpurun.log.
# A run that faults a GPU can take the machine away from everybody: no kernel variant - timing
# knock-outs included - skips the first step (tools/knockout_check.sh shows what it catches).
# Usage: tools/gpu_guarded_run.sh <tag> [--timeout S] -- '<command for the GPU box>'
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:?tag}; shift
TIMEOUT=900
if [ "${1:-}" = "--timeout" ]; then TIMEOUT=$2; shift 2; fi
[ "${1:-}" = "--" ] && shift
CMD=${1:?command}
cd "$R"
mkdir -p gpurun_out
LOG=gpurun_out/$TAG.gpurun.log
make -C orb_slam3_amd/csrc -j8 > /dev/null || { echo "library build failed"; exit 1; }
make -C tests/hostsim -j8 > /dev/null || { echo "host simulation build failed"; exit 1; }
TESTS=${GUARD_TESTS:-"tests/test_gpu_extract.py tests/test_gpu_match.py tests/test_zz_gpu_fast_passes.py"}
for guard in end start; do
  if ! ORBX_HOSTSIM=1 HOSTSIM_GUARD=$guard HOSTSIM_FILL=0xFF timeout 1500 python -m pytest $TESTS -m gpu -x -q \
       > gpurun_out/$TAG.guard_$guard.log 2>&1; then
    echo "guard run ($guard) failed - nothing sent to the GPU; see gpurun_out/$TAG.guard_$guard.log"
    tail -5 gpurun_out/$TAG.guard_$guard.log
    exit 1
  fi
done
echo "guard runs clean: $(tail -1 gpurun_out/$TAG.guard_end.log)"
{ echo "# $(date -Is) $CMD"; /usr/local/graft/bin/gpurun --timeout "$TIMEOUT" -- "$CMD"; echo "# exit $?"; } 2>&1 | tee "$LOG" | tail -40
