#!/bin/bash
# build, and only if that succeeded run the given command on a GPU box:  tools/brun.sh '<command>'
cd /root/repo || exit 1
out=$(python -c "import __graft_entry__ as g; g.build()" 2>&1)
if ! echo "$out" | grep -q "^built "; then echo "$out" | grep -iE "error|spill|failed" | head -8; echo "BUILD FAILED"; exit 1; fi
/usr/local/graft/bin/gpurun --timeout ${GTIMEOUT:-900} -- "$1" 2>&1 | grep -v "^\[gpurun\] send"
