#!/bin/bash
# Development happens in /tmp/loans_wip (outside the tree: no tool double-counts it): a gpurun call snapshots /root/repo some minutes AFTER it
# was started (when a box is free), so the main tree must not be edited while a call is outstanding.
#   tools/wip.sh pull     main -> /tmp/loans_wip   (start of a piece of work)
#   tools/wip.sh push     /tmp/loans_wip -> main   (right before a gpurun call / a commit; never while a call is in flight)
set -e
cd /root/repo
PATHS="loans_amd tests tools include oracle bench.py train_sheep_localizer.py evaluate.py __graft_entry__.py"
copy() {  # copy <from> <to>: mirror the code paths (no rsync in this image)
  for p in $PATHS; do
    rm -rf "$2/$p"
    cp -a "$1/$p" "$2/$p"
  done
  find "$2" -name __pycache__ -prune -exec rm -rf {} + 2>/dev/null || true
}
case "$1" in
  pull) mkdir -p /tmp/loans_wip; copy . /tmp/loans_wip ;;
  push) if gpurun --status | grep -q '"in_flight": 1'; then echo "a gpurun call is in flight: not pushing"; exit 1; fi
        copy /tmp/loans_wip . ;;
  *) echo "usage: $0 pull|push"; exit 2 ;;
esac
