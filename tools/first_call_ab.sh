#!/bin/bash
# A/B of the warm-up's parts on one box: N fresh processes of tools/trace_first_call.py per setting, first and second
# awry_count_batch / awry_locate_batch of each.   usage: tools/first_call_ab.sh <out log> [processes per setting] [settings, e.g. AC]
OUT=$1; N=${2:-4}
: > "$OUT"; : > "$OUT.warmup"
run() {  # label, env assignments...
  local label=$1; shift
  for i in $(seq 1 $N); do
    env "$@" AWRY_TRACE_HOST=1 timeout -k 10 120 python3 tools/trace_first_call.py > /tmp/fc.log 2>&1 || { echo "FAILED $label"; tail -5 /tmp/fc.log; exit 1; }
    grep -h "warm-up, count lane" /tmp/fc.log | sed "s/^/$label | /" >> "$OUT.warmup"
    echo "$label | $(grep -m1 'set_devices' /tmp/fc.log | sed 's/.*set_devices //') | $(grep '^count call [12]:' /tmp/fc.log | sed 's/count call //' | tr '\n' ' ') | $(grep '^locate call [12]:' /tmp/fc.log | sed 's/locate call //; s/ (.*//' | tr '\n' ' ')" >> "$OUT"
  done
}
SETTINGS=${3:-ABCDEF}   # G..K: the bisect settings
[[ $SETTINGS == *A* ]] && run "A default" AWRY_X=0
[[ $SETTINGS == *B* ]] && run "B pinned pool 192 MB (the old set)" AWRY_PINNED_PREWARM_MB=192
[[ $SETTINGS == *C* ]] && run "C no locate warm-up" AWRY_PREWARM_LOCATE=0
[[ $SETTINGS == *D* ]] && run "D small hit buffers" AWRY_PREWARM_HITS=0
[[ $SETTINGS == *E* ]] && run "E = B+C+D (the old warm-up)" AWRY_PINNED_PREWARM_MB=192 AWRY_PREWARM_LOCATE=0 AWRY_PREWARM_HITS=0
[[ $SETTINGS == *F* ]] && run "F no warm-up at all" AWRY_PREWARM=0
# which part of the locate warm-up moves the one-time stall into the first count call (AWRY_PREWARM_LOCATE bit mask)
[[ $SETTINGS == *G* ]] && run "G locate warm-up: reads probe + listed pass only" AWRY_PREWARM_LOCATE=1
[[ $SETTINGS == *H* ]] && run "H locate warm-up: scan + locate pass only" AWRY_PREWARM_LOCATE=2
[[ $SETTINGS == *I* ]] && run "I locate warm-up: copies into pool memory only" AWRY_PREWARM_LOCATE=4
[[ $SETTINGS == *J* ]] && run "J locate warm-up: kernels, no pool copies" AWRY_PREWARM_LOCATE=3

[[ $SETTINGS == *K* ]] && run "K default without the real chunks through the count path at the end" AWRY_PREWARM_REALCOUNT=0
cat "$OUT"; cat "$OUT.warmup"
