#!/bin/bash
# A/B of a compile-time switch on the GPU box: runs the device-resident legs of bench.py with the library as committed, then
# rebuilds it with -D<FLAG> (hipcc is on the box) and runs them again.  usage: tools/ab_build_flag.sh FLAG[,FLAG2...] [locate reads]
FLAG=$1; READS=${2:-20000000}
cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys
r=json.loads(sys.stdin.read()); v=r['variants']; lo=r['locate']; a=r.get('amino',{})
print('$1: headline %.2f G q/s | present %.2f | present_lf %.2f | unseeded %.2f | walk %.2f G hits/s | dense %.2f | sv count %.2f G reads/s | sv locate %.2f | lf count %.3f | amino %.2f / %.2f' % (
 r['value']/1e9, v['present_queries']['queries_per_s']/1e9, v['present_queries_lf_steps_only']['queries_per_s']/1e9, v['unseeded']['queries_per_s']/1e9,
 lo['sa_ratio_8']['hits_per_s']/1e9, lo['sa_ratio_1']['hits_per_s']/1e9, lo['seed_and_verify']['count_phase_reads_per_s']/1e9, lo['seed_and_verify']['hits_per_s']/1e9,
 lo['count_phase_reads_per_s']/1e9, a.get('random',{}).get('queries_per_s',0)/1e9, a.get('present',{}).get('queries_per_s',0)/1e9))"; }
python bench.py --no-pmc --cpu-seconds 0 --steps 30 --warmup 5 --locate-reads $READS 2>/dev/null | show base
cp awry_amd/build.py /tmp/build.py.orig
for F in ${FLAG//,/ }; do   # several flags, comma separated: one rebuild + run each
  cp /tmp/build.py.orig awry_amd/build.py
  sed -i "s/common = \[\"-O3\"/common = [\"-D$F\", \"-O3\"/" awry_amd/build.py
  python -m awry_amd.build --force > /dev/null 2>&1 || { echo "build with -D$F failed"; exit 1; }
  python bench.py --no-pmc --cpu-seconds 0 --steps 30 --warmup 5 --locate-reads $READS 2>/dev/null | show "$F"
done
cp /tmp/build.py.orig awry_amd/build.py
