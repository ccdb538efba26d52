#!/bin/bash
# developer tool: same-box A/B of library variants and / or environment switches with the ORDER ROTATED from repetition to
# repetition, and means / medians at the end.  (The fixed-order scripts -- ab_env.sh, ab_variants.sh -- have a position
# bias on this pool: the first run after a pause is the slowest, the third of three the fastest, by up to 8 us of a
# 0.67-ms cycle; round 4 found two "wins" and one "loss" that were nothing else.)
# usage: REPS=8 tools/ab_rotate.sh default variants/lib_a.so "MHHIP_X=1" ...     (an argument with '=' is an environment
#        switch for the tree's library, anything else a library; `default` = the tree's library, no switch)
cd /root/repo
A=("$@"); N=${#A[@]}; REPS=${REPS:-6}
LOG=$(mktemp)
for ((rep = 0; rep < REPS; rep++)); do for ((k = 0; k < N; k++)); do
  a=${A[$(( (k + rep) % N ))]}
  unset MHHIP_LIB; envs=""
  case "$a" in default) ;; *=*) envs="$a" ;; *) export MHHIP_LIB=$(realpath $a) ;; esac
  echo -n "$a " | tee -a $LOG
  env $envs python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-fit 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['organic_scene']['ms_per_step'])" | tee -a $LOG
done; done
python3 - $LOG <<'PY'
import sys, collections, statistics as S
st, og = collections.defaultdict(list), collections.defaultdict(list)
for l in open(sys.argv[1]):
    p = l.rsplit(None, 2)
    if len(p) == 3:
        st[p[0]].append(float(p[1])); og[p[0]].append(float(p[2]))
for k in st:
    print('%-40s static mean %.4f median %.4f   organic mean %.4f median %.4f   (%d runs)' % (k, S.mean(st[k]), S.median(st[k]), S.mean(og[k]), S.median(og[k]), len(st[k])))
PY
