#!/bin/bash
# Stage times of the eigensolver under a few launch-parameter settings (GPU box; run through gpurun from the repo root).
# usage: sweep_eigh_knobs.sh n "VAR=value" "VAR=value" ...   (the first run is the default setting)
n=$1; shift
python3 scripts/time_eigh.py $n 2>&1 | grep "^n=" | sed "s/^/[default] /"
for kv in "$@"; do
    env $kv python3 scripts/time_eigh.py $n 2>&1 | grep "^n=" | sed "s/^/[$kv] /"
done
