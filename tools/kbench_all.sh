set -e
for w in cfg1 cfg2 cfg3 cfg3_hw cfg5; do echo "== $w"; timeout -k 10 120 python tools/kbench.py --workload $w; done
