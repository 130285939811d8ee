# usage: bash tools/exp_run.sh [exp numbers...]; runs the product library first
timeout -k 10 120 python tools/perf_probe.py --configs 512:1:512 --packed 1 --profile 1 --steps 2000 || exit 1
timeout -k 10 120 python tools/perf_probe.py --configs 512:1:512 --packed 1 --profile 0 --steps 2000 || exit 1
for n in "$@"; do echo "EXP $n"; SPICEY_HIP_LIB=build/exp/libspicey_hip_exp$n.so timeout -k 10 120 python tools/perf_probe.py --configs 512:1:512 --packed 1 --profile 1 --steps 2000 || exit 1; done
