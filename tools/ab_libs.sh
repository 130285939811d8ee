# usage: bash tools/ab_libs.sh libA.so libB.so  — same-session A/B of two builds of libspicey_hip.so (build/exp/)
for rep in 1 2 3; do for lib in "$@"; do echo "LIB $lib"; SPICEY_HIP_LIB=$lib timeout -k 10 100 python tools/perf_probe.py --configs 512:1:512 --packed 1 --steps 3000 || exit 1; done; done
for rep in 1 2 3; do for lib in "$@"; do echo "LIB $lib"; SPICEY_HIP_LIB=$lib timeout -k 10 100 python tools/perf_probe.py --configs 1:1:1024 --packed 0 --steps 3000 || exit 1; done; done
